"""CPU oracle for the EdgeLine-YOLO detection forward path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.  The shipped package (`edge-yolo_amd/`) never does:
its ops raise if the HIP library is missing.

It is a from-scratch restatement, in plain torch-CPU fp32 functional calls (the reference itself is
a torch program, so ATen CPU conv/softmax/topk/interpolate ARE the reference arithmetic), of the
reference files listed in SURVEY.md §8(c).  Every function cites the reference file:line it follows
(paths relative to /root/reference/ultralytics/).

Pinning: `tests/golden/*.npz` were produced by importing the real reference in the build container
(`tests/golden/make_golden.py`, recipe = SURVEY.md Appendix C) on weights/inputs synthesised by
`synthdata.py` (repo root); `tests/test_oracle_golden.py` checks this oracle against them.
The one boundary that is NOT pinned is `torchvision.ops.nms` (third-party, torchvision==0.17.2,
absent from the reference tree and from this image): `oracle/nms.py::tv_nms` restates its published
CPU algorithm; "parity unpinned" at that boundary.  The surrounding `non_max_suppression` logic is
pinned by goldens produced from the reference's own ops.py with `tv_nms` plugged in.
"""
