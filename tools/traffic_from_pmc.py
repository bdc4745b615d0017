"""Developer tool: per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py ->
profiles/<tag>_pmc_traffic.json, keyed by the kernel labels bench.py's `roofline` uses.
usage: traffic_from_pmc.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

Units / corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are KB per dispatch (derived from
TCC_EA0_RDREQ / WRREQ: 128-byte memory-side requests tallied at 64 B); the guide documents the factor 2 for 16 B/lane streaming
reads and calls other widths uncalibrated, so it was CALIBRATED here (tools/micro/fetch_calib.hip, profiles/r02_fetch_size_calibration.txt):
bytes / FETCH_SIZE = 2.000 for fully coalesced streaming reads of 16, 8, 4 AND 2 bytes per lane -- the factor belongs to the
memory-side request size, not to the lane width -- so every kernel's FETCH_SIZE is doubled.  What the doubled figure counts is whole
128-byte lines: a kernel that touches partial lines fetches more than its useful bytes (288-byte row segments at a 1280-byte pitch,
the MFMA stem's patch rows: 1.36x the useful bytes).  WRITE_SIZE is exact for 16-byte stores.  Infinity-Cache hits are counted,
i.e. this is traffic at the L2's memory side, not DRAM-only."""
import collections
import csv
import json
import re
import sys


def label_of(name):
    m = re.search(r"conv_ws_kernelIDF16_Li(\d+)ELi(\d+)ELi(\d+)E", name)
    if m:
        return f"conv_ws_kernel<f16,{m[1]},{m[2]},{m[3]}>"
    m = re.search(r"conv_pwn_kernelILi(\d+)ELi(\d+)E", name) or re.search(r"conv_pwn_kernel<(\d+), (\d+)>", name)
    if m:
        return f"conv_pwn_kernel<f16,{m[1]},{m[2]}>"
    m = re.search(r"conv_pwc_kernelILi(\d+)E", name) or re.search(r"conv_pwc_kernel<(\d+)>", name)
    if m:
        return f"conv_pwc_kernel<f16,{m[1]}>"
    m = re.search(r"dsb_pair_kernelILi(\d+)ELi(\d+)E", name) or re.search(r"dsb_pair_kernel<(\d+), (\d+)", name)
    if m:
        return f"dsb_pair_kernel<{m[1]},{m[2]}>"
    if "wavelet_z_kernel" in name:
        return "wavelet_z_kernel"
    if "block_tile_kernel" in name:
        return "block_tile_kernel"
    if "stem_pair_kernel" in name:
        return "stem_pair_kernel"
    if "pw3b_kernel" in name or "pw3_kernel" in name:
        return "pw3_kernel"
    m = re.search(r"conv_pwr_kernelIDF16_Li(\d+)ELi(\d+)E", name)
    if m:
        return f"conv_pwr_kernel<f16,{m[1]},{m[2]}>"
    m = re.search(r"conv_pw_kernelIDF16_Li(\d+)E", name)
    if m:
        return f"conv_pw_kernel<f16,{m[1]}>"
    m = re.search(r"conv3_tile_kernelIDF16_Li(\d+)ELi(\d+)E", name)
    if m:
        return f"conv3_tile_kernel<f16,{m[1]},{m[2]}>"
    m = re.search(r"conv3s_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name) or re.search(r"conv3s_kernel<(\d+), (\d+), (\d+), (\d+)", name)
    if m:
        return f"conv3s_kernel<{m[1]},{m[2]},{m[4]}>"
    m = re.search(r"conv3p_kernelILi(\d+)E", name) or re.search(r"conv3p_kernel<(\d+)", name)
    if m:
        return f"conv3p_kernel<{m[1]}>"
    m = re.search(r"conv3r_kernelILi(\d+)ELi(\d+)E", name) or re.search(r"conv3r_kernel<(\d+), (\d+)>", name)
    if m:
        return f"conv3r_kernel<{m[1]},{m[2]}>"
    m = re.search(r"conv3_halo_kernelIDF16_Li(\d+)ELi(\d+)E", name)
    if m:
        return f"conv3_halo_kernel<f16,{m[1]},{m[2]}>"
    m = re.search(r"conv_small_kernelIDF16_Li(\d+)ELi(\d+)E", name)
    if m:
        return f"conv_small_kernel<f16,{m[1]},{m[2]}>"
    m = re.search(r"dsconv_strip_kernel(?:ILi|<)(\d+)", name)
    if m:
        return f"dsconv_strip_kernel<{m[1]}>"
    m = re.search(r"dsconv_tz_kernel(?:ILi|<)(\d+)", name)
    if m:
        return f"dsconv_tz_kernel<{m[1]}>"
    m = re.search(r"dsconv_kernelIDF16_Li(\d+)E", name)
    if m:
        return f"dsconv_kernel<{m[1]}>"
    if "stem_mfma" in name:
        return "stem_kernel"
    m = re.search(r"dwconv_kernelIDF16_Li(\d+)E", name)
    if m:
        return f"dwconv_kernel<{m[1]}>"
    if "dwconv3_strip" in name:
        return "dwconv_kernel<3>"
    for k in ("stem_kernel", "dwt_kernel", "head_decode_kernel", "linattn_kernel", "sppf_kernel", "copy_kernel", "softattn_kernel"):
        if k in name:
            return k
    for k in ("nf_select_kernel", "nf_mask_kernel", "nf_resolve_kernel", "scale_img_kernel", "tta_merge_kernel"):
        if k in name:
            return k
    if "nms_select" in name:
        return "nms_select_greedy_kernel"
    if "nms_" in name:
        return "nms(score+sort_greedy)"
    if "block_kernel" in name:
        return "block_kernel"
    if "conv_pw2" in name:
        return "conv_pw2_kernel"
    if "linattn_mfma" in name:
        return "linattn_kernel"
    return None


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        lab = label_of(r["Kernel_Name"])
        if lab:
            agg[lab][0] += 1
            agg[lab][1] += float(r["Counter_Value"])
    return agg


if __name__ == "__main__":
    f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for lab in sorted(set(f) | set(w)):
        nf, kf = f.get(lab, [0, 0.0])
        nw, kw = w.get(lab, [0, 0.0])
        fetch = 2.0 * 1024 * kf / max(nf, 1)
        write = 1024 * kw / max(nw, 1)
        out[lab] = {"launches_profiled": nf, "fetch_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write),
                    "hbm_bytes_per_launch": round(fetch + write), "fetch_kb_raw_avg": round(kf / max(nf, 1), 1)}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --no-pipeline; FETCH_SIZE x 2 = 128-byte lines fetched "
                         "(calibrated for 2/4/8/16 B per lane coalesced reads: profiles/r02_fetch_size_calibration.txt; nf_select / nf_mask / nf_resolve are the three kernels of the `nms_fast(...)` operator row of bench.py's roofline table)",
               "kernels": out}, open(sys.argv[3], "w"), indent=1)
    print(f"{len(out)} kernels -> {sys.argv[3]}")
