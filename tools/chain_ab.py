"""Developer tool: the bench step with / without the C2PSA cv1 -> qkv pointwise chain.  python tools/chain_ab.py"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys; sys.path.insert(0, %r); from edge_yolo_amd.nn.modules import block; block._Chains.pw_chains = %s; import bench; "
        "bench.main(['--steps', '40', '--no-api', '--no-cpu-baseline', '--no-roofline'] + sys.argv[1:])")
for chains in ("('proj_ffn_cv2',)", "('proj_ffn_cv2', 'cv1_qkv')", "('proj_ffn_cv2',)", "('proj_ffn_cv2', 'cv1_qkv')"):
    for extra in ([], ["--no-pipeline"]):
        out = subprocess.run([sys.executable, "-c", code % (root, chains)] + extra, capture_output=True, text=True).stdout
        line = [l for l in out.splitlines() if l.startswith("{")][-1]
        print(chains, "single-graph" if extra else "pipelined", json.loads(line)["ms_per_step"], flush=True)
