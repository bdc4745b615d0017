"""Developer tool: print the block-program rows (and the step summary) of a bench JSON line."""
import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], "img/s", d["ms_per_step"], "ms/step |", d["config"]["pipeline"][:120])
r = d.get("roofline")
if r:
    for row in r["table"]:
        if "block" in row["kernel"] or "nms" in row["kernel"] or "head_decode" in row["kernel"]:
            print(f'  {row["kernel"]:34s} {row["avg_launch_us"]:9.1f} us  x{row["launches_per_step"]}')
    print("  step:", r["step"])
