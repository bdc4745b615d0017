"""Summarise a rocprofv3 --kernel-trace CSV of bench.py: per-kernel totals for the LAST step and its top dispatches."""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(f'{d}/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'nms_select' in r['Kernel_Name']]
step = rows[idx[-2] + 1: idx[-1] + 1]
agg = collections.defaultdict(lambda: [0, 0.0])
disp = []
for r in step:
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    name = r['Kernel_Name'].split('(')[0]
    name = name.replace('_Z14conv_ws_kernelIDF16_', 'ws<').replace('_Z17conv_small_kernelIDF16_', 'small<').replace('_Z17conv3_halo_kernelIDF16_', 'halo<').replace('_Z13dsconv_kernelIDF16_', 'dsconv<').replace('EEv5ConvP', '>').replace('EEv3DsP', '>')
    agg[name][0] += 1; agg[name][1] += dur
    disp.append((dur, name, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Grid_Size_Y'], r['Grid_Size_Z']))
tot = sum(v[1] for v in agg.values())
print(f"step: {len(step)} dispatches, {tot:.0f} us of kernel time")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 18]:
    print(f"  {k[:58]:58s} n={v[0]:3d} {v[1]:7.1f} us {100 * v[1] / tot:5.1f}%")
print("top dispatches:", ", ".join(f"{n[:28]}[{g}x{y}x{z}]={d:.0f}" for d, n, g, y, z in sorted(disp, reverse=True)[:14]))
