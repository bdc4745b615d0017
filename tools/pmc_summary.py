"""Developer tool: per-kernel summary of a rocprofv3 --pmc counter_collection.csv (SQ wave-cycle breakdown)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:70] + ' grid=' + r.get('Grid_Size', '')][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    if 'conv' not in k and 'kernel' not in k:
        continue
    m = {c: sum(x) / len(x) for c, x in v.items()}
    w, wc = m['SQ_WAVES'], m['SQ_WAVE_CYCLES']
    print(k)
    print('   waves %d  cycles/wave %.0f  sq_busy/32 %.0f  valu/wave %.0f | wait_any %.0f%% wait_inst %.0f%% (lds %.0f%%) active %.0f%%' % (
        w, 4 * wc / w, m['SQ_BUSY_CYCLES'] / 32, m['SQ_INSTS_VALU'] / w, 100 * m['SQ_WAIT_ANY'] / wc, 100 * m['SQ_WAIT_INST_ANY'] / wc,
        100 * m.get('SQ_WAIT_INST_LDS', 0) / wc, 100 * m['SQ_ACTIVE_INST_ANY'] / wc))
