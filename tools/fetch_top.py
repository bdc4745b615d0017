"""Developer tool: per-kernel FETCH_SIZE (x2, see traffic_from_pmc.py) of a rocprofv3 --pmc FETCH_SIZE counter_collection.csv, 8 profiled steps assumed."""
import collections, csv, sys
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE":
        agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
print("total fetch per step %.0f MB" % sum(2 * sum(v) / 1024 / 8 for v in agg.values()))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:24]:
    print("%-60s n=%3d avg fetch %.1f MB" % (k, len(v), 2 * sum(v) / len(v) / 1024))
