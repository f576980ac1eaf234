"""Per-kernel averages of every counter in one or more rocprofv3 counter_collection.csv files (one file per --pmc pass).
usage: pmc_table.py <name filter> file.csv [file.csv ...]"""
import collections, csv, sys
flt = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        if flt in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(k)
    for name in sorted(c):
        v = c[name]
        print(f"    {name:32s} launches {len(v):3d}  average per launch {sum(v)/len(v):.6e}")
