"""Developer tool: sum rocprofv3 --pmc counter_collection csv per (kernel, grid) -> mean per launch."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r["Kernel_Name"].split("(")[0][-28:]
    g = (n, r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Dispatch_Id"))
    agg[(n, r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    if len(sys.argv) > 2 and sys.argv[2] not in k[0]: continue
    print(k, {c: round(sum(x) / len(x), 1) for c, x in v.items()}, "n", len(next(iter(v.values()))))
