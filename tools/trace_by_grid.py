"""Developer tool: average kernel durations of a rocprofv3 --kernel-trace csv, grouped by (kernel, grid y)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0][-40:]
    gy = next((int(r[k]) for k in r if k.lower() == "grid_size_y"), 1)
    gx = next((int(r[k]) for k in r if k.lower() == "grid_size_x"), 1)
    agg[(n, gx, gy)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    print(k, len(v), "avg %.1f us" % (sum(v) / len(v)))
