"""Per (kernel, grid size) durations of a rocprofv3 kernel trace between the two marker launches (clock probe): which LAUNCHES of a kernel family carry its time.
Usage: python scripts/trace_by_grid.py <kernel_trace.csv> [iterations] [name substring ...]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1
subs = sys.argv[3:]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "clock_probe" in r["Kernel_Name"]]
if len(marks) >= 2:
    rows = rows[marks[-2] + 1:marks[-1]]
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("tedspad::", "").split("(")[0]
    if subs and not any(s in name for s in subs):
        continue
    key = (name[:70], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * max(1, int(r.get("Grid_Size_Y", 1)) // max(1, int(r.get("Workgroup_Size_Y", 1)))))
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in agg.values())
print("%d launches, %.2f ms per iteration" % (sum(a[0] for a in agg.values()) / iters, tot / iters / 1e3))
for (name, wgs), (n, us) in sorted(agg.items(), key=lambda t: -t[1][1])[:60]:
    print("%-72s wgs %7d  x%5.1f  avg %8.1f us  total %8.1f us/iter" % (name, wgs, n / iters, us / n, us / iters))
