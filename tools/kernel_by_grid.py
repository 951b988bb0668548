"""Per (kernel, grid, workgroup) launch class of a rocprofv3 kernel trace: launches, average / total duration - the in-step duration of the
streaming kernels BY TENSOR SHAPE (their grid encodes it), which the per-kernel stats average away.
  python tools/kernel_by_grid.py <kernel_trace.csv> [min_total_us]"""
import csv, re, sys, collections
rows = csv.DictReader(open(sys.argv[1]))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
acc = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r.get("Kernel_Name") or r.get("Name") or "")
    name = re.sub(r"^void ", "", name).split("(")[0][:60]
    g = (r.get("Grid_Size_X") or r.get("Grid_Size") or "", r.get("Grid_Size_Y") or "", r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "")
    d = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    a = acc[(name,) + g]
    a[0] += 1; a[1] += d
tot = sum(v[1] for v in acc.values())
print("# %d launch classes, %.1f ms of kernel time in the trace; classes with >= %.0f us in total" % (len(acc), tot / 1e3, min_us))
print("# total_us  launches  avg_us  kernel  grid_x grid_y wg_x")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if v[1] < min_us:
        break
    print("%10.1f %7d %8.1f  %-60s %s %s %s" % (v[1], v[0], v[1] / v[0], k[0], k[1], k[2], k[3]))
