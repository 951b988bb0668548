"""Kernel-by-kernel difference of two launch censuses (tools/launch_census.py outputs): launches / step and us / step, before -> after.
  python tools/census_delta.py profiles/r05_launch_census.txt profiles/r06_launch_census.txt > profiles/r06_census_delta.txt"""
import re, sys
def load(p):
    d, head = {}, ""
    for l in open(p):
        if l.startswith('#'):
            head = head or l.strip()
            continue
        m = re.match(r'\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)', l)
        if m:
            d[m.group(4).strip()] = (float(m.group(1)), float(m.group(2)), float(m.group(3)))
    return d, head
a, ha = load(sys.argv[1]); b, hb = load(sys.argv[2])
print("# before:", sys.argv[1], ha); print("# after: ", sys.argv[2], hb)
print("# totals (kernels above the census's 0.05 launches / step cut): %.1f -> %.1f launches / step, %.2f -> %.2f ms of kernel time / step"
      % (sum(v[0] for v in a.values()), sum(v[0] for v in b.values()), sum(v[2] for v in a.values()) / 1e3, sum(v[2] for v in b.values()) / 1e3))
print("# launches/step before -> after      us/step before -> after   (delta)   kernel")
for k in sorted(set(a) | set(b), key=lambda k: (b.get(k, (0, 0, 0))[2] - a.get(k, (0, 0, 0))[2])):
    x, y = a.get(k, (0, 0, 0)), b.get(k, (0, 0, 0))
    if abs(y[2] - x[2]) < 2.0 and abs(y[0] - x[0]) < 0.3:
        continue
    print("%8.2f -> %6.2f   %9.1f -> %8.1f   (%+7.1f)   %s" % (x[0], y[0], x[2], y[2], y[2] - x[2], k[:90]))
