"""Per-kernel time per step from a rocprofv3 rocpd database (kernel trace of `bench.py --steps K --warmup W`)."""
import sqlite3, re, collections, sys
db = sqlite3.connect(sys.argv[1]); steps = int(sys.argv[2]); warm = int(sys.argv[3]); top = int(sys.argv[4]) if len(sys.argv) > 4 else 45
rows = list(db.cursor().execute("select name, start, end from kernels order by start"))
n = len(rows); i0 = int(n * warm / (warm + steps))
def short(name):
    name = re.sub(r'^void ', '', name).replace('(anonymous namespace)::', '')
    m = re.match(r'([\w:]+(<[^(]*>)?)', name)
    return (m.group(1) if m else name)[:90]
agg = collections.defaultdict(lambda: [0, 0]); tot = 0
for name, s, e in rows[i0:]:
    k = short(name); agg[k][0] += e - s; agg[k][1] += 1; tot += e - s
span = rows[-1][2] - rows[i0][1]
gaps = sum(max(0, rows[i + 1][1] - rows[i][2]) for i in range(i0, n - 1))
print("busy %.2f ms/step, %.0f launches/step, span %.2f ms/step, idle gaps %.2f ms/step" % (tot / steps / 1e6, (n - i0) / steps, span / steps / 1e6, gaps / steps / 1e6))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print('%8.3f ms/step %7.1f /step %8.1f us  %s' % (v[0] / steps / 1e6, v[1] / steps, v[0] / v[1] / 1e3, k))
