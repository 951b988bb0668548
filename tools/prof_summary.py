"""Per-kernel time per step from a rocprofv3 rocpd database (kernel trace of `bench.py --steps K --warmup W`)."""
import sqlite3, re, collections, sys

if len(sys.argv) > 2 and sys.argv[1] == "gd":
    # python tools/prof_summary.py gd <conv_shapes.txt>  - recompute bench.py's roofline figures (dominant family, G / D / G+D conv stack, per
    # network) from the per-shape dump `HWG_CONV_DUMP=<file> python bench.py ...` writes for the profiled cycles of that same run
    import ast
    PEAK = 157.3
    steps, fam, net = None, {}, {}
    for line in open(sys.argv[2]):
        if line.startswith("#"):
            m = re.search(r"\((\d+) steps\)", line)
            steps = int(m.group(1)) if m else steps
            continue
        head, rest = line.split(None, 4)[:4], line.split(None, 4)[4]
        kind, tup = rest.split(" ", 1)
        a, b = tup.rfind(")"), None
        shape = ast.literal_eval(tup[: a + 1])
        tail = tup[a + 1:].split()
        sec, n = float(head[0]) * 1e-3, int(head[1])
        work = float(tail[0]) if tail else float(head[3]) * 1e12 * sec      # (dumps before round 5 carry TFLOP/s to one decimal only)
        f = fam.setdefault(kind, [0.0, 0.0, 0]); f[0] += work; f[1] += sec; f[2] += n
        g = net.setdefault(str(shape[-1]), [0.0, 0.0]); g[1] += sec
        if "reduce" not in kind:
            g[0] += work
    mf = {k: v for k, v in fam.items() if "reduce" not in k and "direct" not in k}
    dom = max(mf, key=lambda k: mf[k][1])
    fl, sec, n = fam[dom]
    print("profiled steps: %s" % steps)
    print("dominant family %s: %.3f TFLOP/s = %.4f of %.1f, %d launches, %.2f us average, %.4f GFLOP per launch" % (dom, fl / sec / 1e12, fl / sec / 1e12 / PEAK, PEAK, n, sec / n * 1e6, fl / n / 1e9))
    for k, (fl_, sec_, n_) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print("   %-28s %8.3f ms  %6d launches  %8.2f us  %s" % (k, sec_ * 1e3, n_, sec_ / n_ * 1e6, ("%.1f GB/s" % (fl_ / sec_ / 1e9)) if "reduce" in k else ("%.1f TFLOP/s" % (fl_ / sec_ / 1e12))))
    G, D = net.get("G", [0.0, 0.0]), net.get("D", [0.0, 0.0])
    gfl, gsec = G[0] + D[0], G[1] + D[1]
    print("G+D conv stack: %.3f TFLOP/s = %.4f of peak; %.1f GFLOP and %.3f ms per step; generator %.3f, discriminator %.3f TFLOP/s" % (
        gfl / gsec / 1e12, gfl / gsec / 1e12 / PEAK, gfl / 1e9 / max(steps or 1, 1), gsec * 1e3 / max(steps or 1, 1), G[0] / G[1] / 1e12, D[0] / D[1] / 1e12))
    for k, v in sorted(net.items(), key=lambda kv: -kv[1][1]):
        print("   %-10s %7.3f ms per step  %6.1f TFLOP/s" % (k, v[1] * 1e3 / max(steps or 1, 1), v[0] / v[1] / 1e12))
    sys.exit(0)
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
