"""GPU idle time per curriculum lesson from a rocprofv3 kernel trace (CSV with Start_Timestamp / End_Timestamp per kernel):
  rocprofv3 --kernel-trace -f csv -d DIR -o kt -- python3 bench.py ...      then      python tools/gpu_idle.py DIR/*_kernel_trace.csv
A step ends with its optimizer launch (mt_adam_kernel); per lesson (step index mod 7, counted back from the last step so that warm-up does
not matter) it prints span, busy time (union of the kernel intervals over all streams), idle = span - busy, launches, and the idle time
split by gap length. The profiler adds host time per launch, so the idle share is an upper bound on the unprofiled run's."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
ends = [i for i, k in enumerate(ks) if "mt_adam_kernel" in k[2]]
steps = []
for a, b in zip(ends[:-1], ends[1:]):
    steps.append(ks[a + 1:b + 1])
steps = steps[-(len(steps) // 7 * 7):]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 7
steps = steps[skip:]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0, [0.0, 0.0, 0.0, 0.0]])
for si, st in enumerate(steps):
    lesson = si % 7
    t0 = st[0][0]; t1 = max(k[1] for k in st)
    busy = 0; cur_s, cur_e = st[0][0], st[0][1]
    gaps = [0.0, 0.0, 0.0, 0.0]
    for s, e, _ in st[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            g = (s - cur_e) / 1e3
            gaps[0 if g < 3 else 1 if g < 10 else 2 if g < 100 else 3] += g
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    a = agg[lesson]
    a[0] += 1; a[1] += (t1 - t0) / 1e6; a[2] += busy / 1e6; a[3] += len(st)
    for i in range(4): a[4][i] += gaps[i] / 1e3
tot = [0.0, 0.0]
print("lesson  steps  span ms  busy ms  idle ms  idle%  launches   idle by gap: <3us  3-10us  10-100us  >100us (ms)")
for l in sorted(agg):
    n, span, busy, launches, gaps = agg[l]
    print("%6d %6d %8.2f %8.2f %8.2f %6.1f %9.0f   %16.2f %7.2f %9.2f %7.2f" % (l, n, span / n, busy / n, (span - busy) / n, 100 * (span - busy) / span, launches / n,
                                                                 *[g / n for g in gaps]))
    tot[0] += span / n; tot[1] += busy / n
print("cycle: span %.2f ms, busy %.2f ms, idle %.1f %%  (per step %.2f / %.2f ms)" % (tot[0], tot[1], 100 * (tot[0] - tot[1]) / tot[0], tot[0] / 7, tot[1] / 7))
