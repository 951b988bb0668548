"""Launches per training step by kernel, from a rocprofv3 kernel-stats CSV of a bench.py run and that run's JSON line (steps + warm-up steps
run; + the profiled cycles unless HWG_BENCH_NO_PROF was set):  python tools/launch_census.py kernel_stats.csv bench_stdout.log [extra_steps]"""
import csv, json, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
line = [l for l in open(sys.argv[2]) if l.startswith('{"metric"')][-1]
j = json.loads(line)
extra = int(sys.argv[3]) if len(sys.argv) > 3 else (j.get("roofline") or {}).get("profiled_steps", 0)
pre = int(j.get("recorder_warmup_steps") or 0)                       # (round 6: the recorder warm-up and the whole-cycle secondary run in the traced process too)
whole = int((j.get("whole_cycles") or {}).get("steps") or 0)
steps = j["steps"] + j["warmup"] + extra + pre + whole
tot_calls = sum(int(r["Calls"]) for r in rows)
tot_ns = sum(float(r["TotalDurationNs"]) for r in rows)
print("# %d steps in the traced run (%d timed + %d warm-up + %d recorder warm-up + %d whole-cycle secondary + %d profiled): %.1f launches and %.2f ms of kernel time per step"
      % (steps, j["steps"], j["warmup"], pre, whole, extra, tot_calls / steps, tot_ns / steps / 1e6))
print("# launches/step   avg us   us/step   kernel")
for r in sorted(rows, key=lambda r: -int(r["Calls"])):
    c = int(r["Calls"]) / steps
    if c < 0.05:
        continue
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"^void ", "", name).split("(")[0][:80]
    print("%10.2f %10.1f %9.1f   %s" % (c, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / steps / 1e3, name))
