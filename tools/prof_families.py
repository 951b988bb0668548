"""Per-kernel-family roofline evidence from rocprofv3 runs of one bench command (tools/collect_profiles_r3.sh):
   kernel-trace stats CSV (time), three PMC databases (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE)
-> for every kernel: launches, average duration, share of GPU time, HBM-side bytes per launch ((2 x FETCH_SIZE + WRITE_SIZE) x 1024, the
   gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-B request of wide reads), achieved GB/s against 8 TB/s, and
   the matrix-core busy fraction SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs).
usage: prof_families.py stats.csv fetch.db write.db mfma.db out.json out.txt"""
import csv, json, re, sqlite3, sys


def short(name):
    name = re.sub(r'^void ', '', name).replace('(anonymous namespace)::', '')
    m = re.match(r'([\w:]+(<[^(]*>)?)', name)
    return (m.group(1) if m else name)[:80]


def counters(path, names):
    out = {}
    try:
        c = sqlite3.connect(path).cursor()
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
        tab = "counters_collection" if "counters_collection" in tabs else next(t for t in tabs if "counter" in t and "collect" in t)
        for kname, cname, n, s in c.execute("select kernel_name, counter_name, count(*), sum(value) from %s group by kernel_name, counter_name" % tab):
            if cname in names:
                d = out.setdefault(short(kname), {})
                d[cname] = d.get(cname, 0.0) + s
                d["n_" + cname] = d.get("n_" + cname, 0) + n
    except Exception as e:      # noqa: BLE001 - a missing pass leaves its columns empty
        print("counters(%s): %r" % (path, e), file=sys.stderr)
    return out


stats_csv, fdb, wdb, mdb, out_json, out_txt = sys.argv[1:7]
fam = {}
for row in csv.DictReader(open(stats_csv)):
    k = short(row["Name"])
    f = fam.setdefault(k, {"calls": 0, "ns": 0.0})
    f["calls"] += int(row["Calls"]); f["ns"] += float(row["TotalDurationNs"])
total_ns = sum(f["ns"] for f in fam.values())
fetch = counters(fdb, ("FETCH_SIZE",)); write = counters(wdb, ("WRITE_SIZE",)); mfma = counters(mdb, ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))
rows = []
for k, f in sorted(fam.items(), key=lambda kv: -kv[1]["ns"]):
    r = {"kernel": k, "calls": f["calls"], "avg_us": f["ns"] / f["calls"] / 1e3, "time_frac": f["ns"] / total_ns}
    fe, wr = fetch.get(k), write.get(k)
    if fe and wr and fe.get("n_FETCH_SIZE") and wr.get("n_WRITE_SIZE"):
        b = (2.0 * fe["FETCH_SIZE"] / fe["n_FETCH_SIZE"] + wr["WRITE_SIZE"] / wr["n_WRITE_SIZE"]) * 1024.0
        r["hbm_bytes_per_launch"] = b
        r["achieved_GBps"] = b / (r["avg_us"] * 1e-6) / 1e9
        r["frac_of_8TBps"] = r["achieved_GBps"] / 8000.0
    m = mfma.get(k)
    if m and m.get("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE comes back as ONE value per dispatch that is the SUM over the 8 XCDs' GRBMs (calibrated on kernels whose issued
        # matrix-core work is known from their FLOP count: the Winograd weight gradient issues 80 TFLOP/s = 0.51 of peak and reads 0.51
        # with this normalisation, 0.064 without): active cycles of the launch = value / 8; 1024 SIMDs each retire at most one MFMA pass
        # per SQ_VALU_MFMA_BUSY cycle
        r["mfma_busy_frac"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    rows.append(r)
json.dump({"total_gpu_ms": total_ns / 1e6, "kernels": rows}, open(out_json, "w"), indent=0)
with open(out_txt, "w") as fh:
    fh.write("# %-62s %7s %9s %6s %12s %9s %7s %9s\n" % ("kernel", "calls", "avg us", "time%", "HBM B/launch", "GB/s", "of 8TB", "MFMA busy"))
    for r in rows[:70]:
        fh.write("%-64s %7d %9.1f %6.2f %12s %9s %7s %9s\n" % (
            r["kernel"], r["calls"], r["avg_us"], 100 * r["time_frac"],
            "%.3e" % r["hbm_bytes_per_launch"] if "hbm_bytes_per_launch" in r else "-",
            "%.0f" % r["achieved_GBps"] if "achieved_GBps" in r else "-",
            "%.3f" % r["frac_of_8TBps"] if "frac_of_8TBps" in r else "-",
            "%.3f" % r["mfma_busy_frac"] if "mfma_busy_frac" in r else "-"))
print(open(out_txt).read()[:3000])
