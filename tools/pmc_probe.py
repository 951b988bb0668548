"""Counters of probe runs (rocprofv3 --kernel-trace --pmc ... -- python3 tools/conv_probe.py ...): per (kernel, grid) the average of every
counter per dispatch and the dispatch duration, over all databases given (one PMC pass each).
usage: pmc_probe.py pass1.db [pass2.db ...]"""
import re, sqlite3, sys


def short(name):
    name = re.sub(r'^void ', '', name).replace('(anonymous namespace)::', '')
    m = re.match(r'([\w:]+(<[^(]*>)?)', name)
    return (m.group(1) if m else name)[:70]


rows = {}
for path in sys.argv[1:]:
    c = sqlite3.connect(path).cursor()
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    kn = "kernel_name" if "kernel_name" in cols else "name"
    dur = "(end - start)" if "end" in cols and "start" in cols else "0"
    q = "select %s, grid_size_x, grid_size_y, grid_size_z, counter_name, count(*), sum(value), sum(%s) from counters_collection group by 1,2,3,4,5" % (kn, dur)
    for name, gx, gy, gz, cname, n, s, d in c.execute(q):
        if "mfma" not in name and "wino" not in name and "wgrad" not in name:
            continue
        r = rows.setdefault((short(name), gx, gy, gz), {})
        r[cname] = s / n
        if d:
            r["us"] = d / n / 1e3
keys = sorted({k for r in rows.values() for k in r})
print("%-58s %-18s " % ("kernel", "grid") + " ".join("%14s" % k[:14] for k in keys))
for (name, gx, gy, gz), r in sorted(rows.items()):
    print("%-58s %-18s " % (name, "%dx%dx%d" % (gx, gy, gz)) + " ".join("%14.4g" % r.get(k, float("nan")) for k in keys))
    if "SQ_WAVE_CYCLES" in r and "SQ_BUSY_CYCLES" in r:
        extra = []
        for a, b in (("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"), ("SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES"), ("SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES"), ("SQ_INST_CYCLES_VMEM", "SQ_WAVE_CYCLES"),
                     ("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"), ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")):
            if a in r and b in r and r[b]:
                extra.append("%s/%s %.3f" % (a, b, r[a] / r[b]))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in r and "GRBM_GUI_ACTIVE" in r:
            extra.append("mfma_busy %.3f" % (r["SQ_VALU_MFMA_BUSY_CYCLES"] / (r["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)))
        print("      " + "; ".join(extra))
