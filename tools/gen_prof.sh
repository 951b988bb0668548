#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/genprof; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/gen_bench.py 64 48; python tools/gen_bench.py 8 96; python tools/gen_bench.py 256 16
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -f csv -- python3 tools/gen_bench.py 64 48 > $OUT/kt.log 2>&1
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/kt
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print("total kernel ms %.1f, launches %d (56 generate calls)" % (tot/1e6, calls))
for r in sorted(rows, key=lambda r:-float(r["TotalDurationNs"]))[:30]:
    print("%6.2f%% %7d %9.1f us  %s" % (100*float(r["TotalDurationNs"])/tot, int(r["Calls"]), float(r["AverageNs"])/1e3, r["Name"][:110]))
PY
