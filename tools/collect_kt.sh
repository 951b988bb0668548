#!/bin/bash
# quick kernel-trace stats of the default bench workload (gpurun): gpurun_out/kt/kernel_stats.csv + top lines
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-kt}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
HWG_BENCH_NO_MINNEC=1 HWG_BENCH_NO_PROF=1 timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -f csv -- python3 bench.py --steps 28 --warmup 7 --no-cpu-baseline --no-gen > $OUT/kt.log 2>&1
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/kt
tail -1 $OUT/kt.log | cut -c1-300
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print("total kernel ms %.1f, launches %d" % (tot/1e6, calls))
for r in sorted(rows, key=lambda r:-float(r["TotalDurationNs"]))[:45]:
    print("%6.2f%% %7d %9.1f us  %s" % (100*float(r["TotalDurationNs"])/tot, int(r["Calls"]), float(r["AverageNs"])/1e3, r["Name"][:110]))
PY
