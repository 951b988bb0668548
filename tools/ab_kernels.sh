#!/bin/bash
# A/B of kernel variants on the bench step: for every library given (paths relative to the repo root), the rocprofv3 kernel-trace averages
# of the kernels whose name matches $AB_FILTER.  usage: AB_FILTER="moments|apply" bash tools/ab_kernels.sh libA.so libB.so ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  tag=$(basename $lib .so)
  rm -rf gpurun_out/ab_$tag
  HWG_LIB_OVERRIDE=$PWD/$lib HWG_BENCH_NO_MINNEC=1 timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/ab_$tag -o kt -f csv -- python3 bench.py --steps 14 --warmup 7 --no-cpu-baseline --no-gen > gpurun_out/ab_$tag.log 2>&1
  f=$(find gpurun_out/ab_$tag -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$tag" <<PY
import csv, sys, os, re
rows = list(csv.DictReader(open(sys.argv[1])))
flt = re.compile(os.environ.get("AB_FILTER", "."))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
sel = [r for r in rows if flt.search(r["Name"])]
print("%s: all kernels %.3f ms per step; matching %.3f ms per step" % (sys.argv[2], tot / 21 / 1e6, sum(int(r["TotalDurationNs"]) for r in sel) / 21 / 1e6))
for r in sorted(sel, key=lambda r: -int(r["TotalDurationNs"]))[:12]:
    name = re.sub(r"^void ", "", r["Name"]).replace("(anonymous namespace)::", "").split("(")[0]
    print("   %-44s calls %5s avg %8.2f us" % (name[:44], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf gpurun_out/ab_$tag
done
