#!/bin/bash
# Round-3 measurement artefacts, run on the GPU box (gpurun); outputs under gpurun_out/r3prof/ (copy what is to be judged into profiles/).
#  1. bench line + rocprofv3 kernel-trace stats of the default workload, PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy) -> per-family table
#  2. bench line + kernel stats of the other BASELINE configs: iam_gan_b1a1_w512, rimes_gan_b4a2_w256_1024, iam_auto_b28_w512
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
STEPS="--steps 28 --warmup 7"
timeout 600 python bench.py --steps 70 --warmup 14 > $OUT/bench_line.json 2> $OUT/bench_err.log
HWG_CONV_DUMP=$OUT/conv_shapes.txt timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -f csv -- python3 bench.py $STEPS --no-cpu-baseline --no-gen > $OUT/kt.log 2>&1
cp $OUT/kt/*kernel_stats.csv $OUT/kernel_stats_b4a2_w512.csv 2>/dev/null || find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_b4a2_w512.csv \;
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$tag -o p -- python3 bench.py --steps 7 --warmup 7 --no-cpu-baseline --no-gen > $OUT/pmc_$tag.log 2>&1
done
python tools/prof_families.py $OUT/kernel_stats_b4a2_w512.csv $(find $OUT/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $OUT/pmc_WRITE_SIZE -name "*.db" | head -1) $(find $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES -name "*.db" | head -1) $OUT/families.json $OUT/families.txt > $OUT/families.log 2>&1
python tools/pmc_traffic.py $(find $OUT/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $OUT/pmc_WRITE_SIZE -name "*.db" | head -1) > $OUT/pmc_traffic.jsonl 2>&1
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES
for wl in iam_gan_b1a1_w512 rimes_gan_b4a2_w256_1024 iam_auto_b28_w512; do
  timeout 400 python bench.py --workload $wl $STEPS --no-cpu-baseline > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt_$wl -o kt -f csv -- python3 bench.py --workload $wl --steps 14 --warmup 7 --no-cpu-baseline --no-gen > $OUT/kt_$wl.log 2>&1
  find $OUT/kt_$wl -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$wl.csv \;
  rm -rf $OUT/kt_$wl
done
rm -rf $OUT/kt
python tools/norm_bw.py > $OUT/norm_bw.txt 2>&1
ls -la $OUT | head -40
head -30 $OUT/families.txt
for f in $OUT/bench_*.json; do tail -1 $f | cut -c1-200; done
