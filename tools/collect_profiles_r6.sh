#!/bin/bash
# Round-6 measurement artefacts, run on the GPU box (gpurun); outputs under gpurun_out/r6prof/ (copy what is to be judged into profiles/).
#  1. bench line + rocprofv3 kernel-trace stats of the default workload, PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy) -> per-family table
#  2. bench line + kernel stats of the other BASELINE configs: iam_gan_b1a1_w512, rimes_gan_b4a2_w256_1024, iam_auto_b28_w512
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
STEPS="--steps 28 --warmup 7"
HWG_CONV_DUMP=$OUT/conv_shapes.txt timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_err.log
python tools/prof_summary.py gd $OUT/conv_shapes.txt > $OUT/gd_recomputed.txt 2>&1
timeout 600 python bench.py --steps 140 --warmup 14 --no-cpu-baseline --no-other-workloads > $OUT/bench_line_long.json 2>> $OUT/bench_err.log
HWG_BENCH_NO_MINNEC=1 HWG_CONV_DUMP=$OUT/conv_shapes_under_tracer.txt timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -f csv -- python3 bench.py $STEPS --no-cpu-baseline --no-gen --no-other-workloads > $OUT/kt.log 2>&1
cp $OUT/kt/*kernel_stats.csv $OUT/kernel_stats_b4a2_w512.csv 2>/dev/null || find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_b4a2_w512.csv \;
python tools/launch_census.py $OUT/kernel_stats_b4a2_w512.csv $OUT/kt.log > $OUT/launch_census.txt 2>&1
KT=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
[ -n "$KT" ] && python tools/kernel_by_grid.py $KT 300 > $OUT/kernel_by_grid.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  HWG_BENCH_NO_MINNEC=1 timeout 400 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$tag -o p -- python3 bench.py --steps 7 --warmup 7 --no-cpu-baseline --no-gen --no-other-workloads > $OUT/pmc_$tag.log 2>&1
done
for c in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU"; do
  tag=$(echo $c | cut -d' ' -f1)
  HWG_BENCH_NO_MINNEC=1 timeout 400 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$tag -o p -- python3 bench.py --steps 7 --warmup 7 --no-cpu-baseline --no-gen --no-other-workloads > $OUT/pmc_$tag.log 2>&1
done
python tools/pmc_probe.py $(find $OUT/pmc_SQ_INSTS_LDS $OUT/pmc_SQ_BUSY_CYCLES -name "*.db") > $OUT/sq_counters.txt 2>&1
rm -rf $OUT/pmc_SQ_INSTS_LDS $OUT/pmc_SQ_BUSY_CYCLES
python tools/prof_families.py $OUT/kernel_stats_b4a2_w512.csv $(find $OUT/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $OUT/pmc_WRITE_SIZE -name "*.db" | head -1) $(find $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES -name "*.db" | head -1) $OUT/families.json $OUT/families.txt > $OUT/families.log 2>&1
python tools/pmc_traffic.py $(find $OUT/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $OUT/pmc_WRITE_SIZE -name "*.db" | head -1) > $OUT/pmc_traffic.jsonl 2>&1
python - <<PY
import json
rows = [json.loads(l) for l in open("$OUT/pmc_traffic.jsonl") if l.startswith("{")]
json.dump({"workload": "iam_gan_b4a2_w512",
           "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 7 --warmup 7 --no-cpu-baseline --no-gen",
           "correction": "gfx950: bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports half of wide coalesced reads)",
           "kernels": {r["kernel"]: r for r in rows}}, open("$OUT/pmc_traffic.json", "w"), indent=1)
PY
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES
# per-shape traffic of the conv kernels: the bench run's top shapes replayed one by one under the same two counters
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/shapes/fetch -o sf -- python3 tools/pmc_shapes.py run $OUT/conv_shapes.txt > $OUT/shapes_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/shapes/write -o sw -- python3 tools/pmc_shapes.py run $OUT/conv_shapes.txt > $OUT/shapes_write.log 2>&1
python tools/pmc_shapes.py parse $OUT/conv_shapes.txt $OUT/shapes $OUT/pmc_shapes.json > $OUT/shapes_parse.log 2>&1
rm -rf $OUT/shapes
for wl in iam_gan_b1a1_w512 rimes_gan_b4a2_w256_1024 iam_auto_b28_w512; do
  timeout 400 python bench.py --workload $wl $STEPS --no-cpu-baseline --no-other-workloads > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
  HWG_BENCH_NO_MINNEC=1 timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt_$wl -o kt -f csv -- python3 bench.py --workload $wl --steps 14 --warmup 7 --no-cpu-baseline --no-gen --no-other-workloads > $OUT/kt_$wl.log 2>&1
  find $OUT/kt_$wl -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$wl.csv \;
  rm -rf $OUT/kt_$wl
done
rm -rf $OUT/kt
python tools/norm_bw.py > $OUT/norm_bw.txt 2>&1
for pf in c1 c1fwd onerow; do
  PROBE="$(grep -v "^#" tools/probes/probe_r6_$pf.txt | tr "\n" ";")" timeout 300 python tools/conv_probe.py > $OUT/probe_$pf.txt 2>&1
done
python tools/host_time.py > $OUT/host_time.txt 2>&1
python tools/call_trace.py $OUT/call_trace.txt > /dev/null 2>&1
# parity summary: teacher-forced trainer groups, pre-training trainers, measured gate flips, per-engine forward error
rm -f $OUT/parity_summary.txt
HWG_PARITY_SUMMARY=$OUT/parity_summary.txt timeout 1500 python -m pytest tests/test_trainer_lessons_gpu.py tests/test_pretrain_trainers_gpu.py tests/test_pipeline_gpu.py tests/test_ops_gpu.py tests/test_modules_gpu.py -q -k "teacher_forced or match_reference_per_tensor or forcing or pretrain or reference_trainer or gate_flips or adversarial or full_size or module_parity" > $OUT/parity_tests.log 2>&1
tail -3 $OUT/parity_tests.log
ls -la $OUT | head -40
head -30 $OUT/families.txt
for f in $OUT/bench_*.json; do tail -1 $f | cut -c1-200; done
