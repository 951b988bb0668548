#!/bin/bash
# copy the judged artefacts of tools/collect_profiles_r6.sh from gpurun_out/r6prof into profiles/r06_*   (run in the build container)
set -eu
cd "$(dirname "$0")/.."
S=gpurun_out/r6prof
grep '^{"metric"' $S/bench_line.json | tail -1 > profiles/r06_bench_line.json
grep '^{"metric"' $S/bench_line_long.json | tail -1 > profiles/r06_bench_line_140steps.json
for wl in iam_gan_b1a1_w512 rimes_gan_b4a2_w256_1024 iam_auto_b28_w512; do
  grep '^{"metric"' $S/bench_$wl.json | tail -1 > profiles/r06_bench_line_$wl.json
  cp $S/kernel_stats_$wl.csv profiles/r06_kernel_stats_$wl.csv
done
cp $S/conv_shapes.txt profiles/r06_conv_shapes_b4a2_w512.txt
cp $S/gd_recomputed.txt profiles/r06_gd_recomputed.txt
cp $S/kernel_stats_b4a2_w512.csv profiles/r06_kernel_stats_b4a2_w512.csv
cp $S/launch_census.txt profiles/r06_launch_census.txt
cp $S/kernel_by_grid.txt profiles/r06_kernel_by_grid.txt
cp $S/pmc_traffic.json profiles/r06_pmc_traffic.json
cp $S/pmc_shapes.json profiles/r06_pmc_shapes.json 2>/dev/null || echo "no pmc_shapes.json"
cp $S/families.json profiles/r06_families.json
cp $S/families.txt profiles/r06_families.txt
cp $S/sq_counters.txt profiles/r06_sq_counters.txt
cp $S/parity_summary.txt profiles/r06_parity_summary.txt
cp $S/host_time.txt profiles/r06_host_time.txt
cp $S/call_trace.txt profiles/r06_call_trace.txt
cp $S/norm_bw.txt profiles/r06_norm_bw.txt
for pf in c1 c1fwd onerow; do cp $S/probe_$pf.txt profiles/r06_probe_$pf.txt; done
python tools/census_delta.py profiles/r05_launch_census.txt profiles/r06_launch_census.txt > profiles/r06_census_delta.txt
ls -la profiles/r06_* | awk '{print $5, $9}'
