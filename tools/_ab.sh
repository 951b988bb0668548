cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for les in 0 1 3; do
rocprofv3 --kernel-trace --stats -d gpurun_out/les$les -o l -f csv -- python3 tools/lesson_profile.py $les 20 > gpurun_out/les$les.log 2>&1
rm -f gpurun_out/les$les/l_kernel_trace.csv
done
