set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r5_trace4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5_trace4 -o run -- python3 tools/cheap_step.py xrans10 10 > gpurun_out/r5_cheap_trace_line.txt 2> gpurun_out/r5_trace4.err || { tail -20 gpurun_out/r5_trace4.err; exit 1; }
S=$(find gpurun_out/r5_trace4 -name "*kernel_stats.csv" | head -1)
cp $S gpurun_out/r5_cheap_kernel_stats.csv
rm -rf gpurun_out/r5_trace4
cat gpurun_out/r5_cheap_trace_line.txt; head -12 gpurun_out/r5_cheap_kernel_stats.csv | cut -c1-160
