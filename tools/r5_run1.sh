set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r5_t4.log 2>&1 || { tail -30 gpurun_out/r5_t4.log; exit 1; }
tail -3 gpurun_out/r5_t4.log
python tools/cnn_gap_probe.py gpurun_out/r5_cnn_gap_probe.json > gpurun_out/r5_cnn_gap_probe.log 2>&1 || tail -20 gpurun_out/r5_cnn_gap_probe.log
rm -rf gpurun_out/r5_trace1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5_trace1 -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-pcie-legs > gpurun_out/r5_trace1_line.json 2> gpurun_out/r5_trace1.err || tail -20 gpurun_out/r5_trace1.err
find gpurun_out/r5_trace1 -name "*.csv" | head
T=$(find gpurun_out/r5_trace1 -name "*kernel_trace.csv" | head -1)
python tools/trace_passes.py $T gpurun_out/r5_trace1_passes.json > gpurun_out/r5_trace1_passes.log 2>&1 || tail gpurun_out/r5_trace1_passes.log
python tools/trace_levels.py $T gpurun_out/r5_trace1_dispatch_groups.json > /dev/null
S=$(find gpurun_out/r5_trace1 -name "*kernel_stats.csv" | head -1)
cp $S gpurun_out/r5_trace1_kernel_stats.csv
rm -rf gpurun_out/r5_trace1     # the raw trace is large
tail -c 1500 gpurun_out/r5_trace1_passes.log
