set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r5_t26.log 2>&1 || { tail -40 gpurun_out/r5_t26.log; exit 1; }
tail -3 gpurun_out/r5_t26.log
mkdir -p gpurun_out/r5_profiles
CMD="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-pcie-legs --no-ac-leg"
rm -rf gpurun_out/r5_trace3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5_trace3 -o run -- python3 $CMD > gpurun_out/r5_profiles/bench_xrans10_trace_line.json 2> gpurun_out/r5_trace3.err || { tail -20 gpurun_out/r5_trace3.err; exit 1; }
T=$(find gpurun_out/r5_trace3 -name "*kernel_trace.csv" | head -1)
S=$(find gpurun_out/r5_trace3 -name "*kernel_stats.csv" | head -1)
python tools/trace_passes.py $T gpurun_out/r5_profiles/bench_xrans10_cnn_passes.json > /dev/null
python tools/trace_levels.py $T gpurun_out/r5_profiles/bench_xrans10_dispatch_groups.json > /dev/null
cp $S gpurun_out/r5_profiles/bench_xrans10_kernel_stats.csv
python tools/cnn_rocprof.py gpurun_out/r5_profiles/bench_xrans10_kernel_stats.csv gpurun_out/r5_profiles/cnn_rocprof.json "rocprofv3 --kernel-trace --stats -- python3 $CMD" | tail -12
rm -rf gpurun_out/r5_trace3
echo "== PMC bench"
bash tools/collect_pmc.sh gpurun_out/r5_pmc
python tools/pmc_traffic.py gpurun_out/r5_pmc gpurun_out/r5_profiles > /dev/null
echo "== PMC table"
bash tools/collect_pmc.sh gpurun_out/r5_pmc_tab table
mv gpurun_out/r5_profiles/pmc_summary.csv gpurun_out/r5_profiles/pmc_bench_summary.csv
python tools/pmc_traffic.py gpurun_out/r5_pmc_tab gpurun_out/r5_profiles table | tail -5
mv gpurun_out/r5_profiles/pmc_summary.csv gpurun_out/r5_profiles/pmc_table_kernel_summary.csv
rm -rf gpurun_out/r5_pmc gpurun_out/r5_pmc_tab
ls -la gpurun_out/r5_profiles
