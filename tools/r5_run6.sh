set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 700 python tests/fuzz_parity.py 1800 71 gpurun_out/r5_fuzz_1800_per_image_counts.json > gpurun_out/r5_fuzz4.log 2>&1 || { tail -5 gpurun_out/r5_fuzz4.log; exit 1; }
tail -2 gpurun_out/r5_fuzz4.log
