set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r5_t21.log 2>&1 || { tail -40 gpurun_out/r5_t21.log; exit 1; }
tail -3 gpurun_out/r5_t21.log
timeout -k 10 400 python tests/fuzz_parity.py 500 81 gpurun_out/r5_fuzz_500_corrupt.json > gpurun_out/r5_fuzz5.log 2>&1 || { tail -8 gpurun_out/r5_fuzz5.log; exit 1; }
tail -2 gpurun_out/r5_fuzz5.log
