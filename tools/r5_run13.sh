set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/r5_t24.log 2>&1 || { tail -40 gpurun_out/r5_t24.log; exit 1; }
tail -3 gpurun_out/r5_t24.log
