set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python tests/fuzz_parity.py 2800 61 gpurun_out/r5_fuzz_2800_all.json > gpurun_out/r5_fuzz3.log 2>&1 || { tail -5 gpurun_out/r5_fuzz3.log; exit 1; }
tail -2 gpurun_out/r5_fuzz3.log
