set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 700 python tests/fuzz_parity.py 1800 101 gpurun_out/r5_fuzz_1800_final.json > gpurun_out/r5_fuzz7.log 2>&1 || { tail -8 gpurun_out/r5_fuzz7.log; exit 1; }
tail -1 gpurun_out/r5_fuzz7.log
timeout -k 10 300 python tests/fuzz_parity.py 600 102 gpurun_out/r5_fuzz_600_xwide_corrupt.json xwide corrupt > gpurun_out/r5_fuzz8.log 2>&1 || { tail -8 gpurun_out/r5_fuzz8.log; exit 1; }
tail -1 gpurun_out/r5_fuzz8.log
