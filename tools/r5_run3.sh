set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r5_t7.log 2>&1 || { tail -40 gpurun_out/r5_t7.log; exit 1; }
tail -3 gpurun_out/r5_t7.log
timeout -k 10 420 python tests/fuzz_parity.py 700 51 gpurun_out/r5_fuzz_700_all.json > gpurun_out/r5_fuzz1.log 2>&1 || { tail -5 gpurun_out/r5_fuzz1.log; exit 1; }
tail -2 gpurun_out/r5_fuzz1.log
timeout -k 10 300 python tests/fuzz_parity.py 500 52 gpurun_out/r5_fuzz_500_xwide.json xwide > gpurun_out/r5_fuzz2.log 2>&1 || { tail -5 gpurun_out/r5_fuzz2.log; exit 1; }
tail -2 gpurun_out/r5_fuzz2.log
python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5_bench3.json 2> gpurun_out/r5_bench3.err
python -c "
import json; d=json.loads(open('gpurun_out/r5_bench3.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
