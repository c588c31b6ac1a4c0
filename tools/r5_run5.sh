set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/ab_ragged.py gpurun_out/r5_ab_ragged2.json > gpurun_out/r5_ab_ragged2.log 2>&1 || { tail -20 gpurun_out/r5_ab_ragged2.log; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r5_ab_ragged2.json'))
for k,v in d.items():
    if isinstance(v,dict): print(k, v.get('megapixels'), v['enc_ms'], v['dec_ms'], v['encdec_mpix_s'], v['dec_kernel_ms'].get('rans_stage'), v['enc_kernel_ms'].get('rans_encode'))
"
python tools/bench_api_mixed.py 24 500 > gpurun_out/r5_api_mixed2.json 2> gpurun_out/r5_api_mixed2.err || { tail -20 gpurun_out/r5_api_mixed2.err; exit 1; }
cat gpurun_out/r5_api_mixed2.json | head -c 900; echo
python -m pytest tests -m gpu -x -q > gpurun_out/r5_t12.log 2>&1 || { tail -40 gpurun_out/r5_t12.log; exit 1; }
tail -3 gpurun_out/r5_t12.log
python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5_bench5.json 2> gpurun_out/r5_bench5.err
python -c "
import json; d=json.loads(open('gpurun_out/r5_bench5.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
