set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r5_t27.log 2>&1 || { tail -40 gpurun_out/r5_t27.log; exit 1; }
tail -2 gpurun_out/r5_t27.log
timeout -k 10 300 python tools/probe_cheap_content.py gpurun_out/r5_probe_cheap5.json > gpurun_out/r5_probe_cheap5.log 2>&1 || { tail -20 gpurun_out/r5_probe_cheap5.log; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5_probe_cheap5.json'))
for k in ('sharp','single'):
    for m,v in d[k].items():
        if isinstance(v,dict) and m!='ac': print(k, m, v['encdec_mpix_s'], v['enc_ms'], v['dec_ms'], v.get('encode_kernel_ms',{}).get('rans_encode'), v.get('decode_kernel_ms',{}).get('rans_tail'))
PY
timeout -k 10 300 python tests/fuzz_parity.py 400 111 gpurun_out/r5_fuzz_400_c.json > gpurun_out/r5_fuzz9.log 2>&1 || { tail -8 gpurun_out/r5_fuzz9.log; exit 1; }
tail -1 gpurun_out/r5_fuzz9.log
timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5_bench20.json 2> gpurun_out/r5_bench20.err
python -c "
import json; d=json.loads(open('gpurun_out/r5_bench20.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
