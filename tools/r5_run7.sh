set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "rans or tail or mixed or poisoned or integrity or degenerate or full_size or agent" > gpurun_out/r5_t20.log 2>&1 || { tail -40 gpurun_out/r5_t20.log; exit 1; }
tail -3 gpurun_out/r5_t20.log
timeout -k 10 300 python tools/probe_cheap_content.py gpurun_out/r5_probe_cheap2.json > gpurun_out/r5_probe_cheap2.log 2>&1 || { tail -20 gpurun_out/r5_probe_cheap2.log; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5_probe_cheap2.json'))
for k in ('sharp','single'):
    for m,v in d[k].items():
        if isinstance(v,dict): print(k, m, v['encdec_mpix_s'], v['enc_ms'], v['dec_ms'], v.get('decode_kernel_ms'))
PY
timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r5_bench7.json 2> gpurun_out/r5_bench7.err
python -c "
import json; d=json.loads(open('gpurun_out/r5_bench7.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
