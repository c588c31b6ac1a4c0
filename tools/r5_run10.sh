set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/probe_cheap_content.py gpurun_out/r5_probe_cheap3.json > gpurun_out/r5_probe_cheap3.log 2>&1 || { tail -20 gpurun_out/r5_probe_cheap3.log; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5_probe_cheap3.json'))
for k in ('sharp','single'):
    for m,v in d[k].items():
        if isinstance(v,dict): print(k, m, v['encdec_mpix_s'], v['enc_ms'], v['dec_ms'], v['bpp_delta_vs_ac_container'], v.get('decode_kernel_ms'), v.get('encode_kernel_ms'))
PY
