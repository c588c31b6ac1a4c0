set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r5_bench14.json 2> gpurun_out/r5_bench14.err || { tail -20 gpurun_out/r5_bench14.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/r5_bench14.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); print(json.dumps(d.get('model_drawn'))[:1800]); print(d['api_path_mixed']['mixed_batches'], d['api_path']['batched'])"
