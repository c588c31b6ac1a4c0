set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 500 python bench.py > gpurun_out/r5_final_line.json 2> gpurun_out/r5_final.err || { tail -20 gpurun_out/r5_final.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/r5_final_line.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_rocprof'], d['roofline']['whole_path_frac'], d['meets_north_star'])
print('api', d['api_path']['batched']['mpix_s'], 'mixed', d['api_path_mixed']['mixed_batches']['mpix_s'], 'model_drawn', d['model_drawn']['xrans10']['encdec_mpix_s'], d['model_drawn']['rans10']['encdec_mpix_s'])
print('single', d['single_image']['in_budget'], 'pcie', d['value_pcie_inclusive'], 'cpu', d['cpu_baseline']['value'])
print([k for k,v in d.items() if isinstance(v,dict) and 'skipped' in v])"
