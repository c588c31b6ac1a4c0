set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "agent or bench_line" 2>&1 | tail -3
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda", 0); torch.cuda.set_device(dev); torch.manual_seed(1337)
for rep in range(3):
    r = bench.api_path_mixed_leg(torch, dev, 24, 500, balance=True)
    print("mixed", r["mixed_batches"]["mpix_s"], r["repeats_mpix_s"], r["mixed_batches"]["gpu_dec_ms_per_image"], flush=True)
r = bench.api_path_leg(torch, dev, 24, 512, 768)
print("api_path", r["batched"])
PY
