set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "agent or bench_line" > gpurun_out/r5_t25.log 2>&1 || { tail -40 gpurun_out/r5_t25.log; exit 1; }
tail -2 gpurun_out/r5_t25.log
python - <<'PY'
import json, sys, os, time
sys.path.insert(0, os.getcwd())
import torch, bench
from llicti_amd.graphs.models import LLICTI_nets as LN
dev = torch.device("cuda", 0); torch.cuda.set_device(dev); torch.manual_seed(1337)
orig = LN.LLICTI.note_content
for rep in range(2):
    for name, fn in (("with_note", orig), ("no_note", lambda self, *a: None)):
        LN.LLICTI.note_content = fn
        r = bench.api_path_mixed_leg(torch, dev, 24, 500, balance=True)
        print(name, r["mixed_batches"]["mpix_s"], r["repeats_mpix_s"], r["mixed_batches"]["gpu_dec_ms_per_image"], flush=True)
LN.LLICTI.note_content = orig
r = bench.api_path_leg(torch, dev, 24, 512, 768)
print("api_path", r["batched"])
PY
