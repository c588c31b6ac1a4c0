#!/usr/bin/env python3
"""One 768x512 image (BASELINE.json configs[1]) encoded + decoded N times in one container, for a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d OUT -o si -- python3 tools/single_image_trace.py rans128 50 [tile rows]
then tools/trace_levels.py / tools/trace_gaps.py on the trace.  Prints the wall time per encode and per decode (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llicti_amd.codec import HipCodec, mode_of_name
from llicti_amd.config import default_config
from llicti_amd.graphs.models.LLICTI_nets import LLICTI

name = sys.argv[1] if len(sys.argv) > 1 else "rans128"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # llicti_set_tuning("cnn_tile_rows"): 0 = automatic, 16 / 8 / 4 forced
torch.manual_seed(1337)
codec = HipCodec(torch.device("cuda", 0))
codec.load_state_dict(LLICTI(default_config()).state_dict())
mode = mode_of_name(name)
codec.set_tuning("cnn_tile_rows", rows)
rgb = torch.randint(0, 256, (1, 3, 512, 768), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
for _ in range(5):
    cont, seg = codec.encode(rgb, mode=mode)
    rec = codec.decode(cont, seg, 512, 768, mode=mode)
codec.check()
assert torch.equal(rec, rgb)
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
te = td = 0.0
for _ in range(n):
    e[0].record(); cont, seg = codec.encode(rgb, mode=mode); e[1].record(); rec = codec.decode(cont, seg, 512, 768, mode=mode); e[2].record()
    torch.cuda.synchronize()
    te += e[0].elapsed_time(e[1]); td += e[1].elapsed_time(e[2])
print(f"{name}: encode {te / n:.3f} ms, decode {td / n:.3f} ms, {0.393216 / ((te + td) / n) * 1e3:.1f} MPix/s")
