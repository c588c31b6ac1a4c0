#!/bin/bash
# The CPU oracle under AddressSanitizer + UBSan (CPU only; GPU sanitizers are not available on the pool): both containers, every
# stream-count family incl. wide streams, and decodes of corrupted streams (which may fail, but must not read or write out of bounds).
# Not collected by pytest (a minute of CPU); run from the repo root:  bash tests/sanitize_oracle.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/llicti_asan
mkdir -p "$OUT"
gcc -O1 -g -fPIC -std=c11 -mfma -mavx2 -ffp-contract=off -fno-fast-math -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer \
    -shared -o "$OUT/liboracle.so" "$ROOT/oracle/llicti_oracle.c" -lm
cat > "$OUT/run.py" <<PY
import sys
sys.path.insert(0, "$ROOT"); sys.path.insert(0, "$ROOT/tests")
import numpy as np
from oracle import oracle as orc
orc._SO = "$OUT/liboracle.so"
orc.build = lambda force=False: orc._SO
from llicti_amd.weights import pack_state_dict
from helpers import make_image
rng = np.random.default_rng(1)
for wname in ("trainedlike", "rand1337"):
    W = orc.Weights(pack_state_dict(dict(np.load(f"$ROOT/tests/golden/weights_{wname}.npz"))))
    for kind, H, Wd in (("smooth", 67, 93), ("noise", 33, 64), ("noise", 96, 130), ("smooth", 32, 32)):
        rgb = make_image(kind, H, Wd, 3)
        assert np.array_equal(orc.decode_image(orc.encode_image(rgb, W), W), rgb)
        for M, wide in ((1, False), (4, False), (10, False), (32, False), (64, False), (128, False), (1, True), (3, True), (10, True), (30, True)):
            bl = orc.encode_image_rans(rgb, W, M, wide)
            assert np.array_equal(orc.decode_image_rans(bl, W), rgb), (kind, M, wide)
            rows = [list(r) for r in bl]                     # one flipped bit in the first non-trivial stream
            hit = next(((r, c) for r in range(1, 6) for c in range(9) if len(rows[r][c]) > 8), None)
            if hit:
                b = bytearray(rows[hit[0]][hit[1]]); b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8)); rows[hit[0]][hit[1]] = bytes(b)
                try:
                    orc.decode_image_rans(rows, W)
                except RuntimeError:
                    pass
        print(wname, kind, H, Wd, "ok", flush=True)
print("sanitizer run clean")
PY
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 python3 "$OUT/run.py"
