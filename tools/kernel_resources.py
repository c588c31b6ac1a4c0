"""Register, spill, scratch and LDS figures of every kernel of the library, from hipcc's own metadata.

`python tools/kernel_resources.py [--json out.json] [extra hipcc flags]` compiles llicti_hip.hip with `-S --cuda-device-only`
(device assembly only: works without a GPU) and prints, per kernel, what the `amdhsa.kernels` metadata says.
tests/test_host_cpu.py::test_no_scratch_in_mfma_and_stage_kernels asserts on it (no spills, no private
segment in any band CNN or stage-decoder instantiation).
"""
from __future__ import annotations

import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "llicti_amd", "csrc", "llicti_hip.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value"]     # llicti_amd/_lib.py HIPCC_FLAGS minus -shared -fPIC
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size")


def device_asm(extra=()):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "llicti.s")
        subprocess.check_call([hipcc] + FLAGS + list(extra) + ["-S", "--cuda-device-only", "-o", out, SRC], stderr=subprocess.DEVNULL)
        with open(out) as f:
            return f.read()


def demangle(names):
    import shutil
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt", path="/opt/rocm/lib/llvm/bin")
    if not filt:
        return list(names)
    out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
    return [re.sub(r"\(.*", "", o).replace("void ", "") for o in out[:len(names)]]


def kernel_resources(extra=()):
    """[{name, demangled, vgpr_count, ...}] for every kernel of the device code object."""
    txt = device_asm(extra)
    md = txt[txt.index("amdhsa.kernels:"):]
    md = md[:md.index("amdhsa.target")] if "amdhsa.target" in md else md
    rows = []
    for blk in re.split(r"\n  - (?=\.)", md)[1:]:
        m = re.search(r"\.name:\s+(\S+)", blk)
        if not m:
            continue
        row = {"name": m.group(1)}
        for f in FIELDS:
            mm = re.search(r"\.%s:\s+(\d+)" % f, blk)
            row[f] = int(mm.group(1)) if mm else 0
        row["uses_dynamic_stack"] = bool(re.search(r"\.uses_dynamic_stack:\s+true", blk))
        rows.append(row)
    for r, d in zip(rows, demangle([r["name"] for r in rows])):
        r["demangled"] = d
    return rows


def main(argv):
    js = None
    if "--json" in argv:
        i = argv.index("--json")
        js = argv[i + 1]
        argv = argv[:i] + argv[i + 2:]
    rows = kernel_resources(argv)
    for r in rows:
        print(f"{r['demangled'][:78]:78s} v={r['vgpr_count']:3d} a={r['agpr_count']:3d} s={r['sgpr_count']:3d} "
              f"vspill={r['vgpr_spill_count']:3d} sspill={r['sgpr_spill_count']:3d} scratch={r['private_segment_fixed_size']:4d} "
              f"lds={r['group_segment_fixed_size']:6d} wg={r['max_flat_workgroup_size']}")
    if js:
        with open(js, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
