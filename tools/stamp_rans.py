#!/usr/bin/env python3
"""Diagnostic build: rans_decode_stage_kernel with s_memtime stamps (prologue / hint / proof / barrier / update,
cycles per step, printed by workgroup 5 of the level-0 stages).  Writes build/lib_stamp.so; run with
LLICTI_HIP_SO=$PWD/build/lib_stamp.so python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras"""
import os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
work = os.path.join(root, "build", "stamp_src")
shutil.rmtree(work, ignore_errors=True)
shutil.copytree(os.path.join(root, "llicti_amd", "csrc"), work)
main = open(os.path.join(work, "llicti_hip.hip")).read().replace('#include "../../include/llicti_hip.h"', f'#include "{root}/include/llicti_hip.h"')
open(os.path.join(work, "llicti_hip.hip"), "w").write(main)
s = open(os.path.join(work, "rans_coder.hpp")).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b, 1)
rep("    Raw cur = fetch(0);\n    for (int k = 0; k < K; ++k) {\n        const int chunk0 = 64 * (m + k * M);",
    "    Raw cur = fetch(0);\n    unsigned long long T[6] = {0,0,0,0,0,0}; int nprobe = 0;\n    for (int k = 0; k < K; ++k) {\n        unsigned long long s0 = __builtin_amdgcn_s_memtime();\n        const int chunk0 = 64 * (m + k * M);")
rep("                // 1. hint: 5-ary search on the approximate table.  Every lane of the group holds ALL five components in\n", "                unsigned long long s1 = __builtin_amdgcn_s_memtime(); T[0] += s1 - s0;\n")
rep("                // 2. proof with the exact spec arithmetic: entries glo and glo + 1 in one round (independent chains);\n", "                unsigned long long s2 = __builtin_amdgcn_s_memtime(); T[1] += s2 - s1;\n")
rep("                    const uint32_t e = group_cdf_entry(A, B, gr, probe);\n", "                    const uint32_t e = group_cdf_entry(A, B, gr, probe); ++nprobe;\n")
rep("                if (!have_lo) vlo = group_cdf_entry(A, B, gr, 0);\n", "                if (!have_lo) vlo = group_cdf_entry(A, B, gr, 0);\n                unsigned long long s3 = __builtin_amdgcn_s_memtime(); T[2] += s3 - s2;\n")
rep("        __syncthreads();\n        {\n            const bool active = chunk0 + lane < nc && 64 * k + lane < tail_from;\n",
    "        unsigned long long s4 = __builtin_amdgcn_s_memtime();\n        __syncthreads();\n        unsigned long long s5 = __builtin_amdgcn_s_memtime(); T[3] += s5 - s4;\n        {\n            const bool active = chunk0 + lane < nc && 64 * k + lane < tail_from;\n")
rep("        cur = nxt;\n    }\n", "        cur = nxt;\n        unsigned long long s6 = __builtin_amdgcn_s_memtime(); T[4] += s6 - s5; T[5] += s6 - s0;\n    }\n    if (K >= 90 && blockIdx.x == 5 && (lane & 3) == 0) { int mx = nprobe; for (int o = 32; o >= 4; o >>= 1) mx = max(mx, __shfl_xor(mx, o)); if (lane == 0) printf(\"w%d K=%d prol %llu hint %llu proof %llu bar %llu upd %llu total %llu | exact probes/step: lane0 %.2f max-group %.2f\\n\", wave, K, T[0]/K, T[1]/K, T[2]/K, T[3]/K, T[4]/K, T[5]/K, (float)nprobe / K, (float)mx / K); }\n")
open(os.path.join(work, "rans_coder.hpp"), "w").write(s)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-value",
                       "-o", os.path.join(root, "build", "lib_stamp.so"), os.path.join(work, "llicti_hip.hip")])
print("built build/lib_stamp.so")
