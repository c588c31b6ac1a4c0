#!/usr/bin/env python3
"""Diagnostic build: band_params_kernel with s_memtime stamps per phase of a tile (barrier wait / layer 0 /
layers 1+2 / store), cycles per tile, printed by workgroup 0's waves 0 and 8 on launches of >= 64 tiles per
workgroup.  Writes build/lib_stamp_cnn.so; run tools/bench_cnn.py with LLICTI_HIP_SO pointing at it."""
import os, shutil, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
work = os.path.join(root, "build", "stamp_cnn_src")
shutil.rmtree(work, ignore_errors=True)
shutil.copytree(os.path.join(root, "llicti_amd", "csrc"), work)
main = open(os.path.join(work, "llicti_hip.hip")).read().replace('#include "../../include/llicti_hip.h"', f'#include "{root}/include/llicti_hip.h"')
open(os.path.join(work, "llicti_hip.hip"), "w").write(main)
s = open(os.path.join(work, "band_cnn.hpp")).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b, 1)
rep("        __syncthreads();\n", "        const unsigned long long q0 = __builtin_amdgcn_s_memtime();\n        __syncthreads();\n        const unsigned long long q1 = __builtin_amdgcn_s_memtime(); TT[0] += q1 - q0;\n")
rep("        if constexpr (CNN_STAGE_SITES > 1) stage_next(1);\n", "        const unsigned long long q2 = __builtin_amdgcn_s_memtime(); TT[1] += q2 - q1;\n        if constexpr (CNN_STAGE_SITES > 1) stage_next(1);\n")
rep("        // D row 4q + r = output 4q + r of this head; params[pos][head][16]\n", "        const unsigned long long q3 = __builtin_amdgcn_s_memtime(); TT[2] += q3 - q2;\n")
rep("        cur ^= 1;\n", "        cur ^= 1;\n        const unsigned long long q4 = __builtin_amdgcn_s_memtime(); TT[3] += q4 - q3; ++ntile;\n")
rep("    int cur = 0;\n", "    int cur = 0;\n    unsigned long long TT[4] = {0, 0, 0, 0}; int ntile = 0;\n")
# print at the end of the kernel: find the end of the tile loop: the line after 'cur ^= 1;' block closes with '    }\n}'
i = s.index("        const unsigned long long q4")
j = s.index("\n    }\n", i)
s = s[:j + 7] + "    if (ntile >= 64 && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (wave == 0 || wave == 5 || wave == 10 || wave == 15)) printf(\"band %d w%d tiles %d: barrier %llu layer0(+stage0,bias) %llu relu+stage+layers12 %llu store %llu per tile, total %llu\\n\", BAND, wave, ntile, TT[0]/ntile, TT[1]/ntile, TT[2]/ntile, TT[3]/ntile, (TT[0]+TT[1]+TT[2]+TT[3])/ntile);\n" + s[j + 7:]
open(os.path.join(work, "band_cnn.hpp"), "w").write(s)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-value",
                       "-o", os.path.join(root, "build", "lib_stamp_cnn.so"), os.path.join(work, "llicti_hip.hip")])
print("built build/lib_stamp_cnn.so")
