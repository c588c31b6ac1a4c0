// numerics.hpp -- device side of the "LLICTI-MI355X numerics spec v1" (DESIGN.md section 4).
//
// Every fp32 operation, its order and its rounding are fixed by the spec so that the gfx950 kernels,
// run in any launch shape, produce the same 16-bit CDF tables (hence the same bitstreams) as any other
// conforming implementation.  Build with -ffp-contract=off: the only fused operations are the
// explicit __builtin_fmaf calls below; '/' is IEEE division (hipcc default for HIP).
//
// What it restates (reference file:line):
//   entropy_layer_nets.py:197-203   sigma/weight lower bounds, weight normalisation, mixture CDF
//   LLICTI_nets.py:385-392          cross-channel mean update
//   LLICTI_nets.py:941-942          sample grid, ends pushed out by 20 grey levels
//   LLICTI_nets.py:955-983          round(cdf * (65536-(Lp-1))) -> 16-bit wrap -> + index
//   compressai GaussianConditional._standardized_cumulative: 0.5 * erfc(-(2**-0.5) * x)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace llicti {

// polynomial coefficients: tools/gen_numerics_coeffs.py
__device__ __forceinline__ float exp_spec(float y)          // y in [-49, 0]
{
    const float j = __builtin_rintf(y * 0x1.715476p+0f);
    float f = __builtin_fmaf(j, -0x1.62e4p-1f, y);
    f = __builtin_fmaf(j, -0x1.7f7d1cp-20f, f);
    float q = 0x1.6d4328p-10f;
    q = __builtin_fmaf(q, f, 0x1.120b74p-7f);
    q = __builtin_fmaf(q, f, 0x1.5554eap-5f);
    q = __builtin_fmaf(q, f, 0x1.5554dcp-3f);
    q = __builtin_fmaf(q, f, 0x1.0p-1f);
    const float f2 = f * f;
    q = __builtin_fmaf(q, f2, f);
    q = q + 1.0f;
    const int ji = (int)j;
    return __int_as_float(__float_as_int(q) + (ji << 23));
}

// 1.0f / d for d in [2, 9], bit for bit: hipcc expands an IEEE fp32 division into v_div_scale x2, v_rcp, four fma, a mul,
// v_div_fmas and v_div_fixup; on this range both scales are 1, the fix-up is the identity and the mul is by 1.0, so
// the seven operations below ARE that expansion.  Exhaustively checked on the GPU for every float in [2, 9]
// (tools/hipchecks/check_recip.hip: 0 mismatches).  Saves five vector operations per erfc.
__device__ __forceinline__ float recip_2_9(float d)
{
    float y = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    float q = y;
    float r = __builtin_fmaf(-d, q, 1.0f);
    q = __builtin_fmaf(r, y, q);
    r = __builtin_fmaf(-d, q, 1.0f);
    return __builtin_fmaf(r, y, q);
}

__device__ __forceinline__ float erfc_pos_body(float x)     // x >= 0 (the value is used only for x < 7): x + 2 in [2, 9]
{
    const float r = recip_2_9(x + 2.0f);
    const float t = (x - 2.0f) * r;
    float p = 0x1.73901ap-15f;
    p = __builtin_fmaf(p, t, -0x1.d399c2p-18f);
    p = __builtin_fmaf(p, t, -0x1.3ddc9cp-11f);
    p = __builtin_fmaf(p, t, -0x1.a18b0cp-11f);
    p = __builtin_fmaf(p, t, 0x1.93b74ap-9f);
    p = __builtin_fmaf(p, t, 0x1.a7e3c6p-8f);
    p = __builtin_fmaf(p, t, -0x1.606bd2p-6f);
    p = __builtin_fmaf(p, t, -0x1.2a869p-5f);
    p = __builtin_fmaf(p, t, 0x1.1e2deep-2f);
    p = __builtin_fmaf(p, t, -0x1.5fd388p-1f);
    p = __builtin_fmaf(p, t, 0x1.058672p+0f);
    const float s = x * x;
    const float e = __builtin_fmaf(x, x, -s);
    float ex = exp_spec(-s);
    ex = __builtin_fmaf(-e, ex, ex);
    return (p * ex) * r;
}

__device__ __forceinline__ float erfc_pos(float x)          // x >= 0;  := 0 for x >= 7 (and NaN)
{
    if (!(x < 7.0f)) return 0.0f;
    return erfc_pos_body(x);
}

// same bits as erfc_spec, without control flow: the saturated result is selected, not branched to
__device__ __forceinline__ float erfc_spec_nobranch(float x)
{
    const float a = __builtin_fabsf(x);
    const float body = erfc_pos_body(a < 7.0f ? a : 0.0f);    // clamp: keeps exp_spec's exponent arithmetic in range
    const float v = (a < 7.0f) ? body : 0.0f;
    return (x < 0.0f) ? 2.0f - v : v;
}

__device__ __forceinline__ float erfc_spec(float x)
{
    const float v = erfc_pos(__builtin_fabsf(x));
    return (x < 0.0f) ? 2.0f - v : v;
}

constexpr float kScaleBound = (float)(0.11 / 255.0);
constexpr float kWeightBound = 1e-6f;
constexpr float kNegRsqrt2 = (float)(-0.70710678118654752440);

struct Mix { float rsig[5], mu[5], wn[5]; };     // rsig = 1 / max(sigma, bound): one IEEE division per mixture

// The raw CNN outputs in HBM are CHANNEL-PLANAR per (level, band): params[b][ch][pos], ch = 0 .. 63 (4 heads x 16, 15 used), pos
// = i * w + j over the band grid -- so that whatever walks consecutive positions (a decoder wavefront's symbols of a step, a thread
// per position) reads consecutive floats of one channel plane, fully coalesced, and touches only the channels it needs.  ParRow is
// one position of that layout; par[ch] is ONE load at plane stride.  (Round 3 kept position-major rows of 64 floats: a decoder lane
// then gathered single dwords at a 256-byte stride and every colour pass pulled whole rows, 1.6-3.1x the bytes it used.)
//   head 0 sigma | head 1 mu | head 2 weight (each Y, Co, Cg x 5 mixtures) | head 3 a, b, d x 5
// i.e. reference channel o (LLICTI_nets.py:381-387) sits at ch = (o / 15) * 16 + o % 15.
struct ParRow {
    const float *img;       // &params[b][0][0]: wave-uniform wherever a wave works on one image
    long stride;            // floats between channel planes: h * w
    int pos;                // the position (the only per-lane part of an address: img + ch * stride stays on the scalar unit)
    __device__ __forceinline__ float operator[](int ch) const { return (img + (long)ch * stride)[pos]; }
};
__device__ __forceinline__ ParRow par_row(const float *params, int b, long npos, long pos)
{
    return ParRow{ params + (long)b * 64 * npos, npos, (int)pos };
}
// The same position for code that reads it inside a LOOP (the stage decoders' step loops): ONE scalar base (the image's planes) + a 32-bit
// per-lane byte offset -- `global_load_dword v, v_off, s[base]`.  63 planes of the largest band grid (4080 x 4080 positions) are 4.2e9 bytes: inside 32
// bits unsigned.  With ParRow the compiler keeps a 64-bit scalar base PER PLANE across such a loop -- 50 SGPRs it does not have: two v_readlane of a
// spilled pair, a wait state and a 64-bit add in front of every load (219 + 144 spill moves in the lane decoder's disassembly, 103 + 35 with this form;
// its stage launches 1.84 -> 1.80 ms).  One-shot readers (the pairs and table kernels: a thread reads its position once) are better off with ParRow:
// their plane bases are computed once on the scalar unit (cdf_pairs_kernel 0.556 -> 0.566 ms with this form, so it keeps the other).
struct ParRow32 {
    const float *img;       // &params[b][0][0]: wave-uniform
    uint32_t stride4;       // BYTES between channel planes: 4 h w
    uint32_t pos4;          // the position, in bytes
    __device__ __forceinline__ float operator[](int ch) const
    {
        return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(img) + (uint32_t)(pos4 + (uint32_t)ch * stride4));
    }
};
__device__ __forceinline__ ParRow32 par_row32(const float *params, int b, long npos, long pos)
{
    return ParRow32{ params + (long)b * 64 * npos, 4u * (uint32_t)npos, 4u * (uint32_t)pos };
}

// par: a ParRow, or a plain array of the position's 64 values (same indexing)
template <class PAR>
__device__ __forceinline__ void mix_prepare(const PAR &par, int clr, float yv, float cov, Mix &m)
{
    float w[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float sg = par[5 * clr + k];
        float mu = par[16 + 5 * clr + k];
        const float wk = par[32 + 5 * clr + k];
        if (clr == 1) {
            const float t = par[48 + k] * yv;
            mu = mu + t;
        } else if (clr == 2) {
            const float t1 = par[48 + 5 + k] * yv;
            const float t2 = par[48 + 10 + k] * cov;
            const float t = t1 + t2;
            mu = mu + t;
        }
        m.rsig[k] = 1.0f / ((sg > kScaleBound) ? sg : kScaleBound);
        m.mu[k] = mu;
        w[k] = (wk > kWeightBound) ? wk : kWeightBound;
    }
    const float s = (((w[0] + w[1]) + w[2]) + w[3]) + w[4];
    const float den = 1e-9f + s;
#pragma unroll
    for (int k = 0; k < 5; ++k) m.wn[k] = w[k] / den;
}

__device__ __forceinline__ float mix_cdf(const Mix &m, float pt)
{
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float z = (pt - m.mu[k]) * m.rsig[k];
        const float c = 0.5f * erfc_spec_nobranch(kNegRsqrt2 * z);    // same bits as erfc_spec; the five chains interleave
        const float t = m.wn[k] * c;
        acc = (k == 0) ? t : acc + t;
    }
    return acc;
}

// per-stream constants of the sample grid
struct Grid {
    int minv, Lp;
    float p_first, p_last, scale;
};
__device__ __forceinline__ Grid make_grid(int minv, int maxv)
{
    Grid g;
    g.minv = minv;
    g.Lp = maxv - minv + 2;
    g.p_first = (float)(((double)minv - 0.5 - 20.0) / 255.0);   // Python-double scalars in the reference
    g.p_last = (float)(((double)maxv + 0.5 + 20.0) / 255.0);
    g.scale = (float)(65536 - (g.Lp - 1));
    return g;
}
// x / 255.0f for half-integers |x| <= 350 (every regular sample point), in three operations instead of the
// ~10 of an IEEE division: q = x * RN(1/255), one fma residual, one fma correction.  Bit-identical to the
// division on that whole domain (exhaustive check: tests/test_host_cpu.py::test_div255_shortcut_is_exact).
__device__ __forceinline__ float div255_exact(float x)
{
    const float r = 0x1.010102p-8f;
    const float q = x * r;
    const float rem = __builtin_fmaf(-q, 255.0f, x);
    return __builtin_fmaf(rem, r, q);
}
__device__ __forceinline__ float sample_pt(const Grid &g, int i)
{
    if (i == 0) return g.p_first;
    if (i == g.Lp - 1) return g.p_last;
    return div255_exact((float)g.minv - 0.5f + (float)i);
}
__device__ __forceinline__ uint32_t cdf_entry(const Mix &m, const Grid &g, int i)
{
    const float q = __builtin_rintf(mix_cdf(m, sample_pt(g, i)) * g.scale);
    return (uint32_t)((int)q + i) & 0xFFFFu;
}

}  // namespace llicti
