// likelihood.hpp -- validation likelihood path (LLICTI.forward): float lift and self-information.
// Part of the single translation unit llicti_hip.hip (included in order; not a stand-alone header).
#pragma once

// ------------------------------------------------------------------------------------------------ likelihood path
// LLICTI.forward (LLICTI_nets.py:101-123): the float lift of the training path -- elementwise IEEE fp32 with
// torch.round (half to even) on Co * 255 / 2 (:40-49), then Y - 127/255 (:110) -- and, per band, the mixture
// likelihood of every target pixel (get_self_infos :862-880, :933-935; GaussianConditionalLosslessGMM.forward,
// entropy_layer_nets.py:160-183; _likelihood_fk :117-139) as -log2.
__global__ __launch_bounds__(256) void lift_train_kernel(const uint8_t *__restrict__ rgb, long plane, float *__restrict__ fplanes)
{
    const int b = blockIdx.y;
    const uint8_t *src = rgb + (long)b * 3 * plane;
    float *dst = fplanes + (long)b * 3 * plane;
    const float meanY = (float)(127.0 / 255.0);
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < plane; p += (long)gridDim.x * blockDim.x) {
        const float R = (float)src[p] / 255.0f, G = (float)src[plane + p] / 255.0f, Bl = (float)src[2 * plane + p] / 255.0f;
        const float Co = R - Bl;
        const float t = Bl + __builtin_rintf(Co * 255.0f / 2.0f) / 255.0f;
        const float Cg = G - t;
        const float Y = t + __builtin_rintf(Cg * 255.0f / 2.0f) / 255.0f;
        dst[p] = Y - meanY;
        dst[plane + p] = Co;
        dst[2 * plane + p] = Cg;
    }
}

struct SelfGeom { int B, H, W, lvl, h, w, oi, oj, Hl, Wl; long plane; };

// thread per band-grid position: out [B][3][h][w] (Y, Co, Cg) in bits
__global__ __launch_bounds__(256) void selfinfo_kernel(const float *__restrict__ fplanes, const float *__restrict__ params, SelfGeom s,
                                                       float *__restrict__ out)
{
    const int b = blockIdx.y;
    const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= (long)s.h * s.w) return;
    const int i = (int)(n / s.w), j = (int)(n - (long)i * s.w);
    const ParRow par = par_row(params, b, (long)s.h * s.w, n);
    int rr = 2 * i + s.oi, cc = 2 * j + s.oj;
    if (rr >= s.Hl) rr -= 2;                       // lazyDWT's replicate pad of the odd edge (pad=True geometry)
    if (cc >= s.Wl) cc -= 2;
    const long off = (long)b * 3 * s.plane + ((long)rr << s.lvl) * s.W + ((long)cc << s.lvl);
    float v[3];
    v[0] = fplanes[off]; v[1] = fplanes[off + s.plane]; v[2] = fplanes[off + 2 * s.plane];
    const float half = (float)(0.5 / 255.0);
#pragma unroll
    for (int clr = 0; clr < 3; ++clr) {
        float wv[5], lik[5], wsum = 0.0f;
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            float sg = par[5 * clr + m], mu = par[16 + 5 * clr + m];
            if (clr == 1) { const float t = par[48 + m] * v[0]; mu = mu + t; }
            else if (clr == 2) { const float t1 = par[48 + 5 + m] * v[0]; const float t2 = par[48 + 10 + m] * v[1]; const float t = t1 + t2; mu = mu + t; }
            sg = (sg > kScaleBound) ? sg : kScaleBound;
            const float d = __builtin_fabsf(v[clr] - mu);
            const float up = 0.5f * erfc_spec(kNegRsqrt2 * ((half - d) / sg));
            const float lo = 0.5f * erfc_spec(kNegRsqrt2 * ((-half - d) / sg));
            lik[m] = up - lo;
            const float wk = par[32 + 5 * clr + m];
            wv[m] = (wk > kWeightBound) ? wk : kWeightBound;
            wsum = (m == 0) ? wv[m] : wsum + wv[m];
        }
        float L = 0.0f;
#pragma unroll
        for (int m = 0; m < 5; ++m) { const float t = (wv[m] / wsum) * lik[m]; L = (m == 0) ? t : L + t; }
        if (!(L > 1e-9f)) L = 1e-9f;               // likelihood_lower_bound
        out[(((long)b * 3 + clr) * s.h + i) * s.w + j] = -__builtin_log2f(L);
    }
}
