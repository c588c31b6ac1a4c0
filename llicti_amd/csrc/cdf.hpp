// cdf.hpp -- mixture CDF kernels: encoder pairs and full uint16 tables (K6-K9).
// Part of the single translation unit llicti_hip.hip (included in order; not a stand-alone header).
#pragma once

// ------------------------------------------------------------------------------------------------ CDF kernels
// image b's stage geometry: its entry of the call's table, or (kernel-level entry points: B images of one size, tight arrays) derived from b
__device__ __forceinline__ StageGeom stage_at(const StageGeom &s, const StageGeom *__restrict__ sv, int b)
{
    if (sv) return sv[b];
    StageGeom r = s;
    r.img_off = (long)b * 3 * s.plane;
    r.par_off = (long)b * LLICTI_PARAM_STRIDE * s.h * s.w;
    r.pair_off = (long)b * s.hc * s.wc;
    return r;
}

__device__ __forceinline__ void clr_range(const int32_t *mm, int clr, int &minv, int &maxv, int &shift)
{
    // LLICTI_nets.py:394-395, :544-547: Y uses the fixed range [-127,128], Co/Cg the image's own [min,max]
    if (clr == 0) { minv = -127; maxv = 128; shift = 127; }
    else { minv = mm[clr - 1]; maxv = mm[2 + clr - 1]; shift = -minv; }
}

// encoder: thread per coded position; the two entries the coder reads, for Y, Co, Cg
constexpr int kPairsThreads = 256;       // a workgroup walks 256 consecutive positions (1 KB of every channel plane); 64 ... 1024 measured: 0.61 / 0.62 / 0.60 / 0.64 / 0.69 ms per encode
__device__ __forceinline__ void cdf_pairs_body(const int16_t *__restrict__ planes, const float *__restrict__ params,
                                               const int32_t *__restrict__ minmax, const StageGeom &s, uint32_t *__restrict__ pairs)
{
    // Channel-planar CNN outputs (numerics.hpp: ParRow): the 64 lanes of a wave walk 64 consecutive positions, so each of a thread's
    // 60 parameter loads is one fully coalesced 256-byte wave access -- no LDS staging (round 3 moved position-major rows through LDS).
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int np = s.h * s.w;
    const int p0 = (blockIdx.x * (blockDim.x >> 6) + wave) * 64;
    if (p0 >= np) return;                                         // whole wave (a smaller image of a mixed batch: whole workgroups)
    const ParRow par = par_row(params + s.par_off, 0, np, min(p0 + lane, np - 1));
    const int p = p0 + lane;
    const int i = p / s.w, j = p - i * s.w;
    if (p >= np || i >= s.hc || j >= s.wc) return;                // padded row / column of the band grid: not coded
    const long n = (long)i * s.wc + j;
    const long off = s.img_off + ((long)(2 * i + s.oi) << s.lvl) * s.W + ((long)(2 * j + s.oj) << s.lvl);
    const int vy = planes[off], vco = planes[off + s.plane], vcg = planes[off + 2 * s.plane];
    const float yv = div255_exact((float)vy), cov = div255_exact((float)vco);      // (integers in [-255, 255]: bit-identical to the division, numerics.hpp)
    const int32_t *mm = minmax + 4 * b;
#pragma unroll
    for (int clr = 0; clr < 3; ++clr) {
        int minv, maxv, shift;
        clr_range(mm, clr, minv, maxv, shift);
        const Grid gr = make_grid(minv, maxv);
        const int v = (clr == 0) ? vy : (clr == 1) ? vco : vcg;
        const int sym = v + shift;
        Mix m;
        mix_prepare(par, clr, yv, cov, m);
        const uint32_t lo = cdf_entry(m, gr, sym);
        const uint32_t hi1 = cdf_entry(m, gr, min(sym + 1, gr.Lp - 2));     // unconditional (clamped): no divergent branch around ten erfc chains
        const uint32_t hi = (sym == gr.Lp - 2) ? 0u : hi1;
        pairs[(long)clr * s.pair_cs + s.pair_off + n] = (hi << 16) | lo;      // [clr][image][n]
    }
}
__global__ __launch_bounds__(kPairsThreads) void cdf_pairs_kernel(const int16_t *__restrict__ planes, const float *__restrict__ params,
                                                        const int32_t *__restrict__ minmax, StageGeom s, const StageGeom *__restrict__ sv,
                                                        uint32_t *__restrict__ pairs)
{
    cdf_pairs_body(planes, params, minmax, stage_at(s, sv, blockIdx.y), pairs);
}
// The three bands of one level in ONE launch (blockIdx.z = band; the band grids of a level have the same size, the coded crops differ):
// the encoder's levels 4..1, whose three CNN outputs fit side by side into the buffer level 0 needs anyway -- two launches fewer per level.
struct PairsBands { const StageGeom *sv[3]; const float *params[3]; uint32_t *pairs[3]; };      // sv: the band's per-image table
__global__ __launch_bounds__(kPairsThreads) void cdf_pairs_bands_kernel(const int16_t *__restrict__ planes, PairsBands a,
                                                              const int32_t *__restrict__ minmax)
{
    const int band = blockIdx.z;
    cdf_pairs_body(planes, a.params[band], minmax, a.sv[band][blockIdx.y], a.pairs[band]);
}

// decoder / seam export: full Lp-entry rows (entries >= Lp padded with 0xFFFF).  Persistent wavefronts, one
// row per wave iteration, lane l owns entry 64k + l of block k.  Per row the wave derives, for every mixture
// component, a conservative index interval outside which erfc_spec is exactly 0 (below) or 2 (above):
// x = -(p - mu) * rsig / sqrt2 is monotone in the sample index, |x| >= 7 saturates, and the interval is widened
// by 2 entries against rounding.  A (block, component) pair outside the interval contributes the constant 0
// or wn (bit-identical to evaluating erfc_spec there); only pairs that overlap it run the polynomial.
constexpr int kTabWaves = 4;

__global__ __launch_bounds__(64 * kTabWaves) void cdf_table_kernel(const int16_t *__restrict__ planes, const float *__restrict__ params,
                                                                   const int32_t *__restrict__ minmax, StageGeom s, int clr,
                                                                   uint16_t *__restrict__ tables, int row_stride,
                                                                   int n0, int cnt, int cap_rows)
{
    const int b = blockIdx.y;
    const int nc = s.hc * s.wc;
    const int lane = threadIdx.x & 63;
    const int wave0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * kTabWaves + (threadIdx.x >> 6)));
    const int nwaves = gridDim.x * kTabWaves;
    int minv, maxv, shift;
    clr_range(minmax + 4 * b, clr, minv, maxv, shift);
    const Grid gr = make_grid(minv, maxv);
    const int nblk = (row_stride + 63) >> 6;             // <= 8
    float pt[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) pt[k] = sample_pt(gr, min(64 * k + lane, gr.Lp - 1));
    const float fmin = (float)minv;
    const long img = (long)b * 3 * s.plane;
    const int mi = min(lane, 4);                         // lanes 0..4 prepare one mixture component each

    // rows [n0, n0 + cnt) of every image -> tables[b][cap_rows][row_stride] (row n at index n - n0).  A wavefront takes kTabRun CONSECUTIVE rows at a
    // time: with the channel-planar params a row's 15 - 25 parameters are single floats of as many planes, and eight consecutive rows share each of
    // their 32-byte sectors -- dealt out one row per wavefront (round 5) the eight rows went to eight workgroups on eight XCDs, whose L2s each fetched
    // the sector from HBM: FETCH_SIZE 3 - 12x the 60 - 100 bytes a row needs (profiles/r6/pmc_table_kernel_summary.csv).
    constexpr int kTabRun = 8;
    const int n_end = min(nc, n0 + cnt);
    for (int nb = n0 + kTabRun * wave0; nb < n_end; nb += kTabRun * nwaves)
    for (int n = nb; n < min(nb + kTabRun, n_end); ++n) {
        const int i = n / s.wc, j = n - i * s.wc;
        const ParRow par = par_row(params, b, (long)s.h * s.w, (long)i * s.w + j);
        const long off = img + ((long)(2 * i + s.oi) << s.lvl) * s.W + ((long)(2 * j + s.oj) << s.lvl);
        // component mi, prepared exactly as mix_prepare() does
        const float sgm = par[5 * clr + mi];
        float mu = par[16 + 5 * clr + mi];
        const float wk = par[32 + 5 * clr + mi];
        if (clr == 1) {
            const float t = par[48 + mi] * ((float)planes[off] / 255.0f);
            mu = mu + t;
        } else if (clr == 2) {
            const float t1 = par[48 + 5 + mi] * ((float)planes[off] / 255.0f);
            const float t2 = par[48 + 10 + mi] * ((float)planes[off + s.plane] / 255.0f);
            const float t = t1 + t2;
            mu = mu + t;
        }
        const float sg = (sgm > kScaleBound) ? sgm : kScaleBound;
        const float rsig = 1.0f / sg;
        const float w = (wk > kWeightBound) ? wk : kWeightBound;
        float ssum = w + dpp_row_shl(w, 1);              // (((w0 + w1) + w2) + w3) + w4 in lane 0
        ssum = ssum + dpp_row_shl(w, 2);
        ssum = ssum + dpp_row_shl(w, 3);
        ssum = ssum + dpp_row_shl(w, 4);
        ssum = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ssum)));
        const float wn = w / (1e-9f + ssum);
        // saturation interval in entry coordinates: entry i samples (minv - 0.5 + i) / 255 (the two pushed-out
        // end points are further out on their own side, hence at least as saturated as this says)
        const float c = mu * 255.0f - fmin + 0.5f, hw = 9.8994949f * 255.0f * sg + 2.0f;    // 7 * sqrt2
        int lo = -1, hi = 1 << 20;                       // entries <= lo: erfc = 0;  entries >= hi: erfc = 2
        if (c - hw > -1.0f && c - hw < 1e6f) lo = (int)(c - hw);
        if (c + hw > -1e6f && c + hw < 1e6f) hi = (int)(c + hw) + 1;
        if (!(c == c) || !(hw == hw)) { lo = -1; hi = 1 << 20; }
        float mu_[5], rs_[5], wn_[5];
        int lo_[5], hi_[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            mu_[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mu), k));
            rs_[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rsig), k));
            wn_[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wn), k));
            lo_[k] = __builtin_amdgcn_readlane(lo, k);
            hi_[k] = __builtin_amdgcn_readlane(hi, k);
        }
        uint16_t *row = tables + ((long)b * cap_rows + (n - n0)) * row_stride;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k < nblk) {
                const int e = 64 * k + lane;
                const int ec = min(e, gr.Lp - 1);
                // index span of the block's sample points; entries 0 and Lp-1 sit 20 grey levels further out
                const int bmin = (k == 0) ? -20 : 64 * k;
                const int bmax = (64 * k + 63 >= gr.Lp - 1) ? gr.Lp + 19 : 64 * k + 63;
                float acc = 0.0f;
#pragma unroll
                for (int m = 0; m < 5; ++m) {
                    float t;
                    if (bmax <= lo_[m]) t = 0.0f;                        // wn * (0.5 * 0)
                    else if (bmin >= hi_[m]) t = wn_[m];                 // wn * (0.5 * 2)
                    else {
                        const float z = (pt[k] - mu_[m]) * rs_[m];
                        t = wn_[m] * (0.5f * erfc_spec(kNegRsqrt2 * z));
                    }
                    acc = (m == 0) ? t : acc + t;
                }
                const float q = __builtin_rintf(acc * gr.scale);
                const uint32_t v = (uint32_t)((int)q + ec) & 0xFFFFu;
                if (e < row_stride) row[e] = (uint16_t)((e < gr.Lp) ? v : 0xFFFFu);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ anchors (AC decode)
// The whole-batch AC decoder does not need the full row: it needs to FIND the symbol, and for that every 8th entry
// (64 "anchors" for Lp <= 512) is enough -- the 8 entries between two anchors are evaluated by the decoding wave itself,
// on all 64 lanes at once (8 entries x 5 mixture components + 3 idle lanes per entry, ac_decode_anchor_kernel), with
// the same arithmetic, hence the same bits.  This kernel therefore does 1/8 of the table kernel's erfc work and
// writes 208 instead of 528 / 1024 bytes per symbol:
//   row n (kAnchorRow bytes):  [0, 128)   uint16 anchor[l] = entry[min(8 l, Lp - 1)],  l = 0 .. 63
//                              [128, 208) float4 (mu + cross-channel update, 1 / max(sigma, bound), normalised weight, 0) x 5

__global__ __launch_bounds__(64 * kTabWaves) void cdf_anchor_kernel(const int16_t *__restrict__ planes, const float *__restrict__ params,
                                                                    const int32_t *__restrict__ minmax, StageGeom s, int clr,
                                                                    uint8_t *__restrict__ rows, int n0, int cnt, int cap_rows)
{
    const int b = blockIdx.y;
    const int nc = s.hc * s.wc;
    const int lane = threadIdx.x & 63;
    const int wave0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * kTabWaves + (threadIdx.x >> 6)));
    const int nwaves = gridDim.x * kTabWaves;
    int minv, maxv, shift;
    clr_range(minmax + 4 * b, clr, minv, maxv, shift);
    const Grid gr = make_grid(minv, maxv);
    const int ec = min(8 * lane, gr.Lp - 1);
    const float pt = sample_pt(gr, ec);
    const long img = (long)b * 3 * s.plane;
    const int mi = min(lane, 4);                         // lanes 0..4 prepare one mixture component each
    for (int n = n0 + wave0; n < min(nc, n0 + cnt); n += nwaves) {
        const int i = n / s.wc, j = n - i * s.wc;
        const ParRow par = par_row(params, b, (long)s.h * s.w, (long)i * s.w + j);
        const long off = img + ((long)(2 * i + s.oi) << s.lvl) * s.W + ((long)(2 * j + s.oj) << s.lvl);
        // component mi, prepared exactly as mix_prepare() does
        const float sgm = par[5 * clr + mi];
        float mu = par[16 + 5 * clr + mi];
        const float wk = par[32 + 5 * clr + mi];
        if (clr == 1) {
            const float t = par[48 + mi] * ((float)planes[off] / 255.0f);
            mu = mu + t;
        } else if (clr == 2) {
            const float t1 = par[48 + 5 + mi] * ((float)planes[off] / 255.0f);
            const float t2 = par[48 + 10 + mi] * ((float)planes[off + s.plane] / 255.0f);
            const float t = t1 + t2;
            mu = mu + t;
        }
        const float sg = (sgm > kScaleBound) ? sgm : kScaleBound;
        const float rsig = 1.0f / sg;
        const float w = (wk > kWeightBound) ? wk : kWeightBound;
        float ssum = w + dpp_row_shl(w, 1);              // (((w0 + w1) + w2) + w3) + w4 in lane 0
        ssum = ssum + dpp_row_shl(w, 2);
        ssum = ssum + dpp_row_shl(w, 3);
        ssum = ssum + dpp_row_shl(w, 4);
        ssum = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ssum)));
        const float wn = w / (1e-9f + ssum);
        float acc = 0.0f;
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            const float mu_m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mu), m));
            const float rs_m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rsig), m));
            const float wn_m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wn), m));
            const float z = (pt - mu_m) * rs_m;
            const float t = wn_m * (0.5f * erfc_spec(kNegRsqrt2 * z));
            acc = (m == 0) ? t : acc + t;
        }
        const float q = __builtin_rintf(acc * gr.scale);
        const uint32_t v = (uint32_t)((int)q + ec) & 0xFFFFu;
        uint8_t *row = rows + ((long)b * cap_rows + (n - n0)) * kAnchorRow;
        reinterpret_cast<uint16_t *>(row)[lane] = (uint16_t)v;
        if (lane < 5) reinterpret_cast<float4 *>(row + 128)[lane] = make_float4(mu, rsig, wn, 0.0f);
    }
}
