// rans_coder.hpp -- LLICTI-rANS v2 container: 64-way interleaved rANS encoder and the table-free stage decoder.
// Part of the single translation unit llicti_hip.hip (included in order; not a stand-alone header).
#pragma once

// ------------------------------------------------------------------------------------------------ rANS container
// "LLICTI-rANS v2" (new format of this build; BASELINE.json north_star: "torchac replaced by a HIP rANS
// coder").  Same CDFs and symbols as the AC container; each image has M independent streams, each a
// 64-way interleaved rANS coder (32-bit states, 16-bit words, 16-bit probabilities) driven by ONE
// wavefront: lane l of stream m codes symbol n = 64c + l of every chunk c = m (mod M) of every stage.
// Words are shared by the 64 lanes in lane order (ballot + mbcnt prefix), so a whole stage decodes in
// ceil(nc / 64M) wave steps instead of nc serial symbols.
// v2 "absorbing start": the first symbol a lane's encoder codes (the LAST one its decoder decodes) starts from state
// x = freq(symbol) instead of 2^16, which makes the coded state 2^16 + c_low -- the 16 bits a rANS state holds at least
// then carry that symbol instead of nothing (saves one symbol's information per lane: ~95 of ~190 bytes per stream on
// noise).  The decoder reads no renormalisation word after a lane's last symbol and checks that the state left is freq.
__device__ __forceinline__ int lanes_below(uint64_t mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__global__ __launch_bounds__(64) void rans_encode_kernel(const uint32_t *__restrict__ pairs, const StreamDesc *__restrict__ desc,
                                                         int B, int M, uint8_t *__restrict__ slots, const long *__restrict__ rslot_off,
                                                         int rslot_cap, int32_t *__restrict__ rinfo, int32_t *status)
{
    const int sidx = blockIdx.x, b = sidx / M, m = sidx - b * M, lane = threadIdx.x;
    uint16_t *w16 = reinterpret_cast<uint16_t *>(slots + rslot_off[sidx]);
    long p = rslot_cap / 2;                         // word cursor, moving backwards from the end of the slot
    uint32_t x = 1u << 16;
    bool started = false;                           // v2: has this lane coded its first symbol yet?
    int bad = 0;
    for (int st = LLICTI_NSTREAMS - 1; st >= 0; --st) {      // rANS is LIFO: last decoded symbol first
        const StreamDesc d = desc[(long)st * B + b];
        const int nchunks = (d.n + 63) >> 6;
        if (nchunks <= m) continue;
        const int K = (nchunks - m + M - 1) / M;
        const uint32_t *pp = pairs + d.pair_off;
        // The pair loads do not depend on the coder state: four steps are kept in flight in four registers with
        // FIXED roles (the loop is unrolled by four; a rotating ring r0 = r1 ... makes the compiler copy the
        // newest load, i.e. wait for it with s_waitcnt vmcnt(0) in every step).  Loads are unconditional
        // (clamped address); the raw value is masked only where it is consumed.
        auto fetch = [&](int k) -> uint32_t { return pp[min(64 * (m + max(k, 0) * M) + lane, d.n - 1)]; };
        auto code = [&](int k, uint32_t raw) {
            const int n = 64 * (m + k * M) + lane;
            const bool active = k >= 0 && n < d.n;
            const uint32_t v = active ? raw : 0x00010000u;
            const uint32_t lo = v & 0xFFFFu;
            uint32_t hi = v >> 16;
            if (hi == 0) hi = 0x10000u;
            uint32_t freq = hi - lo;
            if (active && (freq == 0 || hi < lo)) { bad = 1; freq = 1; }
            const bool emit = active && started && ((uint64_t)x >= ((uint64_t)freq << 16));
            const uint64_t E = ballot64(emit);
            p -= __builtin_popcountll(E);
            if (p < 144) { bad = 2; p = 144; }                   // room for the state header (<= 140 words)
            if (emit) { w16[p + lanes_below(E)] = (uint16_t)(x & 0xFFFFu); x >>= 16; }
            if (active) {
                if (!started) { x = freq; started = true; }          // absorbing start: codes to 2^16 + lo
                // x < freq << 16 here, so the quotient fits 16 bits: a float reciprocal estimate is off by at most
                // one, and one signed remainder test repairs it (8 operations instead of a 32-bit division)
                uint32_t q = (uint32_t)((float)x * __builtin_amdgcn_rcpf((float)freq));
                int32_t r = (int32_t)(x - q * freq);
                if (r < 0) { q -= 1; r += (int32_t)freq; }
                else if (r >= (int32_t)freq) { q += 1; r -= (int32_t)freq; }
                x = (q << 16) + (uint32_t)r + lo;
            }
        };
        uint32_t r0 = fetch(K - 1), r1 = fetch(K - 2), r2 = fetch(K - 3), r3 = fetch(K - 4);
        for (int k = K - 1; k >= 0; k -= 4) {                 // steps k, k-1, k-2, k-3 (those below 0 are no-ops)
            code(k, r0);     r0 = fetch(k - 4);
            code(k - 1, r1); r1 = fetch(k - 5);
            code(k - 2, r2); r2 = fetch(k - 6);
            code(k - 3, r3); r3 = fetch(k - 7);
        }
    }
    // v2 compact flush of the 64 final states (x >> 16 is log-uniform in [1, 2^16): its bit length costs 4 bits, its
    // leading one nothing): 64 nibbles nb = bitlen(x >> 16) - 1 | 64 x uint16 low halves | nb mantissa bits per lane, lane
    // order, LSB first, zero padded to 16 bits  ->  160 .. 280 bytes instead of 256 (220 on average)
    __shared__ uint32_t fl_bits[32];
    if (lane < 32) fl_bits[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    const uint32_t hi16 = x >> 16;
    const int nb = 31 - __clz((int)hi16);                     // 0 .. 15 (hi16 >= 1)
    const uint32_t mant = hi16 & ((1u << nb) - 1u);
    int pre = nb;                                             // inclusive prefix sum over the lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(pre, d); if (lane >= d) pre += t; }
    const int total_bits = __shfl(pre, 63);
    const int bpos = pre - nb;
    if (nb > 0) {
        atomicOr(&fl_bits[bpos >> 5], mant << (bpos & 31));
        if ((bpos & 31) + nb > 32) atomicOr(&fl_bits[(bpos >> 5) + 1], mant >> (32 - (bpos & 31)));
    }
    __builtin_amdgcn_wave_barrier();
    const int nw = (total_bits + 15) >> 4;                     // mantissa words (16 bit)
    p -= 80 + nw;
    if (p < 0) { bad = 2; p = 0; }
    const int nb_next = __shfl_down(nb, 1);
    uint8_t *hdr = reinterpret_cast<uint8_t *>(w16 + p);
    if ((lane & 1) == 0) hdr[lane >> 1] = (uint8_t)(nb | (nb_next << 4));
    w16[p + 16 + lane] = (uint16_t)(x & 0xFFFFu);
    if (lane < nw) w16[p + 80 + lane] = (uint16_t)(fl_bits[lane >> 1] >> (16 * (lane & 1)));
    if (lane == 0) { rinfo[2 * sidx] = (int32_t)(2 * p); rinfo[2 * sidx + 1] = (int32_t)(rslot_cap - 2 * p); }
    if (bad) atomicExch(&status[0], bad == 1 ? LLICTI_EFORMAT : LLICTI_ENOSPACE);
}

// decode: parse the compact state header of every stream (see rans_encode_kernel) -> 64 states, word cursor 0 and the
// byte offset of the stream's first 16-bit word
__global__ __launch_bounds__(64) void rans_init_kernel(const uint8_t *__restrict__ slots, const long *__restrict__ rslot_off,
                                                       const int32_t *__restrict__ seg_len, int M,
                                                       uint32_t *__restrict__ rstate, uint32_t *__restrict__ rpos,
                                                       uint32_t *__restrict__ rwoff, int32_t *status)
{
    const int sidx = blockIdx.x, lane = threadIdx.x;
    const int b = sidx / M, m = sidx - b * M;
    const uint8_t *hdr = slots + rslot_off[sidx];
    const uint16_t *h16 = reinterpret_cast<const uint16_t *>(hdr);
    const int n = seg_len[(long)b * LLICTI_NSEG + 4 + m];          // validated >= 160 by rans_unpack_kernel (else a harmless header was written)
    const int nb = (hdr[lane >> 1] >> (4 * (lane & 1))) & 15;
    int pre = nb;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(pre, d); if (lane >= d) pre += t; }
    const int total_bits = __shfl(pre, 63);
    const int hbytes = 160 + 2 * ((total_bits + 15) >> 4);
    uint32_t x = 1u << 16;
    if (hbytes <= max(n, 160)) {
        const int bpos = pre - nb;
        const uint32_t win = (uint32_t)h16[80 + (bpos >> 4)] | ((uint32_t)h16[80 + (bpos >> 4) + 1] << 16);   // slot is zero padded past n
        const uint32_t mant = (win >> (bpos & 15)) & ((1u << nb) - 1u);
        x = (((1u << nb) | mant) << 16) | (uint32_t)h16[16 + lane];
    } else if (lane == 0) atomicExch(&status[0], LLICTI_EFORMAT);
    rstate[(long)sidx * 64 + lane] = x;
    if (lane == 0) { rpos[sidx] = 0; rwoff[sidx] = (uint32_t)min(hbytes, 280); }
}

// One stage (level, band, colour channel) of all images.  One workgroup of 4 wavefronts per stream (one per
// SIMD: with the stage VALU-issue bound, the busiest SIMD sets the pace, so waves per workgroup is a multiple
// of 4).  Every wave keeps its own copy of the 64 rANS states (the update is cheap and identical in all of
// them), so a step needs ONE barrier.  A wave resolves 16 of the step's 64 symbols, 4 lanes per symbol: lanes
// 0..2 of a group evaluate mixture components 0..2 of the probed table entry, lane 3 components 3 and 4; the
// five terms are summed in the spec's order over DPP row shifts, and a ballot hands the comparison to the
// group's lanes.  The symbol is first located with a CHEAP approximate CDF (Abramowitz-Stegun 7.1.26 erfc on
// v_rcp / v_exp, ~0.01 table counts of error) by bisection, then PROVEN with the exact spec arithmetic:
// entry[s] <= slot < entry[s+1] is checked with cdf_entry()'s operations, and if the guess is off the exact
// search gallops away from it and bisects -- so the result is bit-identical to an exact search whatever the
// approximation does.  The 64 (c_low, c_high) pairs meet in a ping-pong LDS buffer, after which every wave
// updates and renormalises its state copy.  No table in HBM.
constexpr int kRansWaves = 4;

// (((tA0 + tA1) + tA2) + tA3) + tB3 of the 4-lane group starting at this lane (meaningful in the group's first lane)
__device__ __forceinline__ float dpp_sum5(float tA, float tB)
{
    float acc = tA + dpp_row_shl(tA, 1);
    acc = acc + dpp_row_shl(tA, 2);
    acc = acc + dpp_row_shl(tA, 3);
    acc = acc + dpp_row_shl(tB, 3);
    return acc;
}
__device__ __forceinline__ float quad_lane0(float v)    // broadcast lane (l & ~3) to its quad
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x00, 0xF, 0xF, true));   // quad_perm [0,0,0,0]
}

template <int J>
__device__ __forceinline__ float quad_bcast(float v)     // broadcast lane (l & ~3) + J to its quad
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), J * 0x55, 0xF, 0xF, true));   // quad_perm [J,J,J,J]
}

struct Comp { float mu, rsig, wn; };

// exact table entry i (numerics spec); valid in the group's first lane.  erfc_spec_nobranch returns the same
// bits as erfc_spec (the saturation test selects the result instead of skipping the polynomial), which lets
// the two components' dependent chains interleave.
__device__ __forceinline__ uint32_t group_cdf_entry(const Comp &A, const Comp &B, const Grid &g, int i)
{
    const float pt = sample_pt(g, i);
    const float tA = A.wn * (0.5f * erfc_spec_nobranch(kNegRsqrt2 * ((pt - A.mu) * A.rsig)));
    const float tB = B.wn * (0.5f * erfc_spec_nobranch(kNegRsqrt2 * ((pt - B.mu) * B.rsig)));
    const float q = __builtin_rintf(dpp_sum5(tA, tB) * g.scale);
    return (uint32_t)((int)q + i) & 0xFFFFu;
}

// Approximate table entry i, 0 < i < Lp - 1 (search hint only -- never used as a result): Abramowitz-Stegun
// 7.1.26 erfc (|error| <= 1.5e-7) on v_rcp_f32 / v_exp_f32, with everything that does not depend on the sample
// point folded into per-component constants: x' = sqrt(log2 e) * x = c1 * pt + c0, exp(-x^2) = exp2(-x'^2),
// term = wh * erfc (wh = wn / 2).  15 vector operations per component and probe.
struct CompFast { float c1, c0, wh, wn; };
__device__ __forceinline__ CompFast comp_fast(const Comp &c)
{
    CompFast f;
    f.c1 = (kNegRsqrt2 * 1.2011224087864498f) * c.rsig;
    f.c0 = -c.mu * f.c1;
    f.wh = 0.5f * c.wn;
    f.wn = c.wn;
    return f;
}
__device__ __forceinline__ float term_fast(const CompFast &c, float pt)
{
    const float x = __builtin_fmaf(pt, c.c1, c.c0);
    const float a = __builtin_fabsf(x);
    const float u = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f / 1.2011224087864498f, a, 1.0f));
    float p = __builtin_fmaf(1.061405429f, u, -1.453152027f);
    p = __builtin_fmaf(p, u, 1.421413741f);
    p = __builtin_fmaf(p, u, -0.284496736f);
    p = __builtin_fmaf(p, u, 0.254829592f);
    const float E = ((p * u) * __builtin_amdgcn_exp2f(-(a * a))) * c.wh;
    return (x < 0.0f) ? c.wn - E : E;
}
__device__ __forceinline__ int group_cdf_entry_fast(const CompFast &A, const CompFast &B, float fbase, float scale, int i)
{
    const float pt = div255_exact(fbase + (float)i);     // the exact sample point: near a narrow component the CDF moves by counts per ulp of pt
    return (int)__builtin_rintf(dpp_sum5(term_fast(A, pt), term_fast(B, pt)) * scale) + i;
}

template <int CLR>
__global__ __launch_bounds__(64 * kRansWaves) void rans_decode_stage_kernel(const float *__restrict__ params, StageGeom sg, int M,
                                                               const uint8_t *__restrict__ slots, const long *__restrict__ rslot_off,
                                                               int rslot_cap, uint32_t *__restrict__ rstate, uint32_t *__restrict__ rpos,
                                                               const uint32_t *__restrict__ rwoff,
                                                               int16_t *__restrict__ planes, float *__restrict__ fplanes,
                                                               const int32_t *__restrict__ minmax, int later_max, int32_t *status)
{
    __shared__ uint32_t sh_res[2][64][2];        // ping-pong by step parity: [0] = c_low, [1] = c_high
    const int sidx = blockIdx.x, b = sidx / M, m = sidx - b * M;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nc = sg.hc * sg.wc;
    const int nchunks = (nc + 63) >> 6;
    if (nchunks <= m) return;                    // whole workgroup
    const int K = (nchunks - m + M - 1) / M;
    uint32_t x = rstate[(long)sidx * 64 + lane], pos = rpos[sidx];      // every wave: its own copy
    const uint32_t woff = rwoff[sidx];                                  // bytes of the compact state header (160 .. 280, even)
    const uint16_t *words = reinterpret_cast<const uint16_t *>(slots + rslot_off[sidx] + woff);
    const uint32_t max_words = ((uint32_t)rslot_cap - woff) / 2;
    constexpr int clr = CLR;                     // compile-time: no branch (hence no register merge, hence no s_waitcnt vmcnt(0)) next to the prefetch loads
    int minv, maxv, shift;
    clr_range(minmax + 4 * b, clr, minv, maxv, shift);
    const Grid gr = make_grid(minv, maxv);
    const int max_symbol = gr.Lp - 2;
    const float fbase = (float)minv - 0.5f;
    const long img = (long)b * 3 * sg.plane;
    const int mA = lane & 3;                                            // component A of this lane; component B is 4 (read from lane 3 only)
    const int gsym = 16 * wave + (lane >> 2);                           // symbol (lane of the stream) this 4-lane group resolves
    const int gbit = lane & ~3;                                         // ballot bit of the group's first lane
    const bool head = (mA == 0);
    // Raw CNN outputs / prior-channel pixels of this group's symbol in step k: requested one step ahead, so the
    // memory round trip runs under the previous step's search instead of in front of this one's.
    struct Raw { float sgA, muA, wkA, a0A, a1A, sgB, muB, wkB, a0B, a1B, y, co; long off; bool on; };
    auto fetch = [&](int k) -> Raw {
        Raw r;
        const int n = min(64 * (m + k * M) + gsym, nc - 1);          // clamped: the loads are unconditional
        const int i = div_wc(sg, n), j = n - i * sg.wc;          // multiply-shift: a runtime division costs ~25 of the step's ~900 instructions
        const float *par = params + ((long)b * sg.h * sg.w + (long)i * sg.w + j) * LLICTI_PARAM_STRIDE;
        r.off = img + ((long)(2 * i + sg.oi) << sg.lvl) * sg.W + ((long)(2 * j + sg.oj) << sg.lvl);
        r.sgA = par[5 * clr + mA]; r.muA = par[16 + 5 * clr + mA]; r.wkA = par[32 + 5 * clr + mA];
        r.sgB = par[5 * clr + 4];  r.muB = par[16 + 5 * clr + 4];  r.wkB = par[32 + 5 * clr + 4];
        r.a0A = r.a1A = r.a0B = r.a1B = r.y = r.co = 0.0f;
        if constexpr (clr == 1) { r.a0A = par[48 + mA]; r.a0B = par[48 + 4]; r.y = fplanes[r.off]; }
        else if constexpr (clr == 2) {
            r.a0A = par[48 + 5 + mA]; r.a1A = par[48 + 10 + mA]; r.a0B = par[48 + 5 + 4]; r.a1B = par[48 + 10 + 4];
            r.y = fplanes[r.off]; r.co = fplanes[r.off + sg.plane];
        }
        r.on = (k < K) && (64 * (m + k * M) + gsym) < nc;
        return r;
    };
    // component (sigma, mu, w) -> (mu with the cross-channel update, 1 / max(sigma, bound), max(w, bound)), as mix_prepare()
    auto prep = [&](float sgm, float mu, float wk, float a0, float a1, float y, float co, float &w) -> Comp {
        Comp cpt;
        if constexpr (clr == 1) { const float t = a0 * y; mu = mu + t; }
        else if constexpr (clr == 2) { const float t1 = a0 * y; const float t2 = a1 * co; const float t = t1 + t2; mu = mu + t; }
        cpt.mu = mu;
        cpt.rsig = 1.0f / ((sgm > kScaleBound) ? sgm : kScaleBound);
        w = (wk > kWeightBound) ? wk : kWeightBound;
        cpt.wn = 0.0f;
        return cpt;
    };
    // Stream words: lane l holds word wbase + l, a second register the 64 after them; a step consumes at most
    // 64 words, pulled with ds_bpermute instead of a dependent global load.
    uint32_t wbase = pos & ~63u;
    auto load_words = [&](uint32_t w0) -> uint32_t { return words[min(w0 + (uint32_t)lane, max_words - 1)]; };
    uint32_t win0 = load_words(wbase), win1 = load_words(wbase + 64);
    Raw cur = fetch(0);
    for (int k = 0; k < K; ++k) {
        const int chunk0 = 64 * (m + k * M);
        const Raw nxt = fetch(min(k + 1, K - 1));
        // slot of this group's symbol = low half of the state in lane gsym of this wave's copy
        const uint32_t slot = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * gsym, (int)x) & 0xFFFFu;
        {
            if (cur.on) {                        // uniform within the group
                const long off = cur.off;
                float wA, wB;
                Comp A = prep(cur.sgA, cur.muA, cur.wkA, cur.a0A, cur.a1A, cur.y, cur.co, wA);
                Comp B = prep(cur.sgB, cur.muB, cur.wkB, cur.a0B, cur.a1B, cur.y, cur.co, wB);
                const float ssum = quad_lane0(dpp_sum5(wA, wB));     // (((w0 + w1) + w2) + w3) + w4
                const float den = 1e-9f + ssum;
                A.wn = wA / den;
                B.wn = wB / den;

                // 1. hint: 5-ary search on the approximate table.  Every lane of the group holds ALL five components in
                //    fast form (quad broadcasts) and evaluates the whole approximate mixture at its OWN probe point, so a
                //    round costs one 5-term evaluation (no cross-lane sum) and cuts the bracket to a fifth: 4 rounds for
                //    Lp = 512 instead of 9 bisection rounds.  The phase is instruction-issue bound (in-kernel stamps), and
                //    4 x ~85 instructions are fewer than 9 x 56.
                const CompFast Af = comp_fast(A), Bf = comp_fast(B);
                CompFast F0, F1, F2, F3;
                F0.c1 = quad_bcast<0>(Af.c1); F0.c0 = quad_bcast<0>(Af.c0); F0.wh = quad_bcast<0>(Af.wh); F0.wn = quad_bcast<0>(Af.wn);
                F1.c1 = quad_bcast<1>(Af.c1); F1.c0 = quad_bcast<1>(Af.c0); F1.wh = quad_bcast<1>(Af.wh); F1.wn = quad_bcast<1>(Af.wn);
                F2.c1 = quad_bcast<2>(Af.c1); F2.c0 = quad_bcast<2>(Af.c0); F2.wh = quad_bcast<2>(Af.wh); F2.wn = quad_bcast<2>(Af.wn);
                F3.c1 = quad_bcast<3>(Af.c1); F3.c0 = quad_bcast<3>(Af.c0); F3.wh = quad_bcast<3>(Af.wh); F3.wn = quad_bcast<3>(Af.wn);
                int glo = 0, ghi = max_symbol + 1;
                while (ghi - glo > 1) {
                    const int stp = (ghi - glo + 4) / 5;                   // >= 1; the last part is the (smaller) remainder
                    auto probe_at = [&](int j) { return min(glo + stp * (j + 1), ghi - 1); };
                    const int pi = probe_at(mA);
                    const float pt = div255_exact(fbase + (float)pi);     // the exact sample point: near a narrow component the CDF moves by counts per ulp of pt
                    float sum = term_fast(F0, pt);
                    sum += term_fast(F1, pt);
                    sum += term_fast(F2, pt);
                    sum += term_fast(F3, pt);
                    sum += term_fast(Bf, pt);
                    const int e = (int)__builtin_rintf(sum * gr.scale) + pi;
                    const uint64_t bal = ballot64(e <= (int)slot);
                    // the four probes are ordered, so the passes form a prefix of the quad's lanes (if the approximation
                    // ever breaks that, the hint is merely wrong: the proof below decides)
                    const int np = __builtin_popcount((uint32_t)(bal >> gbit) & 0xFu);
                    const int nlo = (np > 0) ? probe_at(np - 1) : glo;
                    const int nhi = (np < 4) ? probe_at(np) : ghi;
                    glo = nlo; ghi = nhi;
                }
                // 2. proof with the exact spec arithmetic: entries glo and glo + 1 in one round (independent chains);
                //    if the hint is off, gallop away from it and bisect
                int lo = 0, hi = max_symbol + 1;
                uint32_t vlo = 0, vhi = 0x10000u;                    // meaningful in the group's first lane only
                bool have_lo = false, have_hi = false;
                {
                    const int s1 = glo, s2 = min(glo + 1, max_symbol);
                    // components 0..3 of both entries in their own lanes; component 4 of entry s1 in the group's
                    // lane 0 and of entry s2 in lane 1 (every lane holds component 4's parameters): three
                    // evaluations per lane instead of four
                    const float p1 = sample_pt(gr, s1), p2 = sample_pt(gr, s2);
                    const float pX = (mA == 1) ? p2 : p1;
                    const float t1 = A.wn * (0.5f * erfc_spec_nobranch(kNegRsqrt2 * ((p1 - A.mu) * A.rsig)));
                    const float t2 = A.wn * (0.5f * erfc_spec_nobranch(kNegRsqrt2 * ((p2 - A.mu) * A.rsig)));
                    const float tX = B.wn * (0.5f * erfc_spec_nobranch(kNegRsqrt2 * ((pX - B.mu) * B.rsig)));
                    float a1 = t1 + dpp_row_shl(t1, 1);              // (((t0 + t1) + t2) + t3) + t4, in the group's lane 0
                    a1 = a1 + dpp_row_shl(t1, 2);
                    a1 = a1 + dpp_row_shl(t1, 3);
                    a1 = a1 + tX;
                    float a2 = t2 + dpp_row_shl(t2, 1);
                    a2 = a2 + dpp_row_shl(t2, 2);
                    a2 = a2 + dpp_row_shl(t2, 3);
                    a2 = a2 + dpp_row_shl(tX, 1);
                    const uint32_t eA = (uint32_t)((int)__builtin_rintf(a1 * gr.scale) + s1) & 0xFFFFu;
                    const uint32_t eB = (uint32_t)((int)__builtin_rintf(a2 * gr.scale) + s2) & 0xFFFFu;
                    const bool bA = (ballot64(eA <= slot) >> gbit) & 1ull;
                    const bool bB = (ballot64(eB <= slot) >> gbit) & 1ull;
                    const bool leA = (s1 == 0) || bA;                // entry 0 is the floor of the search (torchac: left = 0)
                    const bool leB = (s1 + 1 <= max_symbol) && bB;   // past the top symbol: c_high = 0x10000 by definition
                    if (leA) {
                        lo = s1; vlo = eA; have_lo = true;
                        if (leB) { lo = s2; vlo = eB; }
                        else if (s1 + 1 <= max_symbol) { hi = s2; vhi = eB; have_hi = true; }
                    } else { hi = s1; vhi = eA; have_hi = true; }
                }
                int step = 2;
                while (hi - lo > 1) {
                    int probe;
                    if (have_lo && have_hi) probe = (lo + hi) >> 1;
                    else if (have_lo) { probe = min(lo + step, hi - 1); step <<= 1; }
                    else { probe = max(hi - step, lo + 1); step <<= 1; }
                    const uint32_t e = group_cdf_entry(A, B, gr, probe);
                    const uint64_t bal = ballot64(e <= slot);
                    if ((bal >> gbit) & 1ull) { lo = probe; vlo = e; have_lo = true; } else { hi = probe; vhi = e; have_hi = true; }
                }
                if (!have_lo) vlo = group_cdf_entry(A, B, gr, 0);
                if (head) {
                    sh_res[k & 1][gsym][0] = vlo;
                    sh_res[k & 1][gsym][1] = vhi;
                    const int v = lo - shift;
                    planes[off + (long)clr * sg.plane] = (int16_t)v;
                    fplanes[off + (long)clr * sg.plane] = (float)v / 255.0f;
                }
            }
        }
        __syncthreads();
        {
            const bool active = chunk0 + lane < nc;
            // v2: a lane's last symbol (no later slot in this stage, none in any later stage: lane l of stream m is active
            // in a stage of n symbols iff 64 m + l < n) is followed by no read, and must leave the encoder's start state freq
            const bool fin = active && (chunk0 + lane + 64 * M >= nc) && (64 * m + lane >= later_max);
            if (active) {
                const uint32_t vlo = sh_res[k & 1][lane][0], vhi = sh_res[k & 1][lane][1];
                x = (vhi - vlo) * (x >> 16) + (x & 0xFFFFu) - vlo;
                if (fin && x != vhi - vlo && wave == 0) atomicExch(&status[0], LLICTI_EFORMAT);
            }
            const bool need = active && !fin && x < 0x10000u;
            const uint64_t E = ballot64(need);
            const uint32_t idx = pos + (uint32_t)lanes_below(E);
            const uint32_t rel = idx - wbase;                                   // < 128
            const uint32_t wa = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (int)(rel & 63u), (int)win0);
            const uint32_t wb = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (int)(rel & 63u), (int)win1);
            if (need) {
                const uint32_t wv = (idx < max_words) ? ((rel < 64u) ? wa : wb) : 0u;
                x = (x << 16) | wv;
            }
            pos += (uint32_t)__builtin_popcountll(E);
            if (pos - wbase >= 64u) { wbase += 64u; win0 = win1; win1 = load_words(wbase + 64); }
        }
        cur = nxt;
    }
    if (wave == 0) {
        rstate[(long)sidx * 64 + lane] = x;
        if (lane == 0) rpos[sidx] = pos;
    }
}

__global__ __launch_bounds__(256) void rans_pack_kernel(const uint8_t *__restrict__ slots, const long *__restrict__ rslot_off,
                                                        const int32_t *__restrict__ rinfo, int M, int hdr_bytes,
                                                        uint8_t *__restrict__ out, long out_stride, int32_t *__restrict__ seg_len, int32_t *status)
{
    const int m = blockIdx.x, b = blockIdx.y;
    long dst = hdr_bytes;
    for (int k = 0; k < m; ++k) dst += rinfo[2 * (b * M + k) + 1];
    const int n = rinfo[2 * (b * M + m) + 1];
    if (dst + n > out_stride) { if (threadIdx.x == 0) atomicExch(&status[0], LLICTI_ENOSPACE); return; }
    const uint8_t *src = slots + rslot_off[b * M + m] + rinfo[2 * (b * M + m)];
    uint8_t *o = out + (long)b * out_stride + dst;
    block_copy_bytes(o, src, n);
    if (threadIdx.x == 0) {
        seg_len[(long)b * LLICTI_NSEG + 4 + m] = n;
        if (m == 0) for (int k = 4 + M; k < LLICTI_NSEG; ++k) seg_len[(long)b * LLICTI_NSEG + k] = 0;
    }
}

__global__ __launch_bounds__(256) void rans_unpack_kernel(const uint8_t *__restrict__ in, long in_stride, const int32_t *__restrict__ seg_len,
                                                          int M, uint8_t *__restrict__ slots, const long *__restrict__ rslot_off,
                                                          int rslot_cap, int32_t *status)
{
    const int m = blockIdx.x, b = blockIdx.y;
    const int32_t *sl = seg_len + (long)b * LLICTI_NSEG;
    long src = 0;
    bool bad = false;
    for (int k = 0; k < 4 + m; ++k) {           // see unpack_kernel: every earlier entry is validated too
        const int v = sl[k];
        if (v < 0 || v > in_stride) bad = true;
        src += v;
        if (src < 0 || src > in_stride) { bad = true; src = 0; }
    }
    int n = sl[4 + m];
    uint8_t *o = slots + rslot_off[b * M + m];
    if (bad || n < 160 || n > rslot_cap || src + n > in_stride) {
        if (threadIdx.x == 0) atomicExch(&status[0], LLICTI_EFORMAT);
        for (int t = threadIdx.x; t < 160; t += blockDim.x) o[t] = 0;                       // states = 1 << 16, no mantissa bits: harmless
        n = 160;
    } else {
        const uint8_t *p = in + (long)b * in_stride + src;
        block_copy_bytes(o, p, n);
    }
    const int padded = min(rslot_cap, n + 64);
    for (int t = n + threadIdx.x; t < padded; t += blockDim.x) o[t] = 0;
}
