// rans_coder.hpp -- LLICTI-rANS v3 container: 64- or 128-way interleaved, bit-granular rANS encoder, the table-free stage decoders
// and the tail decoder.  Part of the single translation unit llicti_hip.hip (included in order; not a stand-alone header).
#pragma once
#include <type_traits>

// ------------------------------------------------------------------------------------------------ rANS container
// "LLICTI-rANS v3" (format of this build; BASELINE.json north_star: "torchac replaced by a HIP rANS coder"; spec and CPU
// restatement: oracle/llicti_oracle.h / .c).  Same CDFs and symbols as the AC container; each image has M independent
// streams, each an L-way interleaved rANS coder (L = 64 lanes, or 128 for wide streams) encoded by ONE wavefront: lane l of stream m
// codes symbol n = Lc + l of every chunk c = m (mod M) of every stage, so a whole stage decodes in ceil(nc / LM) steps.
// What v3 changes against v2 (16-bit words, states in [2^16, 2^32), ~60 bytes of start / flush overhead per stream):
//   * states live in [2^31, 2^32) and renormalise BIT by bit (0..16 bits per symbol): x / freq >= 2^15, so the coder loses
//     ~2^-16 of a symbol's length like the range coder does (v2 lost ~0.005 bpp on smooth content), and a final state is
//     31 bits flat -- no length field;
//   * a lane's INITIAL state carries payload instead of nothing: the last T symbols of the stream's last stage are coded by a
//     single-state "tail" coder whose output (<= 31 L bits) is cut into the L x 31 bits the lanes start from.  The decoder
//     is left with those states after the last stage, reassembles the tail stream and decodes its T symbols serially.
// Cost over the ideal code length: ~6 bytes per stream (v2: ~60) -- ten streams are within 0.001 bpp of the AC container.
// Stream bytes:  u16 (T | pad << 11) | bit region, read DOWN from its top minus pad unused bits | L x 31-bit final states.

__device__ __forceinline__ int lanes_below(uint64_t mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// inclusive prefix sum over the 64 lanes (DPP: row_shr 1, 2, 4, 8 inside each row of 16, then row_bcast 15 / 31)
__device__ __forceinline__ int wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);      // lane 15 of rows 0 / 2 -> rows 1 / 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);      // lane 31 -> rows 2, 3
    return v;
}

// encoder renormalisation: the smallest n with (x >> n) < freq << 16, x in [2^31, 2^32), 1 <= freq <= 2^16
__device__ __forceinline__ int rans_emit_bits(uint32_t x, uint32_t freq)
{
    if (freq >= 0x10000u) return 0;
    const int n0 = __clz((int)freq) - 16;                 // 32 - bit length of (freq << 16)
    return n0 + (((x >> n0) >= (freq << 16)) ? 1 : 0);
}
// the same for ANY state 0 <= x < 2^32 (the one-chain tail of an xwide v4 stream starts from a small state: nothing to emit until it has grown)
__device__ __forceinline__ int rans_emit_bits_any(uint32_t x, uint32_t freq)
{
    if (freq >= 0x10000u) return 0;
    const int n0 = max(__clz((int)freq) - 16 - __clz((int)x), 0);
    return n0 + (((x >> n0) >= (freq << 16)) ? 1 : 0);
}
// C(s, x) = (x / freq) << 16 + x % freq + lo for x < freq << 16: the quotient fits 16 bits, so a float reciprocal estimate
// is off by at most one and one signed remainder test repairs it (8 operations instead of a 32-bit division)
__device__ __forceinline__ uint32_t rans_push(uint32_t x, uint32_t lo, uint32_t freq)
{
    const uint32_t q = (uint32_t)((float)x * __builtin_amdgcn_rcpf((float)freq));
    const int32_t r = (int32_t)(x - q * freq);
    const bool neg = r < 0, big = r >= (int32_t)freq;                            // selects, not branches: the push is on the coder's dependency chain
    const uint32_t q2 = q + (big ? 1u : 0u) - (neg ? 1u : 0u);
    const int32_t r2 = r + (neg ? (int32_t)freq : 0) - (big ? (int32_t)freq : 0);
    return (q2 << 16) + (uint32_t)r2 + lo;
}
// n bits (0 <= n <= 32) at bit position pos of a little-endian dword array in LDS
__device__ __forceinline__ uint32_t lds_get_bits(const uint32_t *buf, int pos, int n)
{
    const uint64_t w = (uint64_t)buf[pos >> 5] | ((uint64_t)buf[(pos >> 5) + 1] << 32);
    return (uint32_t)(w >> (pos & 31)) & (uint32_t)((1ull << n) - 1ull);
}
__device__ __forceinline__ void lds_or_bits(uint32_t *buf, int pos, int n, uint32_t v)     // v < 2^n
{
    if (n == 0) return;
    atomicOr(&buf[pos >> 5], v << (pos & 31));
    if ((pos & 31) + n > 32) atomicOr(&buf[(pos >> 5) + 1], v >> (32 - (pos & 31)));
}

// "auto" xwide encodes (LLICTI_MODE_RANS_X_AUTO; host_types.hpp: rans_auto_pick): the weights 16 - floor(log2 freq) of the symbols of an image's
// LAST stage -- the pairs are all there before the coder runs -- are summed into ssum[b] (zeroed by header_write_kernel at the start of the call;
// kAutoSlices workgroups per image: one per image walked its 98,304 pairs of a 768x512 image in 384 dependent trips, 0.2 ms of a 0.3 ms coder), and
// every coder / pack workgroup works the image's stream count out of the sum itself.  A pure function of the image.
constexpr int kAutoSlices = 48;
__global__ __launch_bounds__(256) void choose_streams_kernel(const uint32_t *__restrict__ pairs, const StreamDesc *__restrict__ desc, int B,
                                                             unsigned long long *__restrict__ ssum)
{
    const int b = blockIdx.y;
    const StreamDesc dl = desc[(long)(LLICTI_NSTREAMS - 1) * B + b];
    const uint32_t *pl = pairs + dl.pair_off;
    long long s = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < dl.n; i += 256 * kAutoSlices) {
        const uint32_t v = pl[i];
        const uint32_t lo = v & 0xFFFFu, hi = (v >> 16) ? (v >> 16) : 0x10000u;
        s += __clz((int)max(hi - lo, 1u)) - 15;
    }
    __shared__ long long sh[256];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k]; __syncthreads(); }
    if (threadIdx.x == 0 && sh[0]) atomicAdd(&ssum[b], (unsigned long long)sh[0]);
}
// the image's stream count: the table's (fixed by the caller), or -- "auto" -- what rans_auto_pick makes of the size rule's count and the sum above
__device__ __forceinline__ int image_stream_count(const unsigned long long *ssum, const ImgGeo *iv, const StreamDesc *desc, int B, int b, int M_table)
{
    if (!ssum) return M_table;
    const int Mlo = iv[b].Mlo;
    return Mlo ? rans_auto_pick(Mlo, (long long)ssum[b], (long long)desc[(long)(LLICTI_NSTREAMS - 1) * B + b].n) : M_table;
}

// One wavefront per 64 lanes of a stream (a wide stream: two, one per sub-chunk; their bit fields interleave, so the step's bit
// totals meet in LDS behind one barrier).  Slot layout (rslot_off is 64-byte aligned): [0,2) unused | [2,4) T | [4, 4 + nbytes) bit region
// (dword aligned) | 248 Q bytes of final states; rinfo = (2, 2 + nbytes + 248 Q) for rans_pack_kernel.
template <int Q>
__global__ __launch_bounds__(64 * Q) void rans_encode_kernel(const uint32_t *__restrict__ pairs, const StreamDesc *__restrict__ desc,
                                                         int B, const StreamRef *__restrict__ sref, uint8_t *__restrict__ slots, const long *__restrict__ rslot_off,
                                                         int rslot_cap, int32_t *__restrict__ rinfo, int32_t *status,
                                                         const StageGeom *__restrict__ sglv, const int16_t *__restrict__ planes, const int32_t *__restrict__ minmax,
                                                         const unsigned long long *__restrict__ ssum, const ImgGeo *__restrict__ iv)
{
    using GEO = RansGeo<Q>;
    constexpr int L = GEO::kLanes;
    const int M_img = image_stream_count(ssum, iv, desc, B, sref[blockIdx.x].b, sref[blockIdx.x].M);
    if (sref[blockIdx.x].m >= M_img) {                                   // "auto": the image got fewer streams than the table holds for it -- this one does not exist
        if (threadIdx.x == 0) { rinfo[2 * blockIdx.x] = 4; rinfo[2 * blockIdx.x + 1] = 0; }
        return;                                                          // (whole workgroup, in front of every barrier)
    }
    __shared__ uint32_t sh_pay[64 * Q + kRansSpillMax / 32 + 8];      // the tail coder's output (xwide v4: payload ++ spill) / the final states (62 Q dwords used, the rest slack)
    // dwords: one flush = one dword per thread; a step adds at most kFlush / 2, a round of four steps 2 kFlush.  Ring of 8 flush units: when a
    // round's barrier opens, up to 3 kFlush dwords are waiting (a leftover below one unit + the previous round's), the round's own ORs reach
    // 2 kFlush + 1 further -- below wbase + 5 kFlush + 2 -- while slower wavefronts may still be reading / zeroing [wbase, wbase + 2 kFlush):
    // with 4 units (the size the one-step-per-round form needed) those ORs wrapped onto the units being flushed whenever the wavefronts
    // of a stream drifted apart -- never in a test that had the chip to itself, at once under a second context's kernels (bench.py's
    // overlapped_streams leg caught it; tests/test_hip_parity.py::test_two_contexts_concurrently_bitexact now does).
    constexpr int kFlush = 64 * Q, kWin = 8 * kFlush;
    __shared__ uint32_t sh_win[kWin];               // staging RING of the bit region: stream dword d at sh_win[d & (kWin - 1)], d in [wbase, wbase + kWin)
    __shared__ __attribute__((aligned(16))) int sh_tot[2][Q][4];       // a round's four bit totals per sub-chunk (ping-pong by round parity)
    const int sidx = blockIdx.x;
    const StreamRef sr_ = sref[sidx];               // the stream's image, its index among the image's streams, the image's stream count (images of a call may differ)
    const int b = sr_.b, m = sr_.m, M = M_img;
    const int tid = threadIdx.x, lane = tid & 63, wq = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wavefront wq codes sub-chunk wq (stream lanes 64 wq .. 64 wq + 63); scalar: what depends on it alone branches on the scalar unit
    const StageGeom sgl = sglv[b];                  // the image's last stage (level 0, band x10): an xwide stream's seed symbols are read from its pixels
    uint8_t *slot = slots + rslot_off[sidx];
    uint32_t *out32 = reinterpret_cast<uint32_t *>(slot + 4);
    const int cap_dw = (rslot_cap - 4 - GEO::kPayBytes - 8) >> 2;      // dwords the bit region may take
    int bad = 0;
    sh_pay[tid] = 0;
    if (tid < kRansSpillMax / 32 + 8) sh_pay[64 * Q + tid] = 0;
#pragma unroll
    for (int t = 0; t < kWin; t += kFlush) sh_win[t + tid] = 0;
    __syncthreads();

    // 1. tail: the stream's last T symbols of the last stage, last symbol first, single state; bits go UP from bit 0 of
    //    the payload, the final state (32 bits, leading one = highest set bit of the payload) on top.  The recursion is serial and
    //    wave-uniform (every wavefront runs it, on its scalar unit; thread 0 writes): what a symbol costs is its dependency chain, so
    //    the chain holds nothing but the state update -- the pairs come out of a register by v_readlane (64 at a time, the next 64 in
    //    flight), the emitted bits collect in a 64-bit accumulator and reach LDS a dword at a time.  (Round 3 read every pair from LDS
    //    and ORed every bit field into LDS with an atomic: ~700 cycles a symbol; an xwide stream has ~620 tail symbols.)
    const StreamDesc dl = desc[(long)(LLICTI_NSTREAMS - 1) * B + b];
    const int cnt = rans_stream_count(dl.n, m, M, L);
    int T = 0;
    int tail_single = 0;                            // xwide: the tail has one chain, not two (bit 8 of the stream's header field)
    int alen = GEO::kPayBits;                       // xwide v4: bits of the tail coder's output; what exceeds the payload starts the main bit region
    if constexpr (kSeeded<Q>) {
        // xwide v4 (host_types.hpp, oracle/llicti_oracle.h): ONE chain that starts from the stream's last symbol itself (raw) and emits nothing while
        // its state is small, or TWO seeded chains where symbols are expensive -- chain c on wavefront c.  The chains are independent but for the
        // stop rule, which looks at both: they run in blocks of 32 steps, record (field, bits) of every step in symbol order and their states of the
        // last two blocks, and meet at a barrier after each block.  The output ("arena") is not cut to the payload: symbols are taken until, at a
        // multiple of 32 counted from the stream's end, it has reached the payload's 7,936 bits -- so the recursion runs one block past the block that
        // got there -- and what exceeds the payload (< 512 bits, the spill) is handed to the main coder as the bottom of its bit region.  Where
        // exactly the rule stops -- T -- and where each field goes are prefix sums over the records, done by all 256 threads.
        __shared__ uint32_t sh_fld[kRansTailMaxX + 1];  // (field, bits) of every tail step, in symbol order: up to 8,160 of them (a tail of 2,047 symbols
                                                        // fills the 7,936-bit payload only at 3.9 bits per symbol; the trained model's last stage costs 1.7)
        __shared__ uint32_t sh_xs[2][2][34];            // [block parity][chain]: the state at the block's start, then behind each of its steps
        __shared__ int sh_used[2][2], sh_scan[2][4], sh_cut[3];
        const uint32_t *pl = pairs + dl.pair_off;
        int minv, maxv, shift;
        clr_range(minmax + 4 * b, 2, minv, maxv, shift);
        const int A = maxv - minv + 1;
        uint32_t pw;
        const int ns = rans_seed_count(A, pw);
        // one chain or two (the rule of oracle/llicti_oracle.c, rans_tail_encode_x: a second chain pays when symbols are expensive): on the
        // stream's last up to 64 symbols, two iff ns * mean(16 - floor(log2 freq)) >= 32 + ns / 2 -- every wavefront computes it for itself
        int nch = 1;
        if (cnt >= 2 * ns) {
            const int k64 = min(cnt, 64);
            int wgt = 0;
            if (lane < k64) {
                const int q = cnt - 1 - lane;
                const uint32_t v = pl[L * (m + (q / L) * M) + (q % L)];
                const uint32_t lo = v & 0xFFFFu, hi = (v >> 16) ? (v >> 16) : 0x10000u;
                wgt = __clz((int)max(hi - lo, 1u)) - 15;
            }
            const int wsum = __builtin_amdgcn_readlane(wave_incl_scan(wgt), 63);
            if (2 * wsum * ns >= k64 * (64 + ns)) nch = 2;
        }
        tail_single = (nch == 1);
        const int sn = (nch == 2) ? ns : 1;                                // raw symbols in a chain's start state
        const int fixed = (nch == 2) ? 64 : 33;                            // the final states; one chain: + its end marker
        const int NS = min(nch * sn, cnt);
        const int ncod = max(min(cnt, kRansTailMaxX) - nch * sn, 0);       // candidates j = nch sn + idx, idx < ncod; chain c takes idx = nch i + c
        const int n_own = (wq < nch) ? (ncod + nch - 1 - wq) / nch : 0;
        const int nblk = ((ncod + nch - 1) / nch + 31) >> 5;               // chain A's steps, in blocks
        uint32_t xc = (nch == 2) ? (1u << 31) : 0u;
        if (wq < nch) {
            const int j = wq * sn + lane;
            int term = 0;
            if (lane < sn && j < cnt) {
                const int q = cnt - 1 - j;
                const int n = L * (m + (q / L) * M) + (q % L);
                const int pi = div_wc(sgl, n), pj = n - pi * sgl.wc;
                int sv = (int)planes[sgl.img_off + 2 * sgl.plane + ((long)(2 * pi + sgl.oi) << sgl.lvl) * sgl.W + ((long)(2 * pj + sgl.oj) << sgl.lvl)] + shift;
                if (sv < 0 || sv >= A) { bad = 1; sv = 0; }
                uint32_t mul = 1;
                for (int e = 0; e < lane; ++e) mul *= (uint32_t)A;
                term = (int)((uint32_t)sv * mul);                         // the sum is below A^n <= 2^31
            }
            xc += (uint32_t)__builtin_amdgcn_readlane(wave_incl_scan(term), 63);
        }
        auto fetch_blk = [&](int blk) -> uint32_t {     // lane t < 32: step 32 blk + t of the wavefront's chain
            const int i = 32 * blk + (lane & 31);
            const int q = cnt - 1 - (nch * sn + nch * i + wq);
            return (i < n_own) ? pl[L * (m + (q / L) * M) + (q % L)] : 0u;
        };
        int used = 0, blk = 0, prev_tot = 0;
        uint32_t raw = fetch_blk(0);
        if (wq < 2 && lane == 0) { sh_xs[0][wq][0] = xc; sh_used[0][wq] = 0; }      // (a stream with nothing to code: the start states are the final ones)
        if (nblk > 0)
        for (;; ++blk) {                                // workgroup-uniform
            const uint32_t rawn = fetch_blk(blk + 1);
            if (wq < 2) {                               // (an idle chain B records no steps and reports 0 bits)
                const int nst = min(32, n_own - 32 * blk);
                if (lane == 0) sh_xs[blk & 1][wq][0] = xc;
                uint32_t recv = 0, xsv = 0;             // lane t: step t's record and the state behind it -- written to LDS once per block, not once per step
                for (int t = 0; t < nst; ++t) {
                    const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)raw, t);
                    const uint32_t lo = v & 0xFFFFu;
                    uint32_t hi = v >> 16;
                    if (hi == 0) hi = 0x10000u;
                    uint32_t freq = hi - lo;
                    if (freq == 0 || hi < lo) { bad = 1; freq = 1; }
                    const int nb = rans_emit_bits_any(xc, freq);
                    recv = (lane == t) ? ((xc & ((1u << nb) - 1u)) | ((uint32_t)nb << 16)) : recv;
                    used += nb;
                    xc = rans_push(xc >> nb, lo, freq);
                    xsv = (lane == t) ? xc : xsv;
                }
                if (lane < nst) { sh_fld[nch * (32 * blk + lane) + wq] = recv; sh_xs[blk & 1][wq][lane + 1] = xsv; }
                if (lane == 0) sh_used[blk & 1][wq] = used;
            }
            __syncthreads();
            raw = rawn;
            const bool full_before = blk > 0 && prev_tot + fixed >= GEO::kPayBits;      // the payload was full a block ago: the stopping multiple of 32 lies in this block or the one before
            prev_tot = sh_used[blk & 1][0] + sh_used[blk & 1][1];
            if (full_before || blk + 1 >= nblk) break;
        }
        const int nrun = (nblk > 0) ? min(32 * nch * (blk + 1), ncod) : 0;     // records there are
        // Where the rule stops (tcod) and where each field goes are prefix sums over the records: thread tid takes records
        // 2048 q + 8 tid .. + 7 of quarter q (a long tail of a cheap source has up to four), the chains' bit counts before a quarter carried
        // over from the quarters in front of it.
        auto onb = [&](int e) -> bool { return nch == 2 && (e & 1); };      // record 2048 q + 8 tid + e belongs to chain B (2048 and 8 are even)
        const int nq = (nrun + 2047) >> 11;
        int nbv[8];
        uint32_t fv[8];
        int base_a = 0, base_b = 0;                     // bits of chain A / chain B in the quarters before the current one
        // exclusive prefixes (ea, eb) of this thread's records of quarter q; leaves the quarter's totals in tot_a / tot_b.  Two barriers.
        auto scan_quarter = [&](int q, int &ea, int &eb, int &tot_a, int &tot_b) {
            int sa = 0, sb = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int idx = 2048 * q + 8 * tid + e;
                const uint32_t r = (idx < nrun) ? sh_fld[idx] : 0u;
                nbv[e] = (int)(r >> 16); fv[e] = r & 0xFFFFu;
                if (onb(e)) sb += nbv[e]; else sa += nbv[e];
            }
            const int ia = wave_incl_scan(sa), ib = wave_incl_scan(sb);
            __syncthreads();                            // (the previous use of sh_scan has been read)
            if (lane == 63) { sh_scan[0][wq] = ia; sh_scan[1][wq] = ib; }
            __syncthreads();
            ea = base_a + ia - sa; eb = base_b + ib - sb;
            tot_a = 0; tot_b = 0;
            for (int w2 = 0; w2 < 4; ++w2) {
                if (w2 < wq) { ea += sh_scan[0][w2]; eb += sh_scan[1][w2]; }
                tot_a += sh_scan[0][w2]; tot_b += sh_scan[1][w2];
            }
        };
        if (tid == 0) sh_cut[0] = nrun;
        __syncthreads();
        for (int q = 0; q < nq; ++q) {                  // 1. the first record at a multiple of 32 symbols (from the stream's end) in front of which the arena has reached the payload
            int ea, eb, ta, tb2;
            scan_quarter(q, ea, eb, ta, tb2);
            int ca = ea, cb = eb, first = 0x7FFFFFFF;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int idx = 2048 * q + 8 * tid + e;
                if (((NS + idx) & (kRansTailBlock - 1)) == 0 && ca + cb + fixed >= GEO::kPayBits && idx < nrun && first == 0x7FFFFFFF) first = idx;
                if (onb(e)) cb += nbv[e]; else ca += nbv[e];
            }
            if (first != 0x7FFFFFFF) atomicMin(&sh_cut[0], first);
            base_a += ta; base_b += tb2;
        }
        __syncthreads();
        const int tcod = sh_cut[0];                     // coded symbols
        base_a = 0; base_b = 0;
        if (tid == 0 && tcod == nrun) { sh_cut[1] = -1; sh_cut[2] = -1; }
        for (int q = 0; q < nq; ++q) {                  // 2. the chains' bits up to there
            int ea, eb, ta, tb2;
            scan_quarter(q, ea, eb, ta, tb2);
            int ca = ea, cb = eb;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (2048 * q + 8 * tid + e == tcod) { sh_cut[1] = ca; sh_cut[2] = cb; }
                if (onb(e)) cb += nbv[e]; else ca += nbv[e];
            }
            base_a += ta; base_b += tb2;
        }
        __syncthreads();
        if (tid == 0 && (sh_cut[1] < 0 || nq == 0)) { sh_cut[1] = base_a; sh_cut[2] = base_b; }      // every record is coded: the totals
        __syncthreads();
        const int used_a = sh_cut[1], used_b = sh_cut[2];
        alen = max(GEO::kPayBits, fixed + used_a + used_b);     // the arena: payload ++ spill
        base_a = 0; base_b = 0;
        for (int q = 0; q < nq; ++q) {                  // 3. the fields to their places: chain A's upwards from bit 32 in the decoder's reading order, chain B's below its state
            int ea, eb, ta, tb2;
            scan_quarter(q, ea, eb, ta, tb2);
            int ca = ea, cb = eb;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (2048 * q + 8 * tid + e < tcod) {
                    const int pos = onb(e) ? alen - 32 - used_b + cb : 32 + used_a - ca - nbv[e];
                    lds_or_bits(sh_pay, pos, nbv[e], fv[e]);
                }
                if (onb(e)) cb += nbv[e]; else ca += nbv[e];
            }
            base_a += ta; base_b += tb2;
        }
        if (tid == 0) {
            // the chains' states behind their last coded step: s steps in, i.e. entry s - 32 b of block b = min(s / 32, the last block run)
            auto state_of = [&](int c, int steps) -> uint32_t { const int bb = min(steps >> 5, blk); return sh_xs[bb & 1][c][steps - 32 * bb]; };
            sh_pay[0] = state_of(0, (tcod + nch - 1) / nch);
            if (nch == 2) {
                const uint32_t xb = state_of(1, tcod >> 1);
                lds_or_bits(sh_pay, alen - 32, 16, xb & 0xFFFFu);
                lds_or_bits(sh_pay, alen - 16, 16, xb >> 16);
            } else lds_or_bits(sh_pay, 32 + used_a, 1, 1u);      // one chain: the end marker behind its last field -- the arena's highest set bit
        }
        T = NS + tcod;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < kWin; t += kFlush) sh_win[t + tid] = 0;
        __syncthreads();
        // the spill -- arena bits [7936, alen): whole dwords of sh_pay, the payload being 248 of them -- is the bottom of the main bit region
        if (tid < kRansSpillMax / 32 + 1) sh_win[tid] = sh_pay[GEO::kPayDw + tid];
        __syncthreads();
    } else {
        const uint32_t *pl = pairs + dl.pair_off;
        auto fetch_blk = [&](int q1) -> uint32_t {      // lane t: the t-th symbol from the end of the q1 symbols that are left
            const int q = q1 - 1 - lane;
            return (q >= 0) ? pl[L * (m + (q / L) * M) + (q % L)] : 0u;
        };
        uint32_t xt = 1u << 31;
        int tb = 0;
        uint64_t acc = 0;                               // emitted bits not yet in LDS: the low accn (< 32) bits
        int accn = 0, wdw = 0;
        bool full = false;
        uint32_t raw = fetch_blk(cnt);
        for (int q1 = cnt; q1 > 0 && !full; q1 -= 64) {
            const uint32_t rawn = fetch_blk(q1 - 64);
            const int nblk = min(64, q1);
            for (int t = 0; t < nblk; ++t) {
                if (T >= kRansTailMax) { full = true; break; }
                const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)raw, t);
                const uint32_t lo = v & 0xFFFFu;
                uint32_t hi = v >> 16;
                if (hi == 0) hi = 0x10000u;
                uint32_t freq = hi - lo;
                if (freq == 0 || hi < lo) { bad = 1; freq = 1; }
                if (T == 0) xt = freq << 15;            // absorbing start: the first pushed symbol codes to 2^31 + c_low, no bits
                const int nb = (T == 0) ? 0 : rans_emit_bits(xt, freq);      // (the closed form needs x >= 2^31)
                if (tb + nb + 32 > GEO::kPayBits) { full = true; break; }
                acc |= (uint64_t)(xt & ((1u << nb) - 1u)) << accn;
                accn += nb;
                tb += nb;
                if (accn >= 32) {                        // wave-uniform
                    if (tid == 0) sh_pay[wdw] = (uint32_t)acc;
                    ++wdw; acc >>= 32; accn -= 32;
                }
                xt = rans_push(xt >> nb, lo, freq);
                ++T;
            }
            raw = rawn;
        }
        acc |= (uint64_t)xt << accn;                     // the final state on top: accn + 32 <= 63 bits
        if (tid == 0) { sh_pay[wdw] = (uint32_t)acc; if (accn > 0) sh_pay[wdw + 1] = (uint32_t)(acc >> 32); }
        __syncthreads();
    }
    // 2. the lanes start from the payload (lane l of wavefront wq = stream lane 64 wq + l)
    uint32_t x = (1u << 31) | lds_get_bits(sh_pay, kRansStateBits * tid, kRansStateBits);
    const int tail_from = cnt - T;                  // sequence position (L k + stream lane) of the first tail symbol

    // 3. main coder, last decoded symbol first; bits go UP from bit 0 of the bit region
    int bp = alen - GEO::kPayBits, wbase = 0;       // bit cursor (xwide v4: above the tail's spill); first dword of the staging window (workgroup-uniform)
    int par = 0;                                    // parity of the coded steps (sh_tot)
    // A stage's steps of this stream: K of them (0: the stage has no chunk for stream m), its pairs, the first tail position
    struct Stg { const uint32_t *pp; int n, K, lim; };
    auto stage_of = [&](int st) -> Stg {
        if (st < 0) return Stg{ pairs, 1, 0, 0 };
        const StreamDesc d = desc[(long)st * B + b];
        const int nchunks = (d.n + L - 1) / L;
        return Stg{ pairs + d.pair_off, max(d.n, 1), (nchunks <= m) ? 0 : (nchunks - m + M - 1) / M,
                    (st == LLICTI_NSTREAMS - 1) ? tail_from : 0x7FFFFFFF };
    };
    // Pair loads are unconditional (clamped address); the raw value is masked only where it is consumed.
    auto fetch_of = [&](const Stg &g, int k) -> uint32_t { return g.pp[min(L * (m + max(k, 0) * M) + tid, g.n - 1)]; };
    // The pair loads do not depend on the coder state: two rounds (eight steps) are kept in flight in registers with FIXED roles (a
    // rotating ring makes the compiler copy the newest load, i.e. wait for it with s_waitcnt vmcnt(0) in every round) -- and a stage's
    // first eight are requested before the PREVIOUS stage's rounds run, so that the 45 stages do not each start with a memory round trip.
    Stg sg_cur = stage_of(LLICTI_NSTREAMS - 1);
    uint32_t a0 = fetch_of(sg_cur, sg_cur.K - 1), a1 = fetch_of(sg_cur, sg_cur.K - 2), a2 = fetch_of(sg_cur, sg_cur.K - 3), a3 = fetch_of(sg_cur, sg_cur.K - 4);
    uint32_t b0 = fetch_of(sg_cur, sg_cur.K - 5), b1 = fetch_of(sg_cur, sg_cur.K - 6), b2 = fetch_of(sg_cur, sg_cur.K - 7), b3 = fetch_of(sg_cur, sg_cur.K - 8);
    for (int st = LLICTI_NSTREAMS - 1; st >= 0; --st) {      // rANS is LIFO: last decoded symbol first
        const Stg d = sg_cur;
        const Stg sg_nxt = stage_of(st - 1);
        const uint32_t n0 = fetch_of(sg_nxt, sg_nxt.K - 1), n1 = fetch_of(sg_nxt, sg_nxt.K - 2), n2 = fetch_of(sg_nxt, sg_nxt.K - 3), n3 = fetch_of(sg_nxt, sg_nxt.K - 4);
        const uint32_t n4 = fetch_of(sg_nxt, sg_nxt.K - 5), n5 = fetch_of(sg_nxt, sg_nxt.K - 6), n6 = fetch_of(sg_nxt, sg_nxt.K - 7), n7 = fetch_of(sg_nxt, sg_nxt.K - 8);
        const int K = d.K;
        const int lim = d.lim;
        auto fetch = [&](int k) -> uint32_t { return fetch_of(d, k); };
        // FOUR steps per round.  The state recurrence of a lane -- x -> bits to emit -> push -> x -- does not depend on where the bits
        // go, and it is the only serial chain the coder has (~25 dependent operations a step); WHERE they go needs a prefix sum over
        // the stream's lanes, i.e. for a multi-wavefront stream an LDS exchange behind a barrier -- a ~350-cycle round trip that the
        // per-step form put on that chain (a step took ~2,000 cycles for ~90 instructions).  So a round first runs the recurrence
        // four steps ahead, keeping (bits, count) of each step in registers, then places the four steps' fields with ONE exchange:
        // four independent prefix sums, one 16-byte LDS write per wavefront, one barrier, four pairs of ORs.  No divergent branch: an
        // inactive lane (past the stage's end, a tail symbol, a step below 0) codes the neutral pair (freq 2^16: no bits, push discarded).
        auto round4 = [&](int k, uint32_t raw0, uint32_t raw1, uint32_t raw2, uint32_t raw3) {
            const uint32_t raws[4] = { raw0, raw1, raw2, raw3 };
            uint32_t fld[4];
            int nbs[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {                            // steps k, k - 1, k - 2, k - 3: the recurrence
                const int kk = k - j;
                const int n = L * (m + kk * M) + tid;
                const bool active = kk >= 0 && n < d.n && L * kk + tid < lim;      // (d.n >= 1: a clamped length of an empty stage, which has K = 0 and never gets here)
                const uint32_t v = active ? raws[j] : 0u;            // (lo, c_high) = (0, 2^16 stored as 0): freq 2^16, no bits
                const uint32_t lo = v & 0xFFFFu;
                uint32_t hi = v >> 16;
                hi = (hi == 0) ? 0x10000u : hi;
                uint32_t freq = hi - lo;
                const bool wrong = (freq == 0) | (hi < lo);
                bad = wrong ? 1 : bad;
                freq = wrong ? 1u : freq;
                const int nb = rans_emit_bits(x, freq);
                nbs[j] = nb;
                fld[j] = x & ((1u << nb) - 1u);
                const uint32_t xn = rans_push(x >> nb, lo, freq);
                x = active ? xn : x;
            }
            // placement: the decoder renormalises stream-lane-ascending reading DOWN -- the highest lane's bits lowest, i.e. sub-chunk
            // Q - 1 first -- and step k's bits below step k - 1's
            int incl[4], total[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { incl[j] = wave_incl_scan(nbs[j]); total[j] = __builtin_amdgcn_readlane(incl[j], 63); }
            int below[4] = { 0, 0, 0, 0 }, step_total[4] = { total[0], total[1], total[2], total[3] };
            if constexpr (Q > 1) {
                if (lane == 0) *reinterpret_cast<int4 *>(&sh_tot[par][wq][0]) = make_int4(total[0], total[1], total[2], total[3]);
                lds_barrier();                                       // not __syncthreads(): the pair loads of the next rounds stay in flight
#pragma unroll
                for (int j = 0; j < 4; ++j) step_total[j] = 0;
#pragma unroll
                for (int q2 = 0; q2 < Q; ++q2) {
                    const int4 t4 = *reinterpret_cast<const int4 *>(&sh_tot[par][q2][0]);
                    const int t[4] = { t4.x, t4.y, t4.z, t4.w };
#pragma unroll
                    for (int j = 0; j < 4; ++j) { step_total[j] += t[j]; below[j] += (q2 > wq) ? t[j] : 0; }
                }
                par ^= 1;
            }
            // The ring's lowest kFlush dwords are complete once the cursor has passed them -- and every wavefront's ORs of the PREVIOUS
            // round are done once this round's barrier (above; Q = 1: same wavefront, program order) has been passed: they are written
            // out and zeroed here, one dword per thread, with no barrier of their own (this round's ORs start at bp, beyond them, and end
            // below wbase + 5 kFlush + 2: inside the ring of 8 units; a round adds at most 2 kFlush dwords).
            while (bp - 32 * wbase >= 32 * kFlush) {                 // workgroup-uniform; at most twice
                const int slot_i = (wbase + tid) & (kWin - 1);
                if (wbase + kFlush <= cap_dw) out32[wbase + tid] = sh_win[slot_i]; else bad = 2;
                sh_win[slot_i] = 0;
                wbase += kFlush;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pos = bp + below[j] + (total[j] - incl[j]);        // nb = 0: ORs zeros
                const int sh = pos & 31;
                atomicOr(&sh_win[(pos >> 5) & (kWin - 1)], fld[j] << sh);
                atomicOr(&sh_win[((pos >> 5) + 1) & (kWin - 1)], (uint32_t)(((uint64_t)fld[j] << sh) >> 32));
                bp += step_total[j];
            }
        };
        for (int k = K - 1; k >= 0; k -= 8) {                 // steps k .. k - 7 (those below 0 are no-ops)
            round4(k, a0, a1, a2, a3);
            a0 = fetch(k - 8); a1 = fetch(k - 9); a2 = fetch(k - 10); a3 = fetch(k - 11);
            if (k - 4 >= 0) round4(k - 4, b0, b1, b2, b3);         // workgroup-uniform
            b0 = fetch(k - 12); b1 = fetch(k - 13); b2 = fetch(k - 14); b3 = fetch(k - 15);
        }
        a0 = n0; a1 = n1; a2 = n2; a3 = n3; b0 = n4; b1 = n5; b2 = n6; b3 = n7;
        sg_cur = sg_nxt;
    }
    // 4. the rest of the ring, the 64 Q final states (31 bits each), T | pad (xwide v4: the header field on top of the bit region instead)
    __syncthreads();
    if constexpr (kSeeded<Q>) {
        // 8 bits T / 32 rounded up, 1 bit "one chain", 1 end-marker bit -- ORed into the ring above the main coder's last bit (every wavefront's
        // ORs of the last round are behind the barrier; the ring holds at least 3 flush units beyond the cursor)
        if (tid == 0) {
            const uint32_t f10 = (uint32_t)((T + kRansTailBlock - 1) / kRansTailBlock) | ((uint32_t)tail_single << 8) | (1u << 9);
            const int sh = bp & 31;
            atomicOr(&sh_win[(bp >> 5) & (kWin - 1)], f10 << sh);
            atomicOr(&sh_win[((bp >> 5) + 1) & (kWin - 1)], (uint32_t)(((uint64_t)f10 << sh) >> 32));
        }
        bp += 10;
        __syncthreads();
    }
    const int nbytes = (bp + 7) >> 3;
    sh_pay[tid] = 0;
    __syncthreads();
    lds_or_bits(sh_pay, kRansStateBits * tid, 16, x & 0xFFFFu);
    lds_or_bits(sh_pay, kRansStateBits * tid + 16, kRansStateBits - 16, (x >> 16) & 0x7FFFu);
    __syncthreads();
    const int ndw = (nbytes >> 2) - wbase;                    // whole ring dwords still to write (< 2 kFlush); then 0..3 bytes
    const bool over = (nbytes >> 2) + 1 > cap_dw;
    if (over) bad = 2;
    else {
        for (int t = tid; t < ndw; t += 64 * Q) out32[wbase + t] = sh_win[(wbase + t) & (kWin - 1)];
        if (tid < (nbytes & 3)) slot[4 + (nbytes & ~3) + tid] = (uint8_t)(sh_win[(wbase + ndw) & (kWin - 1)] >> (8 * tid));
        uint8_t *fs = slot + 4 + nbytes;
        for (int t = tid; t < GEO::kPayBytes; t += 64 * Q) fs[t] = (uint8_t)(sh_pay[t >> 2] >> (8 * (t & 3)));
    }
    if (tid == 0) {
        if constexpr (kSeeded<Q>) {
            // the stream is bit region | states: it starts at the region (slot + 4)
            rinfo[2 * sidx] = 4; rinfo[2 * sidx + 1] = (bad == 2) ? 0 : nbytes + GEO::kPayBytes;
        } else {
            const int t16 = (T & 0x7FF) | ((8 * nbytes - bp) << 11);      // pad: unused (zero) bits on top of the region's last byte
            slot[2] = (uint8_t)(t16 & 0xFF); slot[3] = (uint8_t)(t16 >> 8);
            rinfo[2 * sidx] = 2; rinfo[2 * sidx + 1] = (bad == 2) ? 0 : 2 + nbytes + GEO::kPayBytes;      // overflowed slot (never with the plan's sizing): nothing to pack, ENOSPACE is latched
        }
    }
    if (bad) atomicExch(&status[0], bad == 1 ? LLICTI_EFORMAT : LLICTI_ENOSPACE);
}

// decode: parse a stream (copied by rans_unpack_kernel so that its bit region starts at slot + 4, dword aligned; its validated length is in rpos):
// T and the bit cursor, the 64 Q states.  64 / 128 lanes: u16 (T | pad << 11) | bit region | states.  xwide v4: bit region | states, the
// region's highest set bit its end marker, the 9 bits below it the header field (T / 32 rounded up -- T = min(32 field, the stream's share of
// the last stage) -- and the one-chain flag), the main coder's bits below that.
template <int Q>
__global__ __launch_bounds__(64) void rans_init_kernel(const uint8_t *__restrict__ slots, const long *__restrict__ rslot_off,
                                                       const StreamRef *__restrict__ sref, uint32_t *__restrict__ rstate, uint32_t *__restrict__ rpos,
                                                       uint32_t *__restrict__ rtail, int32_t *status, const StageGeom *__restrict__ sglv)
{
    using GEO = RansGeo<Q>;
    const int sidx = blockIdx.x, lane = threadIdx.x;
    const uint8_t *slot = slots + rslot_off[sidx];
    const int n = (int)rpos[sidx];                                 // >= GEO::kMinStream
    bool bad = false;
    int T = 0, single = 0, cur = 0, nbytes;
    if constexpr (kSeeded<Q>) {
        nbytes = n - GEO::kPayBytes;                               // >= 2
        const uint8_t *reg = slot + 4;
        const int lastb = reg[nbytes - 1];
        const int top = 8 * (nbytes - 1) + 31 - __clz(lastb | 1);      // the end marker
        if (lastb == 0 || top < 9) bad = true;
        else {
            const int p0 = top - 9;
            const uint32_t w = (uint32_t)reg[p0 >> 3] | ((uint32_t)reg[(p0 >> 3) + 1] << 8) | ((uint32_t)reg[min((p0 >> 3) + 2, nbytes - 1)] << 16);
            const uint32_t f9 = (w >> (p0 & 7)) & 0x1FFu;
            const StreamRef sr_ = sref[sidx];
            const StageGeom sgl = sglv[sr_.b];                     // the image's last stage: a tail is at most the stream's share of it
            T = min(kRansTailBlock * (int)(f9 & 0xFFu), rans_stream_count(sgl.hc * sgl.wc, sr_.m, sr_.M, GEO::kLanes));
            single = (int)(f9 >> 8);
            cur = p0;
        }
    } else {
        const int t16 = slot[2] | (slot[3] << 8);
        T = t16 & 0x7FF;
        nbytes = n - 2 - GEO::kPayBytes;
        const int pad = (t16 >> 11) & 7;
        if ((t16 >> 14) || nbytes < 0 || (nbytes == 0 && pad)) { bad = true; T = 0; nbytes = max(nbytes, 0); }
        cur = bad ? 0 : 8 * nbytes - pad;
    }
    const uint8_t *fs = slot + 4 + nbytes;
#pragma unroll
    for (int qq = 0; qq < Q; ++qq) {
        const int bpos = kRansStateBits * (64 * qq + lane);
        uint64_t w = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) w |= (uint64_t)fs[min((bpos >> 3) + k, GEO::kPayBytes - 1)] << (8 * k);
        rstate[((long)sidx * Q + qq) * 64 + lane] = (1u << 31) | ((uint32_t)(w >> (bpos & 7)) & 0x7FFFFFFFu);
    }
    if (lane == 0) { rpos[sidx] = (uint32_t)cur; rtail[sidx] = (uint32_t)T | ((uint32_t)single << 16); }
    if (bad && lane == 0) flag_image(status, sref[sidx].b, LLICTI_EFORMAT);
}

// One stage (level, band, colour channel) of all images.  One workgroup of 4 wavefronts per stream (one per
// SIMD: with the stage VALU-issue bound, the busiest SIMD sets the pace, so waves per workgroup is a multiple
// of 4).  Every wave keeps its own copy of the 64 rANS states (the update is cheap and identical in all of
// them), so a step needs ONE barrier.  A wave resolves 16 of the step's 64 symbols, 4 lanes per symbol: every lane of a
// group holds all five mixture components and probes its own table entry.  The symbol is first located with a CHEAP
// approximate CDF (Abramowitz-Stegun 7.1.26 erfc on v_rcp / v_exp, ~0.01 table counts of error) by 5-ary search, then
// PROVEN with the exact spec arithmetic: entry[s] <= slot < entry[s+1] is checked with cdf_entry()'s operations, and if
// the guess is off the exact search gallops away from it and bisects -- so the result is bit-identical to an exact
// search whatever the approximation does.  The 64 (c_low, c_high) pairs meet in a ping-pong LDS buffer, after which
// every wave updates its state copy and renormalises it bit-granularly: lane l needs clz(x) bits, a wave prefix sum
// places them in the stream (read DOWN: the encoder wrote upwards), pulled from a 128-dword register window.
// No table in HBM.
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
// The same term read from a table of the normal CDF (llicti_ctx::d_phi_lut copied into LDS: Phi(z) on [-kPhiLutZ, kPhiLutZ] as (value,
// difference to the next) pairs, linear interpolation, error <= 1e-6): table coordinate by one fma with per-component constants, clamp,
// truncate, fract, one 8-byte LDS read, two fma -- 7 vector operations instead of 15 (two of them quarter-rate).  All three stage decoders'
// hints use it (rans_decode_stage_lane_kernel has the details and the measurements).
struct CompLut { float c1, c0, wn; };
__device__ __forceinline__ CompLut comp_lut(const Comp &c)
{
    constexpr float kS = kPhiLutN / (2.0f * (float)kPhiLutZ);
    CompLut f;
    f.c1 = c.rsig * kS;
    f.c0 = __builtin_fmaf(-c.mu, f.c1, (float)kPhiLutZ * kS);
    f.wn = c.wn;
    return f;
}
__device__ __forceinline__ float term_lut(const CompLut &c, float pt, const float2 *lut)
{
    // u in [0, kPhiLutN): v_med3_f32 returns one of its operands and v_cvt_u32_f32 turns a NaN into 0 -- the index stays inside the table
    const float u = __builtin_amdgcn_fmed3f(__builtin_fmaf(pt, c.c1, c.c0), 0.0f, (float)kPhiLutN - 0.0009765625f);
    const float2 e2 = lut[(uint32_t)u];
    return c.wn * __builtin_fmaf(e2.y, __builtin_amdgcn_fractf(u), e2.x);
}
__device__ __forceinline__ int group_cdf_entry_fast(const CompFast &A, const CompFast &B, float fbase, float scale, int i)
{
    const float pt = div255_exact(fbase + (float)i);     // the exact sample point: near a narrow component the CDF moves by counts per ulp of pt
    return (int)__builtin_rintf(dpp_sum5(term_fast(A, pt), term_fast(B, pt)) * scale) + i;
}

__global__ __launch_bounds__(64 * kRansWaves) void rans_decode_stage_kernel(const float *__restrict__ params, const StageGeom *__restrict__ sgv, const StreamRef *__restrict__ sref,
                                                               const float2 *__restrict__ phi_lut,
                                                               const uint8_t *__restrict__ slots, const long *__restrict__ rslot_off,
                                                               int rslot_cap, uint32_t *__restrict__ rstate, uint32_t *__restrict__ rpos,
                                                               const uint32_t *__restrict__ rtail,
                                                               int16_t *__restrict__ planes, float *__restrict__ fplanes,
                                                               const int32_t *__restrict__ minmax, int last_stage, int32_t *status)
{
    __shared__ uint32_t sh_res[2][64][2];        // ping-pong by step parity: [0] = c_low, [1] = c_high
    __shared__ float2 sh_lut[kPhiLutN];          // the hint's normal CDF (term_lut)
    const int sidx = blockIdx.x;
    const StreamRef sr_ = sref[sidx];               // the stream's image, its index among the image's streams, the image's stream count (images of a call may differ)
    const int b = sr_.b, m = sr_.m, M = sr_.M;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const StageGeom sg = sgv[b];                 // the image's own stage geometry (the images of a call may differ in size)
    const int nc = sg.hc * sg.wc;
    const int nchunks = (nc + 63) >> 6;
    if (nchunks <= m) return;                    // whole workgroup
    for (int j = threadIdx.x; j < kPhiLutN; j += 64 * kRansWaves) sh_lut[j] = phi_lut[j];
    __syncthreads();
    const int K = (nchunks - m + M - 1) / M;
    uint32_t x = rstate[(long)sidx * 64 + lane];                       // every wave: its own copy
    int bcur = (int)rpos[sidx];                                          // bit cursor in the stream's bit region, moving DOWN
    const uint32_t *bitw = reinterpret_cast<const uint32_t *>(slots + rslot_off[sidx] + 4);
    const int max_dw = (rslot_cap - 4) >> 2;
    // Stream bits: register window of 128 dwords below wtop (a multiple of 64): lane l of winA holds dword wtop - 64 + l,
    // of winB dword wtop - 128 + l; bcur stays in (32 (wtop - 64), 32 wtop] and a step consumes at most 1024 bits, pulled
    // with ds_bpermute instead of a dependent global load.
    int wtop = max(64, (((bcur + 31) >> 5) + 63) & ~63);
    auto load_dw = [&](int d0) -> uint32_t { return bitw[min(max(d0 + lane, 0), max_dw - 1)]; };
    uint32_t winA = load_dw(wtop - 64), winB = load_dw(wtop - 128);
    int badx = 0;                                                        // a lane saw a state below 2^15 after its update (clz > 16): malformed stream
    // One pass per colour channel, Y -> Co -> Cg, in ONE launch per band (see rans_decode_stage_pair_kernel): a stream's chunk of Co
    // needs only the Y pixels of the same chunk, decoded by this very wavefront a pass earlier.
    auto pass = [&](auto tag) {
    constexpr int clr = decltype(tag)::value;    // compile-time: no branch (hence no register merge, hence no s_waitcnt vmcnt(0)) next to the prefetch loads
    // the stream's tail symbols (last stage only) are not in the main stream: sequence position 64 k + lane >= tail_from
    const int tail_from = (last_stage && clr == 2) ? rans_stream_count(nc, m, M, 64) - (int)(rtail[sidx] & 0xFFFFu) : 0x7FFFFFFF;
    int minv, maxv, shift;
    clr_range(minmax + 4 * b, clr, minv, maxv, shift);
    const Grid gr = make_grid(minv, maxv);
    const int max_symbol = gr.Lp - 2;
    const float fbase = (float)minv - 0.5f;
    const long img = sg.img_off;
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
        const ParRow32 par = par_row32(params + sg.par_off, 0, (long)sg.h * sg.w, (long)i * sg.w + j);
        r.off = img + ((long)(2 * i + sg.oi) << sg.lvl) * sg.W + ((long)(2 * j + sg.oj) << sg.lvl);
        r.sgA = par[5 * clr + mA]; r.muA = par[16 + 5 * clr + mA]; r.wkA = par[32 + 5 * clr + mA];
        r.sgB = par[5 * clr + 4];  r.muB = par[16 + 5 * clr + 4];  r.wkB = par[32 + 5 * clr + 4];
        r.a0A = r.a1A = r.a0B = r.a1B = r.y = r.co = 0.0f;
        if constexpr (clr == 1) { r.a0A = par[48 + mA]; r.a0B = par[48 + 4]; r.y = fplanes[r.off]; }
        else if constexpr (clr == 2) {
            r.a0A = par[48 + 5 + mA]; r.a1A = par[48 + 10 + mA]; r.a0B = par[48 + 5 + 4]; r.a1B = par[48 + 10 + 4];
            r.y = fplanes[r.off]; r.co = fplanes[r.off + sg.plane];
        }
        r.on = (k < K) && (64 * (m + k * M) + gsym) < nc && (64 * k + gsym) < tail_from;
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
    Raw cur = fetch(0);
    for (int k = 0; k < K; ++k) {
        const int chunk0 = 64 * (m + k * M);
        Raw nxt = fetch(min(k + 1, K - 1));
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
                const CompLut Af = comp_lut(A), Bf = comp_lut(B);
                CompLut F0, F1, F2, F3;
                F0.c1 = quad_bcast<0>(Af.c1); F0.c0 = quad_bcast<0>(Af.c0); F0.wn = quad_bcast<0>(Af.wn);
                F1.c1 = quad_bcast<1>(Af.c1); F1.c0 = quad_bcast<1>(Af.c0); F1.wn = quad_bcast<1>(Af.wn);
                F2.c1 = quad_bcast<2>(Af.c1); F2.c0 = quad_bcast<2>(Af.c0); F2.wn = quad_bcast<2>(Af.wn);
                F3.c1 = quad_bcast<3>(Af.c1); F3.c0 = quad_bcast<3>(Af.c0); F3.wn = quad_bcast<3>(Af.wn);
                int glo = 0, ghi = max_symbol + 1;
                while (ghi - glo > 1) {
                    const int stp = (ghi - glo + 4) / 5;                   // >= 1; the last part is the (smaller) remainder
                    auto probe_at = [&](int j) { return min(glo + stp * (j + 1), ghi - 1); };
                    const int pi = probe_at(mA);
                    const float pt = div255_exact(fbase + (float)pi);     // the exact sample point: near a narrow component the CDF moves by counts per ulp of pt
                    float sum = term_lut(F0, pt, sh_lut);
                    sum += term_lut(F1, pt, sh_lut);
                    sum += term_lut(F2, pt, sh_lut);
                    sum += term_lut(F3, pt, sh_lut);
                    sum += term_lut(Bf, pt, sh_lut);
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
                    fplanes[off + (long)clr * sg.plane] = div255_exact((float)v);
                }
            }
        }
        // (the next step's operands are pinned in their registers here, in front of this step's window reload: see rans_decode_stage_lane_kernel)
        asm volatile("" : "+v"(nxt.sgA), "+v"(nxt.muA), "+v"(nxt.wkA), "+v"(nxt.a0A), "+v"(nxt.a1A), "+v"(nxt.sgB), "+v"(nxt.muB), "+v"(nxt.wkB), "+v"(nxt.a0B), "+v"(nxt.a1B), "+v"(nxt.y), "+v"(nxt.co));
        lds_barrier();      // LDS words only cross here: global loads / stores in flight stay in flight (common.hpp)
        {
            const bool active = chunk0 + lane < nc && 64 * k + lane < tail_from;
            int nb = 0;
            if (active) {
                const uint32_t vlo = sh_res[k & 1][lane][0], vhi = sh_res[k & 1][lane][1];
                x = (vhi - vlo) * (x >> 16) + (x & 0xFFFFu) - vlo;            // in [freq << 15, freq << 16)
                const int lz = __clz((int)x);
                badx |= lz > 16;                                               // only a corrupt stream: the oracle rejects it right here, so flag the image (below)
                nb = min(lz, 16);
            }
            const int incl = wave_incl_scan(nb);
            const int bpos = bcur - incl;                                       // this lane's bits: [bpos, bpos + nb)
            const int d = bpos >> 5;                                           // wtop - 128 <= d < wtop on a well-formed stream
            const uint32_t a0 = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (d & 63), (int)winA);
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (d & 63), (int)winB);
            const uint32_t a1 = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * ((d + 1) & 63), (int)winA);
            const uint32_t b1 = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * ((d + 1) & 63), (int)winB);
            const uint32_t w0 = (d >= wtop - 64) ? a0 : b0;
            const uint32_t w1 = (d + 1 >= wtop - 64) ? a1 : b1;
            const uint32_t bits = __builtin_amdgcn_alignbit(w1, w0, (uint32_t)(bpos & 31)) & ((1u << nb) - 1u);
            x = (x << nb) | bits;
            bcur -= __builtin_amdgcn_readlane(incl, 63);
            if (bcur <= 32 * (wtop - 64) && wtop > 64) { wtop -= 64; winA = winB; winB = load_dw(wtop - 128); }
        }
        cur = nxt;
    }
    };
    pass(std::integral_constant<int, 0>{});
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __syncthreads();
    pass(std::integral_constant<int, 1>{});
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __syncthreads();
    pass(std::integral_constant<int, 2>{});
    if (wave == 0) {
        rstate[(long)sidx * 64 + lane] = x;
        const bool anybad = ballot64(badx != 0) != 0;
        if (lane == 0) {
            rpos[sidx] = (uint32_t)max(bcur, 0);
            if (bcur < 0 || anybad) flag_image(status, b, LLICTI_EFORMAT);     // the stream ran out of bits, or a state fell out of [2^31, 2^32)
        }
    }
}

// Wide streams (Q = 2, 128 lanes): FOUR wavefronts per stream, TWO lanes per symbol.  The stage is bound by vector instructions per
// symbol (a CU's issue saturates at two wavefronts per SIMD), and four lanes per symbol spend them on overhead: four 5-ary rounds
// are 16 mixture evaluations per symbol where a pair's six ternary rounds are 12, and a wave's per-step fixed costs (fetch, the
// state update of all 128 lanes) are shared by 32 symbols instead of 16.  A pair's lanes each own three mixture components
// (0,1,2 / 2,3,4), prepare them as mix_prepare() does and swap the results (quad_perm [1,0,3,2]); both then hold all five in
// canonical order.  Hint: ternary search on the approximate CDF, one probe per lane.  Proof: lane 0 evaluates the exact entry s,
// lane 1 entry s + 1, each all five terms in the spec's order -- no cross-lane sum.  Bit-identical to the four-lane kernel.
// One launch decodes a band's three stages (Y, Co, Cg passes).
__device__ __forceinline__ float pair_swap(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
}
__device__ __forceinline__ uint32_t pair_swap_u(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}

__global__ __launch_bounds__(64 * kRansWaves) void rans_decode_stage_pair_kernel(const float *__restrict__ params, const StageGeom *__restrict__ sgv, const StreamRef *__restrict__ sref,
                                                               const float2 *__restrict__ phi_lut,
                                                               const uint8_t *__restrict__ slots, const long *__restrict__ rslot_off,
                                                               int rslot_cap, uint32_t *__restrict__ rstate, uint32_t *__restrict__ rpos,
                                                               const uint32_t *__restrict__ rtail,
                                                               int16_t *__restrict__ planes, float *__restrict__ fplanes,
                                                               const int32_t *__restrict__ minmax, int last_stage, int32_t *status)
{
    constexpr int Q = 2, L = 64 * Q;
    __shared__ uint32_t sh_res[2][L][2];         // ping-pong by step parity: [0] = c_low, [1] = c_high
    __shared__ float2 sh_lut[kPhiLutN];          // the hint's normal CDF (term_lut)
    const int sidx = blockIdx.x;
    const StreamRef sr_ = sref[sidx];               // the stream's image, its index among the image's streams, the image's stream count (images of a call may differ)
    const int b = sr_.b, m = sr_.m, M = sr_.M;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const StageGeom sg = sgv[b];                 // the image's own stage geometry (the images of a call may differ in size)
    const int nc = sg.hc * sg.wc;
    const int nchunks = (nc + L - 1) / L;
    if (nchunks <= m) return;                    // whole workgroup
    for (int j = threadIdx.x; j < kPhiLutN; j += 64 * kRansWaves) sh_lut[j] = phi_lut[j];
    __syncthreads();
    const int K = (nchunks - m + M - 1) / M;
    uint32_t x[Q];                                                      // every wave: its own copy of the stream's 128 states
#pragma unroll
    for (int qq = 0; qq < Q; ++qq) x[qq] = rstate[((long)sidx * Q + qq) * 64 + lane];
    int bcur = (int)rpos[sidx];                                          // bit cursor in the stream's bit region, moving DOWN
    const uint32_t *bitw = reinterpret_cast<const uint32_t *>(slots + rslot_off[sidx] + 4);
    const int max_dw = (rslot_cap - 4) >> 2;
    int wtop = max(64, (((bcur + 31) >> 5) + 63) & ~63);
    auto load_dw = [&](int d0) -> uint32_t { return bitw[min(max(d0 + lane, 0), max_dw - 1)]; };
    uint32_t winA = load_dw(wtop - 64), winB = load_dw(wtop - 128);
    int badx = 0;                                                        // a lane saw a state below 2^15 after its update (clz > 16): malformed stream
    // One pass per colour channel, Y -> Co -> Cg: a stream's chunk of Co needs only the Y pixels of the same chunk (the cross-channel
    // mean update reads the SAME position, LLICTI_nets.py:474-477), which this very wavefront decoded a pass earlier -- so the band's
    // three stages are one launch (15 per decode instead of 45: a launch's ramp-up and its wait for the slowest stream are paid once).
    // The channel is a compile-time constant of the pass: no branch next to the prefetch loads.
    auto pass = [&](auto tag) {
    constexpr int clr = decltype(tag)::value;
    const int tail_from = (last_stage && clr == 2) ? rans_stream_count(nc, m, M, L) - (int)(rtail[sidx] & 0xFFFFu) : 0x7FFFFFFF;
    int minv, maxv, shift;
    clr_range(minmax + 4 * b, clr, minv, maxv, shift);
    const Grid gr = make_grid(minv, maxv);
    const int max_symbol = gr.Lp - 2;
    const float fbase = (float)minv - 0.5f;
    const long img = sg.img_off;
    const int pl = lane & 1;                                            // lane of the pair
    const bool odd = pl != 0;
    const int gsym = 32 * wave + (lane >> 1);                           // stream lane (0 .. 127) this pair resolves
    const int gbit = lane & ~1;                                         // ballot bit of the pair's first lane
    const bool head = !odd;
    const int c0 = 2 * pl;                                              // the lane's own components: c0, c0 + 1, c0 + 2
    struct Raw { float sg[3], mu[3], wk[3], a0[3], a1[3], y, co; long off; bool on; };
    auto fetch = [&](int k) -> Raw {
        Raw r;
        const int n = min(L * (m + k * M) + gsym, nc - 1);           // clamped: the loads are unconditional
        const int i = div_wc(sg, n), j = n - i * sg.wc;
        const ParRow32 par = par_row32(params + sg.par_off, 0, (long)sg.h * sg.w, (long)i * sg.w + j);
        r.off = img + ((long)(2 * i + sg.oi) << sg.lvl) * sg.W + ((long)(2 * j + sg.oj) << sg.lvl);
        r.y = r.co = 0.0f;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            r.sg[t] = par[5 * clr + c0 + t]; r.mu[t] = par[16 + 5 * clr + c0 + t]; r.wk[t] = par[32 + 5 * clr + c0 + t];
            r.a0[t] = r.a1[t] = 0.0f;
            if constexpr (clr == 1) r.a0[t] = par[48 + c0 + t];
            else if constexpr (clr == 2) { r.a0[t] = par[48 + 5 + c0 + t]; r.a1[t] = par[48 + 10 + c0 + t]; }
        }
        if constexpr (clr == 1) r.y = fplanes[r.off];
        else if constexpr (clr == 2) { r.y = fplanes[r.off]; r.co = fplanes[r.off + sg.plane]; }
        r.on = (k < K) && (L * (m + k * M) + gsym) < nc && (L * k + gsym) < tail_from;
        return r;
    };
    // own[0..2] of the two lanes -> the five components in canonical order (component 2 is computed by both)
    auto canon = [&](const float own[3], float out[5]) {
        const float r0 = pair_swap(own[0]), r1 = pair_swap(own[1]), r2 = pair_swap(own[2]);
        out[0] = odd ? r0 : own[0];
        out[1] = odd ? r1 : own[1];
        out[2] = odd ? own[0] : own[2];
        out[3] = odd ? own[1] : r1;
        out[4] = odd ? own[2] : r2;
    };
    Raw cur = fetch(0);
    for (int k = 0; k < K; ++k) {
        const int chunk0 = L * (m + k * M);
        Raw nxt = fetch(min(k + 1, K - 1));
        const uint32_t xs = (wave < 2) ? x[0] : x[1];
        const uint32_t slot = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (gsym & 63), (int)xs) & 0xFFFFu;
        if (cur.on) {                            // uniform within the pair
            const long off = cur.off;
            // the lane's three components as mix_prepare() has them
            float mu3[3], rs3[3], w3[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                float mu = cur.mu[t];
                if constexpr (clr == 1) { const float tt = cur.a0[t] * cur.y; mu = mu + tt; }
                else if constexpr (clr == 2) { const float t1 = cur.a0[t] * cur.y; const float t2 = cur.a1[t] * cur.co; const float tt = t1 + t2; mu = mu + tt; }
                mu3[t] = mu;
                rs3[t] = 1.0f / ((cur.sg[t] > kScaleBound) ? cur.sg[t] : kScaleBound);
                w3[t] = (cur.wk[t] > kWeightBound) ? cur.wk[t] : kWeightBound;
            }
            float w5[5], mu5[5], rs5[5], wn5[5];
            canon(w3, w5);
            const float ssum = (((w5[0] + w5[1]) + w5[2]) + w5[3]) + w5[4];
            const float den = 1e-9f + ssum;
            float wn3[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) wn3[t] = w3[t] / den;
            canon(mu3, mu5); canon(rs3, rs5); canon(wn3, wn5);
            CompLut F[5];
#pragma unroll
            for (int c = 0; c < 5; ++c) { Comp cc; cc.mu = mu5[c]; cc.rsig = rs5[c]; cc.wn = wn5[c]; F[c] = comp_lut(cc); }

            // 1. hint: ternary search on the approximate table, one probe per lane of the pair
            int glo = 0, ghi = max_symbol + 1;
            while (ghi - glo > 1) {
                const int stp = (ghi - glo + 2) / 3;                    // >= 1
                const int q1 = min(glo + stp, ghi - 1), q2 = min(glo + 2 * stp, ghi - 1);
                const int pi = odd ? q2 : q1;
                const float pt = div255_exact(fbase + (float)pi);
                float sum = term_lut(F[0], pt, sh_lut);
                sum += term_lut(F[1], pt, sh_lut);
                sum += term_lut(F[2], pt, sh_lut);
                sum += term_lut(F[3], pt, sh_lut);
                sum += term_lut(F[4], pt, sh_lut);
                const int e = (int)__builtin_rintf(sum * gr.scale) + pi;
                const uint64_t bal = ballot64(e <= (int)slot);
                const int np = __builtin_popcount((uint32_t)(bal >> gbit) & 0x3u);      // ordered probes: the passes form a prefix
                const int nlo = (np == 0) ? glo : (np == 1) ? q1 : q2;
                const int nhi = (np == 0) ? q1 : (np == 1) ? q2 : ghi;
                glo = nlo; ghi = nhi;
            }
            // exact table entry i, all five terms in this lane, cdf_entry()'s operations in its order
            auto entry_exact = [&](int i) -> uint32_t {
                const float pt = sample_pt(gr, i);
                const float t0 = wn5[0] * (0.5f * erfc_spec_nobranch(kNegRsqrt2 * ((pt - mu5[0]) * rs5[0])));
                const float t1 = wn5[1] * (0.5f * erfc_spec_nobranch(kNegRsqrt2 * ((pt - mu5[1]) * rs5[1])));
                const float t2 = wn5[2] * (0.5f * erfc_spec_nobranch(kNegRsqrt2 * ((pt - mu5[2]) * rs5[2])));
                const float t3 = wn5[3] * (0.5f * erfc_spec_nobranch(kNegRsqrt2 * ((pt - mu5[3]) * rs5[3])));
                const float t4 = wn5[4] * (0.5f * erfc_spec_nobranch(kNegRsqrt2 * ((pt - mu5[4]) * rs5[4])));
                const float acc = (((t0 + t1) + t2) + t3) + t4;
                return (uint32_t)((int)__builtin_rintf(acc * gr.scale) + i) & 0xFFFFu;
            };
            // 2. proof: lane 0 evaluates entry glo, lane 1 entry glo + 1; if the hint is off, gallop away from it and bisect
            int lo = 0, hi = max_symbol + 1;
            uint32_t vlo = 0, vhi = 0x10000u;
            bool have_lo = false, have_hi = false;
            {
                const int s1 = glo, s2 = min(glo + 1, max_symbol);
                const uint32_t eM = entry_exact(odd ? s2 : s1);
                const uint32_t eO = pair_swap_u(eM);
                const uint32_t eA = odd ? eO : eM, eB = odd ? eM : eO;
                const bool leA = (s1 == 0) || (eA <= slot);              // entry 0 is the floor of the search (torchac: left = 0)
                const bool leB = (s1 + 1 <= max_symbol) && (eB <= slot); // past the top symbol: c_high = 0x10000 by definition
                if (leA) {
                    lo = s1; vlo = eA; have_lo = true;
                    if (leB) { lo = s2; vlo = eB; }
                    else if (s1 + 1 <= max_symbol) { hi = s2; vhi = eB; have_hi = true; }
                } else { hi = s1; vhi = eA; have_hi = true; }
            }
            int step = 2;
            while (hi - lo > 1) {                                        // the same probe in both lanes (uniform in the pair)
                int probe;
                if (have_lo && have_hi) probe = (lo + hi) >> 1;
                else if (have_lo) { probe = min(lo + step, hi - 1); step <<= 1; }
                else { probe = max(hi - step, lo + 1); step <<= 1; }
                const uint32_t e = entry_exact(probe);
                if (e <= slot) { lo = probe; vlo = e; have_lo = true; } else { hi = probe; vhi = e; have_hi = true; }
            }
            if (!have_lo) vlo = entry_exact(0);
            if (head) {
                sh_res[k & 1][gsym][0] = vlo;
                sh_res[k & 1][gsym][1] = vhi;
                const int v = lo - shift;
                planes[off + (long)clr * sg.plane] = (int16_t)v;
                fplanes[off + (long)clr * sg.plane] = div255_exact((float)v);
            }
        }
        // (the next step's operands are pinned in their registers here, in front of this step's window reload: see rans_decode_stage_lane_kernel)
#pragma unroll
        for (int t = 0; t < 3; ++t) asm volatile("" : "+v"(nxt.sg[t]), "+v"(nxt.mu[t]), "+v"(nxt.wk[t]), "+v"(nxt.a0[t]), "+v"(nxt.a1[t]));
        asm volatile("" : "+v"(nxt.y), "+v"(nxt.co));
        lds_barrier();      // LDS words only cross here: global loads / stores in flight stay in flight (common.hpp)
        {
            int below = 0;                                                         // bits of the lower sub-chunk of this step
#pragma unroll
            for (int qq = 0; qq < Q; ++qq) {
                const int sl = 64 * qq + lane;                                     // stream lane
                const bool active = chunk0 + sl < nc && L * k + sl < tail_from;
                int nb = 0;
                if (active) {
                    const uint32_t vlo = sh_res[k & 1][sl][0], vhi = sh_res[k & 1][sl][1];
                    x[qq] = (vhi - vlo) * (x[qq] >> 16) + (x[qq] & 0xFFFFu) - vlo;  // in [freq << 15, freq << 16)
                    const int lz = __clz((int)x[qq]);
                    badx |= lz > 16;                                               // only a corrupt stream: the oracle rejects it right here, so flag the image (below)
                    nb = min(lz, 16);
                }
                const int incl = wave_incl_scan(nb);
                const int bpos = bcur - below - incl;                              // this lane's bits: [bpos, bpos + nb)
                const int d = bpos >> 5;                                           // wtop - 128 <= d < wtop on a well-formed stream
                const uint32_t a0 = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (d & 63), (int)winA);
                const uint32_t b0 = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (d & 63), (int)winB);
                const uint32_t a1 = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * ((d + 1) & 63), (int)winA);
                const uint32_t b1 = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * ((d + 1) & 63), (int)winB);
                const uint32_t w0 = (d >= wtop - 64) ? a0 : b0;
                const uint32_t w1 = (d + 1 >= wtop - 64) ? a1 : b1;
                const uint32_t bits = __builtin_amdgcn_alignbit(w1, w0, (uint32_t)(bpos & 31)) & ((1u << nb) - 1u);
                x[qq] = (x[qq] << nb) | bits;
                below += __builtin_amdgcn_readlane(incl, 63);
            }
            bcur -= below;
            if (bcur <= 32 * (wtop - 64) && wtop > 64) { wtop -= 64; winA = winB; winB = load_dw(wtop - 128); }
        }
        cur = nxt;
    }
    };
    // Between passes: the pixels a wavefront stored are loaded again by the SAME wavefront (same lanes, same positions), through the
    // same CU's L1 / the same XCD's L2 -- a workgroup-scope fence orders them (an agent-scope __threadfence() writes the L2 back: +0.45 ms
    // per decode); the barrier frees the ping-pong result buffers.
    pass(std::integral_constant<int, 0>{});
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __syncthreads();
    pass(std::integral_constant<int, 1>{});
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __syncthreads();
    pass(std::integral_constant<int, 2>{});
    if (wave == 0) {
#pragma unroll
        for (int qq = 0; qq < Q; ++qq) rstate[((long)sidx * Q + qq) * 64 + lane] = x[qq];
        const bool anybad = ballot64(badx != 0) != 0;
        if (lane == 0) {
            rpos[sidx] = (uint32_t)max(bcur, 0);
            if (bcur < 0 || anybad) flag_image(status, b, LLICTI_EFORMAT);     // the stream ran out of bits, or a state fell out of [2^31, 2^32)
        }
    }
}

// XWIDE streams (Q = 4, 256 lanes): ONE lane per symbol, one wavefront per 64 stream lanes -- four per workgroup, one per SIMD.
// The stage decoders are bound by vector instructions ISSUED per symbol (a lone wavefront per SIMD issues one every ~4 cycles whatever
// it does), and lanes that share a symbol spend them on each other: the pair kernel runs 12 approximate mixture evaluations (six ternary
// rounds x two lanes) and two preparations per symbol, this kernel 9 (binary search) and one; the two exact entries c_low, c_high are the
// same ten erfc either way.  ~26 instead of ~41 issued instructions per symbol.  What makes one lane per symbol possible without halving
// the wavefronts per workgroup is the lane count of the STREAM (bytes are per stream, 0.06 bit per lane), and what makes it cheap is the
// channel-planar `params` layout: a wavefront's 64 symbols are 64 consecutive positions, so every parameter load is one coalesced
// 256-byte access whose address is scalar base + lane position.
// Each wavefront owns the states of its 64 stream lanes (no copies): a step's symbols never leave their lane -- slot, search, c_low /
// c_high and the state update all happen in it -- and the only exchange is the four wavefronts' bit totals (LDS, one barrier per step),
// from which every lane knows where its bits start.  The stream's bits come from an LDS ring of 512 dwords (a step takes at most
// 128), refilled 128 dwords at a time: requested at the end of a step, stored at the end of the next, used after the barrier that follows.
template <int Q>
__global__ __launch_bounds__(64 * Q) void rans_decode_stage_lane_kernel(const float *__restrict__ params, const StageGeom *__restrict__ sgv, const StreamRef *__restrict__ sref,
                                                               const uint8_t *__restrict__ slots, const long *__restrict__ rslot_off,
                                                               int rslot_cap, uint32_t *__restrict__ rstate, uint32_t *__restrict__ rpos,
                                                               const uint32_t *__restrict__ rtail,
                                                               int16_t *__restrict__ planes, float *__restrict__ fplanes,
                                                               const int32_t *__restrict__ minmax, int last_stage, int32_t *status,
                                                               const float2 *__restrict__ phi_lut)
{
    constexpr int L = 64 * Q;
    constexpr int kRing = 512, kRefill = 128;    // dwords; a step consumes at most 16 L bits = L / 2 dwords
    static_assert(L / 2 <= kRefill && kRing >= 4 * kRefill && 64 * Q >= kRefill, "ring sizing");
    __shared__ uint32_t sh_ring[kRing];          // stream dword d at sh_ring[d & (kRing - 1)]
    __shared__ int sh_tot[2][Q];                 // a step's bit totals per wavefront (ping-pong by step parity)
    // The hint's mixture evaluations are 45 of the ~55 a symbol costs (nine probes x five components), and on the vector unit each is 15
    // operations, two of them quarter-rate (rcp, exp2).  The normal CDF of ONE variable is a table: Phi(z) on [-6, 6] in kPhiLutN steps as
    // (value, difference to the next) pairs (the context's, computed once on the host in double), 16 KB of LDS, read with linear
    // interpolation -- error <= h^2 / 8 max|Phi''| = 1e-6, a fifteenth of a count of the 16-bit table (the vector form's Abramowitz-Stegun
    // fit: 1.5e-7; either way the hint is proved or corrected by exact entries).  A term is then fma (table coordinate), clamp, floor,
    // convert, min, fract, one 8-byte LDS read, two fma: ~1,220 instead of ~1,510 vector instructions per symbol, stage launches -6 %.
    constexpr int kLutN = kPhiLutN;
    constexpr float kLutZ = (float)kPhiLutZ, kLutS = kLutN / (2.0f * kLutZ);
    __shared__ float2 sh_lut[kLutN];
    const int sidx = blockIdx.x;
    const StreamRef sr_ = sref[sidx];               // the stream's image, its index among the image's streams, the image's stream count (images of a call may differ)
    const int b = sr_.b, m = sr_.m, M = sr_.M;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;      // tid = lane of the stream
    const StageGeom sg = sgv[b];                 // the image's own stage geometry (the images of a call may differ in size)
    const int nc = sg.hc * sg.wc;
    const int nchunks = (nc + L - 1) / L;
    if (nchunks <= m) return;                    // whole workgroup
    for (int j = tid; j < kLutN; j += L) sh_lut[j] = phi_lut[j];
    const int K = (nchunks - m + M - 1) / M;
    uint32_t x = rstate[(long)sidx * L + tid];
    int bcur = (int)rpos[sidx];                  // bit cursor in the stream's bit region, moving DOWN
    const uint32_t *bitw = reinterpret_cast<const uint32_t *>(slots + rslot_off[sidx] + 4);
    const int max_dw = (rslot_cap - 4) >> 2;
    auto load_dw = [&](int d) -> uint32_t { return bitw[min(max(d, 0), max_dw - 1)]; };
    // ring window [wlo, wlo + kRing): everything a step can touch, dwords (bcur >> 5) - L / 2 .. (bcur >> 5) + 1, with two steps of slack below
    int wlo = ((((bcur >> 5) + 2) + kRefill - 1) & ~(kRefill - 1)) - kRing;
    for (int t = tid; t < kRing; t += L) sh_ring[(wlo + t) & (kRing - 1)] = load_dw(wlo + t);
    uint32_t pf = 0;                             // a refill in flight: dwords [wlo, wlo + kRefill) of the NEW wlo, lanes 0 .. kRefill - 1
    bool pend = false;
    int badx = 0;                                // a lane saw a state below 2^15 after its update (clz > 16): malformed stream
    int par = 0;                                 // step parity over the whole launch (sh_tot)
    __syncthreads();

    auto pass = [&](auto tag) {
    constexpr int clr = decltype(tag)::value;    // compile-time: no branch next to the prefetch loads
    const int tail_from = (last_stage && clr == 2) ? rans_stream_count(nc, m, M, L) - (int)(rtail[sidx] & 0xFFFFu) : 0x7FFFFFFF;
    int minv, maxv, shift;
    clr_range(minmax + 4 * b, clr, minv, maxv, shift);
    const Grid gr = make_grid(minv, maxv);
    const int max_symbol = gr.Lp - 2;
    const float fbase = (float)minv - 0.5f;
    const long img = sg.img_off;
    const long npos = (long)sg.h * sg.w;
    struct Raw { float sg[5], mu[5], wk[5], a0[5], a1[5], y, co; long off; bool on; };
    auto fetch = [&](int k) -> Raw {
        Raw r;
        const int n = min(L * (m + k * M) + tid, nc - 1);            // clamped: the loads are unconditional
        const int i = div_wc(sg, n), j = n - i * sg.wc;
        const ParRow32 prw = par_row32(params + sg.par_off, 0, npos, (long)i * sg.w + j);
        r.off = img + ((long)(2 * i + sg.oi) << sg.lvl) * sg.W + ((long)(2 * j + sg.oj) << sg.lvl);
        r.y = r.co = 0.0f;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            r.sg[t] = prw[5 * clr + t]; r.mu[t] = prw[16 + 5 * clr + t]; r.wk[t] = prw[32 + 5 * clr + t];
            r.a0[t] = r.a1[t] = 0.0f;
            if constexpr (clr == 1) r.a0[t] = prw[48 + t];
            else if constexpr (clr == 2) { r.a0[t] = prw[48 + 5 + t]; r.a1[t] = prw[48 + 10 + t]; }
        }
        if constexpr (clr == 1) r.y = fplanes[r.off];
        else if constexpr (clr == 2) { r.y = fplanes[r.off]; r.co = fplanes[r.off + sg.plane]; }
        r.on = (k < K) && (L * (m + k * M) + tid) < nc && (L * k + tid) < tail_from;
        return r;
    };
    Raw cur = fetch(0);
    for (int k = 0; k < K; ++k) {
        const uint32_t slot = x & 0xFFFFu;
        uint32_t vlo = 0, vhi = 0x10000u;
        // the five components as mix_prepare() has them -- computed for every lane (a clamped position's values where the lane is off):
        // that makes the raw CNN outputs dead right here, so the NEXT step's are loaded into the same registers now and have the whole
        // search to arrive (no second register set, no copies)
        float mu5[5], rs5[5], wn5[5], w5[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            float mu = cur.mu[t];
            if constexpr (clr == 1) { const float tt = cur.a0[t] * cur.y; mu = mu + tt; }
            else if constexpr (clr == 2) { const float t1 = cur.a0[t] * cur.y; const float t2 = cur.a1[t] * cur.co; const float tt = t1 + t2; mu = mu + tt; }
            mu5[t] = mu;
            rs5[t] = 1.0f / ((cur.sg[t] > kScaleBound) ? cur.sg[t] : kScaleBound);
            w5[t] = (cur.wk[t] > kWeightBound) ? cur.wk[t] : kWeightBound;
        }
        const long off_k = cur.off;
        const bool on_k = cur.on;
        cur = fetch(min(k + 1, K - 1));
        if (on_k) {
            const float ssum = (((w5[0] + w5[1]) + w5[2]) + w5[3]) + w5[4];
            const float den = 1e-9f + ssum;
            float c1[5], c0[5];                  // table coordinate of component t at sample point pt: u = pt c1 + c0
#pragma unroll
            for (int t = 0; t < 5; ++t) { wn5[t] = w5[t] / den; c1[t] = rs5[t] * kLutS; c0[t] = __builtin_fmaf(-mu5[t], c1[t], kLutZ * kLutS); }

            // 1. hint: binary search on the approximate table (probes 1 .. max_symbol: regular sample points only)
            int glo = 0, ghi = max_symbol + 1;
            while (ghi - glo > 1) {
                const int pi = (glo + ghi) >> 1;
                const float pt = div255_exact(fbase + (float)pi);
                float sum = 0.0f;
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    // u in [0, kLutN) -- v_med3_f32 returns one of its operands, so a NaN coordinate comes out as NaN, 0 or the bound, and v_cvt_u32_f32
                    // (which truncates) turns a NaN into 0: the index stays inside the table whatever the CNN produced; v_fract_f32 is what is left
                    const float u = __builtin_amdgcn_fmed3f(__builtin_fmaf(pt, c1[t], c0[t]), 0.0f, (float)kLutN - 0.0009765625f);
                    const float2 e2 = sh_lut[(uint32_t)u];
                    sum = __builtin_fmaf(wn5[t], __builtin_fmaf(e2.y, __builtin_amdgcn_fractf(u), e2.x), sum);
                }
                const int e = (int)__builtin_rintf(sum * gr.scale) + pi;
                const bool le = e <= (int)slot;
                glo = le ? pi : glo;
                ghi = le ? ghi : pi;
            }
            // exact table entry i, cdf_entry()'s operations in its order
            auto entry_exact = [&](int i) -> uint32_t {
                const float pt = sample_pt(gr, i);
                const float t0 = wn5[0] * (0.5f * erfc_spec_nobranch(kNegRsqrt2 * ((pt - mu5[0]) * rs5[0])));
                const float t1 = wn5[1] * (0.5f * erfc_spec_nobranch(kNegRsqrt2 * ((pt - mu5[1]) * rs5[1])));
                const float t2 = wn5[2] * (0.5f * erfc_spec_nobranch(kNegRsqrt2 * ((pt - mu5[2]) * rs5[2])));
                const float t3 = wn5[3] * (0.5f * erfc_spec_nobranch(kNegRsqrt2 * ((pt - mu5[3]) * rs5[3])));
                const float t4 = wn5[4] * (0.5f * erfc_spec_nobranch(kNegRsqrt2 * ((pt - mu5[4]) * rs5[4])));
                const float acc = (((t0 + t1) + t2) + t3) + t4;
                return (uint32_t)((int)__builtin_rintf(acc * gr.scale) + i) & 0xFFFFu;
            };
            // 2. the two entries the state update needs anyway, entry[s] and entry[s + 1], are also the proof of the hint;
            //    if it is off, gallop away from it and bisect with exact entries (bit-identical to an exact search whatever the hint was)
            int lo = 0, hi = max_symbol + 1;
            bool have_lo = false, have_hi = false;
            {
                const int s1 = glo, s2 = min(glo + 1, max_symbol);
                const uint32_t eA = entry_exact(s1), eB = entry_exact(s2);
                const bool leA = (s1 == 0) || (eA <= slot);              // entry 0 is the floor of the search (torchac: left = 0)
                const bool leB = (s1 + 1 <= max_symbol) && (eB <= slot); // past the top symbol: c_high = 0x10000 by definition
                if (leA) {
                    lo = s1; vlo = eA; have_lo = true;
                    if (leB) { lo = s2; vlo = eB; }
                    else if (s1 + 1 <= max_symbol) { hi = s2; vhi = eB; have_hi = true; }
                } else { hi = s1; vhi = eA; have_hi = true; }
            }
            int step = 1;                        // (a hint that is off is off by one: the first gallop probe is the neighbour)
            while (hi - lo > 1) {
                int probe;
                if (have_lo && have_hi) probe = (lo + hi) >> 1;
                else if (have_lo) { probe = min(lo + step, hi - 1); step <<= 1; }
                else { probe = max(hi - step, lo + 1); step <<= 1; }
                const uint32_t e = entry_exact(probe);
                if (e <= slot) { lo = probe; vlo = e; have_lo = true; } else { hi = probe; vhi = e; have_hi = true; }
            }
            if (!have_lo) vlo = entry_exact(0);
            const int v = lo - shift;
            planes[off_k + (long)clr * sg.plane] = (int16_t)v;
            fplanes[off_k + (long)clr * sg.plane] = div255_exact((float)v);
        }
        // The next step's CNN outputs were requested ~8,000 cycles ago and are pinned in their registers HERE, in front of the ring's refill
        // load: left alone, the compiler moves them into the loop-carried registers at the bottom of the loop, behind that load and the
        // pixel stores, and waits for them with s_waitcnt vmcnt(0) -- in-order counting makes that a wait for the refill load just issued
        // and for the stores (a memory round trip per step: most of the 22 % of wave cycles this kernel spent parked).
#pragma unroll
        for (int t = 0; t < 5; ++t) asm volatile("" : "+v"(cur.sg[t]), "+v"(cur.mu[t]), "+v"(cur.wk[t]), "+v"(cur.a0[t]), "+v"(cur.a1[t]));
        asm volatile("" : "+v"(cur.y), "+v"(cur.co));
        // state update of this lane, bit-granular renormalisation: lane l of the stream takes its clz(x) bits below those of lanes < l
        int nb = 0;
        if (on_k) {
            x = (vhi - vlo) * (x >> 16) + slot - vlo;                          // in [freq << 15, freq << 16)
            const int lz = __clz((int)x);
            badx |= lz > 16;                                                   // only a corrupt stream: the oracle rejects it too
            nb = min(lz, 16);
        }
        const int incl = wave_incl_scan(nb);
        if (lane == 63) sh_tot[par][wave] = incl;
        lds_barrier();                  // (also: last step's ring refill is visible)  not __syncthreads(): the pixel stores above and the next step's parameter loads stay in flight
        int below = 0, step_total = 0;
#pragma unroll
        for (int q2 = 0; q2 < Q; ++q2) { const int t2 = sh_tot[par][q2]; step_total += t2; below += (q2 < wave) ? t2 : 0; }
        par ^= 1;
        {
            const int bpos = bcur - below - incl;                              // this lane's bits: [bpos, bpos + nb)
            const int d = bpos >> 5;
            const uint32_t w0 = sh_ring[d & (kRing - 1)], w1 = sh_ring[(d + 1) & (kRing - 1)];
            const uint32_t bits = __builtin_amdgcn_alignbit(w1, w0, (uint32_t)(bpos & 31)) & ((1u << nb) - 1u);
            x = (x << nb) | bits;
        }
        bcur -= step_total;
        // ring: store the refill requested a step ago (its slots held dwords >= wlo + kRing: above anything this step's readers touch),
        // then request the next one if fewer than two steps of dwords are left below the cursor
        if (pend) { if (tid < kRefill) sh_ring[(wlo + tid) & (kRing - 1)] = pf; pend = false; }
        if ((bcur >> 5) - 2 * kRefill < wlo) { wlo -= kRefill; if (tid < kRefill) pf = load_dw(wlo + tid); pend = true; }
    }
    };
    // Between passes: the pixels a lane stored are loaded again by the SAME lane (same position) -- a workgroup-scope fence orders them
    pass(std::integral_constant<int, 0>{});
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    pass(std::integral_constant<int, 1>{});
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    pass(std::integral_constant<int, 2>{});
    rstate[(long)sidx * L + tid] = x;
    const int anybad = __syncthreads_or(badx);
    if (tid == 0) {
        rpos[sidx] = (uint32_t)max(bcur, 0);
        if (bcur < 0 || anybad) flag_image(status, b, LLICTI_EFORMAT);     // the stream ran out of bits, or a state fell out of [2^31, 2^32)
    }
}

// After the last stage the 64 Q states of a stream ARE its tail stream (31 bits each, the tail coder's final state on top,
// its leading one the highest set bit).  The T tail symbols -- all of the last stage's Cg channel (level 0, band x10) -- come out
// of ONE coder state, serially; a lone wavefront issues one instruction every ~4 cycles whatever it does, so everything that does not
// depend on the coder state is done by other wavefronts of the workgroup:
//   * kTailAhead wavefronts per chain PREPARE symbols: the position's CNN outputs, the component's mean / 1 / sigma / normalised
//     weight exactly as mix_prepare() has them, the approximate mixture at the 64 anchors 8 l (Lp <= 512) -- and, round 5, a
//     SPECULATED WINDOW: the twelve exact table entries around the mixture's median (where the approximate table crosses 2^15).
//     Round r + 1's symbols are prepared (into the other half of a ping-pong LDS buffer) while
//   * wavefront 0 DECODES round r's: if the slot lies inside the speculated window the symbol is proved by one compare, a ballot and
//     two v_readlane, and the state update runs on the scalar unit (the state is wave-uniform); otherwise the anchors pick a bucket,
//     12 x 5 lanes evaluate the 12 exact entries around it (lane = 5 e + mc: mixture component mc of window entry e) and a ballot
//     proves the symbol (an exact 13-ary search takes over when that hint is wrong too).  Either way the symbol is the one an exact
//     search of the whole row returns: the entries are the same bits wherever they are computed.  The renormalisation's bits come
//     from two payload dwords requested at the top of the symbol (they depend on the cursor only); the round's pixels are stored
//     together.
// What the window buys depends on the source: on uniform noise (12.8 bits per symbol, tails of ~620 symbols) it is not offered (the
// median's bucket holds a fiftieth of the mass) and nothing changes; on a source as cheap as the reference's trained model (1.7 bits
// per symbol: tails of ~4,700 symbols, all on one chain) the tail launch of 24 x 10 xwide streams goes from 2.70 to 1.29 ms, 64-lane
// streams 0.66 -> 0.34 (profiles/r5/tail_speculation.json: coder alone 1.2 ms, preparing wavefronts alone 1.2 ms).
// One barrier per round.  Checks: the main region was read to its last bit, the tail state ends at its start state (freq << 15 of
// the symbol the tail encoder began with; 2^31 when T = 0) with no bit left.  Xwide streams (two seeded chains, above): a second set of
// wavefronts runs chain B; the start states are seeds -- checked to be below A^n, with zero digits where the stream has no symbol -- the
// chains' cursors must not have crossed and the payload between them must be zero.  A ONE-chain xwide stream (cheap symbols: the second
// chain would cost its final state) leaves chain B's wavefronts idle: its preparing wavefronts join chain A's, a round is 2 kTailAhead symbols.
#ifndef TAIL_AHEAD
#define TAIL_AHEAD 4                            // (build switch for A/B runs: 2 / 3 / 4 / 5 measured, profiles/r5/tail_speculation.json)
#endif
#ifndef TAIL_SPEC
#define TAIL_SPEC 1                             // the preparing wavefronts offer a speculated window (0: the coder always evaluates its own -- the kernel of rounds 3-4)
#endif
constexpr int kTailAhead = TAIL_AHEAD;           // symbols per round = preparing wavefronts

template <int Q> constexpr int kTailChains = kSeeded<Q> ? 2 : 1;      // xwide: two chains, each with its own coder + preparing wavefronts

template <int Q>
__global__ __launch_bounds__(64 * (1 + kTailAhead) * kTailChains<Q>) void rans_tail_kernel(const float *__restrict__ params, const StageGeom *__restrict__ sgv, const StreamRef *__restrict__ sref,
                                                       const uint32_t *__restrict__ rstate, const uint32_t *__restrict__ rpos,
                                                       const uint32_t *__restrict__ rtail,
                                                       int16_t *__restrict__ planes, float *__restrict__ fplanes,
                                                       const int32_t *__restrict__ minmax, int32_t *status,
                                                       const uint8_t *__restrict__ slots, const long *__restrict__ rslot_off)
{
    using GEO = RansGeo<Q>;
    constexpr int L = 64 * Q, NCH = kTailChains<Q>;
    constexpr int kSpillDw = kSeeded<Q> ? kRansSpillMax / 32 + 1 : 0;      // xwide v4: the arena is the payload ++ the spill the main decoder left at the bottom of its region
    __shared__ uint32_t sh_pay[64 * Q + 2 + kSpillDw];
    constexpr int NSL = NCH * kTailAhead;               // prepared symbols per round and buffer: slot = chain kTailAhead + i
    __shared__ float sh_cmp[2][NSL][16];                // [0..4] mu, [5..9] 1 / sigma, [10..14] normalised weight of the five components
    __shared__ int sh_e1[2][NSL][64];                   // approximate table entry at anchor 8 l
    __shared__ long sh_off[2][NSL];                     // the symbol's pixel
    __shared__ int sh_spec[2][NSL][16];                 // [0..11] the exact entries of the speculated window, [12] its first index (-1: none)
    __shared__ int sh_cur[2];                           // xwide: where the two chains stopped reading
    const int sidx = blockIdx.x;
    const StreamRef sr_ = sref[sidx];               // the stream's image, its index among the image's streams, the image's stream count (images of a call may differ)
    const int b = sr_.b, m = sr_.m, M = sr_.M;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (scalar: roles, chains and symbol counts are wave-uniform, their branches scalar)
    // role 0: the chain's coder.  A workgroup's wavefronts go to the four SIMDs round robin: with two chains the coders are wavefronts 0 and 1 (their
    // own SIMD each, shared with one preparing wavefront), not 0 and 4 (the same SIMD, taking turns at its issue port)
    const int role = wave / NCH;
    const StageGeom sg = sgv[b];                 // the image's own stage geometry (the images of a call may differ in size)
    const int nc = sg.hc * sg.wc;
    const int cnt = rans_stream_count(nc, m, M, L);
    const int rt = (int)(rtail[sidx] & 0xFFFFu);
    const int nch = (kSeeded<Q> && !(rtail[sidx] >> 16)) ? 2 : 1;      // chains this stream's tail was coded with (xwide: its flag says one or two)
    // A one-chain xwide stream (well-predicted content: the second chain would cost its final state) has twice the symbols on its chain and a
    // second set of wavefronts with nothing to do: the idle chain's preparing wavefronts join chain 0's, and a round is 2 kTailAhead symbols.
    const bool pool = NCH == 2 && nch == 1;
    const int chain = (pool && role != 0) ? 0 : wave % NCH;
    const int SA = pool ? NSL : kTailAhead;             // symbols of a chain per round = its preparing wavefronts
    // 64 / 128 lanes: the main region must have been read to its last bit.  xwide v4: what is left of it IS the tail coder's spill (its length the cursor)
    const int E = kSeeded<Q> ? (int)rpos[sidx] : 0;
    bool bad = (kSeeded<Q> ? E >= kRansSpillMax : rpos[sidx] != 0) || rt > cnt;
    const int alen = GEO::kPayBits + (bad ? 0 : E);     // the arena
    const int Tall = min(rt, cnt);                      // the stream's tail symbols: the coded ones, then (xwide) the seeds'
    int minv, maxv, shift;
    clr_range(minmax + 4 * b, 2, minv, maxv, shift);
    const Grid gr = make_grid(minv, maxv);
    const int max_symbol = gr.Lp - 2;
    uint32_t pw = 1;
    const int ns = kSeeded<Q> ? rans_seed_count(max_symbol + 1, pw) : 0;
    const int sn = kSeeded<Q> ? (nch == 2 ? ns : 1) : 0;     // raw symbols in a chain's start state (xwide v4: n per chain of two, the stream's last symbol for one chain)
    const int NS = min(nch * sn, cnt);                  // seed symbols (0 for the older stream kinds)
    bad = bad || Tall < NS || (nch == 2 && cnt < 2 * ns);
    const bool live = chain < nch;                      // (the second set of wavefronts idles through a one-chain stream's rounds)
    const int Tc = max(Tall - NS, 0);                   // coded symbols; the one with index idx (j = nch ns + idx from the stream's end) is on chain idx % nch
    const int T = live ? (Tc + nch - 1 - chain) / nch : 0;      // this chain's
    const int R = ((Tc + nch - 1) / nch + SA - 1) / SA;      // rounds (chain 0 has the most symbols)
    const long img = sg.img_off;
    const int mc = lane % 5, we = lane / 5;
    auto pixel_of = [&](int j) -> long {                 // the stream's j-th symbol from its end
        const int q = min(max(cnt - 1 - j, 0), max(cnt - 1, 0));
        const int n = min(L * (m + (q / L) * M) + (q % L), nc - 1);
        const int pi = div_wc(sg, n), pj = n - pi * sg.wc;
        return ((long)pi << 32) | (uint32_t)pj;
    };

    if (role != 0) {
        // ---- preparing wavefronts: symbol t = SA r + i of round r (t counts the chain's symbols in decoding order), prepared into slot sl
        const int i = pool ? wave - NCH : role - 1;
        const int sl = chain * kTailAhead + i;
        struct Row { float sg, mu, wk, bb, dd, y, co; long off; };
        auto fetch = [&](int t) -> Row {
            Row r;
            const long pp = pixel_of(nch * sn + chain + nch * (T - 1 - t));
            const int pi = (int)(pp >> 32), pj = (int)(uint32_t)pp;
            const ParRow src = par_row(params + sg.par_off, 0, (long)sg.h * sg.w, (long)pi * sg.w + pj);
            r.off = img + ((long)(2 * pi + sg.oi) << sg.lvl) * sg.W + ((long)(2 * pj + sg.oj) << sg.lvl);
            r.sg = src[10 + mc]; r.mu = src[16 + 10 + mc]; r.wk = src[32 + 10 + mc];      // the Cg channel's sigma, mu, weight ...
            r.bb = src[48 + 5 + mc]; r.dd = src[48 + 10 + mc];                                // ... and its cross-channel factors
            r.y = fplanes[r.off]; r.co = fplanes[r.off + sg.plane];
            return r;
        };
        auto prepare = [&](const Row &row, int buf) {
            // component mc, exactly as mix_prepare() does
            const float t1 = row.bb * row.y;
            const float t2 = row.dd * row.co;
            const float tt = t1 + t2;
            const float mu = row.mu + tt;
            const float rsig = 1.0f / ((row.sg > kScaleBound) ? row.sg : kScaleBound);
            const float w = (row.wk > kWeightBound) ? row.wk : kWeightBound;
            float wk5[5], mu5[5], rs5[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) wk5[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w), k));
            const float ssum = (((wk5[0] + wk5[1]) + wk5[2]) + wk5[3]) + wk5[4];
            const float wn = w / (1e-9f + ssum);
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                mu5[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mu), k));
                rs5[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rsig), k));
                wk5[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wn), k));
            }
            // the approximate mixture (all five components in every lane) at anchor 8 l
            const int i1 = min(8 * lane, max_symbol);
            const float pt1 = sample_pt(gr, i1);
            float sum = 0.0f;
#pragma unroll
            for (int k = 0; k < 5; ++k) { Comp ck; ck.mu = mu5[k]; ck.rsig = rs5[k]; ck.wn = wk5[k]; sum += term_fast(comp_fast(ck), pt1); }
            const int e1v = (int)__builtin_rintf(sum * gr.scale) + i1;
            sh_e1[buf][sl][lane] = e1v;
            if (lane < 5) { sh_cmp[buf][sl][lane] = mu; sh_cmp[buf][sl][5 + lane] = rsig; sh_cmp[buf][sl][10 + lane] = wn; }
            if (lane == 0) sh_off[buf][sl] = row.off;
            // The speculated window.  The coder's per-symbol chain is slot -> bucket -> 12 exact entries (an erfc and five cross-lane
            // moves) -> ballot -> state update, ~2,500 cycles of which the exact entries are most -- and they depend on the coder state
            // only through WHERE the window lies.  On a well-predicted source (the reference's trained model spends 1.7 bits per symbol of
            // this channel) that is nearly always around the mixture's median, which is known here: the anchor bucket in which the
            // approximate table crosses 2^15, refined linearly.  So the window's twelve entries are computed in THIS wavefront, with
            // entry_at()'s operations in its order, and the coder only has to look: slot inside [entry 0, entry 11) proves the symbol
            // exactly as its own window would (same entries, same ballot); outside, it falls back to its bucket search.  Not offered when
            // the median's bucket holds less than an eighth of the mass (noise: the window would miss nine times in ten).
            const int lst = __builtin_amdgcn_readfirstlane(__builtin_popcountll(ballot64(lane == 0 || (8 * lane <= max_symbol && e1v <= 0x8000))) - 1);
            const int eL = __builtin_amdgcn_readlane(e1v, lst);
            const int eH = (lst < 63 && 8 * (lst + 1) <= max_symbol) ? __builtin_amdgcn_readlane(e1v, min(lst + 1, 63)) : 0x10000;
            int wbs = -1;
            if (TAIL_SPEC && eH - eL >= 0x2000) {         // (wave-uniform)
                const float fr = 8.0f * (float)(0x8000 - eL) * __builtin_amdgcn_rcpf((float)(eH - eL));
                wbs = max(8 * lst + (int)__builtin_amdgcn_fmed3f(fr, 0.0f, 8.0f) - 5, 0);
                const int idx = wbs + min(we, 11);
                const float ptx = sample_pt(gr, min(idx, gr.Lp - 1));
                const float term = wn * (0.5f * erfc_spec_nobranch(kNegRsqrt2 * ((ptx - mu) * rsig)));
                const int g0 = 4 * 5 * min(we, 11);
                const float a0 = __int_as_float(__builtin_amdgcn_ds_bpermute(g0, __float_as_int(term)));
                const float a1 = __int_as_float(__builtin_amdgcn_ds_bpermute(g0 + 4, __float_as_int(term)));
                const float a2 = __int_as_float(__builtin_amdgcn_ds_bpermute(g0 + 8, __float_as_int(term)));
                const float a3 = __int_as_float(__builtin_amdgcn_ds_bpermute(g0 + 12, __float_as_int(term)));
                const float a4 = __int_as_float(__builtin_amdgcn_ds_bpermute(g0 + 16, __float_as_int(term)));
                const float acc = (((a0 + a1) + a2) + a3) + a4;
                const float qf = __builtin_rintf(acc * gr.scale);
                const uint32_t ent = (idx <= max_symbol) ? ((uint32_t)((int)qf + idx) & 0xFFFFu) : 0x10000u;
                if (mc == 0 && we < 12) sh_spec[buf][sl][we] = (int)ent;
            }
            if (lane == 0) sh_spec[buf][sl][12] = wbs;
        };
        Row rowA = fetch(i);
        Row rowB = fetch(SA + i);
        prepare(rowA, 0);
        __syncthreads();                                  // (payload assembled by wavefront 0)
        __syncthreads();                                  // round 0 is prepared
        for (int r = 0; r < R; ++r) {
            const Row rowC = fetch(SA * (r + 2) + i);            // two rounds ahead: the loads run under a whole round
            if (SA * (r + 1) + i < T) prepare(rowB, (r + 1) & 1);
            rowB = rowC;
            lds_barrier();      // LDS words only cross here: global loads / stores in flight stay in flight (common.hpp)
        }
        if constexpr (kSeeded<Q>) lds_barrier();          // (the coders exchange their cursors)
        return;
    }

    // ---- role 0: the chain's coder (wavefront 0 assembles the payload for both)
    if (wave == 0) {
#pragma unroll
        for (int qq = 0; qq < Q; ++qq) sh_pay[64 * qq + lane] = 0;
        if (lane < 2) sh_pay[64 * Q + lane] = 0;
        __builtin_amdgcn_wave_barrier();                  // same wavefront, in-order LDS
#pragma unroll
        for (int qq = 0; qq < Q; ++qq) {
            const uint32_t xl = rstate[((long)sidx * Q + qq) * 64 + lane] & 0x7FFFFFFFu;
            lds_or_bits(sh_pay, kRansStateBits * (64 * qq + lane), 16, xl & 0xFFFFu);
            lds_or_bits(sh_pay, kRansStateBits * (64 * qq + lane) + 16, kRansStateBits - 16, xl >> 16);
        }
        if constexpr (kSeeded<Q>) {
            // the spill: bits [0, E) of the stream's bit region (slot + 4), behind the payload's 248 dwords; what lies above it in the region is not the tail's
            if (lane < kSpillDw) {
                const uint32_t w = reinterpret_cast<const uint32_t *>(slots + rslot_off[sidx] + 4)[lane];
                const int nb = min(max(alen - GEO::kPayBits - 32 * lane, 0), 32);
                sh_pay[GEO::kPayDw + lane] = nb >= 32 ? w : (w & ((1u << nb) - 1u));
            }
        }
    }
    __syncthreads();
    uint32_t xt;
    int tc;                                               // bit cursor: legacy chains and xwide chain B read DOWN to it, xwide chain A reads UP from it
    int a_end = alen;                                     // xwide, one chain: its end marker (the bits it may read end there)
    if constexpr (kSeeded<Q>) {
        // final states at fixed places: chain A's in bits [0, 32), chain B's in the arena's top 32 bits (the state is wave-uniform: kept scalar)
        xt = !live ? (1u << 31) : (uint32_t)__builtin_amdgcn_readfirstlane((int)(chain ? lds_get_bits(sh_pay, alen - 32, 32) : sh_pay[0]));
        tc = chain ? alen - (live ? 32 : 0) : 32;
        if (nch == 2) { if (!(xt >> 31)) { bad = true; xt |= 1u << 31; } }
        else {
            // one chain: the arena's highest set bit above the state is the chain's end marker (a spill ends with it)
            int top = -1;
            for (int d0 = 64 * ((GEO::kPayDw + kSpillDw - 1) / 64); d0 >= 0 && top < 0; d0 -= 64) {
                const int d = d0 + lane;
                const uint32_t w = (d >= 1 && d < GEO::kPayDw + kSpillDw) ? sh_pay[d] : 0u;
                const uint64_t nz = ballot64(w != 0);
                if (nz) {
                    const int hl = 63 - __clzll((long long)nz);
                    top = 32 * (d0 + hl) + 31 - __clz((int)__builtin_amdgcn_readlane((int)w, hl));
                }
            }
            if (top < 32 || (alen > GEO::kPayBits && top != alen - 1)) { bad = true; top = 32; }
            a_end = top;
        }
    } else {
        int top = -1;                                                          // the payload's highest set bit
#pragma unroll
        for (int qq = Q - 1; qq >= 0; --qq) {
            const uint64_t nz = ballot64(sh_pay[64 * qq + lane] != 0);
            if (top < 0 && nz) {
                const int hd = 64 * qq + 63 - __clzll((long long)nz);
                top = 32 * hd + 31 - __clz((int)sh_pay[hd]);
            }
        }
        // a malformed stream still gets its T tail pixels written (from whatever state there is): the output of a flagged image
        // must not depend on what the workspace held
        if (top < 31) bad = true;
        xt = (top >= 31) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_get_bits(sh_pay, top - 31, 32)) : (1u << 31);
        tc = (top >= 31) ? top - 31 : 0;
    }
    __syncthreads();                                      // round 0 is prepared
    for (int r = 0; r < R; ++r) {
        const int buf = r & 1;
        int vsym = 0;
        for (int i = 0; i < SA && SA * r + i < T; ++i) {
            const int sl = chain * kTailAhead + i;
            const uint32_t slot = xt & 0xFFFFu;
            // 0. the window the preparing wavefront speculated on (around the mixture's median), if it offered one: its twelve exact entries are there
            //    already -- lane k < 12 reads entry k, lane 12 the window's first index -- and the symbol is the last entry <= slot, proved when its
            //    successor is in the window too.  Everything from here to the next state is wave-uniform and meant for the scalar unit: ballot,
            //    count, two v_readlane, the state update.
            const int wv = sh_spec[buf][sl][min(lane, 12)];
            // the <= 16 bits the renormalisation will take lie in two dwords that only depend on the cursor: requested here, next to the window, so that
            // the LDS round trip runs under the search instead of behind the state update
            const bool up = kSeeded<Q> && chain == 0;            // xwide chain A reads UP from its cursor, everything else DOWN to it
            const int wpos = min(max(up ? tc : tc - 16, 0), alen);               // (a corrupt stream's cursor stays inside the arena's array)
            const uint32_t bw0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh_pay[wpos >> 5]);
            const uint32_t bw1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh_pay[(wpos >> 5) + 1]);
            auto take_bits = [&](int pos, int n) -> uint32_t {   // bits [pos, pos + n) of the payload, n <= 16, inside the window
                const uint64_t w = ((uint64_t)bw1 << 32) | bw0;
                return (uint32_t)(w >> (pos - (wpos & ~31))) & ((1u << n) - 1u);
            };
            int wb = __builtin_amdgcn_readlane(wv, 12), np = 0;
            // (entries past the top symbol are 2^16, never <= slot; entry 0 is the floor of the search: counted whatever it holds)
            if (wb >= 0) np = __builtin_popcountll((ballot64((uint32_t)wv <= slot) & 0xFFFull) | (wb == 0 ? 1ull : 0ull));
            uint32_t vlo, vhi;
            if (np >= 1 && np <= 11) {
                vlo = (uint32_t)__builtin_amdgcn_readlane(wv, np - 1);
                vhi = (uint32_t)__builtin_amdgcn_readlane(wv, np);
            } else {
                const float mu = sh_cmp[buf][sl][mc], rsig = sh_cmp[buf][sl][5 + mc], wn = sh_cmp[buf][sl][10 + mc];
                const int e1 = sh_e1[buf][sl][lane];
                // exact entry idx (uniform in the lane's group of five): valid in every lane of the group
                auto entry_at = [&](int idx) -> uint32_t {
                    const float pt = sample_pt(gr, min(idx, gr.Lp - 1));
                    const float term = wn * (0.5f * erfc_spec_nobranch(kNegRsqrt2 * ((pt - mu) * rsig)));
                    const int g0 = 4 * 5 * min(we, 11);
                    const float a0 = __int_as_float(__builtin_amdgcn_ds_bpermute(g0, __float_as_int(term)));
                    const float a1 = __int_as_float(__builtin_amdgcn_ds_bpermute(g0 + 4, __float_as_int(term)));
                    const float a2 = __int_as_float(__builtin_amdgcn_ds_bpermute(g0 + 8, __float_as_int(term)));
                    const float a3 = __int_as_float(__builtin_amdgcn_ds_bpermute(g0 + 12, __float_as_int(term)));
                    const float a4 = __int_as_float(__builtin_amdgcn_ds_bpermute(g0 + 16, __float_as_int(term)));
                    const float acc = (((a0 + a1) + a2) + a3) + a4;
                    const float qf = __builtin_rintf(acc * gr.scale);
                    return (idx <= max_symbol) ? ((uint32_t)((int)qf + idx) & 0xFFFFu) : 0x10000u;      // past the top symbol: c_high = 2^16
                };
                // 1. hint: the bucket of 8 entries the prepared anchors put the slot in
                const uint64_t p1 = ballot64(lane == 0 || (8 * lane <= max_symbol && e1 <= (int)slot));
                wb = max(8 * (__builtin_popcountll(p1) - 1) - 2, 0);
                // 2. proof: the 12 exact entries wb .. wb + 11; the symbol is the last one <= slot, its successor must be in the window too
                uint32_t ent = entry_at(wb + we);
                const uint64_t pw12 = ballot64(mc == 0 && we < 12 && wb + we <= max_symbol && (ent <= slot || wb + we == 0));
                np = __builtin_popcountll(pw12);
                if (np == 0 || np == 12) {
                    // the hint was wrong (only absurd mixtures get here): exact 13-ary search from scratch, then the window at the result
                    int lo = 0, hi = max_symbol + 1;
                    while (hi - lo > 1) {
                        const int stp = (hi - lo + 12) / 13;
                        const int pi = min(lo + stp * (min(we, 11) + 1), hi - 1);
                        const uint32_t e = entry_at(pi);
                        const uint64_t pb = ballot64(mc == 0 && we < 12 && e <= slot);
                        const int k = __builtin_popcountll(pb);                        // probes are ordered: the passes form a prefix
                        const int nlo = (k > 0) ? min(lo + stp * k, hi - 1) : lo;
                        const int nhi = (k < 12) ? min(lo + stp * (k + 1), hi - 1) : hi;
                        lo = nlo; hi = nhi;
                    }
                    wb = lo;
                    ent = entry_at(wb + we);
                    np = 1;
                }
                vlo = (uint32_t)__builtin_amdgcn_readlane((int)ent, 5 * (np - 1));
                vhi = (uint32_t)__builtin_amdgcn_readlane((int)ent, 5 * np);
            }
            vsym = (lane == i) ? wb + np - 1 - shift : vsym;      // the round's pixels are stored together, lane i symbol i (one vector store per round instead of a masked block per symbol)
            xt = (vhi - vlo) * (xt >> 16) + slot - vlo;
            if constexpr (kSeeded<Q>) {
                int nb = __clz((int)xt);
                if (nch == 2) {
                    const int avail = chain ? tc - 32 : alen - 32 - tc;      // (never into the other chain's state; whether the chains crossed is checked at the end)
                    if (nb > 16 || avail < nb) { bad = true; nb = max(min(nb, min(avail, 16)), 0); }   // corrupt: keep going on what is there
                    if (chain) tc -= nb;
                    xt = ((xt << nb) | take_bits(tc, nb)) | (1u << 31);
                    if (!chain) tc += nb;
                } else {
                    // one chain: it started small and emitted nothing until its state had grown -- so once the bits below the end marker are used
                    // up the decoder is in that silent start: it takes what is left (the encoder's first emission: < 16 bits) and then nothing
                    const int avail = a_end - tc;
                    if (nb > avail) nb = max(avail, 0);
                    else if (nb > 16) bad = true;                     // bits were left, so the encoder's state was in [2^31, 2^32): corrupt
                    if (nb > 16) { bad = true; nb = 16; }
                    xt = (xt << nb) | take_bits(tc, nb);
                    tc += nb;
                }
            } else if (SA * r + i == T - 1) bad = bad || xt != (vhi - vlo) << 15;      // the encoder's first symbol: absorbing start, no bits
            else {
                int nb = __clz((int)xt);
                if (nb > 16 || tc < nb) { bad = true; nb = max(min(nb, min(tc, 16)), 0); }  // corrupt: keep going on what is there
                tc -= nb;
                xt = ((xt << nb) | take_bits(tc, nb)) | (1u << 31);
            }
        }
        if (lane < SA && SA * r + lane < T) {
            const long off = sh_off[buf][chain * kTailAhead + lane];
            planes[off + 2 * sg.plane] = (int16_t)vsym;
            fplanes[off + 2 * sg.plane] = div255_exact((float)vsym);      // (an integer in [-255, 255]: bit-identical to the division, numerics.hpp)
        }
        lds_barrier();      // LDS words only cross here: global loads / stores in flight stay in flight (common.hpp)
    }
    if constexpr (kSeeded<Q>) {
        // the chain is back at its start state.  Two chains: 2^31 | its n seed symbols in radix A (lane i: digit i = the stream's (chain n + i)-th symbol
        // from the end).  One chain: the stream's last symbol itself (zero if the stream has none), every bit below the end marker read.
        const uint32_t A = (uint32_t)(max_symbol + 1), v = (nch == 2) ? (xt & 0x7FFFFFFFu) : xt;
        bad = bad || (live && (nch == 2 ? v >= pw : (v >= A || tc != a_end)));
        uint32_t div = 1;
        for (int e = 0; e < min(lane, sn); ++e) div *= A;
        const int dg = (lane < sn) ? (int)((v / div) % A) : 0;
        const int j = chain * sn + lane;
        if (live && lane < sn && j < NS) {
            const long pp = pixel_of(j);
            const int pi = (int)(pp >> 32), pj = (int)(uint32_t)pp;
            const long off = img + ((long)(2 * pi + sg.oi) << sg.lvl) * sg.W + ((long)(2 * pj + sg.oj) << sg.lvl);
            const int pv = dg - shift;
            planes[off + 2 * sg.plane] = (int16_t)pv;
            fplanes[off + 2 * sg.plane] = div255_exact((float)pv);
        }
        bad = bad || ballot64(live && lane < sn && j >= NS && dg != 0) != 0;      // digits of symbols the stream does not have
        if (lane == 0) sh_cur[chain] = tc;
        lds_barrier();
        if (chain == 0 && nch == 2) {
            // the chains must not have crossed, and what lies between them is zero
            const int ca = sh_cur[0], cb = sh_cur[1];
            uint32_t nz = 0;
            for (int d = lane; d < GEO::kPayDw + kSpillDw; d += 64) {
                const int lo = max(ca, 32 * d), hi = min(cb, 32 * d + 32);
                if (lo < hi) nz |= sh_pay[d] & ((hi - lo == 32) ? 0xFFFFFFFFu : (((1u << (hi - lo)) - 1u) << (lo - 32 * d)));
            }
            bad = bad || ca > cb || ballot64(nz != 0) != 0;
        }
        if (bad && lane == 0) flag_image(status, b, LLICTI_EFORMAT);
    } else {
        if (T == 0) bad = bad || xt != (1u << 31);
        if ((bad || tc != 0) && lane == 0) flag_image(status, b, LLICTI_EFORMAT);
    }
}

__global__ __launch_bounds__(256) void rans_pack_kernel(const uint8_t *__restrict__ slots, const long *__restrict__ rslot_off,
                                                        const int32_t *__restrict__ rinfo, const StreamRef *__restrict__ sref, const ImgGeo *__restrict__ iv,
                                                        uint8_t *__restrict__ out, long out_stride, int32_t *__restrict__ seg_len, int32_t *status,
                                                        const unsigned long long *__restrict__ ssum, const StreamDesc *__restrict__ desc, int B)
{
    const StreamRef sr_ = sref[blockIdx.x];
    const int b = sr_.b, m = sr_.m, M = image_stream_count(ssum, iv, desc, B, sr_.b, sr_.M), s0 = sr_.sbase;      // (s0 + k: stream k of this image)
    const int hdr_bytes = iv[b].hdr_bytes;
    if (ssum && m == 0 && threadIdx.x == 0) {
        // "auto": the header was written before the count was picked -- the pad field's bits 10 .. 15 say how many streams the image has
        uint8_t *oh = out + (long)b * out_stride;
        const int pf = (iv[b].padint & 0x3FF) | (M << 10);              // (auto counts are <= 32: the field holds them as they are)
        oh[15] = (uint8_t)(pf & 0xFF); oh[16] = (uint8_t)((pf >> 8) & 0xFF);
    }
    if (m >= M) {                                                       // a stream the image did not get: an empty segment
        if (threadIdx.x == 0) seg_len[(long)b * LLICTI_NSEG + 4 + m] = 0;
        return;
    }
    const int G = rans_group(M), sg = m / G;
    long dst = hdr_bytes + (G > 1 ? 4L * G * (sg + 1) : 0);
    for (int k = 0; k < m; ++k) dst += rinfo[2 * (s0 + k) + 1];
    const int n = rinfo[2 * (s0 + m) + 1];
    if (dst + n > out_stride) { if (threadIdx.x == 0) atomicExch(&status[0], LLICTI_ENOSPACE); return; }
    const uint8_t *src = slots + rslot_off[s0 + m] + rinfo[2 * (s0 + m)];
    uint8_t *o = out + (long)b * out_stride + dst;
    block_copy_bytes(o, src, n);
    if (threadIdx.x == 0) {
        if (G == 1) seg_len[(long)b * LLICTI_NSEG + 4 + m] = n;
        else if (m % G == 0) {                                  // the segment's length table and total
            uint8_t *tab = o - 4 * G;
            int tot = 4 * G;
            for (int k = 0; k < G; ++k) {
                const int len = rinfo[2 * (s0 + m + k) + 1];
                tab[4 * k] = (uint8_t)(len & 0xFF); tab[4 * k + 1] = (uint8_t)((len >> 8) & 0xFF);
                tab[4 * k + 2] = (uint8_t)((len >> 16) & 0xFF); tab[4 * k + 3] = (uint8_t)((len >> 24) & 0xFF);
                tot += len;
            }
            seg_len[(long)b * LLICTI_NSEG + 4 + sg] = tot;
        }
        if (m == 0) for (int k = 4 + M / G; k < LLICTI_NSEG; ++k) seg_len[(long)b * LLICTI_NSEG + k] = 0;
    }
}

__global__ __launch_bounds__(256) void rans_unpack_kernel(const uint8_t *__restrict__ in, long in_stride, const int32_t *__restrict__ seg_len,
                                                          const StreamRef *__restrict__ sref, int min_stream, uint8_t *__restrict__ slots, const long *__restrict__ rslot_off,
                                                          int rslot_cap, uint32_t *__restrict__ rpos, int32_t *status, int stream_off)
{
    const StreamRef sr_ = sref[blockIdx.x];
    const int b = sr_.b, m = sr_.m, M = sr_.M, s0 = sr_.sbase;
    const int G = rans_group(M), sg = m / G;
    const int32_t *sl = seg_len + (long)b * LLICTI_NSEG;
    long src = 0;
    bool bad = false;
    for (int k = 0; k < 4 + sg; ++k) {          // see unpack_kernel: every earlier entry is validated too
        const int v = sl[k];
        if (v < 0 || v > in_stride) bad = true;
        src += v;
        if (src < 0 || src > in_stride) { bad = true; src = 0; }
    }
    int n = sl[4 + sg];
    if (G > 1) {
        // the segment's table: G stream lengths, each a whole stream, together exactly the segment
        const int seg_n = n;
        if (bad || seg_n < 4 * G || src + seg_n > in_stride) { bad = true; n = 0; }
        else {
            const uint8_t *tab = in + (long)b * in_stride + src;
            long off = 4L * G, tot = 4L * G;
            for (int k = 0; k < G; ++k) {
                const long len = (long)tab[4 * k] | ((long)tab[4 * k + 1] << 8) | ((long)tab[4 * k + 2] << 16) | ((long)tab[4 * k + 3] << 24);
                if (len < min_stream || len > seg_n) bad = true;
                if (k < m % G) off += len;
                if (k == m % G) n = (int)min(len, (long)seg_n);
                tot += len;
            }
            if (tot != seg_n) bad = true;
            src += off;
        }
    }
    uint8_t *o = slots + rslot_off[s0 + m] + stream_off;         // the bit region lands dword aligned at slot + 4: stream offset 2 behind the u16 of a 64- / 128-lane stream, 0 in an xwide v4 stream
    if (bad || n < min_stream || n + 4 + 64 > rslot_cap || src + n > in_stride) {
        if (threadIdx.x == 0) flag_image(status, b, LLICTI_EFORMAT);
        n = min_stream;                                            // a harmless stream: T = 0, no bits, states 2^31
        for (int t = threadIdx.x; t < n; t += blockDim.x) o[t] = 0;
    } else {
        const uint8_t *p = in + (long)b * in_stride + src;
        block_copy_bytes(o, p, n);
    }
    const int padded = min(rslot_cap - stream_off, n + 64);
    for (int t = n + threadIdx.x; t < padded; t += blockDim.x) o[t] = 0;
    if (threadIdx.x == 0) rpos[s0 + m] = (uint32_t)n;            // the length rans_init_kernel parses
}
