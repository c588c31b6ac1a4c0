// ac_coder.hpp -- torchac-algorithm range coder: encoders (one wave per stream) and the LDS-ring decoder (K10-K11).
// Part of the single translation unit llicti_hip.hip (included in order; not a stand-alone header).
#pragma once

// ------------------------------------------------------------------------------------------------ arithmetic coder
// torchac 0.9.3 algorithm: 32-bit low/high, 16-bit CDFs, pending-bit carry
// handling, MSB-first bits.  The bit-at-a-time renormalisation loop is evaluated in closed form:
//   n1 = clz(low ^ high)                      leading bits on which low and high agree  (E1/E2 steps)
//   n2 = min(clo(low' << 1), clz(high' << 1)) following "01.. / 10.." underflow steps   (E3 steps)
struct BitWriter {          // MSB-first bit stream, flushed 32 bits at a time into a 4-byte aligned slot
    uint32_t *out; int cap_words; int pos; uint64_t acc; int nb; int overflow;
    __device__ __forceinline__ void put(uint32_t bits, int k)      // k <= 32, nb < 32 on entry
    {
        acc = (acc << k) | bits; nb += k;
        if (nb >= 32) {
            const uint32_t w = (uint32_t)(acc >> (nb - 32));
            if (pos < cap_words) out[pos] = __builtin_bswap32(w); else overflow = 1;
            ++pos; nb -= 32;
        }
    }
    __device__ __forceinline__ void put_run(uint32_t bit, uint32_t count)
    {
        while (count > 0) {
            const int k = count > 24 ? 24 : (int)count;
            put(bit ? ((1u << k) - 1u) : 0u, k);
            count -= k;
        }
    }
    // pad with zero bits to a byte boundary; returns the stream length in bytes
    __device__ __forceinline__ int finish()
    {
        const int nbytes = (nb + 7) >> 3;
        if (nbytes > 0) {
            const uint32_t w = (uint32_t)(acc << (32 - nb));       // left-aligned remaining bits, zero padded
            if (pos < cap_words) out[pos] = __builtin_bswap32(w); else overflow = 1;
        }
        return 4 * pos + nbytes;
    }
};

// (span * c) >> 16 (mod 2^32) with span = r + 1 (r = high - low, possibly 0xFFFFFFFF) and c <= 0x10000,
// on full-rate 24-bit multiplies: r = rh * 2^16 + rl  =>  rh*c + ((rl*c + c) >> 16); no term overflows for c < 2^16
__device__ __forceinline__ uint32_t scale16(uint32_t r, uint32_t c)
{
    if (c == 0x10000u) return r + 1u;
    return __umul24(r >> 16, c) + ((__umul24(r & 0xFFFFu, c) + c) >> 16);
}

struct AcEnc {
    uint32_t low, high, pending;
    __device__ __forceinline__ void init() { low = 0; high = 0xFFFFFFFFu; pending = 0; }
    __device__ __forceinline__ void put(BitWriter &bw, uint32_t c_low, uint32_t c_high)
    {
        const uint32_t r = high - low;
        high = (low - 1) + scale16(r, c_high);
        low = low + scale16(r, c_low);
        int n1 = __clz((int)(low ^ high));
        if (n1 > 31) n1 = 31;
        if (n1 > 0) {
            const uint32_t b = low >> 31;
            bw.put(b, 1);
            bw.put_run(b ^ 1u, pending);
            pending = 0;
            if (n1 > 1) bw.put((low << 1) >> (33 - n1), n1 - 1);
            low <<= n1;
            high = (high << n1) | ((1u << n1) - 1u);
        }
        int n2 = min(__clz((int)~(low << 1)), __clz((int)(high << 1)));
        if (n2 > 31) n2 = 31;
        if (n2 > 0) {
            pending += n2;
            low = (low << n2) & 0x7FFFFFFFu;
            high = ((high << n2) | ((1u << n2) - 1u)) | 0x80000000u;
        }
    }
    __device__ __forceinline__ void finish(BitWriter &bw)
    {
        pending += 1;
        const uint32_t b = (low < 0x40000000u) ? 0u : 1u;
        bw.put(b, 1);
        bw.put_run(b ^ 1u, pending);
    }
};

__global__ __launch_bounds__(64) void ac_encode_pairs_kernel(const uint32_t *__restrict__ pairs, const StreamDesc *__restrict__ desc,
                                                             int n_streams, uint8_t *__restrict__ slots,
                                                             int32_t *__restrict__ slot_len, int32_t *status)
{
    // one wavefront per stream, one working lane: the coder is a serial chain, and a lone lane per wave runs
    // it without divergence on its own SIMD (64 streams sharing a wave executed both sides of every branch)
    const int s = blockIdx.x;
    if (s >= n_streams || threadIdx.x != 0) return;
    const StreamDesc d = desc[s];
    const uint32_t *p = pairs + d.pair_off;
    BitWriter bw = { reinterpret_cast<uint32_t *>(slots + d.out_off), d.cap / 4, 0, 0, 0, 0 };
    AcEnc e;
    e.init();
    int i = 0;
    for (; i + 8 <= d.n; i += 8) {              // 8 pairs in flight: the loads do not depend on the coder state
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[i + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t c_high = v[k] >> 16;
            if (c_high == 0) c_high = 0x10000u;
            e.put(bw, v[k] & 0xFFFFu, c_high);
        }
    }
    for (; i < d.n; ++i) {
        const uint32_t v = p[i];
        uint32_t c_high = v >> 16;
        if (c_high == 0) c_high = 0x10000u;
        e.put(bw, v & 0xFFFFu, c_high);
    }
    e.finish(bw);
    slot_len[s] = bw.finish();
    if (bw.overflow) atomicExch(&status[0], LLICTI_ENOSPACE);
}

// torchac seam: explicit tables + symbols, one lane per stream
__global__ __launch_bounds__(64) void ac_encode_tables_kernel(const uint16_t *__restrict__ cdf, int Lp, int row_stride,
                                                              const int16_t *__restrict__ sym, int n_streams, long N,
                                                              uint8_t *__restrict__ out, long out_stride,
                                                              int32_t *__restrict__ len, int32_t *status)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    const uint16_t *tab = cdf + (long)s * N * row_stride;
    const int16_t *sy = sym + (long)s * N;
    BitWriter bw = { reinterpret_cast<uint32_t *>(out + (long)s * out_stride), (int)(out_stride / 4), 0, 0, 0, 0 };
    AcEnc e;
    e.init();
    const int max_symbol = Lp - 2;
    for (long i = 0; i < N; ++i) {
        const int v = sy[i];
        const uint32_t c_low = tab[i * row_stride + v];
        const uint32_t c_high = (v == max_symbol) ? 0x10000u : (uint32_t)tab[i * row_stride + v + 1];
        e.put(bw, c_low, c_high);
    }
    e.finish(bw);
    len[s] = bw.finish();
    if (bw.overflow) atomicExch(&status[0], LLICTI_ENOSPACE);
}

// Decoder: one wavefront per stream; everything below is wave-uniform except the table row, of which each
// lane holds 8 entries.  torchac decodes  count = ((value-low+1)*65536 - 1) / span  and binary-searches the
// row for it; since  entry <= count  <=>  (span*entry >> 16) <= value-low  (integers), the 64-bit division is
// replaced by one multiply-compare per candidate: round 1 tests every lane's first entry (ballot -> the
// lane L holding the symbol), round 2 the 8 entries of lane L (readlane + ballot).  On a strictly
// increasing row this is the index torchac's search returns.  8 rows are kept in flight in registers.
struct DecOut {
    int16_t *sym;            // [n_streams][N] or nullptr
    const int32_t *len;      // [n_streams] stream lengths in bytes, or nullptr (then the slot is zero padded by the caller)
    int16_t *planes;         // [B][3][H][W] or nullptr
    float *fplanes;
    const int32_t *minmax;   // [B][4]
    StageGeom sg;
    int clr;
};

__device__ __forceinline__ uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }
__device__ __forceinline__ uint32_t pick16(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, int e)   // entry e of 8 packed in 4 words
{
    const uint32_t w = (e & 4) ? ((e & 2) ? w3 : w2) : ((e & 2) ? w1 : w0);
    return (e & 1) ? (w >> 16) : (w & 0xFFFFu);
}

constexpr int kDecRing = 8;            // table rows in flight per stream (LDS ring, 1 KB each)

#define VMCNT_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// LDS read the compiler cannot see: in front of a visible ds_read of memory an LDS-DMA may have written it
// inserts s_waitcnt vmcnt(0) (all transfers), which would defeat the ring; the explicit counts above order
// this read after the one transfer it needs.
__device__ __forceinline__ u32x4 lds_read_b128_hidden(const void *p)
{
    u32x4 v;
    const uint32_t a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}

// A launch decodes symbols [n0, n0 + cnt) of every stream (n0 a multiple of 64); a stream's coder state
// travels between the launches of a stage in `state` (8 words per stream), so that a stage can be cut into
// chunks and the three colour channels of a band decoded as a pipeline (Co's tables need Y's pixels, Cg's need
// Y's and Co's -- of the SAME positions only).  Table rows of the chunk: cdf[stream][cap_rows][row_stride].
struct AcChunk { int n0, cnt, n_total, cap_rows; uint32_t *state; };

__global__ __launch_bounds__(64) void ac_decode_kernel(const uint16_t *__restrict__ cdf, int Lp_fixed, int row_stride,
                                                       const uint8_t *__restrict__ in, long in_stride, AcChunk ck, DecOut o)
{
    // Table rows reach the wave through an LDS ring filled by LDS-DMA (global_load_lds_dwordx4: one
    // instruction moves a whole 1 KB row), waited for with explicit vmcnt counts: rows held in registers
    // made the compiler copy them around behind an s_waitcnt vmcnt(0), i.e. one full memory latency per symbol.
    __shared__ uint4 ring[kDecRing][64];
    const int s = blockIdx.x;
    const int lane = threadIdx.x;
    const int N = ck.cnt;                                // symbols of this launch (local index nl; absolute n0 + nl)
    const bool first = ck.n0 == 0, last = ck.n0 + ck.cnt >= ck.n_total;
    const uint32_t *words = reinterpret_cast<const uint32_t *>(in + (long)s * in_stride);
    // streams are zero padded: reads past the end return 0 bits like torchac's get()
    int Lp = Lp_fixed, shift = 0;
    if (o.planes) {
        int minv, maxv;
        clr_range(o.minmax + 4 * s, o.clr, minv, maxv, shift);
        Lp = maxv - minv + 2;
    }
    const uint32_t max_symbol = (uint32_t)(Lp - 2);
    const uint16_t *tab = cdf + (long)s * ck.cap_rows * row_stride;
    const int vec_per_row = row_stride >> 3;             // uint4 (8 entries) per row
    // every lane transfers (lanes past the row re-read its last vector; their entries fail idx <= max_symbol)
    const int lane_vec = min(lane, vec_per_row - 1);
    auto dma_row = [&](int n, int slot) {
        const uint4 *src = reinterpret_cast<const uint4 *>(tab + (long)min(n, N - 1) * row_stride) + lane_vec;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)&ring[slot][0], 16, 0, 0);
    };

    // Bitstream window: lane l holds word (wbase + l) of the stream; the coder pulls its next 32 bits with one
    // readlane.  Every 64 words (~150 symbols) the window is reloaded synchronously: one memory latency per
    // 150 symbols, and no register with a load in flight across loop iterations (those make the compiler
    // emit s_waitcnt vmcnt(0) at every merge point, which would drain the row ring as well).
    // Words the stream really has: with a length (torchac seam) everything past it reads as zero bits, whatever the
    // buffer holds there -- the last word is masked to its valid bytes; without one the slot is zero padded (unpack_kernel).
    const int len_b = o.len ? max(0, min(o.len[s], (int)in_stride)) : (int)in_stride;
    const int in_words = max(1, (len_b + 3) >> 2);
    const uint32_t tail_mask = (len_b & 3) ? ~(0xFFFFFFFFu >> (8 * (len_b & 3))) : 0xFFFFFFFFu;
    auto load_win = [&](int w0) -> uint32_t {           // big-endian words, swapped once per reload
        const int wi = min(w0 + lane, in_words - 1);
        const uint32_t w = bswap32(words[wi]);
        if (w0 + lane >= in_words) return 0u;           // past the stream: zero bits (also for the first three words of a stream of <= 8 bytes)
        return (wi == in_words - 1) ? (len_b > 0 ? (w & tail_mask) : 0u) : w;
    };
    uint32_t *st = ck.state ? ck.state + 8 * s : nullptr;
    int wpos = first ? 3 : (int)st[6];                  // next word to pull (wave-uniform)
    uint32_t win_cur = load_win(first ? 0 : (wpos & ~63));
    asm volatile("" : "+v"(win_cur));
    auto next_word = [&]() -> uint32_t {
        uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)win_cur, wpos & 63);
        if (wpos >= in_words) w = 0;                    // reads past the slot return 0 bits
        ++wpos;
        if ((wpos & 63) == 0) { win_cur = load_win(wpos); asm volatile("" : "+v"(win_cur)); }   // wait for it here, not at every later pull
        return w;
    };
#pragma unroll
    for (int k = 0; k < kDecRing; ++k) dma_row(k, k);
    uint32_t value = (uint32_t)__builtin_amdgcn_readlane((int)win_cur, 0);
    const uint32_t w1_ = (uint32_t)__builtin_amdgcn_readlane((int)win_cur, 1), w2_ = (uint32_t)__builtin_amdgcn_readlane((int)win_cur, 2);
    uint64_t buf = ((uint64_t)w1_ << 32) | w2_;        // next 64 bits, MSB first (readlane returns a signed int: no sign extension here)
    int have = 64;
    uint32_t low = 0, high = 0xFFFFFFFFu;
    if (!first) {                                       // resume where the previous chunk of this stage stopped
        low = st[0]; high = st[1]; value = st[2]; buf = ((uint64_t)st[3] << 32) | st[4]; have = (int)st[5];
    }
    const bool lane_ok = 8u * (uint32_t)lane <= max_symbol;      // this lane's first entry is a real table entry
    const int e = lane & 7;

    // decoded symbols are parked one per lane and written out every 64 symbols (one store wave instead of 64)
    int mysym = 0;
    auto flush = [&](int n_first, int count) {
        if (lane < count) {
            const int n = n_first + lane;
            if (o.sym) o.sym[(long)s * ck.n_total + n] = (int16_t)mysym;
            if (o.planes) {
                const int i = div_wc(o.sg, n), j = n - i * o.sg.wc;
                const long off = (long)s * 3 * o.sg.plane + (long)o.clr * o.sg.plane +
                                 ((long)(2 * i + o.sg.oi) << o.sg.lvl) * o.sg.W + ((long)(2 * j + o.sg.oj) << o.sg.lvl);
                const int v = mysym - shift;                       // _convert_int16cpu_to_float32gpu, LLICTI_nets.py:559-568
                o.planes[off] = (int16_t)v;
                o.fplanes[off] = (float)v / 255.0f;
            }
        }
    };

    VMCNT_WAIT(7);                                      // row 0 has landed (kDecRing - 1 younger transfers)
    u32x4 cur = lds_read_b128_hidden(&ring[0][lane]);
    for (int n = 0; n < N; ++n) {
        const int slot = n & (kDecRing - 1);
        VMCNT_WAIT(6);                                  // row n + 1 has landed (kDecRing - 2 younger transfers, or more waited for)
        const u32x4 nxt = lds_read_b128_hidden(&ring[(n + 1) & (kDecRing - 1)][lane]);
        const uint32_t r = high - low, T = value - low;
        const uint32_t rh = r >> 16, rl = r & 0xFFFFu;
        // (span * c) >> 16 for a table entry c < 2^16 (see scale16); the scaled values double as the interval
        // update below: low += scaled(c_low), high = low - 1 + scaled(c_high)
        // round 1: first entry of every lane (entry 0 always qualifies: torchac's search starts at left = 0)
        const uint32_t c1 = cur.x & 0xFFFFu;
        const uint32_t sc1 = __umul24(rh, c1) + ((__umul24(rl, c1) + c1) >> 16);
        const bool p1 = (lane == 0) || (lane_ok && sc1 <= T);
        const int L = __builtin_popcountll(ballot64(p1)) - 1;
        const uint32_t w0 = __builtin_amdgcn_readlane(cur.x, L), w1 = __builtin_amdgcn_readlane(cur.y, L);
        const uint32_t w2 = __builtin_amdgcn_readlane(cur.z, L), w3 = __builtin_amdgcn_readlane(cur.w, L);
        // round 2: the 8 entries of lane L, one per lane e = lane & 7
        const uint32_t c2 = pick16(w0, w1, w2, w3, e);
        const uint32_t sc2 = __umul24(rh, c2) + ((__umul24(rl, c2) + c2) >> 16);
        const uint32_t idx = 8u * (uint32_t)L + (uint32_t)e;
        const bool p2 = (e == 0) || (idx <= max_symbol && sc2 <= T);
        const int es = __builtin_popcount((uint32_t)ballot64(p2) & 0xFFu) - 1;
        const uint32_t sidx = 8u * (uint32_t)L + (uint32_t)es;
        const uint32_t low_add = __builtin_amdgcn_readlane(sc2, es);
        const uint32_t hi_in = __builtin_amdgcn_readlane(sc2, (es + 1) & 7);      // entry sidx + 1 when es < 7
        const uint32_t hi_nx = __builtin_amdgcn_readlane(sc1, (L + 1) & 63);     // ... when it is the next lane's first entry
        const uint32_t high_add = (sidx == max_symbol) ? r + 1u : (es == 7 ? hi_nx : hi_in);   // top symbol: c_high = 0x10000
        if (lane == (n & 63)) mysym = (int)sidx;
        if ((n & 63) == 63) flush(ck.n0 + n - 63, 64);
        // slot's row sits in `cur` (read one iteration ago): refill it with row n + kDecRing
        dma_row(n + kDecRing, slot);
        cur = nxt;
        if (n == N - 1 && last) break;                  // torchac does not update after the stream's last symbol
        high = (low - 1) + high_add;
        low = low + low_add;
        // renormalisation in closed form: n1 leading bits on which low and high agree (E1 / E2 steps), then n2
        // underflow steps (E3: low = 01.., high = 10..); both shifts of `value` happen at once, with one refill
        int n1 = __clz((int)(low ^ high));
        if (n1 > 31) n1 = 31;
        const uint32_t low1 = low << n1, high1 = (high << n1) | ((1u << n1) - 1u);
        int n2 = min(__clz((int)~(low1 << 1)), __clz((int)(high1 << 1)));
        if (n2 > 31) n2 = 31;
        const int nsh = n1 + n2;
        if (nsh > 0) {
            if (nsh < 32) {
                const uint32_t e3 = n2 > 0 ? 0x80000000u : 0u;
                low = (low1 << n2) & ~e3;
                high = ((high1 << n2) | ((1u << n2) - 1u)) | e3;
                value = ((value << nsh) | (uint32_t)(buf >> (64 - nsh))) ^ e3;
                buf <<= nsh; have -= nsh;
                if (have <= 32) {
                    buf |= (uint64_t)next_word() << (32 - have);
                    have += 32;
                }
            } else {                                    // >= 32 bits consumed by one symbol: never with 16-bit tables, kept for completeness
                low = low1; high = high1;
                if (n1 > 0) {
                    value = (value << n1) | (uint32_t)(buf >> (64 - n1));
                    buf <<= n1; have -= n1;
                    if (have <= 32) { buf |= (uint64_t)next_word() << (32 - have); have += 32; }
                }
                low = (low << n2) & 0x7FFFFFFFu;
                high = ((high << n2) | ((1u << n2) - 1u)) | 0x80000000u;
                value = ((value << n2) ^ 0x80000000u) | (uint32_t)(buf >> (64 - n2));
                buf <<= n2; have -= n2;
                if (have <= 32) { buf |= (uint64_t)next_word() << (32 - have); have += 32; }
            }
        }
    }
    if (N & 63) flush(ck.n0 + (N & ~63), N & 63);
    if (!last && lane == 0) {
        st[0] = low; st[1] = high; st[2] = value; st[3] = (uint32_t)(buf >> 32); st[4] = (uint32_t)buf; st[5] = (uint32_t)have; st[6] = (uint32_t)wpos;
    }
    VMCNT_WAIT(0);                                      // no transfer may still target this workgroup's LDS at exit
}

// ------------------------------------------------------------------------------------------------ anchor decoder
// Whole-batch AC decode without full tables (cdf.hpp: cdf_anchor_kernel).  Same coder, same state carry between chunk
// launches, same LDS ring / explicit vmcnt discipline as ac_decode_kernel; what changes is the search:
//   round 1: lane l holds anchor[l] = entry[8 l]; ballot of (span * anchor >> 16) <= value - low -> bucket L
//   round 2: the 8 entries 8L .. 8L+7 are EVALUATED here, exactly (numerics spec), one (entry, mixture component) pair
//            per lane: lane 8e + m computes term m of entry 8L + e, the five terms are summed in the spec's order over
//            DPP row shifts into lane 8e, integerised, scaled and voted on.  Entry 8L + 8 (c_high when the symbol is the
//            bucket's last) is the next anchor.
// Bit-identical to a search over the full row: the entries are the same bits wherever they are computed.
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_read_anchor_hidden(uint32_t a_addr, uint32_t c_addr, uint32_t &anc, u32x4 &cmp)
{
    asm volatile("ds_read_u16 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(anc), "=&v"(cmp) : "v"(a_addr), "v"(c_addr) : "memory");
}

__global__ __launch_bounds__(64) void ac_decode_anchor_kernel(const uint8_t *__restrict__ rows, const uint8_t *__restrict__ in,
                                                              long in_stride, AcChunk ck, DecOut o)
{
    __shared__ __attribute__((aligned(16))) uint8_t ring[kDecRing][256];      // kAnchorRow = 208 bytes used per slot
    const int s = blockIdx.x;
    const int lane = threadIdx.x;
    const int N = ck.cnt;
    const bool first = ck.n0 == 0, last = ck.n0 + ck.cnt >= ck.n_total;
    const uint32_t *words = reinterpret_cast<const uint32_t *>(in + (long)s * in_stride);
    int minv, maxv, shift;
    clr_range(o.minmax + 4 * s, o.clr, minv, maxv, shift);
    const Grid gr = make_grid(minv, maxv);
    const uint32_t max_symbol = (uint32_t)(gr.Lp - 2);
    const float fbase = (float)minv - 0.5f;
    const uint8_t *tab = rows + (long)s * ck.cap_rows * kAnchorRow;
    auto dma_row = [&](int n, int slot) {                     // 13 lanes x 16 bytes = one 208-byte row
        if (lane < kAnchorRow / 16) {
            const uint8_t *src = tab + (long)min(n, N - 1) * kAnchorRow + 16 * lane;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)&ring[slot][0], 16, 0, 0);
        }
    };
    // words the stream really has (o.len: the lengths unpack_kernel validated): everything past them reads as zero bits, whatever
    // the slot holds there -- a desynchronised decode of a malformed container must not depend on stale workspace bytes
    const int len_b = o.len ? max(0, min(o.len[s], (int)in_stride)) : (int)in_stride;
    const int in_words = max(1, (len_b + 3) >> 2);
    const uint32_t tail_mask = (len_b & 3) ? ~(0xFFFFFFFFu >> (8 * (len_b & 3))) : 0xFFFFFFFFu;
    auto load_win = [&](int w0) -> uint32_t {
        const int wi = min(w0 + lane, in_words - 1);
        const uint32_t w = bswap32(words[wi]);
        if (w0 + lane >= in_words) return 0u;
        return (wi == in_words - 1) ? (len_b > 0 ? (w & tail_mask) : 0u) : w;
    };
    uint32_t *st = ck.state + 8 * s;
    int wpos = first ? 3 : (int)st[6];
    uint32_t win_cur = load_win(first ? 0 : (wpos & ~63));
    asm volatile("" : "+v"(win_cur));
    auto next_word = [&]() -> uint32_t {
        uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)win_cur, wpos & 63);
        if (wpos >= in_words) w = 0;
        ++wpos;
        if ((wpos & 63) == 0) { win_cur = load_win(wpos); asm volatile("" : "+v"(win_cur)); }
        return w;
    };
#pragma unroll
    for (int k = 0; k < kDecRing; ++k) dma_row(k, k);
    uint32_t value = (uint32_t)__builtin_amdgcn_readlane((int)win_cur, 0);
    const uint32_t w1_ = (uint32_t)__builtin_amdgcn_readlane((int)win_cur, 1), w2_ = (uint32_t)__builtin_amdgcn_readlane((int)win_cur, 2);
    uint64_t buf = ((uint64_t)w1_ << 32) | w2_;
    int have = 64;
    uint32_t low = 0, high = 0xFFFFFFFFu;
    if (!first) { low = st[0]; high = st[1]; value = st[2]; buf = ((uint64_t)st[3] << 32) | st[4]; have = (int)st[5]; }
    const bool lane_ok = 8u * (uint32_t)lane <= max_symbol;
    const int e = lane >> 3, m = lane & 7;
    const bool head = (m == 0);
    const uint32_t ring0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)&ring[0][0];
    const uint32_t a_off = 2u * (uint32_t)lane, c_off = 128u + 16u * (uint32_t)min(m, 4);

    int mysym = 0;
    auto flush = [&](int n_first, int count) {
        if (lane < count) {
            const int n = n_first + lane;
            const int i = div_wc(o.sg, n), j = n - i * o.sg.wc;
            const long off = (long)s * 3 * o.sg.plane + (long)o.clr * o.sg.plane +
                             ((long)(2 * i + o.sg.oi) << o.sg.lvl) * o.sg.W + ((long)(2 * j + o.sg.oj) << o.sg.lvl);
            const int v = mysym - shift;
            o.planes[off] = (int16_t)v;
            o.fplanes[off] = (float)v / 255.0f;
        }
    };

    VMCNT_WAIT(7);
    uint32_t cur_a;
    u32x4 cur_c;
    lds_read_anchor_hidden(ring0 + a_off, ring0 + c_off, cur_a, cur_c);
    for (int n = 0; n < N; ++n) {
        const int slot = n & (kDecRing - 1);
        VMCNT_WAIT(6);
        uint32_t nxt_a;
        u32x4 nxt_c;
        const uint32_t nb = ring0 + 256u * (uint32_t)((n + 1) & (kDecRing - 1));
        lds_read_anchor_hidden(nb + a_off, nb + c_off, nxt_a, nxt_c);
        const uint32_t r = high - low, T = value - low;
        const uint32_t rh = r >> 16, rl = r & 0xFFFFu;
        // round 1: anchors
        const uint32_t sc1 = __umul24(rh, cur_a) + ((__umul24(rl, cur_a) + cur_a) >> 16);
        const bool p1 = (lane == 0) || (lane_ok && sc1 <= T);
        const int L = __builtin_popcountll(ballot64(p1)) - 1;
        // round 2: entries 8L + e, term m in lane 8e + m (cdf_entry()'s operations, in its order)
        const int i = 8 * L + e;
        const float pt = (i == 0) ? gr.p_first : div255_exact(fbase + (float)i);
        const float c_mu = __uint_as_float(cur_c.x), c_rs = __uint_as_float(cur_c.y), c_wn = __uint_as_float(cur_c.z);
        const float t = c_wn * (0.5f * erfc_spec_nobranch(kNegRsqrt2 * ((pt - c_mu) * c_rs)));
        float acc = t + dpp_row_shl(t, 1);
        acc = acc + dpp_row_shl(t, 2);
        acc = acc + dpp_row_shl(t, 3);
        acc = acc + dpp_row_shl(t, 4);
        const uint32_t c2 = (uint32_t)((int)__builtin_rintf(acc * gr.scale) + i) & 0xFFFFu;
        const uint32_t sc2 = __umul24(rh, c2) + ((__umul24(rl, c2) + c2) >> 16);
        const bool p2 = head && ((e == 0) || ((uint32_t)i <= max_symbol && sc2 <= T));
        const int es = __builtin_popcountll(ballot64(p2)) - 1;
        const uint32_t sidx = 8u * (uint32_t)L + (uint32_t)es;
        const uint32_t low_add = __builtin_amdgcn_readlane(sc2, 8 * es);
        const uint32_t hi_in = __builtin_amdgcn_readlane(sc2, (8 * (es + 1)) & 63);
        const uint32_t hi_nx = __builtin_amdgcn_readlane(sc1, (L + 1) & 63);
        const uint32_t high_add = (sidx == max_symbol) ? r + 1u : (es == 7 ? hi_nx : hi_in);
        if (lane == (n & 63)) mysym = (int)sidx;
        if ((n & 63) == 63) flush(ck.n0 + n - 63, 64);
        dma_row(n + kDecRing, slot);
        cur_a = nxt_a;
        cur_c = nxt_c;
        if (n == N - 1 && last) break;
        high = (low - 1) + high_add;
        low = low + low_add;
        int n1 = __clz((int)(low ^ high));
        if (n1 > 31) n1 = 31;
        const uint32_t low1 = low << n1, high1 = (high << n1) | ((1u << n1) - 1u);
        int n2 = min(__clz((int)~(low1 << 1)), __clz((int)(high1 << 1)));
        if (n2 > 31) n2 = 31;
        const int nsh = n1 + n2;
        if (nsh > 0) {
            if (nsh < 32) {
                const uint32_t e3 = n2 > 0 ? 0x80000000u : 0u;
                low = (low1 << n2) & ~e3;
                high = ((high1 << n2) | ((1u << n2) - 1u)) | e3;
                value = ((value << nsh) | (uint32_t)(buf >> (64 - nsh))) ^ e3;
                buf <<= nsh; have -= nsh;
                if (have <= 32) { buf |= (uint64_t)next_word() << (32 - have); have += 32; }
            } else {
                low = low1; high = high1;
                if (n1 > 0) {
                    value = (value << n1) | (uint32_t)(buf >> (64 - n1));
                    buf <<= n1; have -= n1;
                    if (have <= 32) { buf |= (uint64_t)next_word() << (32 - have); have += 32; }
                }
                low = (low << n2) & 0x7FFFFFFFu;
                high = ((high << n2) | ((1u << n2) - 1u)) | 0x80000000u;
                value = ((value << n2) ^ 0x80000000u) | (uint32_t)(buf >> (64 - n2));
                buf <<= n2; have -= n2;
                if (have <= 32) { buf |= (uint64_t)next_word() << (32 - have); have += 32; }
            }
        }
    }
    if (N & 63) flush(ck.n0 + (N & ~63), N & 63);
    if (!last && lane == 0) {
        st[0] = low; st[1] = high; st[2] = value; st[3] = (uint32_t)(buf >> 32); st[4] = (uint32_t)buf; st[5] = (uint32_t)have; st[6] = (uint32_t)wpos;
    }
    VMCNT_WAIT(0);
}
