// host_types.hpp -- what the HOST side of the library and the kernels share: error reporting, geometry structs (one image, one level,
// one stage of a whole-batch call), stream descriptors and the constants of the container formats.  Plain C++17 with no HIP dependency:
// g++ compiles this file and host_plan.hpp under AddressSanitizer / UBSan (tests/sanitize_host.sh).  In the HIP build it is part of the
// single translation unit llicti_hip.hip (included through common.hpp).
#pragma once
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/llicti_hip.h"

#if defined(__HIPCC__)
#define LLICTI_HD __host__ __device__ __forceinline__
#else
#define LLICTI_HD inline
#endif

// ------------------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}



// ------------------------------------------------------------------------------------------------ geometry
struct Geom {
    int B, H, W, lvl;
    int Hl, Wl, h, w, padH, padW;
    long plane;   // H*W
    // Whole-batch calls keep one Geom PER IMAGE in a device table (the images of a call may differ in size: llicti_encode_images_v); these two
    // fields are only meaningful there.  The kernel-level entry points pass one Geom by value and derive both from the image index.
    long pix_off; // first element of the image's [3][H][W] block in planes / fplanes
    long par_off; // first float of the image's [64][h * w] block in the CNN-output buffer of this level
};
// One image of a whole-batch call (device table, llicti_hip.hip: Plan).
struct ImgGeo {
    int H, W, h4, w4, padint, hdr_bytes;      // hdr_bytes = 17 + 3 h4 w4 (LLICTI_nets.py:347-350)
    int M, sbase;                             // rANS containers: the image's stream count (its header says so: images of one call may differ) and its first stream
    int byte0;                                // header byte 0 of its container (AC: the number of scales; rANS: the tag)
    int Mlo;                                  // "auto" xwide encodes (LLICTI_MODE_RANS_X_AUTO): the count the image's size gives; M is then the LARGEST the encoder may
                                              // pick -- it picks per image, on the device, from what the image's last stage costs (rans_auto_hi / rans_auto_min,
                                              // choose_streams_kernel) -- and the table holds M streams for it.  0: the count is M, fixed by the caller
    long plane;                               // H * W
    long pix_off;                             // first element of the image's [3][H][W] block in planes / fplanes (workspace)
    long rgb_off;                             // first byte of its [3][H][W] block in the caller's RGB buffer
};
static Geom make_geom(int B, int H, int W, int lvl)
{
    Geom g;
    g.B = B; g.H = H; g.W = W; g.lvl = lvl;
    const int st = 1 << lvl;
    g.Hl = (H + st - 1) / st;
    g.Wl = (W + st - 1) / st;
    g.h = (g.Hl + 1) / 2;
    g.w = (g.Wl + 1) / 2;
    g.padH = g.Hl & 1;
    g.padW = g.Wl & 1;
    g.plane = (long)H * W;
    g.pix_off = 0; g.par_off = 0;
    return g;
}
static void coded_dims(const Geom &g, int band, int *hc, int *wc)
{
    *hc = (band == 0 || band == 2) ? g.h - g.padH : g.h;   // LLICTI_nets.py:396-397
    *wc = (band == 0 || band == 1) ? g.w - g.padW : g.w;
}
static int check_dims(int B, int H, int W)
{
    if (B < 1 || H < 32 || W < 32 || H > 8160 || W > 8160) return fail(LLICTI_EINVAL, "bad shape B=%d H=%d W=%d (need B>=1, 32<=H,W<=8160)", B, H, W);
    return 0;
}


// ------------------------------------------------------------------------------------------------ stages, streams, tiles
struct StageGeom {      // one (level, band): band grid, coded crop, full-res addressing
    int B, H, W, lvl, h, w, hc, wc, oi, oj;
    long plane;
    uint32_t wc_mul;    // n / wc for 0 <= n < 2^31 without a division: div_by_magic(n, wc_mul, wc_sh)
    int wc_sh;
    // Per-image placement (whole-batch calls keep one StageGeom per image in a device table -- the images of a call may differ in size; a
    // kernel-level entry point passes ONE by value and stage_at() derives these from the image index):
    long img_off;       // first element of the image's [3][H][W] block in planes / fplanes
    long par_off;       // first float of its [64][h * w] block in the CNN-output buffer of this (level, band)
    long pair_off;      // first pair of its Y stream in the (level, band)'s pairs [clr][image][n] ...
    long pair_cs;       // ... and the distance between two colour channels there (the images' coded positions together)
};
// exact unsigned division by an invariant divisor (Granlund / Montgomery, the 33-bit multiplier form), 1 <= d < 2^31,
// 0 <= n < 2^31: l = ceil(log2 d), mul = floor(2^32 (2^l - d) / d) + 1, t = mulhi(n, mul), q = (t + ((n - t) >> 1)) >> (l - 1).
// d == 1 has no 33-bit form with a non-negative shift: it is encoded as (mul, sh) = (0, -1) and div_by_magic() returns n.
static void div_magic(uint32_t d, uint32_t *mul, int *sh)
{
    int l = 0;
    while ((1ull << l) < d) ++l;                                   // l = ceil(log2 d)
    if (l == 0) { *mul = 0; *sh = -1; return; }                     // d == 1
    *mul = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    *sh = l - 1;
}
LLICTI_HD uint32_t div_by_magic(uint32_t n, uint32_t mul, int sh)
{
    if (sh < 0) return n;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t t = __umulhi(n, mul);
#else
    const uint32_t t = (uint32_t)(((uint64_t)n * mul) >> 32);
#endif
    return (t + ((n - t) >> 1)) >> sh;
}
LLICTI_HD int div_wc(const StageGeom &s, int n) { return (int)div_by_magic((uint32_t)n, s.wc_mul, s.wc_sh); }
// host self-test (llicti_selftest): every divisor of the format's range against '/', at the values where a magic division breaks first
static int selftest_div_magic()
{
    for (uint32_t d = 1; d <= 8192; ++d) {
        uint32_t mul; int sh;
        div_magic(d, &mul, &sh);
        const uint32_t probes[] = { 0u, 1u, d - 1, d, d + 1, 2 * d - 1, 2 * d, 4080u * 4080u - 1, 4080u * 4080u, (1u << 24) - 1, (1u << 24) + d,
                                    0x7FFFFFFFu / d * d - 1, 0x7FFFFFFFu / d * d, 0x7FFFFFFFu };
        for (uint32_t n : probes) if ((n >> 31) == 0 && div_by_magic(n, mul, sh) != n / d) return (int)d;
        for (uint32_t k = 0; k < 4096; ++k) { const uint32_t n = k * 524287u + d; if ((n >> 31) == 0 && div_by_magic(n, mul, sh) != n / d) return (int)d; }
    }
    return 0;
}
static StageGeom make_stage(const Geom &g, int band)
{
    static const int OI[4] = { 0, 1, 0, 1 }, OJ[4] = { 0, 1, 1, 0 };
    StageGeom s;
    s.B = g.B; s.H = g.H; s.W = g.W; s.lvl = g.lvl; s.h = g.h; s.w = g.w; s.plane = g.plane;
    coded_dims(g, band, &s.hc, &s.wc);
    s.oi = OI[band + 1]; s.oj = OJ[band + 1];
    div_magic((uint32_t)s.wc, &s.wc_mul, &s.wc_sh);
    s.img_off = g.pix_off; s.par_off = g.par_off; s.pair_off = 0; s.pair_cs = 0;
    return s;
}

struct StreamDesc {     // one arithmetic-coded stream of the whole-batch encoder
    long pair_off;      // first (c_low, c_high) pair, in uint32 units
    long out_off;       // slot offset in bytes
    int n;              // symbols
    int cap;            // slot capacity in bytes
};


struct TileRef { int img, yx; };        // one tile of a mixed-size band-CNN launch: image, tile row << 16 | tile column
struct StreamRef { int b, m, M, sbase; };   // one rANS stream of a call: image, stream of the image, the image's stream count, the image's first stream

// ------------------------------------------------------------------------------------------------ container format constants
constexpr int kLiftMaxParts = 65536;        // entries of the partials scratch: B * gridDim.x <= this

constexpr int kAnchorRow = 208;

constexpr int kRansStateBits = 31;
constexpr int kRansTailMax = 2047;          // tail symbols of a 64- / 128-lane stream (the 11-bit T field)
constexpr int kRansTailBlock = 32;          // xwide v4: a tail is a multiple of 32 symbols (or the stream's whole share of the last stage): its field is T / 32 rounded up, 8 bits
constexpr int kRansTailMaxX = 255 * kRansTailBlock;      // 8,160: ... of an xwide stream
constexpr int kRansSpillMax = 512;          // xwide v4: bits by which the tail coder's output may exceed the payload (< 32 symbols x 16 bits); they lie at the bottom of the main bit region
// A stream has 64 Q lanes: Q = 1, Q = 2 ("wide" streams: two 64-symbol sub-chunks per step, decoded two lanes per symbol by
// rans_decode_stage_pair_kernel, at the price of a tail twice as long) or Q = 4 ("xwide": 256 lanes, decoded ONE lane per symbol by
// rans_decode_stage_lane_kernel, four wavefronts per stream).  Symbol n of a stage sits in chunk n / 64Q.
template <int Q> struct RansGeo {
    static constexpr int kLanes = 64 * Q;
    static constexpr int kPayBits = kLanes * kRansStateBits;      // 1984 / 3968 / 7936: what the initial states carry (the tail stream)
    static constexpr int kPayBytes = kPayBits / 8;                // 248 / 496 / 992
    static constexpr int kPayDw = (kPayBits + 31) / 32;           // 62 / 124 / 248
    static constexpr int kMinStream = 2 + kPayBytes;              // T | pad (xwide v4: the header field under its end marker -- ten bits of the bit region), states
};
constexpr int kRansPayBytesMax = RansGeo<4>::kPayBytes;
constexpr int kPhiLutN = 2048;                   // the lane decoder's hint table (llicti_ctx::d_phi_lut)
constexpr double kPhiLutZ = 6.0;
// xwide streams (Q = 4) only, "v4" since round 6 -- the older stream kinds keep their v3 bytes (spec: oracle/llicti_oracle.h, "tail, xwide v4").
//   arena    the tail coder's output is not cut to the payload: tail symbols are taken (j = 0: the stream's last) until, at a multiple of 32, the
//            output has reached the payload's 7,936 bits; what exceeds them (the "spill", < 512 bits) lies at the bottom of the main bit region,
//            where the main decoder -- reading down -- leaves it: its final cursor IS the spill's length;
//   chains   one or two (the encoder's integer rule on the stream's last 64 symbols; bit 8 of the header field = one).  Two: seeded as in v3 --
//            A = number of symbol values of the image's Cg channel, n = rans_seed_count(A), chain A starts from 2^31 | sum sym(i) A^i (i < n),
//            chain B from that of sym(n + i), symbol j >= 2 n on A if j is even -- states in the arena's lowest and highest 32 bits, fields
//            towards each other.  One: the chain starts from x = sym(0), raw, and emits nothing while its state is below the interval; fields
//            up from bit 32, one end-marker bit above the last -- the arena's highest set bit;
//   header   on top of the bit region: 8 bits ceil(T / 32), 1 bit "one chain", 1 end-marker bit, zeros to the byte boundary.
template <int Q> constexpr bool kSeeded = (Q == 4);
constexpr int kSeedMax = 31;
LLICTI_HD int rans_seed_count(int A, uint32_t &pw)      // n and A^n
{
    int n = 0;
    uint64_t p = 1;
    while (n < kSeedMax && p * (uint64_t)A <= (1ull << 31)) { p *= (uint64_t)A; ++n; }
    pw = (uint32_t)p;
    return n;
}

// symbols of stream m in a stage of nc symbols (chunks of L symbols m, m + M, ...; only the stage's last chunk can be partial)
LLICTI_HD int rans_stream_count(int nc, int m, int M, int L)
{
    const int nchunks = (nc + L - 1) / L;
    if (nchunks <= m) return 0;
    const int K = (nchunks - m + M - 1) / M;
    const int last = m + (K - 1) * M;
    return L * K - ((last == nchunks - 1 && (nc % L)) ? L - (nc % L) : 0);
}


// M <= 32: stream m is segment 4 + m of the container.  M = 64 / 128 (latency modes for single / large images; the reference's list
// has 45 stream slots): G = M / 32 streams share segment 4 + m / G = G little-endian u32 stream lengths, then the G streams.
LLICTI_HD int rans_group(int M) { return M > 32 ? M / 32 : 1; }

// "auto" xwide encodes: the stream count of an image is chosen by the ENCODER, per image, from the image itself -- its size (Mlo, the caller's
// rule: llicti_amd.codec.image_streams, sized for natural-like content at ~4.7 bytes per stream) and what the symbols of its LAST stage cost,
// S = sum over its n symbols of (16 - floor(log2 freq)) (an integer; it overstates the ideal bits by about half a bit per symbol):
//   expensive symbols (S >= 11 n: uniform noise under the sigma-floor weights, ~12 bits each): an xwide v4 stream costs ~2.5 bytes there (two
//     seeded tail chains carry six raw symbols), so a third more streams fit the same byte budget: rans_auto_hi(Mlo);
//   cheap symbols (S < 4 n: the class of the reference's trained model on natural images, 1.7 bits each): a stream costs ~5.1 bytes and its
//     serial tail is ~5,000 symbols long: two thirds of the streams, rans_auto_cheap(Mlo);
//   and whatever the class, a count whose payloads of 7,936 bits the last stage cannot fill with a tenth to spare (2 S - n < 2 * 8,704 M)
//     falls to half the size rule's, rans_auto_min(Mlo) -- every unfilled payload bit is a wasted bit.
// A pure function of the image: the same image gets the same container whatever it is coded with, next to or after (oracle: orc_auto_streams).
LLICTI_HD int rans_auto_hi(int Mlo) { const int h = Mlo + (Mlo + 2) / 3; return h > 32 ? 32 : h; }
LLICTI_HD int rans_auto_cheap(int Mlo) { return (2 * Mlo + 2) / 3; }
LLICTI_HD int rans_auto_min(int Mlo) { return (Mlo + 1) / 2; }
LLICTI_HD int rans_auto_pick(int Mlo, long long S, long long n)
{
    if (n <= 0) return Mlo;
    int M = Mlo;
    if (S >= 11 * n) M = rans_auto_hi(Mlo);
    else if (S < 4 * n) M = rans_auto_cheap(Mlo);
    if (2 * S - n < 2LL * 8704 * M) M = (M > Mlo && 2 * S - n >= 2LL * 8704 * Mlo) ? Mlo : rans_auto_min(Mlo);
    return M;
}

