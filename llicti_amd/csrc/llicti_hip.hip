// llicti_hip.hip -- gfx950 (MI355X / CDNA4) kernels and the C-ABI of include/llicti_hip.h.
//
// Kernels (reference call sites in include/llicti_hip.h):
//   lift_kernel / unlift_kernel       integer YCoCg-R lift, min/max, float planes          (HBM bound)
//   band_params_kernel<BAND>          interpolator CNN: 3 chained fp32-MFMA GEMMs per pixel tile,
//                                     weights of one 88-channel head resident in LDS        (MFMA bound)
//   cdf_pairs_kernel                  encoder: the two table entries per symbol (10 erfc)   (VALU)
//   cdf_table_kernel                  decoder: full Lp-entry uint16 rows                    (HBM / VALU bound)
//   ac_encode_*_kernel                torchac-algorithm range encoder, one lane per stream  (latency bound)
//   ac_decode_kernel                  matching decoder, one wavefront per stream            (latency bound)
//   header / pack / unpack kernels    container assembly in HBM
//
// No CPU path exists in this library: every entry point needs a gfx950 device.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/llicti_hip.h"
#include "numerics.hpp"

using namespace llicti;

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
#define HIPCHK(x)                                                                                    \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) return fail(LLICTI_EHIP, "%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern "C" const char *llicti_last_error(void) { return g_err.c_str(); }
extern "C" const char *llicti_version(void) { return "llicti_hip 0.1 (gfx950, numerics spec v1)"; }

// ------------------------------------------------------------------------------------------------ geometry
struct Geom {
    int B, H, W, lvl;
    int Hl, Wl, h, w, padH, padW;
    long plane;   // H*W
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
    return g;
}
static void coded_dims(const Geom &g, int band, int *hc, int *wc)
{
    *hc = (band == 0 || band == 2) ? g.h - g.padH : g.h;   // LLICTI_nets.py:396-397
    *wc = (band == 0 || band == 1) ? g.w - g.padW : g.w;
}
extern "C" int llicti_level_geom(int H, int W, int lvl, int band, int *Hl, int *Wl, int *h, int *w,
                                 int *padH, int *padW, int *hc, int *wc)
{
    if (H < 1 || W < 1 || lvl < 0 || lvl >= LLICTI_NLEVELS || band < 0 || band > 2) return fail(LLICTI_EINVAL, "level_geom: bad argument");
    Geom g = make_geom(1, H, W, lvl);
    if (Hl) *Hl = g.Hl;
    if (Wl) *Wl = g.Wl;
    if (h) *h = g.h;
    if (w) *w = g.w;
    if (padH) *padH = g.padH;
    if (padW) *padW = g.padW;
    int a, b;
    coded_dims(g, band, &a, &b);
    if (hc) *hc = a;
    if (wc) *wc = b;
    return LLICTI_OK;
}
static int check_dims(int B, int H, int W)
{
    if (B < 1 || H < 32 || W < 32 || H > 8160 || W > 8160) return fail(LLICTI_EINVAL, "bad shape B=%d H=%d W=%d (need B>=1, 32<=H,W<=8160)", B, H, W);
    return 0;
}

// source sub-bands in lazyDWT cat order x00, x11, x01, x10 (LLICTI_nets.py:241); band b predicts source b+1
// (row, column) phase of source s: (0,0), (1,1), (0,1), (1,0) -- computed, not looked up: a table load inside
// the CNN's staging loop would put an s_waitcnt vmcnt(0) between consecutive LDS-DMA pieces
__device__ __forceinline__ int src_oi(int s) { return s & 1; }
__device__ __forceinline__ int src_oj(int s) { return ((s + 1) >> 1) & 1; }

// ------------------------------------------------------------------------------------------------ lift
__global__ void minmax_init_kernel(int32_t *mm, int B)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) { mm[4 * i + 0] = 32767; mm[4 * i + 1] = 32767; mm[4 * i + 2] = -32768; mm[4 * i + 3] = -32768; }
}

// 4 pixels per thread when the plane size allows 4-byte aligned uchar4 / short4 / float4 accesses
template <int VEC>
__global__ __launch_bounds__(256) void lift_kernel(const uint8_t *__restrict__ rgb, long plane, int16_t *__restrict__ planes,
                                                   float *__restrict__ fplanes, int32_t *__restrict__ mm)
{
    const int b = blockIdx.y;
    const uint8_t *src = rgb + (long)b * 3 * plane;
    int16_t *dst = planes + (long)b * 3 * plane;
    float *fdst = fplanes + (long)b * 3 * plane;
    int mnCo = 32767, mnCg = 32767, mxCo = -32768, mxCg = -32768;
    for (long p = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VEC; p < plane; p += (long)gridDim.x * blockDim.x * VEC) {
        uint8_t r[VEC], gch[VEC], bl[VEC];
        if constexpr (VEC == 4) {
            const uchar4 a = *reinterpret_cast<const uchar4 *>(src + p);
            const uchar4 c = *reinterpret_cast<const uchar4 *>(src + plane + p);
            const uchar4 d = *reinterpret_cast<const uchar4 *>(src + 2 * plane + p);
            r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w;
            gch[0] = c.x; gch[1] = c.y; gch[2] = c.z; gch[3] = c.w;
            bl[0] = d.x; bl[1] = d.y; bl[2] = d.z; bl[3] = d.w;
        } else {
            r[0] = src[p]; gch[0] = src[plane + p]; bl[0] = src[2 * plane + p];
        }
        short y[VEC], co[VEC], cg[VEC];
        float fy[VEC], fco[VEC], fcg[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const int R = r[k], G = gch[k], Bl = bl[k];
            const int Co = R - Bl;
            const int t = Bl + (Co >> 1);        // floor division (torch >= 1.13 '//'; JVT YCoCg-R '>> 1')
            const int Cg = G - t;
            const int Y = t + (Cg >> 1) - 127;
            y[k] = (short)Y; co[k] = (short)Co; cg[k] = (short)Cg;
            fy[k] = (float)Y / 255.0f; fco[k] = (float)Co / 255.0f; fcg[k] = (float)Cg / 255.0f;
            mnCo = min(mnCo, Co); mxCo = max(mxCo, Co); mnCg = min(mnCg, Cg); mxCg = max(mxCg, Cg);
        }
        if constexpr (VEC == 4) {
            *reinterpret_cast<short4 *>(dst + p) = make_short4(y[0], y[1], y[2], y[3]);
            *reinterpret_cast<short4 *>(dst + plane + p) = make_short4(co[0], co[1], co[2], co[3]);
            *reinterpret_cast<short4 *>(dst + 2 * plane + p) = make_short4(cg[0], cg[1], cg[2], cg[3]);
            *reinterpret_cast<float4 *>(fdst + p) = make_float4(fy[0], fy[1], fy[2], fy[3]);
            *reinterpret_cast<float4 *>(fdst + plane + p) = make_float4(fco[0], fco[1], fco[2], fco[3]);
            *reinterpret_cast<float4 *>(fdst + 2 * plane + p) = make_float4(fcg[0], fcg[1], fcg[2], fcg[3]);
        } else {
            dst[p] = y[0]; dst[plane + p] = co[0]; dst[2 * plane + p] = cg[0];
            fdst[p] = fy[0]; fdst[plane + p] = fco[0]; fdst[2 * plane + p] = fcg[0];
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        mnCo = min(mnCo, __shfl_xor(mnCo, o)); mxCo = max(mxCo, __shfl_xor(mxCo, o));
        mnCg = min(mnCg, __shfl_xor(mnCg, o)); mxCg = max(mxCg, __shfl_xor(mxCg, o));
    }
    // one set of atomics per workgroup, and only where it would change the running value (a stale read can
    // only be larger than the true minimum / smaller than the true maximum, i.e. conservative): thousands
    // of waves hitting the same 16 bytes otherwise serialise at the memory side
    __shared__ int red[4][4];
    if ((threadIdx.x & 63) == 0) {
        const int wv = threadIdx.x >> 6;
        red[wv][0] = mnCo; red[wv][1] = mnCg; red[wv][2] = mxCo; red[wv][3] = mxCg;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        int v = red[0][k];
        for (int wv = 1; wv < 4; ++wv) v = (k < 2) ? min(v, red[wv][k]) : max(v, red[wv][k]);
        const int cur = __hip_atomic_load(&mm[4 * b + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k < 2) { if (v < cur) atomicMin(&mm[4 * b + k], v); }
        else { if (v > cur) atomicMax(&mm[4 * b + k], v); }
    }
}

__global__ __launch_bounds__(256) void unlift_kernel(const int16_t *__restrict__ planes, long plane, uint8_t *__restrict__ rgb)
{
    const int b = blockIdx.y;
    const int16_t *src = planes + (long)b * 3 * plane;
    uint8_t *dst = rgb + (long)b * 3 * plane;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < plane; p += (long)gridDim.x * blockDim.x) {
        const int Y = src[p] + 127, Co = src[plane + p], Cg = src[2 * plane + p];
        const int t = Y - (Cg >> 1);
        const int G = Cg + t;
        const int Bl = t - (Co >> 1);
        const int R = Bl + Co;
        dst[p] = (uint8_t)R; dst[plane + p] = (uint8_t)G; dst[2 * plane + p] = (uint8_t)Bl;
    }
}

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
    const float *par = params + ((long)b * s.h * s.w + n) * LLICTI_PARAM_STRIDE;
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

// ------------------------------------------------------------------------------------------------ band CNN
// Layer-0 convolutions of band b (LLICTI_nets.py:651-675): source sub-band, kernel size, top / left pad.
struct ConvDef { int src, kh, kw, pt, pl; };
constexpr ConvDef kConvs[3][3] = {
    { { 0, 4, 4, 1, 1 }, { -1, 0, 0, 0, 0 }, { -1, 0, 0, 0, 0 } },
    { { 0, 3, 4, 1, 1 }, { 1, 4, 3, 2, 1 }, { -1, 0, 0, 0, 0 } },
    { { 0, 4, 3, 1, 1 }, { 1, 3, 4, 1, 2 }, { 2, 4, 4, 1, 2 } },
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kHead = 88;          // channels per head (configs/llicti_A.json chs[0])
constexpr int kMT = 6;             // 16-row MFMA tiles per head (88 -> 96, rows >= 88 are zero)
constexpr int kKS1 = 22;           // k-steps of the 88-deep layers (88 / 4)
#ifndef CNN_NT
#define CNN_NT 2
#endif
constexpr int kNT = CNN_NT;                    // pixel tiles (16 positions each) per wavefront
constexpr int kCnnThreads = 64 * (32 / CNN_NT);    // NT=4: 8 wavefronts (2 per SIMD); NT=2: 16 wavefronts (4 per SIMD)
constexpr int kTileH = 16;         // band-grid positions per workgroup tile: 16 rows x 32 columns,
constexpr int kTileW = 32;         //   wave w owns rows 2w, 2w+1 (two 16-column pixel tiles each)
constexpr int kInRows = kTileH + 4;    // taps reach rows i-2 .. i+2 and columns j-2 .. j+2
constexpr int kInCols = kTileW + 4;
constexpr int kInPitch = 48;       // = 16 (mod 32): B-fragment reads that stride by one row stay bank-conflict free
constexpr int kInPlane = kInRows * kInPitch;
constexpr int kParamStride = LLICTI_PARAM_STRIDE;
#ifndef CNN_PREFETCH_L0
#define CNN_PREFETCH_L0 1      // software-pipeline the layer-0 fragments one k-step ahead
#endif
#ifndef CNN_FENCE_L1
#define CNN_FENCE_L1 4         // scheduler fence every N k-steps of layer 1 (0 = none)
#endif
#ifndef CNN_STAGE_SITES
#define CNN_STAGE_SITES 4     // points of the tile at which the wave groups request the next tile's DMA (1, 2 or 4)
#endif
#ifndef CNN_STAGGER
#define CNN_STAGGER 0          // delay waves 4-7 before the first tile (decorrelates the two waves of a SIMD)
#endif

// One MFMA k-step consumes 4 consecutive k of the canonical K order (llicti_amd/weights.py): the kernel's
// length-4 axis.  Lane (q = lane>>4, px = lane&15) therefore reads the staged input tile at
// U + q*S + pixel offset with U, S compile-time constants of the k-step.
struct KStep { int U, S; };
struct KTab { KStep s[30]; int n; };
constexpr KTab make_ktab(int band)
{
    KTab t{};
    int k = 0;
    for (int c = 0; c < 3; ++c) {
        const ConvDef cv = kConvs[band][c];
        if (cv.src < 0) break;
        for (int ci = 0; ci < 3; ++ci) {
            const int plane = (cv.src * 3 + ci) * kInPlane;
            if (cv.kw == 4) {
                for (int ky = 0; ky < cv.kh; ++ky) { t.s[k].U = plane + (ky - cv.pt + 2) * kInPitch + (2 - cv.pl); t.s[k].S = 1; ++k; }
            } else {
                for (int kx = 0; kx < cv.kw; ++kx) { t.s[k].U = plane + (2 - cv.pt) * kInPitch + (kx - cv.pl + 2); t.s[k].S = kInPitch; ++k; }
            }
        }
    }
    t.n = k;
    return t;
}
template <int BAND> inline constexpr KTab kKTab = make_ktab(BAND);

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// Per (band, head) weight pack, in MFMA-fragment order so that the LDS image is lane-linear:
//   bias0 [6][4][4]            acc init of tile T, lane group q, reg r  = b0[16T + 4r + q]
//   W0    [6][K0/4][64]        lane l of tile T, k-step t: W0[chan(T, l&15)][4t + (l>>4)]
//   bias1 [6][4][4]
//   W1    [6][22][64]
//   bias2 [4][4]               acc init of lane group q, reg r = b2[4q + r]
//   W2    [22][64]             lane l, k-step t: W2[l&15][4t + (l>>4)]
// chan(T, rho) = 16T + 4(rho&3) + (rho>>2): this row permutation makes the accumulator registers of one
// layer line up, untouched, as the B operand of the next layer's MFMAs in natural channel order
// (C/D layout of v_mfma_f32_16x16x4_f32: col = lane&15, row = 4(lane>>4) + reg).
static constexpr int pack_floats(int K0) { return 96 + kMT * (K0 / 4) * 64 + 96 + kMT * kKS1 * 64 + 16 + kKS1 * 64; }

template <int K0>
struct PackOff {
    static constexpr int bias0 = 0;
    static constexpr int w0 = 96;
    static constexpr int bias1 = w0 + kMT * (K0 / 4) * 64;
    static constexpr int w1 = bias1 + 96;
    static constexpr int bias2 = w1 + kMT * kKS1 * 64;
    static constexpr int w2 = bias2 + 16;
    static constexpr int total = w2 + kKS1 * 64;
};
static constexpr int cnn_lds_bytes(int band)
{
    const int K0 = band == 0 ? 48 : band == 1 ? 72 : 120;
    return (pack_floats(K0) + 2 * 3 * (band + 1) * kInPlane) * 4;     // weights + double-buffered input tile
}

__device__ __forceinline__ float relu(float x) { return (x > 0.0f) ? x : 0.0f; }
__device__ __forceinline__ f32x4 relu4(f32x4 v) { v[0] = relu(v[0]); v[1] = relu(v[1]); v[2] = relu(v[2]); v[3] = relu(v[3]); return v; }
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int BAND>
__global__ __launch_bounds__(kCnnThreads) void band_params_kernel(const float *__restrict__ fplanes, Geom g,
                                                                  const float *__restrict__ wpack,
                                                                  float *__restrict__ params, int tiles_x, int tiles_y, int n_tiles)
{
    constexpr int K0 = (BAND == 0) ? 48 : (BAND == 1) ? 72 : 120;
    constexpr int NK0 = K0 / 4;
    constexpr int NPL = 3 * (BAND + 1);          // staged input planes: (x00 | x11 | x01) x (Y, Co, Cg)
    using PO = PackOff<K0>;
    static_assert(kKTab<BAND>.n == NK0, "k-step table");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *lds_in = lds + PO::total;

    const int head = blockIdx.y;
    {   // stage this head's pack (lane-linear image: a straight copy)
        const float4 *src = reinterpret_cast<const float4 *>(wpack + (long)head * PO::total);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < PO::total / 4; i += kCnnThreads) dst[i] = src[i];
    }
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q = lane >> 4;
    const int px = lane & 15;
    const int q_row = q * kInPitch;
    const int pix0 = ((wave * kNT) >> 1) * kInPitch + px;      // + (n>>1)*pitch + 16*(n&1) for pixel tile n

    // Input tile: LDS-DMA (global_load_lds), double buffered.  One wave-instruction fills 64 consecutive LDS
    // floats, so the tile image [plane][20 rows][pitch 48] is cut into NPL*15 such pieces (the 12 pad columns
    // of a row are filled with a duplicate of column 35); each lane computes its own clamped source address.
    // Pitch 48 makes four tile rows exactly three 64-float pieces, so a piece's plane, row group and phase are
    // functions of the wave-uniform piece index (scalar arithmetic) and only ~20 vector operations per piece
    // depend on the lane: piece phase t covers row 4q+t from column 16t (lanes below 48-16t) and the head
    // of row 4q+t+1 (the others).
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto stage = [&](int tile, float *dst) {
        const int img = tile / (tiles_x * tiles_y);
        const int trem = tile - img * (tiles_x * tiles_y);
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        const int i0 = ty * kTileH - 2, j0 = tx * kTileW - 2;
        const float *base = fplanes + (long)img * 3 * g.plane;
        for (int u = wave_u; u < NPL * (kInPlane / 64); u += kCnnThreads / 64) {
            const int pl = u / 15, v = u - 15 * pl, q4 = v / 3, t = v - 3 * q4;       // wave-uniform
            const int src = pl / 3, ci = pl - 3 * src;
            const int thr = 48 - 16 * t;
            const bool up = lane >= thr;
            const int cidx = min(up ? lane - thr : lane + 16 * t, kInCols - 1);
            const int r = 4 * q4 + t + (up ? 1 : 0);
            const int bi = max(0, min(i0 + r, g.h - 1));            // the conv's replicate padding, in band coordinates
            const int bj = max(0, min(j0 + cidx, g.w - 1));
            int rr = 2 * bi + src_oi(src), cc = 2 * bj + src_oj(src);
            if (rr >= g.Hl) rr -= 2;                   // lazyDWT's replicate pad of the odd edge (LLICTI_nets.py:226-240)
            if (cc >= g.Wl) cc -= 2;
            const unsigned off = (unsigned)(rr * g.W + cc) << g.lvl;   // < H * W
            const float *gp = base + (long)ci * g.plane + off;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gp,
                                             (__attribute__((address_space(3))) void *)(dst + u * 64), 4, 0, 0);
        }
    };
    static_assert(kInPitch == 48 && kInRows % 4 == 0 && kInPlane / 64 == 15, "piece decomposition assumes pitch 48, 20 rows");
    static_assert(kInPlane % 64 == 0, "tile plane must be a whole number of 64-float pieces");

    const int stage_site = (__builtin_amdgcn_readfirstlane(wave) / (kCnnThreads / 256)) % CNN_STAGE_SITES;
    int cur = 0;
    if ((int)blockIdx.x < n_tiles) stage(blockIdx.x, lds_in);
#if CNN_STAGGER
    if (__builtin_amdgcn_readfirstlane(wave) >= 4) __builtin_amdgcn_s_sleep(CNN_STAGGER);
#endif
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int img = tile / (tiles_x * tiles_y);
        const int trem = tile - img * (tiles_x * tiles_y);
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        const int i0 = ty * kTileH, j0 = tx * kTileW;
        float *lds_cur = lds_in + cur * (NPL * kInPlane);

        // this tile's pieces have landed (each wave drains its own DMA, then the barrier), and every wave has
        // finished reading the other buffer (previous tile) -- which the next tile's DMA may now overwrite
        __syncthreads();
        // The next tile's DMA (address arithmetic + issue: pure VALU / VMEM work) is requested at four
        // different points of the tile, one per wave group: a SIMD hosts one wave of each group, so while
        // one of its waves stages, the other three keep the matrix pipe busy.  (All 16 waves staging right
        // after the barrier left the pipe idle for ~9 % of the tile.)
        const bool more = tile + (int)gridDim.x < n_tiles;
        auto stage_next = [&](int site) {
            if (more && stage_site == site % CNN_STAGE_SITES) stage(tile + gridDim.x, lds_in + (cur ^ 1) * (NPL * kInPlane));
        };
        stage_next(0);

        // ---- layer 0: [96 x K0] x [K0 x 64 pixels]; bias preloaded into the accumulators
        f32x4 a0[kMT][kNT];
#pragma unroll
        for (int T = 0; T < kMT; ++T) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(lds + PO::bias0 + (T * 4 + q) * 4);
#pragma unroll
            for (int n = 0; n < kNT; ++n) a0[T][n] = bv;
        }
#if CNN_PREFETCH_L0
        {
            float a_c[kMT], b_c[kNT];
            {
                constexpr int U = kKTab<BAND>.s[0].U, S = kKTab<BAND>.s[0].S;
                const float *bp = lds_cur + U + pix0 + (S == 1 ? q : q_row);
#pragma unroll
                for (int n = 0; n < kNT; ++n) b_c[n] = bp[(n >> 1) * kInPitch + 16 * (n & 1)];
#pragma unroll
                for (int T = 0; T < kMT; ++T) a_c[T] = lds[PO::w0 + (T * NK0 + 0) * 64 + lane];
            }
            static_for<NK0>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                float a_n[kMT], b_n[kNT];
                if constexpr (t + 1 < NK0) {       // next k-step's fragments are in flight while this one's MFMAs run
                    constexpr int U = kKTab<BAND>.s[t + 1].U, S = kKTab<BAND>.s[t + 1].S;
                    const float *bp = lds_cur + U + pix0 + (S == 1 ? q : q_row);
#pragma unroll
                    for (int n = 0; n < kNT; ++n) b_n[n] = bp[(n >> 1) * kInPitch + 16 * (n & 1)];
#pragma unroll
                    for (int T = 0; T < kMT; ++T) a_n[T] = lds[PO::w0 + (T * NK0 + t + 1) * 64 + lane];
                }
#pragma unroll
                for (int T = 0; T < kMT; ++T)
#pragma unroll
                    for (int n = 0; n < kNT; ++n) a0[T][n] = MFMA4(a_c[T], b_c[n], a0[T][n]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (t + 1 < NK0) {
#pragma unroll
                    for (int T = 0; T < kMT; ++T) a_c[T] = a_n[T];
#pragma unroll
                    for (int n = 0; n < kNT; ++n) b_c[n] = b_n[n];
                }
            });
        }
#else
        static_for<NK0>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int U = kKTab<BAND>.s[t].U, S = kKTab<BAND>.s[t].S;
            const float *bp = lds_cur + U + pix0 + (S == 1 ? q : q_row);
            float bf[kNT];
#pragma unroll
            for (int n = 0; n < kNT; ++n) bf[n] = bp[(n >> 1) * kInPitch + 16 * (n & 1)];
#pragma unroll
            for (int T = 0; T < kMT; ++T) {
                const float a = lds[PO::w0 + (T * NK0 + t) * 64 + lane];
#pragma unroll
                for (int n = 0; n < kNT; ++n) a0[T][n] = MFMA4(a, bf[n], a0[T][n]);
            }
            __builtin_amdgcn_sched_barrier(0);     // one k-step per scheduling region (bounds VGPR pressure)
        });
#endif
        if constexpr (CNN_STAGE_SITES > 1) stage_next(1);
#pragma unroll
        for (int T = 0; T < kMT; ++T)
#pragma unroll
            for (int n = 0; n < kNT; ++n) a0[T][n] = relu4(a0[T][n]);

        // ---- layers 1 and 2, interleaved per 16-channel tile: the accumulator registers of one layer ARE
        //      the B fragments of the next (k-step tt of the consumer = tile tt>>2, register tt&3)
        f32x4 a2[kNT];
        {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(lds + PO::bias2 + q * 4);
#pragma unroll
            for (int n = 0; n < kNT; ++n) a2[n] = bv;
        }
        static_for<kMT>([&](auto Tc) {
            constexpr int T = decltype(Tc)::value;
            if constexpr (T == 2 && CNN_STAGE_SITES > 2) stage_next(2);
            if constexpr (T == 4 && CNN_STAGE_SITES > 2) stage_next(3);
            f32x4 a1[kNT];
            {
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(lds + PO::bias1 + (T * 4 + q) * 4);
#pragma unroll
                for (int n = 0; n < kNT; ++n) a1[n] = bv;
            }
            static_for<kKS1>([&](auto ttc) {
                constexpr int tt = decltype(ttc)::value;
                const float a = lds[PO::w1 + (T * kKS1 + tt) * 64 + lane];
#pragma unroll
                for (int n = 0; n < kNT; ++n) a1[n] = MFMA4(a, a0[tt >> 2][n][tt & 3], a1[n]);
#if CNN_FENCE_L1 > 0
                if constexpr ((tt % CNN_FENCE_L1) == CNN_FENCE_L1 - 1) __builtin_amdgcn_sched_barrier(0);
#endif
            });
#pragma unroll
            for (int n = 0; n < kNT; ++n) a1[n] = relu4(a1[n]);
            static_for<4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                if constexpr (4 * T + r < kKS1) {
                    const float a = lds[PO::w2 + (4 * T + r) * 64 + lane];
#pragma unroll
                    for (int n = 0; n < kNT; ++n) a2[n] = MFMA4(a, a1[n][r], a2[n]);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });

        // D row 4q + r = output 4q + r of this head; params[pos][head][16]
#pragma unroll
        for (int n = 0; n < kNT; ++n) {
            const int i = i0 + ((wave * kNT) >> 1) + (n >> 1), j = j0 + 16 * (n & 1) + px;
            if (i < g.h && j < g.w)
                *reinterpret_cast<f32x4 *>(params + (((long)img * g.h + i) * g.w + j) * kParamStride + head * 16 + 4 * q) = a2[n];
        }
        cur ^= 1;
    }
}

// host: canonical arrays -> fragment-ordered pack of one band (4 heads)
static void pack_band(int K0, const float *w0, const float *b0, const float *w1, const float *b1,
                      const float *w2, const float *b2, std::vector<float> &out)
{
    const int NK0 = K0 / 4;
    const int total = pack_floats(K0);
    out.assign((size_t)4 * total, 0.0f);
    for (int hd = 0; hd < 4; ++hd) {
        float *p = out.data() + (size_t)hd * total;
        float *bias0 = p, *W0 = p + 96, *bias1 = W0 + kMT * NK0 * 64, *W1 = bias1 + 96;
        float *bias2 = W1 + kMT * kKS1 * 64, *W2 = bias2 + 16;
        for (int T = 0; T < kMT; ++T)
            for (int q = 0; q < 4; ++q)
                for (int r = 0; r < 4; ++r) {
                    const int cl = 16 * T + 4 * r + q;
                    bias0[(T * 4 + q) * 4 + r] = (cl < kHead) ? b0[hd * kHead + cl] : 0.0f;
                    bias1[(T * 4 + q) * 4 + r] = (cl < kHead) ? b1[hd * kHead + cl] : 0.0f;
                }
        for (int T = 0; T < kMT; ++T)
            for (int l = 0; l < 64; ++l) {
                const int rho = l & 15, q = l >> 4;
                const int cl = 16 * T + 4 * (rho & 3) + (rho >> 2);
                for (int t = 0; t < NK0; ++t)
                    W0[(T * NK0 + t) * 64 + l] = (cl < kHead) ? w0[(size_t)(hd * kHead + cl) * K0 + 4 * t + q] : 0.0f;
                for (int t = 0; t < kKS1; ++t)
                    W1[(T * kKS1 + t) * 64 + l] = (cl < kHead) ? w1[(size_t)(hd * kHead + cl) * kHead + 4 * t + q] : 0.0f;
            }
        for (int q = 0; q < 4; ++q)
            for (int r = 0; r < 4; ++r) bias2[q * 4 + r] = (4 * q + r < 15) ? b2[hd * 15 + 4 * q + r] : 0.0f;
        for (int l = 0; l < 64; ++l) {
            const int o = l & 15, q = l >> 4;
            for (int t = 0; t < kKS1; ++t) W2[t * 64 + l] = (o < 15) ? w2[(size_t)(hd * 15 + o) * kHead + 4 * t + q] : 0.0f;
        }
    }
}

__device__ __forceinline__ float dpp_row_shl(float v, int n)   // lane i <- lane i+n within a 16-lane row (n = 1..4)
{
    int r;
    const int iv = __float_as_int(v);
    switch (n) {
    case 1: r = __builtin_amdgcn_update_dpp(0, iv, 0x101, 0xF, 0xF, true); break;
    case 2: r = __builtin_amdgcn_update_dpp(0, iv, 0x102, 0xF, 0xF, true); break;
    case 3: r = __builtin_amdgcn_update_dpp(0, iv, 0x103, 0xF, 0xF, true); break;
    default: r = __builtin_amdgcn_update_dpp(0, iv, 0x104, 0xF, 0xF, true); break;
    }
    return __int_as_float(r);
}

// ------------------------------------------------------------------------------------------------ CDF kernels
struct StageGeom {      // one (level, band): band grid, coded crop, full-res addressing
    int B, H, W, lvl, h, w, hc, wc, oi, oj;
    long plane;
};
static StageGeom make_stage(const Geom &g, int band)
{
    static const int OI[4] = { 0, 1, 0, 1 }, OJ[4] = { 0, 1, 1, 0 };
    StageGeom s;
    s.B = g.B; s.H = g.H; s.W = g.W; s.lvl = g.lvl; s.h = g.h; s.w = g.w; s.plane = g.plane;
    coded_dims(g, band, &s.hc, &s.wc);
    s.oi = OI[band + 1]; s.oj = OJ[band + 1];
    return s;
}

__device__ __forceinline__ void clr_range(const int32_t *mm, int clr, int &minv, int &maxv, int &shift)
{
    // LLICTI_nets.py:394-395, :544-547: Y uses the fixed range [-127,128], Co/Cg the image's own [min,max]
    if (clr == 0) { minv = -127; maxv = 128; shift = 127; }
    else { minv = mm[clr - 1]; maxv = mm[2 + clr - 1]; shift = -minv; }
}

// encoder: thread per coded position; the two entries the coder reads, for Y, Co, Cg
__global__ __launch_bounds__(256) void cdf_pairs_kernel(const int16_t *__restrict__ planes, const float *__restrict__ params,
                                                        const int32_t *__restrict__ minmax, StageGeom s,
                                                        uint32_t *__restrict__ pairs)
{
    const int b = blockIdx.y;
    const long nc = (long)s.hc * s.wc;
    const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nc) return;
    const int i = (int)(n / s.wc), j = (int)(n - (long)i * s.wc);
    const float *par = params + ((long)b * s.h * s.w + (long)i * s.w + j) * LLICTI_PARAM_STRIDE;
    const long off = (long)b * 3 * s.plane + ((long)(2 * i + s.oi) << s.lvl) * s.W + ((long)(2 * j + s.oj) << s.lvl);
    const int vy = planes[off], vco = planes[off + s.plane], vcg = planes[off + 2 * s.plane];
    const float yv = (float)vy / 255.0f, cov = (float)vco / 255.0f;
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
        const uint32_t hi = (sym == gr.Lp - 2) ? 0u : cdf_entry(m, gr, sym + 1);
        pairs[((long)clr * s.B + b) * nc + n] = (hi << 16) | lo;
    }
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
                                                                   uint16_t *__restrict__ tables, int row_stride)
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

    for (int n = wave0; n < nc; n += nwaves) {
        const int i = n / s.wc, j = n - i * s.wc;
        const float *par = params + ((long)b * s.h * s.w + (long)i * s.w + j) * LLICTI_PARAM_STRIDE;
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
        uint16_t *row = tables + ((long)b * nc + n) * row_stride;
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

struct StreamDesc {     // one arithmetic-coded stream of the whole-batch encoder
    long pair_off;      // first (c_low, c_high) pair, in uint32 units
    long out_off;       // slot offset in bytes
    int n;              // symbols
    int cap;            // slot capacity in bytes
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

__global__ __launch_bounds__(64) void ac_decode_kernel(const uint16_t *__restrict__ cdf, int Lp_fixed, int row_stride,
                                                       const uint8_t *__restrict__ in, long in_stride,
                                                       const int32_t *__restrict__ len, int len_stride, long N_, DecOut o)
{
    // Table rows reach the wave through an LDS ring filled by LDS-DMA (global_load_lds_dwordx4: one
    // instruction moves a whole 1 KB row), waited for with explicit vmcnt counts: rows held in registers
    // made the compiler copy them around behind an s_waitcnt vmcnt(0), i.e. one full memory latency per symbol.
    __shared__ uint4 ring[kDecRing][64];
    const int s = blockIdx.x;
    const int lane = threadIdx.x;
    const int N = (int)N_;
    const uint32_t *words = reinterpret_cast<const uint32_t *>(in + (long)s * in_stride);
    (void)len; (void)len_stride;   // streams are zero padded: reads past the end return 0 bits like torchac's get()
    int Lp = Lp_fixed, shift = 0;
    if (o.planes) {
        int minv, maxv;
        clr_range(o.minmax + 4 * s, o.clr, minv, maxv, shift);
        Lp = maxv - minv + 2;
    }
    const uint32_t max_symbol = (uint32_t)(Lp - 2);
    const uint16_t *tab = cdf + (long)s * N * row_stride;
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
    const int in_words = (int)(in_stride >> 2);
    auto load_win = [&](int w0) -> uint32_t { return words[min(w0 + lane, in_words - 1)]; };
    uint32_t win_cur = load_win(0);
    asm volatile("" : "+v"(win_cur));
    int wpos = 3;                                       // next word to pull (wave-uniform)
    auto next_word = [&]() -> uint32_t {
        uint32_t w = bswap32((uint32_t)__builtin_amdgcn_readlane((int)win_cur, wpos & 63));
        if (wpos >= in_words) w = 0;                    // reads past the slot return 0 bits
        ++wpos;
        if ((wpos & 63) == 0) { win_cur = load_win(wpos); asm volatile("" : "+v"(win_cur)); }   // wait for it here, not at every later pull
        return w;
    };
#pragma unroll
    for (int k = 0; k < kDecRing; ++k) dma_row(k, k);
    uint32_t value = bswap32((uint32_t)__builtin_amdgcn_readlane((int)win_cur, 0));
    const uint32_t w1_ = bswap32((uint32_t)__builtin_amdgcn_readlane((int)win_cur, 1)), w2_ = bswap32((uint32_t)__builtin_amdgcn_readlane((int)win_cur, 2));
    uint64_t buf = ((uint64_t)w1_ << 32) | w2_;        // next 64 bits, MSB first (readlane returns a signed int: no sign extension here)
    int have = 64;
    uint32_t low = 0, high = 0xFFFFFFFFu;
    const bool lane_ok = 8u * (uint32_t)lane <= max_symbol;      // this lane's first entry is a real table entry
    const int e = lane & 7;

    // decoded symbols are parked one per lane and written out every 64 symbols (one store wave instead of 64)
    int mysym = 0;
    auto flush = [&](int n_first, int count) {
        if (lane < count) {
            const int n = n_first + lane;
            if (o.sym) o.sym[(long)s * N + n] = (int16_t)mysym;
            if (o.planes) {
                const int i = n / o.sg.wc, j = n - i * o.sg.wc;
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
        const int L = __builtin_popcountll(__ballot(p1)) - 1;
        const uint32_t w0 = __builtin_amdgcn_readlane(cur.x, L), w1 = __builtin_amdgcn_readlane(cur.y, L);
        const uint32_t w2 = __builtin_amdgcn_readlane(cur.z, L), w3 = __builtin_amdgcn_readlane(cur.w, L);
        // round 2: the 8 entries of lane L, one per lane e = lane & 7
        const uint32_t c2 = pick16(w0, w1, w2, w3, e);
        const uint32_t sc2 = __umul24(rh, c2) + ((__umul24(rl, c2) + c2) >> 16);
        const uint32_t idx = 8u * (uint32_t)L + (uint32_t)e;
        const bool p2 = (e == 0) || (idx <= max_symbol && sc2 <= T);
        const int es = __builtin_popcount((uint32_t)__ballot(p2) & 0xFFu) - 1;
        const uint32_t sidx = 8u * (uint32_t)L + (uint32_t)es;
        const uint32_t low_add = __builtin_amdgcn_readlane(sc2, es);
        const uint32_t hi_in = __builtin_amdgcn_readlane(sc2, (es + 1) & 7);      // entry sidx + 1 when es < 7
        const uint32_t hi_nx = __builtin_amdgcn_readlane(sc1, (L + 1) & 63);     // ... when it is the next lane's first entry
        const uint32_t high_add = (sidx == max_symbol) ? r + 1u : (es == 7 ? hi_nx : hi_in);   // top symbol: c_high = 0x10000
        if (lane == (n & 63)) mysym = (int)sidx;
        if ((n & 63) == 63) flush(n - 63, 64);
        // slot's row sits in `cur` (read one iteration ago): refill it with row n + kDecRing
        dma_row(n + kDecRing, slot);
        cur = nxt;
        if (n == N - 1) break;
        high = (low - 1) + high_add;
        low = low + low_add;
        int n1 = __clz((int)(low ^ high));
        if (n1 > 31) n1 = 31;
        if (n1 > 0) {
            low <<= n1;
            high = (high << n1) | ((1u << n1) - 1u);
            value = (value << n1) | (uint32_t)(buf >> (64 - n1));
            buf <<= n1; have -= n1;
            if (have <= 32) {
                buf |= (uint64_t)next_word() << (32 - have);
                have += 32;
            }
        }
        int n2 = min(__clz((int)~(low << 1)), __clz((int)(high << 1)));
        if (n2 > 31) n2 = 31;
        if (n2 > 0) {
            low = (low << n2) & 0x7FFFFFFFu;
            high = ((high << n2) | ((1u << n2) - 1u)) | 0x80000000u;
            value = ((value << n2) ^ 0x80000000u) | (uint32_t)(buf >> (64 - n2));
            buf <<= n2; have -= n2;
            if (have <= 32) {
                buf |= (uint64_t)next_word() << (32 - have);
                have += 32;
            }
        }
    }
    if (N & 63) flush(N & ~63, N & 63);
    VMCNT_WAIT(0);                                      // no transfer may still target this workgroup's LDS at exit
}

// ------------------------------------------------------------------------------------------------ rANS container
// "LLICTI-rANS v1" (new format of this build; BASELINE.json north_star: "torchac replaced by a HIP rANS
// coder").  Same CDFs and symbols as the AC container; each image has M independent streams, each a
// 64-way interleaved rANS coder (32-bit states, 16-bit words, 16-bit probabilities) driven by ONE
// wavefront: lane l of stream m codes symbol n = 64c + l of every chunk c = m (mod M) of every stage.
// Words are shared by the 64 lanes in lane order (ballot + mbcnt prefix), so a whole stage decodes in
// ceil(nc / 64M) wave steps instead of nc serial symbols.
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
    int bad = 0;
    for (int st = LLICTI_NSTREAMS - 1; st >= 0; --st) {      // rANS is LIFO: last decoded symbol first
        const StreamDesc d = desc[(long)st * B + b];
        const int nchunks = (d.n + 63) >> 6;
        if (nchunks <= m) continue;
        const int K = (nchunks - m + M - 1) / M;
        const uint32_t *pp = pairs + d.pair_off;
        // the pair loads do not depend on the coder state: keep three steps in flight.  The loads are
        // unconditional (clamped address) and the raw value is masked only where it is consumed: a select
        // next to the load would make the compiler wait for it on the spot
        auto fetch = [&](int k) -> uint32_t { return pp[min(64 * (m + max(k, 0) * M) + lane, d.n - 1)]; };
        uint32_t r0 = fetch(K - 1), r1 = fetch(K - 2), r2 = fetch(K - 3);
        for (int k = K - 1; k >= 0; --k) {
            const int n = 64 * (m + k * M) + lane;
            const bool active = n < d.n;
            const uint32_t v = active ? r0 : 0x00010000u;
            r0 = r1; r1 = r2; r2 = fetch(k - 3);
            const uint32_t lo = v & 0xFFFFu;
            uint32_t hi = v >> 16;
            if (hi == 0) hi = 0x10000u;
            uint32_t freq = hi - lo;
            if (active && (freq == 0 || hi < lo)) { bad = 1; freq = 1; }
            const bool emit = active && ((uint64_t)x >= ((uint64_t)freq << 16));
            const uint64_t E = __ballot(emit);
            p -= __builtin_popcountll(E);
            if (p < 128) { bad = 2; p = 128; }
            if (emit) { w16[p + lanes_below(E)] = (uint16_t)(x & 0xFFFFu); x >>= 16; }
            if (active) {
                const uint32_t q = x / freq;
                x = (q << 16) + (x - q * freq) + lo;
            }
        }
    }
    p -= 128;                                       // 64 final states, little-endian uint32, lane order
    w16[p + 2 * lane] = (uint16_t)(x & 0xFFFFu);
    w16[p + 2 * lane + 1] = (uint16_t)(x >> 16);
    if (lane == 0) { rinfo[2 * sidx] = (int32_t)(2 * p); rinfo[2 * sidx + 1] = (int32_t)(rslot_cap - 2 * p); }
    if (bad) atomicExch(&status[0], bad == 1 ? LLICTI_EFORMAT : LLICTI_ENOSPACE);
}

__global__ __launch_bounds__(64) void rans_init_kernel(const uint8_t *__restrict__ slots, const long *__restrict__ rslot_off,
                                                       uint32_t *__restrict__ rstate, uint32_t *__restrict__ rpos)
{
    const int sidx = blockIdx.x, lane = threadIdx.x;
    rstate[(long)sidx * 64 + lane] = reinterpret_cast<const uint32_t *>(slots + rslot_off[sidx])[lane];
    if (lane == 0) rpos[sidx] = 0;
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
                                                               int16_t *__restrict__ planes, float *__restrict__ fplanes,
                                                               const int32_t *__restrict__ minmax)
{
    __shared__ uint32_t sh_res[2][64][2];        // ping-pong by step parity: [0] = c_low, [1] = c_high
    const int sidx = blockIdx.x, b = sidx / M, m = sidx - b * M;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nc = sg.hc * sg.wc;
    const int nchunks = (nc + 63) >> 6;
    if (nchunks <= m) return;                    // whole workgroup
    const int K = (nchunks - m + M - 1) / M;
    uint32_t x = rstate[(long)sidx * 64 + lane], pos = rpos[sidx];      // every wave: its own copy
    const uint16_t *words = reinterpret_cast<const uint16_t *>(slots + rslot_off[sidx] + 256);
    const uint32_t max_words = (uint32_t)((rslot_cap - 256) / 2);
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
        const int i = n / sg.wc, j = n - i * sg.wc;
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

                // 1. hint: bisection on the approximate table (a 4-ary round with three probes costs three times
                //    a probe: the phase is instruction-issue bound, not latency bound -- measured with in-kernel stamps)
                const CompFast Af = comp_fast(A), Bf = comp_fast(B);
                int glo = 0, ghi = max_symbol + 1;
                while (ghi - glo > 1) {
                    const int mid = (glo + ghi) >> 1;
                    const int e = group_cdf_entry_fast(Af, Bf, fbase, gr.scale, mid);
                    const uint64_t bal = __ballot(e <= (int)slot);
                    if ((bal >> gbit) & 1ull) glo = mid; else ghi = mid;
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
                    const bool bA = (__ballot(eA <= slot) >> gbit) & 1ull;
                    const bool bB = (__ballot(eB <= slot) >> gbit) & 1ull;
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
                    const uint64_t bal = __ballot(e <= slot);
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
            if (active) {
                const uint32_t vlo = sh_res[k & 1][lane][0], vhi = sh_res[k & 1][lane][1];
                x = (vhi - vlo) * (x >> 16) + (x & 0xFFFFu) - vlo;
            }
            const bool need = active && x < 0x10000u;
            const uint64_t E = __ballot(need);
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
    for (int t = threadIdx.x; t < n; t += blockDim.x) o[t] = src[t];
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
    for (int k = 0; k < 4 + m; ++k) src += sl[k];
    int n = sl[4 + m];
    uint8_t *o = slots + rslot_off[b * M + m];
    if (n < 256 || n > rslot_cap || src + n > in_stride) {
        if (threadIdx.x == 0) atomicExch(&status[0], LLICTI_EFORMAT);
        for (int t = threadIdx.x; t < 256; t += blockDim.x) o[t] = (t & 3) == 2 ? 1 : 0;     // states = 1 << 16: harmless
        n = 256;
    } else {
        const uint8_t *p = in + (long)b * in_stride + src;
        for (int t = threadIdx.x; t < n; t += blockDim.x) o[t] = p[t];
    }
    const int padded = min(rslot_cap, n + 64);
    for (int t = n + threadIdx.x; t < padded; t += blockDim.x) o[t] = 0;
}

// ------------------------------------------------------------------------------------------------ container kernels
// encode: header segments straight into the container; seg_len[b][0..3]
__global__ void header_write_kernel(const uint8_t *__restrict__ rgb, const int32_t *__restrict__ minmax, int H, int W,
                                    int h4, int w4, int padint, int byte0, uint8_t *__restrict__ out, long out_stride,
                                    int32_t *__restrict__ seg_len)
{
    const int b = blockIdx.x;
    uint8_t *o = out + (long)b * out_stride;
    const long plane = (long)H * W;
    if (threadIdx.x == 0) {
        o[0] = (uint8_t)byte0; o[1] = (uint8_t)h4; o[2] = (uint8_t)w4;          // LLICTI_nets.py:347 (AC: number of scales)
        const int32_t *mm = minmax + 4 * b;
        const int16_t v[6] = { 0, (int16_t)mm[0], (int16_t)mm[1], 255, (int16_t)mm[2], (int16_t)mm[3] };   // :139, :348
        for (int k = 0; k < 6; ++k) { o[3 + 2 * k] = (uint8_t)(v[k] & 0xFF); o[4 + 2 * k] = (uint8_t)((v[k] >> 8) & 0xFF); }
        o[15] = (uint8_t)(padint & 0xFF); o[16] = (uint8_t)((padint >> 8) & 0xFF);                         // :349
        int32_t *sl = seg_len + (long)b * LLICTI_NSEG;
        sl[0] = 3; sl[1] = 12; sl[2] = 2; sl[3] = 3 * h4 * w4;
    }
    for (int t = threadIdx.x; t < 3 * h4 * w4; t += blockDim.x) {                                        // :248-252, :350
        const int c = t / (h4 * w4), r = t - c * h4 * w4, i = r / w4, j = r - i * w4;
        o[17 + t] = rgb[(long)b * 3 * plane + c * plane + (long)(32 * i) * W + 32 * j];
    }
}

// encode: copy the 45 slots of image b behind its header, tightly; seg_len[b][4..48]
__global__ __launch_bounds__(256) void pack_kernel(const uint8_t *__restrict__ slots, const long *__restrict__ slot_off,
                                                   const int32_t *__restrict__ slot_len, int B, int hdr_bytes,
                                                   uint8_t *__restrict__ out, long out_stride, int32_t *__restrict__ seg_len,
                                                   int32_t *status)
{
    const int st = blockIdx.x, b = blockIdx.y;
    // slot index: streams are stored stage-major, image-minor (see build_plan)
    long dst = hdr_bytes;
    for (int k = 0; k < st; ++k) dst += slot_len[(long)k * B + b];
    const int n = slot_len[(long)st * B + b];
    if (dst + n > out_stride) { if (threadIdx.x == 0) atomicExch(&status[0], LLICTI_ENOSPACE); return; }
    const uint8_t *src = slots + slot_off[(long)st * B + b];
    uint8_t *o = out + (long)b * out_stride + dst;
    for (int t = threadIdx.x; t < n; t += blockDim.x) o[t] = src[t];
    if (threadIdx.x == 0) seg_len[(long)b * LLICTI_NSEG + 4 + st] = n;
}

// decode: parse + validate header, min/max -> minmax[b][4], DC band -> planes at stride 32
__global__ void header_read_kernel(const uint8_t *__restrict__ in, long in_stride, const int32_t *__restrict__ seg_len,
                                   int H, int W, int h4, int w4, int padint, int byte0, int16_t *__restrict__ planes,
                                   float *__restrict__ fplanes, int32_t *__restrict__ minmax, int32_t *status)
{
    const int b = blockIdx.x;
    const uint8_t *p = in + (long)b * in_stride;
    const int32_t *sl = seg_len + (long)b * LLICTI_NSEG;
    const long plane = (long)H * W;
    __shared__ int ok;
    if (threadIdx.x == 0) {
        const int pad = (int)(int16_t)(p[15] | (p[16] << 8));
        ok = (sl[0] == 3 && sl[1] == 12 && sl[2] == 2 && sl[3] == 3 * h4 * w4 &&
              p[0] == byte0 && p[1] == h4 && p[2] == w4 && pad == padint);               // LLICTI_nets.py:423-428
        if (!ok) atomicExch(&status[0], LLICTI_EFORMAT);
        int16_t v[6];
        for (int k = 0; k < 6; ++k) v[k] = (int16_t)(p[3 + 2 * k] | (p[4 + 2 * k] << 8));
        int32_t *mm = minmax + 4 * b;
        mm[0] = v[1]; mm[1] = v[2]; mm[2] = v[4]; mm[3] = v[5];
        if (v[1] > v[4] || v[2] > v[5] || v[1] < -255 || v[2] < -255 || v[4] > 255 || v[5] > 255) {
            atomicExch(&status[0], LLICTI_EFORMAT);
            mm[0] = mm[1] = -255; mm[2] = mm[3] = 255;
        }
    }
    __syncthreads();
    if (!ok) return;
    const uint8_t *dc = p + 17;
    for (int t = threadIdx.x; t < h4 * w4; t += blockDim.x) {                              // :429-430, :443-444
        const int i = t / w4, j = t - i * w4;
        const int R = dc[t], G = dc[h4 * w4 + t], Bl = dc[2 * h4 * w4 + t];
        const int Co = R - Bl, tt = Bl + (Co >> 1), Cg = G - tt, Y = tt + (Cg >> 1) - 127;
        const long off = (long)b * 3 * plane + (long)(32 * i) * W + 32 * j;
        planes[off] = (int16_t)Y; planes[off + plane] = (int16_t)Co; planes[off + 2 * plane] = (int16_t)Cg;
        fplanes[off] = (float)Y / 255.0f; fplanes[off + plane] = (float)Co / 255.0f; fplanes[off + 2 * plane] = (float)Cg / 255.0f;
    }
}

// decode: copy stream st of image b into its 4-byte aligned, zero padded slot
__global__ __launch_bounds__(256) void unpack_kernel(const uint8_t *__restrict__ in, long in_stride, const int32_t *__restrict__ seg_len,
                                                     int B, uint8_t *__restrict__ slots, const long *__restrict__ slot_off,
                                                     const int32_t *__restrict__ slot_cap, int32_t *status)
{
    const int st = blockIdx.x, b = blockIdx.y;
    const int32_t *sl = seg_len + (long)b * LLICTI_NSEG;
    long src = 0;
    for (int k = 0; k < 4 + st; ++k) src += sl[k];
    int n = sl[4 + st];
    const int cap = slot_cap[(long)st * B + b];
    if (n < 0 || n + 16 > cap || src + n > in_stride) { if (threadIdx.x == 0) atomicExch(&status[0], LLICTI_EFORMAT); n = 0; }
    const uint8_t *p = in + (long)b * in_stride + src;
    uint8_t *o = slots + slot_off[(long)st * B + b];
    for (int t = threadIdx.x; t < n; t += blockDim.x) o[t] = p[t];
    const int padded = min(cap, ((n + 3) & ~3) + 16);
    for (int t = n + threadIdx.x; t < padded; t += blockDim.x) o[t] = 0;
}

// ------------------------------------------------------------------------------------------------ context
struct Plan {                 // workspace carving for (B, H, W)
    int B = 0, H = 0, W = 0;
    size_t off_planes, off_fplanes, off_minmax, off_status, off_params, off_pairs, off_slots, off_slot_len, off_tables;
    size_t total;
    std::vector<StreamDesc> desc;       // stage-major, image-minor: index (stage * B + b)
    std::vector<long> slot_off;
    std::vector<int32_t> slot_cap;
    std::vector<long> pair_base;        // per (lvl, band): first pair of [clr][B][nc]
    size_t max_container;
    int M = 0;                          // rANS streams per image (0: AC container only)
    int rslot_cap = 0;
    std::vector<long> rslot_off;        // [B*M] byte offsets into the slots region
    size_t off_rinfo, off_rstate, off_rpos;
};

constexpr int kMaxSub = 4;
struct PlanDev {
    Plan p;
    StreamDesc *d_desc = nullptr;
    long *d_slot_off = nullptr;
    int32_t *d_slot_cap = nullptr;
    long *d_rslot_off = nullptr;
};

struct llicti_ctx {
    int device = 0;
    float *d_pack[3] = { nullptr, nullptr, nullptr };
    bool have[3] = { false, false, false };
    std::map<std::tuple<int, int, int, int>, struct PlanDev *> plans;   // (B, H, W, M) -> plan + its device arrays
    hipStream_t sub[kMaxSub] = { nullptr, nullptr, nullptr, nullptr };    // sub-batch pipelining (decode)
    hipEvent_t ev_fork = nullptr, ev_join[kMaxSub] = { nullptr, nullptr, nullptr, nullptr };
    int pipeline_s = 4;
    bool pipeline = false;     // sub-batch pipelining of decode: measured neutral (co-resident CNN and rANS waves share VALU issue)
    int32_t *d_status = nullptr;      // small persistent status word (for the kernel-level entry points)
    bool profiling = false;
    std::vector<hipEvent_t> ev;       // pairs around band-CNN launches
    int ev_used = 0;
    hipEvent_t ev_call[2] = { nullptr, nullptr };
    float last_ms[4] = { 0, 0, 0, 0 };
    int last_launches = 0;
    bool timing_pending = false;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int stage_index(int lvl, int band, int clr) { return (LLICTI_NLEVELS - 1 - lvl) * 9 + band * 3 + clr; }   // scale 4..0

static void build_plan(Plan &p, int B, int H, int W, int M)
{
    p.B = B; p.H = H; p.W = W; p.M = M;
    const size_t plane = (size_t)H * W;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    p.off_status = take(64);
    p.off_minmax = take((size_t)B * 4 * sizeof(int32_t));
    p.off_planes = take((size_t)B * 3 * plane * sizeof(int16_t));
    p.off_fplanes = take((size_t)B * 3 * plane * sizeof(float));
    Geom g0 = make_geom(B, H, W, 0);
    p.off_params = take((size_t)B * g0.h * g0.w * LLICTI_PARAM_STRIDE * sizeof(float));
    // pairs + slots
    p.desc.assign((size_t)LLICTI_NSTREAMS * B, StreamDesc{});
    p.slot_off.assign((size_t)LLICTI_NSTREAMS * B, 0);
    p.slot_cap.assign((size_t)LLICTI_NSTREAMS * B, 0);
    p.pair_base.assign(LLICTI_NLEVELS * 3, 0);
    long pair_pos = 0, slot_pos = 0;
    size_t container = 17;
    Geom g4 = make_geom(B, H, W, 4);
    container += 3 * (size_t)g4.h * g4.w;
    for (int lvl = LLICTI_NLEVELS - 1; lvl >= 0; --lvl) {
        Geom g = make_geom(B, H, W, lvl);
        for (int band = 0; band < 3; ++band) {
            int hc, wc;
            coded_dims(g, band, &hc, &wc);
            const long nc = (long)hc * wc;
            p.pair_base[lvl * 3 + band] = pair_pos;
            for (int clr = 0; clr < 3; ++clr) {
                const int st = stage_index(lvl, band, clr);
                const int cap = (int)align_up((size_t)(2 * nc + 8 + 16), 16);   // <= 16 bits per symbol + termination + zero pad
                for (int b = 0; b < B; ++b) {
                    StreamDesc &d = p.desc[(size_t)st * B + b];
                    d.pair_off = pair_pos + ((long)clr * B + b) * nc;
                    d.out_off = slot_pos;
                    d.n = (int)nc;
                    d.cap = cap - 16;
                    p.slot_off[(size_t)st * B + b] = slot_pos;
                    p.slot_cap[(size_t)st * B + b] = cap;
                    slot_pos += cap;
                }
                container += (size_t)(2 * nc + 8);
            }
            pair_pos += 3L * B * nc;
        }
    }
    p.max_container = align_up(container + 64 * 45, 16);
    p.off_pairs = take((size_t)pair_pos * sizeof(uint32_t));
    if (M > 0) {
        // worst case of one stream: every symbol emits a 16-bit word; chunks are dealt round-robin, so a
        // stream gets at most ceil(nchunks / M) chunks of every stage
        long syms = 0;
        for (int st = 0; st < LLICTI_NSTREAMS; ++st) {
            const long nchunks = (p.desc[(size_t)st * B].n + 63) / 64;
            syms += (nchunks + M - 1) / M * 64;
        }
        p.rslot_cap = (int)align_up((size_t)(2 * syms + 256 + 64), 64);
        p.rslot_off.assign((size_t)B * M, 0);
        for (long i = 0; i < (long)B * M; ++i) p.rslot_off[i] = (long)i * p.rslot_cap;
        slot_pos = std::max<long>(slot_pos, (long)B * M * p.rslot_cap);
        p.max_container = std::max(p.max_container, align_up((size_t)(17 + 3 * g4.h * g4.w) + (size_t)M * p.rslot_cap, 16));
    }
    p.off_slots = take((size_t)slot_pos);
    p.off_rinfo = take((size_t)B * 32 * 2 * sizeof(int32_t));
    p.off_rstate = take((size_t)B * 32 * 64 * sizeof(uint32_t));
    p.off_rpos = take((size_t)B * 32 * sizeof(uint32_t));
    p.off_slot_len = take((size_t)LLICTI_NSTREAMS * B * sizeof(int32_t));
    int hc0, wc0;
    coded_dims(g0, 1, &hc0, &wc0);
    p.off_tables = take((size_t)B * g0.h * g0.w * 512 * sizeof(uint16_t));
    p.total = o;
}

// mode: 0 = AC container (torchac-compatible, the reference's format); 0x100 | M = rANS container with M
// streams per image, M in {1,2,4,8,16,32}
static int mode_streams(int mode)
{
    if (mode == 0) return 0;
    if ((mode & ~0xFF) != 0x100) return -1;
    const int M = mode & 0xFF;
    if (M < 1 || M > 32 || (M & (M - 1))) return -1;
    return M;
}
static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

static int sub_batches_max(int B, int M)
{
    if (M == 0) return 1;
    if (B % 4 == 0 && B >= 8) return 4;
    if (B % 2 == 0 && B >= 4) return 2;
    return 1;
}

extern "C" size_t llicti_workspace_bytes(int B, int H, int W, int mode)
{
    const int M = mode_streams(mode);
    if (check_dims(B, H, W) || M < 0) return 0;
    Plan p;
    build_plan(p, B, H, W, M);
    size_t need = p.total;
    const int S = sub_batches_max(B, M);
    if (S > 1) {
        Plan q;
        build_plan(q, B / S, H, W, M);
        need = std::max(need, (size_t)S * align_up(q.total, 256));
    }
    return need;
}
extern "C" size_t llicti_max_container_bytes(int H, int W)
{
    if (check_dims(1, H, W)) return 0;
    Plan p;
    build_plan(p, 1, H, W, 32);     // covers every mode
    return p.max_container;
}

extern "C" int llicti_create(llicti_ctx **out, int device)
{
    if (!out) return fail(LLICTI_EINVAL, "create: null ctx pointer");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(LLICTI_ENODEVICE, "no HIP device: this library has no CPU path");
    if (device < 0 || device >= n) return fail(LLICTI_EINVAL, "create: device %d out of range (%d devices)", device, n);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(LLICTI_ENODEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    llicti_ctx *c = new llicti_ctx();
    c->device = device;
    if (const char *e = getenv("LLICTI_PIPELINE")) { c->pipeline = atoi(e) != 0; c->pipeline_s = atoi(e); }     // experiment switch: sub-batch pipelining of decode
    HIPCHK(hipMalloc(&c->d_status, 64));
    HIPCHK(hipMemset(c->d_status, 0, 64));
    HIPCHK(hipEventCreate(&c->ev_call[0]));
    HIPCHK(hipEventCreate(&c->ev_call[1]));
    HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    for (int i = 1; i < kMaxSub; ++i) {
        HIPCHK(hipStreamCreateWithFlags(&c->sub[i], hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
    }
    // the band CNN stages a whole head (up to 86 KB) in LDS
    HIPCHK(hipFuncSetAttribute((const void *)band_params_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, cnn_lds_bytes(0)));
    HIPCHK(hipFuncSetAttribute((const void *)band_params_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, cnn_lds_bytes(1)));
    HIPCHK(hipFuncSetAttribute((const void *)band_params_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, cnn_lds_bytes(2)));
    *out = c;
    return LLICTI_OK;
}

extern "C" int llicti_destroy(llicti_ctx *c)
{
    if (!c) return LLICTI_OK;
    for (int b = 0; b < 3; ++b) if (c->d_pack[b]) hipFree(c->d_pack[b]);
    for (auto &kv : c->plans) {
        PlanDev *pd = kv.second;
        if (pd->d_desc) hipFree(pd->d_desc);
        if (pd->d_slot_off) hipFree(pd->d_slot_off);
        if (pd->d_slot_cap) hipFree(pd->d_slot_cap);
        if (pd->d_rslot_off) hipFree(pd->d_rslot_off);
        delete pd;
    }
    for (int i = 0; i < kMaxSub; ++i) {
        if (c->sub[i]) hipStreamDestroy(c->sub[i]);
        if (c->ev_join[i]) hipEventDestroy(c->ev_join[i]);
    }
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->d_status) hipFree(c->d_status);
    for (auto e : c->ev) hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) if (c->ev_call[i]) hipEventDestroy(c->ev_call[i]);
    delete c;
    return LLICTI_OK;
}

extern "C" int llicti_set_band_weights(llicti_ctx *c, int band, int K0, const float *w0, const float *b0,
                                       const float *w1, const float *b1, const float *w2, const float *b2)
{
    if (!c || band < 0 || band > 2 || !w0 || !b0 || !w1 || !b1 || !w2 || !b2) return fail(LLICTI_EINVAL, "set_band_weights: bad argument");
    static const int K0s[3] = { 48, 72, 120 };
    if (K0 != K0s[band]) return fail(LLICTI_EINVAL, "set_band_weights: band %d needs K0=%d, got %d", band, K0s[band], K0);
    std::vector<float> pk;
    pack_band(K0, w0, b0, w1, b1, w2, b2, pk);
    HIPCHK(hipSetDevice(c->device));
    if (!c->d_pack[band]) HIPCHK(hipMalloc(&c->d_pack[band], pk.size() * sizeof(float)));
    HIPCHK(hipMemcpy(c->d_pack[band], pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
    c->have[band] = true;
    return LLICTI_OK;
}

extern "C" int llicti_set_profiling(llicti_ctx *c, int enable)
{
    if (!c) return fail(LLICTI_EINVAL, "null ctx");
    c->profiling = enable != 0;
    return LLICTI_OK;
}

// ------------------------------------------------------------------------------------------------ launches
static int launch_lift(const uint8_t *d_rgb, int B, int H, int W, int16_t *planes, float *fplanes, int32_t *mm, hipStream_t s)
{
    const long plane = (long)H * W;
    minmax_init_kernel<<<(B + 63) / 64, 64, 0, s>>>(mm, B);
    const bool vec = (plane % 4 == 0) && (((uintptr_t)d_rgb | (uintptr_t)planes | (uintptr_t)fplanes) % 16 == 0);
    if (vec) {
        const int gx = (int)std::min<long>((plane / 4 + 255) / 256, std::max(8, 4096 / B));
        lift_kernel<4><<<dim3(gx, B), 256, 0, s>>>(d_rgb, plane, planes, fplanes, mm);
    } else {
        const int gx = (int)std::min<long>((plane + 255) / 256, std::max(8, 4096 / B));
        lift_kernel<1><<<dim3(gx, B), 256, 0, s>>>(d_rgb, plane, planes, fplanes, mm);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

static int launch_band_params(llicti_ctx *c, const float *fplanes, const Geom &g, int band, float *params, hipStream_t s)
{
    if (!c->have[band]) return fail(LLICTI_ENOWEIGHTS, "band %d weights not set", band);
    const int tiles_x = (g.w + kTileW - 1) / kTileW, tiles_y = (g.h + kTileH - 1) / kTileH;
    const long n_tiles_l = (long)g.B * tiles_x * tiles_y;
    if (n_tiles_l > 0x7FFFFFFFL) return fail(LLICTI_EINVAL, "band_params: too many tiles");
    const int n_tiles = (int)n_tiles_l;
    const int lds_bytes = cnn_lds_bytes(band);
    const int wg_per_cu = std::max(1, std::min(4, (160 * 1024) / lds_bytes));
    int gx = std::min(n_tiles, 256 * wg_per_cu / 4);    // 4 heads in grid.y; persistent loop over tiles
    if (gx < 1) gx = 1;
    dim3 grid((unsigned)gx, 4);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->profiling) {
        if ((int)c->ev.size() < c->ev_used + 2) {
            hipEvent_t a, b;
            HIPCHK(hipEventCreate(&a));
            HIPCHK(hipEventCreate(&b));
            c->ev.push_back(a);
            c->ev.push_back(b);
        }
        e0 = c->ev[c->ev_used]; e1 = c->ev[c->ev_used + 1];
        c->ev_used += 2;
        HIPCHK(hipEventRecord(e0, s));
    }
    switch (band) {
    case 0: band_params_kernel<0><<<grid, kCnnThreads, lds_bytes, s>>>(fplanes, g, c->d_pack[0], params, tiles_x, tiles_y, n_tiles); break;
    case 1: band_params_kernel<1><<<grid, kCnnThreads, lds_bytes, s>>>(fplanes, g, c->d_pack[1], params, tiles_x, tiles_y, n_tiles); break;
    default: band_params_kernel<2><<<grid, kCnnThreads, lds_bytes, s>>>(fplanes, g, c->d_pack[2], params, tiles_x, tiles_y, n_tiles); break;
    }
    if (c->profiling) HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int llicti_lift_u8(llicti_ctx *c, const uint8_t *d_rgb, int B, int H, int W, int16_t *d_planes,
                              float *d_fplanes, int32_t *d_minmax, void *stream)
{
    if (!c || !d_rgb || !d_planes || !d_fplanes || !d_minmax) return fail(LLICTI_EINVAL, "lift: null pointer");
    if (check_dims(B, H, W)) return LLICTI_EINVAL;
    return launch_lift(d_rgb, B, H, W, d_planes, d_fplanes, d_minmax, (hipStream_t)stream);
}

extern "C" int llicti_unlift_u8(llicti_ctx *c, const int16_t *d_planes, int B, int H, int W, uint8_t *d_rgb, void *stream)
{
    if (!c || !d_planes || !d_rgb) return fail(LLICTI_EINVAL, "unlift: null pointer");
    if (check_dims(B, H, W)) return LLICTI_EINVAL;
    const long plane = (long)H * W;
    const int gx = (int)std::min<long>((plane + 255) / 256, 1024);
    unlift_kernel<<<dim3(gx, B), 256, 0, (hipStream_t)stream>>>(d_planes, plane, d_rgb);
    HIPCHK(hipGetLastError());
    return LLICTI_OK;
}

extern "C" int llicti_band_params_f32(llicti_ctx *c, const float *d_fplanes, int B, int H, int W, int lvl, int band,
                                      float *d_params, void *stream)
{
    if (!c || !d_fplanes || !d_params) return fail(LLICTI_EINVAL, "band_params: null pointer");
    if (check_dims(B, H, W)) return LLICTI_EINVAL;
    if (lvl < 0 || lvl >= LLICTI_NLEVELS || band < 0 || band > 2) return fail(LLICTI_EINVAL, "band_params: bad level/band");
    Geom g = make_geom(B, H, W, lvl);
    return launch_band_params(c, d_fplanes, g, band, d_params, (hipStream_t)stream);
}

extern "C" int llicti_lift_train_f32(llicti_ctx *c, const uint8_t *d_rgb, int B, int H, int W, float *d_fplanes, void *stream)
{
    if (!c || !d_rgb || !d_fplanes) return fail(LLICTI_EINVAL, "lift_train: null pointer");
    if (check_dims(B, H, W)) return LLICTI_EINVAL;
    const long plane = (long)H * W;
    const int gx = (int)std::min<long>((plane + 255) / 256, 2048);
    lift_train_kernel<<<dim3(gx, B), 256, 0, (hipStream_t)stream>>>(d_rgb, plane, d_fplanes);
    HIPCHK(hipGetLastError());
    return LLICTI_OK;
}

extern "C" int llicti_selfinfo_f32(llicti_ctx *c, const float *d_fplanes, const float *d_params, int B, int H, int W,
                                   int lvl, int band, float *d_bits, void *stream)
{
    if (!c || !d_fplanes || !d_params || !d_bits) return fail(LLICTI_EINVAL, "selfinfo: null pointer");
    if (check_dims(B, H, W)) return LLICTI_EINVAL;
    if (lvl < 0 || lvl >= LLICTI_NLEVELS || band < 0 || band > 2) return fail(LLICTI_EINVAL, "selfinfo: bad level/band");
    Geom g = make_geom(B, H, W, lvl);
    StageGeom sg = make_stage(g, band);
    SelfGeom s;
    s.B = B; s.H = H; s.W = W; s.lvl = lvl; s.h = g.h; s.w = g.w; s.oi = sg.oi; s.oj = sg.oj; s.Hl = g.Hl; s.Wl = g.Wl; s.plane = g.plane;
    const long n = (long)g.h * g.w;
    selfinfo_kernel<<<dim3((unsigned)((n + 255) / 256), B), 256, 0, (hipStream_t)stream>>>(d_fplanes, d_params, s, d_bits);
    HIPCHK(hipGetLastError());
    return LLICTI_OK;
}

static int launch_cdf_pairs(const int16_t *planes, const float *params, const int32_t *mm, const Geom &g, int band,
                            uint32_t *pairs, hipStream_t s)
{
    StageGeom sg = make_stage(g, band);
    const long nc = (long)sg.hc * sg.wc;
    cdf_pairs_kernel<<<dim3((unsigned)((nc + 255) / 256), g.B), 256, 0, s>>>(planes, params, mm, sg, pairs);
    HIPCHK(hipGetLastError());
    return 0;
}
static int launch_cdf_table(const int16_t *planes, const float *params, const int32_t *mm, const Geom &g, int band, int clr,
                            uint16_t *tables, int row_stride, hipStream_t s)
{
    StageGeom sg = make_stage(g, band);
    const long nc = (long)sg.hc * sg.wc;
    const long want = (nc + kTabWaves - 1) / kTabWaves;                     // one row per wave ...
    const long cap = std::max<long>(1, (256L * 8 * 2) / std::max(1, g.B));   // ... up to ~16 waves per SIMD-quad in flight per image set
    cdf_table_kernel<<<dim3((unsigned)std::min(want, cap), g.B), 64 * kTabWaves, 0, s>>>(planes, params, mm, sg, clr, tables, row_stride);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int llicti_cdf_u16(llicti_ctx *c, const int16_t *d_planes, const float *d_params, const int32_t *d_minmax,
                              int B, int H, int W, int lvl, int band, int clr, uint16_t *d_tables, int row_stride, void *stream)
{
    if (!c || !d_planes || !d_params || !d_minmax || !d_tables) return fail(LLICTI_EINVAL, "cdf_u16: null pointer");
    if (check_dims(B, H, W)) return LLICTI_EINVAL;
    if (lvl < 0 || lvl >= LLICTI_NLEVELS || band < 0 || band > 2 || clr < 0 || clr > 2) return fail(LLICTI_EINVAL, "cdf_u16: bad level/band/clr");
    if (row_stride < 8 || row_stride > 512 || (row_stride & 7)) return fail(LLICTI_EINVAL, "cdf_u16: row_stride must be a multiple of 8 in [8,512]");
    Geom g = make_geom(B, H, W, lvl);
    return launch_cdf_table(d_planes, d_params, d_minmax, g, band, clr, d_tables, row_stride, (hipStream_t)stream);
}

extern "C" int llicti_cdf_pairs_u32(llicti_ctx *c, const int16_t *d_planes, const float *d_params, const int32_t *d_minmax,
                                    int B, int H, int W, int lvl, int band, uint32_t *d_pairs, void *stream)
{
    if (!c || !d_planes || !d_params || !d_minmax || !d_pairs) return fail(LLICTI_EINVAL, "cdf_pairs: null pointer");
    if (check_dims(B, H, W)) return LLICTI_EINVAL;
    if (lvl < 0 || lvl >= LLICTI_NLEVELS || band < 0 || band > 2) return fail(LLICTI_EINVAL, "cdf_pairs: bad level/band");
    Geom g = make_geom(B, H, W, lvl);
    return launch_cdf_pairs(d_planes, d_params, d_minmax, g, band, d_pairs, (hipStream_t)stream);
}

extern "C" int llicti_ac_encode_u16cdf(llicti_ctx *c, const uint16_t *d_cdf, int Lp, int row_stride, const int16_t *d_sym,
                                       int n_streams, long N, uint8_t *d_out, long out_stride, int32_t *d_len, void *stream)
{
    if (!c || !d_cdf || !d_sym || !d_out || !d_len) return fail(LLICTI_EINVAL, "ac_encode: null pointer");
    if (Lp < 2 || Lp > 65536 || row_stride < Lp || n_streams < 1 || N < 1 || out_stride < 8 || (out_stride & 3) || ((uintptr_t)d_out & 3))
        return fail(LLICTI_EINVAL, "ac_encode: bad argument (out_stride and d_out must be multiples of 4)");
    ac_encode_tables_kernel<<<(n_streams + 63) / 64, 64, 0, (hipStream_t)stream>>>(d_cdf, Lp, row_stride, d_sym, n_streams, N, d_out,
                                                                                  out_stride, d_len, c->d_status);
    HIPCHK(hipGetLastError());
    return LLICTI_OK;
}

extern "C" int llicti_ac_decode_u16cdf(llicti_ctx *c, const uint16_t *d_cdf, int Lp, int row_stride, const uint8_t *d_in,
                                       long in_stride, const int32_t *d_len, int n_streams, long N, int16_t *d_sym, void *stream)
{
    if (!c || !d_cdf || !d_in || !d_len || !d_sym) return fail(LLICTI_EINVAL, "ac_decode: null pointer");
    if (Lp < 2 || Lp > 512 || row_stride < Lp || row_stride > 512 || (row_stride & 7) || n_streams < 1 || N < 1 || (in_stride & 3) ||
        ((uintptr_t)d_in & 3) || ((uintptr_t)d_cdf & 15))
        return fail(LLICTI_EINVAL, "ac_decode: bad argument (Lp<=512, row_stride multiple of 8, 4-byte aligned streams, 16-byte aligned tables)");
    DecOut o;
    memset(&o, 0, sizeof o);
    o.sym = d_sym;
    ac_decode_kernel<<<n_streams, 64, 0, (hipStream_t)stream>>>(d_cdf, Lp, row_stride, d_in, in_stride, d_len, 1, N, o);
    HIPCHK(hipGetLastError());
    return LLICTI_OK;
}

// ------------------------------------------------------------------------------------------------ whole batch
static int get_plan(llicti_ctx *c, int B, int H, int W, int M, PlanDev **out)
{
    auto key = std::make_tuple(B, H, W, M);
    auto it = c->plans.find(key);
    if (it != c->plans.end()) { *out = it->second; return 0; }
    if (c->plans.size() >= 16) {      // bounded cache: drop everything (plans are cheap to rebuild)
        HIPCHK(hipDeviceSynchronize());
        for (auto &kv : c->plans) {
            PlanDev *pd = kv.second;
            hipFree(pd->d_desc); hipFree(pd->d_slot_off); hipFree(pd->d_slot_cap);
            if (pd->d_rslot_off) hipFree(pd->d_rslot_off);
            delete pd;
        }
        c->plans.clear();
    }
    PlanDev *pd = new PlanDev();
    build_plan(pd->p, B, H, W, M);
    if (M > 0) {
        HIPCHK(hipMalloc(&pd->d_rslot_off, (size_t)B * M * sizeof(long)));
        HIPCHK(hipMemcpy(pd->d_rslot_off, pd->p.rslot_off.data(), (size_t)B * M * sizeof(long), hipMemcpyHostToDevice));
    }
    const size_t n = (size_t)LLICTI_NSTREAMS * B;
    HIPCHK(hipMalloc(&pd->d_desc, n * sizeof(StreamDesc)));
    HIPCHK(hipMalloc(&pd->d_slot_off, n * sizeof(long)));
    HIPCHK(hipMalloc(&pd->d_slot_cap, n * sizeof(int32_t)));
    HIPCHK(hipMemcpy(pd->d_desc, pd->p.desc.data(), n * sizeof(StreamDesc), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(pd->d_slot_off, pd->p.slot_off.data(), n * sizeof(long), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(pd->d_slot_cap, pd->p.slot_cap.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
    c->plans[key] = pd;
    *out = pd;
    return 0;
}

// decode runs as S sub-batches on S streams: the rANS stage kernels are latency bound and occupy a few per
// cent of the GPU, so one sub-batch's stages overlap another's CNN launches (images are independent)
static int sub_batches(const llicti_ctx *c, int B, int M)
{
    if (!c->pipeline || c->profiling || M == 0) return 1;
    if (c->pipeline_s >= 4 && B % 4 == 0 && B >= 8) return 4;
    if (B % 2 == 0 && B >= 4) return 2;
    return 1;
}

__global__ void latch_status_kernel(const int32_t *status, int32_t *latched)
{
    if (*status != 0) *latched = *status;
}

static int pad_int(int H, int W)
{
    int v = 0;
    for (int l = 0; l < LLICTI_NLEVELS; ++l) {
        Geom g = make_geom(1, H, W, l);
        v = 4 * v + 2 * g.padH + g.padW;       // LLICTI_nets.py:230
    }
    return v;
}

static void begin_call(llicti_ctx *c, hipStream_t s)
{
    c->ev_used = 0;
    if (c->profiling) hipEventRecord(c->ev_call[0], s);
}
static void end_call(llicti_ctx *c, hipStream_t s)
{
    if (c->profiling) { hipEventRecord(c->ev_call[1], s); c->timing_pending = true; }
}

extern "C" int llicti_encode_images(llicti_ctx *c, const uint8_t *d_rgb, int B, int H, int W, int mode,
                                    void *d_workspace, size_t workspace_bytes,
                                    uint8_t *d_out, size_t out_stride, int32_t *d_seg_len, void *stream)
{
    if (!c || !d_rgb || !d_workspace || !d_out || !d_seg_len) return fail(LLICTI_EINVAL, "encode_images: null pointer");
    if (check_dims(B, H, W)) return LLICTI_EINVAL;
    const int M = mode_streams(mode);
    if (M < 0) return fail(LLICTI_EINVAL, "encode_images: unknown mode 0x%x", mode);
    for (int b = 0; b < 3; ++b) if (!c->have[b]) return fail(LLICTI_ENOWEIGHTS, "band %d weights not set", b);
    HIPCHK(hipSetDevice(c->device));
    PlanDev *pd = nullptr;
    if (int rc = get_plan(c, B, H, W, M, &pd)) return rc;
    const Plan &p = pd->p;
    if (workspace_bytes < p.total) return fail(LLICTI_ENOSPACE, "encode_images: workspace %zu < %zu", workspace_bytes, p.total);
    if (out_stride < p.max_container) return fail(LLICTI_ENOSPACE, "encode_images: out_stride %zu < %zu", out_stride, p.max_container);
    hipStream_t s = (hipStream_t)stream;
    uint8_t *ws = (uint8_t *)d_workspace;
    int16_t *planes = (int16_t *)(ws + p.off_planes);
    float *fplanes = (float *)(ws + p.off_fplanes);
    int32_t *mm = (int32_t *)(ws + p.off_minmax);
    int32_t *status = (int32_t *)(ws + p.off_status);
    float *params = (float *)(ws + p.off_params);
    uint32_t *pairs = (uint32_t *)(ws + p.off_pairs);
    uint8_t *slots = ws + p.off_slots;
    int32_t *slot_len = (int32_t *)(ws + p.off_slot_len);

    begin_call(c, s);
    HIPCHK(hipMemsetAsync(status, 0, 64, s));
    if (int rc = launch_lift(d_rgb, B, H, W, planes, fplanes, mm, s)) return rc;
    Geom g4 = make_geom(B, H, W, 4);
    const int byte0 = M ? (0x80 | (ilog2(M) << 4) | LLICTI_NLEVELS) : LLICTI_NLEVELS;
    header_write_kernel<<<B, 256, 0, s>>>(d_rgb, mm, H, W, g4.h, g4.w, pad_int(H, W), byte0, d_out, (long)out_stride, d_seg_len);
    // the encoder has no dependency between stages: every (level, band) reads only original pixels
    for (int lvl = LLICTI_NLEVELS - 1; lvl >= 0; --lvl) {
        Geom g = make_geom(B, H, W, lvl);
        for (int band = 0; band < 3; ++band) {
            if (int rc = launch_band_params(c, fplanes, g, band, params, s)) return rc;
            if (int rc = launch_cdf_pairs(planes, params, mm, g, band, pairs + p.pair_base[lvl * 3 + band], s)) return rc;
        }
    }
    const int hdr_bytes = 17 + 3 * g4.h * g4.w;
    if (M == 0) {
        const int n_streams = LLICTI_NSTREAMS * B;
        ac_encode_pairs_kernel<<<n_streams, 64, 0, s>>>(pairs, pd->d_desc, n_streams, slots, slot_len, status);
        pack_kernel<<<dim3(LLICTI_NSTREAMS, B), 256, 0, s>>>(slots, pd->d_slot_off, slot_len, B, hdr_bytes, d_out, (long)out_stride, d_seg_len, status);
    } else {
        int32_t *rinfo = (int32_t *)(ws + p.off_rinfo);
        rans_encode_kernel<<<B * M, 64, 0, s>>>(pairs, pd->d_desc, B, M, slots, pd->d_rslot_off, p.rslot_cap, rinfo, status);
        rans_pack_kernel<<<dim3(M, B), 256, 0, s>>>(slots, pd->d_rslot_off, rinfo, M, hdr_bytes, d_out, (long)out_stride, d_seg_len, status);
    }
    latch_status_kernel<<<1, 1, 0, s>>>(status, c->d_status);
    HIPCHK(hipGetLastError());
    end_call(c, s);
    return LLICTI_OK;
}

static int decode_sub(llicti_ctx *c, PlanDev *pd, const uint8_t *d_in, size_t in_stride, const int32_t *d_seg_len,
                      int B, int H, int W, int M, uint8_t *ws, uint8_t *d_rgb, hipStream_t s)
{
    const Plan &p = pd->p;
    int16_t *planes = (int16_t *)(ws + p.off_planes);
    float *fplanes = (float *)(ws + p.off_fplanes);
    int32_t *mm = (int32_t *)(ws + p.off_minmax);
    int32_t *status = (int32_t *)(ws + p.off_status);
    float *params = (float *)(ws + p.off_params);
    uint8_t *slots = ws + p.off_slots;
    uint16_t *tables = (uint16_t *)(ws + p.off_tables);

    HIPCHK(hipMemsetAsync(status, 0, 64, s));
    Geom g4 = make_geom(B, H, W, 4);
    const int byte0 = M ? (0x80 | (ilog2(M) << 4) | LLICTI_NLEVELS) : LLICTI_NLEVELS;
    header_read_kernel<<<B, 256, 0, s>>>(d_in, (long)in_stride, d_seg_len, H, W, g4.h, g4.w, pad_int(H, W), byte0, planes, fplanes, mm, status);
    uint32_t *rstate = (uint32_t *)(ws + p.off_rstate);
    uint32_t *rpos = (uint32_t *)(ws + p.off_rpos);
    if (M == 0) {
        unpack_kernel<<<dim3(LLICTI_NSTREAMS, B), 256, 0, s>>>(d_in, (long)in_stride, d_seg_len, B, slots, pd->d_slot_off, pd->d_slot_cap, status);
    } else {
        rans_unpack_kernel<<<dim3(M, B), 256, 0, s>>>(d_in, (long)in_stride, d_seg_len, M, slots, pd->d_rslot_off, p.rslot_cap, status);
        rans_init_kernel<<<B * M, 64, 0, s>>>(slots, pd->d_rslot_off, rstate, rpos);
    }
    // 45 dependent stages (LLICTI_nets.py:440-498): CNN of band b needs bands < b of this level, Co needs Y, Cg needs Y, Co
    for (int lvl = LLICTI_NLEVELS - 1; lvl >= 0; --lvl) {
        Geom g = make_geom(B, H, W, lvl);
        for (int band = 0; band < 3; ++band) {
            if (int rc = launch_band_params(c, fplanes, g, band, params, s)) return rc;
            StageGeom sg = make_stage(g, band);
            const long nc = (long)sg.hc * sg.wc;
            if (M > 0) {
                rans_decode_stage_kernel<0><<<B * M, 64 * kRansWaves, 0, s>>>(params, sg, M, slots, pd->d_rslot_off, p.rslot_cap, rstate, rpos, planes, fplanes, mm);
                rans_decode_stage_kernel<1><<<B * M, 64 * kRansWaves, 0, s>>>(params, sg, M, slots, pd->d_rslot_off, p.rslot_cap, rstate, rpos, planes, fplanes, mm);
                rans_decode_stage_kernel<2><<<B * M, 64 * kRansWaves, 0, s>>>(params, sg, M, slots, pd->d_rslot_off, p.rslot_cap, rstate, rpos, planes, fplanes, mm);
            }
            for (int clr = 0; clr < 3 && M == 0; ++clr) {
                const int row_stride = (clr == 0) ? 264 : 512;      // Y: Lp = 257; Co/Cg: Lp <= 512
                if (int rc = launch_cdf_table(planes, params, mm, g, band, clr, tables, row_stride, s)) return rc;
                const int st = stage_index(lvl, band, clr);
                DecOut o;
                memset(&o, 0, sizeof o);
                o.planes = planes; o.fplanes = fplanes; o.minmax = mm; o.sg = sg; o.clr = clr;
                // the B streams of one stage sit in consecutive slots of equal capacity
                const long in_stride_slots = p.slot_cap[(size_t)st * B];
                ac_decode_kernel<<<B, 64, 0, s>>>(tables, 0, row_stride, slots + p.slot_off[(size_t)st * B], in_stride_slots,
                                                  nullptr, 0, nc, o);
            }
        }
    }
    const long plane = (long)H * W;
    const int gx = (int)std::min<long>((plane + 255) / 256, 1024);
    unlift_kernel<<<dim3(gx, B), 256, 0, s>>>(planes, plane, d_rgb);
    latch_status_kernel<<<1, 1, 0, s>>>(status, c->d_status);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int llicti_decode_images(llicti_ctx *c, const uint8_t *d_in, size_t in_stride, const int32_t *d_seg_len,
                                    int B, int H, int W, int mode, void *d_workspace, size_t workspace_bytes,
                                    uint8_t *d_rgb, void *stream)
{
    if (!c || !d_in || !d_seg_len || !d_workspace || !d_rgb) return fail(LLICTI_EINVAL, "decode_images: null pointer");
    if (check_dims(B, H, W)) return LLICTI_EINVAL;
    const int M = mode_streams(mode);
    if (M < 0) return fail(LLICTI_EINVAL, "decode_images: unknown mode 0x%x", mode);
    for (int b = 0; b < 3; ++b) if (!c->have[b]) return fail(LLICTI_ENOWEIGHTS, "band %d weights not set", b);
    HIPCHK(hipSetDevice(c->device));
    const int S = sub_batches(c, B, M);
    const int Bs = B / S;
    PlanDev *pd = nullptr;
    if (int rc = get_plan(c, Bs, H, W, M, &pd)) return rc;
    const size_t sub_total = align_up(pd->p.total, 256);
    if (workspace_bytes < (size_t)S * sub_total)
        return fail(LLICTI_ENOSPACE, "decode_images: workspace %zu < %zu", workspace_bytes, (size_t)S * sub_total);
    hipStream_t s = (hipStream_t)stream;
    uint8_t *ws = (uint8_t *)d_workspace;
    const size_t plane3 = (size_t)3 * H * W;

    begin_call(c, s);
    if (S > 1) HIPCHK(hipEventRecord(c->ev_fork, s));
    for (int k = 0; k < S; ++k) {
        hipStream_t sk = (k == 0) ? s : c->sub[k];
        if (k > 0) HIPCHK(hipStreamWaitEvent(sk, c->ev_fork, 0));
        if (int rc = decode_sub(c, pd, d_in + (size_t)k * Bs * in_stride, in_stride, d_seg_len + (size_t)k * Bs * LLICTI_NSEG,
                                Bs, H, W, M, ws + (size_t)k * sub_total, d_rgb + (size_t)k * Bs * plane3, sk))
            return rc;
        if (k > 0) HIPCHK(hipEventRecord(c->ev_join[k], sk));
    }
    for (int k = 1; k < S; ++k) HIPCHK(hipStreamWaitEvent(s, c->ev_join[k], 0));
    end_call(c, s);
    return LLICTI_OK;
}

extern "C" int llicti_check_status(llicti_ctx *c, void *stream)
{
    if (!c) return fail(LLICTI_EINVAL, "null ctx");
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    int32_t st = 0;
    HIPCHK(hipMemcpy(&st, c->d_status, 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(c->d_status, 0, 4));
    if (st == LLICTI_EFORMAT) return fail(LLICTI_EFORMAT, "malformed container (header does not match the requested shape, or a stream is too long)");
    if (st == LLICTI_ENOSPACE) return fail(LLICTI_ENOSPACE, "output buffer too small for the encoded streams");
    if (st != 0) return fail(st, "device-side status %d", st);
    return LLICTI_OK;
}

extern "C" int llicti_header_dims(const uint8_t *h, int *H, int *W)
{
    if (!h || !H || !W) return fail(LLICTI_EINVAL, "header_dims: null pointer");
    if (h[0] != LLICTI_NLEVELS && (h[0] & 0x8F) != (0x80 | LLICTI_NLEVELS))
        return fail(LLICTI_EFORMAT, "header: byte 0 = 0x%02x is neither %d scales (AC container) nor a rANS container tag", h[0], LLICTI_NLEVELS);
    int Hc = h[1], Wc = h[2];
    int pad = (int)(int16_t)(h[15] | (h[16] << 8));
    for (int l = LLICTI_NLEVELS - 1; l >= 0; --l) {     // _get_padHW_lev_list, LLICTI_nets.py:533-542
        const int padW = pad & 1; pad >>= 1;
        const int padH = pad & 1; pad >>= 1;
        Hc = 2 * Hc - padH;
        Wc = 2 * Wc - padW;
    }
    *H = Hc; *W = Wc;
    return LLICTI_OK;
}

extern "C" int llicti_last_timing(llicti_ctx *c, float ms[4], int *n_launch)
{
    if (!c || !ms) return fail(LLICTI_EINVAL, "last_timing: null pointer");
    if (c->timing_pending) {
        HIPCHK(hipEventSynchronize(c->ev_call[1]));
        float t = 0;
        HIPCHK(hipEventElapsedTime(&t, c->ev_call[0], c->ev_call[1]));
        c->last_ms[0] = t;
        float sum = 0;
        for (int i = 0; i + 1 < c->ev_used; i += 2) {
            float k = 0;
            HIPCHK(hipEventElapsedTime(&k, c->ev[i], c->ev[i + 1]));
            sum += k;
        }
        c->last_ms[1] = sum;
        c->last_launches = c->ev_used / 2;
        c->timing_pending = false;
    }
    for (int i = 0; i < 4; ++i) ms[i] = c->last_ms[i];
    if (n_launch) *n_launch = c->last_launches;
    return LLICTI_OK;
}
