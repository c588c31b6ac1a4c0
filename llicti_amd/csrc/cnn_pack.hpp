// cnn_pack.hpp -- the band CNN's shape constants and the host-side weight pack (canonical arrays -> MFMA-fragment order).  Plain C++17, no HIP
// dependency (g++ compiles it under sanitizers: tests/sanitize_host.sh); band_cnn.hpp, which holds the kernel, includes it.
#pragma once
#include <vector>

#include "host_types.hpp"

// ------------------------------------------------------------------------------------------------ band CNN
// Layer-0 convolutions of band b (LLICTI_nets.py:651-675): source sub-band, kernel size, top / left pad.
struct ConvDef { int src, kh, kw, pt, pl; };
constexpr ConvDef kConvs[3][3] = {
    { { 0, 4, 4, 1, 1 }, { -1, 0, 0, 0, 0 }, { -1, 0, 0, 0, 0 } },
    { { 0, 3, 4, 1, 1 }, { 1, 4, 3, 2, 1 }, { -1, 0, 0, 0, 0 } },
    { { 0, 4, 3, 1, 1 }, { 1, 3, 4, 1, 2 }, { 2, 4, 4, 1, 2 } },
};

constexpr int kHead = 88;          // channels per head (configs/llicti_A.json chs[0])
constexpr int kMT = 6;             // 16-row MFMA tiles per head (88 -> 96, rows >= 88 are zero)
constexpr int kKS1 = 22;           // k-steps of the 88-deep layers (88 / 4)
#ifndef CNN_NT
#define CNN_NT 2
#endif
constexpr int kNT = CNN_NT;                    // pixel tiles (16 positions each) per wavefront
static_assert(CNN_NT == 2, "one wavefront per tile row (two 16-column pixel tiles)");
// Workgroup tile = TH rows x 32 columns of band-grid positions, one wavefront per row.  TH = 16 (16 wavefronts, 4 per SIMD) is the
// throughput form; TH = 4 (4 wavefronts, one per SIMD, a quarter of the work per workgroup) is the LATENCY form for launches whose
// 16-row tiles would leave most of the chip idle (coarse levels, single images): 4x the workgroups, each done in ~a third of the time;
// TH = 8 sits between them (round 4): the form for launches of one to three rounds of 16-row tiles, whose last round is mostly empty.
constexpr int kTileHMax = 16, kTileHMid = 8, kTileHSmall = 4;
constexpr int kTileW = 32;
constexpr int kInCols = kTileW + 4;    // taps reach rows i-2 .. i+2 and columns j-2 .. j+2
constexpr int kInPitch = 48;       // = 16 (mod 32): B-fragment reads that stride by one row stay bank-conflict free
template <int TH> struct CnnGeo {
    static_assert(TH % 4 == 0 && TH >= 4 && TH <= 16, "tile rows: a whole number of 4-row groups");
    static constexpr int kThreads = 64 * TH;
    static constexpr int kInRows = TH + 4;
    static constexpr int kInPlane = kInRows * kInPitch;      // floats of one staged plane: (TH + 4) / 4 row groups of three 64-float pieces
    static constexpr int kPP = kInPlane / 64;                // pieces per plane: 15 (TH = 16), 6 (TH = 4)
};
constexpr int kParamStride = LLICTI_PARAM_STRIDE;
#ifndef CNN_PREFETCH_L0
#define CNN_PREFETCH_L0 1      // software-pipeline the layer-0 fragments one k-step ahead
#endif
#ifndef CNN_FENCE_L1
#define CNN_FENCE_L1 4         // scheduler fence every N k-steps of layer 1 (0 = none)
#endif
#ifndef CNN_PRIO
#define CNN_PRIO 1            // progress-based wave priority (see the tile loop)
#endif
#ifndef CNN_STAGE_SITES
#define CNN_STAGE_SITES 4     // points of the tile at which the wave groups request the next tile's DMA (1, 2 or 4)
#endif
#ifndef CNN_STAGGER
#define CNN_STAGGER 0          // delay waves 4-7 before the first tile (decorrelates the two waves of a SIMD)
#endif
#ifndef CNN_PREFETCH_L1
#define CNN_PREFETCH_L1 8      // layers 1 / 2: weight fragments requested this many k-steps ahead of their MFMAs (0: at the fence window)
#endif
#ifndef CNN_STAGE_FAST
#define CNN_STAGE_FAST 1       // interior tiles (no clamp can fire): per-lane source offsets of a wave's pieces precomputed once per kernel
#endif
#ifndef CNN_REM4X4
#define CNN_REM4X4 1           // layer 0: channels 80..87 of a head on v_mfma_f32_4x4x1 (16 blocks = 2 x 4 channels x 8 x 4 pixels,
#endif                         //   one k per instruction) instead of a sixth, half-empty 16-row tile: no padded MACs in layer 0
static_assert(!CNN_REM4X4 || (CNN_PREFETCH_L0 && CNN_NT == 2), "the 4x4x1 remainder path is written for the prefetching, NT = 2 form");
constexpr int kMT0 = CNN_REM4X4 ? 5 : 6;        // 16-row tiles of LAYER 0


// Per (band, head) weight pack, in MFMA-fragment order so that the LDS image is lane-linear:
//   bias0 [6][4][4]            acc init of tile T, lane group q, reg r  = b0[16T + 4r + q]
//   W0    [kMT0][K0/4][64]     lane l of tile T, k-step t: W0[chan(T, l&15)][4t + (l>>4)]
//   W0r   [K0/4][8][4]         (CNN_REM4X4) channels 80..87: W0r[t][c][kk] = W0[80 + c][4t + kk]
//   bias0r[8]                  (CNN_REM4X4) b0[80 + c]
//   bias1 [6][4][4]
//   W1    [6][22][64]
//   bias2 [4][4]               acc init of lane group q, reg r = b2[4q + r]
//   W2    [22][64]             lane l, k-step t: W2[l&15][4t + (l>>4)]
// chan(T, rho) = 16T + 4(rho&3) + (rho>>2): this row permutation makes the accumulator registers of one
// layer line up, untouched, as the B operand of the next layer's MFMAs in natural channel order
// (C/D layout of v_mfma_f32_16x16x4_f32: col = lane&15, row = 4(lane>>4) + reg).
static constexpr int rem_floats(int K0) { return CNN_REM4X4 ? K0 * 8 + 8 : 0; }
static constexpr int pack_floats(int K0) { return 96 + kMT0 * (K0 / 4) * 64 + rem_floats(K0) + 96 + kMT * kKS1 * 64 + 16 + kKS1 * 64; }

template <int K0>
struct PackOff {
    static constexpr int bias0 = 0;
    static constexpr int w0 = 96;
    static constexpr int w0r = w0 + kMT0 * (K0 / 4) * 64;        // [K0/4][8][4]
    static constexpr int bias0r = w0r + (CNN_REM4X4 ? K0 * 8 : 0);
    static constexpr int bias1 = w0 + kMT0 * (K0 / 4) * 64 + rem_floats(K0);
    static constexpr int w1 = bias1 + 96;
    static constexpr int bias2 = w1 + kMT * kKS1 * 64;
    static constexpr int w2 = bias2 + 16;
    static constexpr int total = w2 + kKS1 * 64;
};
static constexpr int cnn_lds_bytes(int band, int TH = kTileHMax)
{
    const int K0 = band == 0 ? 48 : band == 1 ? 72 : 120;
    return (pack_floats(K0) + 2 * 3 * (band + 1) * (TH + 4) * kInPitch) * 4;     // weights + double-buffered input tile
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
        float *bias0 = p, *W0 = p + 96, *W0r = W0 + kMT0 * NK0 * 64, *bias0r = W0r + (CNN_REM4X4 ? K0 * 8 : 0);
        float *bias1 = W0 + kMT0 * NK0 * 64 + rem_floats(K0), *W1 = bias1 + 96;
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
                if (T < kMT0)
                    for (int t = 0; t < NK0; ++t)
                        W0[(T * NK0 + t) * 64 + l] = (cl < kHead) ? w0[(size_t)(hd * kHead + cl) * K0 + 4 * t + q] : 0.0f;
                for (int t = 0; t < kKS1; ++t)
                    W1[(T * kKS1 + t) * 64 + l] = (cl < kHead) ? w1[(size_t)(hd * kHead + cl) * kHead + 4 * t + q] : 0.0f;
            }
        if (CNN_REM4X4) {
            for (int t = 0; t < NK0; ++t)
                for (int c = 0; c < 8; ++c)
                    for (int kk = 0; kk < 4; ++kk) W0r[(t * 8 + c) * 4 + kk] = w0[(size_t)(hd * kHead + 80 + c) * K0 + 4 * t + kk];
            for (int c = 0; c < 8; ++c) bias0r[c] = b0[hd * kHead + 80 + c];
        }
        for (int q = 0; q < 4; ++q)
            for (int r = 0; r < 4; ++r) bias2[q * 4 + r] = (4 * q + r < 15) ? b2[hd * 15 + 4 * q + r] : 0.0f;
        for (int l = 0; l < 64; ++l) {
            const int o = l & 15, q = l >> 4;
            for (int t = 0; t < kKS1; ++t) W2[t * 64 + l] = (o < 15) ? w2[(size_t)(hd * 15 + o) * kHead + 4 * t + q] : 0.0f;
        }
    }
}
