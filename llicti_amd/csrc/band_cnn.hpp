// band_cnn.hpp -- interpolator CNN: three chained fp32-MFMA GEMMs per pixel tile (K4-K5).
// Part of the single translation unit llicti_hip.hip (included in order; not a stand-alone header).
#pragma once
#include "cnn_pack.hpp"

// ------------------------------------------------------------------------------------------------ band CNN (kernel)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// One MFMA k-step consumes 4 consecutive k of the canonical K order (llicti_amd/weights.py): the kernel's
// length-4 axis.  Lane (q = lane>>4, px = lane&15) therefore reads the staged input tile at
// U + q*S + pixel offset with U, S compile-time constants of the k-step.
struct KStep { int U, S; };
struct KTab { KStep s[30]; int n; };
constexpr KTab make_ktab(int band, int kInPlane)
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
template <int BAND, int TH> inline constexpr KTab kKTab = make_ktab(BAND, CnnGeo<TH>::kInPlane);

// TIMING-ONLY experiment switches (what would an all-heads-per-workgroup, weight-stationary form buy?  results are WRONG with any of
// them set; never set in the product build -- tools/cnn_ws_experiment.sh builds the variants, profiles/r6/tried_cnn_weight_stationary.json has the numbers):
//   CNN_EXP_STAGE_EVERY = n   the next tile's input is staged for every n-th tile of a workgroup only (n = 4: what ONE staging per tile for all four
//                             heads would cost per head -- today the four head-workgroups of a tile each stage it);
//   CNN_EXP_NO_WFRAG = 1      no weight-fragment LDS reads inside the tile loop (the fragments a wavefront starts with are reused: what weights held
//                             in registers for the whole kernel would save).
#ifndef CNN_EXP_STAGE_EVERY
#define CNN_EXP_STAGE_EVERY 1
#endif
#ifndef CNN_EXP_NO_WFRAG
#define CNN_EXP_NO_WFRAG 0
#endif

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ float relu(float x) { return (x > 0.0f) ? x : 0.0f; }
__device__ __forceinline__ f32x4 relu4(f32x4 v) { v[0] = relu(v[0]); v[1] = relu(v[1]); v[2] = relu(v[2]); v[3] = relu(v[3]); return v; }
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// 16 blocks of D[4x4] = A[4x1] * B[1x4] + C: A[i] in lane 4b + i, B[j] in lane 4b + j, D[i][j] in lane 4b + j, register i; one
// fmaf per element (tools/hipchecks/check_mfma4x4.hip)
#define MFMA1(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)

// priority to switch to when the wave's issued-MFMA count passes a quarter mark of the tile inside (before, after]; -1: none
constexpr int prio_step(int before, int after, int total)
{
    for (int i = 1; i <= 3; ++i) if (before < total * i / 4 && after >= total * i / 4) return 3 - i;
    return -1;
}

// RAGGED = false: B images of ONE size (tile -> image, row, column by division; the image's arrays at b x their size) -- the form every
// call with equal sizes runs.  RAGGED = true (llicti_encode_images_v / _decode_images_v with mixed sizes): the launch walks a TILE LIST --
// tiles[t] = (image, tile row << 16 | tile column), image-major -- and takes each image's geometry and placement from gv[image]; the same
// fmaf chain per position, so the outputs of an image do not depend on what else is in the batch.
template <int BAND, int TH = kTileHMax, bool RAGGED = false>
__global__ __launch_bounds__(CnnGeo<TH>::kThreads) void band_params_kernel(const float *__restrict__ fplanes, Geom g,
                                                                  const float *__restrict__ wpack,
                                                                  float *__restrict__ params, int tiles_x, int tiles_y, int n_tiles,
                                                                  const Geom *__restrict__ gv, const TileRef *__restrict__ tiles)
{
    using GEO = CnnGeo<TH>;
    constexpr int kCnnThreads = GEO::kThreads, kTileH = TH, kInRows = GEO::kInRows, kInPlane = GEO::kInPlane, kPP = GEO::kPP;
    constexpr int K0 = (BAND == 0) ? 48 : (BAND == 1) ? 72 : 120;
    constexpr int NK0 = K0 / 4;
    constexpr int NPL = 3 * (BAND + 1);          // staged input planes: (x00 | x11 | x01) x (Y, Co, Cg)
    constexpr int kMfmaL0 = kMT * NK0 * kNT, kMfmaT12 = (kKS1 + 4) * kNT, kMfmaTile = kMfmaL0 + kMT * kMfmaT12;   // MFMAs of a wave per tile (approx.)
    using PO = PackOff<K0>;
    static_assert(kKTab<BAND, TH>.n == NK0, "k-step table");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *lds_in = lds + PO::total;

    const int head = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    if ((int)blockIdx.x >= n_tiles) return;      // (the host never launches such a workgroup; one must not end with LDS-DMA in flight)
    {   // Stage this head's pack (lane-linear image: a straight copy) by LDS-DMA, 16 bytes per lane: a wavefront requests its 1 KB pieces
        // back to back and nobody waits before the first tile's barrier (in front of which every wave drains its vmcnt explicitly).  A copy through registers is one memory
        // round trip per piece and thread: 20 of them for band 2's 82 KB in a 4-row workgroup -- a quarter of the time of a launch that has
        // one tile per workgroup (coarse levels, single images), and what made 4-row tiles of band 2 no faster than 8-row ones.
        const char *src = reinterpret_cast<const char *>(wpack + (long)head * PO::total);
        constexpr int kFull = PO::total / 256;                                   // whole 1 KB pieces
        for (int p = __builtin_amdgcn_readfirstlane(wave); p < kFull; p += kCnnThreads / 64)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (long)p * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(lds + p * 256), 16, 0, 0);
        constexpr int kRem4 = (PO::total - kFull * 256) / 4;                     // the rest: fewer than 64 float4
        static_assert(PO::total % 4 == 0 && kRem4 < 64, "pack size");
        if ((int)threadIdx.x < kRem4)
            reinterpret_cast<float4 *>(lds)[kFull * 64 + threadIdx.x] = reinterpret_cast<const float4 *>(src)[kFull * 64 + threadIdx.x];
    }
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
    constexpr int kPieces = NPL * kPP;                                               // TH = 16: 45 / 90 / 135
    constexpr int kMyPieces = (kPieces + kCnnThreads / 64 - 1) / (kCnnThreads / 64);  // pieces a wave stages per tile: 3 / 6 / 9
#if CNN_STAGE_FAST
    // For a tile whose whole halo lies inside the band grid (and off the odd edge) no clamp fires, and a lane's source
    // address is  image base + tile origin (both wave-uniform) + a constant of (wave, piece, lane): those constants are
    // computed ONCE per kernel (byte offsets, < 2^32), so that staging an interior tile costs no address arithmetic per
    // piece -- the ~20 vector operations per piece of the general path compete with the MFMAs for the issue port
    // (deletion experiments: staging is ~6 % of the kernel).  Border tiles take the general path.
    // lane-dependent part: only the piece's phase t matters (row t or t + 1 of its 4-row group, column within the row)
    uint32_t lane_t[3];
    if constexpr (!RAGGED) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int thr = 48 - 16 * t;
            const bool up = lane >= thr;
            const int cidx = min(up ? lane - thr : lane + 16 * t, kInCols - 1);
            lane_t[t] = (uint32_t)((((long)(2 * (t + (up ? 1 : 0))) * g.W + 2 * cidx) << g.lvl) * 4);
        }
    }
#endif
    // The mixed-size form takes tile -> (image, tile row, tile column) from the list and the image's geometry from the table (wave-uniform:
    // scalar loads), where it is needed -- in the staging and in the epilogue -- and does not keep it across the tile (the kernel is at its
    // SGPR budget).  The equal-size form is, instruction for instruction, what it was before the list existed.
    // (always_inline: left to the inliner, band 2's 4-row form got a real call -- s_swappc_b64, a 176-byte stack frame -- at each of its five sites)
    auto stage = [&](int tile, float *dst) __attribute__((always_inline)) {
        int img, ty, tx;
        Geom gl;                                                // (mixed sizes only)
        if constexpr (RAGGED) {
            const TileRef t = tiles[tile];
            img = __builtin_amdgcn_readfirstlane(t.img);
            const int yx = __builtin_amdgcn_readfirstlane(t.yx);
            ty = yx >> 16; tx = yx & 0xFFFF;
            gl = gv[img];
        } else {
            img = tile / (tiles_x * tiles_y);
            const int trem = tile - img * (tiles_x * tiles_y);
            ty = trem / tiles_x; tx = trem - ty * tiles_x;
        }
        const Geom &g_k = g;
        const Geom &g = RAGGED ? gl : g_k;                      // (shadows the kernel's: this image's)
        const int i0 = ty * kTileH - 2, j0 = tx * kTileW - 2;
        const float *base = RAGGED ? fplanes + g.pix_off : fplanes + (long)img * 3 * g.plane;
#if CNN_STAGE_FAST
        // mixed sizes: the row pitch is the image's, so a piece's per-lane offset is computed where the piece is requested, from its (scalar) phase --
        // ~8 vector operations per piece, ~70 of a tile's ~10^5 cycles -- instead of being kept for the three phases across the staging: the 16-row
        // form of band 2 sits at the 128-VGPR cap of a 1024-thread workgroup and spilled eight registers over them
        int ln = lane;
        asm volatile("" : "+v"(ln));                            // (opaque per staged tile: left visible, the lane's part of the offsets is hoisted out of the tile loop -- into spilled registers)
        auto lane_off = [&](int t) __attribute__((always_inline)) -> uint32_t {
            const int thr = 48 - 16 * t;
            const bool up = ln >= thr;
            const int cidx = min(up ? ln - thr : ln + 16 * t, kInCols - 1);
            return (uint32_t)((((long)(2 * (t + (up ? 1 : 0))) * g.W + 2 * cidx) << g.lvl) * 4);
        };
        // rows i0 .. i0 + kInRows - 1 and columns j0 .. j0 + kInCols - 1 of the band grid, all strictly inside it and below
        // the last row / column (where lazyDWT's odd-edge pad could apply)
        if (i0 >= 0 && i0 + kInRows - 1 <= g.h - 2 && j0 >= 0 && j0 + kInCols - 1 <= g.w - 2) {
            const char *origin = reinterpret_cast<const char *>(base) + (((long)(2 * i0) * g.W + 2 * j0) << g.lvl) * 4;
#pragma unroll
            for (int k = 0; k < kMyPieces; ++k) {
                const int u = wave_u + k * (kCnnThreads / 64);                           // wave-uniform: the rest of the address is scalar work
                if (u < kPieces) {
                    const int pl = u / kPP, v = u - kPP * pl, q4 = v / 3, t = v - 3 * q4;
                    const int src = pl / 3, ci = pl - 3 * src;
                    const long uoff = ((long)ci * g.plane + (((long)(8 * q4 + src_oi(src)) * g.W + src_oj(src)) << g.lvl)) * 4;
                    uint32_t lo;
                    if constexpr (RAGGED) lo = lane_off(t);
                    else lo = (t == 0) ? lane_t[0] : (t == 1) ? lane_t[1] : lane_t[2];
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(origin + uoff + lo),
                                                     (__attribute__((address_space(3))) void *)(dst + u * 64), 4, 0, 0);
                }
            }
            return;
        }
#endif
        for (int u = wave_u; u < kPieces; u += kCnnThreads / 64) {
            const int pl = u / kPP, v = u - kPP * pl, q4 = v / 3, t = v - 3 * q4;       // wave-uniform
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
    static_assert(kInPitch == 48 && kInRows % 4 == 0 && kPP * 64 == kInPlane && kPP == 3 * (kInRows / 4), "piece decomposition assumes pitch 48 and whole 4-row groups");
    static_assert(kInPlane % 64 == 0, "tile plane must be a whole number of 64-float pieces");

    const int stage_site = (__builtin_amdgcn_readfirstlane(wave) / (kCnnThreads >= 256 ? kCnnThreads / 256 : 1)) % CNN_STAGE_SITES;
    int cur = 0;
    if ((int)blockIdx.x < n_tiles) stage(blockIdx.x, lds_in);
#if CNN_STAGGER
    if (__builtin_amdgcn_readfirstlane(wave) >= 4) __builtin_amdgcn_s_sleep(CNN_STAGGER);
#endif
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int img, ty, tx;
        if constexpr (RAGGED) {
            const TileRef t = tiles[tile];
            img = __builtin_amdgcn_readfirstlane(t.img);
            const int yx = __builtin_amdgcn_readfirstlane(t.yx);
            ty = yx >> 16; tx = yx & 0xFFFF;
        } else {
            img = tile / (tiles_x * tiles_y);
            const int trem = tile - img * (tiles_x * tiles_y);
            ty = trem / tiles_x; tx = trem - ty * tiles_x;
        }
        const int i0 = ty * kTileH, j0 = tx * kTileW;
        float *lds_cur = lds_in + cur * (NPL * kInPlane);

        // this tile's pieces have landed (each wave drains its own DMA, then the barrier), and every wave has
        // finished reading the other buffer (previous tile) -- which the next tile's DMA may now overwrite.  The drain is explicit: LDS-DMA
        // (the tile's pieces, and before the first tile the head's weight pack) is tracked by vmcnt, and a wave must have seen its own
        // requests complete before it tells the others so -- __syncthreads()'s fence happens to wait for vmcnt(0) on this toolchain, the
        // memory model does not promise it (common.hpp's lds_barrier() is exactly a barrier that does not).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // The next tile's DMA (address arithmetic + issue: pure VALU / VMEM work) is requested at four
        // different points of the tile, one per wave group: a SIMD hosts one wave of each group, so while
        // one of its waves stages, the other three keep the matrix pipe busy.  (All 16 waves staging right
        // after the barrier left the pipe idle for ~9 % of the tile.)
        const bool more = tile + (int)gridDim.x < n_tiles && (CNN_EXP_STAGE_EVERY == 1 || ((tile / (int)gridDim.x) % CNN_EXP_STAGE_EVERY) == CNN_EXP_STAGE_EVERY - 1);
        auto stage_next = [&](int site) __attribute__((always_inline)) {
            if (more && stage_site == site % CNN_STAGE_SITES) stage(tile + gridDim.x, lds_in + (cur ^ 1) * (NPL * kInPlane));
        };
        // Wave priority falls as the wave advances through its tile (3, 2, 1, 0 by quarter of the MFMA work): the
        // SIMD's arbiter otherwise serves the OLDEST wave first, which then finishes its tile ~24 % of a tile time
        // before the barrier opens while the youngest runs the last stretch alone, with nobody to fill its LDS /
        // VALU bubbles (in-kernel stamps: barrier wait 23.5k cycles for wave 0, 6.5k for wave 8, tile 99k).
        // A wave that is behind now outranks one that is ahead, so the four waves of a SIMD reach the barrier together.
        if constexpr (CNN_PRIO) __builtin_amdgcn_s_setprio(3);
        stage_next(0);

        // ---- layer 0: [96 x K0] x [K0 x 64 pixels]; bias preloaded into the accumulators
        f32x4 a0[kMT][kNT];
#pragma unroll
        for (int T = 0; T < kMT0; ++T) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(lds + PO::bias0 + (T * 4 + q) * 4);
#pragma unroll
            for (int n = 0; n < kNT; ++n) a0[T][n] = bv;
        }
#if CNN_REM4X4
#pragma unroll
        for (int n = 0; n < kNT; ++n) a0[5][n] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };     // [0], [1] are written after layer 0; [2], [3] never read
#endif
#if CNN_PREFETCH_L0
        {
            float a_c[kMT0], b_c[kNT];
#if CNN_REM4X4
            // channels 80..87 of the head: block b = lane >> 2 = (cg, pg): channels 80 + 4 cg + i (A, lane & 3 = i) x pixels
            // 4 pg + j of the wave's 32-pixel row (B, lane & 3 = j); accumulator register i of lane (b, j) = channel
            // 80 + 4 cg + i at pixel 4 pg + j.  One k per instruction, in k order: the same fmaf chain as the 16x16x4 tiles.
            const int rsub = lane & 3, rcg = lane >> 5, rpix = 4 * ((lane >> 2) & 7) + rsub;
            const float *rb_base = lds_cur + ((wave * kNT) >> 1) * kInPitch + rpix;
            const float *ra_base = lds + PO::w0r + (4 * rcg + rsub) * 4;
            f32x4 dR = *reinterpret_cast<const f32x4 *>(lds + PO::bias0r + 4 * rcg);
            f32x4 ar_c;
            float br_c[4];
#endif
            {
                constexpr int U = kKTab<BAND, TH>.s[0].U, S = kKTab<BAND, TH>.s[0].S;
                const float *bp = lds_cur + U + pix0 + (S == 1 ? q : q_row);
#pragma unroll
                for (int n = 0; n < kNT; ++n) b_c[n] = bp[(n >> 1) * kInPitch + 16 * (n & 1)];
#pragma unroll
                for (int T = 0; T < kMT0; ++T) a_c[T] = lds[PO::w0 + (T * NK0 + 0) * 64 + lane];
#if CNN_REM4X4
                ar_c = *reinterpret_cast<const f32x4 *>(ra_base);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) br_c[kk] = rb_base[U + (S == 1 ? kk : kk * kInPitch)];
#endif
            }
            static_for<NK0>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                if constexpr (CNN_PRIO && prio_step(kMfmaL0 * t / NK0, kMfmaL0 * (t + 1) / NK0, kMfmaTile) >= 0)
                    __builtin_amdgcn_s_setprio(prio_step(kMfmaL0 * t / NK0, kMfmaL0 * (t + 1) / NK0, kMfmaTile));
                float a_n[kMT0], b_n[kNT];
#if CNN_REM4X4
                f32x4 ar_n;
                float br_n[4];
#endif
                if constexpr (t + 1 < NK0) {       // next k-step's fragments are in flight while this one's MFMAs run
                    constexpr int U = kKTab<BAND, TH>.s[t + 1].U, S = kKTab<BAND, TH>.s[t + 1].S;
                    const float *bp = lds_cur + U + pix0 + (S == 1 ? q : q_row);
#pragma unroll
                    for (int n = 0; n < kNT; ++n) b_n[n] = bp[(n >> 1) * kInPitch + 16 * (n & 1)];
#pragma unroll
                    for (int T = 0; T < kMT0; ++T) a_n[T] = CNN_EXP_NO_WFRAG ? a_c[T] : lds[PO::w0 + (T * NK0 + t + 1) * 64 + lane];
#if CNN_REM4X4
                    ar_n = CNN_EXP_NO_WFRAG ? ar_c : *reinterpret_cast<const f32x4 *>(ra_base + (t + 1) * 32);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) br_n[kk] = rb_base[U + (S == 1 ? kk : kk * kInPitch)];
#endif
                }
#if CNN_REM4X4
                // the four 4x4x1 steps of this k-step form ONE dependent chain (k order is the spec): they are spread between
                // the ten independent 16x16x4 MFMAs so that none waits for its predecessor
                // (scheduler fences pin the order: left alone, the machine scheduler clusters the 4x4x1s)
                a0[0][0] = MFMA4(a_c[0], b_c[0], a0[0][0]); a0[0][1] = MFMA4(a_c[0], b_c[1], a0[0][1]);
                dR = MFMA1(ar_c[0], br_c[0], dR);
                __builtin_amdgcn_sched_barrier(0);
                a0[1][0] = MFMA4(a_c[1], b_c[0], a0[1][0]); a0[1][1] = MFMA4(a_c[1], b_c[1], a0[1][1]);
                a0[2][0] = MFMA4(a_c[2], b_c[0], a0[2][0]);
                dR = MFMA1(ar_c[1], br_c[1], dR);
                __builtin_amdgcn_sched_barrier(0);
                a0[2][1] = MFMA4(a_c[2], b_c[1], a0[2][1]);
                a0[3][0] = MFMA4(a_c[3], b_c[0], a0[3][0]); a0[3][1] = MFMA4(a_c[3], b_c[1], a0[3][1]);
                dR = MFMA1(ar_c[2], br_c[2], dR);
                __builtin_amdgcn_sched_barrier(0);
                a0[4][0] = MFMA4(a_c[4], b_c[0], a0[4][0]); a0[4][1] = MFMA4(a_c[4], b_c[1], a0[4][1]);
                dR = MFMA1(ar_c[3], br_c[3], dR);
#else
#pragma unroll
                for (int T = 0; T < kMT0; ++T)
#pragma unroll
                    for (int n = 0; n < kNT; ++n) a0[T][n] = MFMA4(a_c[T], b_c[n], a0[T][n]);
#endif
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (t + 1 < NK0) {
#pragma unroll
                    for (int T = 0; T < kMT0; ++T) a_c[T] = a_n[T];
#pragma unroll
                    for (int n = 0; n < kNT; ++n) b_c[n] = b_n[n];
#if CNN_REM4X4
                    ar_c = ar_n;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) br_c[kk] = br_n[kk];
#endif
                }
            });
#if CNN_REM4X4
            // hand the 8 channels to layer 1 in ITS operand layout: k-step 20 + r of layer 1 wants channel 80 + 4 r + q of pixel
            // 16 n + px in lane (q, px) = register q of lane 32 r + 16 n + px here.  a0[5][n][0..1] are exactly those operands
            // (a0[5][n][2..3] would be k-steps 22, 23: not used, 88 = 22 x 4).
            dR = relu4(dR);
#pragma unroll
            for (int n = 0; n < kNT; ++n)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int src = 4 * (32 * r + 16 * n + px);
                    const float t0 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(dR[0])));
                    const float t1 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(dR[1])));
                    const float t2 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(dR[2])));
                    const float t3 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(dR[3])));
                    a0[5][n][r] = (q == 0) ? t0 : (q == 1) ? t1 : (q == 2) ? t2 : t3;
                }
#endif
        }
#else
        static_for<NK0>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int U = kKTab<BAND, TH>.s[t].U, S = kKTab<BAND, TH>.s[t].S;
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
        for (int T = 0; T < kMT0; ++T)
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
#if CNN_PREFETCH_L1 > 0
        // Layer 1's weight fragments are one linear stream of kMT * kKS1 rows of 64 floats: a ring of D registers keeps the
        // next D k-steps' fragments in flight (without it the compiler requests a fence window's four fragments and waits
        // for them on the spot: one exposed LDS latency per 8 MFMAs, and the tile's slowest wave spent 48k cycles on 39k
        // cycles of MFMA work in these two layers)
        constexpr int D1 = CNN_PREFETCH_L1;
        float ring1[D1];
#pragma unroll
        for (int i = 0; i < D1; ++i) ring1[i] = lds[PO::w1 + i * 64 + lane];
#endif
        static_for<kMT>([&](auto Tc) {
            constexpr int T = decltype(Tc)::value;
            if constexpr (CNN_PRIO && prio_step(kMfmaL0 + T * kMfmaT12, kMfmaL0 + (T + 1) * kMfmaT12, kMfmaTile) >= 0)
                __builtin_amdgcn_s_setprio(prio_step(kMfmaL0 + T * kMfmaT12, kMfmaL0 + (T + 1) * kMfmaT12, kMfmaTile));
            if constexpr (T == 2 && CNN_STAGE_SITES > 2) stage_next(2);
            if constexpr (T == 4 && CNN_STAGE_SITES > 2) stage_next(3);
            f32x4 a1[kNT];
            {
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(lds + PO::bias1 + (T * 4 + q) * 4);
#pragma unroll
                for (int n = 0; n < kNT; ++n) a1[n] = bv;
            }
#if CNN_PREFETCH_L1 > 0
            float a2w[4] = { 0.0f, 0.0f, 0.0f, 0.0f };      // this tile's layer-2 fragments: requested now, used 22 k-steps later
            static_for<4>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                if constexpr (4 * T + r < kKS1) a2w[r] = CNN_EXP_NO_WFRAG ? ring1[r % D1] : lds[PO::w2 + (4 * T + r) * 64 + lane];
            });
#endif
            static_for<kKS1>([&](auto ttc) {
                constexpr int tt = decltype(ttc)::value;
#if CNN_PREFETCH_L1 > 0
                constexpr int i = T * kKS1 + tt;
                const float a = ring1[i % D1];
                if constexpr (i + D1 < kMT * kKS1 && !CNN_EXP_NO_WFRAG) ring1[i % D1] = lds[PO::w1 + (i + D1) * 64 + lane];
#else
                const float a = lds[PO::w1 + (T * kKS1 + tt) * 64 + lane];
#endif
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
#if CNN_PREFETCH_L1 > 0
                    const float a = a2w[r];
#else
                    const float a = lds[PO::w2 + (4 * T + r) * 64 + lane];
#endif
#pragma unroll
                    for (int n = 0; n < kNT; ++n) a2[n] = MFMA4(a, a1[n][r], a2[n]);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });

        // D row 4q + r = output 4q + r of this head -> channel plane head * 16 + 4q + r of params[img][64][h * w] (numerics.hpp: ParRow).
        // A store instruction writes 16 consecutive positions (64 bytes) of four planes; plane 15 of a head does not exist (15 outputs).
        if constexpr (RAGGED) {
            const Geom gi = gv[img];                            // this image's
            const long npos = (long)gi.h * gi.w;
#pragma unroll
            for (int n = 0; n < kNT; ++n) {
                const int i = i0 + ((wave * kNT) >> 1) + (n >> 1), j = j0 + 16 * (n & 1) + px;
                if (i < gi.h && j < gi.w) {
                    float *dst = params + gi.par_off + (long)(head * 16 + 4 * q) * npos + (long)i * gi.w + j;
                    dst[0] = a2[n][0];
                    dst[npos] = a2[n][1];
                    dst[2 * npos] = a2[n][2];
                    if (q < 3) dst[3 * npos] = a2[n][3];
                }
            }
        } else {
            const long npos = (long)g.h * g.w;
#pragma unroll
            for (int n = 0; n < kNT; ++n) {
                const int i = i0 + ((wave * kNT) >> 1) + (n >> 1), j = j0 + 16 * (n & 1) + px;
                if (i < g.h && j < g.w) {
                    float *dst = params + ((long)img * kParamStride + head * 16 + 4 * q) * npos + (long)i * g.w + j;
                    dst[0] = a2[n][0];
                    dst[npos] = a2[n][1];
                    dst[2 * npos] = a2[n][2];
                    if (q < 3) dst[3 * npos] = a2[n][3];
                }
            }
        }
        cur ^= 1;
    }
}

