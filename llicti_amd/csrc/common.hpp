// common.hpp -- errors, geometry and small device helpers shared by every kernel file.
// Part of the single translation unit llicti_hip.hip (included in order; not a stand-alone header).
#pragma once
#include "host_types.hpp"

#define HIPCHK(x)                                                                                    \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) return fail(LLICTI_EHIP, "%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern "C" const char *llicti_last_error(void) { return g_err.c_str(); }
extern "C" const char *llicti_version(void) { return "llicti_hip 0.7 (gfx950, numerics spec v1, rANS containers: v3 for 64 / 128 lanes, v4 for xwide streams -- tail arena with spill, zero-start chain, stream count in the header's pad field, picked per image by the encoder in the auto mode; batches of mixed sizes)"; }

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


// source sub-bands in lazyDWT cat order x00, x11, x01, x10 (LLICTI_nets.py:241); band b predicts source b+1
// (row, column) phase of source s: (0,0), (1,1), (0,1), (1,0) -- computed, not looked up: a table load inside
// the CNN's staging loop would put an s_waitcnt vmcnt(0) between consecutive LDS-DMA pieces
__device__ __forceinline__ int src_oi(int s) { return s & 1; }
__device__ __forceinline__ int src_oj(int s) { return ((s + 1) >> 1) & 1; }


// Device-side failures are latched twice: status[0] for the whole call (llicti_check_status) and status[16 + b] for image b
// (llicti_image_status), so that a caller can drop just the bad image of a batch.
__device__ __forceinline__ void flag_image(int32_t *status, int b, int code)
{
    atomicExch(&status[0], code);
    atomicExch(&status[16 + b], code);
}

// Workgroup barrier that waits for this wavefront's LDS (and scalar) traffic only.  __syncthreads() is a workgroup-scope fence + s_barrier, and
// the fence drains vmcnt too: every global load in flight (the coders' prefetches of the NEXT steps' operands) and every global store
// just issued (decoded pixels, flushed stream words) would have to complete before the barrier -- a memory round trip per coder step.
// Where only LDS words cross between the wavefronts, this is the barrier to use.  ("memory": the compiler keeps LDS accesses on their side.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// wave-wide vote as a 64-bit lane mask, straight from the comparison (HIP's __ballot(int) first materialises the
// predicate as 0/1 in a VGPR and compares it again: two extra vector operations per vote)
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

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

// Block-cooperative byte copy of n bytes with arbitrary (mutually different) alignment of source and destination:
// bytes up to the destination's first 16-byte boundary, then 16-byte stores fed by unaligned 16-byte loads
// (gfx950 global memory handles unaligned dwordx4 accesses in hardware), then the tail bytes.
struct __attribute__((packed, aligned(1))) u128_unaligned { uint32_t v[4]; };
__device__ __forceinline__ void block_copy_bytes(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, int n)
{
    const int head = min(n, (int)((16u - ((uint32_t)(uintptr_t)dst & 15u)) & 15u));
    for (int t = threadIdx.x; t < head; t += blockDim.x) dst[t] = src[t];
    const int nvec = (n - head) >> 4;
    uint4 *d4 = reinterpret_cast<uint4 *>(dst + head);
    const u128_unaligned *s4 = reinterpret_cast<const u128_unaligned *>(src + head);
    for (int t = threadIdx.x; t < nvec; t += blockDim.x) {
        const u128_unaligned v = s4[t];
        d4[t] = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
    }
    for (int t = head + (nvec << 4) + threadIdx.x; t < n; t += blockDim.x) dst[t] = src[t];
}
