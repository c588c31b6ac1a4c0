// llicti_hip.hip -- gfx950 (MI355X / CDNA4) kernels and the C-ABI of include/llicti_hip.h.
// One translation unit: the kernels live in the *.hpp files included below (common, lift, likelihood, band_cnn,
// cdf, ac_coder, rans_coder, container); this file holds the context, the workspace plan and the entry points.
//
// Kernels (reference call sites in include/llicti_hip.h):
//   lift_kernel / unlift_kernel       integer YCoCg-R lift, min/max, float planes          (HBM bound)
//   band_params_kernel<BAND>          interpolator CNN: 3 chained fp32-MFMA GEMMs per pixel tile,
//                                     weights of one 88-channel head resident in LDS        (MFMA bound)
//   cdf_pairs_kernel                  encoder: the two table entries per symbol (10 erfc)   (VALU)
//   cdf_table_kernel                  decoder: full Lp-entry uint16 rows                    (HBM / VALU bound)
//   ac_encode_*_kernel                torchac-algorithm range encoder, one lane per stream  (latency bound)
//   ac_decode_kernel                  matching decoder, one wavefront per stream            (latency bound)
//   rans_encode_kernel<Q>             rANS v3 encoder: 64 Q lanes per stream, one wavefront per 64 (latency bound)
//   rans_decode_stage_*_kernel        rANS v3 stage decoders, table-free: four / two / ONE lane per symbol for streams of 64 / 128 / 256
//                                     lanes (the timed mode: rans_decode_stage_lane_kernel<4>)    (VALU issue bound)
//   rans_tail_kernel<Q>               the serial tail coder behind the lanes' initial states      (latency bound)
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
#include <memory>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/llicti_hip.h"
#include "numerics.hpp"

using namespace llicti;

#include "common.hpp"
#include "lift.hpp"
#include "likelihood.hpp"
#include "band_cnn.hpp"
#include "cdf.hpp"
#include "ac_coder.hpp"
#include "rans_coder.hpp"
#include "container.hpp"
#include "host_plan.hpp"

// ------------------------------------------------------------------------------------------------ context
constexpr int kMaxSub = 3;
constexpr int kMaxPlans = 32;         // plans the context keeps (least recently used goes first)
// Device + pinned host memory of one plan's tables.  Blocks are POOLED and never freed before llicti_destroy: a plan that leaves the cache
// hands its block back, the next new plan takes any block whose last user has finished (hipEventQuery on `done`, recorded behind every call
// that uses the block) -- so that a data set of many image sizes (the reference's own test set has 119 among 500 images) never makes the
// library synchronise the device or call hipFree (which does) once it is warm.
struct PlanBlock {
    uint8_t *dev = nullptr, *host = nullptr;
    size_t cap = 0;
    hipEvent_t done = nullptr;        // behind the last call that used the block
    hipEvent_t uploaded = nullptr;    // behind the table upload ...
    hipStream_t up_stream = nullptr;  // ... on this stream: a call on another stream waits for it
    bool used = false;                // `done` has been recorded at least once
    hipStream_t done_stream = nullptr;      // ... last on this stream: a call on ANOTHER stream waits for `done` before it re-records it, so that the one
                                            // event always lies behind every user of the block (a plan used on stream A, then on B, then evicted
                                            // was recycled as soon as B's record completed, while A's kernels could still be reading the tables)
};
struct PlanDev {
    Plan p;
    PlanBlock blk;
    uint64_t last_use = 0;
    template <class T> const T *dev(size_t off) const { return reinterpret_cast<const T *>(blk.dev + off); }
};

// profiling spans (llicti_set_profiling): what a pair of events brackets
enum ProfCat { PROF_CNN = 0, PROF_RANS_STAGE, PROF_RANS_TAIL, PROF_PAIRS, PROF_RANS_ENC, PROF_AC, PROF_MISC, PROF_NCAT };
static_assert(PROF_NCAT == LLICTI_NPROF, "include/llicti_hip.h: LLICTI_NPROF");

struct llicti_ctx {
    int device = 0;
    float *d_pack[3] = { nullptr, nullptr, nullptr };
    bool have[3] = { false, false, false };
    std::map<std::vector<long>, struct PlanDev *> plans;   // Plan::key -> plan + its device tables (at most kMaxPlans, least recently used out first)
    std::vector<PlanBlock> pool;      // table blocks of plans that left the cache, for the next new plan
    uint64_t use_clock = 0;
    int force_ragged = 0;             // llicti_set_tuning("force_ragged"): equal-size batches take the mixed-size code path too (tests)
    // what the whole-batch calls did to the host / device since llicti_create (llicti_get_counter): a steady stream of calls over many image
    // sizes must add nothing to the first two
    long n_device_sync = 0, n_device_alloc = 0, n_plan_build = 0, n_plan_hit = 0, n_block_wait = 0;
    hipStream_t sub[kMaxSub] = { nullptr, nullptr, nullptr };    // internal streams of the AC decode pipeline ([0] unused: the caller's)
    hipEvent_t ev_ac[2][16] = {};      // AC decode pipeline: chunk c of Y / Co done
    hipEvent_t ev_ac_band = nullptr, ev_ac_end[2] = { nullptr, nullptr };
    int ac_anchor_min_batch = kAcAnchorBatch;
    int32_t *d_img_status = nullptr;  // per-image status words of the last llicti_decode_images call, latched from its workspace
    int img_status_cap = 0, img_status_n = 0;   // (context-owned: the caller may free or reuse the workspace before llicti_image_status)
    int32_t *d_status = nullptr;      // small persistent status word (for the kernel-level entry points)
    int32_t *d_lift_part = nullptr;   // min/max partials of llicti_lift_u8 (1 MB; calls on one context are not concurrent)
    float2 *d_phi_lut = nullptr;      // normal CDF on [-6, 6] as (value, difference to the next) pairs: the lane decoder's hint table (rans_coder.hpp)
    int n_cu = 256;                   // compute units of the device (grid sizing of the persistent kernels)
    int cnn_tile_rows = 0;            // llicti_set_tuning("cnn_tile_rows"): 0 = choose per launch, 16 / 4 = force (tests, A/B)
    int enc_side_levels = 0;          // llicti_set_tuning("enc_side_levels"): 1 = encoder levels 4..1 on a side stream next to level 0, 0 = one queue (default)
    hipEvent_t ev_enc[2] = { nullptr, nullptr };
    bool profiling = false;
    std::vector<hipEvent_t> ev;       // event pool of the profiling spans
    struct Span { hipEvent_t e0, e1; int cat, tag; };
    std::vector<Span> spans;          // spans of the current call: (start, end) events, category, level of a PROF_CNN span (-1 otherwise)
    hipEvent_t last_ev = nullptr;     // the last event recorded by a span (or the call's opening event) ...
    hipStream_t last_ev_stream = nullptr;      // ... and the stream it was recorded on: a span that follows on the same stream starts from it
    float last_cnn_level_ms[LLICTI_NLEVELS] = {};
    int ev_used = 0;                  // events of the pool in use
    hipEvent_t ev_call[2] = { nullptr, nullptr };
    float last_ms[4] = { 0, 0, 0, 0 };
    float last_cat_ms[PROF_NCAT] = {};
    std::vector<float> last_cnn_ms;   // per band-CNN launch, launch order
    int last_launches = 0;
    bool timing_pending = false;
};

// One profiling span: events on `s` before and after the launches it brackets (no-op unless profiling is on).
// A failed event call only loses the measurement.
struct ProfSpan {
    // Spans on one stream are CHAINED: a span starts from the event that closed the previous one (one event per span, not two), so the
    // spans tile the call's time on that stream and a kernel's span holds its dispatch latency instead of a bubble made by the
    // measurement (start and end events around every launch read 5-7 % above rocprofv3's kernel durations; chained: ~1 %).
    llicti_ctx *c;
    hipStream_t s;
    hipEvent_t e1 = nullptr;
    static hipEvent_t pool_get(llicti_ctx *c)
    {
        if (c->ev_used == (int)c->ev.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            c->ev.push_back(e);
        }
        return c->ev[c->ev_used++];
    }
    ProfSpan(llicti_ctx *c_, int cat, hipStream_t s_, int tag = -1) : c(c_), s(s_)
    {
        if (!c->profiling) return;
        hipEvent_t e0 = (c->last_ev && c->last_ev_stream == s) ? c->last_ev : nullptr;
        if (!e0) {
            e0 = pool_get(c);
            if (!e0 || hipEventRecord(e0, s) != hipSuccess) return;
        }
        hipEvent_t e = pool_get(c);
        if (!e) return;
        c->spans.push_back(llicti_ctx::Span{ e0, e, cat, tag });
        e1 = e;
    }
    ~ProfSpan()
    {
        if (!e1) return;
        if (hipEventRecord(e1, s) == hipSuccess) { c->last_ev = e1; c->last_ev_stream = s; }
        else c->spans.pop_back();
    }
    ProfSpan(const ProfSpan &) = delete;
    ProfSpan &operator=(const ProfSpan &) = delete;
};

// Every entry point that touches the device runs under this guard: the context's device becomes current for the
// call and the caller's current device is restored on return (a HipCodec on cuda:N used while the thread's current
// device is another one must not launch there, and must not leave the thread's device changed).
struct DeviceGuard {
    int prev = -1, dev = -1;
    explicit DeviceGuard(const llicti_ctx *c) {
        if (!c) return;
        dev = c->device;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

extern "C" size_t llicti_workspace_bytes_v(int B, const int *Hs, const int *Ws, int mode) { return plan_workspace_bytes_v(B, Hs, Ws, mode); }
extern "C" size_t llicti_workspace_bytes_vm(int B, const int *Hs, const int *Ws, const int *modes) { return plan_workspace_bytes_vm(B, Hs, Ws, modes, B); }
extern "C" size_t llicti_workspace_bytes(int B, int H, int W, int mode) { return plan_workspace_bytes(B, H, W, mode); }
extern "C" size_t llicti_max_container_bytes(int H, int W) { return plan_max_container_bytes(H, W); }
extern "C" int llicti_header_dims(const uint8_t *h, int *H, int *W) { return plan_header_dims(h, H, W); }
extern "C" int llicti_header_mode(const uint8_t *h, int *mode)
{
    if (!h || !mode) return fail(LLICTI_EINVAL, "header_mode: null pointer");
    int H = 0, W = 0;
    if (int rc = plan_header_dims(h, &H, &W)) return rc;          // (rejects what this build does not read, with the reason)
    if (h[0] == LLICTI_NLEVELS) { *mode = 0; return LLICTI_OK; }
    const int v = rans_streams_of_header(h[0], (int)(uint16_t)(h[15] | (h[16] << 8)));
    *mode = ((v >> 8) == 2 ? 0x500 : (v >> 8) == 1 ? 0x300 : 0x100) | (v & 0xFF);
    return LLICTI_OK;
}

extern "C" int llicti_create(llicti_ctx **out, int device)
{
    if (!out) return fail(LLICTI_EINVAL, "create: null ctx pointer");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(LLICTI_ENODEVICE, "no HIP device: this library has no CPU path");
    if (device < 0 || device >= n) return fail(LLICTI_EINVAL, "create: device %d out of range (%d devices)", device, n);
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(LLICTI_ENODEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    llicti_ctx *c = new llicti_ctx();
    c->device = device;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    DeviceGuard guard(c);
    HIPCHK(hipMalloc(&c->d_status, 64));
    HIPCHK(hipMalloc(&c->d_lift_part, (size_t)kLiftMaxParts * 4 * sizeof(int32_t)));
    HIPCHK(hipMemset(c->d_status, 0, 64));
    {
        std::vector<float2> lut(kPhiLutN);
        for (int j = 0; j < kPhiLutN; ++j) {
            const double z0 = -kPhiLutZ + j * (2.0 * kPhiLutZ / kPhiLutN), z1 = -kPhiLutZ + (j + 1) * (2.0 * kPhiLutZ / kPhiLutN);
            const double v0 = 0.5 * std::erfc(-z0 * 0.70710678118654752440), v1 = 0.5 * std::erfc(-z1 * 0.70710678118654752440);
            lut[j] = make_float2((float)v0, (float)(v1 - v0));
        }
        HIPCHK(hipMalloc(&c->d_phi_lut, sizeof(float2) * kPhiLutN));
        HIPCHK(hipMemcpy(c->d_phi_lut, lut.data(), sizeof(float2) * kPhiLutN, hipMemcpyHostToDevice));
    }
    HIPCHK(hipEventCreate(&c->ev_call[0]));
    HIPCHK(hipEventCreate(&c->ev_call[1]));
    for (int k = 0; k < 2; ++k) {
        for (int i = 0; i < 16; ++i) HIPCHK(hipEventCreateWithFlags(&c->ev_ac[k][i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_ac_end[k], hipEventDisableTiming));
    }
    HIPCHK(hipEventCreateWithFlags(&c->ev_ac_band, hipEventDisableTiming));
    for (int k = 0; k < 2; ++k) HIPCHK(hipEventCreateWithFlags(&c->ev_enc[k], hipEventDisableTiming));
    for (int i = 1; i < kMaxSub; ++i) HIPCHK(hipStreamCreateWithFlags(&c->sub[i], hipStreamNonBlocking));
    // the band CNN stages a whole head (up to 86 KB) in LDS
#define LLICTI_CNN_ATTR(BAND, TH_) \
    HIPCHK(hipFuncSetAttribute((const void *)band_params_kernel<BAND, TH_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, cnn_lds_bytes(BAND, TH_))); \
    HIPCHK(hipFuncSetAttribute((const void *)band_params_kernel<BAND, TH_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cnn_lds_bytes(BAND, TH_)))
    LLICTI_CNN_ATTR(0, kTileHMax); LLICTI_CNN_ATTR(1, kTileHMax); LLICTI_CNN_ATTR(2, kTileHMax);
    LLICTI_CNN_ATTR(0, kTileHMid); LLICTI_CNN_ATTR(1, kTileHMid); LLICTI_CNN_ATTR(2, kTileHMid);
    LLICTI_CNN_ATTR(0, kTileHSmall); LLICTI_CNN_ATTR(1, kTileHSmall); LLICTI_CNN_ATTR(2, kTileHSmall);
#undef LLICTI_CNN_ATTR
    *out = c;
    return LLICTI_OK;
}

extern "C" int llicti_destroy(llicti_ctx *c)
{
    if (!c) return LLICTI_OK;
    {
    DeviceGuard guard(c);
    (void)hipDeviceSynchronize();
    for (int b = 0; b < 3; ++b) if (c->d_pack[b]) hipFree(c->d_pack[b]);
    for (auto &kv : c->plans) { c->pool.push_back(kv.second->blk); delete kv.second; }
    c->plans.clear();
    for (PlanBlock &bk : c->pool) {
        if (bk.dev) (void)hipFree(bk.dev);
        if (bk.host) (void)hipHostFree(bk.host);
        if (bk.done) (void)hipEventDestroy(bk.done);
        if (bk.uploaded) (void)hipEventDestroy(bk.uploaded);
    }
    c->pool.clear();
    for (int i = 0; i < kMaxSub; ++i) {
        if (c->sub[i]) hipStreamDestroy(c->sub[i]);
    }
    for (int k = 0; k < 2; ++k) {
        for (int i = 0; i < 16; ++i) if (c->ev_ac[k][i]) hipEventDestroy(c->ev_ac[k][i]);
        if (c->ev_ac_end[k]) hipEventDestroy(c->ev_ac_end[k]);
    }
    if (c->ev_ac_band) hipEventDestroy(c->ev_ac_band);
    for (int k = 0; k < 2; ++k) if (c->ev_enc[k]) hipEventDestroy(c->ev_enc[k]);
    if (c->d_status) hipFree(c->d_status);
    if (c->d_lift_part) hipFree(c->d_lift_part);
    if (c->d_phi_lut) hipFree(c->d_phi_lut);
    if (c->d_img_status) hipFree(c->d_img_status);
    for (auto e : c->ev) hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) if (c->ev_call[i]) hipEventDestroy(c->ev_call[i]);
    }
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
    DeviceGuard guard(c);
    // blocking by contract (include/llicti_hip.h): work in flight on any stream may still read the old weights
    HIPCHK(hipDeviceSynchronize());
    if (!c->d_pack[band]) HIPCHK(hipMalloc(&c->d_pack[band], pk.size() * sizeof(float)));
    HIPCHK(hipMemcpy(c->d_pack[band], pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
    c->have[band] = true;
    return LLICTI_OK;
}

// every cached plan back to the block pool (no device work: the blocks wait there for their last users)
static void drop_plans(llicti_ctx *c)
{
    for (auto &kv : c->plans) { c->pool.push_back(kv.second->blk); delete kv.second; }
    c->plans.clear();
}

extern "C" int llicti_set_tuning(llicti_ctx *c, const char *key, int value)
{
    if (!c || !key) return fail(LLICTI_EINVAL, "set_tuning: null argument");
    if (!strcmp(key, "ac_anchor_min_batch")) {
        if (value < 1) return fail(LLICTI_EINVAL, "set_tuning: ac_anchor_min_batch must be >= 1");
        c->ac_anchor_min_batch = value;
        return LLICTI_OK;
    }
    if (!strcmp(key, "cnn_tile_rows")) {
        if (value != 0 && value != kTileHMax && value != kTileHMid && value != kTileHSmall && value != -1)
            return fail(LLICTI_EINVAL, "set_tuning: cnn_tile_rows must be 0 (automatic), %d, %d or %d (-1: round 3's rule, 16 or 4)", kTileHMax, kTileHMid, kTileHSmall);
        if (c->cnn_tile_rows != value) drop_plans(c);       // (a mixed-size plan holds the tile lists of the forms chosen when it was built)
        c->cnn_tile_rows = value;
        return LLICTI_OK;
    }
    if (!strcmp(key, "enc_side_levels")) {
        if (value < 0 || value > 1) return fail(LLICTI_EINVAL, "set_tuning: enc_side_levels must be 0 or 1");
        c->enc_side_levels = value;
        return LLICTI_OK;
    }
    if (!strcmp(key, "force_ragged")) {
        if (value < 0 || value > 1) return fail(LLICTI_EINVAL, "set_tuning: force_ragged must be 0 or 1");
        c->force_ragged = value;
        return LLICTI_OK;
    }
    return fail(LLICTI_EINVAL, "set_tuning: unknown key '%s'", key);
}

extern "C" int llicti_workspace_planes(llicti_ctx *c, int B, int H, int W, int mode, size_t *off_planes, size_t *off_fplanes)
{
    if (!c || !off_planes || !off_fplanes) return fail(LLICTI_EINVAL, "workspace_planes: null argument");
    if (check_dims(B, H, W)) return LLICTI_EINVAL;
    const int ME = mode_streams(mode);
    if (ME < 0) return fail(LLICTI_EINVAL, "workspace_planes: unknown mode 0x%x", mode);
    std::vector<int> Hs(B, H), Ws(B, W);
    Plan p;
    build_plan(p, B, Hs.data(), Ws.data(), nullptr, ME, c->n_cu, c->cnn_tile_rows, c->force_ragged != 0);
    if (!p.uniform && B > 1) return fail(LLICTI_EINVAL, "workspace_planes: with the tuning switch force_ragged the images of a batch are not tightly packed");
    *off_planes = p.off_planes;
    *off_fplanes = p.off_fplanes;
    return LLICTI_OK;
}

extern "C" int llicti_get_counter(llicti_ctx *c, const char *name, long *value)
{
    if (!c || !name || !value) return fail(LLICTI_EINVAL, "get_counter: null argument");
    if (!strcmp(name, "device_syncs")) *value = c->n_device_sync;
    else if (!strcmp(name, "device_allocs")) *value = c->n_device_alloc;
    else if (!strcmp(name, "plan_builds")) *value = c->n_plan_build;
    else if (!strcmp(name, "plan_hits")) *value = c->n_plan_hit;
    else if (!strcmp(name, "block_waits")) *value = c->n_block_wait;
    else if (!strcmp(name, "plans_cached")) *value = (long)c->plans.size();
    else if (!strcmp(name, "blocks_pooled")) *value = (long)c->pool.size();
    else return fail(LLICTI_EINVAL, "get_counter: unknown counter '%s'", name);
    return LLICTI_OK;
}

extern "C" int llicti_set_profiling(llicti_ctx *c, int enable)
{
    if (!c) return fail(LLICTI_EINVAL, "null ctx");
    c->profiling = enable != 0;
    return LLICTI_OK;
}

// ------------------------------------------------------------------------------------------------ launches
// part: scratch of kLiftMaxParts x 4 int32 (the workspace's for the whole-batch calls, the context's for llicti_lift_u8)
static int launch_lift(const uint8_t *d_rgb, int B, long plane, bool vec_ok, int16_t *planes, float *fplanes, int32_t *mm, int32_t *part, hipStream_t s,
                       int32_t *zero = nullptr, int n_zero = 0, const ImgGeo *iv = nullptr)
{
    // plane: H * W (with a table: of the batch's largest image -- it sizes the grid); vec_ok: every image's plane size and placement allow
    // 4-pixel accesses
    const bool vec = vec_ok && (plane % 4 == 0) && (((uintptr_t)d_rgb | (uintptr_t)planes | (uintptr_t)fplanes) % 16 == 0);
    const long want = vec ? (plane / 4 + 255) / 256 : (plane + 255) / 256;
    const int gx = (int)std::max<long>(1, std::min<long>(std::min<long>(want, std::max(8, 4096 / B)), kLiftMaxParts / B));
    if ((long)B * gx > kLiftMaxParts) return fail(LLICTI_EINVAL, "lift: batch of %d images exceeds the partials scratch", B);
    if (vec) lift_kernel<4><<<dim3(gx, B), 256, 0, s>>>(d_rgb, plane, planes, fplanes, part, zero, n_zero, iv);
    else lift_kernel<1><<<dim3(gx, B), 256, 0, s>>>(d_rgb, plane, planes, fplanes, part, zero, n_zero, iv);
    minmax_reduce_kernel<<<B, 64, 0, s>>>(part, gx, mm);
    HIPCHK(hipGetLastError());
    return 0;
}

// One band-CNN launch.  Equal sizes (tiles == nullptr): B images of g's size, tile -> image by division; mixed sizes: the plan's tile list of
// this (level, band) and its per-image geometry table, form and grid chosen when the plan was built.
static int launch_band_params(llicti_ctx *c, const float *fplanes, const Geom &g, int band, float *params, hipStream_t s,
                              const Geom *gv = nullptr, const TileRef *tiles = nullptr, const TileRun *run = nullptr)
{
    if (!c->have[band]) return fail(LLICTI_ENOWEIGHTS, "band %d weights not set", band);
    const int tiles_x = (g.w + kTileW - 1) / kTileW;
    int TH, gx;
    long n_tiles_l;
    if (tiles) { TH = run->TH; gx = run->gx; n_tiles_l = run->n_tiles; }
    else {
        const TileForm f = choose_tile_form(c->n_cu, c->cnn_tile_rows, band, [&](int th) { return (long)g.B * tiles_x * ((g.h + th - 1) / th); });
        TH = f.TH; gx = f.gx; n_tiles_l = f.n_tiles;
    }
    const int tiles_y = (g.h + TH - 1) / TH;
    if (n_tiles_l > 0x7FFFFFFFL || n_tiles_l < 1) return fail(LLICTI_EINVAL, "band_params: bad tile count");
    const int n_tiles = (int)n_tiles_l;
    const int lds_bytes = cnn_lds_bytes(band, TH);
    const int kCnnThreads = 64 * TH;
    dim3 grid((unsigned)gx, 4);
    ProfSpan span(c, PROF_CNN, s, g.lvl);
#define LLICTI_CNN_LAUNCH(BAND, TH_, RAG) band_params_kernel<BAND, TH_, RAG><<<grid, kCnnThreads, lds_bytes, s>>>(fplanes, g, c->d_pack[BAND], params, tiles_x, tiles_y, n_tiles, gv, tiles)
    const int form = band + (TH == kTileHSmall ? 3 : TH == kTileHMid ? 6 : 0) + (tiles ? 9 : 0);
    switch (form) {
    case 0: LLICTI_CNN_LAUNCH(0, kTileHMax, false); break;
    case 1: LLICTI_CNN_LAUNCH(1, kTileHMax, false); break;
    case 2: LLICTI_CNN_LAUNCH(2, kTileHMax, false); break;
    case 3: LLICTI_CNN_LAUNCH(0, kTileHSmall, false); break;
    case 4: LLICTI_CNN_LAUNCH(1, kTileHSmall, false); break;
    case 5: LLICTI_CNN_LAUNCH(2, kTileHSmall, false); break;
    case 6: LLICTI_CNN_LAUNCH(0, kTileHMid, false); break;
    case 7: LLICTI_CNN_LAUNCH(1, kTileHMid, false); break;
    case 8: LLICTI_CNN_LAUNCH(2, kTileHMid, false); break;
    case 9: LLICTI_CNN_LAUNCH(0, kTileHMax, true); break;
    case 10: LLICTI_CNN_LAUNCH(1, kTileHMax, true); break;
    case 11: LLICTI_CNN_LAUNCH(2, kTileHMax, true); break;
    case 12: LLICTI_CNN_LAUNCH(0, kTileHSmall, true); break;
    case 13: LLICTI_CNN_LAUNCH(1, kTileHSmall, true); break;
    case 14: LLICTI_CNN_LAUNCH(2, kTileHSmall, true); break;
    case 15: LLICTI_CNN_LAUNCH(0, kTileHMid, true); break;
    case 16: LLICTI_CNN_LAUNCH(1, kTileHMid, true); break;
    default: LLICTI_CNN_LAUNCH(2, kTileHMid, true); break;
    }
#undef LLICTI_CNN_LAUNCH
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int llicti_lift_u8(llicti_ctx *c, const uint8_t *d_rgb, int B, int H, int W, int16_t *d_planes,
                              float *d_fplanes, int32_t *d_minmax, void *stream)
{
    if (!c || !d_rgb || !d_planes || !d_fplanes || !d_minmax) return fail(LLICTI_EINVAL, "lift: null pointer");
    DeviceGuard guard(c);
    if (check_dims(B, H, W)) return LLICTI_EINVAL;
    return launch_lift(d_rgb, B, (long)H * W, true, d_planes, d_fplanes, d_minmax, c->d_lift_part, (hipStream_t)stream);
}

extern "C" int llicti_unlift_u8(llicti_ctx *c, const int16_t *d_planes, int B, int H, int W, uint8_t *d_rgb, void *stream)
{
    if (!c || !d_planes || !d_rgb) return fail(LLICTI_EINVAL, "unlift: null pointer");
    DeviceGuard guard(c);
    if (check_dims(B, H, W)) return LLICTI_EINVAL;
    const long plane = (long)H * W;
    const int gx = (int)std::min<long>((plane + 255) / 256, 1024);
    unlift_kernel<<<dim3(gx, B), 256, 0, (hipStream_t)stream>>>(d_planes, plane, d_rgb, nullptr, 0, nullptr, nullptr, nullptr);
    HIPCHK(hipGetLastError());
    return LLICTI_OK;
}

extern "C" int llicti_band_params_f32(llicti_ctx *c, const float *d_fplanes, int B, int H, int W, int lvl, int band,
                                      float *d_params, void *stream)
{
    if (!c || !d_fplanes || !d_params) return fail(LLICTI_EINVAL, "band_params: null pointer");
    DeviceGuard guard(c);
    if (check_dims(B, H, W)) return LLICTI_EINVAL;
    if (lvl < 0 || lvl >= LLICTI_NLEVELS || band < 0 || band > 2) return fail(LLICTI_EINVAL, "band_params: bad level/band");
    Geom g = make_geom(B, H, W, lvl);
    return launch_band_params(c, d_fplanes, g, band, d_params, (hipStream_t)stream);
}

extern "C" int llicti_lift_train_f32(llicti_ctx *c, const uint8_t *d_rgb, int B, int H, int W, float *d_fplanes, void *stream)
{
    if (!c || !d_rgb || !d_fplanes) return fail(LLICTI_EINVAL, "lift_train: null pointer");
    DeviceGuard guard(c);
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
    DeviceGuard guard(c);
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

// kernel-level entry point: B images of one size, pairs [clr][B][n]
static int launch_cdf_pairs(const int16_t *planes, const float *params, const int32_t *mm, const Geom &g, int band,
                            uint32_t *pairs, hipStream_t s)
{
    StageGeom sg = make_stage(g, band);
    sg.pair_cs = (long)g.B * sg.hc * sg.wc;
    const long np = (long)sg.h * sg.w;                    // the kernel walks the band grid (rows of CNN outputs are contiguous there)
    cdf_pairs_kernel<<<dim3((unsigned)((np + kPairsThreads - 1) / kPairsThreads), g.B), kPairsThreads, 0, s>>>(planes, params, mm, sg, nullptr, pairs);
    HIPCHK(hipGetLastError());
    return 0;
}
static int launch_cdf_anchors(const int16_t *planes, const float *params, const int32_t *mm, const Geom &g, int band, int clr,
                              uint8_t *rows, long n0, long cnt, long cap_rows, hipStream_t s)
{
    StageGeom sg = make_stage(g, band);
    const long want = (cnt + kTabWaves - 1) / kTabWaves;
    const long cap = std::max<long>(1, (256L * 8 * 2) / std::max(1, g.B));
    cdf_anchor_kernel<<<dim3((unsigned)std::max<long>(1, std::min(want, cap)), g.B), 64 * kTabWaves, 0, s>>>(planes, params, mm, sg, clr, rows,
                                                                                                   (int)n0, (int)cnt, (int)cap_rows);
    HIPCHK(hipGetLastError());
    return 0;
}
static int launch_cdf_table(const int16_t *planes, const float *params, const int32_t *mm, const Geom &g, int band, int clr,
                            uint16_t *tables, int row_stride, long n0, long cnt, long cap_rows, hipStream_t s)
{
    StageGeom sg = make_stage(g, band);
    const long want = (cnt + kTabWaves - 1) / kTabWaves;                    // one row per wave ...
    const long cap = std::max<long>(1, (256L * 8 * 2) / std::max(1, g.B));   // ... up to ~16 waves per SIMD-quad in flight per image set
    cdf_table_kernel<<<dim3((unsigned)std::max<long>(1, std::min(want, cap)), g.B), 64 * kTabWaves, 0, s>>>(planes, params, mm, sg, clr, tables, row_stride,
                                                                                                  (int)n0, (int)cnt, (int)cap_rows);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int llicti_cdf_u16(llicti_ctx *c, const int16_t *d_planes, const float *d_params, const int32_t *d_minmax,
                              int B, int H, int W, int lvl, int band, int clr, uint16_t *d_tables, int row_stride, void *stream)
{
    if (!c || !d_planes || !d_params || !d_minmax || !d_tables) return fail(LLICTI_EINVAL, "cdf_u16: null pointer");
    DeviceGuard guard(c);
    if (check_dims(B, H, W)) return LLICTI_EINVAL;
    if (lvl < 0 || lvl >= LLICTI_NLEVELS || band < 0 || band > 2 || clr < 0 || clr > 2) return fail(LLICTI_EINVAL, "cdf_u16: bad level/band/clr");
    if (row_stride < 8 || row_stride > 512 || (row_stride & 7)) return fail(LLICTI_EINVAL, "cdf_u16: row_stride must be a multiple of 8 in [8,512]");
    Geom g = make_geom(B, H, W, lvl);
    int hc, wc;
    coded_dims(g, band, &hc, &wc);
    const long nc = (long)hc * wc;
    return launch_cdf_table(d_planes, d_params, d_minmax, g, band, clr, d_tables, row_stride, 0, nc, nc, (hipStream_t)stream);
}

extern "C" int llicti_cdf_pairs_u32(llicti_ctx *c, const int16_t *d_planes, const float *d_params, const int32_t *d_minmax,
                                    int B, int H, int W, int lvl, int band, uint32_t *d_pairs, void *stream)
{
    if (!c || !d_planes || !d_params || !d_minmax || !d_pairs) return fail(LLICTI_EINVAL, "cdf_pairs: null pointer");
    DeviceGuard guard(c);
    if (check_dims(B, H, W)) return LLICTI_EINVAL;
    if (lvl < 0 || lvl >= LLICTI_NLEVELS || band < 0 || band > 2) return fail(LLICTI_EINVAL, "cdf_pairs: bad level/band");
    Geom g = make_geom(B, H, W, lvl);
    return launch_cdf_pairs(d_planes, d_params, d_minmax, g, band, d_pairs, (hipStream_t)stream);
}

extern "C" int llicti_ac_encode_u16cdf(llicti_ctx *c, const uint16_t *d_cdf, int Lp, int row_stride, const int16_t *d_sym,
                                       int n_streams, long N, uint8_t *d_out, long out_stride, int32_t *d_len, void *stream)
{
    if (!c || !d_cdf || !d_sym || !d_out || !d_len) return fail(LLICTI_EINVAL, "ac_encode: null pointer");
    DeviceGuard guard(c);
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
    DeviceGuard guard(c);
    if (Lp < 2 || Lp > 512 || row_stride < Lp || row_stride > 512 || (row_stride & 7) || n_streams < 1 || N < 1 || (in_stride & 3) ||
        ((uintptr_t)d_in & 3) || ((uintptr_t)d_cdf & 15))
        return fail(LLICTI_EINVAL, "ac_decode: bad argument (Lp<=512, row_stride multiple of 8, 4-byte aligned streams, 16-byte aligned tables)");
    DecOut o;
    memset(&o, 0, sizeof o);
    o.sym = d_sym;
    o.len = d_len;          // bytes past d_len[s] read as zero bits (torchac's get()), whatever the buffer holds there
    AcChunk ck = { 0, (int)N, (int)N, (int)N, nullptr };
    ac_decode_kernel<<<n_streams, 64, 0, (hipStream_t)stream>>>(d_cdf, Lp, row_stride, d_in, in_stride, ck, o);
    HIPCHK(hipGetLastError());
    return LLICTI_OK;
}

// ------------------------------------------------------------------------------------------------ whole batch
// A block of >= need bytes for a new plan's tables: one from the pool whose last user has finished, else a new one.  Never hipFree
// (which synchronises the device) before llicti_destroy; the pool is bounded by kMaxPlans + the blocks in flight.
static int acquire_block(llicti_ctx *c, size_t need, PlanBlock *out)
{
    int pick = -1, busy_fit = -1;
    for (int i = 0; i < (int)c->pool.size(); ++i) {
        PlanBlock &bk = c->pool[i];
        if (bk.cap < need) continue;
        if (!bk.used || hipEventQuery(bk.done) == hipSuccess) { if (pick < 0 || bk.cap < c->pool[pick].cap) pick = i; }
        else if (busy_fit < 0) busy_fit = i;
    }
    if (pick < 0 && busy_fit >= 0 && c->pool.size() >= 16) {      // plenty of blocks, all still in flight: wait for the oldest one's user (a stream-level wait)
        ++c->n_block_wait;
        HIPCHK(hipEventSynchronize(c->pool[busy_fit].done));
        pick = busy_fit;
    }
    if (pick >= 0) {
        *out = c->pool[pick];
        c->pool.erase(c->pool.begin() + pick);
        out->used = false;
        return 0;
    }
    PlanBlock bk;
    size_t cap = 256 << 10;
    while (cap < need) cap *= 2;
    bk.cap = cap;
    ++c->n_device_alloc;
    if (hipMalloc(&bk.dev, cap) != hipSuccess) return fail(LLICTI_EHIP, "plan tables: hipMalloc(%zu) failed", cap);
    if (hipHostMalloc(&bk.host, cap, hipHostMallocDefault) != hipSuccess) { (void)hipFree(bk.dev); return fail(LLICTI_EHIP, "plan tables: hipHostMalloc(%zu) failed", cap); }
    if (hipEventCreateWithFlags(&bk.done, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&bk.uploaded, hipEventDisableTiming) != hipSuccess) {
        (void)hipFree(bk.dev); (void)hipHostFree(bk.host);
        if (bk.done) (void)hipEventDestroy(bk.done);
        return fail(LLICTI_EHIP, "plan tables: hipEventCreate failed");
    }
    *out = bk;
    return 0;
}

// The plan of a batch: cached by (mode, sizes, placement).  A miss builds the tables on the host, copies them into a pinned block and
// enqueues ONE asynchronous upload on the call's stream -- no device synchronisation, no allocation once the pool is warm.
static int get_plan(llicti_ctx *c, int B, const int *Hs, const int *Ws, const size_t *rgb_off, int ME, const int *Ms, hipStream_t s, PlanDev **out)
{
    std::vector<long> key;
    key.reserve(3 + 4 * (size_t)B);
    key.push_back(ME); key.push_back(B); key.push_back(c->cnn_tile_rows * 2 + (c->force_ragged ? 1 : 0));
    {
        long pos = 0;
        for (int b = 0; b < B; ++b) {
            key.push_back(Hs[b]); key.push_back(Ws[b]); key.push_back(rgb_off ? (long)rgb_off[b] : pos); key.push_back((Ms && (ME & 0xFF)) ? Ms[b] : (ME & 0xFF));
            pos += 3L * Hs[b] * Ws[b];
        }
    }
    auto it = c->plans.find(key);
    if (it != c->plans.end()) {
        PlanDev *pd = it->second;
        ++c->n_plan_hit;
        pd->last_use = ++c->use_clock;
        if (pd->blk.up_stream != s) HIPCHK(hipStreamWaitEvent(s, pd->blk.uploaded, 0));      // (uploaded on another stream: order behind it)
        if (pd->blk.used && pd->blk.done_stream != s) HIPCHK(hipStreamWaitEvent(s, pd->blk.done, 0));      // (last used on another stream: this call's `done` must lie behind that use too)
        *out = pd;
        return 0;
    }
    std::unique_ptr<PlanDev> pd(new PlanDev());
    Plan &p = pd->p;
    ++c->n_plan_build;
    build_plan(p, B, Hs, Ws, rgb_off, ME, c->n_cu, c->cnn_tile_rows, c->force_ragged != 0, Ms);
    if (p.key != key) return fail(LLICTI_EINVAL, "plan: key mismatch");
    if (p.rslot_off.size() != (size_t)p.nstreams || p.sref.size() != (size_t)p.nstreams)
        return fail(LLICTI_EINVAL, "plan: stream tables have %zu / %zu entries, expected %d", p.rslot_off.size(), p.sref.size(), p.nstreams);
    if (int rc = acquire_block(c, p.d_total, &pd->blk)) return rc;
    uint8_t *h = pd->blk.host;
    auto put = [&](size_t off, const void *src, size_t n) { if (n) memcpy(h + off, src, n); };
    put(p.d_img, p.img.data(), p.img.size() * sizeof(ImgGeo));
    put(p.d_geo, p.geo.data(), p.geo.size() * sizeof(Geom));
    put(p.d_sg, p.sg.data(), p.sg.size() * sizeof(StageGeom));
    put(p.d_desc, p.desc.data(), p.desc.size() * sizeof(StreamDesc));
    put(p.d_slot_off, p.slot_off.data(), p.slot_off.size() * sizeof(long));
    put(p.d_slot_cap, p.slot_cap.data(), p.slot_cap.size() * sizeof(int32_t));
    put(p.d_rslot_off, p.rslot_off.data(), p.rslot_off.size() * sizeof(long));
    put(p.d_tiles, p.tiles.data(), p.tiles.size() * sizeof(TileRef));
    put(p.d_sref, p.sref.data(), p.sref.size() * sizeof(StreamRef));
    bool ok = hipMemcpyAsync(pd->blk.dev, h, p.d_total, hipMemcpyHostToDevice, s) == hipSuccess;
    ok = ok && hipEventRecord(pd->blk.uploaded, s) == hipSuccess;
    ok = ok && hipEventRecord(pd->blk.done, s) == hipSuccess;      // (so that the block is never recycled in front of its own upload)
    pd->blk.up_stream = s;
    pd->blk.done_stream = s;
    pd->blk.used = true;
    if (!ok) { c->pool.push_back(pd->blk); return fail(LLICTI_EHIP, "plan tables: upload failed"); }
    p.tiles.clear(); p.tiles.shrink_to_fit();                      // (the host copy of the largest table is not needed again)
    if ((int)c->plans.size() >= kMaxPlans) {                       // least recently used out; its block waits in the pool for its last user
        auto lru = c->plans.begin();
        for (auto jt = c->plans.begin(); jt != c->plans.end(); ++jt) if (jt->second->last_use < lru->second->last_use) lru = jt;
        c->pool.push_back(lru->second->blk);
        delete lru->second;
        c->plans.erase(lru);
    }
    pd->last_use = ++c->use_clock;
    *out = pd.get();
    c->plans[key] = pd.release();
    return 0;
}

// status[0] -> the context's latched word (llicti_check_status); decode: the B per-image words -> the context's copy (llicti_image_status)
// the call's status words are cleared by a kernel of the call's own queue: hipMemsetAsync goes through the runtime's blit path, which left the queue
// idle for ~17 us in front of every encode and decode (rocprofv3 trace, tools/trace_gaps.py)
__global__ void zero_words_kernel(int32_t *p, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}
__global__ void latch_status_kernel(const int32_t *status, int32_t *latched, int32_t *img_latched, int B)
{
    if (threadIdx.x == 0 && *status != 0) *latched = *status;
    if (img_latched) for (int b = threadIdx.x; b < B; b += blockDim.x) img_latched[b] = status[kStatusHead + b];
}

// The profiled extent of one whole-batch call: the closing event is recorded on EVERY way out (an early error return
// included), so llicti_last_timing never waits on an event of a call that did not record it.
struct CallScope {
    llicti_ctx *c;
    hipStream_t s;
    bool on;
    CallScope(llicti_ctx *c_, hipStream_t s_) : c(c_), s(s_), on(c_->profiling)
    {
        c->ev_used = 0;
        c->spans.clear();
        c->last_ev = nullptr;
        c->timing_pending = false;
        if (on) on = hipEventRecord(c->ev_call[0], s) == hipSuccess;
        if (on) { c->last_ev = c->ev_call[0]; c->last_ev_stream = s; }
    }
    ~CallScope() { if (on && hipEventRecord(c->ev_call[1], s) == hipSuccess) c->timing_pending = true; }
    CallScope(const CallScope &) = delete;
    CallScope &operator=(const CallScope &) = delete;
};

// marks the plan's table block as in use behind everything the call enqueued (get_plan: blocks are recycled, never freed)
struct PlanUse {
    PlanDev *pd; hipStream_t s;
    ~PlanUse() { if (pd && hipEventRecord(pd->blk.done, s) == hipSuccess) { pd->blk.used = true; pd->blk.done_stream = s; } }
};

// modes: one container mode for the call (n_modes = 1) or one per image (n_modes = B: rANS containers of ONE lane kind whose stream counts may differ);
// -> ME of the call (the lane kind, with the first image's count) and, for per-image counts, Ms
static int resolve_modes(const char *who, const int *modes, int n_modes, int B, int *ME_out, std::vector<int> &Ms)
{
    Ms.clear();
    if (!modes || (n_modes != 1 && n_modes != B)) return fail(LLICTI_EINVAL, "%s: modes must hold one mode or one per image", who);
    const int ME0 = mode_streams(modes[0]);
    if (ME0 < 0) return fail(LLICTI_EINVAL, "%s: unknown mode 0x%x", who, modes[0]);
    *ME_out = ME0;
    if (n_modes == 1) return 0;
    bool differ = false;
    for (int b = 0; b < B; ++b) {
        const int ME = mode_streams(modes[b]);
        if (ME < 0) return fail(LLICTI_EINVAL, "%s: unknown mode 0x%x of image %d", who, modes[b], b);
        if ((ME >> 8) != (ME0 >> 8) || ((ME & 0xFF) == 0) != ((ME0 & 0xFF) == 0))
            return fail(LLICTI_EINVAL, "%s: the images of one call share a container kind (reference format, or rANS streams of one lane count); image %d differs", who, b);
        Ms.push_back(ME & 0xFF);
        differ = differ || ME != ME0;
    }
    if (!differ) Ms.clear();
    return 0;
}

extern "C" int llicti_workspace_params_v(llicti_ctx *c, int B, const int *Hs, const int *Ws, const int *modes, int image, size_t *off_params, long *npos)
{
    if (!c || !off_params || !npos || !modes) return fail(LLICTI_EINVAL, "workspace_params_v: null argument");
    if (check_dims_v(B, Hs, Ws)) return LLICTI_EINVAL;
    if (image < 0 || image >= B) return fail(LLICTI_EINVAL, "workspace_params_v: image %d of %d", image, B);
    int ME = 0;
    std::vector<int> Ms;
    if (int rc = resolve_modes("workspace_params_v", modes, B, B, &ME, Ms)) return rc;
    Plan p;
    build_plan(p, B, Hs, Ws, nullptr, ME, c->n_cu, c->cnn_tile_rows, c->force_ragged != 0, Ms.empty() ? nullptr : Ms.data());
    const Geom &g = p.geo[(size_t)0 * B + image];            // level 0: the last level both passes launch
    *off_params = p.off_params + (size_t)g.par_off * sizeof(float);
    *npos = (long)g.h * g.w;
    return LLICTI_OK;
}

static int encode_batch(llicti_ctx *c, const uint8_t *d_rgb, const size_t *rgb_off, int B, const int *Hs, const int *Ws, const int *modes, int n_modes,
                        void *d_workspace, size_t workspace_bytes, uint8_t *d_out, size_t out_stride, int32_t *d_seg_len, void *stream)
{
    if (!c || !d_rgb || !d_workspace || !d_out || !d_seg_len) return fail(LLICTI_EINVAL, "encode_images: null pointer");
    if (check_dims_v(B, Hs, Ws)) return LLICTI_EINVAL;
    int ME = 0;
    std::vector<int> Ms;
    if (int rc = resolve_modes("encode_images", modes, n_modes, B, &ME, Ms)) return rc;
    const int Q = 1 << ((ME >> 8) & 3);
    const bool autoM = (ME & 0x1000) != 0;
    for (int b = 0; b < 3; ++b) if (!c->have[b]) return fail(LLICTI_ENOWEIGHTS, "band %d weights not set", b);
    DeviceGuard guard(c);
    hipStream_t s = (hipStream_t)stream;
    PlanDev *pd = nullptr;
    if (int rc = get_plan(c, B, Hs, Ws, rgb_off, ME, Ms.empty() ? nullptr : Ms.data(), s, &pd)) return rc;
    PlanUse use{ pd, s };
    const Plan &p = pd->p;
    const int M = p.M;
    if (!p.uniform && M == 0) return fail(LLICTI_EINVAL, "encode_images: a batch of mixed sizes needs a rANS container (the reference-format container codes equal sizes per call)");
    if (workspace_bytes < p.total) return fail(LLICTI_ENOSPACE, "encode_images: workspace %zu < %zu", workspace_bytes, p.total);
    if (out_stride < p.max_container) return fail(LLICTI_ENOSPACE, "encode_images: out_stride %zu < %zu", out_stride, p.max_container);
    uint8_t *ws = (uint8_t *)d_workspace;
    int16_t *planes = (int16_t *)(ws + p.off_planes);
    float *fplanes = (float *)(ws + p.off_fplanes);
    int32_t *mm = (int32_t *)(ws + p.off_minmax);
    int32_t *status = (int32_t *)(ws + p.off_status);
    uint32_t *pairs = (uint32_t *)(ws + p.off_pairs);
    uint8_t *slots = ws + p.off_slots;
    int32_t *slot_len = (int32_t *)(ws + p.off_slot_len);
    const ImgGeo *d_img = pd->dev<ImgGeo>(p.d_img);
    const Geom *d_geo = pd->dev<Geom>(p.d_geo);
    const StageGeom *d_sg = pd->dev<StageGeom>(p.d_sg);
    const StreamDesc *d_desc = pd->dev<StreamDesc>(p.d_desc);
    const long *d_rslot_off = pd->dev<long>(p.d_rslot_off);
    const StreamRef *d_sref = pd->dev<StreamRef>(p.d_sref);
    const TileRef *d_tiles = p.uniform ? nullptr : pd->dev<TileRef>(p.d_tiles);

    CallScope call(c, s);
    {
        ProfSpan span(c, PROF_MISC, s);
        // (the lift is the call's first kernel and sets no status: it clears the call's status words on the way)
        if (int rc = launch_lift(d_rgb, B, p.max_plane, p.vec_ok, planes, fplanes, mm, (int32_t *)(ws + p.off_lift_part), s, status, kStatusHead + B, d_img)) return rc;
        header_write_kernel<<<B, 256, 0, s>>>(d_rgb, mm, d_img, d_out, (long)out_stride, d_seg_len, autoM ? (unsigned long long *)(ws + p.off_rpos) : nullptr);
    }
    // The encoder has no dependency between stages: every (level, band) reads only original pixels.  With llicti_set_tuning("enc_side_levels", 1)
    // levels 4..1 (twelve CNN + twelve pairs launches, a quarter of the work) run on a side stream next to level 0's, with their own
    // quarter-size buffer for the CNN outputs.  Two CNN launches cannot share a compute unit (a workgroup holds 81-155 KB of its 160 KB LDS),
    // so this is no second queue of matrix work: what overlaps is the small stuff -- a pairs kernel under the other queue's CNN launch, a
    // launch's ramp-down under the other's ramp-up.  Measured (profiles/r4/ab_encoder_side_stream.json): -0.5 ... -4 % per encode, 10.75 ->
    // 10.68 ms at B = 24 -- 0.3 % of a step, for which every kernel trace of the run shows the side queue's launches with the time they
    // spend waiting for a compute unit inside their durations.  Off by default: the per-kernel evidence is worth more than 0.07 ms.
    const bool side = c->enc_side_levels != 0 && !c->profiling;      // (the profiling spans assume one queue)
    hipStream_t s2 = side ? c->sub[1] : s;
    if (side) {
        HIPCHK(hipEventRecord(c->ev_enc[0], s));      // lift and header are done
        HIPCHK(hipStreamWaitEvent(s2, c->ev_enc[0], 0));
    }
    for (int lvl = LLICTI_NLEVELS - 1; lvl >= 0; --lvl) {
        const Geom &g = p.geo[(size_t)lvl * B];                         // image 0's (equal sizes: every image's)
        const Geom *gv = p.uniform ? nullptr : d_geo + (size_t)lvl * B;
        hipStream_t s = (lvl >= 1) ? s2 : (hipStream_t)stream;          // (shadows the call's stream inside the loop)
        float *params = (lvl >= 1 && side) ? (float *)(ws + p.off_params2) : (float *)(ws + p.off_params);
        const unsigned pairs_gx = (unsigned)((p.lev_maxpos[lvl] + kPairsThreads - 1) / kPairsThreads);
        if (lvl >= 1 && !side) {
            // Levels 4..1: the three bands' CNN outputs side by side in the buffer (each a quarter of what level 0 needs of it, or less),
            // then ONE pairs launch for the three bands: 3 + 1 launches per level instead of 3 + 3.
            PairsBands pb;
            for (int band = 0; band < 3; ++band) {
                float *pband = params + band * p.lev_floats[lvl];
                if (int rc = launch_band_params(c, fplanes, g, band, pband, s, gv, d_tiles ? d_tiles + p.run[lvl * 3 + band].off : nullptr, &p.run[lvl * 3 + band])) return rc;
                pb.sv[band] = d_sg + (size_t)(lvl * 3 + band) * B;
                pb.params[band] = pband;
                pb.pairs[band] = pairs + p.pair_base[lvl * 3 + band];
            }
            ProfSpan span(c, PROF_PAIRS, s);
            cdf_pairs_bands_kernel<<<dim3(pairs_gx, B, 3), kPairsThreads, 0, s>>>(planes, pb, mm);
            HIPCHK(hipGetLastError());
            continue;
        }
        for (int band = 0; band < 3; ++band) {
            // (a launch's CNN outputs are read once, by the pairs kernel right behind it; cutting level 0 into sub-batches whose outputs stay in
            // the memory-side cache was measured and bought nothing: profiles/r4/tried_encoder_subbatch.json)
            if (int rc = launch_band_params(c, fplanes, g, band, params, s, gv, d_tiles ? d_tiles + p.run[lvl * 3 + band].off : nullptr, &p.run[lvl * 3 + band])) return rc;
            ProfSpan span(c, PROF_PAIRS, s);
            const StageGeom *sv = d_sg + (size_t)(lvl * 3 + band) * B;
            cdf_pairs_kernel<<<dim3(pairs_gx, B), kPairsThreads, 0, s>>>(planes, params, mm, p.sg[(size_t)(lvl * 3 + band) * B], sv, pairs + p.pair_base[lvl * 3 + band]);
            HIPCHK(hipGetLastError());
        }
    }
    if (side) {
        HIPCHK(hipEventRecord(c->ev_enc[1], s2));
        HIPCHK(hipStreamWaitEvent(s, c->ev_enc[1], 0));
    }
    if (M == 0) {
        ProfSpan span(c, PROF_AC, s);
        const int n_streams = LLICTI_NSTREAMS * B;
        ac_encode_pairs_kernel<<<n_streams, 64, 0, s>>>(pairs, d_desc, n_streams, slots, slot_len, status);
        pack_kernel<<<dim3(LLICTI_NSTREAMS, B), 256, 0, s>>>(slots, pd->dev<long>(p.d_slot_off), slot_len, B, p.img[0].hdr_bytes, d_out, (long)out_stride, d_seg_len, status);
    } else {
        ProfSpan span(c, PROF_RANS_ENC, s);
        int32_t *rinfo = (int32_t *)(ws + p.off_rinfo);
        const StageGeom *sglv = d_sg + (size_t)(0 * 3 + 2) * B;      // the last stage: an xwide stream's seed symbols are read from its pixels
        const int NS = p.nstreams;                                   // the streams of all images (an image's count is its own: ImgGeo::M)
        // LLICTI_MODE_RANS_X_AUTO: each image's stream count is picked here, on the device, from what its last stage costs (a pure function of
        // the image); the table holds the most it may get, the streams it does not get are empty segments
        unsigned long long *ssum = autoM ? (unsigned long long *)(ws + p.off_rpos) : nullptr;      // (the decoder's cursor array: unused by an encode; B x 8 bytes of its >= B x 128)
        if (autoM) choose_streams_kernel<<<dim3(kAutoSlices, B), 256, 0, s>>>(pairs, d_desc, B, ssum);
        if (Q == 4) rans_encode_kernel<4><<<NS, 256, 0, s>>>(pairs, d_desc, B, d_sref, slots, d_rslot_off, p.rslot_cap, rinfo, status, sglv, planes, mm, ssum, d_img);
        else if (Q == 2) rans_encode_kernel<2><<<NS, 128, 0, s>>>(pairs, d_desc, B, d_sref, slots, d_rslot_off, p.rslot_cap, rinfo, status, sglv, planes, mm, ssum, d_img);
        else rans_encode_kernel<1><<<NS, 64, 0, s>>>(pairs, d_desc, B, d_sref, slots, d_rslot_off, p.rslot_cap, rinfo, status, sglv, planes, mm, ssum, d_img);
        rans_pack_kernel<<<NS, 256, 0, s>>>(slots, d_rslot_off, rinfo, d_sref, d_img, d_out, (long)out_stride, d_seg_len, status, ssum, d_desc, B);
    }
    latch_status_kernel<<<1, 64, 0, s>>>(status, c->d_status, nullptr, 0);
    HIPCHK(hipGetLastError());
    return LLICTI_OK;
}

extern "C" int llicti_encode_images_v(llicti_ctx *c, const uint8_t *d_rgb, const size_t *rgb_off, int B, const int *Hs, const int *Ws, int mode,
                                      void *d_workspace, size_t workspace_bytes,
                                      uint8_t *d_out, size_t out_stride, int32_t *d_seg_len, void *stream)
{
    return encode_batch(c, d_rgb, rgb_off, B, Hs, Ws, &mode, 1, d_workspace, workspace_bytes, d_out, out_stride, d_seg_len, stream);
}

extern "C" int llicti_encode_images_vm(llicti_ctx *c, const uint8_t *d_rgb, const size_t *rgb_off, int B, const int *Hs, const int *Ws, const int *modes,
                                       void *d_workspace, size_t workspace_bytes,
                                       uint8_t *d_out, size_t out_stride, int32_t *d_seg_len, void *stream)
{
    return encode_batch(c, d_rgb, rgb_off, B, Hs, Ws, modes, B, d_workspace, workspace_bytes, d_out, out_stride, d_seg_len, stream);
}

extern "C" int llicti_encode_images(llicti_ctx *c, const uint8_t *d_rgb, int B, int H, int W, int mode,
                                    void *d_workspace, size_t workspace_bytes,
                                    uint8_t *d_out, size_t out_stride, int32_t *d_seg_len, void *stream)
{
    if (check_dims(B, H, W)) return LLICTI_EINVAL;
    std::vector<int> Hs(B, H), Ws(B, W);
    return encode_batch(c, d_rgb, nullptr, B, Hs.data(), Ws.data(), &mode, 1, d_workspace, workspace_bytes, d_out, out_stride, d_seg_len, stream);
}

static int decode_stages(llicti_ctx *c, PlanDev *pd, const uint8_t *d_in, size_t in_stride, const int32_t *d_seg_len,
                         uint8_t *ws, uint8_t *d_rgb, hipStream_t s)
{
    const Plan &p = pd->p;
    const int B = p.B, M = p.M, Q = p.Q;
    int16_t *planes = (int16_t *)(ws + p.off_planes);
    float *fplanes = (float *)(ws + p.off_fplanes);
    int32_t *mm = (int32_t *)(ws + p.off_minmax);
    int32_t *status = (int32_t *)(ws + p.off_status);
    float *params = (float *)(ws + p.off_params);
    uint8_t *slots = ws + p.off_slots;
    uint8_t *tables = ws + p.off_tables;
    uint32_t *acstate = (uint32_t *)(ws + p.off_acstate);
    int32_t *slot_len = (int32_t *)(ws + p.off_slot_len);
    const ImgGeo *d_img = pd->dev<ImgGeo>(p.d_img);
    const Geom *d_geo = pd->dev<Geom>(p.d_geo);
    const StageGeom *d_sg = pd->dev<StageGeom>(p.d_sg);
    const long *d_rslot_off = pd->dev<long>(p.d_rslot_off);
    const StreamRef *d_sref = pd->dev<StreamRef>(p.d_sref);
    const int NS = p.nstreams;
    const TileRef *d_tiles = p.uniform ? nullptr : pd->dev<TileRef>(p.d_tiles);

    zero_words_kernel<<<(kStatusHead + B + 255) / 256, 256, 0, s>>>(status, kStatusHead + B);
    uint32_t *rstate = (uint32_t *)(ws + p.off_rstate);
    uint32_t *rpos = (uint32_t *)(ws + p.off_rpos);
    uint32_t *rtail = (uint32_t *)(ws + p.off_rtail);
    {
        ProfSpan span(c, PROF_MISC, s);
        header_read_kernel<<<B, 256, 0, s>>>(d_in, (long)in_stride, d_seg_len, d_img, planes, fplanes, mm, status);
        if (M == 0) {
            unpack_kernel<<<dim3(LLICTI_NSTREAMS, B), 256, 0, s>>>(d_in, (long)in_stride, d_seg_len, B, slots, pd->dev<long>(p.d_slot_off), pd->dev<int32_t>(p.d_slot_cap), slot_len, status);
        } else {
            const StageGeom *sglv = d_sg + (size_t)(0 * 3 + 2) * B;      // the last stage: an xwide v4 tail is at most the stream's share of it
            rans_unpack_kernel<<<NS, 256, 0, s>>>(d_in, (long)in_stride, d_seg_len, d_sref, 2 + Q * RansGeo<1>::kPayBytes,
                                                  slots, d_rslot_off, p.rslot_cap, rpos, status, Q == 4 ? 4 : 2);
            if (Q == 4) rans_init_kernel<4><<<NS, 64, 0, s>>>(slots, d_rslot_off, d_sref, rstate, rpos, rtail, status, sglv);
            else if (Q == 2) rans_init_kernel<2><<<NS, 64, 0, s>>>(slots, d_rslot_off, d_sref, rstate, rpos, rtail, status, sglv);
            else rans_init_kernel<1><<<NS, 64, 0, s>>>(slots, d_rslot_off, d_sref, rstate, rpos, rtail, status, sglv);
        }
    }
    // 45 dependent stages (LLICTI_nets.py:440-498): CNN of band b needs bands < b of this level, Co needs Y, Cg needs Y, Co
    for (int lvl = LLICTI_NLEVELS - 1; lvl >= 0; --lvl) {
        const Geom &g = p.geo[(size_t)lvl * B];                         // image 0's (equal sizes: every image's)
        const Geom *gv = p.uniform ? nullptr : d_geo + (size_t)lvl * B;
        for (int band = 0; band < 3; ++band) {
            if (int rc = launch_band_params(c, fplanes, g, band, params, s, gv, d_tiles ? d_tiles + p.run[lvl * 3 + band].off : nullptr, &p.run[lvl * 3 + band])) return rc;
            const StageGeom *sgv = d_sg + (size_t)(lvl * 3 + band) * B;
            if (M > 0) {
                const int last = (lvl == 0 && band == 2) ? 1 : 0;      // the last stage's tail symbols are decoded by rans_tail_kernel
                {
                ProfSpan span(c, PROF_RANS_STAGE, s);
                if (Q == 4) {
                    rans_decode_stage_lane_kernel<4><<<NS, 256, 0, s>>>(params, sgv, d_sref, slots, d_rslot_off, p.rslot_cap, rstate, rpos, rtail, planes, fplanes, mm, last, status, c->d_phi_lut);
                } else if (Q == 2) {
                    rans_decode_stage_pair_kernel<<<NS, 64 * kRansWaves, 0, s>>>(params, sgv, d_sref, c->d_phi_lut, slots, d_rslot_off, p.rslot_cap, rstate, rpos, rtail, planes, fplanes, mm, last, status);
                } else {
                    rans_decode_stage_kernel<<<NS, 64 * kRansWaves, 0, s>>>(params, sgv, d_sref, c->d_phi_lut, slots, d_rslot_off, p.rslot_cap, rstate, rpos, rtail, planes, fplanes, mm, last, status);
                }
                }
                if (last) {
                    ProfSpan span(c, PROF_RANS_TAIL, s);
                    if (Q == 4) rans_tail_kernel<4><<<NS, 64 * (1 + kTailAhead) * kTailChains<4>, 0, s>>>(params, sgv, d_sref, rstate, rpos, rtail, planes, fplanes, mm, status, slots, d_rslot_off);
                    else if (Q == 2) rans_tail_kernel<2><<<NS, 64 * (1 + kTailAhead), 0, s>>>(params, sgv, d_sref, rstate, rpos, rtail, planes, fplanes, mm, status, slots, d_rslot_off);
                    else rans_tail_kernel<1><<<NS, 64 * (1 + kTailAhead), 0, s>>>(params, sgv, d_sref, rstate, rpos, rtail, planes, fplanes, mm, status, slots, d_rslot_off);
                }
            }
            if (M == 0) {
                // (equal sizes only)  Y, Co, Cg of this band as a pipeline over chunks of the stage: chunk c of Co needs only chunk c
                // of Y (mean update from the SAME pixel, LLICTI_nets.py:474-477), chunk c of Cg only chunk c of
                // Y and Co.  Three HIP streams; tables are built per chunk into one buffer per colour; the coder
                // state of a stream travels between its chunk launches in `acstate`.
                const StageGeom sg = make_stage(g, band);
                const long nc = (long)sg.hc * sg.wc;
                const int C = ac_chunks(nc);
                const long rows = ac_chunk_rows(nc);
                hipStream_t q[3] = { s, c->sub[1], c->sub[2] };
                if (C > 1) {
                    HIPCHK(hipEventRecord(c->ev_ac_band, s));
                    HIPCHK(hipStreamWaitEvent(q[1], c->ev_ac_band, 0));
                    HIPCHK(hipStreamWaitEvent(q[2], c->ev_ac_band, 0));
                }
                for (int ch = 0; ch < C; ++ch) {
                    const long n0 = (long)ch * rows, cnt = std::min(rows, nc - n0);
                    for (int clr = 0; clr < 3; ++clr) {
                        hipStream_t qs = (C > 1) ? q[clr] : s;
                        if (C > 1 && clr > 0) HIPCHK(hipStreamWaitEvent(qs, c->ev_ac[clr - 1][ch], 0));
                        ProfSpan span(c, PROF_AC, qs);
                        const bool anchors = ac_use_anchors(B, c->ac_anchor_min_batch);
                        const int row_stride = (clr == 0) ? 264 : 512;      // full rows: Y has Lp = 257, Co / Cg Lp <= 512
                        uint8_t *tab = tables + (size_t)clr * B * p.ac_cap_rows * (anchors ? (size_t)kAnchorRow : (size_t)1024);
                        if (anchors) { if (int rc = launch_cdf_anchors(planes, params, mm, g, band, clr, tab, n0, cnt, p.ac_cap_rows, qs)) return rc; }
                        else if (int rc = launch_cdf_table(planes, params, mm, g, band, clr, (uint16_t *)tab, row_stride, n0, cnt, p.ac_cap_rows, qs)) return rc;
                        const int st = stage_index(lvl, band, clr);
                        DecOut o;
                        memset(&o, 0, sizeof o);
                        o.planes = planes; o.fplanes = fplanes; o.minmax = mm; o.sg = sg; o.clr = clr;
                        o.len = slot_len + (size_t)st * B;
                        // the B streams of one stage sit in consecutive slots of equal capacity
                        const long in_stride_slots = p.slot_cap[(size_t)st * B];
                        AcChunk ck = { (int)n0, (int)cnt, (int)nc, (int)p.ac_cap_rows, acstate + (size_t)clr * B * 8 };
                        if (anchors) ac_decode_anchor_kernel<<<B, 64, 0, qs>>>(tab, slots + p.slot_off[(size_t)st * B], in_stride_slots, ck, o);
                        else ac_decode_kernel<<<B, 64, 0, qs>>>((const uint16_t *)tab, 0, row_stride, slots + p.slot_off[(size_t)st * B], in_stride_slots, ck, o);
                        if (C > 1 && clr < 2) HIPCHK(hipEventRecord(c->ev_ac[clr][ch], qs));
                    }
                }
                if (C > 1) {
                    for (int k = 1; k < 3; ++k) {
                        HIPCHK(hipEventRecord(c->ev_ac_end[k - 1], q[k]));
                        HIPCHK(hipStreamWaitEvent(s, c->ev_ac_end[k - 1], 0));
                    }
                }
            }
        }
    }
    const int gx = (int)std::min<long>((p.max_plane + 255) / 256, 1024);
    {
        ProfSpan span(c, PROF_MISC, s);
        // (the call's status words are latched into the context's by this kernel: the workspace is the caller's, it may be gone or reused by the
        // time the words are read)
        unlift_kernel<<<dim3(gx, B), 256, 0, s>>>(planes, p.max_plane, d_rgb, status, kStatusHead, c->d_status, c->d_img_status, d_img);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

static int decode_batch(llicti_ctx *c, const uint8_t *d_in, size_t in_stride, const int32_t *d_seg_len, int B, const int *Hs, const int *Ws,
                        const int *modes, int n_modes, void *d_workspace, size_t workspace_bytes, uint8_t *d_rgb, const size_t *rgb_off, void *stream)
{
    if (!c || !d_in || !d_seg_len || !d_workspace || !d_rgb) return fail(LLICTI_EINVAL, "decode_images: null pointer");
    if (check_dims_v(B, Hs, Ws)) return LLICTI_EINVAL;
    int ME = 0;
    std::vector<int> Ms;
    if (int rc = resolve_modes("decode_images", modes, n_modes, B, &ME, Ms)) return rc;
    if (ME & 0x1000) return fail(LLICTI_EINVAL, "decode_images: LLICTI_MODE_RANS_X_AUTO is an encoder's mode -- a container says how many streams it has (header: llicti_header_mode)");
    for (int b = 0; b < 3; ++b) if (!c->have[b]) return fail(LLICTI_ENOWEIGHTS, "band %d weights not set", b);
    DeviceGuard guard(c);
    hipStream_t s = (hipStream_t)stream;
    PlanDev *pd = nullptr;
    if (int rc = get_plan(c, B, Hs, Ws, rgb_off, ME, Ms.empty() ? nullptr : Ms.data(), s, &pd)) return rc;
    PlanUse use{ pd, s };
    const Plan &p = pd->p;
    if (!p.uniform && p.M == 0) return fail(LLICTI_EINVAL, "decode_images: a batch of mixed sizes needs a rANS container (the reference-format container codes equal sizes per call)");
    for (int b = 0; b < B; ++b)
        if (in_stride < (size_t)p.img[b].hdr_bytes)
            return fail(LLICTI_EINVAL, "decode_images: in_stride %zu is smaller than the %d header bytes of a %dx%d image", in_stride, p.img[b].hdr_bytes, Ws[b], Hs[b]);
    if (workspace_bytes < p.total)
        return fail(LLICTI_ENOSPACE, "decode_images: workspace %zu < %zu", workspace_bytes, p.total);
    uint8_t *ws = (uint8_t *)d_workspace;

    c->img_status_n = 0;
    if (c->img_status_cap < B) {          // grows rarely (a larger batch than any before): blocking is fine here
        ++c->n_device_sync; ++c->n_device_alloc;
        HIPCHK(hipDeviceSynchronize());
        if (c->d_img_status) { (void)hipFree(c->d_img_status); c->d_img_status = nullptr; c->img_status_cap = 0; }
        const int cap = std::max(B, 64);
        HIPCHK(hipMalloc(&c->d_img_status, (size_t)cap * sizeof(int32_t)));
        c->img_status_cap = cap;
    }
    CallScope call(c, s);
    if (int rc = decode_stages(c, pd, d_in, in_stride, d_seg_len, ws, d_rgb, s)) return rc;
    c->img_status_n = B;
    return LLICTI_OK;
}

extern "C" int llicti_decode_images_v(llicti_ctx *c, const uint8_t *d_in, size_t in_stride, const int32_t *d_seg_len,
                                      int B, const int *Hs, const int *Ws, int mode, void *d_workspace, size_t workspace_bytes,
                                      uint8_t *d_rgb, const size_t *rgb_off, void *stream)
{
    return decode_batch(c, d_in, in_stride, d_seg_len, B, Hs, Ws, &mode, 1, d_workspace, workspace_bytes, d_rgb, rgb_off, stream);
}

extern "C" int llicti_decode_images_vm(llicti_ctx *c, const uint8_t *d_in, size_t in_stride, const int32_t *d_seg_len,
                                       int B, const int *Hs, const int *Ws, const int *modes, void *d_workspace, size_t workspace_bytes,
                                       uint8_t *d_rgb, const size_t *rgb_off, void *stream)
{
    return decode_batch(c, d_in, in_stride, d_seg_len, B, Hs, Ws, modes, B, d_workspace, workspace_bytes, d_rgb, rgb_off, stream);
}

extern "C" int llicti_decode_images(llicti_ctx *c, const uint8_t *d_in, size_t in_stride, const int32_t *d_seg_len,
                                    int B, int H, int W, int mode, void *d_workspace, size_t workspace_bytes,
                                    uint8_t *d_rgb, void *stream)
{
    if (check_dims(B, H, W)) return LLICTI_EINVAL;
    std::vector<int> Hs(B, H), Ws(B, W);
    return decode_batch(c, d_in, in_stride, d_seg_len, B, Hs.data(), Ws.data(), &mode, 1, d_workspace, workspace_bytes, d_rgb, nullptr, stream);
}

extern "C" int llicti_check_status(llicti_ctx *c, void *stream)
{
    if (!c) return fail(LLICTI_EINVAL, "null ctx");
    DeviceGuard guard(c);
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    int32_t st = 0;
    HIPCHK(hipMemcpy(&st, c->d_status, 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(c->d_status, 0, 4));
    if (st == LLICTI_EFORMAT) return fail(LLICTI_EFORMAT, "malformed container (header does not match the requested shape, or a stream is too long)");
    if (st == LLICTI_ENOSPACE) return fail(LLICTI_ENOSPACE, "output buffer too small for the encoded streams");
    if (st != 0) return fail(st, "device-side status %d", st);
    return LLICTI_OK;
}

extern "C" int llicti_last_timing(llicti_ctx *c, float ms[4], int *n_launch)
{
    if (!c || !ms) return fail(LLICTI_EINVAL, "last_timing: null pointer");
    DeviceGuard guard(c);
    if (c->timing_pending) {
        HIPCHK(hipEventSynchronize(c->ev_call[1]));
        float t = 0;
        HIPCHK(hipEventElapsedTime(&t, c->ev_call[0], c->ev_call[1]));
        c->last_ms[0] = t;
        for (int k = 0; k < PROF_NCAT; ++k) c->last_cat_ms[k] = 0;
        c->last_cnn_ms.clear();
        for (int l = 0; l < LLICTI_NLEVELS; ++l) c->last_cnn_level_ms[l] = 0;
        for (const llicti_ctx::Span &sp : c->spans) {
            float k = 0;
            HIPCHK(hipEventSynchronize(sp.e1));       // spans of the AC decode pipeline sit on the internal streams
            HIPCHK(hipEventElapsedTime(&k, sp.e0, sp.e1));
            c->last_cat_ms[sp.cat] += k;
            if (sp.cat == PROF_CNN) { c->last_cnn_ms.push_back(k); if (sp.tag >= 0 && sp.tag < LLICTI_NLEVELS) c->last_cnn_level_ms[sp.tag] += k; }
        }
        c->last_ms[1] = c->last_cat_ms[PROF_CNN];
        c->last_launches = (int)c->last_cnn_ms.size();
        c->timing_pending = false;
    }
    for (int i = 0; i < 4; ++i) ms[i] = c->last_ms[i];
    if (n_launch) *n_launch = c->last_launches;
    return LLICTI_OK;
}

extern "C" int llicti_last_timing_detail(llicti_ctx *c, float cat_ms[LLICTI_NPROF], float *cnn_launch_ms, int cnn_cap, int *n_cnn)
{
    if (!c || !cat_ms) return fail(LLICTI_EINVAL, "last_timing_detail: null pointer");
    float ms[4];
    if (int rc = llicti_last_timing(c, ms, nullptr)) return rc;
    for (int k = 0; k < LLICTI_NPROF; ++k) cat_ms[k] = c->last_cat_ms[k];
    const int n = (int)c->last_cnn_ms.size();
    if (cnn_launch_ms) for (int i = 0; i < std::min(n, cnn_cap); ++i) cnn_launch_ms[i] = c->last_cnn_ms[i];
    if (n_cnn) *n_cnn = n;
    return LLICTI_OK;
}

extern "C" int llicti_last_cnn_level_ms(llicti_ctx *c, float level_ms[LLICTI_NLEVELS])
{
    if (!c || !level_ms) return fail(LLICTI_EINVAL, "last_cnn_level_ms: null pointer");
    float ms[4];
    if (int rc = llicti_last_timing(c, ms, nullptr)) return rc;
    for (int l = 0; l < LLICTI_NLEVELS; ++l) level_ms[l] = c->last_cnn_level_ms[l];
    return LLICTI_OK;
}

extern "C" int llicti_image_status(llicti_ctx *c, int32_t *h_status, int n, void *stream)
{
    if (!c || !h_status || n < 1) return fail(LLICTI_EINVAL, "image_status: bad argument");
    DeviceGuard guard(c);
    if (!c->d_img_status || n > c->img_status_n) return fail(LLICTI_EINVAL, "image_status: the last decode held %d images ", c->img_status_n);
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    HIPCHK(hipMemcpy(h_status, c->d_img_status, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return LLICTI_OK;
}

extern "C" int llicti_selftest(void)
{
    if (int d = selftest_div_magic()) return fail(LLICTI_EINVAL, "selftest: magic division by %d differs from '/'", d);
    return LLICTI_OK;
}
