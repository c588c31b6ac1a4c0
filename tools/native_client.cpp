// native_client.cpp -- libllicti_hip.so from a host program with no Python and no PyTorch in the process: the C-ABI of
// include/llicti_hip.h on plain hipMalloc buffers (the drop-in boundary is a C library; torch is only one possible allocator).
//
//   hipcc -O2 -std=c++17 -I include tools/native_client.cpp -L llicti_amd -lllicti_hip -Wl,-rpath,$PWD/llicti_amd -o native_client
//   ./native_client [B] [H] [W] [streams]        (defaults: 4 images of 96 x 160, container = rANS v3 with 3 xwide streams per image)
//
// What it does: seeded pseudo-random weights of the reference's shapes (the round trip is lossless whatever the weights are) and
// images, llicti_encode_images -> llicti_decode_images on a workspace overwritten in between, byte comparison of the decoded
// pixels, the same through the reference-format container, and one deliberately corrupted container (the call must report
// LLICTI_EFORMAT for that image and decode the others).  Exit code 0 = all of it held.  Built and run by
// tests/test_hip_parity.py::test_native_client_without_torch.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "llicti_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define LL_OK(x) do { int r_ = (x); if (r_ != 0) { std::fprintf(stderr, "%s = %d: %s\n", #x, r_, llicti_last_error()); return 3; } } while (0)

static uint32_t g_state = 12345u;
static float frand() { g_state = g_state * 1664525u + 1013904223u; return (float)((g_state >> 8) & 0xFFFF) / 65536.0f - 0.5f; }

int main(int argc, char **argv)
{
    const int B = argc > 1 ? std::atoi(argv[1]) : 4, H = argc > 2 ? std::atoi(argv[2]) : 96, W = argc > 3 ? std::atoi(argv[3]) : 160;
    const int M = argc > 4 ? std::atoi(argv[4]) : 3;
    const int mode_rans = 0x500 | M;                     // 0x100 + 0x200 * 2 (xwide: 256 lanes) | streams per image
    llicti_ctx *ctx = nullptr;
    LL_OK(llicti_create(&ctx, 0));
    std::printf("%s\n", llicti_version());
    // weights: the canonical packed form of include/llicti_hip.h, small enough that the mixtures stay sane
    const int K0[3] = { 48, 72, 120 };
    for (int band = 0; band < 3; ++band) {
        std::vector<float> w0(352 * K0[band]), b0(352), w1(352 * 88), b1(352), w2(60 * 88), b2(60);
        for (auto &v : w0) v = 0.2f * frand();
        for (auto &v : b0) v = 0.1f * frand();
        for (auto &v : w1) v = 0.2f * frand();
        for (auto &v : b1) v = 0.1f * frand();
        for (auto &v : w2) v = 0.05f * frand();
        for (auto &v : b2) v = 0.05f + 0.05f * (frand() + 0.5f);
        LL_OK(llicti_set_band_weights(ctx, band, K0[band], w0.data(), b0.data(), w1.data(), b1.data(), w2.data(), b2.data()));
    }
    const size_t npix = (size_t)B * 3 * H * W;
    std::vector<uint8_t> rgb(npix);
    for (size_t i = 0; i < npix; ++i) { g_state = g_state * 1664525u + 1013904223u; rgb[i] = (uint8_t)(((i / W) % 64) * 2 + ((g_state >> 24) & 31)); }      // a ramp with noise on top
    const size_t stride = llicti_max_container_bytes(H, W);
    if (stride == 0) { std::fprintf(stderr, "unsupported size\n"); return 3; }
    uint8_t *d_rgb = nullptr, *d_out = nullptr, *d_rec = nullptr, *d_ws = nullptr;
    int32_t *d_seg = nullptr;
    HIP_OK(hipMalloc(&d_rgb, npix));
    HIP_OK(hipMalloc(&d_rec, npix));
    HIP_OK(hipMalloc(&d_out, (size_t)B * stride));
    HIP_OK(hipMalloc(&d_seg, (size_t)B * 49 * sizeof(int32_t)));
    HIP_OK(hipMemcpy(d_rgb, rgb.data(), npix, hipMemcpyHostToDevice));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    std::vector<uint8_t> rec(npix);
    int failures = 0;
    for (int mode : { mode_rans, 0 }) {                  // the throughput container, then the reference's format
        const size_t ws_bytes = llicti_workspace_bytes(B, H, W, mode);
        if (ws_bytes == 0) { std::fprintf(stderr, "workspace size: %s\n", llicti_last_error()); return 3; }
        HIP_OK(hipMalloc(&d_ws, ws_bytes));
        HIP_OK(hipMemsetAsync(d_seg, 0, (size_t)B * 49 * sizeof(int32_t), st));
        LL_OK(llicti_encode_images(ctx, d_rgb, B, H, W, mode, d_ws, ws_bytes, d_out, stride, d_seg, st));
        LL_OK(llicti_check_status(ctx, st));
        HIP_OK(hipMemsetAsync(d_ws, 0xA5, ws_bytes, st));                       // the decoder must not find the encoder's planes
        HIP_OK(hipMemsetAsync(d_rec, 0, npix, st));
        LL_OK(llicti_decode_images(ctx, d_out, stride, d_seg, B, H, W, mode, d_ws, ws_bytes, d_rec, st));
        LL_OK(llicti_check_status(ctx, st));
        HIP_OK(hipMemcpy(rec.data(), d_rec, npix, hipMemcpyDeviceToHost));
        std::vector<int32_t> seg((size_t)B * 49);
        HIP_OK(hipMemcpy(seg.data(), d_seg, seg.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        long bytes = 0;
        for (int32_t v : seg) bytes += v;
        const bool ok = std::memcmp(rec.data(), rgb.data(), npix) == 0;
        std::printf("mode 0x%03x: %d x %dx%d, %ld bytes (%.3f bpp), lossless: %s\n", mode, B, W, H, bytes, 8.0 * bytes / ((double)B * H * W), ok ? "yes" : "NO");
        failures += ok ? 0 : 1;
        if (mode != 0 && B >= 2) {
            // one corrupted container: a byte of image 1's first stream flipped -> LLICTI_EFORMAT, image 1 named, the others intact
            const long off = seg[49 + 0] + seg[49 + 1] + seg[49 + 2] + seg[49 + 3] + 40;
            uint8_t byte;
            HIP_OK(hipMemcpy(&byte, d_out + stride + off, 1, hipMemcpyDeviceToHost));
            byte ^= 0x10;
            HIP_OK(hipMemcpy(d_out + stride + off, &byte, 1, hipMemcpyHostToDevice));
            LL_OK(llicti_decode_images(ctx, d_out, stride, d_seg, B, H, W, mode, d_ws, ws_bytes, d_rec, st));
            const int rc = llicti_check_status(ctx, st);
            std::vector<int32_t> status(B);
            LL_OK(llicti_image_status(ctx, status.data(), B, st));
            HIP_OK(hipMemcpy(rec.data(), d_rec, npix, hipMemcpyDeviceToHost));
            const size_t per = (size_t)3 * H * W;
            bool others = std::memcmp(rec.data(), rgb.data(), per) == 0;
            for (int b = 2; b < B; ++b) others = others && std::memcmp(rec.data() + b * per, rgb.data() + b * per, per) == 0;
            bool named = rc == LLICTI_EFORMAT && status[1] == LLICTI_EFORMAT;
            for (int b = 0; b < B; ++b) if (b != 1) named = named && status[b] == 0;
            std::printf("corrupted container of image 1: call reports %d, per-image status names it: %s, other images intact: %s\n", rc, named ? "yes" : "NO", others ? "yes" : "NO");
            failures += (named && others) ? 0 : 1;
        }
        HIP_OK(hipFree(d_ws));
        d_ws = nullptr;
    }
    LL_OK(llicti_destroy(ctx));
    HIP_OK(hipStreamDestroy(st));
    (void)hipFree(d_rgb); (void)hipFree(d_rec); (void)hipFree(d_out); (void)hipFree(d_seg);
    std::printf(failures ? "FAILED\n" : "native client ok\n");
    return failures ? 1 : 0;
}
