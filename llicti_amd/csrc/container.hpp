// container.hpp -- container assembly kernels: header write / read, stream pack / unpack.
// Part of the single translation unit llicti_hip.hip (included in order; not a stand-alone header).
#pragma once

// ------------------------------------------------------------------------------------------------ container kernels
// encode: header segments straight into the container; seg_len[b][0..3]
__global__ void header_write_kernel(const uint8_t *__restrict__ rgb, const int32_t *__restrict__ minmax, const ImgGeo *__restrict__ iv,
                                    uint8_t *__restrict__ out, long out_stride, int32_t *__restrict__ seg_len, unsigned long long *__restrict__ ssum)
{
    const int b = blockIdx.x;
    if (ssum && threadIdx.x == 0) ssum[b] = 0ull;       // "auto" encodes: the image's last-stage cost is summed into it later in the call (choose_streams_kernel)
    uint8_t *o = out + (long)b * out_stride;
    const ImgGeo ig = iv[b];
    const int byte0 = ig.byte0;
    const int W = ig.W, h4 = ig.h4, w4 = ig.w4, padint = ig.padint;
    const long plane = ig.plane;
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
        o[17 + t] = rgb[ig.rgb_off + c * plane + (long)(32 * i) * W + 32 * j];
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
    block_copy_bytes(o, src, n);
    if (threadIdx.x == 0) seg_len[(long)b * LLICTI_NSEG + 4 + st] = n;
}

// decode: parse + validate header, min/max -> minmax[b][4], DC band -> planes at stride 32
__global__ void header_read_kernel(const uint8_t *__restrict__ in, long in_stride, const int32_t *__restrict__ seg_len,
                                   const ImgGeo *__restrict__ iv, int16_t *__restrict__ planes,
                                   float *__restrict__ fplanes, int32_t *__restrict__ minmax, int32_t *status)
{
    const int b = blockIdx.x;
    const uint8_t *p = in + (long)b * in_stride;
    const int32_t *sl = seg_len + (long)b * LLICTI_NSEG;
    const ImgGeo ig = iv[b];
    const int byte0 = ig.byte0;
    const int W = ig.W, h4 = ig.h4, w4 = ig.w4, padint = ig.padint;
    const long plane = ig.plane;
    __shared__ int ok;
    __shared__ int sh_bad;
    __shared__ unsigned long long sh_tot;
    if (threadIdx.x == 0) { sh_bad = 0; sh_tot = 0ull; }
    __syncthreads();
    if (threadIdx.x < LLICTI_NSEG) {               // the 49 lengths in parallel (one thread walking them was most of this kernel's 70 us)
        const int v = sl[threadIdx.x];
        if (v < 0 || v > in_stride) atomicOr(&sh_bad, 1);
        atomicAdd(&sh_tot, (unsigned long long)((v < 0) ? 0 : v));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // every segment length is checked before a single container byte is read: a malformed seg_len must end in
        // LLICTI_EFORMAT, never in an access outside [in, in + in_stride)
        const long tot = (long)sh_tot;
        bool segs_ok = in_stride >= 17 + 3L * h4 * w4 && !sh_bad;
        if (tot > in_stride) segs_ok = false;
        ok = segs_ok && (sl[0] == 3 && sl[1] == 12 && sl[2] == 2 && sl[3] == 3 * h4 * w4);
        if (ok) {
            const int pad = (int)(uint16_t)(p[15] | (p[16] << 8));      // (xwide v4 containers carry their stream count in bits 10 .. 15: ig.padint has them)
            ok = (p[0] == byte0 && p[1] == h4 && p[2] == w4 && pad == padint);           // LLICTI_nets.py:423-428
        }
        if (!ok) flag_image(status, b, LLICTI_EFORMAT);
        int16_t v[6] = { 0, -255, -255, 255, 255, 255 };
        if (ok) for (int k = 0; k < 6; ++k) v[k] = (int16_t)(p[3 + 2 * k] | (p[4 + 2 * k] << 8));
        int32_t *mm = minmax + 4 * b;
        mm[0] = v[1]; mm[1] = v[2]; mm[2] = v[4]; mm[3] = v[5];
        if (v[1] > v[4] || v[2] > v[5] || v[1] < -255 || v[2] < -255 || v[4] > 255 || v[5] > 255) {
            flag_image(status, b, LLICTI_EFORMAT);
            mm[0] = mm[1] = -255; mm[2] = mm[3] = 255;
        }
    }
    __syncthreads();
    // a rejected header still gets a DC band (mid grey): the 45 stages run for every image of the batch, and what they write
    // must not depend on what an earlier call left in the shared workspace (the image is flagged: llicti_image_status)
    const uint8_t *dc = p + 17;
    for (int t = threadIdx.x; t < h4 * w4; t += blockDim.x) {                              // :429-430, :443-444
        const int i = t / w4, j = t - i * w4;
        const int R = ok ? dc[t] : 128, G = ok ? dc[h4 * w4 + t] : 128, Bl = ok ? dc[2 * h4 * w4 + t] : 128;
        const int Co = R - Bl, tt = Bl + (Co >> 1), Cg = G - tt, Y = tt + (Cg >> 1) - 127;
        const long off = ig.pix_off + (long)(32 * i) * W + 32 * j;
        planes[off] = (int16_t)Y; planes[off + plane] = (int16_t)Co; planes[off + 2 * plane] = (int16_t)Cg;
        fplanes[off] = (float)Y / 255.0f; fplanes[off + plane] = (float)Co / 255.0f; fplanes[off + 2 * plane] = (float)Cg / 255.0f;
    }
}

// decode: copy stream st of image b into its 4-byte aligned, zero padded slot
__global__ __launch_bounds__(256) void unpack_kernel(const uint8_t *__restrict__ in, long in_stride, const int32_t *__restrict__ seg_len,
                                                     int B, uint8_t *__restrict__ slots, const long *__restrict__ slot_off,
                                                     const int32_t *__restrict__ slot_cap, int32_t *__restrict__ slot_len, int32_t *status)
{
    const int st = blockIdx.x, b = blockIdx.y;
    const int32_t *sl = seg_len + (long)b * LLICTI_NSEG;
    long src = 0;
    bool bad = false;
    for (int k = 0; k < 4 + st; ++k) {          // a negative or oversized EARLIER entry must not move src outside the container
        const int v = sl[k];
        if (v < 0 || v > in_stride) bad = true;
        src += v;
        if (src < 0 || src > in_stride) { bad = true; src = 0; }
    }
    int n = sl[4 + st];
    const int cap = slot_cap[(long)st * B + b];
    if (bad || n < 0 || n + 16 > cap || src + n > in_stride) { if (threadIdx.x == 0) flag_image(status, b, LLICTI_EFORMAT); n = 0; src = 0; }
    const uint8_t *p = in + (long)b * in_stride + src;
    uint8_t *o = slots + slot_off[(long)st * B + b];
    block_copy_bytes(o, p, n);
    const int padded = min(cap, ((n + 3) & ~3) + 16);
    for (int t = n + threadIdx.x; t < padded; t += blockDim.x) o[t] = 0;
    if (threadIdx.x == 0) slot_len[(long)st * B + b] = n;       // the decoders read nothing past it
}
