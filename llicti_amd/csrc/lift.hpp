// lift.hpp -- integer YCoCg-R lift / unlift, per-image min/max, float planes (K1-K3, K13).
// Part of the single translation unit llicti_hip.hip (included in order; not a stand-alone header).
#pragma once

// ------------------------------------------------------------------------------------------------ lift
// Per-image min / max of Co and Cg in two steps without atomics: every workgroup of lift_kernel leaves its four
// partial values in part[b][blockIdx.x][4], one small workgroup per image folds them.  (Thousands of atomicMin /
// atomicMax on the same 16 bytes of an image serialise at the memory side: 150 of the kernel's 180 us.)

__global__ __launch_bounds__(64) void minmax_reduce_kernel(const int32_t *__restrict__ part, int gx, int32_t *__restrict__ mm)
{
    const int b = blockIdx.x;
    int mnCo = 32767, mnCg = 32767, mxCo = -32768, mxCg = -32768;
    for (int t = threadIdx.x; t < gx; t += 64) {
        const int32_t *q = part + ((long)b * gx + t) * 4;
        mnCo = min(mnCo, q[0]); mnCg = min(mnCg, q[1]); mxCo = max(mxCo, q[2]); mxCg = max(mxCg, q[3]);
    }
    for (int o = 32; o > 0; o >>= 1) {
        mnCo = min(mnCo, __shfl_xor(mnCo, o)); mxCo = max(mxCo, __shfl_xor(mxCo, o));
        mnCg = min(mnCg, __shfl_xor(mnCg, o)); mxCg = max(mxCg, __shfl_xor(mxCg, o));
    }
    if (threadIdx.x == 0) { mm[4 * b + 0] = mnCo; mm[4 * b + 1] = mnCg; mm[4 * b + 2] = mxCo; mm[4 * b + 3] = mxCg; }
}

// 4 pixels per thread when the plane size allows 4-byte aligned uchar4 / short4 / float4 accesses
template <int VEC>
// zero != nullptr (the whole-batch encode, whose first kernel this is): the call's n_zero status words are cleared here, before any kernel that
// can set them is launched, instead of by a launch of its own.
// iv != nullptr (whole-batch calls): image b's size and placement come from the call's table (the images may differ in size); else B
// images of `plane` pixels each, tightly packed in all three arrays.
__global__ __launch_bounds__(256) void lift_kernel(const uint8_t *__restrict__ rgb, long plane, int16_t *__restrict__ planes,
                                                   float *__restrict__ fplanes, int32_t *__restrict__ part, int32_t *__restrict__ zero, int n_zero,
                                                   const ImgGeo *__restrict__ iv)
{
    const int b = blockIdx.y;
    if (zero && blockIdx.x == 0 && b == 0)
        for (int i = threadIdx.x; i < n_zero; i += blockDim.x) zero[i] = 0;
    long src_off = (long)b * 3 * plane, dst_off = src_off;
    if (iv) { plane = iv[b].plane; src_off = iv[b].rgb_off; dst_off = iv[b].pix_off; }
    const uint8_t *src = rgb + src_off;
    int16_t *dst = planes + dst_off;
    float *fdst = fplanes + dst_off;
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
        part[((long)b * gridDim.x + blockIdx.x) * 4 + k] = v;
    }
}

// status != nullptr (the whole-batch decode, whose last kernel this is): the call's status words are latched into the context's here -- every
// kernel that can set them has finished -- instead of by a launch of its own: word 0 if set, and image b's word by the image's first block.
__global__ __launch_bounds__(256) void unlift_kernel(const int16_t *__restrict__ planes, long plane, uint8_t *__restrict__ rgb,
                                                     const int32_t *__restrict__ status, int status_head, int32_t *__restrict__ latched,
                                                     int32_t *__restrict__ img_latched, const ImgGeo *__restrict__ iv)
{
    const int b = blockIdx.y;
    if (status && blockIdx.x == 0 && threadIdx.x == 0) {
        if (b == 0 && status[0] != 0) *latched = status[0];
        if (img_latched) img_latched[b] = status[status_head + b];
    }
    long src_off = (long)b * 3 * plane, dst_off = src_off;
    if (iv) { plane = iv[b].plane; src_off = iv[b].pix_off; dst_off = iv[b].rgb_off; }      // (lift_kernel: the call's per-image table)
    const int16_t *src = planes + src_off;
    uint8_t *dst = rgb + dst_off;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < plane; p += (long)gridDim.x * blockDim.x) {
        const int Y = src[p] + 127, Co = src[plane + p], Cg = src[2 * plane + p];
        const int t = Y - (Cg >> 1);
        const int G = Cg + t;
        const int Bl = t - (Co >> 1);
        const int R = Bl + Co;
        dst[p] = (uint8_t)R; dst[plane + p] = (uint8_t)G; dst[2 * plane + p] = (uint8_t)Bl;
    }
}
