// Operand / result layout of v_mfma_f32_4x4x1_16B_f32 on gfx950, and that it is one fmaf per element (run on the GPU box).
// 16 independent blocks; block b: D_b[4x4] = A_b[4x1] * B_b[1x4] + C_b.  Expected: A[i] in lane 4b + i, B[j] in lane 4b + j,
// D[i][j] in lane 4b + j, register i.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *a, const float *b, const float *c, float *d)
{
    const int l = threadIdx.x;
    f32x4 acc = { c[l * 4 + 0], c[l * 4 + 1], c[l * 4 + 2], c[l * 4 + 3] };
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[l * 4 + i] = acc[i];
}
int main()
{
    float ha[64], hb[64], hc[256], hd[256], *a, *b, *c, *d;
    for (int l = 0; l < 64; ++l) { ha[l] = 1.0f + 0.37f * l + 1e-3f * l * l; hb[l] = -2.0f + 0.11f * l; }
    for (int l = 0; l < 256; ++l) hc[l] = 0.001f * l - 0.1f;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&c, 1024); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice); hipMemcpy(c, hc, 1024, hipMemcpyHostToDevice);
    k<<<1, 64>>>(a, b, c, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int blk = 0; blk < 16; ++blk)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                const float want = fmaf(ha[4 * blk + i], hb[4 * blk + j], hc[(4 * blk + j) * 4 + i]);
                const float got = hd[(4 * blk + j) * 4 + i];
                if (want != got) { if (bad < 5) printf("block %d i %d j %d: want %.9g got %.9g\n", blk, i, j, want, got); ++bad; }
            }
    printf("mfma_f32_4x4x1: %d mismatches against the expected layout / fmaf\n", bad);
    return bad != 0;
}
