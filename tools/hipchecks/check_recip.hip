// Exhaustive check (run on the GPU box): for EVERY float d in [2, 9] the scale-free reciprocal sequence used by
// numerics.hpp::recip_2_9 returns the bits of the IEEE division 1.0f / d (hipcc's v_div_scale / v_div_fmas / v_div_fixup
// expansion, whose scale and fix-up steps are identities on this range).  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
__device__ __forceinline__ float recip_2_9(float d)
{
    float y = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    float q = y;
    float r = __builtin_fmaf(-d, q, 1.0f);
    q = __builtin_fmaf(r, y, q);
    r = __builtin_fmaf(-d, q, 1.0f);
    return __builtin_fmaf(r, y, q);
}
__global__ void check(uint32_t lo, uint32_t hi, unsigned long long *bad, uint32_t *first)
{
    for (uint64_t u = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u <= hi; u += (uint64_t)gridDim.x * blockDim.x) {
        const float d = __uint_as_float((uint32_t)u);
        const float a = 1.0f / d, b = recip_2_9(d);
        if (__float_as_uint(a) != __float_as_uint(b)) { if (atomicAdd(bad, 1ull) == 0) *first = (uint32_t)u; }
    }
}
int main()
{
    float flo = 2.0f, fhi = 9.0f;
    uint32_t lo, hi;
    memcpy(&lo, &flo, 4); memcpy(&hi, &fhi, 4);
    unsigned long long *bad; uint32_t *first;
    hipMalloc(&bad, 8); hipMalloc(&first, 4); hipMemset(bad, 0, 8); hipMemset(first, 0, 4);
    check<<<4096, 256>>>(lo, hi, bad, first);
    unsigned long long hb = 0; uint32_t hf = 0;
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
    printf("recip_2_9: %llu floats in [2, 9] checked, %llu mismatches (first 0x%08x)\n", (unsigned long long)(hi - lo + 1), hb, hf);
    return hb ? 1 : 0;
}
