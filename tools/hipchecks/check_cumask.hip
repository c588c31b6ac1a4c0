// Which compute units does a hipExtStreamCreateWithCUMask stream run on (MI355X, 8 XCDs x 32 CUs)?  Each workgroup records
// XCC_ID and HW_ID (SE / CU) and spins ~20 us so that the grid spreads over every CU it may use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <map>
#include <vector>
__global__ void probe(uint32_t *out)
{
    if (threadIdx.x == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);      // XCC_ID[3:0]
        const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);       // HW_ID
        out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw;
    }
    const long t0 = clock64();
    while (clock64() - t0 < 2000) { }
}
static void run(const char *name, const uint32_t *mask)
{
    hipStream_t s;
    if (mask) { if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("%s: create failed\n", name); return; } }
    else hipStreamCreate(&s);
    const int n = 4096;
    uint32_t *d; hipMalloc(&d, n * 8);
    probe<<<n, 64, 0, s>>>(d);
    hipStreamSynchronize(s);
    std::vector<uint32_t> h(2 * n);
    hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    std::map<uint32_t, std::set<uint32_t>> per;
    for (int i = 0; i < n; ++i) per[h[2 * i] & 15].insert(h[2 * i + 1] & 0xFFFFFF00u ? ((h[2 * i + 1] >> 8) & 0xFFF) : h[2 * i + 1]);
    printf("%-28s", name);
    int tot = 0;
    for (auto &kv : per) { printf(" xcc%u:%zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
    printf("  total %d\n", tot);
    hipFree(d); hipStreamDestroy(s);
}
int main()
{
    uint32_t m[8];
    run("no mask", nullptr);
    for (int i = 0; i < 8; ++i) m[i] = 0; m[0] = 0xFFFFFFFFu; m[1] = 0xFFFFFFFFu; run("bits 0..63", m);
    for (int i = 0; i < 8; ++i) m[i] = 0xFFFFFFFFu; m[0] = 0; m[1] = 0; run("bits 64..255", m);
    for (int i = 0; i < 8; ++i) m[i] = 0; m[0] = 0xFFFFFFFFu; run("bits 0..31", m);
    for (int i = 0; i < 8; ++i) m[i] = 0x11111111u; run("every 4th bit", m);
    for (int i = 0; i < 8; ++i) m[i] = 0x000000FFu; run("bits 32k..32k+7", m);
    for (int i = 0; i < 8; ++i) m[i] = 0; m[0] = 0xFFu; run("bits 0..7", m);
    return 0;
}
