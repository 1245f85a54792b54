// wg_placement.hip -- where do the two waves of a 128-thread workgroup land?  Probe behind the two-cooperating-waves step kernel
// (lcr_kernels2.hip): launches G workgroups of 2 waves with the step kernel's resource shape (LDS bytes, registers per lane), keeps them
// resident for a while and records (XCC, SE, CU, SIMD, wave slot) of every wave from HW_REG_HW_ID / HW_REG_XCC_ID.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/wg_placement.hip -o gpurun_out/wg_placement && gpurun_out/wg_placement 1024 36864 2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int OCC>
__global__ __launch_bounds__(128, OCC) void probe(unsigned *out, int spin, float *sink) {
    extern __shared__ float lds[];
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // burn registers so that the allocation matches the step kernel's (<= 256 for OCC 2, > 256 for OCC 1)
    float acc[OCC == 1 ? 300 : 180];
#pragma unroll
    for (int i = 0; i < (OCC == 1 ? 300 : 180); i++) acc[i] = threadIdx.x * 1e-3f + i;
    long long t0 = clock64();
    while (clock64() - t0 < spin) {
#pragma unroll
        for (int i = 0; i < (OCC == 1 ? 300 : 180); i++) acc[i] = fmaf(acc[i], 1.0001f, 0.5f);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < (OCC == 1 ? 300 : 180); i++) s += acc[i];
    lds[threadIdx.x] = s;
    __syncthreads();
    if (s == 12345.678f) sink[0] = lds[(threadIdx.x + 1) & 127];
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 2 + (threadIdx.x >> 6);
        out[2 * w] = hwid;
        out[2 * w + 1] = xcc;
    }
}

int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 1024;
    const int ldsb = argc > 2 ? atoi(argv[2]) : 36864;
    const int occ = argc > 3 ? atoi(argv[3]) : 2;
    unsigned *d; float *sink;
    CHECK(hipMalloc(&d, sizeof(unsigned) * 4 * G));
    CHECK(hipMalloc(&sink, 16));
    if (occ == 1) { CHECK(hipFuncSetAttribute((const void *)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb)); hipLaunchKernelGGL(probe<1>, dim3(G), dim3(128), ldsb, 0, d, 400000, sink); }
    else { CHECK(hipFuncSetAttribute((const void *)probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb)); hipLaunchKernelGGL(probe<2>, dim3(G), dim3(128), ldsb, 0, d, 400000, sink); }
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned> h(4 * G);
    CHECK(hipMemcpy(h.data(), d, sizeof(unsigned) * 4 * G, hipMemcpyDeviceToHost));
    // HW_ID (gfx9): wave slot [3:0], SIMD [5:4], pipe [7:6], CU [11:8], SH [12], SE [15:13]; XCC_ID [3:0]
    std::map<unsigned long long, std::vector<int>> per_simd;   // key (xcc, se, sh, cu, simd) -> list of (wg*2+wave)
    int same_simd = 0, same_cu = 0;
    std::map<int, int> pair_hist;   // (simd of wave 0) * 4 + (simd of wave 1)
    for (int g = 0; g < G; g++) {
        unsigned a = h[4 * g], ax = h[4 * g + 1] & 15, b = h[4 * g + 2], bx = h[4 * g + 3] & 15;
        auto cu = [](unsigned v, unsigned x) { return ((unsigned long long)x << 16) | ((v >> 8) & 0xff); };
        int sa = (a >> 4) & 3, sb = (b >> 4) & 3;
        same_cu += cu(a, ax) == cu(b, bx);
        same_simd += (cu(a, ax) == cu(b, bx)) && sa == sb;
        pair_hist[sa * 4 + sb]++;
        per_simd[(cu(a, ax) << 2) | sa].push_back(2 * g);
        per_simd[(cu(b, bx) << 2) | sb].push_back(2 * g + 1);
        if (g < 12) printf("wg %4d: wave0 xcc %u se %u cu %2u simd %d slot %u | wave1 xcc %u se %u cu %2u simd %d slot %u\n", g, ax, (a >> 13) & 7, (a >> 8) & 15, sa, a & 15,
                           bx, (b >> 13) & 7, (b >> 8) & 15, sb, b & 15);
    }
    printf("workgroups %d, LDS %d B, occ %d: both waves on one CU %d, on one SIMD %d\n", G, ldsb, occ, same_cu, same_simd);
    for (auto &kv : pair_hist) printf("  (simd of wave0, simd of wave1) = (%d, %d): %d workgroups\n", kv.first / 4, kv.first % 4, kv.second);
    std::map<int, int> occ_hist, role_hist;   // waves per SIMD; per SIMD: number of wave-0s ("arm" role if role = wave index)
    std::map<int, int> slotrule_hist;
    for (auto &kv : per_simd) {
        occ_hist[(int)kv.second.size()]++;
        int w0 = 0;
        for (int w : kv.second) w0 += (w & 1) == 0;
        role_hist[w0 * 10 + (int)kv.second.size()]++;
    }
    printf("SIMDs in use %zu\n", per_simd.size());
    for (auto &kv : occ_hist) printf("  SIMDs hosting %d waves: %d\n", kv.first, kv.second);
    for (auto &kv : role_hist) printf("  SIMDs hosting %d waves of which %d are wave 0 of their workgroup: %d\n", kv.first % 10, kv.first / 10, kv.second);
    // candidate rule: role A = the wave with even (rank(simd) + slot), rank = position in the cyclic order 0,2,1,3; tie -> wave 0
    const int rank[4] = {0, 2, 1, 3};
    std::map<unsigned long long, int> armcount;
    int ties = 0;
    for (int g = 0; g < G; g++) {
        unsigned a = h[4 * g], ax = h[4 * g + 1] & 15, b = h[4 * g + 2], bx = h[4 * g + 3] & 15;
        auto key = [](unsigned v, unsigned x) { return ((((unsigned long long)x << 16) | ((v >> 8) & 0xff)) << 2) | ((v >> 4) & 3); };
        int pa = (rank[(a >> 4) & 3] + (a & 15)) & 1, pb = (rank[(b >> 4) & 3] + (b & 15)) & 1;
        bool a_is_arm = pa == pb ? true : pa == 0;
        ties += pa == pb;
        armcount[a_is_arm ? key(a, ax) : key(b, bx)]++;
    }
    std::map<int, int> ah;
    for (auto &kv : per_simd) ah[armcount.count(kv.first) ? armcount[kv.first] : 0]++;
    printf("rule (rank + slot) parity: ties %d;", ties);
    for (auto &kv : ah) printf("  SIMDs with %d arm waves: %d;", kv.first, kv.second);
    printf("\n");
    return 0;
}
