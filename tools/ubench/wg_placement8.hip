// wg_placement8.hip -- which waves of a LARGE workgroup share a SIMD?  (256 / 512 threads per workgroup, <= 256 registers per lane, LDS as given.)
// Question behind it: can the two cooperating waves of one 64-env group be put on the SAME SIMD (they alternate when their constraint sets are coupled)?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/wg_placement8.hip -o build_exp/wg_placement8 && build_exp/wg_placement8 256 163840 8
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void probe(unsigned *out, int spin, float *sink) {
    extern __shared__ float lds[];
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float acc[180];
#pragma unroll
    for (int i = 0; i < 180; i++) acc[i] = threadIdx.x * 1e-3f + i;
    long long t0 = clock64();
    while (clock64() - t0 < spin) {
#pragma unroll
        for (int i = 0; i < 180; i++) acc[i] = fmaf(acc[i], 1.0001f, 0.5f);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 180; i++) s += acc[i];
    lds[threadIdx.x] = s;
    __syncthreads();
    if (s == 12345.678f) sink[0] = lds[(threadIdx.x + 1) & 127];
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * NW + (threadIdx.x >> 6);
        out[2 * w] = hwid;
        out[2 * w + 1] = xcc;
    }
}

int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 256;
    const int ldsb = argc > 2 ? atoi(argv[2]) : 163840;
    const int NW = argc > 3 ? atoi(argv[3]) : 8;
    unsigned *d; float *sink;
    CHECK(hipMalloc(&d, sizeof(unsigned) * 2 * NW * G));
    CHECK(hipMalloc(&sink, 16));
    if (NW == 8) { CHECK(hipFuncSetAttribute((const void *)probe<8>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb)); hipLaunchKernelGGL(probe<8>, dim3(G), dim3(512), ldsb, 0, d, 400000, sink); }
    else { CHECK(hipFuncSetAttribute((const void *)probe<4>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb)); hipLaunchKernelGGL(probe<4>, dim3(G), dim3(256), ldsb, 0, d, 400000, sink); }
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned> h(2 * NW * G);
    CHECK(hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * NW * G, hipMemcpyDeviceToHost));
    std::map<std::vector<int>, int> pattern;   // SIMD of wave 0..NW-1 -> count
    int one_cu = 0;
    for (int g = 0; g < G; g++) {
        std::vector<int> p;
        bool same = true;
        for (int w = 0; w < NW; w++) {
            unsigned a = h[2 * (g * NW + w)], ax = h[2 * (g * NW + w) + 1] & 15;
            p.push_back((a >> 4) & 3);
            same = same && ((a >> 8) & 0xff) == ((h[2 * g * NW] >> 8) & 0xff) && ax == (h[2 * g * NW + 1] & 15);
        }
        one_cu += same;
        pattern[p]++;
    }
    printf("workgroups %d x %d waves, LDS %d B: all waves on one CU in %d workgroups\n", G, NW, ldsb, one_cu);
    for (auto &kv : pattern) {
        printf("  SIMD of wave 0..%d = (", NW - 1);
        for (int v : kv.first) printf("%d ", v);
        printf("): %d workgroups\n", kv.second);
    }
    return 0;
}
