// write_bw.hip -- ceiling of the frame kernel: how fast can a gfx950 write 15.1 GB of frames that are (mostly) copies of a
// 460 800-B L2-resident background?  Variants: store flavour, bytes per workgroup, waves per workgroup, loads or no loads.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/write_bw.hip -o build_exp/write_bw && build_exp/write_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// one workgroup per "env": copies VEC vectors (16 B each) from src (same for every env) to dst + env*VEC
template <int NT, bool LOAD, int NTMP>
__global__ __launch_bounds__(NT) void k_copy(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, int vec_per_env) {
    u32x4 *d = dst + (size_t)blockIdx.x * vec_per_env;
    u32x4 v = {blockIdx.x, 1u, 2u, 3u};
    for (int i = threadIdx.x; i < vec_per_env; i += NT) {
        if (LOAD) v = src[i];
        if (NTMP == 1) __builtin_nontemporal_store(v, d + i);
        else d[i] = v;
    }
}
// 4x unrolled: four loads in flight, then four stores
template <int NT, int NTMP>
__global__ __launch_bounds__(NT) void k_copy4(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, int vec_per_env) {
    u32x4 *d = dst + (size_t)blockIdx.x * vec_per_env;
    int i = threadIdx.x;
    for (; i + 3 * NT < vec_per_env; i += 4 * NT) {
        u32x4 a = src[i], b = src[i + NT], c = src[i + 2 * NT], e = src[i + 3 * NT];
        if (NTMP) { __builtin_nontemporal_store(a, d + i); __builtin_nontemporal_store(b, d + i + NT); __builtin_nontemporal_store(c, d + i + 2 * NT); __builtin_nontemporal_store(e, d + i + 3 * NT); }
        else { d[i] = a; d[i + NT] = b; d[i + 2 * NT] = c; d[i + 3 * NT] = e; }
    }
    for (; i < vec_per_env; i += NT) { u32x4 a = src[i]; if (NTMP) __builtin_nontemporal_store(a, d + i); else d[i] = a; }
}
template <typename F> static void timeit(const char *name, size_t bytes, F launch) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 5; r++) { CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; }
    printf("%-46s %.3f ms  %.2f TB/s\n", name, best, bytes / best / 1e9);
}
int main() {
    const int n = 32768, vec = 2 * 240 * 320 * 3 / 16;   // 28800 vectors = 460 800 B per env
    const size_t bytes = (size_t)n * vec * 16;
    u32x4 *src, *dst; CHECK(hipMalloc(&src, (size_t)vec * 16)); CHECK(hipMalloc(&dst, bytes));
    CHECK(hipMemset(src, 1, (size_t)vec * 16));
    timeit("256 thr/WG, load+nt store", bytes, [&] { hipLaunchKernelGGL((k_copy<256, true, 1>), dim3(n), dim3(256), 0, 0, src, dst, vec); });
    timeit("256 thr/WG, load+plain store", bytes, [&] { hipLaunchKernelGGL((k_copy<256, true, 0>), dim3(n), dim3(256), 0, 0, src, dst, vec); });
    timeit("256 thr/WG, no load, nt store", bytes, [&] { hipLaunchKernelGGL((k_copy<256, false, 1>), dim3(n), dim3(256), 0, 0, src, dst, vec); });
    timeit("256 thr/WG, no load, plain store", bytes, [&] { hipLaunchKernelGGL((k_copy<256, false, 0>), dim3(n), dim3(256), 0, 0, src, dst, vec); });
    timeit("1024 thr/WG, load+nt store", bytes, [&] { hipLaunchKernelGGL((k_copy<1024, true, 1>), dim3(n), dim3(1024), 0, 0, src, dst, vec); });
    timeit("64 thr/WG x4n, load+nt store (quarter env)", bytes, [&] { hipLaunchKernelGGL((k_copy<64, true, 1>), dim3(n * 4), dim3(64), 0, 0, src, dst, vec / 4); });
    timeit("256 thr/WG, 4 loads then 4 nt stores", bytes, [&] { hipLaunchKernelGGL((k_copy4<256, 1>), dim3(n), dim3(256), 0, 0, src, dst, vec); });
    timeit("256 thr/WG, 4 loads then 4 plain stores", bytes, [&] { hipLaunchKernelGGL((k_copy4<256, 0>), dim3(n), dim3(256), 0, 0, src, dst, vec); });
    timeit("512 thr/WG, 4 loads then 4 nt stores", bytes, [&] { hipLaunchKernelGGL((k_copy4<512, 1>), dim3(n), dim3(512), 0, 0, src, dst, vec); });
    CHECK(hipMemsetAsync(dst, 0, bytes)); CHECK(hipDeviceSynchronize());
    timeit("hipMemsetAsync (runtime fill kernel)", bytes, [&] { CHECK(hipMemsetAsync(dst, 0, bytes)); });
    return 0;
}
