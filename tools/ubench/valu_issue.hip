// valu_issue.hip -- micro-benchmark behind DESIGN.md's step-kernel strategy: how many cycles does one wave64 VALU
// instruction cost on a gfx950 SIMD, as a function of (a) instruction kind (v_fma_f32, v_pk_fma_f32, v_accvgpr moves,
// v_rcp_f32, v_cndmask), (b) independent chains per lane (ILP) and (c) waves resident per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_issue.hip -o gpurun_out/valu_issue && gpurun_out/valu_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int ILP, int KIND>
__global__ __launch_bounds__(64) void k_chain(float *out, long long *cyc, int iters, float a, float b) {
    float acc[ILP];
    f2 acc2[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) { acc[i] = threadIdx.x * 1e-3f + i; acc2[i] = f2{acc[i], acc[i] + 0.5f}; }
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc2[i]) : "v"(f2{a, a}), "v"(f2{b, b}));
                if (KIND == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(acc[i]));
                if (KIND == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(a) : );
                if (KIND == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
                if (KIND == 5) asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %0, a0" : "+v"(acc[i]) : : "a0");
                if (KIND == 6) asm volatile("v_mov_b32 %0, %1" : "=v"(acc[i]) : "v"(acc[(i + 1) % ILP]));
                if (KIND == 7) asm volatile("v_rsq_f32 %0, %0" : "+v"(acc[i]));
                if (KIND == 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc2[i]) : "v"(f2{a, a}));
                if (KIND == 9) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (KIND == 10) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(acc[i]) : "v"(a) : "s20", "s21");
                if (KIND == 11) asm volatile("v_cmp_lt_f32_e64 s[20:21], %1, %2\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(acc[i]) : "v"(a), "v"(b) : "s20", "s21");
                if (KIND == 12) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %2\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
                if (KIND == 13) asm volatile("v_max_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
                if (KIND == 14) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (KIND == 15) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" : : "v"(acc[i]), "v"(a) : "s20", "s21");
                if (KIND == 16) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(acc[i]));
                if (KIND == 17) asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(a));
                if (KIND == 18) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
                if (KIND == 19) asm volatile("v_fma_f32 %0, %0, %1, s20" : "+v"(acc[i]) : "v"(a) : );
                if (KIND == 20) asm volatile("v_mul_f32 %0, 0x3f800347, %0" : "+v"(acc[i]));
            }
        }
    }
    long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += acc[i] + acc2[i].x + acc2[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int ILP, int KIND>
static void run(const char *name, int waves_per_simd, int nsimd) {
    const int blocks = nsimd * waves_per_simd, iters = 4000;
    float *out; long long *cyc;
    CHECK(hipMalloc(&out, (size_t)blocks * 64 * 4));
    CHECK(hipMalloc(&cyc, (size_t)blocks * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_chain<ILP, KIND>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1.0001f, 1e-6f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
    }
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long *h = (long long *)malloc((size_t)blocks * 8);
    CHECK(hipMemcpy(h, cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost));
    double mean = 0; for (int i = 0; i < blocks; i++) mean += (double)h[i]; mean /= blocks;
    const double ninst = (double)iters * 8 * ILP * ((KIND == 5 || KIND == 11 || KIND == 12) ? 2 : 1);
    // per-SIMD cost of one wave-instruction: wall time x clock / (instructions x waves on the SIMD)
    printf("%-14s ilp=%d waves/simd=%d  ms=%.4f  clock64/inst/wave=%.2f  (=> %.2f per inst per SIMD)  wall ns/inst/SIMD=%.3f\n", name, ILP,
           waves_per_simd, ms, mean / ninst, mean / ninst / waves_per_simd, ms * 1e6 / (ninst * waves_per_simd));
    free(h); CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int nsimd = p.multiProcessorCount * 4;
    printf("%s CUs=%d clockRate=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    for (int w : {1, 2}) {
        run<8, 0>("v_fma_f32", w, nsimd); run<8, 1>("v_pk_fma_f32", w, nsimd);
        run<8, 3>("cndmask vcc", w, nsimd); run<8, 10>("cndmask sgpr", w, nsimd); run<8, 11>("cmp+cnd sgpr", w, nsimd); run<8, 12>("cmp+cnd vcc", w, nsimd);
        run<8, 13>("v_max_f32", w, nsimd); run<8, 14>("v_med3_f32", w, nsimd); run<8, 15>("v_cmp only", w, nsimd);
        run<8, 16>("mov_dpp", w, nsimd); run<8, 17>("add_dpp", w, nsimd); run<8, 18>("v_add_u32", w, nsimd);
        run<8, 19>("fma sgpr src", w, nsimd); run<8, 20>("mul literal", w, nsimd);
        run<8, 2>("v_rcp_f32", w, nsimd); run<4, 5>("accvgpr w+r", w, nsimd);
    }
    return 0;
}
