// Are the four DPP steps of lcr_newton_coop.h's row_sum bit-identical in the 16 lanes of a row?   hipcc --offload-arch=gfx950 -O3 -ffast-math tools/ubench/row_sum_check.hip -o row_sum_check && ./row_sum_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
__device__ __forceinline__ float row_sum(float v, int variant) {
    auto dpp = [](float x, auto ctrl_tag) -> float {
        constexpr int ctrl = decltype(ctrl_tag)::value;
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, 0xf, 0xf, false));
    };
    v += dpp(v, std::integral_constant<int, 0xb1>{});
    v += dpp(v, std::integral_constant<int, 0x4e>{});
    if (variant == 0) { v += dpp(v, std::integral_constant<int, 0x141>{}); v += dpp(v, std::integral_constant<int, 0x140>{}); }
    else { v += dpp(v, std::integral_constant<int, 0x124>{}); v += dpp(v, std::integral_constant<int, 0x128>{}); }
    return v;
}
__global__ void k(const float *in, float *out, int variant) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    out[i] = row_sum(in[i], variant);
}
int main() {
    const int n = 64 * 4096;
    float *h = (float *)malloc(n * 4), *o = (float *)malloc(n * 4), *di, *dout;
    srand(1);
    for (int i = 0; i < n; i++) h[i] = ((float)rand() / RAND_MAX - 0.5f) * expf(10.f * ((float)rand() / RAND_MAX - 0.5f));
    hipMalloc(&di, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(di, h, n * 4, hipMemcpyHostToDevice);
    for (int variant = 0; variant < 2; variant++) {
        hipLaunchKernelGGL(k, dim3(n / 64), dim3(64), 0, 0, di, dout, variant);
        hipMemcpy(o, dout, n * 4, hipMemcpyDeviceToHost);
        int bad_rows = 0, wrong = 0;
        for (int r = 0; r < n / 16; r++) {
            bool same = true; double ex = 0;
            for (int l = 0; l < 16; l++) { ex += h[16 * r + l]; if (o[16 * r + l] != o[16 * r]) same = false; }
            bad_rows += same ? 0 : 1;
            if (fabs(o[16 * r] - ex) > 1e-4 * (fabs(ex) + 1.0)) wrong++;
        }
        printf("variant %d (%s): rows whose 16 lanes do not hold the same bits: %d of %d; rows with a wrong sum: %d\n", variant, variant == 0 ? "mirrors" : "rotations", bad_rows, n / 16, wrong);
    }
    return 0;
}
