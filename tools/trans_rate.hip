// Micro-benchmark behind the "third finding" of profiles/r01_summary.md: issue cost of v_exp_f32 / v_rcp_f32 vs plain fp32 VALU on MI355X.
// hipcc --offload-arch=gfx950 -O3 -o tools/scratch/trans_rate tools/trans_rate.hip && gpurun -- ./tools/scratch/trans_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(float *o, int n) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = 0.001f * (threadIdx.x + i) + 0.5f;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = __builtin_fmaf(a[i], 0.999f, 0.001f);
            if (MODE == 1) a[i] = __builtin_amdgcn_exp2f(a[i] * 0.5f) * 0.5f;   // 1 trans + 2 mul
            if (MODE == 2) a[i] = __builtin_amdgcn_rcpf(a[i] + 1.0f) + 0.5f;    // 1 trans + 2 add
            if (MODE == 3) a[i] = a[i] * 0.5f * 0.999f + 0.01f;                 // mul, mul, add (compare with modes 1, 2)
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> static float run(float *o, int waves_per_simd) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 20000;
    dim3 grid(256 * waves_per_simd), block(256);
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, o, 10);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, o, n); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3f * 2.4e9f / (n * 8.0f * waves_per_simd);  // clocks per wave-iteration-element at 2.4 GHz
}
int main() {
    float *o; hipMalloc(&o, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4}) printf("waves/SIMD %d: fma %.2f clk | exp+2mul %.2f | rcp+2add %.2f | 3 plain %.2f\n", w, run<0>(o, w), run<1>(o, w), run<2>(o, w), run<3>(o, w));
    return 0;
}
