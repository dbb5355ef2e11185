// Does the MFMA rate per CU depend on how many CUs are busy (clock / power management)?  G workgroups (one per CU, 8 waves, two
// independent bf16 32x32x16 chains per wave = the matrix pipe saturated) run a fixed number of products; wall time by events.
//   hipcc --offload-arch=gfx950 -O3 -o mb_power mb_power.cpp && ./mb_power
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k(float *out, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (lane + j)); b[j] = (__bf16)(0.02f * (lane - j)); }
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
    float *out; (void)hipMalloc(&out, 4 * 256 * 512);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 200000;    // 3.2 M products per wave: ~50 ms at full clock
    for (int g : {256, 256, 192, 128, 64, 16, 256}) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(g), dim3(512), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        // per SIMD two waves x 16 products per iteration, 32 cycles each
        const double cycles = 2.0 * 16 * 32 * iters;
        printf("%3d workgroups: %7.2f ms -> effective matrix-pipe clock %.0f MHz, %.0f TFLOP/s bf16 dense over the busy CUs (%.0f chip-equivalent)\n", g, ms,
               cycles / (ms * 1e-3) / 1e6, g * 8.0 * 16 * iters * 32768.0 / (ms * 1e-3) / 1e12, 256 * 8.0 * 16 * iters * 32768.0 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
