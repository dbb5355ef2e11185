// Dependent vs interleaved MFMA chains on gfx950: cycles per v_mfma_f32_32x32x16_{f16,bf16} with NACC independent accumulators issued
// round-robin, one or two waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o mb_mfma_chain mb_mfma_chain.cpp && ./mb_mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool F16>
__global__ __launch_bounds__(512) void chain_kernel(float *out, unsigned long long *cyc, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 ah, bh; bf16x8 ab, bb;
    for (int j = 0; j < 8; ++j) { ah[j] = (_Float16)(0.01f * (lane + j)); bh[j] = (_Float16)(0.02f * (lane - j)); ab[j] = (__bf16)(0.01f * (lane + j)); bb[j] = (__bf16)(0.02f * (lane - j)); }
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            if (F16) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[a], 0, 0, 0);
            else acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[a], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    asm volatile("" :: "v"(s));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NACC, bool F16>
static void run(int threads, const char *what) {
    const int blocks = 256, iters = 2000;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads); hipMalloc(&cyc, 8 * blocks * (threads / 64));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((chain_kernel<NACC, F16>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[256 * 8];
    hipMemcpy(h, cyc, 8 * blocks * (threads / 64), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < blocks * (threads / 64); ++i) m += (double)h[i];
    m /= blocks * (threads / 64);
    printf("%-5s accumulators %d, waves per SIMD %d: %.1f cycles per MFMA per wave (%.1f per SIMD-issue)\n", what, NACC, threads / 256, m / (iters * NACC),
           m / (iters * NACC) / (threads / 256));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<1, true>(256, "f16"); run<2, true>(256, "f16"); run<4, true>(256, "f16");
    run<1, false>(256, "bf16"); run<2, false>(256, "bf16"); run<4, false>(256, "bf16");
    run<1, true>(512, "f16"); run<2, true>(512, "f16");
    run<1, false>(512, "bf16"); run<2, false>(512, "bf16");
    return 0;
}
