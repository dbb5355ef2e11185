// Do MFMA and VALU work of DIFFERENT waves of one SIMD overlap on gfx950?  Workgroup = 8 waves (w and w + 4 share a SIMD): waves 0-3 run
// NM back-to-back v_mfma_f32_32x32x16_bf16 on two independent accumulators, waves 4-7 run NV independent v_fma_f32 (8 chains); each
// role is timed alone and together.   hipcc --offload-arch=gfx950 -O3 -o mb_coissue mb_coissue.cpp && ./mb_coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int iters, int do_mfma, int do_valu, int valu_kind) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (lane + j)); b[j] = (__bf16)(0.02f * (lane - j)); }
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = 0.001f * (lane + j);
    double dv[4];
    for (int j = 0; j < 4; ++j) dv[j] = 0.001 * (lane + j);
    const float m = 1.0001f, c = 0.5f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (w < 4) {
        if (do_mfma)
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
                }
            }
    } else if (do_valu) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (valu_kind == 0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(m), "v"(c));
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { asm volatile("v_max_f64 %0, %0, %1" : "+v"(dv[j]) : "v"(dv[(j + 1) & 3])); asm volatile("v_min_f64 %0, %0, %1" : "+v"(dv[j]) : "v"(dv[(j + 2) & 3])); }
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    for (int j = 0; j < 8; ++j) s += v[j];
    for (int j = 0; j < 4; ++j) s += (float)dv[j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

int main() {
    const int blocks = 256, iters = 2000;
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 4 * blocks * 512); (void)hipMalloc(&cyc, 8 * blocks * 8);
    static unsigned long long h[256 * 8];
    for (int kind = 0; kind < 2; ++kind)
        for (int mode = 1; mode <= 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, cyc, iters, mode & 1, (mode >> 1) & 1, kind);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            double tm = 0, tv = 0;
            for (int bI = 0; bI < blocks; ++bI) for (int w = 0; w < 8; ++w) (w < 4 ? tm : tv) += (double)h[bI * 8 + w];
            tm /= blocks * 4; tv /= blocks * 4;
            printf("%s  mfma %d valu %d:  MFMA waves %7.1f cycles per product (8 per iteration), VALU waves %6.2f cycles per instruction (64 per iteration)\n",
                   kind ? "v_max/min_f64" : "v_fma_f32    ", mode & 1, (mode >> 1) & 1, tm / iters / 8, tv / iters / 64);
        }
    return 0;
}
