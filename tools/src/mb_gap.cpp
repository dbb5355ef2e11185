// What fits into the gap behind a v_mfma_f32_32x32x16_bf16 on gfx950?  One instruction stream per wave: NM x { product ; NV vector instructions },
// on one accumulation chain or two alternating ones, with 1 / 2 / 4 waves per SIMD (256 / 512 / 1024 threads per workgroup, one workgroup per CU),
// all 256 CUs running.  Prints shader cycles per product and wave, and per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mb_gap mb_gap.cpp && ./mb_gap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int CHAINS, int KIND>     // KIND 0: independent v_fma_f32 (8 registers round robin), 1: v_exp_f32, 2: dependent v_add_f32 chain, 3: ds_read_b128 (NV <= 2)
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *cyc, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[16384];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (lane + j)); b[j] = (__bf16)(0.02f * (lane - j)); }
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = 0.001f * (lane + j);
    const float m = 1.0001f, c = 0.5f;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) ((float *)lds)[i] = 0.001f * i;
    __syncthreads();
    const unsigned la = (unsigned)(lane * 16);
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 ld0 = {0, 0, 0, 0}, ld1 = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (CHAINS == 2 && (u & 1)) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(u * NV + j) & 7]) : "v"(m), "v"(c));
                if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(u * NV + j) & 7]));
                if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[0]) : "v"(c));
                if (KIND == 3) { if (j == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(ld0) : "v"(la)); else asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(ld1) : "v"(la)); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = ld0[0] + ld1[1];
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 16 + w] = t1 - t0;
}

static float *out; static unsigned long long *cyc; static unsigned long long h[256 * 16];
static double g_tflops = 0;
template <int NV, int CHAINS, int KIND> double run(int wps) {
    const int iters = 1000, threads = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, CHAINS, KIND>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipEventRecord(e0, 0);
    for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL((k<NV, CHAINS, KIND>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    g_tflops = 20.0 * 256 * 4 * wps * iters * 8 * 32768.0 / (ms * 1e-3) / 1e12;
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double t = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 4 * wps; ++w) t += (double)h[b * 16 + w];
    return t / (256.0 * 4 * wps) / iters / 8;
}
template <int NV, int KIND> void row(const char *nm) {
    printf("%-22s NV=%d |", nm, NV);
    for (int wps = 1; wps <= 4; wps *= 2) {
        const double c1 = run<NV, 1, KIND>(wps), c2 = run<NV, 2, KIND>(wps);
        printf("  %d wave(s)/SIMD: one chain %6.1f (%6.1f per SIMD), two chains %6.1f (%6.1f; %5.0f TFLOP/s wall)", wps, c1, c1 / wps, c2, c2 / wps, g_tflops);
    }
    printf("\n");
}
int main() {
    (void)hipMalloc(&out, 4 * 256 * 1024); (void)hipMalloc(&cyc, 8 * 256 * 16);
    printf("shader cycles per product and wave (per SIMD = / waves per SIMD; the matrix pipe needs 32)\n");
    row<0, 0>("products only");
    row<2, 0>("v_fma_f32"); row<4, 0>("v_fma_f32"); row<6, 0>("v_fma_f32"); row<8, 0>("v_fma_f32"); row<12, 0>("v_fma_f32");
    row<2, 1>("v_exp_f32"); row<4, 1>("v_exp_f32");
    row<2, 2>("dependent v_add_f32"); row<4, 2>("dependent v_add_f32"); row<6, 2>("dependent v_add_f32");
    row<1, 3>("ds_read_b128"); row<2, 3>("ds_read_b128");
    return 0;
}
