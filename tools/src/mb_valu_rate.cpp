// VALU issue rate of gfx950 per wave and per SIMD: W waves per SIMD (workgroup of 4 W waves) each run 64 independent-chain instructions
// per iteration; cycles per instruction seen by one wave, for several instruction kinds.
//   hipcc --offload-arch=gfx950 -O3 -o mb_valu_rate mb_valu_rate.cpp && ./mb_valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k(float *out, unsigned long long *cyc, int iters) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float v[8]; f32x2 p[8]; double d[8]; unsigned u[8];
    for (int j = 0; j < 8; ++j) { v[j] = 0.001f * (lane + j); p[j] = (f32x2){v[j], v[j] + 1.f}; d[j] = 0.001 * (lane + j); u[j] = lane * 7 + j; }
    const float m = 1.0001f, c = 0.5f;
    const f32x2 pm = {m, m}, pc = {c, c};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(m), "v"(c));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(pm), "v"(pc));
                if (KIND == 2) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[j]) : "v"(d[(j + 1) & 7]));
                if (KIND == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
                if (KIND == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
                if (KIND == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
                if (KIND == 6) asm volatile("v_cmp_gt_u64 vcc, %0, %1" :: "v"(d[j]), "v"(d[(j + 1) & 7]) : "vcc");
                if (KIND == 7) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(u[j]) : "v"(u[(j + 1) & 7]) : "s20", "s21");
                if (KIND == 8) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j]) : "v"(v[(j + 1) & 7]), "v"(m) : "vcc");
                if (KIND == 9) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[j]) : "v"(v[(j + 1) & 7]));
                if (KIND == 10) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "v"(u[(j + 2) & 7]));
                if (KIND == 11) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
                if (KIND == 12) asm volatile("v_cmp_ge_f32 s[20:21], %0, %1" :: "v"(v[j]), "v"(v[(j + 1) & 7]) : "s20", "s21");
                if (KIND == 13) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[j]) : "v"(m));
                if (KIND == 14) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(v[(j + 1) & 7]), "v"(v[(j + 2) & 7]));
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += v[j] + p[j][0] + p[j][1] + (float)d[j] + (float)u[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 16 + w] = t1 - t0;
}

template <int KIND>
static void run(const char *name, float *out, unsigned long long *cyc) {
    static unsigned long long h[256 * 16];
    printf("%-14s", name);
    for (int wps = 1; wps <= 4; ++wps) {
        const int threads = 256 * wps, blocks = 256, iters = 1000;
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double t = 0;
        for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4 * wps; ++w) t += (double)h[b * 16 + w];
        t /= blocks * 4 * wps;
        printf("  %d wave%s/SIMD: %5.2f cycles per instruction and wave (%5.2f per SIMD)", wps, wps > 1 ? "s" : " ", t / iters / 64, t / iters / 64 / wps);
    }
    printf("\n");
}

int main() {
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 4 * 256 * 1024); (void)hipMalloc(&cyc, 8 * 256 * 16);
    run<0>("v_fma_f32", out, cyc); run<1>("v_pk_fma_f32", out, cyc); run<2>("v_max_f64", out, cyc); run<3>("v_cndmask_b32", out, cyc);
    run<4>("v_add_u32", out, cyc); run<5>("v_exp_f32", out, cyc); run<6>("v_cmp_gt_u64", out, cyc);
    run<7>("cndmask sgpr", out, cyc); run<8>("cmp+cndmask", out, cyc); run<9>("v_max_f32", out, cyc); run<10>("v_bfi_b32", out, cyc);
    run<11>("v_lshl_or_b32", out, cyc); run<12>("v_cmp_f32 sgpr", out, cyc); run<13>("v_mul_f32", out, cyc); run<14>("v_max3_f32", out, cyc);
    return 0;
}
