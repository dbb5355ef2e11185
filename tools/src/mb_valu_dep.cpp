// Dependent chains of v_pk_fma_f32 on gfx950: cycles per instruction of a wave that advances NCH independent chains round-robin
// (NCH = 1: fully dependent), with 1 or 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o mb_valu_dep mb_valu_dep.cpp && ./mb_valu_dep
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NCH>
__global__ void k(float *out, unsigned long long *cyc, int iters) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f32x2 p[NCH], u[NCH];
    for (int j = 0; j < NCH; ++j) { p[j] = (f32x2){0.001f * (lane + j), 0.002f * lane}; u[j] = (f32x2){0.5f, 0.25f}; }
    const f32x2 c = {0.125f, 0.125f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 64 / NCH; ++r)
#pragma unroll
            for (int j = 0; j < NCH; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n\ts_nop 0" : "+v"(p[j]) : "v"(u[j]), "v"(c));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < NCH; ++j) s += p[j][0] + p[j][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 16 + w] = t1 - t0;
}
template <int NCH>
static void run(float *out, unsigned long long *cyc) {
    static unsigned long long h[256 * 16];
    printf("%d chain%s per wave:", NCH, NCH > 1 ? "s" : " ");
    for (int wps = 1; wps <= 2; ++wps) {
        const int threads = 256 * wps, blocks = 256, iters = 1000;
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double t = 0;
        for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4 * wps; ++w) t += (double)h[b * 16 + w];
        t /= blocks * 4 * wps;
        printf("  %d wave%s/SIMD: %5.2f cycles per v_pk_fma_f32 and wave", wps, wps > 1 ? "s" : " ", t / iters / 64);
    }
    printf("\n");
}
int main() {
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 4 * 256 * 512); (void)hipMalloc(&cyc, 8 * 256 * 16);
    run<1>(out, cyc); run<2>(out, cyc); run<4>(out, cyc); run<8>(out, cyc);
    return 0;
}
