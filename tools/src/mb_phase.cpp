// Do the PHASES of different waves of a SIMD overlap?  Each wave loops over "tiles": 12 dependent products, NV vector instructions (16 of them
// v_exp_f32, sums as a dependent chain like a softmax), 12 products on two chains, optionally an LDS round trip in front of each product
// phase and a workgroup barrier per tile.  256-thread workgroups, WPC workgroups per CU (= waves per SIMD), all CUs.
//   hipcc --offload-arch=gfx950 -O3 -o mb_phase mb_phase.cpp && ./mb_phase
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool BARRIER, bool LDSRT, int NRD = 0, bool DMA = false, int BIAS = 0, bool RESTART = false>   // BIAS: accumulator start values from LDS (8 ds_read2_b32); RESTART: a workgroup's prologue / epilogue every 16 tiles      // NRD: ds_read_b128 per product phase (the attention kernel: 8 + 8); DMA: 4 x 1 KiB LDS-DMA per wave and tile
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters, int lds_pad, const char *src = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (lane + j)); b[j] = (__bf16)(0.02f * (lane - j)); }
    f32x16 s, o0, o1;
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; o0[r] = 0.f; o1[r] = 0.f; }
    for (int i = threadIdx.x; i < 2048; i += 256) ((float *)lds)[i] = 0.001f * i;
    __syncthreads();
    float l = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (RESTART && (it & 15) == 0) {
            // epilogue of the previous "workgroup": 32 lanes x 64 B stores per wave (two 64-byte plane rows per query); prologue of the next: the bias table
            // (strided 4-byte loads), the query rows (8 x 16 B per lane), the first tile's DMA and the wait for it
            float *og = out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
#pragma unroll
            for (int r = 0; r < 16; r += 4) *(float4 *)(og + r) = make_float4(o0[r], o0[r + 1], o1[r + 2], o1[r + 3]);
            __syncthreads();
            for (int i = threadIdx.x; i < 1023; i += 256) ((float *)lds)[1024 + i] = ((const float *)src)[(size_t)i * 4 + (blockIdx.x & 3)];
            const float4 *qg = (const float4 *)(src + ((size_t)(blockIdx.x * 25 + (it >> 4)) % 4096) * 131072 + threadIdx.x * 512);
            float4 qv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) qv[i] = qg[i];
#pragma unroll
            for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(qv[j].x + qv[j].y); b[j] = (__bf16)(qv[j].z - qv[j].w); }
            if (DMA) {
                const char *g = src + ((size_t)blockIdx.x * 64 + (it & 63)) * 16384 + w * 4096 + lane * 16;
                char *d = lds + 8192 + (it & 1) * 16384 + w * 4096;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + i * 1024), (__attribute__((address_space(3))) void *)(d + i * 1024), 16, 0, 0);
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
            l = 0.f;
        }
        if (BARRIER) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (DMA) {
            const char *g = src + ((size_t)blockIdx.x * 64 + (it & 63)) * 16384 + w * 4096 + lane * 16;
            char *d = lds + 8192 + ((it + 1) & 1) * 16384 + w * 4096;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + i * 1024), (__attribute__((address_space(3))) void *)(d + i * 1024), 16, 0, 0);
        }
        if (LDSRT) { a = *(const bf16x8 *)(lds + lane * 16); asm volatile("" : "+v"(a)); }
        bf16x8 fr[8];
        if (NRD) {
#pragma unroll
            for (int i = 0; i < NRD; ++i) fr[i & 7] = *(const bf16x8 *)(lds + 8192 + (it & 1) * 16384 + ((i * 1024 + lane * 16) & 16383));
#pragma unroll
            for (int i = 0; i < (NRD < 8 ? NRD : 8); ++i) asm volatile("" : "+v"(fr[i]));
            a = fr[0];
        }
        f32x16 bias16;
        if (BIAS) {     // 1: the accumulators START from the bias; 2: the bias is read in front of the products and ADDED behind them
            const float *tb = (const float *)lds + 1024 + (lane & 31) - (it & 15) * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) bias16[r] = tb[511 - ((r & 3) + 8 * (r >> 2))];
            if (BIAS == 1) s = bias16;
        }
#pragma unroll
        for (int c = 0; c < 12; ++c) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s, 0, 0, 0);
        if (BIAS == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] += bias16[r];
        }
        // "softmax": 16 exp, a dependent sum, the hi / lo split (about 105 vector instructions)
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] * 1e-3f); ps += s[r]; }
        l += ps;
        bf16x8 ph[2], pl[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float x = s[8 * c + j]; const __bf16 hh = (__bf16)x; ph[c][j] = hh; pl[c][j] = (__bf16)(x - (float)hh); }
        if (LDSRT) { b = *(const bf16x8 *)(lds + 4096 + lane * 16); asm volatile("" : "+v"(b)); }
        if (NRD) {
#pragma unroll
            for (int i = 0; i < NRD; ++i) fr[i & 7] = *(const bf16x8 *)(lds + 8192 + (it & 1) * 16384 + ((8192 + i * 1024 + lane * 16) & 16383));
#pragma unroll
            for (int i = 0; i < (NRD < 8 ? NRD : 8); ++i) asm volatile("" : "+v"(fr[i]));
            b = fr[1];
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, ph[c], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, ph[c], o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pl[c], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, pl[c], o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, ph[c], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, ph[c], o1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = o0[r] * 1e-6f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float x = l;
    for (int r = 0; r < 16; ++r) x += o0[r] + o1[r] + s[r];
    out[blockIdx.x * 256 + threadIdx.x] = x;
    if (lane == 0) cyc[blockIdx.x * 4 + w] = t1 - t0;
}

int main() {
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 4 * 1024 * 256 * 16); (void)hipMalloc(&cyc, 8 * 1024 * 4);
    static unsigned long long h[1024 * 4];
    const int iters = 400;
    printf("shader cycles per tile (24 products = 768 of matrix pipe + ~120 vector instructions): per wave, and per SIMD (= / waves per SIMD)\n");
    for (int wpc = 1; wpc <= 4; ++wpc) {
        const int lds_bytes = 160 * 1024 / wpc - 1024;            // LDS limits the workgroups per CU to wpc
        for (int mode = 0; mode < 4; ++mode) {
            const bool bar = mode & 1, rt = mode & 2;
            auto go = [&]() {
#define GO(B, R) { (void)hipFuncSetAttribute((const void *)k<B, R>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); hipLaunchKernelGGL((k<B, R>), dim3(256 * wpc), dim3(256), lds_bytes, 0, out, cyc, iters, 0); }
                if (bar && rt) GO(true, true) else if (bar) GO(true, false) else if (rt) GO(false, true) else GO(false, false)
            };
            go(); go();
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h, cyc, 8 * 256 * wpc * 4, hipMemcpyDeviceToHost);
            double t = 0;
            for (int i = 0; i < 256 * wpc * 4; ++i) t += (double)h[i];
            t /= 256.0 * wpc * 4 * iters;
            printf("%d wave(s) per SIMD, barrier per tile %d, LDS round trips %d: %7.0f per wave, %7.0f per SIMD\n", wpc, (int)bar, (int)rt, t, t / wpc);
        }
    }
    // towards the attention kernel (4 workgroups per CU, barrier per tile): 8 + 8 fragment reads per tile, LDS-DMA of the next tile
    {
        char *src; (void)hipMalloc(&src, (size_t)1024 * 64 * 16384);
        (void)hipMemset(src, 0, (size_t)1024 * 64 * 16384);
        const int wpc = 4, lds_bytes = 160 * 1024 / wpc - 1024;
        auto run = [&](const char *nm, auto kern) {
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(256 * wpc), dim3(256), lds_bytes, 0, out, cyc, iters, 0, (const char *)src);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h, cyc, 8 * 256 * wpc * 4, hipMemcpyDeviceToHost);
            double t = 0;
            for (int i = 0; i < 256 * wpc * 4; ++i) t += (double)h[i];
            t /= 256.0 * wpc * 4 * iters;
            printf("4 waves per SIMD, barrier per tile, %-44s %7.0f per wave, %7.0f per SIMD\n", nm, t, t / wpc);
        };
        run("+ 8 + 8 fragment reads per tile:", k<true, false, 8, false>);
        run("+ LDS-DMA of the next tile (4 KiB per wave):", k<true, true, 0, true>);
        run("+ both:", k<true, false, 8, true>);
        run("+ both + bias as start values from LDS:", k<true, false, 8, true, 1>);
        run("+ both + bias read early, added behind:", k<true, false, 8, true, 2>);
        run("+ start-value form + workgroup restart / 16 tiles:", k<true, false, 8, true, 1, true>);
        run("+ added-behind form + workgroup restart / 16 tiles:", k<true, false, 8, true, 2, true>);
    }
    return 0;
}
