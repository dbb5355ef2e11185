// Does the clock the chip holds under an MFMA-dense loop depend on the MFMA SHAPE (MI355X_MICROARCH.md, DVFS give-back item 7)?
// One wave per SIMD (4 per CU, 256 CUs), random bf16 operands, the split-product pattern of the row-chain kernels: per weight fragment pair
// (hi, lo - read from LDS every time, or kept in registers) three products lo.hi, hi.lo, hi.hi into one accumulator, rows as the B operand
// in registers.  Shape 0: v_mfma_f32_32x32x16_bf16, 32 rows per wave, NCH accumulator chains; shape 1: v_mfma_f32_16x16x32_bf16, two 16-row
// tiles per wave (the same weight fragments feed both), 2 x NCH chains.  Equal MACs, equal LDS bytes, equal register operands per MAC.
// Reports wall time, product rate and the in-kernel clock (delta s_memtime / delta s_memrealtime x 100 MHz, median over workgroups).
//   hipcc --offload-arch=gfx950 -O3 -o mb_shape mb_shape.cpp && ./mb_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define NCH 4          // output-channel blocks of 32 per sweep (accumulator chains)
#define KC 4           // k-chunks of 32 input channels per sweep

template <int SHAPE, bool LDSFED>
__global__ __launch_bounds__(256) void k(const u32x4 *__restrict__ w, const u32x4 *__restrict__ x, float *out, unsigned long long *stamps, int iters) {
    // LDS image of the weights: [KC][NCH][2 planes][2 halves][64 lanes] 16-byte fragments = 4*4*2*2*1 KB = 64 KB
    extern __shared__ u32x4 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < KC * NCH * 4 * 64; i += 256) lds[i] = w[i];
    __syncthreads();
    // rows: B fragments hi / lo for KC k-chunks of 32 channels: two 16-channel fragments per chunk (32x32x16) or one 32-channel fragment per 16-row tile (16x16x32): 4 x 16 B either way
    bf16x8 xh[KC][2], xl[KC][2];
    for (int c = 0; c < KC; ++c)
        for (int h = 0; h < 2; ++h) {
            u32x4 a = x[((blockIdx.x * 4 + wave) * KC * 4 + c * 4 + h * 2) * 64 + lane], b = x[((blockIdx.x * 4 + wave) * KC * 4 + c * 4 + h * 2 + 1) * 64 + lane];
            xh[c][h] = __builtin_bit_cast(bf16x8, a);
            xl[c][h] = __builtin_bit_cast(bf16x8, b);
        }
    f32x16 acc[NCH];
    for (int n = 0; n < NCH; ++n)
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    bf16x8 wr[NCH][4];
    if (!LDSFED)
        for (int n = 0; n < NCH; ++n)
            for (int q = 0; q < 4; ++q) wr[n][q] = __builtin_bit_cast(bf16x8, lds[(n * 4 + q) * 64 + lane]);
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int loff = lane;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(loff));          // the LDS reads stay inside the loop
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            bf16x8 wf[NCH][4];                  // per 32 output channels x 32 input channels: hi / lo planes x two halves, 16 B per lane each
#pragma unroll
            for (int n = 0; n < NCH; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q) wf[n][q] = LDSFED ? __builtin_bit_cast(bf16x8, lds[((c * NCH + n) * 4 + q) * 64 + loff]) : wr[n][q];
            if (SHAPE == 0) {
                // A = 32 channels x 16 k: fragments 0 / 1 = hi / lo of the first k half, 2 / 3 of the second; NCH independent chains, products interleaved
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int n = 0; n < NCH; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[n][2 * h + 1], xh[c][h], acc[n], 0, 0, 0);
#pragma unroll
                    for (int n = 0; n < NCH; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[n][2 * h], xl[c][h], acc[n], 0, 0, 0);
#pragma unroll
                    for (int n = 0; n < NCH; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[n][2 * h], xh[c][h], acc[n], 0, 0, 0);
                }
            } else {
                // A = 16 channels x 32 k: fragments 0 / 1 = hi / lo of the first channel half, 2 / 3 of the second; two row tiles of 16 (xh[c][0 / 1]): 4 NCH chains
                f32x4 cc[NCH][4];
#pragma unroll
                for (int n = 0; n < NCH; ++n)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        for (int r = 0; r < 4; ++r) cc[n][q][r] = acc[n][4 * q + r];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
#pragma unroll
                    for (int n = 0; n < NCH; ++n)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const bf16x8 a = wf[n][2 * (q >> 1) + (p == 0 ? 1 : 0)];
                            const bf16x8 b = p == 1 ? xl[c][q & 1] : xh[c][q & 1];
                            cc[n][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, cc[n][q], 0, 0, 0);
                        }
                }
#pragma unroll
                for (int n = 0; n < NCH; ++n)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        for (int r = 0; r < 4; ++r) acc[n][4 * q + r] = cc[n][q][r];
            }
        }
        // keep the magnitudes bounded without leaving the matrix pipe idle for long
        if ((it & 63) == 63)
            for (int n = 0; n < NCH; ++n)
                for (int r = 0; r < 16; ++r) acc[n][r] *= 1e-3f;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int n = 0; n < NCH; ++n)
        for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int SHAPE, bool LDSFED>
static void run(const char *name, u32x4 *w, u32x4 *x, float *out, unsigned long long *st, int iters, int G) {
    (void)hipFuncSetAttribute((const void *)k<SHAPE, LDSFED>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, LDSFED>), dim3(G), dim3(256), 128 * 1024, 0, w, x, out, st, iters);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(2 * G);
        (void)hipMemcpy(h.data(), st, 16 * G, hipMemcpyDeviceToHost);
        std::vector<double> clk(G), cyc(G);
        for (int g = 0; g < G; ++g) { clk[g] = (double)h[2 * g] / (double)h[2 * g + 1] * 100.0; cyc[g] = (double)h[2 * g]; }
        std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
        const double macs = (double)G * 4 * iters * KC * NCH * 6 * 16384.0;      // per wave and sweep: KC x NCH x 6 products of 32x32x16 MACs
        printf("%-34s %8.2f ms  %7.1f TFLOP/s (16-bit dense)  %6.1f per fp32-class product  in-kernel clock %4.0f MHz  cycles per 32x32x16-equivalent product %.1f\n", name, ms,
               2 * macs / (ms * 1e-3) / 1e12, 2 * macs / 3 / (ms * 1e-3) / 1e12, clk[G / 2], cyc[G / 2] / ((double)iters * KC * NCH * 6));
    }
}

int main(int argc, char **argv) {
    const int G = 256, iters = argc > 1 ? atoi(argv[1]) : 6000;
    const size_t nw = (size_t)KC * NCH * 4 * 64, nx = (size_t)G * 4 * KC * 4 * 64;
    std::vector<u32x4> hw(nw), hx(nx);
    srand(1);
    auto rnd = [&]() {      // two random bf16 in (-1, 1): sign, exponent 120 - 126, random mantissa
        unsigned v = 0;
        for (int h = 0; h < 2; ++h) v |= (((rand() & 1) << 15) | ((120 + rand() % 7) << 7) | (rand() & 127)) << (16 * h);
        return v;
    };
    for (auto &v : hw) v = u32x4{rnd(), rnd(), rnd(), rnd()};
    for (auto &v : hx) v = u32x4{rnd(), rnd(), rnd(), rnd()};
    u32x4 *w, *x; float *out; unsigned long long *st;
    (void)hipMalloc(&w, nw * 16); (void)hipMalloc(&x, nx * 16); (void)hipMalloc(&out, 4 * G * 256); (void)hipMalloc(&st, 16 * G);
    (void)hipMemcpy(w, hw.data(), nw * 16, hipMemcpyHostToDevice); (void)hipMemcpy(x, hx.data(), nx * 16, hipMemcpyHostToDevice);
    for (int round = 0; round < 2; ++round) {
        run<0, false>("32x32x16, weights in registers", w, x, out, st, iters, G);
        run<1, false>("16x16x32, weights in registers", w, x, out, st, iters, G);
        run<0, true>("32x32x16, weights from LDS", w, x, out, st, iters, G);
        run<1, true>("16x16x32, weights from LDS", w, x, out, st, iters, G);
    }
    return 0;
}
