// Swin MLP in one kernel: C = GELU(X . W1^T + b1) . W2^T + b2 + R   (256 -> 1024 -> 256, swin_transformer.py:559-571), bf16x3 on
// split operands like gemm_split.hip, the 1024-wide hidden activation never leaves the CU (gfx950 / CDNA4).
//
// Workgroup = 8 waves (2 x 4) = 128 rows of X, persistent over row tiles.  The hidden dimension is processed in 4 super-chunks of
// 256 = two chunks of 128:
//   phase 1 (8 k-steps of 32):  H^T[256 hidden x 128 rows] = W1c . X^T      wave: (32 + 32) hidden x 64 rows (A = W1 rows, B = X
//                               rows: the accumulator then holds four consecutive hidden units per register group = one 8-byte
//                               store into the A-operand image of phase 2).  One pass over the activation tile feeds both chunks:
//                               the tile is streamed four times per row tile, not eight.
//   per chunk: bias + GELU + hi/lo split -> LDS image H[4 k-slabs][2 planes][128 rows][32] (64-byte rows, XOR swizzle of an A
//              stage), then phase 2 (4 k-steps of 32): Y[128 x 256] += H . W2c^T, wave: 64 rows x 64 columns; the second chunk's
//              accumulators wait in registers meanwhile.
// All global operands arrive by LDS-DMA into two 48 KiB stages (phase 1: X slab 16 KiB + W1 slab 32 KiB, phase 2: W2 slab 32 KiB),
// one step ahead, behind raw barriers (SCP_BARRIER_DMA); biases come through the scalar path (a vector load would drain the DMAs).
// Products, their order and the k order are those of two scp_linear_split calls (fc1 with GELU and split output, fc2 with
// residual): results are bit-identical.  LDS: 2 x 48 KiB stages + 64 KiB H image (reused as the epilogue bounce) = 160 KiB.
// Measured and dropped: 128-hidden chunks with three 32 KiB stages two steps ahead (8 passes over the activation tile: 6 - 9 %
// slower), GELU of the second chunk in the MFMA shadow of the first chunk's phase 2 (58 spilled registers, no gain), packed-fp32
// GELU (no gain), nt cache policy on either stream (13 - 20 % slower).  Round 2: the compiler issues the 12 dependent packed FMAs of a
// GELU polynomial back to back with an s_nop each (12.3 cycles per instruction and wave, tools/src/mb_valu_dep.cpp); two pairs' chains
// alternated with volatile asm (bit-identical) take the GELU phase from 34.9 k to 32.7 k cycles per tile - and the barrier waits from
// 49.6 k to 54.3 k: 1.958 against 1.968 ms per 577 536-row launch.  The tile is paced by its barriers, not by the waves' busy time.
#include <stdlib.h>
#include "scp_internal.h"

typedef __bf16 mbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 mbf16x4 __attribute__((ext_vector_type(4)));
typedef float mf32x16 __attribute__((ext_vector_type(16)));
typedef float mf32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *m_lds_ptr_t;
typedef const __attribute__((address_space(1))) void *m_glb_ptr_t;

__device__ __forceinline__ float mlp_gelu(float y) {   // the exact-erf GELU of gemm_split.hip (same polynomial, same order)
    const float z = y * 0.70710678118654752f;
    const float zc = fminf(fabsf(z), 3.5f);
    const float u = fmaf(zc * zc, 2.0f / 12.25f, -1.0f);
    float p = 1.480935152e-03f;
    p = fmaf(p, u, -3.987360327e-03f);
    p = fmaf(p, u, 4.474287011e-03f);
    p = fmaf(p, u, -7.227925849e-03f);
    p = fmaf(p, u, 1.704961757e-02f);
    p = fmaf(p, u, -3.003174999e-02f);
    p = fmaf(p, u, 4.501544287e-02f);
    p = fmaf(p, u, -6.477065166e-02f);
    p = fmaf(p, u, 8.840217622e-02f);
    p = fmaf(p, u, -1.146127499e-01f);
    p = fmaf(p, u, 1.467501802e-01f);
    p = fmaf(p, u, -2.007010379e-01f);
    p = fmaf(p, u, 4.038729840e-01f);
    const float e = copysignf(p * zc, z);
    const float hy = 0.5f * y;
    return fmaf(hy, e, hy);
}

__device__ __forceinline__ void mlp_dma16(const void *g, char *l) {
    __builtin_amdgcn_global_load_lds((m_glb_ptr_t)g, (m_lds_ptr_t)l, 16, 0, 0);
}

struct MlpArgs {
    const __bf16 *Xhi, *Xlo; int64_t ldx;        // [M][ldx], 256 columns used
    const __bf16 *W1hi, *W1lo;                   // [1024][256]
    const __bf16 *W2hi, *W2lo;                   // [256][1024]
    const float *b1, *b2;
    const float *res; int64_t ldr;               // fp32 [M][ldr] or null
    float *C; int64_t ldc;
    int M;
    int wtiled;                                  // W planes tiled (default) / row-major (SCP_WTILE=0)
};

#define MBM 128
#define M2STAGE 49152
#define M2HOFF (2 * M2STAGE)
template <bool DBG>
__global__ __launch_bounds__(512, 2) void mlp_fused_kernel(const MlpArgs a, unsigned long long *__restrict__ dbg) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;
    const int ntiles = (a.M + MBM - 1) / MBM;
    const int d_row = lane >> 2;
    const int d_q = (lane & 3) ^ ((lane >> 4) & 3);
    const int f_pos0 = (h ^ ((lane >> 2) & 3)) << 4;

    // step s of a tile: super-chunk sc = s / 16, t = s % 16; t < 8: phase 1 k-step t; 8..11: phase 2 of chunk 2 sc; 12..15: of chunk 2 sc + 1
    auto issue = [&](int m0, int s) {
        char *base = smem + (s & 1) * M2STAGE;
        const int sc = s >> 4, t = s & 15;
        if (t < 8) {
            const int k0 = t * 32 + 8 * d_q;
            int m = m0 + 16 * w + d_row;
            m = m < a.M ? m : a.M - 1;
            const int64_t xo = (int64_t)m * a.ldx + k0;
            mlp_dma16(a.Xhi + xo, base + w * 1024);
            mlp_dma16(a.Xlo + xo, base + 8192 + w * 1024);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cr = w + 8 * j;                          // 16-row chunk of the 256 hidden rows
                // W1 planes are TILED (scp_tile_weight_bf16): block (16-row group, 32-element k-slab) = the 1 KiB LDS image of one DMA
                // instruction, consecutive in memory - eight whole cache lines per instruction instead of sixteen half lines
                const int64_t wo = a.wtiled ? ((int64_t)(sc * 16 + cr) * 8 + t) * 512 + lane * 8 : (int64_t)(sc * 256 + 16 * cr + d_row) * 256 + k0;
                mlp_dma16(a.W1hi + wo, base + 16384 + cr * 1024);
                mlp_dma16(a.W1lo + wo, base + 32768 + cr * 1024);
            }
        } else {
            const int c = 2 * sc + ((t - 8) >> 2);
            const int k0s = c * 128 + ((t - 8) & 3) * 32;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cr = w + 8 * j;
                const int64_t wo = a.wtiled ? ((int64_t)cr * 32 + (k0s >> 5)) * 512 + lane * 8      // tiled W2 planes: [16 row groups][32 k-slabs][1 KiB]
                                            : (int64_t)(16 * cr + d_row) * 1024 + k0s + 8 * d_q;
                mlp_dma16(a.W2hi + wo, base + cr * 1024);
                mlp_dma16(a.W2lo + wo, base + 16384 + cr * 1024);
            }
        }
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    issue(tile * MBM, 0);
    // DBG build (tools/mb_mlp_fused.py stamps): shader cycles of this wave in [barrier waits, phase-1 products, GELU + split, phase-2
    // products, epilogue]
    unsigned long long t_bar = 0, t_p1 = 0, t_gelu = 0, t_p2 = 0, t_epi = 0, t_prev = 0;
    if (DBG) t_prev = __builtin_amdgcn_s_memtime();
    auto stamp = [&](unsigned long long &acc_t) { if (DBG) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc_t += t - t_prev; t_prev = t; } };

    for (; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * MBM;
        mf32x16 acc2[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
        mf32x16 acc1[2][2];                                         // [chunk of the pair][row tile]

        for (int sc = 0; sc < 4; ++sc) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc1[q][j][r] = 0.f;
            // ---------------- phase 1: both chunks of the pair ----------------
            for (int t = 0; t < 8; ++t) {
                const int s = sc * 16 + t;
                SCP_BARRIER_DMA(0);
                stamp(t_bar);
                issue(m0, s + 1);
                const char *st = smem + (s & 1) * M2STAGE;
                const int ow = 16384 + (wn * 32 + col) * 64, ox = (wm * 64 + col) * 64;
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    const int po = f_pos0 ^ (kc * 32);
                    mbf16x8 xh[2], xl[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        xh[j] = *(const mbf16x8 *)(st + ox + j * 2048 + po);
                        xl[j] = *(const mbf16x8 *)(st + ox + 8192 + j * 2048 + po);
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const mbf16x8 wh = *(const mbf16x8 *)(st + ow + q * 8192 + po), wl = *(const mbf16x8 *)(st + ow + 16384 + q * 8192 + po);
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc1[q][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl[j], acc1[q][j], 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc1[q][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh[j], acc1[q][j], 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc1[q][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh[j], acc1[q][j], 0, 0, 0);
                    }
                }
                if (DBG) { float keep; asm volatile("v_mov_b32 %0, %1" : "=v"(keep) : "v"(acc1[1][1][15])); asm volatile("" :: "v"(keep)); stamp(t_p1); }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = 2 * sc + q;
                // ---------------- bias + GELU + split of chunk c -> H image (k-slab wn) ----------------
                if (q == 1) { SCP_BARRIER_DMA(0); stamp(t_bar); }  // every wave is done reading chunk A's H image (the DMA in flight
                                                                    // is re-waited at the next step; one step of prefetch is lost here)
                {
                    char *hb = smem + M2HOFF + wn * 16384;
                    typedef const __attribute__((address_space(4))) float *m_const_ptr_t;
                    m_const_ptr_t b1p = (m_const_ptr_t)(uintptr_t)(a.b1 + c * 128 + wn * 32);
                    float sb[32];
#pragma unroll
                    for (int e = 0; e < 32; ++e) sb[e] = b1p[e];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        mf32x4 bv;
#pragma unroll
                        for (int u = 0; u < 4; ++u) bv[u] = h ? sb[8 * g + 4 + u] : sb[8 * g + u];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int xr = wm * 64 + 32 * j + col;
                            mbf16x4 hi, lo;
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const float y = mlp_gelu(acc1[q][j][4 * g + u] + bv[u]);
                                const __bf16 hh = (__bf16)y;
                                hi[u] = hh;
                                lo[u] = (__bf16)(y - (float)hh);
                            }
                            const int off = xr * 64 + ((g ^ ((xr >> 2) & 3)) << 4) + 8 * h;
                            *(mbf16x4 *)(hb + off) = hi;
                            *(mbf16x4 *)(hb + 8192 + off) = lo;
                        }
                    }
                }
                stamp(t_gelu);
                // ---------------- phase 2 of chunk c ----------------
                for (int t = 0; t < 4; ++t) {
                    const int s = sc * 16 + 8 + 4 * q + t;
                    SCP_BARRIER_DMA(0);                             // at t = 0 this also publishes the H image
                    stamp(t_bar);
                    if (s + 1 < 64) issue(m0, s + 1);
                    const char *st = smem + (s & 1) * M2STAGE;
                    const char *hs = smem + M2HOFF + t * 16384;
                    const int oa = (wm * 64 + col) * 64, ob = (wn * 64 + col) * 64;
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc) {
                        const int po = f_pos0 ^ (kc * 32);
                        mbf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            ah[i] = *(const mbf16x8 *)(hs + oa + i * 2048 + po);
                            al[i] = *(const mbf16x8 *)(hs + 8192 + oa + i * 2048 + po);
                            bh[i] = *(const mbf16x8 *)(st + ob + i * 2048 + po);
                            bl[i] = *(const mbf16x8 *)(st + 16384 + ob + i * 2048 + po);
                        }
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc2[i][j], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc2[i][j], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc2[i][j], 0, 0, 0);
                    }
                    if (DBG) { float keep; asm volatile("v_mov_b32 %0, %1" : "=v"(keep) : "v"(acc2[1][1][15])); asm volatile("" :: "v"(keep)); stamp(t_p2); }
                }
            }
        }
        SCP_WAIT_DMA(0);
        __syncthreads();                                            // every wave is done with the stages and the H image
        stamp(t_bar);
        if (tile + (int)gridDim.x < ntiles) issue((tile + gridDim.x) * MBM, 0);

        float *stg = (float *)(smem + M2HOFF + w * 8192);
        float bv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[j] = a.b2[wn * 64 + j * 32 + col];
        const int c4 = (lane & 15) * 4, rsub = lane >> 4;
        const int nb = wn * 64 + c4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mb = m0 + wm * 64 + i * 32 + rsub;
            mf32x4 rr[8];
            if (a.res) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int m = mb + 4 * it;
                    rr[it] = *(const mf32x4 *)(a.res + (int64_t)(m < a.M ? m : a.M - 1) * a.ldr + nb);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * h) * 64 + j * 32 + col] = acc2[i][j][r] + bv[j];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int m = mb + 4 * it;
                mf32x4 y = *(const mf32x4 *)(stg + (4 * it + rsub) * 64 + c4);
                if (a.res) y += rr[it];
                if (m < a.M) *(mf32x4 *)(a.C + (int64_t)m * a.ldc + nb) = y;
            }
        }
        stamp(t_epi);
    }
    if (DBG && dbg && lane == 0) {
        unsigned long long *o = dbg + ((size_t)blockIdx.x * 8 + w) * 8;
        o[0] = t_bar; o[1] = t_p1; o[2] = t_gelu; o[3] = t_p2; o[4] = t_epi; o[5] = (unsigned long long)((ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1);
    }
}

static unsigned long long *g_mlp_dbg = nullptr;   // diagnostic build only (tools/mb_mlp_fused.py): per wave [barrier, phase 1, GELU, phase 2, epilogue, tiles]
extern "C" SCP_API int scp_mlp_debug_buffer(unsigned long long *dev_buf) { g_mlp_dbg = dev_buf; return SCP_OK; }

static int g_mlp_num_cu = 0;

// X planes [M][ldx] (256 columns), W1 planes [1024][256], W2 planes [256][1024] (scp_split_weight_bf16 with Kpad = K),
// b1[1024], b2[256], optional residual fp32 [M][ldr], C fp32 [M][ldc].  Replaces intermediate.dense + GELU + output.dense + residual
// of a Swin block (swin_transformer.py:559-571); bit-identical to scp_linear_split (act 2, split output) + scp_linear_split (residual).
extern "C" SCP_API int scp_mlp_split_fused(const void *Xhi, const void *Xlo, int64_t ldx, const void *W1hi, const void *W1lo, const void *W2hi,
                                           const void *W2lo, const float *b1, const float *b2, const float *residual, int64_t ldr, float *C,
                                           int64_t ldc, int32_t M, void *stream) {
    if (!Xhi || !Xlo || !W1hi || !W1lo || !W2hi || !W2lo || !b1 || !b2 || !C || M <= 0 || (ldx & 7) || ldx < 256 || ldc < 256 || (ldc & 3) ||
        (residual && (ldr < 256 || (ldr & 3))) ||
        (((uintptr_t)Xhi | (uintptr_t)Xlo | (uintptr_t)W1hi | (uintptr_t)W1lo | (uintptr_t)W2hi | (uintptr_t)W2lo | (uintptr_t)C |
          (uintptr_t)residual | (uintptr_t)b1 | (uintptr_t)b2) & 15))
        return SCP_EINVAL;
    if (!g_mlp_num_cu) {
        int dev = 0;
        hipDeviceProp_t p;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipGetDeviceProperties(&p, dev));
        g_mlp_num_cu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    constexpr int LDS = 2 * M2STAGE + 65536;
    static bool configured = false;
    if (!configured) {
        HIP_TRY(hipFuncSetAttribute((const void *)mlp_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        HIP_TRY(hipFuncSetAttribute((const void *)mlp_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        configured = true;
    }
    MlpArgs a;
    a.Xhi = (const __bf16 *)Xhi; a.Xlo = (const __bf16 *)Xlo; a.ldx = ldx;
    a.W1hi = (const __bf16 *)W1hi; a.W1lo = (const __bf16 *)W1lo; a.W2hi = (const __bf16 *)W2hi; a.W2lo = (const __bf16 *)W2lo;
    a.b1 = b1; a.b2 = b2; a.res = residual; a.ldr = ldr; a.C = C; a.ldc = ldc; a.M = M;
    { static int wt = -1; if (wt < 0) { const char *e = getenv("SCP_WTILE"); wt = (e && e[0] == '0') ? 0 : 1; } a.wtiled = wt; }
    const int ntiles = (M + MBM - 1) / MBM;
    const unsigned grid = (unsigned)(ntiles < g_mlp_num_cu ? ntiles : g_mlp_num_cu);
    if (g_mlp_dbg) hipLaunchKernelGGL(mlp_fused_kernel<true>, dim3(grid), dim3(512), LDS, (hipStream_t)stream, a, g_mlp_dbg);
    else hipLaunchKernelGGL(mlp_fused_kernel<false>, dim3(grid), dim3(512), LDS, (hipStream_t)stream, a, (unsigned long long *)nullptr);
    LAUNCH_CHECK();
    return SCP_OK;
}
