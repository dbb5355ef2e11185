// Swin MLP in one kernel: C = GELU(X . W1^T + b1) . W2^T + b2 + R   (256 -> 1024 -> 256, swin_transformer.py:559-571), bf16x3 on
// split operands like gemm_split.hip, the 1024-wide hidden activation never leaves the CU (gfx950 / CDNA4).
//
// Workgroup = 8 waves (2 x 4) = 128 rows of X, persistent over row tiles.  The hidden dimension is processed in 8 chunks of 128:
//   phase 1 (8 k-steps of 32):  H^T[128 hidden x 128 rows] = W1c . X^T      wave: 32 hidden x 64 rows (A = W1 rows, B = X rows:
//                               the accumulator then holds four consecutive hidden units per register group = one 8-byte
//                               store into the A-operand image of phase 2)
//   GELU + bias + hi/lo split -> LDS image H[4 k-slabs][2 planes][128 rows][32] (same 64-byte rows and XOR swizzle as an A stage)
//   phase 2 (4 k-steps of 32):  Y[128 x 256] += H . W2c^T                   wave: 64 rows x 64 columns
// All global operands arrive by LDS-DMA into two 32 KiB stages (phase 1: X slab + W1 slab, phase 2: W2 slab), one step ahead.
// Products, their order and the k order are those of two scp_linear_split calls (fc1 with GELU and split output, fc2 with
// residual): results are bit-identical.  LDS: 3 x 32 KiB stages + 64 KiB H image (reused as the epilogue bounce) = 160 KiB.
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
};

#define MBM 128
#define MSTAGE 32768
#define MNST 3                                   // LDS stages: operands are requested two steps ahead (64 KiB in flight per CU)
#define MHOFF (MNST * MSTAGE)                    // H image: 4 slabs x (hi 8 KiB + lo 8 KiB)

__global__ __launch_bounds__(512, 2) void mlp_fused_kernel(const MlpArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;
    const int ntiles = (a.M + MBM - 1) / MBM;
    const int d_row = lane >> 2;
    const int d_q = (lane & 3) ^ ((lane >> 4) & 3);
    const int f_pos0 = (h ^ ((lane >> 2) & 3)) << 4;

    // step s of a tile: chunk c = s / 12, t = s % 12; t < 8: phase 1 k-step t, else phase 2 k-step t - 8
    auto issue = [&](int m0, int s, int gs) {
        char *base = smem + (gs % MNST) * MSTAGE;
        const int c = s / 12, t = s - 12 * c;
        if (t < 8) {
            const int k0 = t * 32 + 8 * d_q;
            int m = m0 + 16 * w + d_row;
            m = m < a.M ? m : a.M - 1;
            const int64_t xo = (int64_t)m * a.ldx + k0;
            mlp_dma16(a.Xhi + xo, base + w * 1024);
            mlp_dma16(a.Xlo + xo, base + 8192 + w * 1024);
            const int64_t wo = (int64_t)(c * 128 + 16 * w + d_row) * 256 + k0;
            mlp_dma16(a.W1hi + wo, base + 16384 + w * 1024);
            mlp_dma16(a.W1lo + wo, base + 24576 + w * 1024);
        } else {
            const int k0 = c * 128 + (t - 8) * 32 + 8 * d_q;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cr = w + 8 * j;
                const int64_t wo = (int64_t)(16 * cr + d_row) * 1024 + k0;
                mlp_dma16(a.W2hi + wo, base + cr * 1024);
                mlp_dma16(a.W2lo + wo, base + 16384 + cr * 1024);
            }
        }
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    // one continuous stream of steps across this workgroup's tiles: global step gs = 96 * (tile ordinal) + s uses stage gs % 3 and
    // is requested two steps ahead; every wave issues exactly four DMA instructions per step, so "step gs has landed" is
    // vmcnt <= 4 while a younger step is in flight (the counter retires in order: older epilogue loads / stores are covered)
    int gs0 = 0;
    issue(tile * MBM, 0, 0);
    issue(tile * MBM, 1, 1);
    auto ahead = [&](int m0, int s) {                                 // request step s + 2 of the stream; false if there is none
        if (s + 2 < 96) { issue(m0, s + 2, gs0 + s + 2); return true; }
        if (tile + (int)gridDim.x < ntiles) { issue((tile + (int)gridDim.x) * MBM, s + 2 - 96, gs0 + s + 2); return true; }
        return false;
    };
    auto landed = [&](int s) {   // barrier: step s has landed in every wave's view (given what has been requested after it)
        const bool younger = (s + 1 < 96) || (tile + (int)gridDim.x < ntiles);
        if (younger) SCP_BARRIER_DMA(4); else SCP_BARRIER_DMA(0);
    };

    for (; tile < ntiles; tile += gridDim.x, gs0 += 96) {
        const int m0 = tile * MBM;
        mf32x16 acc2[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
        mf32x16 acc1[2];

        for (int c = 0; c < 8; ++c) {
            // ---------------- phase 1: H^T chunk = W1c . X^T ----------------
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[j][r] = 0.f;
            for (int t = 0; t < 8; ++t) {
                const int s = c * 12 + t;
                landed(s);
                ahead(m0, s);
                const char *st = smem + ((gs0 + s) % MNST) * MSTAGE;
                const int ow = 16384 + (wn * 32 + col) * 64, ox = (wm * 64 + col) * 64;
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    const int po = f_pos0 ^ (kc * 32);
                    const mbf16x8 wh = *(const mbf16x8 *)(st + ow + po), wl = *(const mbf16x8 *)(st + ow + 8192 + po);
                    mbf16x8 xh[2], xl[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        xh[j] = *(const mbf16x8 *)(st + ox + j * 2048 + po);
                        xl[j] = *(const mbf16x8 *)(st + ox + 8192 + j * 2048 + po);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl[j], acc1[j], 0, 0, 0);   // x_lo . w_hi
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh[j], acc1[j], 0, 0, 0);   // x_hi . w_lo
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh[j], acc1[j], 0, 0, 0);   // x_hi . w_hi
                }
            }
            // ---------------- bias + GELU + split -> H image (k-slab wn) ----------------
            {
                char *hb = smem + MHOFF + wn * 16384;
                // the 32 biases of this wave's hidden tile through the scalar path (wave-uniform address -> s_load: a vector load
                // here would make the compiler wait vmcnt(0) for it, which also drains the operand DMAs in flight)
                typedef const __attribute__((address_space(4))) float *m_const_ptr_t;   // constant address space: scalar loads
                m_const_ptr_t b1p = (m_const_ptr_t)(uintptr_t)(a.b1 + c * 128 + wn * 32);
                float sb[32];
#pragma unroll
                for (int e = 0; e < 32; ++e) sb[e] = b1p[e];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    mf32x4 bv;                                                 // hidden units 8 g + 4 h + 0..3 of this wave's 32
#pragma unroll
                    for (int u = 0; u < 4; ++u) bv[u] = h ? sb[8 * g + 4 + u] : sb[8 * g + u];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int xr = wm * 64 + 32 * j + col;
                        mbf16x4 hi, lo;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float y = mlp_gelu(acc1[j][4 * g + u] + bv[u]);
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
            // ---------------- phase 2: Y += H . W2c^T ----------------
            for (int t = 8; t < 12; ++t) {
                const int s = c * 12 + t;
                landed(s);                                          // at t = 8 this also publishes the H image
                ahead(m0, s);
                const char *st = smem + ((gs0 + s) % MNST) * MSTAGE;
                const char *hs = smem + MHOFF + (t - 8) * 16384;
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
            }
        }
        __syncthreads();                                            // every wave is done with the H image (the next tile's first two
                                                                    // steps are already streaming into their stages)

        // ---------------- epilogue: + b2, + residual, fp32 rows (bounce through this wave's 8 KiB of the H region) ----------------
        float *stg = (float *)(smem + MHOFF + w * 8192);
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
        // the first barrier of the next tile orders these LDS reads before that tile's H writes (eight steps later anyway)
    }
}

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
    constexpr int LDS = MNST * MSTAGE + 65536;
    static bool configured = false;
    if (!configured) {
        HIP_TRY(hipFuncSetAttribute((const void *)mlp_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        configured = true;
    }
    MlpArgs a;
    a.Xhi = (const __bf16 *)Xhi; a.Xlo = (const __bf16 *)Xlo; a.ldx = ldx;
    a.W1hi = (const __bf16 *)W1hi; a.W1lo = (const __bf16 *)W1lo; a.W2hi = (const __bf16 *)W2hi; a.W2lo = (const __bf16 *)W2lo;
    a.b1 = b1; a.b2 = b2; a.res = residual; a.ldr = ldr; a.C = C; a.ldc = ldc; a.M = M;
    const int ntiles = (M + MBM - 1) / MBM;
    const unsigned grid = (unsigned)(ntiles < g_mlp_num_cu ? ntiles : g_mlp_num_cu);
    hipLaunchKernelGGL(mlp_fused_kernel, dim3(grid), dim3(512), LDS, (hipStream_t)stream, a);
    LAUNCH_CHECK();
    return SCP_OK;
}
