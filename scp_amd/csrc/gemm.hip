// Dense layer on bf16 MFMA with fp32-class accuracy: C = epilogue(A . W^T), "bf16x3" split (gfx950 / CDNA4).
//
// Every fp32 operand x is written as hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 significant bits together) and
//     a.w  ~=  a_hi.w_hi + a_hi.w_lo + a_lo.w_hi          (the dropped lo.lo term is 2^-16 relative)
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16: three MFMAs at 16x the fp32-MFMA rate instead of one fp32 MFMA, i.e. a
// ~5x higher ceiling than v_mfma_f32_32x32x2_f32 for the same product.  Measured end effect on the EHEM logits
// (oracle emulation + tests/test_gpu_model.py): max |dlogit| 4e-5, tolerance 1e-3.  kNN distances are NOT computed this way.
//
// Layout: W is torch's Linear weight [N][K] (K contiguous), pre-split once into two bf16 planes padded to [Npad][Kpad]
// (scp_amd/ops.py); A is fp32 [M][lda] and is split on the fly while staging.  Workgroup = 8 waves = 128 x 128 tile of C,
// BK = 32; a wave owns 64 x 32 = 2 MFMA tiles (16 waves per CU hide the staging latency; measured).  LDS holds four bf16 planes [128][32(+8 pad)] (80-byte rows: the eight
// 16-byte fragment reads of a 16-lane group land in distinct bank slots).  One LDS stage + register prefetch of the next
// k-tile; 2-3 workgroups per CU hide the two barriers per k-tile.
// Epilogue (fused): + bias[n], activation (none | LeakyReLU(0.01) | exact-erf GELU | ReLU), + residual[m][n].
#include "scp_internal.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define BM 128
#define BN 128
#define BK 32
#define LDP 40   // bf16 elements per LDS row (32 + 8 pad) = 80 bytes

enum { ACT_NONE = 0, ACT_LEAKY = 1, ACT_GELU = 2, ACT_RELU = 3 };

template <int act>
__device__ __forceinline__ float apply_act(float y) {
    if (act == ACT_LEAKY) return y > 0.f ? y : 0.01f * y;
    if (act == ACT_GELU) return scp_gelu(y);     // exact-erf GELU to 4.4e-7 absolute: scp_internal.h (round 5; the degree-12 erf polynomial of rounds 1 - 4 is gone)
    if (act == ACT_RELU) return y > 0.f ? y : 0.f;
    return y;
}

// F16 = true is the "f16x3" form of the same kernel: planes are IEEE half (11 significant bits each, 22 together, against 16 for
// bf16) and every row of A and of W is first multiplied by a power of two that puts its largest magnitude into [2^13, 2^14)
// (row_scale_kernel / split_weight_f16_kernel), so that neither plane overflows and the low plane of the elements that matter
// stays normal; the accumulator is multiplied back by the two inverse scales in the epilogue (powers of two: exact).  The
// dropped lo.lo term and the plane rounding are ~2^-22 relative - the error class of an fp32 FMA chain over K = 600 - at the
// MFMA rate of the bf16 form.  Used by OctAttention, whose sqrt(600)-scaled embeddings leave bf16x3 short of the 1e-3 logit
// tolerance.  asc / iasc: scale and inverse scale per row of A; iwsc: inverse scale per row of W.
template <int ACT, bool F16>
__global__ __launch_bounds__(512, 4) void gemm_bf16x3_kernel(const float *__restrict__ A, int64_t lda, const __bf16 *__restrict__ Whi,
                                                            const __bf16 *__restrict__ Wlo, int Kpad, const float *__restrict__ bias,
                                                            const float *__restrict__ res, int64_t ldr, float *__restrict__ C, int64_t ldc,
                                                            int M, int N, int K, const float *__restrict__ asc,
                                                            const float *__restrict__ iasc, const float *__restrict__ iwsc, int wtiled) {
    // two LDS stages x four 16-bit planes (A hi, A lo, B hi, B lo) = 2 x 40 KiB: exactly two workgroups per CU
    __shared__ __attribute__((aligned(16))) __bf16 lds[2][4][BM * LDP];

    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, col = lane & 31, h = lane >> 5;
    const int wm = w >> 2, wn = w & 3;   // 8 waves: 2 (M) x 4 (N), a wave owns 64 x 32 = two 32x32 MFMA tiles
    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs, so give every XCD a contiguous run of tiles with the
    // N tiles of one 128-row stripe back to back - the A stripe (128 x K fp32) is then fetched once per XCD L2, not once per tile.
    const int ntn = (N + BN - 1) / BN, nblk = gridDim.x;
    int lin = blockIdx.x;
    if ((nblk & 7) == 0) lin = (lin & 7) * (nblk >> 3) + (lin >> 3);
    const int m0 = (lin / ntn) * BM, n0 = (lin % ntn) * BN;
    const int nk = Kpad / BK;

    // staging assignment (512 threads): A tile = 128 rows x 8 float4; thread handles rows (tid>>3) + 64*i, float4 column tid&7
    const int a_r = tid >> 3, a_c = tid & 7;
    // B planes = 128 rows x 4 chunks of 8 bf16 (16 B); thread handles row tid>>2, chunk tid&3
    const int b_r = tid >> 2, b_c = tid & 3;

    // two register sets: loads are issued TWO k-tiles ahead of their use (one tile ahead does not cover the L2/HBM latency
    // with only two workgroups per CU)
    f32x4 pa2[2][2];
    bf16x8 pbh2[2][1], pbl2[2][1];
    float rsc[2] = {1.f, 1.f};                           // F16: scale of this thread's two staged rows
    if (F16) {
#pragma unroll
        for (int i = 0; i < 2; ++i) { const int m = m0 + a_r + 64 * i; rsc[i] = asc[m < M ? m : M - 1]; }
    }
    auto issue = [&](int kt, auto &pa, auto &pbh, auto &pbl) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // unconditional load from a clamped address + select: a branch around the load would make hipcc lose count of the
            // outstanding loads and wait vmcnt(0) every iteration (no prefetch at all)
            const int m = m0 + a_r + 64 * i, k = k0 + 4 * a_c;
            const int mc = m < M ? m : M - 1, kc = k < K ? k : K - 4;             // K % 4 == 0
            pa[i] = *(const f32x4 *)(A + (int64_t)mc * lda + kc);                 // masked when it is consumed (commit): a select
                                                                                  // here would make the wave wait for the load at once
        }
        {
            // planes are padded: always in range.  wtiled: the tiled planes of scp_tile_weight_bf16 (1 KiB blocks [16-row group][k-slab], chunk
            // q of row r at position q ^ ((r >> 2) & 3)): a wavefront's 64 loads then cover whole cache lines
            const int wr = n0 + b_r;
            const int64_t off = wtiled ? (((int64_t)(wr >> 4) * (Kpad >> 5) + kt) * 64 + (wr & 15) * 4 + (b_c ^ ((wr >> 2) & 3))) * 8
                                       : (int64_t)wr * Kpad + k0 + 8 * b_c;
            pbh[0] = *(const bf16x8 *)(Whi + off);
            pbl[0] = *(const bf16x8 *)(Wlo + off);
        }
    };
    auto commit = [&](int st, int kt, auto &pa, auto &pbh, auto &pbl) {
        __bf16 *sAh = lds[st][0], *sAl = lds[st][1], *sBh = lds[st][2], *sBl = lds[st][3];
        // the prefetched registers become visible HERE: without this the scheduler hoists the conversions of the next step's
        // commit above the barrier, and with them the wait for loads that should stay in flight across it
        asm volatile("" : "+v"(pa[0]), "+v"(pa[1]), "+v"(pbh[0]), "+v"(pbl[0]));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (!((m0 + a_r + 64 * i < M) && (kt * BK + 4 * a_c < K))) pa[i] = (f32x4){0.f, 0.f, 0.f, 0.f};   // rows / columns beyond the matrix
            const int o = (a_r + 64 * i) * LDP + 4 * a_c;
            if (F16) {
                f16x4 hi, lo;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float x = pa[i][u] * rsc[i];
                    const _Float16 hh = (_Float16)x;
                    hi[u] = hh;
                    lo[u] = (_Float16)(x - (float)hh);
                }
                *(f16x4 *)(sAh + o) = hi;
                *(f16x4 *)(sAl + o) = lo;
            } else {
                bf16x4 hi, lo;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const __bf16 hh = (__bf16)pa[i][u];
                    hi[u] = hh;
                    lo[u] = (__bf16)(pa[i][u] - (float)hh);
                }
                *(bf16x4 *)(sAh + o) = hi;
                *(bf16x4 *)(sAl + o) = lo;
            }
        }
        {
            const int o = b_r * LDP + 8 * b_c;
            *(bf16x8 *)(sBh + o) = pbh[0];
            *(bf16x8 *)(sBl + o) = pbl[0];
        }
    };

    f32x16 acc[2][1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

    issue(0, pa2[0], pbh2[0], pbl2[0]);
    issue(nk > 1 ? 1 : 0, pa2[1], pbh2[1], pbl2[1]);
    commit(0, 0, pa2[0], pbh2[0], pbl2[0]);
    __syncthreads();
    // one k-step; `rf*` = the register set that held tile kt (in LDS since the previous step): refilled with tile kt + 2;
    // `cm*` = the set holding tile kt + 1 (loaded during the previous step): written to the other LDS stage after the MFMAs.
    // The loop is unrolled by two so that both sets are compile-time registers: the wait in front of the commit is then a
    // counted one (the four younger loads of tile kt + 2 stay in flight), and the raw barrier keeps them in flight too
    // (__syncthreads() would drain them: vmcnt(0) in front of s_barrier, scp_internal.h).
    auto step = [&](int kt, auto &rfa, auto &rfh, auto &rfl, auto &cma, auto &cmh, auto &cml) {
        issue(kt + 2 < nk ? kt + 2 : nk - 1, rfa, rfh, rfl);   // unconditional (the last two steps re-read the last tile): a branch
                                                                // around the loads would make the compiler copy - and so wait for -
                                                                // the loaded registers at the join
        const __bf16 *sAh = lds[kt & 1][0], *sAl = lds[kt & 1][1], *sBh = lds[kt & 1][2], *sBl = lds[kt & 1][3];
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            bf16x8 ah[2], al[2], bh[1], bl[1];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ao = (wm * 64 + t * 32 + col) * LDP + kc * 16 + 8 * h;
                ah[t] = *(const bf16x8 *)(sAh + ao);
                al[t] = *(const bf16x8 *)(sAl + ao);
            }
            {
                const int bo = (wn * 32 + col) * LDP + kc * 16 + 8 * h;
                bh[0] = *(const bf16x8 *)(sBh + bo);
                bl[0] = *(const bf16x8 *)(sBl + bo);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 1; ++j) {
                    if (F16) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16((f16x8)al[i], (f16x8)bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16((f16x8)ah[i], (f16x8)bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16((f16x8)ah[i], (f16x8)bh[j], acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
                }
        }
        if (kt + 1 < nk) commit((kt + 1) & 1, kt + 1, cma, cmh, cml);
        SCP_BARRIER_DMA(4);
    };
    for (int kt = 0; kt < nk; kt += 2) {
        step(kt, pa2[0], pbh2[0], pbl2[0], pa2[1], pbh2[1], pbl2[1]);
        if (kt + 1 < nk) step(kt + 1, pa2[1], pbh2[1], pbl2[1], pa2[0], pbh2[0], pbl2[0]);
    }

    // epilogue: the accumulators hold 4 B per lane per row (lane = column n); bounce each wave's 64 x 64 tile through its private
    // slice of the (now idle) LDS so that the residual loads and the stores move 16 B per lane, 4 rows x 256 B per instruction.
    // (the loop's last barrier already retired every read of the operand stages)
    constexpr int LDE = 36;   // floats per staged row: 32 + 4 (keeps float4 reads aligned, spreads banks)
    float *stg = (float *)&lds[0][0][0] + w * (64 * LDE);
    {
        const int n = n0 + wn * 32 + col;
        const float bv = (bias && n < N) ? bias[n] : 0.f;
        const float wsc = F16 ? iwsc[n] : 1.f;               // planes are padded to Npad rows: always in range
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float t = acc[i][0][r];
                if (F16) { const int m = m0 + wm * 64 + ml; t = (t * iasc[m < M ? m : M - 1]) * wsc; }
                stg[ml * LDE + col] = apply_act<ACT>(t + bv);
            }
    }
    __syncthreads();
    const int c4 = (lane & 7) * 4, rsub = lane >> 3;   // 8 lanes x 16 B per row, 8 rows per instruction
    const int nb = n0 + wn * 32 + c4;
    const bool vec_ok = ((ldc & 3) == 0) && (!res || (ldr & 3) == 0) && (nb + 3 < N) && (((uintptr_t)C & 15) == 0) &&
                        (!res || ((uintptr_t)res & 15) == 0);
#pragma unroll 4
    for (int it = 0; it < 8; ++it) {
        const int ml = it * 8 + rsub;
        const int m = m0 + wm * 64 + ml;
        if (m >= M) continue;
        f32x4 y = *(const f32x4 *)(stg + ml * LDE + c4);
        if (vec_ok) {
            if (res) { const f32x4 rr = *(const f32x4 *)(res + (int64_t)m * ldr + nb); y += rr; }
            *(f32x4 *)(C + (int64_t)m * ldc + nb) = y;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (nb + u < N) C[(int64_t)m * ldc + nb + u] = y[u] + (res ? res[(int64_t)m * ldr + nb + u] : 0.f);
        }
    }
}

static int gemm_wtiled() { static int wt = -1; if (wt < 0) { const char *e = getenv("SCP_WTILE"); wt = (e && e[0] == '0') ? 0 : 1; } return wt; }

// split an fp32 weight [N][K] into zero-padded bf16 planes [Npad][Kpad]
__global__ void split_weight_kernel(const float *__restrict__ W, int N, int K, int Npad, int Kpad, __bf16 *__restrict__ hi,
                                    __bf16 *__restrict__ lo) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)Npad * Kpad) return;
    const int n = (int)(i / Kpad), k = (int)(i - (int64_t)n * Kpad);
    const float v = (n < N && k < K) ? W[(int64_t)n * K + k] : 0.f;
    const __bf16 hh = (__bf16)v;
    hi[i] = hh;
    lo[i] = (__bf16)(v - (float)hh);
}

extern "C" SCP_API int scp_split_weight_bf16(const float *W, int32_t N, int32_t K, int32_t Npad, int32_t Kpad, void *hi, void *lo, void *stream) {
    if (!W || !hi || !lo || N <= 0 || K <= 0 || Npad < N || Kpad < K || (Npad % BN) || (Kpad % BK)) return SCP_EINVAL;
    const int64_t tot = (int64_t)Npad * Kpad;
    hipLaunchKernelGGL(split_weight_kernel, dim3((unsigned)cdiv64(tot, 256)), dim3(256), 0, (hipStream_t)stream, W, N, K, Npad, Kpad,
                       (__bf16 *)hi, (__bf16 *)lo);
    LAUNCH_CHECK();
    return SCP_OK;
}

extern "C" SCP_API int scp_linear_bf16x3(const float *A, int64_t lda, const void *Whi, const void *Wlo, int32_t Kpad, const float *bias,
                                         const float *residual, int64_t ldr, float *C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                                         int32_t act, void *stream) {
    if (!A || !Whi || !Wlo || !C || M <= 0 || N <= 0 || K <= 0 || (K & 3) || (lda & 3) || Kpad < K || (Kpad % BK) || act < 0 || act > 3 ||
        ((uintptr_t)A & 15) || lda < K || ldc < N || (residual && ldr < N))
        return SCP_EINVAL;
    const dim3 grid((unsigned)(((N + BN - 1) / BN) * ((M + BM - 1) / BM)));
    SCP_PROF(SCP_PROF_GEMM_ROWS, stream, 2.0 * M * (double)N * K);
#define GO(ACT) hipLaunchKernelGGL((gemm_bf16x3_kernel<ACT, false>), grid, dim3(512), 0, (hipStream_t)stream, A, lda, (const __bf16 *)Whi, \
                              (const __bf16 *)Wlo, Kpad, bias, residual, ldr, C, ldc, M, N, K, nullptr, nullptr, nullptr, gemm_wtiled())
    switch (act) { case ACT_LEAKY: GO(ACT_LEAKY); break; case ACT_GELU: GO(ACT_GELU); break; case ACT_RELU: GO(ACT_RELU); break; default: GO(ACT_NONE); }
#undef GO
    LAUNCH_CHECK();
    return SCP_OK;
}

// ---- f16x3: row scales, weight planes, launch ------------------------------------------------------------------------------------
#define pow2_scale scp_pow2_scale       // scp_internal.h

// A [M][lda] (K % 4 == 0) -> scale and inverse scale per row.  A wavefront takes four consecutive rows and issues all their loads
// before it reduces the first (one row per wavefront keeps too few bytes in flight to reach the bandwidth of HBM).
__global__ __launch_bounds__(256) void row_scale_kernel(const float *__restrict__ A, int64_t lda, int M, int K, float *__restrict__ sc,
                                                       float *__restrict__ isc) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    if (row0 >= M) return;
    float mx[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 4 * lane; k < K; k += 256) {
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *(const f32x4 *)(A + (int64_t)(row0 + i < M ? row0 + i : M - 1) * lda + k);
#pragma unroll
        for (int i = 0; i < 4; ++i) mx[i] = fmaxf(fmaxf(mx[i], fmaxf(fabsf(v[i][0]), fabsf(v[i][1]))), fmaxf(fabsf(v[i][2]), fabsf(v[i][3])));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx[i] = fmaxf(mx[i], __shfl_xor(mx[i], o));
        if (lane == 0 && row0 + i < M) { float s, is; pow2_scale(mx[i], s, is); sc[row0 + i] = s; isc[row0 + i] = is; }
    }
}

// one wavefront per row of W [N][K] -> scaled f16 planes [Npad][Kpad] (zero padded) + inverse row scale [Npad] (1 on padding)
__global__ __launch_bounds__(256) void split_weight_f16_kernel(const float *__restrict__ W, int N, int K, int Npad, int Kpad,
                                                              _Float16 *__restrict__ hi, _Float16 *__restrict__ lo, float *__restrict__ isc) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= Npad) return;
    float mx = 0.f;
    if (row < N)
        for (int k = lane; k < K; k += 64) mx = fmaxf(mx, fabsf(W[(int64_t)row * K + k]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float s, is;
    pow2_scale(mx, s, is);
    if (lane == 0) isc[row] = is;
    for (int k = lane; k < Kpad; k += 64) {
        const float x = (row < N && k < K) ? W[(int64_t)row * K + k] * s : 0.f;
        const _Float16 hh = (_Float16)x;
        hi[(int64_t)row * Kpad + k] = hh;
        lo[(int64_t)row * Kpad + k] = (_Float16)(x - (float)hh);
    }
}

extern "C" SCP_API int scp_split_weight_f16(const float *W, int32_t N, int32_t K, int32_t Npad, int32_t Kpad, void *hi, void *lo,
                                            float *inv_scale, void *stream) {
    if (!W || !hi || !lo || !inv_scale || N <= 0 || K <= 0 || Npad < N || Kpad < K || (Npad % BN) || (Kpad % BK)) return SCP_EINVAL;
    hipLaunchKernelGGL(split_weight_f16_kernel, dim3((unsigned)((Npad + 3) / 4)), dim3(256), 0, (hipStream_t)stream, W, N, K, Npad, Kpad,
                       (_Float16 *)hi, (_Float16 *)lo, inv_scale);
    LAUNCH_CHECK();
    return SCP_OK;
}

// A [M][lda] fp32 (K % 4 == 0, K <= 1024) -> what the f16x3 kernel would stage for these rows, once, in HBM: scale / inverse scale per
// row and the two IEEE-half planes of the scaled row ([M][ldp], columns K .. Kp zero) - the operand format of scp_linear_split_f16.
// Same arithmetic as gemm_bf16x3_kernel<.., true>'s commit (x = a * scale; hi = half(x); lo = half(x - hi)): a layer fed with these
// planes gives the bits of the layer fed with the fp32 rows.  Four rows per wavefront, all loads in flight before the first reduction.
template <int NIT>
__global__ __launch_bounds__(256) void split_rows_f16_kernel(const float *__restrict__ A, int64_t lda, int M, int K, int Kp, _Float16 *__restrict__ hi,
                                                            _Float16 *__restrict__ lo, int64_t ldp, float *__restrict__ sc, float *__restrict__ isc) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    if (row0 >= M) return;
    f32x4 v[NIT][4];
    float mx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int k = 4 * lane + 256 * it;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            v[it][i] = k < K ? *(const f32x4 *)(A + (int64_t)(row0 + i < M ? row0 + i : M - 1) * lda + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            mx[i] = fmaxf(fmaxf(mx[i], fmaxf(fabsf(v[it][i][0]), fabsf(v[it][i][1]))), fmaxf(fabsf(v[it][i][2]), fabsf(v[it][i][3])));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx[i] = fmaxf(mx[i], __shfl_xor(mx[i], o));
        if (row0 + i >= M) continue;
        float s, is;
        pow2_scale(mx[i], s, is);
        if (lane == 0) { sc[row0 + i] = s; isc[row0 + i] = is; }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int k = 4 * lane + 256 * it;
            if (k >= Kp) continue;
            f16x4 h4, l4;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float x = v[it][i][u] * s;
                const _Float16 hh = (_Float16)x;
                h4[u] = hh;
                l4[u] = (_Float16)(x - (float)hh);
            }
            *(f16x4 *)(hi + (int64_t)(row0 + i) * ldp + k) = h4;
            *(f16x4 *)(lo + (int64_t)(row0 + i) * ldp + k) = l4;
        }
    }
}

extern "C" SCP_API int scp_split_rows_f16(const float *A, int64_t lda, int32_t M, int32_t K, void *hi, void *lo, int64_t ldp, float *scale,
                                          float *inv_scale, void *stream) {
    const int Kp = (K + 31) & ~31;
    if (!A || !hi || !lo || !scale || !inv_scale || M <= 0 || K <= 0 || (K & 3) || K > 1024 || (lda & 3) || lda < K || ldp < Kp || (ldp & 7) ||
        ((uintptr_t)A & 15) || (((uintptr_t)hi | (uintptr_t)lo) & 15))
        return SCP_EINVAL;
    const dim3 grid((unsigned)((M + 15) / 16));
    SCP_PROF(SCP_PROF_SPLIT_ROWS, stream, 8.0 * M * (double)K);
#define GOR(N_) hipLaunchKernelGGL((split_rows_f16_kernel<N_>), grid, dim3(256), 0, (hipStream_t)stream, A, lda, M, K, Kp, (_Float16 *)hi, (_Float16 *)lo, \
                                   ldp, scale, inv_scale)
    if (Kp <= 256) GOR(1); else if (Kp <= 512) GOR(2); else if (Kp <= 768) GOR(3); else GOR(4);
#undef GOR
    LAUNCH_CHECK();
    return SCP_OK;
}

// row scales from row maxima that a producing kernel's epilogue took (scp_linear_split_f16_max): bit patterns of max |row| -> scale, 1 / scale
__global__ __launch_bounds__(256) void row_scale_from_max_kernel(const unsigned *__restrict__ mxb, int M, float *__restrict__ sc, float *__restrict__ isc) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float s, is;
    pow2_scale(__uint_as_float(mxb[m]), s, is);
    sc[m] = s; isc[m] = is;
}

extern "C" SCP_API int scp_row_scale_from_max(const uint32_t *row_max, int32_t M, float *scale, float *inv_scale, void *stream) {
    if (!row_max || !scale || !inv_scale || M <= 0) return SCP_EINVAL;
    hipLaunchKernelGGL(row_scale_from_max_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned *)row_max, M, scale, inv_scale);
    LAUNCH_CHECK();
    return SCP_OK;
}

// row scales of an activation on their own (scale [M], 1 / scale [M]): several layers reading the SAME rows (OctAttention's key / value /
// query projections of one embedding tensor) share them through scp_linear_f16x3_scaled
extern "C" SCP_API int scp_row_scale_f16(const float *A, int64_t lda, int32_t M, int32_t K, float *scale, float *inv_scale, void *stream) {
    if (!A || !scale || !inv_scale || M <= 0 || K <= 0 || (K & 3) || (lda & 3) || lda < K || ((uintptr_t)A & 15)) return SCP_EINVAL;
    hipLaunchKernelGGL(row_scale_kernel, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, (hipStream_t)stream, A, lda, M, K, scale, inv_scale);
    LAUNCH_CHECK();
    return SCP_OK;
}

static int linear_f16x3_launch(const float *A, int64_t lda, const void *Whi, const void *Wlo, const float *w_inv_scale, int32_t Kpad,
                               const float *bias, const float *residual, int64_t ldr, float *C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                               int32_t act, float *sc, float *isc, bool compute_scales, void *stream) {
    if (!A || !Whi || !Wlo || !w_inv_scale || !C || !sc || !isc || M <= 0 || N <= 0 || K <= 0 || (K & 3) || (lda & 3) || Kpad < K ||
        (Kpad % BK) || act < 0 || act > 3 || ((uintptr_t)A & 15) || lda < K || ldc < N || (residual && ldr < N))
        return SCP_EINVAL;
    if (compute_scales) hipLaunchKernelGGL(row_scale_kernel, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, (hipStream_t)stream, A, lda, M, K, sc, isc);
    const dim3 grid((unsigned)(((N + BN - 1) / BN) * ((M + BM - 1) / BM)));
    SCP_PROF(SCP_PROF_GEMM_ROWS, stream, 2.0 * M * (double)N * K);
#define GO(ACT) hipLaunchKernelGGL((gemm_bf16x3_kernel<ACT, true>), grid, dim3(512), 0, (hipStream_t)stream, A, lda, (const __bf16 *)Whi, \
                              (const __bf16 *)Wlo, Kpad, bias, residual, ldr, C, ldc, M, N, K, sc, isc, w_inv_scale, gemm_wtiled())
    switch (act) { case ACT_LEAKY: GO(ACT_LEAKY); break; case ACT_GELU: GO(ACT_GELU); break; case ACT_RELU: GO(ACT_RELU); break; default: GO(ACT_NONE); }
#undef GO
    LAUNCH_CHECK();
    return SCP_OK;
}

extern "C" SCP_API int scp_linear_f16x3(const float *A, int64_t lda, const void *Whi, const void *Wlo, const float *w_inv_scale,
                                        int32_t Kpad, const float *bias, const float *residual, int64_t ldr, float *C, int64_t ldc,
                                        int32_t M, int32_t N, int32_t K, int32_t act, float *row_scale_ws, void *stream) {
    if (!row_scale_ws) return SCP_EINVAL;
    return linear_f16x3_launch(A, lda, Whi, Wlo, w_inv_scale, Kpad, bias, residual, ldr, C, ldc, M, N, K, act, row_scale_ws, row_scale_ws + M, true,
                               stream);      // workspace: 2 M floats
}

// the same with the row scales of A given (scp_row_scale_f16 on exactly these rows)
extern "C" SCP_API int scp_linear_f16x3_scaled(const float *A, int64_t lda, const void *Whi, const void *Wlo, const float *w_inv_scale,
                                               int32_t Kpad, const float *bias, const float *residual, int64_t ldr, float *C, int64_t ldc,
                                               int32_t M, int32_t N, int32_t K, int32_t act, const float *scale, const float *inv_scale,
                                               void *stream) {
    return linear_f16x3_launch(A, lda, Whi, Wlo, w_inv_scale, Kpad, bias, residual, ldr, C, ldc, M, N, K, act, (float *)scale, (float *)inv_scale,
                               false, stream);
}

// ================================================================================================================
// Exact fp32 dense layer, batch-invariant: C = act(A . W^T + bias) on v_mfma_f32_32x32x2_f32.
// Used where bf16x3 is not wanted or not possible: the small-K layers (K = 3, 16) and everything that feeds a kNN search
// (edge-conv u/v products, mlp2), so that those features are plain fp32 FMA chains in k order - the same arithmetic as the
// reference's CPU matmul - AND independent of how many rows share the launch: the encoder (one packed launch per frame) and the
// decoder (one window at a time) must produce bit-identical features, which a library GEMM that picks its kernel by problem
// size does not guarantee.  Workgroup = 4 waves = 128 rows x 64 columns; K is staged 32 at a time into LDS de-interleaved as
// [row][even k | odd k] so that MFMA lane half h reads k = 2s + h contiguously (k order preserved); zero padding is exact.
#define FBM 128
#define FBN 64
#define FBK 32
#define FLD 36   // floats per staged row (32 + 4): 16-byte aligned rows, conflict-free ds_read_b128

template <int ACT>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ W,
                                                         const float *__restrict__ bias, float *__restrict__ C, int64_t ldc, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) float sA[FBM * FLD];
    __shared__ __attribute__((aligned(16))) float sB[FBN * FLD];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, col = lane & 31, h = lane >> 5;
    const int ntn = (N + FBN - 1) / FBN;
    const int m0 = (blockIdx.x / ntn) * FBM, n0 = (blockIdx.x % ntn) * FBN;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int nk = (K + FBK - 1) / FBK;
    // staging: element (row, k) -> [row][(k & 1) * 16 + (k >> 1)].  Fast path (K % 4 == 0, 16-byte aligned rows): every thread
    // moves 16-byte pieces - 4 of A and 2 of W per k-tile - and the pieces of tile kt + 1 are loaded into registers BEFORE the
    // MFMAs of tile kt (written to LDS after them), so the global latency hides behind the workgroup's own compute.
    const bool vec = ((K & 3) == 0) && ((lda & 3) == 0) && ((((uintptr_t)A | (uintptr_t)W) & 15) == 0);
    const int pr = tid >> 3, pc = (tid & 7) * 4;          // piece p = tid + 256 i: row pr + 32 i, k columns pc .. pc + 3
    f32x4 ra[4], rb[2];
    auto load_tile = [&](int kt) {
        const int kk = kt * FBK + pc;
        const bool kin = kk < K;                           // K % 4 == 0: a piece is entirely inside or outside
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + pr + 32 * i;
            const int mc = m < M ? m : M - 1, kc = kin ? kk : 0;
            const f32x4 v = *(const f32x4 *)(A + (int64_t)mc * lda + kc);
            ra[i] = (kin && m < M) ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = n0 + pr + 32 * i;
            const int nc = n < N ? n : N - 1, kc = kin ? kk : 0;
            const f32x4 v = *(const f32x4 *)(W + (int64_t)nc * K + kc);
            rb[i] = (kin && n < N) ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_tile = [&]() {
        const int o = pc >> 1;                             // k = pc + u -> column (u & 1) * 16 + o + (u >> 1)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float *d = sA + (pr + 32 * i) * FLD;
            *(float2 *)(d + o) = make_float2(ra[i][0], ra[i][2]);
            *(float2 *)(d + 16 + o) = make_float2(ra[i][1], ra[i][3]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float *d = sB + (pr + 32 * i) * FLD;
            *(float2 *)(d + o) = make_float2(rb[i][0], rb[i][2]);
            *(float2 *)(d + 16 + o) = make_float2(rb[i][1], rb[i][3]);
        }
    };
    if (vec) load_tile(0);
    for (int kt = 0; kt < nk; ++kt) {
        const int k0 = kt * FBK;
        __syncthreads();
        if (vec) {
            store_tile();
        } else {
        for (int e = tid; e < FBM * FBK; e += 256) {
            const int r = e >> 5, k = e & 31;
            const int m = m0 + r, kk = k0 + k;
            sA[r * FLD + (k & 1) * 16 + (k >> 1)] = (m < M && kk < K) ? A[(int64_t)m * lda + kk] : 0.f;
        }
        for (int e = tid; e < FBN * FBK; e += 256) {
            const int r = e >> 5, k = e & 31;
            const int n = n0 + r, kk = k0 + k;
            sB[r * FLD + (k & 1) * 16 + (k >> 1)] = (n < N && kk < K) ? W[(int64_t)n * K + kk] : 0.f;
        }
        }
        __syncthreads();
        if (vec && kt + 1 < nk) load_tile(kt + 1);
        const float *ar = sA + (w * 32 + col) * FLD + h * 16;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 a = *(const f32x4 *)(ar + 4 * g);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 b = *(const f32x4 *)(sB + (j * 32 + col) * FLD + h * 16 + 4 * g);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc[j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + j * 32 + col;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + w * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < M) C[(int64_t)m * ldc + n] = apply_act<ACT>(acc[j][r] + bv);
        }
    }
}

extern "C" SCP_API int scp_linear_f32(const float *A, int64_t lda, const float *W, const float *bias, float *C, int64_t ldc, int32_t M, int32_t N,
                                      int32_t K, int32_t act, void *stream) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0 || lda < K || ldc < N || act < 0 || act > 3) return SCP_EINVAL;
    const dim3 grid((unsigned)(((N + FBN - 1) / FBN) * ((M + FBM - 1) / FBM)));
    SCP_PROF(SCP_PROF_GEMM_F32, stream, 2.0 * M * (double)N * K);
#define GOF(ACT) hipLaunchKernelGGL(gemm_f32_kernel<ACT>, grid, dim3(256), 0, (hipStream_t)stream, A, lda, W, bias, C, ldc, M, N, K)
    switch (act) { case ACT_LEAKY: GOF(ACT_LEAKY); break; case ACT_GELU: GOF(ACT_GELU); break; case ACT_RELU: GOF(ACT_RELU); break; default: GOF(ACT_NONE); }
#undef GOF
    LAUNCH_CHECK();
    return SCP_OK;
}
