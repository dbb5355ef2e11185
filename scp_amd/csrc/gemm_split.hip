// Dense layer on bf16 MFMA with fp32-class accuracy, BOTH operands pre-split ("bf16x3", gfx950 / CDNA4).
//
//     C = epilogue(A . W^T),   a.w ~= a_lo.w_hi + a_hi.w_lo + a_hi.w_hi   (fp32 accumulate, same order as gemm.hip)
//
// An activation travels between the kernels of the context model as two bf16 planes hi = bf16(x), lo = bf16(x - hi) (the
// producing kernel's epilogue writes them: LayerNorm, attention, this GEMM, the row gathers) - the same 4 bytes per element
// as fp32, but the consumer needs no conversion: every operand tile goes global -> LDS by LDS-DMA (global_load_lds_dwordx4),
// no VGPRs, no VALU.  That is what lets one workgroup per CU run a 256-row tile:
//
//   workgroup = 8 waves, tile BM x BN = 256 x 256 (waves 2 x 4) or 256 x 128 (waves 4 x 2), wave tile (TM x 32) x 64,
//   K step = 32 elements x 2 planes = 64 B rows (the byte geometry of a plain bf16 BK = 64 GEMM), two LDS stages:
//   stage = planes A_hi, A_lo [BM][32], W_hi, W_lo [BN][32]; 16-byte chunk q of row r sits at chunk q ^ ((r >> 2) & 3), so the
//   16 rows a ds_read_b128 lane group touches cover all 64 banks.  The LDS image is lane-linear per LDS-DMA instruction
//   (16 rows x 64 B = 1 KiB); the XOR is applied to the per-lane SOURCE address.
//   Loop: wait own DMA (vmcnt 0) -> barrier -> start DMA of step k+1 into the other stage -> 2 x {12 ds_read_b128,
//   3 x TM x 2 MFMA 32x32x16}.  Persistent: a workgroup walks tiles with stride gridDim.x and starts the first DMA of its next
//   tile before the epilogue of the current one, so the HBM latency of a tile's first step hides behind the stores.
//   Epilogue: accumulators -> wave-private LDS slice (32 x 64 fp32, rows of 256 B = one bank sweep) -> 16-byte rows:
//   + bias, activation, + residual, written as fp32 and/or as hi/lo planes (columns N .. round32(N) are zero filled: the next
//   layer's K padding).
#include <stdlib.h>
#include "scp_internal.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 sf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { ACT_NONE = 0, ACT_LEAKY = 1, ACT_GELU = 2, ACT_RELU = 3 };

template <int act>
__device__ __forceinline__ float apply_act_s(float y) {
    if (act == ACT_LEAKY) return y > 0.f ? y : 0.01f * y;
    if (act == ACT_GELU) return scp_gelu(y);     // exact-erf GELU to 4.4e-7 absolute: scp_internal.h (round 5; the degree-12 erf polynomial of rounds 1 - 4 is gone)
    if (act == ACT_RELU) return y > 0.f ? y : 0.f;
    return y;
}

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;

__device__ __forceinline__ void dma16(const void *g, char *l) {
    __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
}

struct GemmSplitArgs {
    const __bf16 *Ahi, *Alo; int64_t lda;        // activation planes [M][lda], lda % 8 == 0, columns K .. Kpad finite (zero)
    const __bf16 *Whi, *Wlo; int Kpad;           // weight planes [Npad][Kpad], zero padded, Npad % 256 == 0, Kpad % 32 == 0
    const float *bias;                           // [N] or null
    const float *res; int64_t ldr;               // fp32 residual [M][ldr] or null
    const int64_t *res_map;                      // optional: row m adds residual row res_map[m] (gathered residual)
    int res_first;                               // 1: activation AFTER the residual add, act(A.W^T + bias + residual)
    const int64_t *out_map;                      // optional: fp32 row m is written to C row out_map[m] (< 0: dropped)
    float *C; int64_t ldc;                       // fp32 output or null
    __bf16 *Ohi, *Olo; int64_t ldo;              // split output planes or null
    int M, N;
    int vec_ok;                                  // fp32 rows of C / residual are 16-byte aligned
    int ncols_out;                               // split output: columns written (N rounded up to 32, <= ldo): zero filled beyond N
    int wtiled;                                  // weight planes in the tiled layout (default) / row-major [Npad][Kpad] (SCP_WTILE=0)
    const float *a_isc, *w_isc;                  // F16 kernels: inverse power-of-two scale per activation row [M] / weight row [Npad]
    // F16 kernels, optional maxima of the OUTPUT taken in the epilogue (atomicMax on the bit patterns of |y|; the caller zeroes them): what the next
    // layer's power-of-two scales are made from, without a pass over the output (OctAttention: row scales of linear1's output for linear2, max |v|
    // of the key | value projection for the attention kernel's V planes)
    unsigned *row_max;                           // [M]: max |C[m][:]| or null
    unsigned *col_max;                           // one word: max |C[m][n]| over m < cm_rows, cm_lo <= n < cm_hi, or null
    int cm_lo, cm_hi, cm_rows;
};

// EXT: the epilogue extensions (gathered residual before the activation, scattered output rows) are compiled only into the
// variant that needs them - as run-time options they cost every dense layer ~12 % (measured)
// F16: the planes are IEEE half of power-of-two scaled rows (scp_split_rows_f16 / scp_split_weight_f16: 22 significant bits per operand
// instead of 16), the products run on v_mfma_f32_32x32x16_f16 and the accumulator is multiplied back by the two inverse scales in the
// epilogue - the arithmetic of gemm_bf16x3_kernel<.., true> (gemm.hip), same products in the same order: bit-identical results, without
// its fp32 -> plane conversion in every tile that reads a row.  OctAttention's dense layers (models/oct_attention.py).
template <int WM, int WN, int TM, int ACT, bool EXT, bool F16 = false>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_split_kernel(const GemmSplitArgs a) {
    constexpr int NW = WM * WN;                            // waves per workgroup: 8 (one workgroup per CU) or 4 (two per CU)
    constexpr int TN = 2;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int STAGE = (BM + BN) * 128;                 // bytes: 2 planes x 64 B rows
    constexpr int MAPOFF = STAGE + (STAGE > WM * WN * 8192 ? STAGE : WM * WN * 8192);   // EXT: row maps behind the stages / bounce slices
    constexpr int OFF_AL = BM * 64, OFF_BH = 2 * BM * 64, OFF_BL = 2 * BM * 64 + BN * 64;
    constexpr int NA = BM / (16 * NW), NB = BN / (16 * NW);            // LDS-DMA instructions per plane per wave
    static_assert(NW == 8 || NW == 4, "4 or 8 waves");
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction; tell the compiler (LDS-DMA bases live in M0)
    const int wm = w / WN, wn = w % WN;
    const int nk = a.Kpad >> 5;
    const int tn = (a.N + BN - 1) / BN, tm = (a.M + BM - 1) / BM;
    const int ntiles = tn * tm;

    // DMA lane constants: lane l of an instruction fills chunk l of its 1 KiB = row l >> 2, chunk position l & 3
    const int d_row = lane >> 2;
    const int d_q = (lane & 3) ^ ((lane >> 4) & 3);        // logical k-chunk stored at that position
    // fragment read lane constants
    const int f_pos0 = (h ^ ((lane >> 2) & 3)) << 4;       // byte offset of chunk (kc = 0, half h) in this lane's row; kc = 1: ^ 32

    auto tile_coords = [&](int t, int &m0, int &n0) {
        // XCD-aware order inside each round of gridDim.x tiles: workgroup b runs on XCD b & 7, so give every XCD a contiguous
        // run of tiles (the N tiles of one row stripe back to back): the stripe is fetched once per XCD L2
        // (full rounds only - the ragged last round keeps the identity, so the map stays a bijection)
        const int g = gridDim.x;
        const int round = t / g, b = t - round * g;
        int lin = t;
        if ((g & 7) == 0 && (round + 1) * g <= ntiles) lin = round * g + (b & 7) * (g >> 3) + (b >> 3);
        m0 = (lin / tn) * BM;
        n0 = (lin % tn) * BN;
    };

    auto stage_issue = [&](int st, int m0, int n0, int kt) {
        char *base = smem + st * STAGE;
        const int k0 = kt * 32 + 8 * d_q;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int cr = w + NW * j;                     // 16-row chunk of the plane
            int m = m0 + 16 * cr + d_row;
            m = m < a.M ? m : a.M - 1;
            const int64_t off = (int64_t)m * a.lda + k0;
            dma16(a.Ahi + off, base + cr * 1024);
            dma16(a.Alo + off, base + OFF_AL + cr * 1024);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int cr = w + NW * j;
            // weight planes are TILED (native._tile_planes): block (16-row group, 32-element k-slab) = the 1 KiB LDS image of one DMA
            // instruction, consecutive in memory, blocks ordered [row group][k-slab]
            const int64_t off = a.wtiled ? ((int64_t)((n0 >> 4) + cr) * (a.Kpad >> 5) + kt) * 512 + lane * 8 : (int64_t)(n0 + 16 * cr + d_row) * a.Kpad + k0;
            dma16(a.Whi + off, base + OFF_BH + cr * 1024);
            dma16(a.Wlo + off, base + OFF_BL + cr * 1024);
        }
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    // two waves share a SIMD (w and w + 4): the upper half always wins MFMA arbitration, so the pair falls out of phase - one
    // runs its MFMA batch while the other waits for its fragment reads - instead of both stalling on LDS at the same time

    int m0, n0;
    tile_coords(tile, m0, n0);
    stage_issue(0, m0, n0, 0);

    for (; tile < ntiles; tile += gridDim.x) {
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // ---- software-pipelined k loop -----------------------------------------------------------------------------------
        // A k-step (32 k x 2 planes) is four sub-phases of TM MFMA tiles x 3 products: (kc, p) = k-chunk of 16 x half of the
        // wave's row tiles.  Every sub-phase first issues the fragment reads of the NEXT one (second register set), then runs
        // its MFMAs, so LDS latency hides behind the matrix pipe.  One barrier per k-step, ahead of the last sub-phase: by then
        // every read of the current stage has been issued and waited for, the DMA of step kt + 1 (issued a k-step ago) has
        // landed, and the DMA of step kt + 2 may overwrite the current stage.
        constexpr int HM = TM / 2;
        bf16x8 Ah[2][HM], Al[2][HM], Bh[2][TN], Bl[2][TN];
        const int fa = (wm * (TM * 32) + col) * 64, fb = OFF_BH + (wn * 64 + col) * 64;
        auto load_b = [&](int buf, int sbo, int kc) {
            const int ob = sbo + fb + (f_pos0 ^ (kc * 32));
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                Bh[buf][j] = *(const bf16x8 *)(smem + ob + j * 2048);
                Bl[buf][j] = *(const bf16x8 *)(smem + ob + (OFF_BL - OFF_BH) + j * 2048);
            }
        };
        auto load_a = [&](int buf, int sbo, int kc, int p) {
            const int oa = sbo + fa + (f_pos0 ^ (kc * 32)) + p * (HM * 2048);
#pragma unroll
            for (int i = 0; i < HM; ++i) {
                Ah[buf][i] = *(const bf16x8 *)(smem + oa + i * 2048);
                Al[buf][i] = *(const bf16x8 *)(smem + oa + OFF_AL + i * 2048);
            }
        };
        auto mfma_sub = [&](int ab, int bb, int p) {
#pragma unroll
            for (int i = 0; i < HM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[p * HM + i][j] = F16 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sf16x8, Al[ab][i]), __builtin_bit_cast(sf16x8, Bh[bb][j]), acc[p * HM + i][j], 0, 0, 0)
                                             : __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al[ab][i], Bh[bb][j], acc[p * HM + i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < HM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[p * HM + i][j] = F16 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sf16x8, Ah[ab][i]), __builtin_bit_cast(sf16x8, Bl[bb][j]), acc[p * HM + i][j], 0, 0, 0)
                                             : __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[ab][i], Bl[bb][j], acc[p * HM + i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < HM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[p * HM + i][j] = F16 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sf16x8, Ah[ab][i]), __builtin_bit_cast(sf16x8, Bh[bb][j]), acc[p * HM + i][j], 0, 0, 0)
                                             : __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[ab][i], Bh[bb][j], acc[p * HM + i][j], 0, 0, 0);
        };

        SCP_WAIT_DMA(0);
        __syncthreads();   // DMA of step 0 landed; every wave is past the previous tile's epilogue
        if (nk > 1) stage_issue(1, m0, n0, 1);
        if (EXT) {   // the tile's residual / output row maps -> LDS now (latency hidden by the k loop): the epilogue would otherwise
                     // chain two dependent global loads (map entry, then the row it names) per row
            int64_t *mapbuf = (int64_t *)(smem + MAPOFF);
            if (tid < BM) {
                const int m = m0 + tid < a.M ? m0 + tid : a.M - 1;
                if (a.res_map) mapbuf[tid] = a.res_map[m];
                if (a.out_map) mapbuf[BM + tid] = a.out_map[m];
            }
        }
        load_b(0, 0, 0);
        load_a(0, 0, 0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            const int sbo = (kt & 1) * STAGE, sbn = STAGE - sbo;
            load_a(1, sbo, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_sub(0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_b(1, sbo, 1);
            load_a(0, sbo, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_sub(1, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            load_a(1, sbo, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_sub(0, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < nk) {
                SCP_WAIT_DMA(0);   // the DMA of step kt + 1 (issued a k-step ago) has landed
                __syncthreads();
                if (kt + 2 < nk) stage_issue(kt & 1, m0, n0, kt + 2);
                load_b(0, sbn, 0);
                load_a(0, sbn, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_sub(1, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        SCP_WAIT_DMA(0);
        __syncthreads();   // every wave is done with both stages
        const int cm0 = m0, cn0 = n0;
        if (tile + (int)gridDim.x < ntiles) {   // first step of the next tile: in flight during the epilogue (stage 0)
            tile_coords(tile + gridDim.x, m0, n0);
            stage_issue(0, m0, n0, 0);
        }

        // ---- epilogue through this wave's private 8 KiB slice of stage 1 ---------------------------------------------------
        float *stg = (float *)(smem + STAGE + w * 8192);
        float bv[TN], wsc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = cn0 + wn * 64 + j * 32 + col;
            bv[j] = (a.bias && n < a.N) ? a.bias[n] : 0.f;
            wsc[j] = F16 ? a.w_isc[n] : 1.f;                       // [Npad]: always in range
        }
        const int c4 = (lane & 15) * 4, rsub = lane >> 4;
        const int nb = cn0 + wn * 64 + c4;
        const bool full = (cn0 + wn * 64 + 64 <= a.N) && a.vec_ok;        // wave-uniform: the wave's 64 columns are all real
        const bool has_res = a.res != nullptr, has_c = a.C != nullptr, has_o = a.Ohi != nullptr;
        const bool maxes = F16 && (a.row_max || a.col_max);     // (bf16 instantiations: compiled out)
        float cmx = 0.f;
        auto take_max = [&](const f32x4 &y, int m, bool okrow) {   // all 64 lanes call this; lanes of rows beyond M contribute 0
            float mx = okrow ? fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3]))) : 0.f;
            if (a.col_max && okrow && m < a.cm_rows) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (nb + u >= a.cm_lo && nb + u < a.cm_hi) cmx = fmaxf(cmx, fabsf(y[u]));
            }
            if (a.row_max) {   // the 16 lanes of a row (same lane >> 4) hold its 64 columns of this wave
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
                if ((lane & 15) == 0 && okrow) atomicMax(a.row_max + m, __float_as_uint(fminf(mx, 3.0e38f)));
            }
        };
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = cm0 + wm * (TM * 32) + i * 32 + rsub;     // row of it = 0; it adds 4
            f32x4 rr[8];
            if (has_res && full) {   // residual rows first: eight independent 16-byte loads in flight (row clamped, no branch)
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int m = mb + 4 * it;
                    const int mc = m < a.M ? m : a.M - 1;
                    const int64_t rrow = (EXT && a.res_map) ? ((const int64_t *)(smem + MAPOFF))[mc - cm0] : (int64_t)mc;
                    rr[it] = *(const f32x4 *)(a.res + rrow * a.ldr + nb);
                }
            }
            float ia[16];
            if (F16) {   // inverse scale of the accumulator registers' rows (powers of two: the two multiplications are exact)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = cm0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    ia[r] = a.a_isc[m < a.M ? m : a.M - 1];
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float t0 = (F16 ? (acc[i][j][r] * ia[r]) * wsc[j] : acc[i][j][r]) + bv[j];
                    stg[ml * 64 + j * 32 + col] = (EXT && a.res_first) ? t0 : apply_act_s<ACT>(t0);
                }
            if (full) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int m = mb + 4 * it;
                    f32x4 y = *(const f32x4 *)(stg + (4 * it + rsub) * 64 + c4);
                    if (has_res) y += rr[it];
                    if (EXT && a.res_first) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) y[u] = apply_act_s<ACT>(y[u]);
                    }
                    if (maxes) take_max(y, m, m < a.M);
                    if (has_c && m < a.M) {
                        const int64_t orow = (EXT && a.out_map) ? ((const int64_t *)(smem + MAPOFF))[BM + m - cm0] : (int64_t)m;
                        if (orow >= 0) *(f32x4 *)(a.C + orow * a.ldc + nb) = y;
                    }
                    if (has_o) {
                        bf16x4 hi, lo;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const __bf16 hh = (__bf16)y[u];
                            hi[u] = hh;
                            lo[u] = (__bf16)(y[u] - (float)hh);
                        }
                        if (m < a.M) {
                            *(bf16x4 *)(a.Ohi + (int64_t)m * a.ldo + nb) = hi;
                            *(bf16x4 *)(a.Olo + (int64_t)m * a.ldo + nb) = lo;
                        }
                    }
                }
            } else {   // ragged N edge (N = 240 / 255 / 128 under a wider tile): element-wise, rare
                for (int it = 0; it < 8; ++it) {
                    const int m = mb + 4 * it;
                    f32x4 y = *(const f32x4 *)(stg + (4 * it + rsub) * 64 + c4);
                    if (m >= a.M) {
                        if (maxes) take_max(y, m, false);
                        continue;
                    }
                    for (int u = 0; u < 4; ++u) {
                        if (nb + u < a.N) {
                            if (has_res) y[u] += a.res[((EXT && a.res_map) ? a.res_map[m] : (int64_t)m) * a.ldr + nb + u];
                            if (EXT && a.res_first) y[u] = apply_act_s<ACT>(y[u]);
                            if (has_c) {
                                const int64_t orow = (EXT && a.out_map) ? a.out_map[m] : (int64_t)m;
                                if (orow >= 0) a.C[orow * a.ldc + nb + u] = y[u];
                            }
                        } else y[u] = 0.f;
                    }
                    if (maxes) take_max(y, m, true);
                    if (has_o && nb < a.ncols_out) {
                        bf16x4 hi, lo;
                        for (int u = 0; u < 4; ++u) {
                            const __bf16 hh = (__bf16)y[u];
                            hi[u] = hh;
                            lo[u] = (__bf16)(y[u] - (float)hh);
                        }
                        *(bf16x4 *)(a.Ohi + (int64_t)m * a.ldo + nb) = hi;
                        *(bf16x4 *)(a.Olo + (int64_t)m * a.ldo + nb) = lo;
                    }
                }
            }
        }
        if (maxes && a.col_max) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cmx = fmaxf(cmx, __shfl_xor(cmx, o));
            if (lane == 0 && cmx > 0.f) atomicMax(a.col_max, __float_as_uint(fminf(cmx, 3.0e38f)));
        }
        // the next tile's first barrier orders these LDS reads before the DMA that refills stage 1
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The two FINEST stages of a layer over concat_states (ehem.py:75-86,100-136) in ONE launch (round 6).
//
//     out[t] = act(A0[t] . W0^T + A1[t >> 1] . W1^T + bias + res[res_map[t]])
//
// Until round 5 the stage-1 product left as an fp32 [M / 2][N] partial sum that the stage-0 launch read back through the parent map:
// both launches were HBM-bound on that intermediate (4 KB written per stage-1 row, 2 KB read per stage-0 row, N = 1024).  Here a
// 256-token tile computes the products of its OWN 128 parents first and keeps them in the accumulators:
//   * the tile's tokens are dealt to the MFMA rows so that a parent and its two children share (lane, register): token tau of a
//     wave's 128 (tau = 4 rr + i) sits in row rr of row tile i = tau & 3; parent p = tau >> 1 = 2 rr + (i >> 1) sits in row rr of
//     parent tile i >> 1.  The permutation costs nothing: LDS-DMA takes a per-lane source address, the epilogue computes the row.
//   * coarse phase (K = 256 of stage 1): row tiles 0, 1 of the wave = its parent tiles, 24 MFMAs per k-step;
//   * acc[3] = acc[2] = parent tile 1, acc[1] = acc[0] = parent tile 0 (register copies), then the stage-0 product continues the
//     four chains.  Summation order of an output element: stage-1 k ascending, then stage-0 k ascending (+ bias, + residual): the
//     same for every row whatever the launch holds (batch invariance).
// 6 block-products per tile instead of 4 + 2 in two launches, no intermediate, one epilogue; K depth 512 - 768 instead of 256.
// Rows of window padding (parent map entry arbitrary) produce finite garbage nobody reads, as before.
// Measured (profiles/r6_gemm_split_shapes.txt): ancient_mlp 1 413 + 658 -> 1 816 us, prob_pred_mlp2 792 + 285 -> 941 us per L16-m frame.  Also measured
// and dropped in round 6: a start stagger of the persistent workgroups (a quarter of them delayed by 1, 2, 3 x 4 - 16 us once, so that the groups'
// epilogues fall into each other's k loops): no change on any shape (8.93 / 8.91 / 8.93 / 9.20 ms for 0 / 4 / 8 / 16 us units).
struct GemmHierArgs {
    const __bf16 *A0hi, *A0lo; int64_t lda0; int K0pad;     // stage-0 planes [M][lda0] (K0pad % 32 == 0)
    const __bf16 *A1hi, *A1lo; int64_t lda1; int64_t M1;    // stage-1 planes [M1][lda1], K = 256
    const __bf16 *W0hi, *W0lo, *W1hi, *W1lo;                // tiled weight planes [Npad][K0pad], [Npad][256]
    const int64_t *parent;                                  // [M]: stage-1 row of stage-0 row m
    const float *bias;                                      // [N] or null
    const float *res; int64_t ldr; const int64_t *res_map;  // optional fp32 residual rows res[res_map[m]] (the coarser stages' partial sum)
    __bf16 *Ohi, *Olo; int64_t ldo;                         // split output planes [M][ldo]
    int M, N;
};

template <int ACT>
__global__ __launch_bounds__(512, 2) void gemm_hier2_kernel(const GemmHierArgs a) {
    constexpr int WN = 4, TM = 4, TN = 2, NW = 8, HM = 2;
    constexpr int BM = 256, BN = 256;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int MAPOFF = 2 * STAGE;
    constexpr int OFF_AL = BM * 64, OFF_BH = 2 * BM * 64, OFF_BL = 2 * BM * 64 + BN * 64;
    constexpr int NK1 = 8;                                  // k-steps of the coarse phase (K = 256)
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w / WN, wn = w % WN;
    const int nk0 = a.K0pad >> 5, nkt = NK1 + nk0;
    const int tn = (a.N + BN - 1) / BN, tm = (a.M + BM - 1) / BM;
    const int ntiles = tn * tm;
    const int d_row = lane >> 2;
    const int d_q = (lane & 3) ^ ((lane >> 4) & 3);
    const int f_pos0 = (h ^ ((lane >> 2) & 3)) << 4;

    auto tile_coords = [&](int t, int &m0, int &n0) {       // the XCD-aware order of gemm_split_kernel
        const int g = gridDim.x;
        const int round = t / g, b = t - round * g;
        int lin = t;
        if ((g & 7) == 0 && (round + 1) * g <= ntiles) lin = round * g + (b & 7) * (g >> 3) + (b >> 3);
        m0 = (lin / tn) * BM;
        n0 = (lin % tn) * BN;
    };
    // LDS row R of the A tile (row tile i = (R >> 5) & 3 of wave-row group R >> 7, row rr = R & 31):
    //   fine phase: token m0 + 128 (R >> 7) + 4 rr + i;   coarse phase (i < 2): parent row pb + 64 (R >> 7) + 2 rr + i
    auto stage_issue = [&](int st, int m0, int n0, int64_t pb, int kt) {
        char *base = smem + st * STAGE;
        const bool coarse = kt < NK1;
        const int ks = coarse ? kt : kt - NK1;
        const int k0 = ks * 32 + 8 * d_q;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cr = w + NW * j;                      // 16-row chunk of the plane; its row tile: (cr >> 1) & 3 = (w >> 1) & 3
            const int R = 16 * cr + d_row, rr = R & 31, it = (R >> 5) & 3, g = R >> 7;
            if (coarse) {
                if (w < 4) {                                // wave-uniform: only the parent tiles are filled
                    int64_t r1 = pb + 64 * g + 2 * rr + it;
                    r1 = r1 < a.M1 ? r1 : a.M1 - 1;
                    const int64_t off = r1 * a.lda1 + k0;
                    dma16(a.A1hi + off, base + cr * 1024);
                    dma16(a.A1lo + off, base + OFF_AL + cr * 1024);
                }
            } else {
                int m = m0 + 128 * g + 4 * rr + it;
                m = m < a.M ? m : a.M - 1;
                const int64_t off = (int64_t)m * a.lda0 + k0;
                dma16(a.A0hi + off, base + cr * 1024);
                dma16(a.A0lo + off, base + OFF_AL + cr * 1024);
            }
        }
        const __bf16 *Wh = coarse ? a.W1hi : a.W0hi, *Wl = coarse ? a.W1lo : a.W0lo;
        const int nks = coarse ? NK1 : nk0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cr = w + NW * j;
            const int64_t off = ((int64_t)((n0 >> 4) + cr) * nks + ks) * 512 + lane * 8;
            dma16(Wh + off, base + OFF_BH + cr * 1024);
            dma16(Wl + off, base + OFF_BL + cr * 1024);
        }
    };
    auto parent_of = [&](int m0) { return a.parent[m0 < a.M ? m0 : a.M - 1]; };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    int m0, n0;
    tile_coords(tile, m0, n0);
    int64_t pb = parent_of(m0);
    stage_issue(0, m0, n0, pb, 0);

    for (; tile < ntiles; tile += gridDim.x) {
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        bf16x8 Ah[2][HM], Al[2][HM], Bh[2][TN], Bl[2][TN];
        const int fa = (wm * (TM * 32) + col) * 64, fb = OFF_BH + (wn * 64 + col) * 64;
        auto load_b = [&](int buf, int sbo, int kc) {
            const int ob = sbo + fb + (f_pos0 ^ (kc * 32));
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                Bh[buf][j] = *(const bf16x8 *)(smem + ob + j * 2048);
                Bl[buf][j] = *(const bf16x8 *)(smem + ob + (OFF_BL - OFF_BH) + j * 2048);
            }
        };
        auto load_a = [&](int buf, int sbo, int kc, int p) {
            const int oa = sbo + fa + (f_pos0 ^ (kc * 32)) + p * (HM * 2048);
#pragma unroll
            for (int i = 0; i < HM; ++i) {
                Ah[buf][i] = *(const bf16x8 *)(smem + oa + i * 2048);
                Al[buf][i] = *(const bf16x8 *)(smem + oa + OFF_AL + i * 2048);
            }
        };
        auto mfma_sub = [&](int ab, int bb, int p) {         // same order of the three partial products as gemm_split_kernel: lo.hi, hi.lo, hi.hi
#pragma unroll
            for (int i = 0; i < HM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[p * HM + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al[ab][i], Bh[bb][j], acc[p * HM + i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < HM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[p * HM + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[ab][i], Bl[bb][j], acc[p * HM + i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < HM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[p * HM + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[ab][i], Bh[bb][j], acc[p * HM + i][j], 0, 0, 0);
        };
        // step kt is in stage kt & 1; behind its barrier the DMA of step kt + 1 has landed, step kt + 2 may overwrite the current stage and the
        // first fragments of step kt + 1 are read ((kc 0, row tiles 0 - 1): the first sub-phase of either kind of step)
        auto turn = [&](int kt) {
            if (kt + 1 < nkt) {
                SCP_WAIT_DMA(0);
                __syncthreads();
                if (kt + 2 < nkt) stage_issue(kt & 1, m0, n0, pb, kt + 2);
                const int sbn = STAGE - (kt & 1) * STAGE;
                load_b(0, sbn, 0);
                load_a(0, sbn, 0, 0);
            }
        };

        SCP_WAIT_DMA(0);
        __syncthreads();   // DMA of step 0 landed; every wave is past the previous tile's epilogue
        stage_issue(1, m0, n0, pb, 1);
        {   // the tile's residual row map -> LDS (latency hidden by the k loop), indexed by token offset
            int64_t *mapbuf = (int64_t *)(smem + MAPOFF);
            if (tid < BM && a.res_map) mapbuf[tid] = a.res_map[m0 + tid < a.M ? m0 + tid : a.M - 1];
        }
        load_b(0, 0, 0);
        load_a(0, 0, 0, 0);
        // ---- coarse phase: the tile's 128 parents (two row tiles per wave), K = 256 of stage 1 ---------------------------
        for (int kt = 0; kt < NK1; ++kt) {
            const int sbo = (kt & 1) * STAGE;
            load_b(1, sbo, 1);
            load_a(1, sbo, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_sub(0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            turn(kt);
            __builtin_amdgcn_sched_barrier(0);
            mfma_sub(1, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // parent tile 1 -> row tiles 2, 3; parent tile 0 -> row tiles 0, 1
#pragma unroll
        for (int j = 0; j < TN; ++j) { acc[3][j] = acc[1][j]; acc[2][j] = acc[1][j]; acc[1][j] = acc[0][j]; }
        // ---- fine phase: the stage-0 product continues the four chains (the k loop of gemm_split_kernel) ---------------------
        for (int kt = NK1; kt < nkt; ++kt) {
            const int sbo = (kt & 1) * STAGE;
            load_a(1, sbo, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_sub(0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_b(1, sbo, 1);
            load_a(0, sbo, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_sub(1, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            load_a(1, sbo, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_sub(0, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            turn(kt);
            __builtin_amdgcn_sched_barrier(0);
            mfma_sub(1, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        SCP_WAIT_DMA(0);
        __syncthreads();   // every wave is done with both stages
        const int cm0 = m0, cn0 = n0;
        if (tile + (int)gridDim.x < ntiles) {   // first step of the next tile: in flight during the epilogue (stage 0)
            tile_coords(tile + gridDim.x, m0, n0);
            pb = parent_of(m0);
            stage_issue(0, m0, n0, pb, 0);
        }

        // ---- epilogue through this wave's private 8 KiB slice of stage 1: act(acc + bias + residual) -> hi / lo planes ---------
        float *stg = (float *)(smem + STAGE + w * 8192);
        float bv[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = cn0 + wn * 64 + j * 32 + col;
            bv[j] = (a.bias && n < a.N) ? a.bias[n] : 0.f;
        }
        const int c4 = (lane & 15) * 4, rsub = lane >> 4;
        const int nb = cn0 + wn * 64 + c4;
        const bool colsok = nb < a.N;                        // N % 4 == 0 (launcher): a lane's four columns are real or not together
        const bool has_res = a.res != nullptr;
        const int64_t *mapbuf = (const int64_t *)(smem + MAPOFF);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int tb = wm * 128 + i;                     // token offset of row rr of this row tile: tb + 4 rr
            f32x4 rr4[8];
            if (has_res && colsok) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int t = tb + 4 * (4 * it + rsub);
                    const int m = cm0 + t < a.M ? cm0 + t : a.M - 1;
                    const int64_t rrow = a.res_map ? mapbuf[m - cm0] : (int64_t)m;
                    rr4[it] = *(const f32x4 *)(a.res + rrow * a.ldr + nb);
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = (r & 3) + 8 * (r >> 2) + 4 * h;
                    stg[ml * 64 + j * 32 + col] = acc[i][j][r] + bv[j];
                }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int m = cm0 + tb + 4 * (4 * it + rsub);
                f32x4 y = *(const f32x4 *)(stg + (4 * it + rsub) * 64 + c4);
                if (has_res && colsok) y += rr4[it];
                bf16x4 hi, lo;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float v = colsok ? apply_act_s<ACT>(y[u]) : 0.f;
                    const __bf16 hh = (__bf16)v;
                    hi[u] = hh;
                    lo[u] = (__bf16)(v - (float)hh);
                }
                if (m < a.M && nb < a.ldo) {
                    *(bf16x4 *)(a.Ohi + (int64_t)m * a.ldo + nb) = hi;
                    *(bf16x4 *)(a.Olo + (int64_t)m * a.ldo + nb) = lo;
                }
            }
        }
        // the next tile's first barrier orders these LDS reads before the DMA that refills stage 1
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// fp32 rows -> hi/lo planes, optionally gathered: out[r][0:C] = split(src[idx ? idx[r] : r][0:C]); idx == n_src -> zero row
__global__ __launch_bounds__(256) void split_rows_kernel(const float *__restrict__ src, int64_t lds_, int64_t n_src, const int64_t *__restrict__ idx,
                                                        int C4, int Cp4, __bf16 *__restrict__ hi, __bf16 *__restrict__ lo, int64_t ldo, int64_t total) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const int64_t r = g / Cp4;
    const int c = (int)(g - r * Cp4);
    const int64_t s = idx ? idx[r] : r;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c < C4 && s < n_src) v = *(const f32x4 *)(src + s * lds_ + 4 * c);
    bf16x4 vh, vl;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const __bf16 hh = (__bf16)v[u];
        vh[u] = hh;
        vl[u] = (__bf16)(v[u] - (float)hh);
    }
    *(bf16x4 *)(hi + r * ldo + 4 * c) = vh;
    *(bf16x4 *)(lo + r * ldo + 4 * c) = vl;
}

// weight plane [Npad][Kpad] (row-major, bf16) -> the TILED layout the LDS-DMA loops stream: blocks of 1 KiB ordered [16-row group][32-element
// k-slab], each block the LDS image of one DMA instruction - row r of the group at bytes 64 r, its logical 16-byte chunk q at position
// q ^ ((r >> 2) & 3) (the read swizzle of the fragment loads).  One DMA instruction then reads eight whole cache lines instead of sixteen
// half lines: with every CU streaming, 48 KiB land in 1 190 instead of 2 020 cycles (tools/src/mb_lds_fill.cpp).
__global__ __launch_bounds__(256) void tile_weight_kernel(const uint4 *__restrict__ src, int Npad, int Kpad, uint4 *__restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;            // destination chunk (16 bytes = 8 elements)
    const int64_t total = (int64_t)Npad * Kpad / 8;
    if (i >= total) return;
    const int p = (int)(i & 3), r = (int)((i >> 2) & 15);
    const int64_t blk = i >> 6;
    const int nks = Kpad >> 5;
    const int ks = (int)(blk % nks);
    const int64_t rb = blk / nks;
    const int q = p ^ ((r >> 2) & 3);
    dst[i] = src[((rb * 16 + r) * (int64_t)Kpad + 32 * ks + 8 * q) >> 3];
}

extern "C" SCP_API int scp_tile_weight_bf16(const void *plane, int32_t Npad, int32_t Kpad, void *tiled, void *stream) {
    if (!plane || !tiled || Npad <= 0 || Kpad <= 0 || (Npad & 15) || (Kpad & 31) || (((uintptr_t)plane | (uintptr_t)tiled) & 15)) return SCP_EINVAL;
    const int64_t total = (int64_t)Npad * Kpad / 8;
    hipLaunchKernelGGL(tile_weight_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, (const uint4 *)plane, Npad, Kpad, (uint4 *)tiled);
    LAUNCH_CHECK();
    return SCP_OK;
}

extern "C" SCP_API int scp_split_rows(const float *src, int64_t ld_src, int64_t n_src, const int64_t *idx, int32_t C, void *hi, void *lo,
                                      int64_t ldo, int64_t rows, void *stream) {
    if (!src || !hi || !lo || rows < 0 || C <= 0 || (C & 3) || (ld_src & 3) || (ldo & 7) || ldo < C || (((uintptr_t)src) & 15) ||
        (((uintptr_t)hi | (uintptr_t)lo) & 7))
        return SCP_EINVAL;
    if (rows == 0) return SCP_OK;
    int Cp = (C + 31) & ~31;          // zero fill up to the next multiple of 32 (the consumer's K padding) when the row has room
    if (Cp > ldo) Cp = C;
    const int64_t total = rows * (Cp / 4);
    SCP_PROF(SCP_PROF_SPLIT_ROWS, stream, 8.0 * rows * C);
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, src, ld_src, n_src, idx, C / 4,
                       Cp / 4, (__bf16 *)hi, (__bf16 *)lo, ldo, total);
    LAUNCH_CHECK();
    return SCP_OK;
}

static int g_num_cu = 0;
template <int WM, int WN, int TM, bool EXT, bool F16 = false>
static int launch_cfg(const GemmSplitArgs &ga, int act, hipStream_t st, double work) {
    constexpr int BM = WM * TM * 32, BN = WN * 64;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int BOUNCE = WM * WN * 8192;
    constexpr int LDS = STAGE + (STAGE > BOUNCE ? STAGE : BOUNCE) + (EXT ? 2 * BM * 8 : 0);   // stage 0 + max(stage 1, the epilogue's 8 KiB bounce slice per wave) [+ row maps]
    static bool configured = false;
    if (!configured) {
        HIP_TRY(hipFuncSetAttribute((const void *)gemm_split_kernel<WM, WN, TM, ACT_NONE, EXT, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        HIP_TRY(hipFuncSetAttribute((const void *)gemm_split_kernel<WM, WN, TM, ACT_RELU, EXT, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        if (!F16) {
            HIP_TRY(hipFuncSetAttribute((const void *)gemm_split_kernel<WM, WN, TM, ACT_LEAKY, EXT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
            HIP_TRY(hipFuncSetAttribute((const void *)gemm_split_kernel<WM, WN, TM, ACT_GELU, EXT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        }
        configured = true;
    }
    const int64_t ntiles = cdiv64(ga.M, BM) * cdiv64(ga.N, BN);
    const int64_t slots = (int64_t)g_num_cu * (WM * WN == 4 ? 2 : 1);
    const unsigned grid = (unsigned)(ntiles < slots ? ntiles : slots);
    SCP_PROF(SCP_PROF_GEMM_SPLIT, st, work);
#define GOS(ACT, F) hipLaunchKernelGGL((gemm_split_kernel<WM, WN, TM, ACT, EXT, F>), dim3(grid), dim3(WM * WN * 64), LDS, st, ga)
    if (F16) { if (act == ACT_RELU) GOS(ACT_RELU, F16); else GOS(ACT_NONE, F16); }
    else switch (act) { case ACT_LEAKY: GOS(ACT_LEAKY, false); break; case ACT_GELU: GOS(ACT_GELU, false); break; case ACT_RELU: GOS(ACT_RELU, false); break; default: GOS(ACT_NONE, false); }
#undef GOS
    LAUNCH_CHECK();
    return SCP_OK;
}

static int linear_split_impl(const void *Ahi, const void *Alo, int64_t lda, const void *Whi, const void *Wlo, int32_t Npad, int32_t Kpad,
                             const float *bias, const float *residual, int64_t ldr, const int64_t *res_map, int32_t res_first, const int64_t *out_map, float *C, int64_t ldc,
                             void *Ohi, void *Olo, int64_t ldo, int32_t M, int32_t N, int32_t K, int32_t act, int32_t cfg, void *stream) {
    if (!Ahi || !Alo || !Whi || !Wlo || (!C && !Ohi) || ((Ohi == nullptr) != (Olo == nullptr)) || M <= 0 || N <= 0 || K <= 0 || (lda & 7) ||
        Kpad < K || (Kpad & 31) || lda < Kpad || (Npad & 255) || Npad < N || act < 0 || act > 3 || (C && ldc < N) ||
        (residual && ldr < N) || (Ohi && ((ldo & 7) || ldo < N)) ||
        (((uintptr_t)Ahi | (uintptr_t)Alo | (uintptr_t)Whi | (uintptr_t)Wlo) & 15) || (((uintptr_t)C | (uintptr_t)residual) & 3) ||
        (((uintptr_t)Ohi | (uintptr_t)Olo) & 7))
        return SCP_EINVAL;
    if (!g_num_cu) {
        int dev = 0;
        hipDeviceProp_t p;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipGetDeviceProperties(&p, dev));
        g_num_cu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    GemmSplitArgs ga;
    ga.a_isc = nullptr; ga.w_isc = nullptr; ga.row_max = nullptr; ga.col_max = nullptr; ga.cm_lo = ga.cm_hi = ga.cm_rows = 0;
    ga.Ahi = (const __bf16 *)Ahi; ga.Alo = (const __bf16 *)Alo; ga.lda = lda;
    ga.Whi = (const __bf16 *)Whi; ga.Wlo = (const __bf16 *)Wlo; ga.Kpad = Kpad;
    ga.bias = bias; ga.res = residual; ga.ldr = ldr; ga.C = C; ga.ldc = ldc;
    ga.res_map = residual ? res_map : nullptr; ga.res_first = (residual && res_first) ? 1 : 0;
    ga.out_map = C ? out_map : nullptr;
    ga.Ohi = (__bf16 *)Ohi; ga.Olo = (__bf16 *)Olo; ga.ldo = ldo; ga.M = M; ga.N = N;
    int nco = (N + 31) & ~31;
    if (nco > ldo) nco = (N + 3) & ~3;
    ga.ncols_out = nco;
    { static int wt = -1; if (wt < 0) { const char *e = getenv("SCP_WTILE"); wt = (e && e[0] == '0') ? 0 : 1; } ga.wtiled = wt; }
    if (cfg & ~0xff) {
        // timing probes of tools/mb_gemm_split.py (RESULTS ARE WRONG with them): 0x10000 = every activation row reads row 0 (operands
        // cache resident), 0x20000 = no output traffic.  Only honoured when SCP_GEMM_PROBE is set; an error otherwise.
        static int probe = -1;
        if (probe < 0) probe = getenv("SCP_GEMM_PROBE") ? 1 : 0;
        if (!probe || (cfg & ~0x300ff)) return SCP_EINVAL;
        if (cfg & 0x10000) ga.lda = 0;
        if (cfg & 0x20000) { ga.C = nullptr; ga.Ohi = ga.Olo = nullptr; ga.res = nullptr; }
    }
    cfg &= 255;
    ga.vec_ok = !((C && ((ldc & 3) || ((uintptr_t)C & 15))) || (residual && ((ldr & 3) || ((uintptr_t)residual & 15))));
    hipStream_t st = (hipStream_t)stream;
    // cfg 0 = automatic: the 256 x 128 tile where a 256-wide one would leave the chip's last round mostly empty or N <= 128
    const bool ext = ga.res_map || ga.res_first || ga.out_map;
    if (cfg == 0) {
        cfg = (N <= 128) ? 2 : 1;
        // short launches (the decoder's one-window forwards): when 256 x 256 tiles would leave most CUs idle, smaller tiles shorten the launch - it
        // lasts one tile's k loop + epilogue either way.  Every configuration accumulates an output element in the same k order: identical bits
        // (tests/test_gpu_model.py::test_linear_split_matches_fp32_activation_kernel, every cfg).  SCP_GEMM_SMALL=0: A/B bracket.
        static int small = -1;
        if (small < 0) { const char *e = getenv("SCP_GEMM_SMALL"); small = (e && e[0] == '0') ? 0 : 1; }
        if (small && cdiv64(M, 256) * cdiv64(N, 256) * 2 <= g_num_cu) cfg = ext ? 2 : 3;
    }
    const double work = 2.0 * M * (double)N * K;
    if (cfg == 2) return ext ? launch_cfg<4, 2, 2, true>(ga, act, st, work) : launch_cfg<4, 2, 2, false>(ga, act, st, work);
    if (cfg == 3) return ext ? SCP_EINVAL : launch_cfg<2, 2, 2, false>(ga, act, st, work);   // 128 x 128, 4 waves, two workgroups per CU
    return ext ? launch_cfg<2, 4, 4, true>(ga, act, st, work) : launch_cfg<2, 4, 4, false>(ga, act, st, work);
}

extern "C" SCP_API int scp_linear_split(const void *Ahi, const void *Alo, int64_t lda, const void *Whi, const void *Wlo, int32_t Npad, int32_t Kpad,
                                        const float *bias, const float *residual, int64_t ldr, float *C, int64_t ldc, void *Ohi, void *Olo,
                                        int64_t ldo, int32_t M, int32_t N, int32_t K, int32_t act, int32_t cfg, void *stream) {
    return linear_split_impl(Ahi, Alo, lda, Whi, Wlo, Npad, Kpad, bias, residual, ldr, nullptr, 0, nullptr, C, ldc, Ohi, Olo, ldo, M, N, K, act, cfg, stream);
}

// f16x3 form (OctAttention's dense layers, oct_attention.py:48-83 / attention_model.py:97-125): A planes + inverse row scales from
// scp_split_rows_f16, W planes (tiled) + inverse row scales from scp_split_weight_f16 with Npad % 256 == 0; act: 0 none, 3 ReLU.
// Bit-identical to scp_linear_f16x3_scaled on the fp32 rows the planes were made from.
static int linear_split_f16_impl(const void *Ahi, const void *Alo, int64_t lda, const float *a_inv_scale, const void *Whi, const void *Wlo,
                                 const float *w_inv_scale, int32_t Npad, int32_t Kpad, const float *bias, const float *residual, int64_t ldr,
                                 float *C, int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t act, int32_t cfg, uint32_t *row_max, uint32_t *col_max,
                                 int32_t col_lo, int32_t col_hi, int32_t col_rows, void *stream) {
    if (col_max && (col_lo < 0 || col_hi > N || col_lo >= col_hi || col_rows <= 0)) return SCP_EINVAL;
    if ((((uintptr_t)row_max | (uintptr_t)col_max) & 3)) return SCP_EINVAL;
    if (!Ahi || !Alo || !a_inv_scale || !Whi || !Wlo || !w_inv_scale || !C || M <= 0 || N <= 0 || K <= 0 || (lda & 7) || Kpad < K || (Kpad & 31) || cfg < 0 || cfg > 3 ||
        lda < Kpad || (Npad & 255) || Npad < N || (act != ACT_NONE && act != ACT_RELU) || ldc < N || (residual && ldr < N) ||
        (((uintptr_t)Ahi | (uintptr_t)Alo | (uintptr_t)Whi | (uintptr_t)Wlo) & 15) || (((uintptr_t)C | (uintptr_t)residual) & 3))
        return SCP_EINVAL;
    if (!g_num_cu) {
        int dev = 0;
        hipDeviceProp_t p;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipGetDeviceProperties(&p, dev));
        g_num_cu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    GemmSplitArgs ga = {};
    ga.Ahi = (const __bf16 *)Ahi; ga.Alo = (const __bf16 *)Alo; ga.lda = lda;
    ga.Whi = (const __bf16 *)Whi; ga.Wlo = (const __bf16 *)Wlo; ga.Kpad = Kpad;
    ga.bias = bias; ga.res = residual; ga.ldr = ldr; ga.C = C; ga.ldc = ldc; ga.M = M; ga.N = N;
    ga.a_isc = a_inv_scale; ga.w_isc = w_inv_scale;
    ga.row_max = row_max; ga.col_max = col_max; ga.cm_lo = col_lo; ga.cm_hi = col_hi; ga.cm_rows = col_rows;
    { static int wt = -1; if (wt < 0) { const char *e = getenv("SCP_WTILE"); wt = (e && e[0] == '0') ? 0 : 1; } ga.wtiled = wt; }
    ga.vec_ok = !((ldc & 3) || ((uintptr_t)C & 15) || (residual && ((ldr & 3) || ((uintptr_t)residual & 15))));
    // cfg 0 = automatic.  3: 128 x 128 tiles, 4 waves, two workgroups per CU: N = 600 / 300 waste 6 % / 22 % of a 128-wide column tiling
    // where 256-wide tiles waste 22 % / 41 % (tools/mb_oa_gemm.py: 660 against 711 us at M = 245 760, N = K = 600 with the bf16 planes);
    // 2: 256 rows x 128 columns, 8 waves (the same column waste, 1.33 x fewer LDS-DMA bytes per flop); 1: 256 x 256 (N a multiple of 256,
    // or wide concatenated projections such as key | value, N = 1200: 6.7 % waste, half the fill of the 128 x 128 tile).
    const double work = 2.0 * M * (double)N * K;
    hipStream_t st = (hipStream_t)stream;
    if (cfg == 0) cfg = (N > 768 || (N & 255) == 0) ? 1 : 3;
    if (cfg == 1) return launch_cfg<2, 4, 4, false, true>(ga, act, st, work);
    if (cfg == 2) return launch_cfg<4, 2, 2, false, true>(ga, act, st, work);
    return launch_cfg<2, 2, 2, false, true>(ga, act, st, work);
}

extern "C" SCP_API int scp_linear_split_f16(const void *Ahi, const void *Alo, int64_t lda, const float *a_inv_scale, const void *Whi, const void *Wlo,
                                            const float *w_inv_scale, int32_t Npad, int32_t Kpad, const float *bias, const float *residual, int64_t ldr,
                                            float *C, int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t act, int32_t cfg, void *stream) {
    return linear_split_f16_impl(Ahi, Alo, lda, a_inv_scale, Whi, Wlo, w_inv_scale, Npad, Kpad, bias, residual, ldr, C, ldc, M, N, K, act, cfg, nullptr, nullptr, 0, 0,
                                 0, stream);
}

// scp_linear_split_f16 that also takes maxima of its output in the epilogue (atomicMax on the bit patterns of |C|; the caller zeroes the words): row_max [M]
// = max |C[m][:]|; col_max (one word) = max |C[m][n]| over m < col_rows, col_lo <= n < col_hi.  The power-of-two scales of the NEXT f16x3 layer come from
// them (scp_row_scale_from_max; scp_octattn_attention_f16x3_vmax) without a pass over C.  Same C, bit for bit.
extern "C" SCP_API int scp_linear_split_f16_max(const void *Ahi, const void *Alo, int64_t lda, const float *a_inv_scale, const void *Whi, const void *Wlo,
                                                const float *w_inv_scale, int32_t Npad, int32_t Kpad, const float *bias, const float *residual, int64_t ldr,
                                                float *C, int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t act, int32_t cfg, uint32_t *row_max,
                                                uint32_t *col_max, int32_t col_lo, int32_t col_hi, int32_t col_rows, void *stream) {
    if (!row_max && !col_max) return SCP_EINVAL;
    return linear_split_f16_impl(Ahi, Alo, lda, a_inv_scale, Whi, Wlo, w_inv_scale, Npad, Kpad, bias, residual, ldr, C, ldc, M, N, K, act, cfg, row_max, col_max,
                                 col_lo, col_hi, col_rows, stream);
}

// the same with a GATHERED residual added BEFORE the activation: out[m] = act(A[m] . W^T + bias + residual[res_map[m]]).
// Lets a layer whose input is a concatenation of per-stage features sampled at token >> s (concat_states, ehem.py:75-86) run as one
// small product per stage at the stage's own resolution, each adding the coarser stages' partial sum through the parent-row map.
extern "C" SCP_API int scp_linear_split_gather(const void *Ahi, const void *Alo, int64_t lda, const void *Whi, const void *Wlo, int32_t Npad,
                                               int32_t Kpad, const float *bias, const float *residual, int64_t ldr, const int64_t *res_map, float *C,
                                               int64_t ldc, void *Ohi, void *Olo, int64_t ldo, int32_t M, int32_t N, int32_t K, int32_t act,
                                               int32_t cfg, void *stream) {
    return linear_split_impl(Ahi, Alo, lda, Whi, Wlo, Npad, Kpad, bias, residual, ldr, res_map, 1, nullptr, C, ldc, Ohi, Olo, ldo, M, N, K, act, cfg, stream);
}

// the same as scp_linear_split with SCATTERED fp32 output rows: row m is written to C row out_map[m] (negative: dropped).  The last
// layer of the probability heads writes its rows straight to their positions in the frame's coding-order table (encode.py:126-131).
extern "C" SCP_API int scp_linear_split_scatter(const void *Ahi, const void *Alo, int64_t lda, const void *Whi, const void *Wlo, int32_t Npad,
                                                int32_t Kpad, const float *bias, const int64_t *out_map, float *C, int64_t ldc, int32_t M, int32_t N,
                                                int32_t K, int32_t act, int32_t cfg, void *stream) {
    if (!out_map || !C) return SCP_EINVAL;
    return linear_split_impl(Ahi, Alo, lda, Whi, Wlo, Npad, Kpad, bias, nullptr, 0, nullptr, 0, out_map, C, ldc, nullptr, nullptr, 0, M, N, K, act, cfg,
                             stream);
}

// out = act(A0 . W0^T + A1[parent] . W1^T + bias + res[res_map]) as split planes: the two finest stages of a layer over concat_states in one launch
// (gemm_hier2_kernel).  A0: planes [M][lda0] with K0pad in {256, 512} columns, A1: planes [M1][lda1] with 256 columns, W0 / W1: tiled weight planes
// [Npad][K0pad] / [Npad][256] (scp_split_weight_bf16 + scp_tile_weight_bf16), parent: int64 [M], res: fp32 rows (optional, gathered through res_map when given),
// act: 0 none, 1 LeakyReLU(0.01).  M % 256 == 0 (rows of the packed layout come in 512s), N % 4 == 0, Npad % 256 == 0.
extern "C" SCP_API int scp_linear_split_hier2(const void *A0hi, const void *A0lo, int64_t lda0, int32_t K0pad, const void *A1hi, const void *A1lo, int64_t lda1,
                                              int64_t M1, const void *W0hi, const void *W0lo, const void *W1hi, const void *W1lo, int32_t Npad,
                                              const int64_t *parent, const float *bias, const float *res, int64_t ldr, const int64_t *res_map, void *Ohi,
                                              void *Olo, int64_t ldo, int32_t M, int32_t N, int32_t act, void *stream) {
    if (!A0hi || !A0lo || !A1hi || !A1lo || !W0hi || !W0lo || !W1hi || !W1lo || !parent || !Ohi || !Olo || M <= 0 || (M & 255) || M1 <= 0 || N <= 0 || (N & 3) ||
        (Npad & 255) || Npad < N || K0pad < 32 || (K0pad & 31) || lda0 < K0pad || (lda0 & 7) || lda1 < 256 || (lda1 & 7) || ldo < N || (ldo & 7) ||
        (act != ACT_NONE && act != ACT_LEAKY) || (res && (ldr < N || (ldr & 3) || ((uintptr_t)res & 15))) || (!res && res_map) ||
        (((uintptr_t)A0hi | (uintptr_t)A0lo | (uintptr_t)A1hi | (uintptr_t)A1lo | (uintptr_t)W0hi | (uintptr_t)W0lo | (uintptr_t)W1hi | (uintptr_t)W1lo) & 15) ||
        (((uintptr_t)Ohi | (uintptr_t)Olo) & 7))
        return SCP_EINVAL;
    if (!g_num_cu) {
        int dev = 0;
        hipDeviceProp_t p;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipGetDeviceProperties(&p, dev));
        g_num_cu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    constexpr int LDS = 2 * (256 + 256) * 128 + 256 * 8;
    static bool configured = false;
    if (!configured) {
        HIP_TRY(hipFuncSetAttribute((const void *)gemm_hier2_kernel<ACT_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        HIP_TRY(hipFuncSetAttribute((const void *)gemm_hier2_kernel<ACT_LEAKY>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        configured = true;
    }
    GemmHierArgs a;
    a.A0hi = (const __bf16 *)A0hi; a.A0lo = (const __bf16 *)A0lo; a.lda0 = lda0; a.K0pad = K0pad;
    a.A1hi = (const __bf16 *)A1hi; a.A1lo = (const __bf16 *)A1lo; a.lda1 = lda1; a.M1 = M1;
    a.W0hi = (const __bf16 *)W0hi; a.W0lo = (const __bf16 *)W0lo; a.W1hi = (const __bf16 *)W1hi; a.W1lo = (const __bf16 *)W1lo;
    a.parent = parent; a.bias = bias; a.res = res; a.ldr = ldr; a.res_map = res ? res_map : nullptr;
    a.Ohi = (__bf16 *)Ohi; a.Olo = (__bf16 *)Olo; a.ldo = ldo; a.M = M; a.N = N;
    const int64_t ntiles = cdiv64(M, 256) * cdiv64(N, 256);
    const unsigned grid = (unsigned)(ntiles < g_num_cu ? ntiles : g_num_cu);
    hipStream_t st = (hipStream_t)stream;
    // algorithmic flops: the stage-0 product at M rows + the stage-1 product at M / 2 rows
    SCP_PROF(SCP_PROF_GEMM_SPLIT, st, 2.0 * M * (double)N * (K0pad + 128.0));
    if (act == ACT_LEAKY) hipLaunchKernelGGL(gemm_hier2_kernel<ACT_LEAKY>, dim3(grid), dim3(512), LDS, st, a);
    else hipLaunchKernelGGL(gemm_hier2_kernel<ACT_NONE>, dim3(grid), dim3(512), LDS, st, a);
    LAUNCH_CHECK();
    return SCP_OK;
}
