// "Row-chain" kernels of the Swin blocks (gfx950 / CDNA4): a workgroup owns 128 token rows, each of its 4 waves 32 of them, and a
// wave keeps ITS rows in registers from the first load to the last store - as the B operand (the MFMA column = lane & 31 = the row)
// of every product.  Weights stream through LDS as the A operand, shared by the 4 waves:
//
//     C^T[out channel][row] += W[out channel][k] . X^T[k][row]          v_mfma_f32_32x32x16_bf16, bf16x3 (hi/lo split operands)
//
// The accumulator of such a product holds, per lane, 16 output channels OF THE LANE'S OWN ROW - which is exactly the shape of a B
// fragment of the next product (k = those channels).  So a chain LayerNorm -> dense -> GELU -> dense never leaves the register
// file: the k order of the consuming weight is permuted once, when the weight is tiled (rc_perm16 below), and LayerNorm's row
// statistics are one exchange between the two lanes that share a row (lane ^ 32).
//
//   scp_swin_ln_linear  : out = LayerNorm(x) . W^T + b      (layernorm_before + query|key|value, swin_transformer.py:443-501,654-660;
//                         layernorm(query) + query of the cross layers).  The LayerNorm affine is folded into the weight by the
//                         caller: W' = W diag(gamma), wbeta = W beta; rows the window pads AFTER LayerNorm (valid = 0,
//                         swin_transformer.py:638-641) come out as b alone.  Replaces layernorm_rows_kernel + gemm_split_kernel:
//                         the normalised rows never exist in HBM (no plane write, no plane read: 2 KB per row and LayerNorm).
//   scp_swin_post_attn  : x2 = x1 + fc2(GELU(fc1(LayerNorm(x1)))),  x1 = x + proj(o)   (attention.output.dense + residual,
//                         layernorm_after, intermediate.dense + GELU, output.dense + residual: swin_transformer.py:503-571,
//                         662-706) in ONE launch: reads the attention output planes and the residual stream, writes the residual
//                         stream - 3 KB per row instead of 8 (+ 3.6 of re-reads) for the three launches it replaces.
//
// Geometry: 256 threads, ONE wave per SIMD (the resident rows are 128 - 256 registers of a wave's 512).  LDS: a ring of four 32 KiB
// weight slots (one slot = 32 weight rows x 256 k, or 256 weight rows x 32 k: hi and lo planes as the 1 KiB LDS-DMA blocks of
// scp_tile_weight_bf16), filled by LDS-DMA one step (two slots, 96 MFMAs per wave) ahead behind a raw barrier; 16 KiB of
// wave-private bounce buffers for full-line stores; biases.
// Two accumulation chains always alternate in the matrix pipe (a lone dependent chain of v_mfma_f32_32x32x16 issues every 45 - 52
// cycles instead of 32, tools/src/mb_mfma_chain.cpp).
// Results are per row: independent of what else is in the launch and of the row's position in its tile (batch invariance, which
// the decoder relies on).
#include <stdlib.h>
#include "scp_internal.h"

typedef __bf16 rbf16x8 __attribute__((ext_vector_type(8)));
typedef float rf32x16 __attribute__((ext_vector_type(16)));
typedef float rf32x4 __attribute__((ext_vector_type(4)));
typedef int ri32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *rc_lds_ptr_t;
typedef const __attribute__((address_space(1))) void *rc_glb_ptr_t;

#define RC_ROWS 128                 // rows per workgroup tile
#define RC_SLOT 32768               // bytes of one weight slot (hi plane 16 KiB + lo plane 16 KiB)
#define RC_RING (4 * RC_SLOT)
#define RC_BOUNCE 4096              // per wave: 32 rows x 32 channels fp32
#define RC_OFF_BOUNCE RC_RING
#define RC_OFF_BIAS (RC_RING + 4 * RC_BOUNCE)
#define RC_LDS (RC_OFF_BIAS + 16384)

__device__ __forceinline__ void rc_dma16(const void *g, char *l) {
    __builtin_amdgcn_global_load_lds((rc_glb_ptr_t)g, (rc_lds_ptr_t)l, 16, 0, 0);
}

// hi/lo split of an fp32 value (the same arithmetic as every other producer of split operands: hi = bf16(x), lo = bf16(x - hi))
__device__ __forceinline__ void rc_split8(const float *f, rbf16x8 &hi, rbf16x8 &lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float v = f[i];
        asm volatile("" : "+v"(v));            // the rounded fp32 value (no FMA contraction into the subtraction below)
        const __bf16 hh = (__bf16)v;
        hi[i] = hh;
        lo[i] = (__bf16)(v - (float)hh);
    }
}

// A fragment of weight row (lane & 31) of a 32-row slot block, k-step s (16 k): the tiled image of scp_tile_weight_bf16
//   ROWCHUNK slot ([32 weight rows][256 k]):  plane = [2 row groups][8 k-slabs] x 1 KiB
//   KCHUNK   slot ([256 weight rows][32 k]):  plane = [16 row groups] x 1 KiB, m-block b = row groups 2b, 2b + 1
struct RcLane {
    int lane, col, h, w;
    int frag;          // byte offset of this lane's 16-byte chunk inside a 1 KiB block for k-chunk 0 (k-chunk 1: ^ 32)
    int rg;            // (col >> 4) : which 16-row group of a 32-row block
};

__device__ __forceinline__ RcLane rc_lane() {
    RcLane L;
    const int tid = threadIdx.x;
    L.lane = tid & 63; L.col = L.lane & 31; L.h = L.lane >> 5;
    L.w = __builtin_amdgcn_readfirstlane(tid >> 6);
    L.frag = (L.col & 15) * 64 + ((L.h ^ ((L.col >> 2) & 3)) << 4);
    L.rg = L.col >> 4;
    return L;
}

// one ROWCHUNK slot: weight rows [32 g, 32 g + 32) of a tiled plane pair with K = 256 (8 k-slabs): 16 consecutive KiB per plane
__device__ __forceinline__ void rc_issue_rowchunk(const RcLane &L, const __bf16 *Whi, const __bf16 *Wlo, int g, char *slot) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = L.w + 4 * j;                                   // 1 KiB block of the plane
        const int64_t off = ((int64_t)g * 16 + p) * 512 + L.lane * 8;
        rc_dma16(Whi + off, slot + p * 1024);
        rc_dma16(Wlo + off, slot + 16384 + p * 1024);
    }
}

// one KCHUNK slot: k-slab c of all 256 weight rows of a tiled plane pair with nks k-slabs per row group
__device__ __forceinline__ void rc_issue_kchunk(const RcLane &L, const __bf16 *Whi, const __bf16 *Wlo, int c, int nks, char *slot) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = L.w + 4 * j;                                   // row group
        const int64_t off = ((int64_t)p * nks + c) * 512 + L.lane * 8;
        rc_dma16(Whi + off, slot + p * 1024);
        rc_dma16(Wlo + off, slot + 16384 + p * 1024);
    }
}

__device__ __forceinline__ rbf16x8 rc_frag_rowchunk(const RcLane &L, const char *slot, int plane, int s) {
    return *(const rbf16x8 *)(slot + plane * 16384 + (L.rg * 8 + (s >> 1)) * 1024 + (L.frag ^ ((s & 1) << 5)));
}

__device__ __forceinline__ rbf16x8 rc_frag_kchunk(const RcLane &L, const char *slot, int plane, int b, int t) {
    return *(const rbf16x8 *)(slot + plane * 16384 + (2 * b + L.rg) * 1024 + (L.frag ^ (t << 5)));
}

#define RC_DS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")

// bf16x3 product step on two alternating chains (same order of the three partial products as gemm_split.hip: lo.hi, hi.lo, hi.hi)
__device__ __forceinline__ void rc_mfma3x2(rf32x16 &c0, rf32x16 &c1, const rbf16x8 &a0h, const rbf16x8 &a0l, const rbf16x8 &a1h,
                                           const rbf16x8 &a1l, const rbf16x8 &bh, const rbf16x8 &bl) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0l, bh, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l, bh, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, bl, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, bl, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h, bh, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, bh, c1, 0, 0, 0);
}

// Store one m-block (32 output channels x the wave's 32 rows, accumulator layout: lane = row, reg r = channel 8 (r >> 2) + 4 h +
// (r & 3)) as fp32 rows: through the wave's private 4 KiB bounce buffer (16-byte chunk c of row n at chunk c ^ (n & 7): conflict-free
// both ways), so that every global store instruction writes 8 rows x one whole 128-byte line.  `rs` addresses the tile's first
// row of the output (buffer resource: rows beyond M are dropped by the range check, no branch).
__device__ __forceinline__ void rc_store_block(const RcLane &L, char *bounce, const rf32x4 v[4], __amdgpu_buffer_rsrc_t rs, int64_t ldo_bytes,
                                               int row_in_tile0, int ch0) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
        *(rf32x4 *)(bounce + L.col * 128 + (((2 * q + L.h) ^ (L.col & 7)) << 4)) = v[q];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int rho = 8 * it + (L.lane >> 3), kap = L.lane & 7;
        const rf32x4 y = *(const rf32x4 *)(bounce + rho * 128 + ((kap ^ (rho & 7)) << 4));
        const int64_t off = (int64_t)(row_in_tile0 + rho) * ldo_bytes + (ch0 + 4 * kap) * 4;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ri32x4, y), rs, (int)off, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the bounce buffer is free again
}

// ---------------------------------------------------------------------------------------------------------------------------------
// LayerNorm statistics of the wave's rows: the lane holds 128 of its row's 256 channels, lane ^ 32 the other 128.  Two-pass
// (mean, then centred squares) in float32 like layernorm_rows_kernel; returns (mean, rstd).
__device__ __forceinline__ void rc_ln_stats(const float *v /*[128]*/, float eps, float &mean, float &rstd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 128; i += 4) s += (v[i] + v[i + 1]) + (v[i + 2] + v[i + 3]);
    s += __shfl_xor(s, 32);
    mean = s * (1.0f / 256.0f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 128; ++i) { const float d = v[i] - mean; q += d * d; }
    q += __shfl_xor(q, 32);
    rstd = rsqrtf(q * (1.0f / 256.0f) + eps);
}

struct RcLnLinArgs {
    const float *x; int64_t ldx;            // [M][ldx] fp32 rows, 256 channels
    const float *valid;                     // [M] multiplier applied AFTER LayerNorm (0 / 1) or null
    const __bf16 *Whi, *Wlo;                // tiled planes of W' = W diag(gamma), [Npad][256]
    const float *bias, *wbeta;              // [N]: b and W beta (either may be null)
    float *out; int64_t ldo;                // [M][ldo] fp32
    int M, N;                               // N % 128 == 0
    float eps;
    int probe;                              // timing probes (tools/mb_rowchain_probe.py, SCP_RC_PROBE; RESULTS ARE WRONG): 1 stores dropped, 2 no DMA, 8 no bounce / stores
    unsigned long long *dbg;                // diagnostic stamps (scp_rc_debug_buffer): per wave [barrier waits, steps, LayerNorm, drain, tiles]
};

#define RC_DS_WRITE(addr, val, off) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(val), "n"(off) : "memory")

// One STEP of the LN + linear kernel: 64 output channels (two 32-row weight slots of ring half `half`) x the wave's 32 rows, in 16
// slices of one k-step each = 6 MFMAs on two alternating chains.  A wave is alone on its SIMD and issues in order, so everything
// that is not an MFMA is dealt, one instruction at a time, into the GAPS between the MFMAs (tools/src/mb_vmem_issue.cpp, cycles per
// slice beside 192 of matrix pipe: four ds_read_b128 in front of the six MFMAs 230, one per gap 208; one LDS-DMA in front 262, in
// a gap 237, two in two gaps 243):
//   gaps 0 - 3 of slice s : the four fragment reads of k-step s + 1 (second register set; slice 15 reads k-step 0 of the NEXT step
//                           from the other ring half, which the barrier at the top of slice 15 has just published)
//   side work, also in gaps: the step after next's ... no: the NEXT step's 16 LDS-DMA pieces, 8 in slice w and 8 in slice w + 4 (the
//                           waves take turns: 64 B per cycle is all the CU's address unit takes); the PREVIOUS step's results (p0,
//                           p1) through the bounce buffer: block 0 written at slice 8, read back at 9, stored (4 x 8 rows x 128 B)
//                           at 10, block 1 at 11, 12, 13; with LOADX the next tile's rows are requested at 14 and 15.
// vmcnt counts loads, LDS-DMA and stores together IN ISSUE ORDER: the DMA pieces are the oldest of a step, so the barrier of slice 15
// waits for them and leaves the stores (and row loads) behind them in flight.
// All LDS traffic of the loop is inline asm with hand-counted waits: the compiler's wait insertion cannot see across asm, and what it
// inserts for its own LDS reads drains the read-ahead.
#define RC_SB __builtin_amdgcn_sched_barrier(0)
#ifndef RC_STORE_AUX
#define RC_STORE_AUX 0              // cache policy of the output stores (gfx950: 1 = sc0, 2 = nt, 16 = sc1)
#endif
template <bool LOADX, int PROBE>
__device__ __forceinline__ void rc_ll_step(const RcLane &L, const RcLnLinArgs &a, char *smem, int half, rf32x16 &c0, rf32x16 &c1,
                                           rf32x16 &p0, rf32x16 &p1, const rbf16x8 (&Xh)[16], const rbf16x8 (&Xl)[16], int g_next,
                                           __amdgpu_buffer_rsrc_t rs_prev, int voff, int ldo_bytes, int ch0_prev, const unsigned (&bw)[4],
                                           unsigned br, const float *xsrc, float (&v)[128], rbf16x8 (&A)[2][4], const float *tb_next, unsigned long long *st,
                                           __amdgpu_buffer_rsrc_t wr_hi, __amdgpu_buffer_rsrc_t wr_lo) {
    const unsigned ab = (unsigned)(uintptr_t)(rc_lds_ptr_t)(smem + half * (2 * RC_SLOT)) + L.rg * 8192;
    const unsigned ad0 = ab + L.frag, ad1 = ab + (L.frag ^ 32);
    const unsigned an0 = (unsigned)(uintptr_t)(rc_lds_ptr_t)(smem + (half ^ 1) * (2 * RC_SLOT)) + L.rg * 8192 + L.frag;   // k-step 0 of the next step
    char *nxt = smem + (half ^ 1) * (2 * RC_SLOT);
    rf32x4 rd[4];
    unsigned long long ts0 = 0;
    if (st) ts0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        if (st && (s == 1 || s == 8 || s == 15)) { const unsigned long long t = __builtin_amdgcn_s_memtime(); st[s == 1 ? 0 : s == 8 ? 1 : 2] += t - ts0; ts0 = t; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the fragments of this k-step (requested a slice ago)
        if (s == 15) {                                              // next step's slots landed; every wave is done reading this half's
            if (PROBE) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            else if (LOADX) asm volatile("s_waitcnt vmcnt(24)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
            if (st) { const unsigned long long t = __builtin_amdgcn_s_memtime(); st[3] += t - ts0; ts0 = t; }
        }
        RC_SB;
        const int nb = (s + 1) & 1, ns = (s + 1) & 15;
        const unsigned ra = (s == 15) ? an0 : ((ns & 1) ? ad1 : ad0);
#pragma unroll
        for (int gap = 0; gap < 6; ++gap) {
            // ---- the MFMA of this gap (product order lo.hi, hi.lo, hi.hi on both chains, as everywhere) -------------------------------
            if (gap == 0) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s & 1][1], Xh[s], c0, 0, 0, 0);
            if (gap == 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s & 1][3], Xh[s], c1, 0, 0, 0);
            if (gap == 2) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s & 1][0], Xl[s], c0, 0, 0, 0);
            if (gap == 3) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s & 1][2], Xl[s], c1, 0, 0, 0);
            if (gap == 4) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s & 1][0], Xh[s], c0, 0, 0, 0);
            if (gap == 5) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s & 1][2], Xh[s], c1, 0, 0, 0);
            RC_SB;
            // ---- what rides in the gap behind it ----------------------------------------------------------------------------------------
            if (gap < 4) {                                          // fragment gap of the next k-step: (slot gap >> 1, plane gap & 1)
                if (gap == 0) RC_DS_READ(A[nb][0], ra, (ns >> 1) * 1024);
                if (gap == 1) RC_DS_READ(A[nb][1], ra, 16384 + (ns >> 1) * 1024);
                if (gap == 2) RC_DS_READ(A[nb][2], ra, RC_SLOT + (ns >> 1) * 1024);
                if (gap == 3) RC_DS_READ(A[nb][3], ra, RC_SLOT + 16384 + (ns >> 1) * 1024);
            }
            if (s < 8 && (s & 3) == L.w && gap < 4 && !(PROBE & 2)) {   // 2 of the next step's LDS-DMA pieces: slot s >> 2, block w + 4 gap
                // buffer form: per-lane offset (16 lane) in one VGPR for all pieces, the piece's offset in an SGPR - the global form
                // spends two 64-bit VALU adds per piece on its address (32 of the ~48 cycles a piece then costs the wave)
                const int slot = s >> 2, pb = L.w + 4 * gap;
                const int soff = ((g_next + slot) * 16 + pb) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wr_hi, (rc_lds_ptr_t)(nxt + slot * RC_SLOT + pb * 1024), 16, L.lane * 16, soff, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wr_lo, (rc_lds_ptr_t)(nxt + slot * RC_SLOT + 16384 + pb * 1024), 16, L.lane * 16, soff, 0, 0);
            }
            if ((s == 8 || s == 11) && gap < 4 && !(PROBE & 8)) { // previous results -> bounce buffer, one 32-channel block at a time
                const rf32x16 &pp = (s == 8) ? p0 : p1;
                const rf32x4 t = {pp[4 * gap], pp[4 * gap + 1], pp[4 * gap + 2], pp[4 * gap + 3]};
                RC_DS_WRITE(bw[gap], t, 0);
            }
            if ((s == 9 || s == 12) && gap < 4 && !(PROBE & 8))   // ... the block's four row groups back (8 rows x 128 B each) ...
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rd[gap]) : "v"(br), "n"(gap * 1024) : "memory");
            if ((s == 10 || s == 13) && gap < 4 && !(PROBE & 8))  // ... and out
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ri32x4, rd[gap]), rs_prev, voff,
                                                       gap * 8 * ldo_bytes + (ch0_prev + 32 * (s == 13 ? 1 : 0)) * 4, RC_STORE_AUX);
            if (LOADX && s >= 14 && gap < 4 && !(PROBE & 4)) {                      // the next tile's rows into the registers LayerNorm has freed
#pragma unroll
                for (int t = 8 * (s - 14) + 2 * gap; t < 8 * (s - 14) + 2 * gap + 2; ++t) {
                    const rf32x4 q0 = *(const rf32x4 *)(xsrc + 16 * t), q1 = *(const rf32x4 *)(xsrc + 16 * t + 4);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { v[8 * t + u] = q0[u]; v[8 * t + 4 + u] = q1[u]; }
                }
            }
            if (s == 14 && gap >= 4) {                              // the previous results are out: those registers become the next
#pragma unroll                                                      // step's accumulators and start from its bias
                for (int q = 2 * (gap - 4); q < 2 * (gap - 4) + 2; ++q) {
                    const rf32x4 b0 = *(const rf32x4 *)(tb_next + 8 * q), b1 = *(const rf32x4 *)(tb_next + 32 + 8 * q);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { p0[4 * q + u] = b0[u]; p1[4 * q + u] = b1[u]; }
                }
            }
            RC_SB;
        }
    }
    if (st) { const unsigned long long t = __builtin_amdgcn_s_memtime(); st[4] += t - ts0; }
}

template <int PROBE>
__global__ __launch_bounds__(256, 1) void rc_ln_linear_kernel(const RcLnLinArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const RcLane L = rc_lane();
    const int ntiles = (a.M + RC_ROWS - 1) / RC_ROWS;
    const int nsteps = a.N >> 6;                                     // 64 output channels (two slots) per step; even
    float *sb = (float *)(smem + RC_OFF_BIAS), *sbw = sb + 1024;     // b and b + W beta
    for (int i = threadIdx.x; i < a.N; i += 256) {
        const float b = a.bias ? a.bias[i] : 0.f;
        sb[i] = b; sbw[i] = b + (a.wbeta ? a.wbeta[i] : 0.f);
    }
    __syncthreads();
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    char *bounce = smem + RC_OFF_BOUNCE + L.w * RC_BOUNCE;
    unsigned bw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bw[q] = (unsigned)(uintptr_t)(rc_lds_ptr_t)(bounce + L.col * 128 + (((2 * q + L.h) ^ (L.col & 7)) << 4));
    const unsigned br0 = (unsigned)(uintptr_t)(rc_lds_ptr_t)(bounce + (L.lane >> 3) * 128 + (((L.lane & 7) ^ ((L.lane >> 3) & 7)) << 4));
    const int ldo_bytes = (int)(a.ldo * 4);
    const int voff = (32 * L.w + (L.lane >> 3)) * ldo_bytes + (L.lane & 7) * 16;
    const __amdgpu_buffer_rsrc_t rs_none = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0, 0x00020000);   // every store out of range: dropped

    const int wbytes = ((a.N + 255) & ~255) * 512;                  // one weight plane: [Npad][256] bf16
    const __amdgpu_buffer_rsrc_t wr_hi = __builtin_amdgcn_make_buffer_rsrc((void *)a.Whi, 0, wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr_lo = __builtin_amdgcn_make_buffer_rsrc((void *)a.Wlo, 0, wbytes, 0x00020000);
    unsigned long long t_bar = 0, t_step = 0, t_ln = 0, t_drain = 0, t_prev = 0, n_tiles = 0;
    unsigned long long tsl[5] = {0, 0, 0, 0, 0};
    const bool DBG = a.dbg != nullptr;
    if (DBG) t_prev = __builtin_amdgcn_s_memtime();
    auto stamp = [&](unsigned long long &acc_t) { if (DBG) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc_t += t - t_prev; t_prev = t; } };

    // the first step's slots and the first tile's rows (k = 16 s + 8 h + i of row `col`: natural k order); every later tile's rows are
    // requested during the second-to-last step of the tile before it
    rc_issue_rowchunk(L, a.Whi, a.Wlo, 0, smem);
    rc_issue_rowchunk(L, a.Whi, a.Wlo, 1, smem + RC_SLOT);
    float v[128];
    auto xrow = [&](int t) {
        const int r = t * RC_ROWS + 32 * L.w + L.col;
        return a.x + (int64_t)(r < a.M ? r : a.M - 1) * a.ldx + 8 * L.h;
    };
    {
        const float *src = xrow(tile);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const rf32x4 q0 = *(const rf32x4 *)(src + 16 * s), q1 = *(const rf32x4 *)(src + 16 * s + 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) { v[8 * s + u] = q0[u]; v[8 * s + 4 + u] = q1[u]; }
        }
    }
    rbf16x8 A[2][4];                                                // weight fragments: the set in use and the one being read ahead
    {
        SCP_BARRIER_DMA(0);                                         // (also drains the row loads: once per launch)
        const unsigned ad0 = (unsigned)(uintptr_t)(rc_lds_ptr_t)(smem) + L.rg * 8192 + L.frag;
        RC_DS_READ(A[0][0], ad0, 0); RC_DS_READ(A[0][1], ad0, 16384); RC_DS_READ(A[0][2], ad0, RC_SLOT); RC_DS_READ(A[0][3], ad0, RC_SLOT + 16384);
    }
    int gstep = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * RC_ROWS;
        const int row = m0 + 32 * L.w + L.col;
        const int rowc = row < a.M ? row : a.M - 1;
        float keep = (row < a.M) ? 1.0f : 0.0f;
        if (a.valid) keep *= a.valid[rowc];
        float mean, rstd;
        rc_ln_stats(v, a.eps, mean, rstd);
        const float sc = rstd * keep;
        rbf16x8 Xh[16], Xl[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float f[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = (v[8 * s + i] - mean) * sc;
            rc_split8(f, Xh[s], Xl[s]);
        }
        stamp(t_ln);
        const float *tb = (keep != 0.f ? sbw : sb) + 4 * L.h;       // rows the window pads after LayerNorm get b alone
        // output rows of this tile through a buffer resource (range check instead of a branch on row < M)
        const int64_t rows_left = (int64_t)a.M - m0;
        const int64_t span = (rows_left < RC_ROWS ? rows_left : RC_ROWS) * a.ldo * 4;
        const __amdgpu_buffer_rsrc_t rs = (a.probe & 1) ? rs_none : __builtin_amdgcn_make_buffer_rsrc(a.out + (int64_t)m0 * a.ldo, 0, (int)span, 0x00020000);
        const bool more = tile + (int)gridDim.x < ntiles;
        const float *xnext = xrow(more ? tile + (int)gridDim.x : tile);

        rf32x16 acc[2][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {                               // the first step's accumulators start from its bias (the later ones' are
            const rf32x4 b0 = *(const rf32x4 *)(tb + 8 * q), b1 = *(const rf32x4 *)(tb + 32 + 8 * q);   // set inside the step before)
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc[0][0][4 * q + u] = b0[u]; acc[0][1][4 * q + u] = b1[u]; }
        }
        for (int j = 0; j < nsteps; j += 2) {
            // step j -> acc[0], stores acc[1] (step j - 1); step j + 1 -> acc[1], stores acc[0]
            if (j + 2 == nsteps)
                rc_ll_step<true, PROBE>(L, a, smem, gstep & 1, acc[0][0], acc[0][1], acc[1][0], acc[1][1], Xh, Xl, 2 * (j + 1), j ? rs : rs_none, voff,
                                 ldo_bytes, 64 * (j - 1), bw, br0, xnext, v, A, tb + 64 * (j + 1), DBG ? tsl : nullptr, wr_hi, wr_lo);
            else
                rc_ll_step<false, PROBE>(L, a, smem, gstep & 1, acc[0][0], acc[0][1], acc[1][0], acc[1][1], Xh, Xl, 2 * (j + 1), j ? rs : rs_none, voff,
                                  ldo_bytes, 64 * (j - 1), bw, br0, xnext, v, A, tb + 64 * (j + 1), DBG ? tsl : nullptr, wr_hi, wr_lo);
            ++gstep;
            rc_ll_step<false, PROBE>(L, a, smem, gstep & 1, acc[1][0], acc[1][1], acc[0][0], acc[0][1], Xh, Xl, (j + 2 < nsteps) ? 2 * (j + 2) : 0, rs, voff,
                              ldo_bytes, 64 * j, bw, br0, xnext, v, A, tb + 64 * ((j + 2 < nsteps) ? j + 2 : 0), DBG ? tsl : nullptr, wr_hi, wr_lo);
            ++gstep;
        }
        stamp(t_step);
        // ---- drain: the last step's results (nothing left to overlap with in this tile) ----------------------------------------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            rf32x4 o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int u = 0; u < 4; ++u) o[q][u] = acc[1][blk][4 * q + u];
            rc_store_block(L, bounce, o, rs, a.ldo * 4, 32 * L.w, 64 * (nsteps - 1) + 32 * blk);
        }
        stamp(t_drain);
        ++n_tiles;
    }
    if (DBG && L.lane == 0) {
        unsigned long long *o = a.dbg + ((size_t)blockIdx.x * 4 + L.w) * 8;
        o[0] = tsl[3]; o[1] = t_step; o[2] = t_ln; o[3] = t_drain; o[4] = n_tiles; o[5] = tsl[0]; o[6] = tsl[1]; o[7] = tsl[2] + tsl[4];
    }
    SCP_WAIT_DMA(0);                                                // the slots requested for a tile that does not exist
}

static unsigned long long *g_rc_dbg = nullptr;   // diagnostic only (tools/mb_rowchain_probe.py): [workgroup][wave][8] cycle sums
extern "C" SCP_API int scp_rc_debug_buffer(unsigned long long *dev_buf) { g_rc_dbg = dev_buf; return SCP_OK; }

static int g_rc_num_cu = 0;
static int rc_num_cu() {
    if (!g_rc_num_cu) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
        g_rc_num_cu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    return g_rc_num_cu;
}

// out[m] = valid[m] * LayerNorm_noaffine(x[m]) . W'^T + bias + valid[m] * wbeta     (x: fp32 [M][ldx], 256 channels)
// W' planes: scp_split_weight_bf16 + scp_tile_weight_bf16 of W diag(gamma) ([Npad][256]).  N % 128 == 0, N <= 1024.
extern "C" SCP_API int scp_swin_ln_linear(const float *x, int64_t ldx, const float *valid, const void *Whi, const void *Wlo, const float *bias,
                                          const float *wbeta, float eps, float *out, int64_t ldo, int32_t M, int32_t N, void *stream) {
    if (!x || !Whi || !Wlo || !out || M <= 0 || N <= 0 || (N & 127) || N > 1024 || ldx < 256 || (ldx & 3) || ldo < N || (ldo & 3) ||
        (((uintptr_t)x | (uintptr_t)out | (uintptr_t)Whi | (uintptr_t)Wlo) & 15) || (int64_t)RC_ROWS * ldo * 4 > 0x7fffffffLL)
        return SCP_EINVAL;
    static bool configured = false;
    if (!configured) {
        HIP_TRY(hipFuncSetAttribute((const void *)rc_ln_linear_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS));
        HIP_TRY(hipFuncSetAttribute((const void *)rc_ln_linear_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS));
        HIP_TRY(hipFuncSetAttribute((const void *)rc_ln_linear_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS));
        HIP_TRY(hipFuncSetAttribute((const void *)rc_ln_linear_kernel<14>, hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS));
        configured = true;
    }
    RcLnLinArgs a;
    a.x = x; a.ldx = ldx; a.valid = valid; a.Whi = (const __bf16 *)Whi; a.Wlo = (const __bf16 *)Wlo; a.bias = bias; a.wbeta = wbeta;
    a.out = out; a.ldo = ldo; a.M = M; a.N = N; a.eps = eps;
    { static int pr = -1; if (pr < 0) { const char *e = getenv("SCP_RC_PROBE"); pr = e ? atoi(e) : 0; } a.probe = pr; }
    if (a.probe) { const char *e = getenv("SCP_RC_PROBE"); a.probe = e ? atoi(e) : 0; }
    a.dbg = g_rc_dbg;
    const int ntiles = (M + RC_ROWS - 1) / RC_ROWS;
    const int ncu = rc_num_cu();
    const dim3 grid((unsigned)(ntiles < ncu ? ntiles : ncu));
    // SCP_RC_PROBE (tools/mb_rowchain_probe.py; RESULTS ARE WRONG): builds without the LDS-DMA (2), the bounce + stores (8), or both and the row prefetch (14)
    if ((a.probe & 14) == 14) hipLaunchKernelGGL(rc_ln_linear_kernel<14>, grid, dim3(256), RC_LDS, (hipStream_t)stream, a);
    else if (a.probe & 8) hipLaunchKernelGGL(rc_ln_linear_kernel<8>, grid, dim3(256), RC_LDS, (hipStream_t)stream, a);
    else if (a.probe & 2) hipLaunchKernelGGL(rc_ln_linear_kernel<2>, grid, dim3(256), RC_LDS, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(rc_ln_linear_kernel<0>, grid, dim3(256), RC_LDS, (hipStream_t)stream, a);
    LAUNCH_CHECK();
    return SCP_OK;
}
