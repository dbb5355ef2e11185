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

// hi/lo split of fp32 values (the arithmetic of every producer of split operands: hi = bf16(x), lo = bf16(x - hi)), two at a time:
// v_cvt_pk_bf16_f32 packs the pair, so a fragment is assembled from four 32-bit words - element-wise conversion made the compiler
// hold every 16-bit half in a register of its own until a v_perm_b32 packed it (hundreds of spills in the LayerNorm section).
typedef float rf32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 rbf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned ru32x4 __attribute__((ext_vector_type(4)));
typedef unsigned ru32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned rc_pack2(float a, float b) {
    const rf32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, rbf16x2));
}
__device__ __forceinline__ void rc_split2(float a, float b, unsigned &hi, unsigned &lo) {
    asm volatile("" : "+v"(a), "+v"(b));        // the rounded fp32 values (no FMA contraction into the subtractions below)
    hi = rc_pack2(a, b);
    lo = rc_pack2(a - __builtin_bit_cast(float, hi << 16), b - __builtin_bit_cast(float, hi & 0xffff0000u));
}
__device__ __forceinline__ void rc_split8(const float *f, rbf16x8 &hi, rbf16x8 &lo) {
    ru32x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) { unsigned a, b; rc_split2(f[2 * i], f[2 * i + 1], a, b); h[i] = a; l[i] = b; }
    hi = __builtin_bit_cast(rbf16x8, h);
    lo = __builtin_bit_cast(rbf16x8, l);
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

// The first fragments of the NEXT step are read behind the last MFMA of a step, all four and their wait in ONE asm statement: the
// compiler treats an asm output as valid the moment the statement ends, and across a step boundary it does move these registers (a
// v_accvgpr_write right behind the read that fills them copied stale bytes: errors of one lo plane, 1e-4, that came and went with
// unrelated edits).  Inside a step - one basic block - nothing may touch a fragment between its read and the counted wait:
// tests/test_rowchain_isa.py scans the generated code for exactly that.
#define RC_DS_READ4_WAIT(d0, a0, o0, d1, a1, o1, d2, a2, o2, d3, a3, o3)                                                              \
    asm volatile("ds_read_b128 %0, %4 offset:%8\n\tds_read_b128 %1, %5 offset:%9\n\tds_read_b128 %2, %6 offset:%10\n\t"                   \
                 "ds_read_b128 %3, %7 offset:%11\n\ts_waitcnt lgkmcnt(0)"                                                              \
                 : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(o0), "n"(o1), "n"(o2), "n"(o3) : "memory")
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
__device__ __forceinline__ void rc_store_block(const RcLane &L, char *bounce, const rf32x4 v[4], __amdgpu_buffer_rsrc_t rs, int ldo_bytes, int voff,
                                               int ch0) {
    // voff = (32 w + (lane >> 3)) * ldo_bytes + (lane & 7) * 16: the one per-lane offset; row group and channel block go into the scalar
    // offset of the store (32 per-lane offsets, one per store of a tile, were hoisted, spilled, and each store then waited for its
    // reload - and with it, vmcnt being in order, for every store before it: 20 k cycles per tile)
#pragma unroll
    for (int q = 0; q < 4; ++q)
        *(rf32x4 *)(bounce + L.col * 128 + (((2 * q + L.h) ^ (L.col & 7)) << 4)) = v[q];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    rf32x4 y[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int rho = 8 * it + (L.lane >> 3), kap = L.lane & 7;
        y[it] = *(const rf32x4 *)(bounce + rho * 128 + ((kap ^ (rho & 7)) << 4));
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ri32x4, y[it]), rs, voff, it * 8 * ldo_bytes + ch0 * 4, 0);
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
    // rc_ln_linear_kernel<., KV = true> (scp_swin_ln_qkv): the first nq steps (64 channels each) are the query, written to `out` as above;
    // the next four are the key heads, the last four the value heads, written as bf16 hi / lo planes in the layout of the plane-fed
    // attention (csrc/attn.hip: swin_attn_planes_kernel): planes = [4][Tp][256] bf16 = K hi, K lo, V^T hi, V^T lo; plane_bytes = Tp * 512
    __bf16 *planes; int64_t plane_bytes; int nq;
    // Round 5, short launches (the decoder's one-window forwards: 4 - 64 tiles on 256 CUs, every launch as long as ONE tile's serial chain of
    // steps): `ngroups` workgroups share a tile, each runs LayerNorm on the tile's rows and then its own run of nsteps / ngroups steps (an even
    // number, never straddling query / key / value).  Every output channel is still one accumulation chain of one wave: identical bits.
    int ngroups;
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
// Behind a step's barrier the four waves of a workgroup run the same instruction stream in lockstep, one per SIMD - and reach every one of the
// next step's 16 LDS-DMA pieces (1 KiB each: 16 cycles of the CU's one 64 B / cycle address path) in the same cycle: a piece then holds its wave's
// issue until the three in front of it are through.  RC_SKEW > 0 delays wave w by w x RC_SKEW rounds of ~16 cycles right behind the barrier
// (a scalar loop inside the barrier's own asm statement: the step stays one basic block for the compiler), so the waves' pieces arrive in turn.
#ifndef RC_SKEW
#define RC_SKEW 0
#endif
#if RC_SKEW > 0
#define RC_STEP_BARRIER(WAITS, w)                                                                                                          \
    do {                                                                                                                                   \
        int _rc_n;                                                                                                                         \
        asm volatile("s_waitcnt " WAITS "\n\ts_barrier\n\ts_mov_b32 %0, %1\n\ts_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lrcskew_done%=\n"          \
                     ".Lrcskew_loop%=:\n\ts_nop 7\n\ts_sub_u32 %0, %0, 1\n\ts_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 .Lrcskew_loop%=\n"              \
                     ".Lrcskew_done%=:"                                                                                                   \
                     : "=&s"(_rc_n) : "s"((w) * RC_SKEW) : "memory", "scc");                                                                \
    } while (0)
#else
#ifdef RC_PROBE_NOBAR
#define RC_STEP_BARRIER(WAITS, w) asm volatile("s_waitcnt " WAITS ::: "memory")
#else
#define RC_STEP_BARRIER(WAITS, w) asm volatile("s_waitcnt " WAITS "\n\ts_barrier" ::: "memory")
#endif
#endif
// Timing probes of the post-attention kernel's steps (WRONG results; tools/r5_step_probe.sh): RC_PROBE_NODMA no LDS-DMA pieces inside the steps,
// RC_PROBE_NOFRAG no fragment reads (and no waits for them) inside the steps, RC_PROBE_NOBAR no workgroup barrier per step
#ifdef RC_PROBE_NODMA
#define RC_PROBE_DMA_ON 0
#else
#define RC_PROBE_DMA_ON 1
#endif
#ifdef RC_PROBE_NOFRAG
#define RC_PROBE_FRAG_ON 0
#else
#define RC_PROBE_FRAG_ON 1
#endif
// RC_DMA_SPREAD: the 16 pieces of a step over slices 0 - 11 (two per slice in slices 0 - 3, one per slice afterwards; the barrier of slice 15
// must find them landed) instead of two per slice over the first eight
#ifdef RC_DMA_SPREAD
#define RC_DMA_P(s, gap) ((s) < 4 ? 2 * (s) + ((gap) == 3) : (s) + 4)
#define RC_DMA_HERE(s, gap) (((s) < 4 && ((gap) == 1 || (gap) == 3)) || ((s) >= 4 && (s) < 12 && (gap) == 1))
#define RC_DMA_SLOT(s, gap) (RC_DMA_P(s, gap) >> 3)
#define RC_DMA_IDX(s, gap) ((RC_DMA_P(s, gap) >> 1) & 3)
#define RC_DMA_PLANE(s, gap) (RC_DMA_P(s, gap) & 1)
#else
#define RC_DMA_HERE(s, gap) ((s) < 8 && ((gap) == 1 || (gap) == 3))
#define RC_DMA_SLOT(s, gap) ((s) >> 2)
#define RC_DMA_IDX(s, gap) ((s) & 3)
#define RC_DMA_PLANE(s, gap) ((gap) >> 1)
#endif
#ifdef RC_WAIT0
#define RC_FRAG_WAIT() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define RC_FRAG_WAIT() asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory")
#endif
#ifndef RC_STORE_AUX
#define RC_STORE_AUX 0              // cache policy of the output stores (gfx950: 1 = sc0, 2 = nt, 16 = sc1)
#endif
// SWAP (the value steps of rc_ln_linear_kernel<., KV>): the operands change places - A = the wave's rows, B = the weight fragment (the
// two fragment layouts are the same registers) - so the accumulator holds C[row][channel]: lane = CHANNEL, register r = row 8 (r >> 2) +
// 4 h + (r & 3), which is the order V^T tiles are stored in; same products, same k order, same values.  NSWAP: the NEXT step is such a
// step, its accumulators start from b[channel] (rows the window pads: b alone, others b + W beta; maskh = the rows' valid bits >> 4 h).
template <bool LOADX, int PROBE, bool SWAP = false, bool NSWAP = false>
__device__ __forceinline__ void rc_ll_step(const RcLane &L, const RcLnLinArgs &a, char *smem, int half, rf32x16 &c0, rf32x16 &c1,
                                           rf32x16 &p0, rf32x16 &p1, const rbf16x8 (&Xh)[16], const rbf16x8 (&Xl)[16], int g_next,
                                           __amdgpu_buffer_rsrc_t rs_prev, int voff, int ldo_bytes, int ch0_prev, const unsigned (&bw)[4],
                                           unsigned br, const float *xsrc, float (&v)[128], rbf16x8 (&A)[2][4], const float *tb_next, unsigned long long *st,
                                           __amdgpu_buffer_rsrc_t wr_hi, __amdgpu_buffer_rsrc_t wr_lo, const float *sbn = nullptr, unsigned maskh = 0) {
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
        if (s == 15) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                              // next step's slots landed; every wave is done reading this half's
            if (PROBE) RC_STEP_BARRIER("vmcnt(0)", L.w);
            else if (LOADX) RC_STEP_BARRIER("vmcnt(24)", L.w);
            else RC_STEP_BARRIER("vmcnt(8)", L.w);
            if (st) { const unsigned long long t = __builtin_amdgcn_s_memtime(); st[3] += t - ts0; ts0 = t; }
        }
        RC_SB;
        const int nb = (s + 1) & 1, ns = (s + 1) & 15;
        const unsigned ra = (ns & 1) ? ad1 : ad0;
#pragma unroll
        for (int gap = 0; gap < 6; ++gap) {
            // ---- the MFMA of this gap (product order lo.hi, hi.lo, hi.hi on both chains, as everywhere).  The four fragments of this
            // k-step were requested one per gap of the slice before, in the order their first MFMA needs them: before each of the first
            // four MFMAs the oldest outstanding read is the one it needs - a whole slice old - and the three younger ones stay in flight
            if (gap < 4) { RC_FRAG_WAIT(); RC_SB; }
#define RC_LL_MFMA(c, wf, xf) c = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, wf, c, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, c, 0, 0, 0)
            if (gap == 0) RC_LL_MFMA(c0, A[s & 1][1], Xh[s]);
            if (gap == 1) RC_LL_MFMA(c1, A[s & 1][3], Xh[s]);
            if (gap == 2) RC_LL_MFMA(c0, A[s & 1][0], Xl[s]);
            if (gap == 3) RC_LL_MFMA(c1, A[s & 1][2], Xl[s]);
            if (gap == 4) RC_LL_MFMA(c0, A[s & 1][0], Xh[s]);
            if (gap == 5) RC_LL_MFMA(c1, A[s & 1][2], Xh[s]);
#undef RC_LL_MFMA
            RC_SB;
            // ---- what rides in the gap behind it ----------------------------------------------------------------------------------------
            if (gap < 4 && s < 15) {                                // a fragment of the next k-step, in the order of use: lo 0, lo 1, hi 0, hi 1
                if (gap == 0) RC_DS_READ(A[nb][1], ra, 16384 + (ns >> 1) * 1024);
                if (gap == 1) RC_DS_READ(A[nb][3], ra, RC_SLOT + 16384 + (ns >> 1) * 1024);
                if (gap == 2) RC_DS_READ(A[nb][0], ra, (ns >> 1) * 1024);
                if (gap == 3) RC_DS_READ(A[nb][2], ra, RC_SLOT + (ns >> 1) * 1024);
            }
            if (s < 8 && (gap == 1 || gap == 3) && !(PROBE & 2)) {   // one of the next step's 16 LDS-DMA pieces: slot s >> 2, block w + 4 (s & 3)
                // (buffer form: the per-lane offset 16 lane in one VGPR for all pieces, the piece's offset in an SGPR; no wave-dependent
                // branch, so the step is one basic block - see rc_dma_piece)
                const int slot = s >> 2, pb = L.w + 4 * (s & 3), plane = gap >> 1;
                const int soff = ((g_next + slot) * 16 + pb) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(plane ? wr_lo : wr_hi, (rc_lds_ptr_t)(nxt + slot * RC_SLOT + plane * 16384 + pb * 1024), 16,
                                                         L.lane * 16, soff, 0, 0);
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
            // (Measured and dropped, round 4: the same 32 loads one per gap over slices 8 - 15 instead of four per gap in slices 14 and 15: they then
            // collide with the deferred epilogue - 0.995 / 0.655 against 0.973 / 0.595 ms per 590 848 rows at N = 768 / 512.)
            if (LOADX && s >= 14 && gap < 4 && !(PROBE & 4)) {                      // the next tile's rows into the registers LayerNorm has freed
#pragma unroll
                for (int t = 8 * (s - 14) + 2 * gap; t < 8 * (s - 14) + 2 * gap + 2; ++t) {
                    const rf32x4 q0 = *(const rf32x4 *)(xsrc + 16 * t), q1 = *(const rf32x4 *)(xsrc + 16 * t + 4);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { v[8 * t + u] = q0[u]; v[8 * t + 4 + u] = q1[u]; }
                }
            }
            if (NSWAP && s == 14 && gap >= 4) {                     // a value step comes next: lane = channel, register = row
                const float bb = sbn[32 * (gap - 4)], bwv = sbn[32 * (gap - 4) + 1024];
                rf32x16 &pp = (gap == 4) ? p0 : p1;
#pragma unroll
                for (int r = 0; r < 16; ++r) pp[r] = ((maskh >> (8 * (r >> 2) + (r & 3))) & 1u) ? bwv : bb;
            }
            if (!NSWAP && s == 14 && gap >= 4) {                    // the previous results are out: those registers become the next
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
    RC_DS_READ4_WAIT(A[0][1], an0, 16384, A[0][3], an0, RC_SLOT + 16384, A[0][0], an0, 0, A[0][2], an0, RC_SLOT);   // next step, k-step 0
    if (st) { const unsigned long long t = __builtin_amdgcn_s_memtime(); st[4] += t - ts0; }
}

template <int PROBE, bool KV = false>
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
    const int ngr = a.ngroups > 1 ? a.ngroups : 1;
    int tile = blockIdx.x / ngr;
    const int tstride = (int)gridDim.x / ngr;                        // (the grid is a multiple of ngroups)
    const int j0 = (int)(blockIdx.x % ngr) * (nsteps / ngr), j1 = j0 + nsteps / ngr;     // this workgroup's steps
    if (tile >= ntiles) return;
    char *bounce = smem + RC_OFF_BOUNCE + L.w * RC_BOUNCE;
    unsigned bw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bw[q] = (unsigned)(uintptr_t)(rc_lds_ptr_t)(bounce + L.col * 128 + (((2 * q + L.h) ^ (L.col & 7)) << 4));
    const unsigned br0 = (unsigned)(uintptr_t)(rc_lds_ptr_t)(bounce + (L.lane >> 3) * 128 + (((L.lane & 7) ^ ((L.lane >> 3) & 7)) << 4));
    const int ldo_bytes = (int)(a.ldo * 4);
    const int voff = (32 * L.w + (L.lane >> 3)) * ldo_bytes + (L.lane & 7) * 16;
    const __amdgpu_buffer_rsrc_t rs_none = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0, 0x00020000);   // every store out of range: dropped
    const int pbytes = (int)a.plane_bytes;
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void *)a.planes, 0, KV ? (int)(4 * a.plane_bytes) : 0, 0x00020000);

    const int wbytes = ((a.N + 255) & ~255) * 512;                  // one weight plane: [Npad][256] bf16
    const __amdgpu_buffer_rsrc_t wr_hi = __builtin_amdgcn_make_buffer_rsrc((void *)a.Whi, 0, wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr_lo = __builtin_amdgcn_make_buffer_rsrc((void *)a.Wlo, 0, wbytes, 0x00020000);
    unsigned long long t_step = 0, t_ln = 0, t_drain = 0, t_prev = 0, n_tiles = 0;
    unsigned long long tsl[5] = {0, 0, 0, 0, 0};
    const bool DBG = a.dbg != nullptr;
    if (DBG) t_prev = __builtin_amdgcn_s_memtime();
    auto stamp = [&](unsigned long long &acc_t) { if (DBG) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc_t += t - t_prev; t_prev = t; } };

    // the first step's slots and the first tile's rows (k = 16 s + 8 h + i of row `col`: natural k order); every later tile's rows are
    // requested during the second-to-last step of the tile before it
    rc_issue_rowchunk(L, a.Whi, a.Wlo, 2 * j0, smem);
    rc_issue_rowchunk(L, a.Whi, a.Wlo, 2 * j0 + 1, smem + RC_SLOT);
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
        RC_DS_READ4_WAIT(A[0][1], ad0, 16384, A[0][3], ad0, RC_SLOT + 16384, A[0][0], ad0, 0, A[0][2], ad0, RC_SLOT);
    }
    int gstep = 0;
    for (; tile < ntiles; tile += tstride) {
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
        const bool more = tile + tstride < ntiles;
        const float *xnext = xrow(more ? tile + tstride : tile);

        rf32x16 acc[2][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {                               // the first step's accumulators start from its bias (the later ones' are
            const rf32x4 b0 = *(const rf32x4 *)(tb + 64 * j0 + 8 * q), b1 = *(const rf32x4 *)(tb + 64 * j0 + 32 + 8 * q);   // set inside the step before)
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc[0][0][4 * q + u] = b0[u]; acc[0][1][4 * q + u] = b1[u]; }
        }
        if (!KV) {
        for (int j = j0; j < j1; j += 2) {
            // step j -> acc[0], stores acc[1] (step j - 1); step j + 1 -> acc[1], stores acc[0]
            if (j + 2 == j1)
                rc_ll_step<true, PROBE>(L, a, smem, gstep & 1, acc[0][0], acc[0][1], acc[1][0], acc[1][1], Xh, Xl, 2 * (j + 1), j > j0 ? rs : rs_none, voff,
                                 ldo_bytes, 64 * (j - 1), bw, br0, xnext, v, A, tb + 64 * (j + 1), DBG ? tsl : nullptr, wr_hi, wr_lo);
            else
                rc_ll_step<false, PROBE>(L, a, smem, gstep & 1, acc[0][0], acc[0][1], acc[1][0], acc[1][1], Xh, Xl, 2 * (j + 1), j > j0 ? rs : rs_none, voff,
                                  ldo_bytes, 64 * (j - 1), bw, br0, xnext, v, A, tb + 64 * (j + 1), DBG ? tsl : nullptr, wr_hi, wr_lo);
            ++gstep;
            rc_ll_step<false, PROBE>(L, a, smem, gstep & 1, acc[1][0], acc[1][1], acc[0][0], acc[0][1], Xh, Xl, (j + 2 < j1) ? 2 * (j + 2) : 2 * j0, rs, voff,
                              ldo_bytes, 64 * j, bw, br0, xnext, v, A, tb + 64 * ((j + 2 < j1) ? j + 2 : j0), DBG ? tsl : nullptr, wr_hi, wr_lo);
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
            rc_store_block(L, bounce, o, rs, ldo_bytes, voff, 64 * (j1 - 1) + 32 * blk);
        }
        } else {
        // ---- query | key | value (or key | value) with the keys and values leaving as attention planes --------------------------------------
        // Roles come in runs of an even number of steps (nq in {0, 4}, four key heads, four value heads), so a pair of steps never
        // straddles two roles.  Query steps store through the deferred epilogue of the step behind them as above; a key or value head is
        // written out right behind its step - both planes through the bounce buffer back to back (a wave's LDS operations execute in order:
        // no wait between the write of one plane, its read-back and the write of the next) - and the step behind it then stores nothing
        // (rs_none).  Measured and dropped: the key / value epilogue dealt into the gaps of the next step like the query's (nine
        // instantiations of the step in one kernel: the register allocator parks row fragments in AGPRs and copies them back - 230
        // v_accvgpr moves and 70 scratch accesses in a value step - 1.14 against 0.93 ms per 590 848 rows).
        const int nq = a.nq;
        const unsigned kmask = (unsigned)__builtin_amdgcn_ballot_w64(keep != 0.f);      // valid bits of the wave's 32 rows (lanes 0 - 31)
        const unsigned maskh = kmask >> (4 * L.h);
        const float *sbl = sb + L.col;                               // + 64 j (+ 32): b of the lane's channel; + 1024: b + W beta
        const int kvoff = (m0 + 32 * L.w + (L.lane >> 3)) * 512 + (L.lane & 7) * 16;     // key planes: row 8 it + (lane >> 3), chunk lane & 7
        const int vblk = ((m0 >> 5) + L.w) * 4;                      // value planes: this wave's 32-token block, head 0
        auto split_pair = [&](float x0, float x1, int plane) {
            const unsigned hh = rc_pack2(x0, x1);
            return plane ? rc_pack2(x0 - __builtin_bit_cast(float, hh << 16), x1 - __builtin_bit_cast(float, hh & 0xffff0000u)) : hh;
        };
        auto store_k = [&](const rf32x16 &c0, const rf32x16 &c1, int head) {
            ri32x4 y[2][4];
#pragma unroll
            for (int plane = 0; plane < 2; ++plane) {
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const rf32x16 &c = b ? c1 : c0;
                        *(ru32x2 *)(bounce + L.col * 128 + (((4 * b + q) ^ (L.col & 7)) << 4) + 8 * L.h) =
                            (ru32x2){split_pair(c[4 * q], c[4 * q + 1], plane), split_pair(c[4 * q + 2], c[4 * q + 3], plane)};
                    }
#pragma unroll
                for (int it = 0; it < 4; ++it) y[plane][it] = *(const ri32x4 *)(bounce + it * 1024 + L.lane * 16);
            }
#pragma unroll
            for (int plane = 0; plane < 2; ++plane)
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    __builtin_amdgcn_raw_buffer_store_b128(y[plane][it], prs, kvoff, it * 8 * 512 + head * 128 + plane * pbytes, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        auto store_v = [&](const rf32x16 &c0, const rf32x16 &c1, int head) {
            ri32x4 y[2][4];
#pragma unroll
            for (int plane = 0; plane < 2; ++plane) {
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        const rf32x16 &c = b ? c1 : c0;
                        const int d = L.col + 32 * b, R = d >> 1, sl = (d & 1) * 4 + 2 * cc + L.h;
                        *(ru32x4 *)(bounce + R * 128 + ((sl ^ (R & 7)) << 4)) =
                            (ru32x4){split_pair(c[8 * cc], c[8 * cc + 1], plane), split_pair(c[8 * cc + 2], c[8 * cc + 3], plane),
                                     split_pair(c[8 * cc + 4], c[8 * cc + 5], plane), split_pair(c[8 * cc + 6], c[8 * cc + 7], plane)};
                    }
#pragma unroll
                for (int it = 0; it < 4; ++it) y[plane][it] = *(const ri32x4 *)(bounce + it * 1024 + L.lane * 16);
            }
#pragma unroll
            for (int plane = 0; plane < 2; ++plane)
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    __builtin_amdgcn_raw_buffer_store_b128(y[plane][it], prs, L.lane * 16, it * 1024 + (vblk + head) * 4096 + (2 + plane) * pbytes, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        auto role = [&](int j) { return j < nq ? 0 : (j < nq + 4 ? 1 : 2); };                     // 0 query, 1 key, 2 value
        if (role(j0) == 2) {                                         // a workgroup whose run starts at a value step (ngroups > 1): its first accumulators
#pragma unroll                                                       // in the swapped layout (lane = channel, register = row), as rc_ll_step<.., NSWAP> sets them
            for (int blk = 0; blk < 2; ++blk) {
                const float bb = sbl[64 * j0 + 32 * blk], bwv = sbl[64 * j0 + 32 * blk + 1024];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][blk][r] = ((maskh >> (8 * (r >> 2) + (r & 3))) & 1u) ? bwv : bb;
            }
        }
        auto post = [&](int j, const rf32x16 &c0, const rf32x16 &c1) {
            const int r = role(j);
            if (r == 1) store_k(c0, c1, j - nq);
            else if (r == 2) store_v(c0, c1, j - nq - 4);
        };
        for (int j = j0; j < j1; j += 2) {
            // step j -> acc[0], step j + 1 -> acc[1]; a query step's results are stored by the step behind it
            const int jn = j + 2 < j1 ? j + 2 : j0;                  // the step behind this pair (the next tile's first one at the end of the run)
            const bool val = role(j) == 2, nval = role(jn) == 2;
            const __amdgpu_buffer_rsrc_t rsa = (j > j0 && role(j - 1) == 0) ? rs : rs_none, rsb = role(j) == 0 ? rs : rs_none;
            const int gn = 2 * jn;
            const float *tbn = tb + 64 * jn, *sbn1 = sbl + 64 * (j + 1), *sbn2 = sbl + 64 * jn;
#define RC_STEP_A(LX, SW) rc_ll_step<LX, PROBE, SW, SW>(L, a, smem, gstep & 1, acc[0][0], acc[0][1], acc[1][0], acc[1][1], Xh, Xl, 2 * (j + 1), rsa, voff, ldo_bytes, \
                                                64 * (j - 1), bw, br0, xnext, v, A, tb + 64 * (j + 1), DBG ? tsl : nullptr, wr_hi, wr_lo, sbn1, maskh)
#define RC_STEP_B(SW, NSW) rc_ll_step<false, PROBE, SW, NSW>(L, a, smem, gstep & 1, acc[1][0], acc[1][1], acc[0][0], acc[0][1], Xh, Xl, gn, rsb, voff, ldo_bytes, 64 * j, bw, \
                                                     br0, xnext, v, A, tbn, DBG ? tsl : nullptr, wr_hi, wr_lo, sbn2, maskh)
            if (!val) RC_STEP_A(false, false);
            else if (j + 2 == j1) RC_STEP_A(true, true);
            else RC_STEP_A(false, true);
            ++gstep;
            post(j, acc[0][0], acc[0][1]);
            if (!val) { if (nval) RC_STEP_B(false, true); else RC_STEP_B(false, false); }
            else { if (nval) RC_STEP_B(true, true); else RC_STEP_B(true, false); }
            ++gstep;
            post(j + 1, acc[1][0], acc[1][1]);
#undef RC_STEP_A
#undef RC_STEP_B
        }
        stamp(t_step);
        if (role(j1 - 1) == 0) {                                     // a run that ENDS on a query step (ngroups > 1): nobody behind it stores its results
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                rf32x4 o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int u = 0; u < 4; ++u) o[q][u] = acc[1][blk][4 * q + u];
                rc_store_block(L, bounce, o, rs, ldo_bytes, voff, 64 * (j1 - 1) + 32 * blk);
            }
        }
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

// =================================================================================================================================
// scp_swin_post_attn: everything of a Swin block behind the attention, one launch (see the head of this file).
//
// Per 128-row tile, per wave (32 rows), 37 steps of 96 MFMAs, each step fed by one ring half (two 32 KiB weight slots):
//   steps 0 - 3   phase 0   Y[2j], Y[2j+1] += Wp[64 j .. 64 j + 63][:] . O^T          (B = the attention output rows, read as planes)
//   then          Y += x + bp  (x1 = the new residual stream), LayerNorm statistics, normalised rows -> hi/lo B fragments IN PLACE of O
//   step 4        P1(0), P1(1)   acc1[c] = W1'[32 c .. 32 c + 31][:] . X^T              (hidden chunk c = 32 hidden units)
//   steps 5 - 36  body(c), c = 0 .. 31: per slice one k-step of P1(c + 2) (chain acc1[c & 1]) and one of P2(c): Y[blk] += W2[32 blk ..]
//                 [hidden chunk c] . H(c)^T (chain Y[blk], blk = slice >> 1), and in the gaps GELU(c + 1): acc1[(c + 1) & 1] (started from
//                 b1') -> GELU (rc_gelu_stage) -> hi/lo split -> the B fragments of P2(c + 1).  The accumulator layout IS the fragment layout (lane =
//                 row, 8 consecutive registers = one k-step) up to a fixed permutation of each 16 channels (rc_perm16), which the caller
//                 applies to the K axis of W1' and to the hidden axis of W2 when it tiles them.
//   then          Y + b2 -> fp32 rows out (bounce buffer, whole 128-byte lines).
// The next tile's attention rows are requested into the fragment registers when the last P1 is done (body 30), its residual rows at
// the start of the tile; they are older than the LDS-DMA pieces requested after them, so the step barriers (vmcnt(0)) cover them.
struct RcPostArgs {
    const __bf16 *Ohi, *Olo; int64_t ldo_in;    // attention output planes [M][ldo_in] (256 columns)
    const float *x; int64_t ldx;                // residual stream in, fp32 [M][ldx]
    const void *W;                              // ONE buffer: tiled planes proj hi | fc1 hi | fc2 hi | proj lo | fc1 lo | fc2 lo (RC_W_*):
                                                //   proj [256][256]; fc1 = s (W1 diag(gamma))[:, perm16] [1024][256]; fc2 = W2[:, perm16] / s [256][1024]
                                                //   (s = scp_gelu_prescale(): the activation is evaluated in the variable s y, scp_internal.h)
    const float *bp, *b1, *b2;                  // [256], [1024] (= s (b1 + W1 beta)), [256]
    float *out; int64_t ldc;                    // residual stream out, fp32 [M][ldc] (may be x)
    int M;
    float eps;
    unsigned long long *dbg;
    int dbg_mode;                               // SCP_RC_DUMP: 1 = write the normalised rows instead of the result, 2 = mean / rstd in columns 0, 1
    const int32_t *tile_list; int n_list;       // optional: the 128-row tiles to process, ascending (others - tiles of nothing but window padding - are left alone)
};

#define RC_W_PROJ 0
#define RC_W_FC1 (256 * 512)
#define RC_W_FC2 (RC_W_FC1 + 1024 * 512)
#define RC_W_PLANE (RC_W_FC2 + 256 * 2048)      // bytes of the three hi planes; the lo planes follow in the same order
// LDS-DMA piece pb (1 KiB) of a slot: byte offset base + pb * stride into the weight buffer (lo plane: + RC_W_PLANE).  One buffer
// resource for every weight of the block: nothing selects between descriptors at run time, and four SGPRs hold it instead of 24.
struct RcSlotSrc { int base, stride; };

// The 16 LDS-DMA pieces a wave requests per step: piece pb = w + 4 (s & 3) of slot s >> 2, slices s = 0 .. 7, its hi plane in gap 1 and
// its lo plane in gap 3.  No wave-dependent branch: a step stays ONE basic block, so the compiler has no place to put a copy (or a
// spill) of a fragment register between the inline-asm read that fills it and the hand-placed wait - it would copy stale bytes.
__device__ __forceinline__ void rc_dma_piece(const RcLane &L, __amdgpu_buffer_rsrc_t wr, const RcSlotSrc &src, char *slot_lds, int s, int plane) {
    const int pb = L.w + 4 * (s & 3);
    const int soff = src.base + pb * src.stride + plane * RC_W_PLANE;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (rc_lds_ptr_t)(slot_lds + plane * 16384 + pb * 1024), 16, L.lane * 16, soff, 0, 0);
}

// lane addresses of the fragment reads in a ring half: row-chunk slots (k-chunk 0 / 1) and k-chunk slots (t = 0 / 1)
struct RcFragAddr { unsigned r0, r1, k0, k1; };
__device__ __forceinline__ RcFragAddr rc_frag_addr(const RcLane &L, char *half_base) {
    RcFragAddr f;
    const unsigned b = (unsigned)(uintptr_t)(rc_lds_ptr_t)half_base;
    f.r0 = b + L.rg * 8192 + L.frag; f.r1 = b + L.rg * 8192 + (L.frag ^ 32);
    f.k0 = b + L.rg * 1024 + L.frag; f.k1 = b + L.rg * 1024 + (L.frag ^ 32);
    return f;
}

// Measured and dropped: the MFMAs as inline asm with every operand's register file pinned (accumulators and half of the resident B
// fragments in AGPRs, everything the VALU touches in VGPRs).  gfx950 gives a wave 256 + 256 registers and only MFMA operands may sit in
// the second half; the compiler keeps the two live chains in a[0:31] and parks everything else in the other AGPRs through
// v_accvgpr moves (about 2.7 per MFMA gap).  With asm MFMAs it no longer inserts the wait states between a VALU write of a register
// (its own v_accvgpr copies for the "a" constraints, the zeroing of an accumulator) and the MFMA that reads it: wrong results that
// come and go with timing, and the build that added the nops by hand was slower (186 k against 171 k cycles per tile for the MLP).
#ifdef RC_PROBE_SHAPE16
// Timing probe (WRONG results; tools/r5_shape_probe.sh): every 32x32x16 product as two v_mfma_f32_16x16x32_bf16 on the same operand registers and two quarters of
// the same accumulator - the MACs, the operand reads and the dependency distances of the real step, the other MFMA shape (profiles/r5_mfma_shape.md)
__device__ __forceinline__ void rc_mfma_shape16(rf32x16 &acc, const rbf16x8 &a, const rbf16x8 &b) {
    rf32x4 q0 = {acc[0], acc[1], acc[2], acc[3]}, q1 = {acc[4], acc[5], acc[6], acc[7]};
    q0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, q0, 0, 0, 0);
    q1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, q1, 0, 0, 0);
    acc[0] = q0[0]; acc[1] = q0[1]; acc[2] = q0[2]; acc[3] = q0[3];
    acc[4] = q1[0]; acc[5] = q1[1]; acc[6] = q1[2]; acc[7] = q1[3];
}
#define RC_MFMA_AVA(acc, a, b) rc_mfma_shape16(acc, a, b)
#define RC_MFMA_AVV(acc, a, b) rc_mfma_shape16(acc, a, __builtin_bit_cast(rbf16x8, b))
#else
#define RC_MFMA_AVA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0)
#define RC_MFMA_AVV(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(rbf16x8, b), acc, 0, 0, 0)
#endif
#define RC_MFMA_DRAIN() do { } while (0)

// GEMM step: two row-chunk slots, two chains (c0: slot 0, c1: slot 1) over the 16 k-steps of the resident B fragments.  Used for
// phase 0 (B = attention rows, chains = Y blocks) and for P1(0), P1(1).  NEXT_BODY: the step after this one is a body step (its first
// fragments are a row-chunk and a k-chunk fragment) - they are read in slice 15, after the barrier that publishes the next ring half.
template <bool NEXT_BODY>
__device__ __forceinline__ void rc_gemm_step(const RcLane &L, char *smem, int half, rf32x16 &c0, rf32x16 &c1, const rbf16x8 (&Bh)[16],
                                             const rbf16x8 (&Bl)[16], rbf16x8 (&A)[2][4], __amdgpu_buffer_rsrc_t wr, const RcSlotSrc &n0,
                                             const RcSlotSrc &n1) {
    const RcFragAddr f = rc_frag_addr(L, smem + half * (2 * RC_SLOT)), fn = rc_frag_addr(L, smem + (half ^ 1) * (2 * RC_SLOT));
    char *nxt = smem + (half ^ 1) * (2 * RC_SLOT);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        if (s == 15) RC_STEP_BARRIER("vmcnt(0) lgkmcnt(0)", L.w);
        RC_SB;
        const int nb = (s + 1) & 1, ns = (s + 1) & 15;
#pragma unroll
        for (int gap = 0; gap < 6; ++gap) {
            if (gap < 4 && RC_PROBE_FRAG_ON) { RC_FRAG_WAIT(); RC_SB; }   // the fragment this MFMA is the first to use (see rc_ll_step)
            if (gap == 0) RC_MFMA_AVA(c0, A[s & 1][1], Bh[s]);
            if (gap == 1) RC_MFMA_AVA(c1, A[s & 1][3], Bh[s]);
            if (gap == 2) RC_MFMA_AVV(c0, A[s & 1][0], Bl[s]);
            if (gap == 3) RC_MFMA_AVV(c1, A[s & 1][2], Bl[s]);
            if (gap == 4) RC_MFMA_AVA(c0, A[s & 1][0], Bh[s]);
            if (gap == 5) RC_MFMA_AVA(c1, A[s & 1][2], Bh[s]);
            RC_SB;
            if (s < 15 && RC_PROBE_FRAG_ON) {
                const unsigned ra = (ns & 1) ? f.r1 : f.r0;
                if (gap == 0) RC_DS_READ(A[nb][1], ra, 16384 + (ns >> 1) * 1024);
                if (gap == 1) RC_DS_READ(A[nb][3], ra, RC_SLOT + 16384 + (ns >> 1) * 1024);
                if (gap == 2) RC_DS_READ(A[nb][0], ra, (ns >> 1) * 1024);
                if (gap == 3) RC_DS_READ(A[nb][2], ra, RC_SLOT + (ns >> 1) * 1024);
            }
            if (RC_DMA_HERE(s, gap) && RC_PROBE_DMA_ON) rc_dma_piece(L, wr, RC_DMA_SLOT(s, gap) ? n1 : n0, nxt + RC_DMA_SLOT(s, gap) * RC_SLOT, RC_DMA_IDX(s, gap), RC_DMA_PLANE(s, gap));
            RC_SB;
        }
    }
    // first fragments of the next step, from the other ring half (published by the barrier of slice 15)
    if (NEXT_BODY) RC_DS_READ4_WAIT(A[0][1], fn.r0, 16384, A[0][3], fn.k0, RC_SLOT + 16384, A[0][0], fn.r0, 0, A[0][2], fn.k0, RC_SLOT);
    else RC_DS_READ4_WAIT(A[0][1], fn.r0, 16384, A[0][3], fn.r0, RC_SLOT + 16384, A[0][0], fn.r0, 0, A[0][2], fn.r0, RC_SLOT);
}

// GELU (scp_internal.h: scp_gelu_scaled - max(y', 0) - |y'| exp2(-y'^2) / P4(|y'|); fc1 arrives scaled by s, fc2 by 1 / s), two elements at
// a time, cut into 12 stages: one stage per MFMA gap, so the 26 instructions of a pair (20 of the activation, 6 of the hi / lo split)
// ride in the gaps of two slices.  st = (slice & 1) * 6 + gap.  The accumulator already holds fc1's bias (its start value).
// Rounds 3 - 4 ran the degree-12 erf polynomial here: 48 instructions per pair, 384 of the ~470 vector instructions that shared a
// step's 96 MFMA gaps; the two transcendentals of this form (v_exp_f32, v_rcp_f32) sit six and three stages in front of
// their first use.
struct RcGelu { float ea, eb, pa, pb, wa, wb, xa, xb, ga, gb; unsigned hw; };
__device__ __forceinline__ void rc_gelu_stage(int st, RcGelu &g, float ya, float yb, unsigned &wh, unsigned &wl) {
#ifdef RC_NOGELU   // timing probe (WRONG results): the activation reduced to the split - what the GELU's 10 arithmetic stages cost the launch
    if (st == 0) { g.ga = ya; g.gb = yb; asm volatile("" : "+v"(g.ga), "+v"(g.gb)); }
    if (st == 10) { g.hw = rc_pack2(g.ga, g.gb); wh = g.hw; asm volatile("" : "+v"(g.hw)); }
    if (st == 11) wl = rc_pack2(g.ga - __builtin_bit_cast(float, g.hw << 16), g.gb - __builtin_bit_cast(float, g.hw & 0xffff0000u));
    return;
#endif
    switch (st) {
    case 0: g.ea = -ya * ya; g.eb = -yb * yb; break;
    case 1: g.ea = __builtin_amdgcn_exp2f(g.ea); g.eb = __builtin_amdgcn_exp2f(g.eb); break;
    case 2: g.pa = fmaf(SCP_GELU_C4, __builtin_fabsf(ya), SCP_GELU_C3); g.pb = fmaf(SCP_GELU_C4, __builtin_fabsf(yb), SCP_GELU_C3); break;
    case 3: g.pa = fmaf(g.pa, __builtin_fabsf(ya), SCP_GELU_C2); g.pb = fmaf(g.pb, __builtin_fabsf(yb), SCP_GELU_C2); break;
    case 4: g.pa = fmaf(g.pa, __builtin_fabsf(ya), SCP_GELU_C1); g.pb = fmaf(g.pb, __builtin_fabsf(yb), SCP_GELU_C1); break;
    case 5: g.pa = fmaf(g.pa, __builtin_fabsf(ya), SCP_GELU_C0); g.pb = fmaf(g.pb, __builtin_fabsf(yb), SCP_GELU_C0); break;
    case 6: g.pa = __builtin_amdgcn_rcpf(g.pa); g.pb = __builtin_amdgcn_rcpf(g.pb); break;
    case 7: g.wa = __builtin_fabsf(ya) * g.ea; g.wb = __builtin_fabsf(yb) * g.eb; break;
    case 8: asm("v_max_f32_e32 %0, 0, %1" : "=v"(g.xa) : "v"(ya)); asm("v_max_f32_e32 %0, 0, %1" : "=v"(g.xb) : "v"(yb)); break;   // (fmaxf: + a canonicalising v_max per operand)
    case 9: g.ga = fmaf(-g.wa, g.pa, g.xa); g.gb = fmaf(-g.wb, g.pb, g.xb); break;
    case 10: g.hw = rc_pack2(g.ga, g.gb); wh = g.hw; break;
    default: wl = rc_pack2(g.ga - __builtin_bit_cast(float, g.hw << 16), g.gb - __builtin_bit_cast(float, g.hw & 0xffff0000u)); break;
    }
    // Pin the stage where it is written: the values are pure arithmetic, and without this the optimiser sinks all twelve stages of a pair
    // to the last one's gap (and packs them into v_pk_fma_f32, which is dearer beside MFMAs) - one gap of 26 VALU instructions with the
    // matrix pipe idle instead of 2 - 3 in each of 12 gaps.  An empty asm that "modifies" the live state ends every stage.
    switch (st) {
    case 0: case 1: asm volatile("" : "+v"(g.ea), "+v"(g.eb)); break;
    case 2: case 3: case 4: case 5: case 6: asm volatile("" : "+v"(g.pa), "+v"(g.pb)); break;
    case 7: asm volatile("" : "+v"(g.wa), "+v"(g.wb)); break;
    case 8: asm volatile("" : "+v"(g.xa), "+v"(g.xb)); break;
    case 9: asm volatile("" : "+v"(g.ga), "+v"(g.gb)); break;
    case 10: asm volatile("" : "+v"(g.ga), "+v"(g.gb), "+v"(g.hw)); break;
    default: break;
    }
}

// Body step c: slot 0 = W1' rows [32 (c + 2), +32) (row chunk), slot 1 = W2 hidden slab c (k chunk).
//   HAS_P1: P1(c + 2) -> a1n (false for the last two bodies; a1n arrives holding its bias);  GELU of a1g (= P1(c + 1)) -> Hn (fragments of P2(c + 1));
//   P2(c): Y[blk] += W2 . Hc.   NEXT: 0 = next step is a body, 1 = next step is a GEMM step (phase 0 of the next tile).
template <bool HAS_P1, bool HAS_GELU, int NEXT>
__device__ __forceinline__ void rc_body_step(const RcLane &L, char *smem, int half, rf32x16 (&Y)[8], rf32x16 &a1n, const rf32x16 &a1g,
                                             const ru32x4 (&Hch)[2], const ru32x4 (&Hcl)[2], ru32x4 (&Hnh)[2],
                                             ru32x4 (&Hnl)[2], const rbf16x8 (&Xh)[16], const rbf16x8 (&Xl)[16], rbf16x8 (&A)[2][4],
                                             __amdgpu_buffer_rsrc_t wr, const RcSlotSrc &n0, const RcSlotSrc &n1) {
    const RcFragAddr f = rc_frag_addr(L, smem + half * (2 * RC_SLOT)), fn = rc_frag_addr(L, smem + (half ^ 1) * (2 * RC_SLOT));
    char *nxt = smem + (half ^ 1) * (2 * RC_SLOT);
    RcGelu g;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        if (s == 15) RC_STEP_BARRIER("vmcnt(0) lgkmcnt(0)", L.w);
        RC_SB;
        const int nb = (s + 1) & 1, ns = (s + 1) & 15;
        const int blk = s >> 1, t = s & 1;
#pragma unroll
        for (int gap = 0; gap < 6; ++gap) {
            // the fragment this MFMA is the first to use (see rc_ll_step); where P1 is over only the two W2 fragments are read per slice -
            // a read nothing uses is a register the compiler hands to the next VALU instruction while the read is still in flight
            if (HAS_P1 && gap < 4 && RC_PROBE_FRAG_ON) { RC_FRAG_WAIT(); RC_SB; }
            if (!HAS_P1 && (gap == 1 || gap == 3) && RC_PROBE_FRAG_ON) { asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory"); RC_SB; }
            // P1 k-step s on chain a1n (fragments A[.][0] hi, A[.][1] lo) and P2 (blk, t) on chain Y[blk] (A[.][2] hi, A[.][3] lo), alternating
            if (gap == 0 && HAS_P1) RC_MFMA_AVA(a1n, A[s & 1][1], Xh[s]);
            if (gap == 1) RC_MFMA_AVV(Y[blk], A[s & 1][3], Hch[t]);
            if (gap == 2 && HAS_P1) RC_MFMA_AVV(a1n, A[s & 1][0], Xl[s]);
            if (gap == 3) RC_MFMA_AVV(Y[blk], A[s & 1][2], Hcl[t]);
            if (gap == 4 && HAS_P1) RC_MFMA_AVA(a1n, A[s & 1][0], Xh[s]);
            if (gap == 5) RC_MFMA_AVV(Y[blk], A[s & 1][2], Hch[t]);
            RC_SB;
            if (s < 15 && RC_PROBE_FRAG_ON) {
                if (gap == 0 && HAS_P1) RC_DS_READ(A[nb][1], (ns & 1) ? f.r1 : f.r0, 16384 + (ns >> 1) * 1024);
                if (gap == 1) RC_DS_READ(A[nb][3], (ns & 1) ? f.k1 : f.k0, RC_SLOT + 16384 + (ns >> 1) * 2048);
                if (gap == 2 && HAS_P1) RC_DS_READ(A[nb][0], (ns & 1) ? f.r1 : f.r0, (ns >> 1) * 1024);
                if (gap == 3) RC_DS_READ(A[nb][2], (ns & 1) ? f.k1 : f.k0, RC_SLOT + (ns >> 1) * 2048);
            }
            if (RC_DMA_HERE(s, gap) && RC_PROBE_DMA_ON) rc_dma_piece(L, wr, RC_DMA_SLOT(s, gap) ? n1 : n0, nxt + RC_DMA_SLOT(s, gap) * RC_SLOT, RC_DMA_IDX(s, gap), RC_DMA_PLANE(s, gap));
            if (HAS_GELU) {   // element pair (2 k, 2 k + 1) of the finished P1 chain, k = s >> 1: registers 2 k, 2 k + 1 = fragment (k >> 2), elements 2 (k & 3), + 1
                const int k = s >> 1, st = (s & 1) * 6 + gap;
                unsigned wh = 0, wl = 0;
                rc_gelu_stage(st, g, a1g[2 * k], a1g[2 * k + 1], wh, wl);
                if (st == 10) Hnh[k >> 2][k & 3] = wh;
                if (st == 11) Hnl[k >> 2][k & 3] = wl;
            }
            RC_SB;
        }
    }
    // first fragments of the next step, from the other ring half (published by the barrier of slice 15)
    if (NEXT == 0) RC_DS_READ4_WAIT(A[0][1], fn.r0, 16384, A[0][3], fn.k0, RC_SLOT + 16384, A[0][0], fn.r0, 0, A[0][2], fn.k0, RC_SLOT);
    else RC_DS_READ4_WAIT(A[0][1], fn.r0, 16384, A[0][3], fn.r0, RC_SLOT + 16384, A[0][0], fn.r0, 0, A[0][2], fn.r0, RC_SLOT);
}

__global__ __launch_bounds__(256, 1) void rc_post_attn_kernel(const RcPostArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const RcLane L = rc_lane();
    const int ntiles = (a.M + RC_ROWS - 1) / RC_ROWS;
    float *sbp = (float *)(smem + RC_OFF_BIAS), *sb2 = sbp + 256, *sb1 = sbp + 512;      // bp[256], b2[256], b1'[1024]
    for (int i = threadIdx.x; i < 256; i += 256) { sbp[i] = a.bp[i]; sb2[i] = a.b2[i]; }
    for (int i = threadIdx.x; i < 1024; i += 256) sb1[i] = a.b1[i];
    __syncthreads();
    // tile sequence of this workgroup: every gridDim.x-th entry of the tile list (all tiles without a list)
    const int nlist = a.tile_list ? a.n_list : ntiles;
    auto tile_of = [&](int i) { return a.tile_list ? a.tile_list[i] : i; };
    int ti = blockIdx.x;
    if (ti >= nlist) return;
    int tile = tile_of(ti);
    char *bounce = smem + RC_OFF_BOUNCE + L.w * RC_BOUNCE;
    const int ldc_bytes = (int)(a.ldc * 4);
    const int voff = (32 * L.w + (L.lane >> 3)) * ldc_bytes + (L.lane & 7) * 16;

    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)a.W, 0, 2 * RC_W_PLANE, 0x00020000);
    auto src_p = [&](int g) { RcSlotSrc r = {RC_W_PROJ + g * 16384, 1024}; return r; };             // proj rows [32 g, +32): 16 consecutive KiB
    auto src_1 = [&](int c) { RcSlotSrc r = {RC_W_FC1 + c * 16384, 1024}; return r; };              // fc1 rows [32 c, +32)
    auto src_2 = [&](int c) { RcSlotSrc r = {RC_W_FC2 + c * 1024, 32 * 1024}; return r; };          // fc2 hidden slab c (32 columns) of the 16 row groups

    unsigned long long t_p0 = 0, t_ln = 0, t_mlp = 0, t_epi = 0, t_prev = 0, n_tiles = 0;
    const bool DBG = a.dbg != nullptr;
    unsigned long long t_first = 0, r_first = 0;
    if (DBG) { t_prev = t_first = __builtin_amdgcn_s_memtime(); r_first = __builtin_amdgcn_s_memrealtime(); }
    auto stamp = [&](unsigned long long &acc_t) { if (DBG) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc_t += t - t_prev; t_prev = t; } };

    // (Measured and dropped: starting the workgroups an eighth of a tile time apart, so that the 256 epilogues - 128 KB of row stores per
    // CU, 20 k cycles per tile - do not hit HBM together: 2.366 against 2.376 ms per 590 848 rows, epilogue 20.9 k against 20.8 k cycles: the
    // epilogue is not bandwidth-bound.)
    // first step's slots (proj rows 0 .. 63) and the first tile's attention rows (B fragments: k = 16 s + 8 h + i, natural order)
    {
        const RcSlotSrc s0 = src_p(0), s1 = src_p(1);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int plane = 0; plane < 2; ++plane) { rc_dma_piece(L, wr, s0, smem, q, plane); rc_dma_piece(L, wr, s1, smem + RC_SLOT, q, plane); }
    }
    rbf16x8 Bh[16], Bl[16];                                         // attention rows during phase 0, normalised rows afterwards
    // The attention rows of a tile are fetched COALESCED like the residual rows (see phase 0): raw chunk [4 p + it] = row 8 it + (lane >> 3),
    // 16-byte chunk lane & 7 of the row's 128-byte line p (k-steps 4 p .. 4 p + 3), and turned into B fragments (lane = row, k = 16 s + 8 h
    // + i: chunk 2 (s & 3) + h of line s >> 2) through the bounce buffer once the tile before is stored: eight passes of 4 writes + 4 reads.
    auto load_o = [&](int t) {
        const int r0 = t * RC_ROWS + 32 * L.w + (L.lane >> 3);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = r0 + 8 * it;
            const int64_t o = (int64_t)(r < a.M ? r : a.M - 1) * a.ldo_in + (L.lane & 7) * 8;
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) { Bh[4 * pp + it] = *(const rbf16x8 *)(a.Ohi + o + 64 * pp); Bl[4 * pp + it] = *(const rbf16x8 *)(a.Olo + o + 64 * pp); }
        }
    };
    auto transpose_o = [&]() {
#pragma unroll
        for (int plane = 0; plane < 2; ++plane)
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int rho = 8 * it + (L.lane >> 3), kap = L.lane & 7;
                    *(rbf16x8 *)(bounce + rho * 128 + ((kap ^ (rho & 7)) << 4)) = plane ? Bl[4 * pp + it] : Bh[4 * pp + it];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const rbf16x8 fr = *(const rbf16x8 *)(bounce + L.col * 128 + (((2 * q + L.h) ^ (L.col & 7)) << 4));
                    if (plane) Bl[4 * pp + q] = fr; else Bh[4 * pp + q] = fr;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
    };
    load_o(tile);
    transpose_o();
    rbf16x8 A[2][4];
    {
        SCP_BARRIER_DMA(0);
        const RcFragAddr f = rc_frag_addr(L, smem);
        RC_DS_READ4_WAIT(A[0][1], f.r0, 16384, A[0][3], f.r0, RC_SLOT + 16384, A[0][0], f.r0, 0, A[0][2], f.r0, RC_SLOT);
    }
    int gstep = 0;
    for (; ti < nlist; ti += gridDim.x) {
        tile = tile_of(ti);
        const int m0 = tile * RC_ROWS;
        const bool more = ti + (int)gridDim.x < nlist;
        const int tile_next = more ? tile_of(ti + (int)gridDim.x) : tile;
        rf32x16 Y[8];
        // ---- phase 0: x1 = x + bp + proj(attention rows): four steps of two 32-channel blocks (ONE copy of the step's code; the unrolled
        // form that accumulates straight into Y[2 j], Y[2 j + 1] is 680 instructions longer and no faster).  The residual
        // rows of a step's 64 channels (accumulator layout: channel 32 b + 8 q + 4 h + u) are requested in front of the step - older than
        // its LDS-DMA pieces, so the step's barrier covers them - and added behind it.
        // The residual rows are fetched COALESCED - an instruction reads 8 rows x one whole 128-byte line (row 8 it + (lane >> 3), chunk
        // lane & 7 of a 32-channel block) - and turned into the accumulator layout through the wave's bounce buffer behind the step (the
        // reverse of rc_store_block).  Every lane fetching from its OWN row (32 rows x 32 bytes per instruction, every line touched by
        // four instructions) cost 10 % of the launch: 31.3 k cycles of phase 0 against 21.2 k with the loads compiled out, and 10 k more in
        // the MLP phase, whose weight DMA shares the address path; coalesced: 29.2 k and 152 k (2.29 -> 2.22 ms per 590 848 rows).
        // Measured and dropped: the rows of step j + 1 requested inside step j behind its DMA pieces (barrier with vmcnt(8), two
        // alternating register sets, phase 0 unrolled by two): the second set pushes the steps into scratch spills - 78 k cycles.
        const int xrow = m0 + 32 * L.w + (L.lane >> 3);
        const float *xsrc = a.x + (L.lane & 7) * 4;
        for (int j = 0; j < 4; ++j) {
            rf32x4 xc[2][4];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int r = xrow + 8 * it;
                    xc[b][it] = *(const rf32x4 *)(xsrc + (int64_t)(r < a.M ? r : a.M - 1) * a.ldx + 64 * j + 32 * b);
                }
            rf32x16 c0, c1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
            rc_gemm_step<false>(L, smem, gstep & 1, c0, c1, Bh, Bl, A, wr, j < 3 ? src_p(2 * j + 2) : src_1(0), j < 3 ? src_p(2 * j + 3) : src_1(1));
            ++gstep;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int rho = 8 * it + (L.lane >> 3), kap = L.lane & 7;
                    *(rf32x4 *)(bounce + rho * 128 + ((kap ^ (rho & 7)) << 4)) = xc[b][it];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const rf32x4 xq = *(const rf32x4 *)(bounce + L.col * 128 + (((2 * q + L.h) ^ (L.col & 7)) << 4));
                    const rf32x4 bq = *(const rf32x4 *)(sbp + 64 * j + 32 * b + 8 * q + 4 * L.h);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (b == 0) c0[4 * q + u] += xq[u] + bq[u];
                        else c1[4 * q + u] += xq[u] + bq[u];
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the buffer is free for the next block
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            switch (j) {
            case 0: Y[0] = c0; Y[1] = c1; break;
            case 1: Y[2] = c0; Y[3] = c1; break;
            case 2: Y[4] = c0; Y[5] = c1; break;
            default: Y[6] = c0; Y[7] = c1; break;
            }
        }
        stamp(t_p0);
        // ---- LayerNorm statistics of x1 over the row (this lane's 128 channels + lane ^ 32's) -------------------------------------------------
        float sum = 0.f;
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int r = 0; r < 16; r += 4) sum += (Y[b][r] + Y[b][r + 1]) + (Y[b][r + 2] + Y[b][r + 3]);
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / 256.0f);
        float sq = 0.f;
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = Y[b][r] - mean; sq += d * d; }
        sq += __shfl_xor(sq, 32);
        const float rstd = rsqrtf(sq * (1.0f / 256.0f) + a.eps);
        // normalised rows -> B fragments of fc1 (k-step 2 b + t = registers 8 t .. 8 t + 7 of block b); Y += b2 (the MLP output bias)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float fr[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) fr[i] = (Y[b][8 * t + i] - mean) * rstd;
                rc_split8(fr, Bh[2 * b + t], Bl[2 * b + t]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const rf32x4 bb = *(const rf32x4 *)(sb2 + 32 * b + 8 * q + 4 * L.h);
#pragma unroll
                for (int u = 0; u < 4; ++u) Y[b][4 * q + u] += bb[u];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(t_ln);
        if (a.dbg_mode) {
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int t = r >> 3, i = r & 7;
                    Y[b][r] = (a.dbg_mode == 1) ? (float)Bh[2 * b + t][i] + (float)Bl[2 * b + t][i] : (r == 0 ? mean : rstd);
                }
        }
        // ---- P1(0), P1(1): every P1 chain starts from its 32 biases (b1' = s (b1 + W1 beta), accumulator order: register 4 q + u =
        // channel 8 q + 4 h + u of the chunk) ----------------------------------------------------------------------------------------
        rf32x16 a1[2];
        auto bias_into = [&](rf32x16 &acc, int c) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const rf32x4 bq = *(const rf32x4 *)(sb1 + 32 * c + 8 * q + 4 * L.h);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[4 * q + u] = bq[u];
            }
        };
        bias_into(a1[0], 0);
        bias_into(a1[1], 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!a.dbg_mode) {
        rc_gemm_step<true>(L, smem, gstep & 1, a1[0], a1[1], Bh, Bl, A, wr, src_1(2), src_2(0)); ++gstep;
        // ---- GELU(0) (nothing to hide it behind yet) -----------------------------------------------------------------------------------
        ru32x4 Hh[1][2], Hl[1][2];                                     // H(c): hi / lo fragments (bf16 pairs) of the two k-steps of a hidden chunk
        {
            RcGelu g;
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int st = 0; st < 12; ++st)
                {
                    unsigned wh = 0, wl = 0;
                    rc_gelu_stage(st, g, a1[0][2 * k], a1[0][2 * k + 1], wh, wl);
                    if (st == 10) Hh[0][k >> 2][k & 3] = wh;
                    if (st == 11) Hl[0][k >> 2][k & 3] = wl;
                }
        }
        // ---- bodies: H(c) in (Hch, Hcl), P1(c + 1) in a1g; body c leaves H(c + 1) in (Hnh, Hnl) and P1(c + 2) in a1n; then they swap -----
        ru32x4 Hnh[2], Hnl[2];
        for (int c = 0; c < 30; ++c) {
            rf32x16 a1n;
            bias_into(a1n, c + 2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            rc_body_step<true, true, 0>(L, smem, gstep & 1, Y, a1n, a1[1], Hh[0], Hl[0], Hnh, Hnl, Bh, Bl, A, wr,
                                        c + 3 < 32 ? src_1(c + 3) : src_2(c + 1), src_2(c + 1));
            ++gstep;
            a1[1] = a1n;
#pragma unroll
            for (int t = 0; t < 2; ++t) { Hh[0][t] = Hnh[t]; Hl[0][t] = Hnl[t]; }
        }
        // body 30: no P1 left; GELU(31).  The fragment registers are free: the next tile's attention rows are requested here.  (They cost 7 k
        // cycles per tile - probe without them: MLP phase 144.8 k against 151.9 k.  Requested INSIDE the step behind its DMA pieces, one per
        // MFMA gap of slices 8 - 15 with the barrier leaving them in flight, they cost 7.5 k more: 159.4 k.)
        load_o(tile_next);
        rc_body_step<false, true, 0>(L, smem, gstep & 1, Y, a1[0], a1[1], Hh[0], Hl[0], Hnh, Hnl, Bh, Bl, A, wr, src_2(31), src_2(31));
        ++gstep;
        // body 31: P2 only; the step after it is phase 0 of the next tile
        rc_body_step<false, false, 1>(L, smem, gstep & 1, Y, a1[0], a1[1], Hnh, Hnl, Hh[0], Hl[0], Bh, Bl, A, wr, src_p(0), src_p(1));
        ++gstep;
        }
        stamp(t_mlp);
        // ---- x2 = Y -> fp32 rows -------------------------------------------------------------------------------------------------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int64_t rows_left = (int64_t)a.M - m0;
        const int64_t span = (rows_left < RC_ROWS ? rows_left : RC_ROWS) * a.ldc * 4;
#ifdef RC_EPI_NOSTORE      // timing probe (WRONG results): every output store out of range, dropped at the address unit - 2.17 against 2.26 ms per 590 848 rows (4 %)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0, 0x00020000);
#else
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.out + (int64_t)m0 * a.ldc, 0, (int)span, 0x00020000);
#endif
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            rf32x4 o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int u = 0; u < 4; ++u) o[q][u] = Y[b][4 * q + u];
            rc_store_block(L, bounce, o, rs, ldc_bytes, voff, 32 * b);
        }
        // (Measured and dropped, round 4: four blocks per pass through the ring half body 31 has just finished with - 16 KiB of bounce space per wave,
        // 2 + 2 exposed LDS round trips for the stores and the transposition instead of 16 + 16, a barrier behind it - 2.167 - 2.184 against 2.158 ms per
        // 590 848 rows: the epilogue's 20 k cycles are not LDS latency either.)
        transpose_o();                                             // the next tile's attention rows (requested in front of body 30) -> B fragments
        stamp(t_epi);
        ++n_tiles;
    }
    if (DBG && L.lane == 0) {
        unsigned long long *o = a.dbg + ((size_t)blockIdx.x * 4 + L.w) * 8;
        o[0] = t_p0; o[1] = t_mlp; o[2] = t_ln; o[3] = t_epi; o[4] = n_tiles;
        o[5] = __builtin_amdgcn_s_memtime() - t_first; o[6] = __builtin_amdgcn_s_memrealtime() - r_first;     // shader clock = o[5] / o[6] x 100 MHz
    }
    SCP_WAIT_DMA(0);
}

// =================================================================================================================================
// rc_post_attn_wide_kernel: the SAME block as rc_post_attn_kernel for SHORT launches (round 5; the decoder's one-window forwards: a few dozen
// 128-row tiles on 256 CUs, every launch as long as one wave's serial chain of 3 552 products - 93 us whatever M is).  Here a workgroup owns
// 32 rows and its four waves split the OUTPUT CHANNELS of every product instead of the rows:
//   proj   : wave w computes x1 blocks 2 w, 2 w + 1 (64 of the 256 channels) for all 32 rows              96 products
//   (x1 blocks meet in LDS; every wave then holds the whole rows, runs LayerNorm on them and keeps the normalised rows as B fragments)
//   fc1    : wave w computes hidden chunks w, w + 4, .., w + 28 (two chains at a time), GELU, hi / lo split -> H(c) into LDS     384 products
//   fc2    : wave w accumulates output blocks 2 w, 2 w + 1 over all 32 hidden chunks in order                                     384 products
// 864 products per wave instead of 3 552.  Every output element is still ONE wave's accumulation chain with the same operands in the same
// order as in rc_post_attn_kernel (same start values, same three partial products per k-step, same GELU routine, same LayerNorm reduction
// order), so a row's result has the same bits whichever kernel computed it - the decoder (short launches) and the encoder (one packed
// launch) keep agreeing on every integer CDF (tests/test_gpu_model.py::test_wide_post_attn_kernel_has_the_chain_kernels_bits).
// The weights do not go through LDS (every wave needs different ones; 128 KiB of the LDS hold H): a lane reads its 16-byte fragment of the
// tiled planes straight from L2, a few slices ahead.
// fragments in flight: k-steps (proj, fc1: 16 registers each) / hidden chunks (fc2: 32 registers each).  Measured 4 / 2, 8 / 4, 12 / 4, 8 / 8, 16 / 4:
// 46.5 / 46.9 / 49.3 / 49.4 / 51.6 us per 512-row launch - the kernel is not waiting for these loads; a lone wave per SIMD pays for ISSUING them
// (a vector load to registers holds the wave's issue for ~200 cycles, tools/src/mb_vmem_issue.cpp): 47 - 57 us against the chain kernel's 79 - 88
#ifndef RCW_D1
#define RCW_D1 4
#endif
#ifndef RCW_D2
#define RCW_D2 2
#endif
#define RCW_H_BYTES (32 * 4096)                     // H(c): [32 chunks][64 lanes][4 x 16 B: hi t = 0, 1, lo t = 0, 1]
__global__ __launch_bounds__(256, 1) void rc_post_attn_wide_kernel(const RcPostArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const RcLane L = rc_lane();
    float *sbp = (float *)(smem + RC_OFF_BIAS), *sb2 = sbp + 256, *sb1 = sbp + 512;      // bp[256], b2[256], b1'[1024]
    for (int i = threadIdx.x; i < 256; i += 256) { sbp[i] = a.bp[i]; sb2[i] = a.b2[i]; }
    for (int i = threadIdx.x; i < 1024; i += 256) sb1[i] = a.b1[i];
    __syncthreads();
    // item = a 32-row tile: sub-tile (item & 3) of 128-row tile tile_list[item >> 2] (all tiles without a list)
    const int n128 = a.tile_list ? a.n_list : (a.M + RC_ROWS - 1) / RC_ROWS;
    char *bounce = smem + RC_OFF_BOUNCE + L.w * RC_BOUNCE;
    char *hbuf = smem;                                                                    // H(c); the x1 blocks meet here first
    const int ldc_bytes = (int)(a.ldc * 4);
    const int voff = (L.lane >> 3) * ldc_bytes + (L.lane & 7) * 16;
    // Weight fragments come by buffer loads: ONE resource for the whole weight buffer, the lane's offset inside a slot image in a VGPR (two of
    // them: k-step parity), everything else - slot, k-slab, plane - in the scalar offset.  (With plain pointers the compiler precomputes a 64-bit
    // address per fragment and spills them: 230 scratch accesses per step.)
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)a.W, 0, 2 * RC_W_PLANE, 0x00020000);
    const int vrc0 = L.rg * 8192 + L.frag, vrc1 = L.rg * 8192 + (L.frag ^ 32);          // row-chunk slot image: + (s >> 1) * 1024, k-step parity s & 1
    const int vkc0 = L.rg * 32768 + L.frag, vkc1 = L.rg * 32768 + (L.frag ^ 32);        // fc2 k-slab pieces: row group rg of a block, t = 0 / 1
    auto wload = [&](int voff, int soff) { return __builtin_bit_cast(rbf16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, voff, soff, 0)); };
    for (int item = blockIdx.x; item < 4 * n128; item += gridDim.x) {
        const int t128 = a.tile_list ? a.tile_list[item >> 2] : (item >> 2);
        const int m0 = t128 * RC_ROWS + 32 * (item & 3);
        if (m0 >= a.M) continue;                                                           // (uniform over the workgroup)
        // ---- the tile's attention rows as B fragments (every wave all 32 rows): coalesced fetch + transposition as in the chain kernel ----
        rbf16x8 Bh[16], Bl[16];
        {
            const int r0 = m0 + (L.lane >> 3);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = r0 + 8 * it;
                const int64_t o = (int64_t)(r < a.M ? r : a.M - 1) * a.ldo_in + (L.lane & 7) * 8;
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) { Bh[4 * pp + it] = *(const rbf16x8 *)(a.Ohi + o + 64 * pp); Bl[4 * pp + it] = *(const rbf16x8 *)(a.Olo + o + 64 * pp); }
            }
#pragma unroll
            for (int plane = 0; plane < 2; ++plane)
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int rho = 8 * it + (L.lane >> 3), kap = L.lane & 7;
                        *(rbf16x8 *)(bounce + rho * 128 + ((kap ^ (rho & 7)) << 4)) = plane ? Bl[4 * pp + it] : Bh[4 * pp + it];
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const rbf16x8 fr = *(const rbf16x8 *)(bounce + L.col * 128 + (((2 * q + L.h) ^ (L.col & 7)) << 4));
                        if (plane) Bl[4 * pp + q] = fr; else Bh[4 * pp + q] = fr;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
        }
        // ---- proj: blocks 2 w (chain c0) and 2 w + 1 (chain c1); weight rows [64 w, 64 w + 64) = row-chunk slots 2 w, 2 w + 1 ----------------
        rf32x16 c0, c1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
        {
            const int w0 = RC_W_PROJ + (2 * L.w) * 16384, w1 = w0 + 16384;
            constexpr int D = RCW_D1;                                                     // slices of fragments in flight
            rbf16x8 F[D][4];                                                              // [.][0] hi slot 0, [1] lo slot 0, [2] hi slot 1, [3] lo slot 1
            auto fetch = [&](int s, rbf16x8 (&d)[4]) {
                const int v = (s & 1) ? vrc1 : vrc0, o = (s >> 1) * 1024;
                d[0] = wload(v, w0 + o); d[1] = wload(v, w0 + RC_W_PLANE + o);
                d[2] = wload(v, w1 + o); d[3] = wload(v, w1 + RC_W_PLANE + o);
            };
#pragma unroll
            for (int s = 0; s < D; ++s) fetch(s, F[s]);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                rbf16x8 (&A)[4] = F[s % D];
                RC_MFMA_AVA(c0, A[1], Bh[s]); RC_MFMA_AVA(c1, A[3], Bh[s]);
                RC_MFMA_AVA(c0, A[0], Bl[s]); RC_MFMA_AVA(c1, A[2], Bl[s]);
                RC_MFMA_AVA(c0, A[0], Bh[s]); RC_MFMA_AVA(c1, A[2], Bh[s]);
                if (s + D < 16) fetch(s + D, F[s % D]);
                RC_SB;
            }
        }
        // x1 = x + bp + proj: the lane's own row, channels 64 w + 32 b + 8 q + 4 h + u
        {
            const int r = m0 + L.col;
            const float *xr = a.x + (int64_t)(r < a.M ? r : a.M - 1) * a.ldx + 64 * L.w + 4 * L.h;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const rf32x4 xq = *(const rf32x4 *)(xr + 32 * b + 8 * q);
                    const rf32x4 bq = *(const rf32x4 *)(sbp + 64 * L.w + 32 * b + 8 * q + 4 * L.h);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (b == 0) c0[4 * q + u] += xq[u] + bq[u];
                        else c1[4 * q + u] += xq[u] + bq[u];
                    }
                }
        }
        // ---- the eight x1 blocks meet in LDS ([block][lane][16 floats]); every wave takes all of them ------------------------------------------
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *(rf32x4 *)(hbuf + ((2 * L.w) * 64 + L.lane) * 64 + 16 * q) = (rf32x4){c0[4 * q], c0[4 * q + 1], c0[4 * q + 2], c0[4 * q + 3]};
            *(rf32x4 *)(hbuf + ((2 * L.w + 1) * 64 + L.lane) * 64 + 16 * q) = (rf32x4){c1[4 * q], c1[4 * q + 1], c1[4 * q + 2], c1[4 * q + 3]};
        }
        __syncthreads();
        // ---- LayerNorm statistics and the normalised rows: the chain kernel's arithmetic, operation for operation.  The blocks are re-read from
        // LDS in every pass (a wave has 256 registers the vector unit can read; eight blocks + the fragments would be all of them) --------------
        auto yblock = [&](int b, rf32x16 &y) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const rf32x4 t = *(const rf32x4 *)(hbuf + (b * 64 + L.lane) * 64 + 16 * q);
#pragma unroll
                for (int u = 0; u < 4; ++u) y[4 * q + u] = t[u];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        float sum = 0.f;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            rf32x16 y;
            yblock(b, y);
#pragma unroll
            for (int r = 0; r < 16; r += 4) sum += (y[r] + y[r + 1]) + (y[r + 2] + y[r + 3]);
        }
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / 256.0f);
        float sq = 0.f;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            rf32x16 y;
            yblock(b, y);
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = y[r] - mean; sq += d * d; }
        }
        sq += __shfl_xor(sq, 32);
        const float rstd = rsqrtf(sq * (1.0f / 256.0f) + a.eps);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            rf32x16 y;
            yblock(b, y);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float fr[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) fr[i] = (y[8 * t + i] - mean) * rstd;
                rc_split8(fr, Bh[2 * b + t], Bl[2 * b + t]);
            }
        }
        // this wave's two output blocks start from x1 + b2
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            rf32x16 y;
            yblock(2 * L.w + b, y);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const rf32x4 bb = *(const rf32x4 *)(sb2 + 32 * (2 * L.w + b) + 8 * q + 4 * L.h);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (b == 0) c0[4 * q + u] = y[4 * q + u] + bb[u];
                    else c1[4 * q + u] = y[4 * q + u] + bb[u];
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                                                   // the exchange area becomes H
        // ---- fc1 + GELU: hidden chunks w + 8 i (chain a1a) and w + 8 i + 4 (chain a1b), i = 0 .. 3 -------------------------------------------------
        for (int i = 0; i < 4; ++i) {
            const int ca = L.w + 8 * i, cb = ca + 4;
            rf32x16 a1a, a1b;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const rf32x4 ba = *(const rf32x4 *)(sb1 + 32 * ca + 8 * q + 4 * L.h), bb = *(const rf32x4 *)(sb1 + 32 * cb + 8 * q + 4 * L.h);
#pragma unroll
                for (int u = 0; u < 4; ++u) { a1a[4 * q + u] = ba[u]; a1b[4 * q + u] = bb[u]; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int w0 = RC_W_FC1 + ca * 16384, w1 = RC_W_FC1 + cb * 16384;
            constexpr int D = RCW_D1;
            rbf16x8 F[D][4];
            auto fetch = [&](int s, rbf16x8 (&d)[4]) {
                const int v = (s & 1) ? vrc1 : vrc0, o = (s >> 1) * 1024;
                d[0] = wload(v, w0 + o); d[1] = wload(v, w0 + RC_W_PLANE + o);
                d[2] = wload(v, w1 + o); d[3] = wload(v, w1 + RC_W_PLANE + o);
            };
#pragma unroll
            for (int s = 0; s < D; ++s) fetch(s, F[s]);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                rbf16x8 (&A)[4] = F[s % D];
                RC_MFMA_AVA(a1a, A[1], Bh[s]); RC_MFMA_AVA(a1b, A[3], Bh[s]);
                RC_MFMA_AVA(a1a, A[0], Bl[s]); RC_MFMA_AVA(a1b, A[2], Bl[s]);
                RC_MFMA_AVA(a1a, A[0], Bh[s]); RC_MFMA_AVA(a1b, A[2], Bh[s]);
                if (s + D < 16) fetch(s + D, F[s % D]);
                RC_SB;
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const rf32x16 &acc = half ? a1b : a1a;
                ru32x4 Hh[2], Hl[2];
                RcGelu g;
#pragma unroll
                for (int k = 0; k < 8; ++k)
#pragma unroll
                    for (int st = 0; st < 12; ++st) {
                        unsigned wh = 0, wl = 0;
                        rc_gelu_stage(st, g, acc[2 * k], acc[2 * k + 1], wh, wl);
                        if (st == 10) Hh[k >> 2][k & 3] = wh;
                        if (st == 11) Hl[k >> 2][k & 3] = wl;
                    }
                char *hp = hbuf + ((half ? cb : ca) * 64 + L.lane) * 64;
                *(ru32x4 *)(hp) = Hh[0]; *(ru32x4 *)(hp + 16) = Hh[1]; *(ru32x4 *)(hp + 32) = Hl[0]; *(ru32x4 *)(hp + 48) = Hl[1];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- fc2: blocks 2 w (c0), 2 w + 1 (c1) over the hidden chunks in order; W2 k-slab c, row groups 2 blk + rg --------------------------------
        {
            const int wf = RC_W_FC2 + (2 * (2 * L.w)) * 32768;                           // block 2 w (row groups 4 w, 4 w + 1); block 2 w + 1: + 65536
            constexpr int D = RCW_D2;                                                     // chunks of fragments in flight
            rbf16x8 F[D][8];                                                              // [t][plane][blk]: index 4 t + 2 plane + blk
            auto fetch = [&](int c, rbf16x8 (&d)[8]) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int plane = 0; plane < 2; ++plane)
#pragma unroll
                        for (int blk = 0; blk < 2; ++blk)
                            d[4 * t + 2 * plane + blk] = wload(t ? vkc1 : vkc0, wf + c * 1024 + plane * RC_W_PLANE + blk * 65536);
            };
#pragma unroll
            for (int c = 0; c < D; ++c) fetch(c, F[c]);
            auto chunk = [&](int c, rbf16x8 (&A)[8]) {
                const char *hp = hbuf + (c * 64 + L.lane) * 64;
                const ru32x4 Hh0 = *(const ru32x4 *)(hp), Hh1 = *(const ru32x4 *)(hp + 16), Hl0 = *(const ru32x4 *)(hp + 32), Hl1 = *(const ru32x4 *)(hp + 48);
                // t = 0, then t = 1; per t: lo . hi, hi . lo, hi . hi (A index 4 t + 2 plane + blk)
                RC_MFMA_AVV(c0, A[2], Hh0); RC_MFMA_AVV(c1, A[3], Hh0);
                RC_MFMA_AVV(c0, A[0], Hl0); RC_MFMA_AVV(c1, A[1], Hl0);
                RC_MFMA_AVV(c0, A[0], Hh0); RC_MFMA_AVV(c1, A[1], Hh0);
                RC_MFMA_AVV(c0, A[6], Hh1); RC_MFMA_AVV(c1, A[7], Hh1);
                RC_MFMA_AVV(c0, A[4], Hl1); RC_MFMA_AVV(c1, A[5], Hl1);
                RC_MFMA_AVV(c0, A[4], Hh1); RC_MFMA_AVV(c1, A[5], Hh1);
                if (c + D < 32) fetch(c + D, A);
                RC_SB;
            };
            for (int c = 0; c < 32; c += D) {
#pragma unroll
                for (int k = 0; k < D; ++k) chunk(c + k, F[k]);
            }
        }
        // ---- x2 -> fp32 rows (the wave's 64 channels of the tile's 32 rows) ----------------------------------------------------------------------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int64_t rows_left = (int64_t)a.M - m0;
        const int64_t span = (rows_left < 32 ? rows_left : 32) * a.ldc * 4;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.out + (int64_t)m0 * a.ldc, 0, (int)span, 0x00020000);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            rf32x4 o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int u = 0; u < 4; ++u) o[q][u] = b ? c1[4 * q + u] : c0[4 * q + u];
            rc_store_block(L, bounce, o, rs, ldc_bytes, voff, 64 * L.w + 32 * b);
        }
        __syncthreads();                                                                   // H is free for the next item
    }
}

// =================================================================================================================================
// scp_swin_merge: SwinPatchMerging (swin_transformer.py:350-384) in one launch - gather the (even, odd) token of every pair, LayerNorm
// over the 512 concatenated channels, 512 -> 256 reduction (no bias) - for M merged rows.  Replaces layernorm_rows(gather) + gemm_split.
// A wave holds its 32 merged rows; the K = 512 product runs as two K = 256 halves over the SAME eight accumulator blocks Y: half 0
// (the even tokens' channels) as fragments while the raw odd-token rows wait in registers, then those become the fragments of half 1.
// Weights: W' = W diag(gamma) as two [256][256] matrices (columns 0 - 255 / 256 - 511), each tiled like the attention projection, in a
// buffer with RC_W_PLANE bytes between the hi and the lo planes (hi: W'0 at 0, W'1 at 128 KiB); the accumulators start from W beta.
struct RcMergeArgs {
    const float *x; int64_t ldx; int64_t n_src;     // source rows [n_src][ldx], 256 channels; index n_src = a row of zeros
    const int64_t *ia, *ib;                         // [M]: even / odd token of the pair
    const void *W; const float *wbeta;              // see above; [256]
    float *out; int64_t ldo; int M; float eps;
};

__global__ __launch_bounds__(256, 1) void rc_merge_kernel(const RcMergeArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const RcLane L = rc_lane();
    const int ntiles = (a.M + RC_ROWS - 1) / RC_ROWS;
    float *swb = (float *)(smem + RC_OFF_BIAS);
    for (int i = threadIdx.x; i < 256; i += 256) swb[i] = a.wbeta ? a.wbeta[i] : 0.f;
    __syncthreads();
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    char *bounce = smem + RC_OFF_BOUNCE + L.w * RC_BOUNCE;
    const int ldo_bytes = (int)(a.ldo * 4);
    const int voff = (32 * L.w + (L.lane >> 3)) * ldo_bytes + (L.lane & 7) * 16;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)a.W, 0, 2 * RC_W_PLANE, 0x00020000);
    auto src = [&](int t) { RcSlotSrc r = {((t >> 3) & 1) * 131072 + (t & 7) * 16384, 1024}; return r; };   // slot t = 8 kh + r: rows [32 r, +32) of half kh
    {   // the first step's slots
        const RcSlotSrc s0 = src(0), s1 = src(1);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int plane = 0; plane < 2; ++plane) { rc_dma_piece(L, wr, s0, smem, q, plane); rc_dma_piece(L, wr, s1, smem + RC_SLOT, q, plane); }
    }
    rbf16x8 A[2][4];
    {
        SCP_BARRIER_DMA(0);
        const RcFragAddr f = rc_frag_addr(L, smem);
        RC_DS_READ4_WAIT(A[0][1], f.r0, 16384, A[0][3], f.r0, RC_SLOT + 16384, A[0][0], f.r0, 0, A[0][2], f.r0, RC_SLOT);
    }
    int gstep = 0;
    auto load_half = [&](const int64_t *idx, int t, float (&v)[128]) {
        const int r = t * RC_ROWS + 32 * L.w + L.col;
        const int64_t i = idx[r < a.M ? r : a.M - 1];
        const float k = i < a.n_src ? 1.f : 0.f;
        const float *p = a.x + (i < a.n_src ? i : a.n_src - 1) * a.ldx + 8 * L.h;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const rf32x4 q0 = *(const rf32x4 *)(p + 16 * s), q1 = *(const rf32x4 *)(p + 16 * s + 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) { v[8 * s + u] = q0[u] * k; v[8 * s + 4 + u] = q1[u] * k; }
        }
    };
    // (Measured and dropped: the next tile's even rows requested behind step 3, when their registers are free - 0.420 against 0.396 ms
    // per 303 616 merged rows: the loads are older than step 4's weight pieces, so its barrier waits for them.)
    for (; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * RC_ROWS;
        float va[128], vb[128];
        load_half(a.ia, tile, va);
        load_half(a.ib, tile, vb);
        // LayerNorm statistics over the 512 channels (this lane's 256 + lane ^ 32's), two-pass like layernorm_rows_kernel
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 128; i += 4) sum += ((va[i] + va[i + 1]) + (va[i + 2] + va[i + 3])) + ((vb[i] + vb[i + 1]) + (vb[i + 2] + vb[i + 3]));
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / 512.0f);
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 128; ++i) { const float da = va[i] - mean, db = vb[i] - mean; sq += da * da + db * db; }
        sq += __shfl_xor(sq, 32);
        const float rstd = rsqrtf(sq * (1.0f / 512.0f) + a.eps);
        rbf16x8 Xh[16], Xl[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float f[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = (va[8 * s + i] - mean) * rstd;
            rc_split8(f, Xh[s], Xl[s]);
        }
        rf32x16 Y[8];
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const rf32x4 bb = *(const rf32x4 *)(swb + 32 * b + 8 * q + 4 * L.h);
#pragma unroll
                for (int u = 0; u < 4; ++u) Y[b][4 * q + u] = bb[u];
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // eight steps: half kh, output channels [64 g, +64) -> Y[2 g], Y[2 g + 1]; a step requests the next one's slots (t + 1, wrapping
        // into the next tile's first step)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int g = t & 3, tn = (t + 1) & 7;
            rc_gemm_step<false>(L, smem, gstep & 1, Y[2 * g], Y[2 * g + 1], Xh, Xl, A, wr, src(2 * (tn & 3) + 8 * (tn >> 2)), src(2 * (tn & 3) + 1 + 8 * (tn >> 2)));
            ++gstep;
            if (t == 3) {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    float f[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) f[i] = (vb[8 * s + i] - mean) * rstd;
                    rc_split8(f, Xh[s], Xl[s]);
                }
            }
        }
        // ---- Y -> fp32 rows ----------------------------------------------------------------------------------------------------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int64_t rows_left = (int64_t)a.M - m0;
        const int64_t span = (rows_left < RC_ROWS ? rows_left : RC_ROWS) * a.ldo * 4;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.out + (int64_t)m0 * a.ldo, 0, (int)span, 0x00020000);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            rf32x4 o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int u = 0; u < 4; ++u) o[q][u] = Y[b][4 * q + u];
            rc_store_block(L, bounce, o, rs, ldo_bytes, voff, 32 * b);
        }
    }
    SCP_WAIT_DMA(0);
}

extern "C" SCP_API int scp_swin_merge(const float *x, int64_t ldx, int64_t n_src, const int64_t *ia, const int64_t *ib, const void *W, const float *wbeta,
                                      float eps, float *out, int64_t ldo, int32_t M, void *stream);

// =================================================================================================================================
// scp_geo_edge_mlps: the two edge MLPs of the geometry feature generator (dgcnn.py:121-151; ehem.py geo_feat_generator.edge_mlp1 / edge_mlp2)
// for M points in one launch:   e1 = mlp1(cat(pos1, pos2, pos3))  (448 -> 256 -> 256 -> 256)
//                               out = mlp2(cat(pos3, e1))          (512 -> 256 -> 256 -> 128)       LeakyReLU(0.01) between the layers
// Six dense layers, 30 steps of rc_gemm_step per 128-row tile; every activation between two layers stays in the accumulators and
// becomes the next layer's B fragments (columns of the next weight in accumulator order, rc_perm16), e1 never exists in memory.
// Replaces six split-GEMM launches and four fp32 -> plane conversion launches.  Weights: eight tiled [256][256] matrices (hi planes at
// m * 128 KiB, lo planes RC_W_PLANE bytes behind): 0 / 1 = mlp1 layer 1 columns [0, 256) / [256, 448) (zero padded); 2, 3 = mlp1 layers
// 2, 3 (perm16 columns); 4 = mlp2 layer 1 columns [256, 512) (e1: perm16); 5 = its columns [0, 256) (pos3); 6 = mlp2 layer 2 (perm16);
// 7 = mlp2 layer 3 (128 rows, perm16 columns).  bias: b11 | b12 | b13 | b21 | b22 (256 each) | b23 (128).
struct RcEdgeArgs {
    const float *p1, *p2, *p3; int64_t ld1, ld2, ld3;
    const void *W; const float *bias;
    float *out; int64_t ldo; int M;
};

__device__ __forceinline__ float rc_leaky(float y) { return y > 0.f ? y : 0.01f * y; }

__global__ __launch_bounds__(256, 1) void rc_edge_mlp_kernel(const RcEdgeArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const RcLane L = rc_lane();
    const int ntiles = (a.M + RC_ROWS - 1) / RC_ROWS;
    float *sb = (float *)(smem + RC_OFF_BIAS);
    for (int i = threadIdx.x; i < 5 * 256 + 128; i += 256) sb[i] = a.bias[i];
    __syncthreads();
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    char *bounce = smem + RC_OFF_BOUNCE + L.w * RC_BOUNCE;
    const int ldo_bytes = (int)(a.ldo * 4);
    const int voff = (32 * L.w + (L.lane >> 3)) * ldo_bytes + (L.lane & 7) * 16;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)a.W, 0, 2 * RC_W_PLANE, 0x00020000);
    // step t = 4 m + g (m < 7), 28 + g (m = 7, g < 2): rows [64 g, +64) of matrix m; the step behind the last is the next tile's first
    // (the two k halves of a first layer alternate per 64 output channels, so that an accumulator pair is finished before the next starts)
    auto src = [&](int t, int k) {
        int m, g;
        if (t < 8) { m = t & 1; g = t >> 1; }
        else if (t < 16) { m = t >> 2; g = t & 3; }
        else if (t < 24) { m = 4 + (t & 1); g = (t - 16) >> 1; }
        else if (t < 28) { m = 6; g = t & 3; }
        else { m = 7; g = t - 28; }
        RcSlotSrc r = {m * 131072 + (2 * g + k) * 16384, 1024};
        return r;
    };
    {
        const RcSlotSrc s0 = src(0, 0), s1 = src(0, 1);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int plane = 0; plane < 2; ++plane) { rc_dma_piece(L, wr, s0, smem, q, plane); rc_dma_piece(L, wr, s1, smem + RC_SLOT, q, plane); }
    }
    rbf16x8 A[2][4];
    {
        SCP_BARRIER_DMA(0);
        const RcFragAddr f = rc_frag_addr(L, smem);
        RC_DS_READ4_WAIT(A[0][1], f.r0, 16384, A[0][3], f.r0, RC_SLOT + 16384, A[0][0], f.r0, 0, A[0][2], f.r0, RC_SLOT);
    }
    int gstep = 0, t = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * RC_ROWS;
        const int row = m0 + 32 * L.w + L.col;
        const int64_t rowc = row < a.M ? row : a.M - 1;
        const float *q1 = a.p1 + rowc * a.ld1 + 8 * L.h, *q2 = a.p2 + rowc * a.ld2 + 8 * L.h, *q3 = a.p3 + rowc * a.ld3 + 8 * L.h;
        rbf16x8 Xh[16], Xl[16], Zh[16], Zl[16];
        rf32x16 Y[8];
        auto frag_of = [&](const float *p, rbf16x8 &hi, rbf16x8 &lo) {          // 8 consecutive channels of this lane's row
            const rf32x4 v0 = *(const rf32x4 *)p, v1 = *(const rf32x4 *)(p + 4);
            const float f[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            rc_split8(f, hi, lo);
        };
        auto to_frags = [&](rbf16x8 (&Fh)[16], rbf16x8 (&Fl)[16], bool act) {   // Y (accumulator layout) -> B fragments of the next layer
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    float fr[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) fr[i] = act ? rc_leaky(Y[b][8 * tt + i]) : Y[b][8 * tt + i];
                    rc_split8(fr, Fh[2 * b + tt], Fl[2 * b + tt]);
                }
        };
        // One pair of accumulator blocks at a time: start from the bias, one step per k half, write the pair into Y (written, never read,
        // through the run-time switch: read that way the compiler keeps Y in scratch memory).  ONE copy of a step's code per use.
#define RC_STEP(Fh, Fl) { const int tn = t + 1 < 30 ? t + 1 : 0; rc_gemm_step<false>(L, smem, gstep & 1, c0, c1, Fh, Fl, A, wr, src(tn, 0), src(tn, 1)); ++gstep; t = tn; }
#define RC_LAYER(LAYER, NG, STEPS)                                                                                                             \
        for (int g = 0; g < (NG); ++g) {                                                                                                       \
            rf32x16 c0, c1;                                                                                                                    \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                                    \
                const rf32x4 b0 = *(const rf32x4 *)(sb + 256 * (LAYER) + 64 * g + 8 * q + 4 * L.h), b1 = *(const rf32x4 *)(sb + 256 * (LAYER) + 64 * g + 32 + 8 * q + 4 * L.h); \
                _Pragma("unroll") for (int u = 0; u < 4; ++u) { c0[4 * q + u] = b0[u]; c1[4 * q + u] = b1[u]; }                                 \
            }                                                                                                                                  \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                                 \
            STEPS                                                                                                                              \
            switch (g) { case 0: Y[0] = c0; Y[1] = c1; break; case 1: Y[2] = c0; Y[3] = c1; break; case 2: Y[4] = c0; Y[5] = c1; break; default: Y[6] = c0; Y[7] = c1; break; } \
        }
        // ---- mlp1 layer 1: cat(pos1 (64), pos2 (128), pos3 (256)) as k halves [0, 256) and [256, 448) + 64 zero columns ----------------------------
#pragma unroll
        for (int s = 0; s < 16; ++s) frag_of(s < 4 ? q1 + 16 * s : (s < 12 ? q2 + 16 * (s - 4) : q3 + 16 * (s - 12)), Xh[s], Xl[s]);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (s < 12) frag_of(q3 + 64 + 16 * s, Zh[s], Zl[s]);
            else { Zh[s] = __builtin_bit_cast(rbf16x8, (ru32x4){0u, 0u, 0u, 0u}); Zl[s] = Zh[s]; }
        }
        RC_LAYER(0, 4, RC_STEP(Xh, Xl) RC_STEP(Zh, Zl))
        to_frags(Xh, Xl, true);
        RC_LAYER(1, 4, RC_STEP(Xh, Xl))                                          // mlp1 layer 2
        to_frags(Xh, Xl, true);
        RC_LAYER(2, 4, RC_STEP(Xh, Xl))                                          // mlp1 layer 3 -> e1
        // ---- mlp2 layer 1: cat(pos3, e1): e1 straight from the accumulators, pos3 read again ------------------------------------------------------
        to_frags(Xh, Xl, false);
#pragma unroll
        for (int s = 0; s < 16; ++s) frag_of(q3 + 16 * s, Zh[s], Zl[s]);
        RC_LAYER(3, 4, RC_STEP(Xh, Xl) RC_STEP(Zh, Zl))
        to_frags(Xh, Xl, true);
        RC_LAYER(4, 4, RC_STEP(Xh, Xl))                                          // mlp2 layer 2
        to_frags(Xh, Xl, true);
        RC_LAYER(5, 2, RC_STEP(Xh, Xl))                                          // mlp2 layer 3: 128 outputs
#undef RC_LAYER
#undef RC_STEP
        // ---- Y[0 .. 3] -> fp32 rows ------------------------------------------------------------------------------------------------------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int64_t rows_left = (int64_t)a.M - m0;
        const int64_t span = (rows_left < RC_ROWS ? rows_left : RC_ROWS) * a.ldo * 4;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.out + (int64_t)m0 * a.ldo, 0, (int)span, 0x00020000);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            rf32x4 o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int u = 0; u < 4; ++u) o[q][u] = Y[b][4 * q + u];
            rc_store_block(L, bounce, o, rs, ldo_bytes, voff, 32 * b);
        }
    }
    SCP_WAIT_DMA(0);
}

// =================================================================================================================================
// scp_mlp3_rows: a three-layer head Sequential(Linear, LeakyReLU, Linear, LeakyReLU, Linear) on 256-channel rows in ONE launch (round 6):
// prob_pred_mlp1 (256 -> 256 -> 256 -> 255, ehem.py:113-115: even-node logits) and pre_attn_mlp (256 -> 256 -> 240 -> 240, ehem.py:117-121).
// As three split-GEMM launches these layers are HBM-bound (K = N = 256: 2 KB of planes moved per row and layer for 0.39 MFLOP, 0.20 - 0.24 of
// the matrix roof at 3 TB/s); chained through the accumulators (rc_edge_mlp_kernel's scheme) a row is read once (1 KB, fp32, optionally
// gathered: the even tokens of the self branch) and its result written once (optionally scattered: logits straight to their coding-order rows).
// 12 steps of rc_gemm_step per 128-row tile.  Weights: three tiled [256][256] matrices (rows / columns beyond the layer's zero; hi planes at
// m * 128 KiB, lo planes RC_W_PLANE bytes behind; layers 2, 3 with their columns in accumulator order, rc_perm16); bias: 3 x 256 (zero padded).
// Results per row: independent of what else is in the launch.
struct RcMlp3Args {
    const float *x; int64_t ldx; const int64_t *in_map; int64_t n_src;   // row m reads x[in_map ? in_map[m] : m] (clamped to n_src - 1)
    const void *W; const float *bias;
    float *out; int64_t ldo; const int64_t *out_map;                     // row m -> out[out_map ? out_map[m] : m]; negative: dropped
    int M, N;                                                            // N: output columns written (<= 256, % 4 == 0)
};

__global__ __launch_bounds__(256, 1) void rc_mlp3_kernel(const RcMlp3Args a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const RcLane L = rc_lane();
    const int ntiles = (a.M + RC_ROWS - 1) / RC_ROWS;
    float *sb = (float *)(smem + RC_OFF_BIAS);
    for (int i = threadIdx.x; i < 3 * 256; i += 256) sb[i] = a.bias[i];
    __syncthreads();
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    char *bounce = smem + RC_OFF_BOUNCE + L.w * RC_BOUNCE;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)a.W, 0, 2 * RC_W_PLANE, 0x00020000);
    // step t = 4 m + g: rows [64 g, +64) of matrix m; the step behind the last is the next tile's first
    auto src = [&](int t, int k) {
        RcSlotSrc r = {(t >> 2) * 131072 + (2 * (t & 3) + k) * 16384, 1024};
        return r;
    };
    {
        const RcSlotSrc s0 = src(0, 0), s1 = src(0, 1);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int plane = 0; plane < 2; ++plane) { rc_dma_piece(L, wr, s0, smem, q, plane); rc_dma_piece(L, wr, s1, smem + RC_SLOT, q, plane); }
    }
    rbf16x8 A[2][4];
    {
        SCP_BARRIER_DMA(0);
        const RcFragAddr f = rc_frag_addr(L, smem);
        RC_DS_READ4_WAIT(A[0][1], f.r0, 16384, A[0][3], f.r0, RC_SLOT + 16384, A[0][0], f.r0, 0, A[0][2], f.r0, RC_SLOT);
    }
    int gstep = 0, t = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * RC_ROWS;
        const int row = m0 + 32 * L.w + L.col;
        int64_t srow = row < a.M ? row : a.M - 1;
        if (a.in_map) srow = a.in_map[srow];
        srow = srow < a.n_src ? (srow < 0 ? 0 : srow) : a.n_src - 1;
        const float *q0 = a.x + srow * a.ldx + 8 * L.h;
        rbf16x8 Xh[16], Xl[16];
        rf32x16 Y[8];
        auto to_frags = [&]() {   // LeakyReLU(Y) (accumulator layout) -> B fragments of the next layer
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    float fr[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) fr[i] = rc_leaky(Y[b][8 * tt + i]);
                    rc_split8(fr, Xh[2 * b + tt], Xl[2 * b + tt]);
                }
        };
#define RC_STEP3() { const int tn = t + 1 < 12 ? t + 1 : 0; rc_gemm_step<false>(L, smem, gstep & 1, c0, c1, Xh, Xl, A, wr, src(tn, 0), src(tn, 1)); ++gstep; t = tn; }
#define RC_BIAS3(LAYER)                                                                                                                        \
            rf32x16 c0, c1;                                                                                                                    \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                                    \
                const rf32x4 b0 = *(const rf32x4 *)(sb + 256 * (LAYER) + 64 * g + 8 * q + 4 * L.h), b1 = *(const rf32x4 *)(sb + 256 * (LAYER) + 64 * g + 32 + 8 * q + 4 * L.h); \
                _Pragma("unroll") for (int u = 0; u < 4; ++u) { c0[4 * q + u] = b0[u]; c1[4 * q + u] = b1[u]; }                                 \
            }                                                                                                                                  \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define RC_LAYER3(LAYER)                                                                                                                       \
        for (int g = 0; g < 4; ++g) {                                                                                                          \
            RC_BIAS3(LAYER)                                                                                                                    \
            RC_STEP3()                                                                                                                         \
            switch (g) { case 0: Y[0] = c0; Y[1] = c1; break; case 1: Y[2] = c0; Y[3] = c1; break; case 2: Y[4] = c0; Y[5] = c1; break; default: Y[6] = c0; Y[7] = c1; break; } \
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const rf32x4 v0 = *(const rf32x4 *)(q0 + 16 * s), v1 = *(const rf32x4 *)(q0 + 16 * s + 4);
            const float f[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            rc_split8(f, Xh[s], Xl[s]);
        }
        RC_LAYER3(0)
        to_frags();
        RC_LAYER3(1)
        to_frags();
        // the rows this lane stores (4 groups of 8 rows of the wave's 32) and where they go
        const int kap = L.lane & 7;
        float *orow[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = m0 + 32 * L.w + 8 * it + (L.lane >> 3);
            const int64_t o = r < a.M ? (a.out_map ? a.out_map[r] : (int64_t)r) : -1;
            orow[it] = o >= 0 ? a.out + o * a.ldo + 4 * kap : nullptr;
        }
        // one block (32 output channels x the wave's 32 rows, accumulator layout) -> fp32 rows, columns [ch0, ch0 + 32) below N: through the wave's
        // bounce buffer (rc_store_block's exchange), stored by row pointer
        auto store_block = [&](const rf32x16 &c, int ch0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                rf32x4 v;
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = c[4 * q + u];
                *(rf32x4 *)(bounce + L.col * 128 + (((2 * q + L.h) ^ (L.col & 7)) << 4)) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            rf32x4 y[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int rho = 8 * it + (L.lane >> 3);
                y[it] = *(const rf32x4 *)(bounce + rho * 128 + ((kap ^ (rho & 7)) << 4));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 4; ++it)
                if (orow[it] && ch0 + 4 * kap < a.N) *(rf32x4 *)(orow[it] + ch0) = y[it];
        };
        // last layer: every pair of blocks leaves as soon as its step is done (kept in Y until the end of the layer, the 128 accumulator registers stay
        // live beside the 128 fragment registers through four steps: the compiler spilled 240 of them)
        for (int g = 0; g < 4; ++g) {
            RC_BIAS3(2)
            RC_STEP3()
            store_block(c0, 64 * g);
            store_block(c1, 64 * g + 32);
        }
#undef RC_LAYER3
#undef RC_BIAS3
#undef RC_STEP3
    }
    SCP_WAIT_DMA(0);
}

extern "C" SCP_API double scp_gelu_prescale(void) { return (double)SCP_GELU_S; }

// workgroups per tile of a short rc_ln_linear launch (RcLnLinArgs.ngroups): the largest divisor g of nsteps / 2 with ntiles * g <= the CU count
// (every workgroup then owns exactly one (tile, run of steps) and the launch lasts LayerNorm + nsteps / g steps instead of LayerNorm + nsteps)
static int rc_ll_groups(int ntiles, int nsteps, int ncu) {
    static int on = -1;
    if (on < 0) { const char *e = getenv("SCP_RC_GROUPS"); on = (e && e[0] == '0') ? 0 : 1; }      // SCP_RC_GROUPS=0: A/B bracket (identical bits)
    int best = 1;
    if (on)
        for (int g = 2; g <= nsteps / 2; ++g)
            if ((nsteps / 2) % g == 0 && (int64_t)ntiles * g <= ncu) best = g;
    return best;
}

static unsigned long long *g_rc_dbg = nullptr;   // diagnostic only (tools/mb_rowchain_probe.py): [workgroup][wave][8] cycle sums
extern "C" SCP_API int scp_rc_debug_buffer(unsigned long long *dev_buf) { g_rc_dbg = dev_buf; return SCP_OK; }

static int g_rc_num_cu = 0, g_rc_grid = 0;
// scp_debug.h: which kernel scp_swin_post_attn launches: -1 (default) by size, 0 the chain kernel always, 1 the wide kernel always (tests; SCP_RC_WIDE)
static int g_rc_wide = -2;
static int rc_wide_mode() {
    if (g_rc_wide == -2) { const char *e = getenv("SCP_RC_WIDE"); g_rc_wide = e ? atoi(e) : -1; if (g_rc_wide < -1 || g_rc_wide > 1) g_rc_wide = -1; }
    return g_rc_wide;
}
extern "C" SCP_API int scp_rc_set_wide(int32_t mode) { g_rc_wide = (mode < -1 || mode > 1) ? -1 : mode; return SCP_OK; }
// measurement hook (scp_debug.h): number of persistent workgroups of the row-chain launches; 0 = one per CU of the device.  For streams
// created with a CU mask (tools/mb_cumask.py) - a persistent grid larger than the stream's CU set would run in two rounds.
extern "C" SCP_API int scp_rc_set_grid(int32_t workgroups) { g_rc_grid = workgroups > 0 ? workgroups : 0; return SCP_OK; }
static int rc_num_cu() {
    if (g_rc_grid) return g_rc_grid;
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
        configured = true;
    }
    RcLnLinArgs a;
    a.x = x; a.ldx = ldx; a.valid = valid; a.Whi = (const __bf16 *)Whi; a.Wlo = (const __bf16 *)Wlo; a.bias = bias; a.wbeta = wbeta;
    a.out = out; a.ldo = ldo; a.M = M; a.N = N; a.eps = eps;
    a.planes = nullptr; a.plane_bytes = 0; a.nq = 0;
    { static int pr = -1; if (pr < 0) { const char *e = getenv("SCP_RC_PROBE"); pr = e ? atoi(e) : 0; } a.probe = pr; }
    if (a.probe) { const char *e = getenv("SCP_RC_PROBE"); a.probe = e ? atoi(e) : 0; }
    a.dbg = g_rc_dbg;
    const int ntiles = (M + RC_ROWS - 1) / RC_ROWS;
    const int ncu = rc_num_cu();
    SCP_PROF(SCP_PROF_LN_LINEAR, stream, 2.0 * M * 256.0 * N);
    a.ngroups = rc_ll_groups(ntiles, N / 64, ncu);
    hipLaunchKernelGGL(rc_ln_linear_kernel<0>, dim3((unsigned)(a.ngroups > 1 ? ntiles * a.ngroups : (ntiles < ncu ? ntiles : ncu))), dim3(256), RC_LDS, (hipStream_t)stream, a);
    LAUNCH_CHECK();
    return SCP_OK;
}

// scp_swin_ln_linear for the query | key | value projection of a Swin block (N = 768) or the key | value projection of a cross layer
// (N = 512) with the keys and values leaving as the bf16 hi / lo planes of the plane-fed window attention (scp_swin_attention_packed_planes):
// q: fp32 [M][ldq] (N = 768 only), planes: [4][Tp][256] bf16 (K hi, K lo, V^T hi, V^T lo; Tp >= M rows, Tp % 32 == 0).  M % 128 == 0 (rows of
// the packed layout come in 512s).  The planes hold the bits scp_swin_kv_planes makes of scp_swin_ln_linear's fp32 k and v.
extern "C" SCP_API int scp_swin_ln_qkv(const float *x, int64_t ldx, const float *valid, const void *Whi, const void *Wlo, const float *bias,
                                       const float *wbeta, float eps, float *q, int64_t ldq, void *planes, int64_t Tp, int32_t M, int32_t N, void *stream) {
    if (!x || !Whi || !Wlo || !planes || M <= 0 || (M & 127) || (N != 768 && N != 512) || (N == 768 && (!q || ldq < 256 || (ldq & 3))) || ldx < 256 ||
        (ldx & 3) || Tp < M || (Tp & 31) || (((uintptr_t)x | (uintptr_t)q | (uintptr_t)Whi | (uintptr_t)Wlo | (uintptr_t)planes) & 15) ||
        (q && (int64_t)RC_ROWS * ldq * 4 > 0x7fffffffLL) || 4 * Tp * 512 > 0x7fffffffLL)
        return SCP_EINVAL;
    static bool configured = false;
    if (!configured) {
        HIP_TRY(hipFuncSetAttribute((const void *)rc_ln_linear_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS));
        configured = true;
    }
    RcLnLinArgs a;
    a.x = x; a.ldx = ldx; a.valid = valid; a.Whi = (const __bf16 *)Whi; a.Wlo = (const __bf16 *)Wlo; a.bias = bias; a.wbeta = wbeta;
    a.out = q; a.ldo = q ? ldq : 256; a.M = M; a.N = N; a.eps = eps; a.probe = 0; a.dbg = nullptr;
    a.planes = (__bf16 *)planes; a.plane_bytes = Tp * 512; a.nq = N / 64 - 8;
    const int ntiles = M / RC_ROWS, ncu = rc_num_cu();
    SCP_PROF(SCP_PROF_LN_LINEAR, stream, 2.0 * M * 256.0 * N);
    a.ngroups = rc_ll_groups(ntiles, N / 64, ncu);
    hipLaunchKernelGGL((rc_ln_linear_kernel<0, true>), dim3((unsigned)(a.ngroups > 1 ? ntiles * a.ngroups : (ntiles < ncu ? ntiles : ncu))), dim3(256), RC_LDS, (hipStream_t)stream, a);
    LAUNCH_CHECK();
    return SCP_OK;
}

// x2 = x1 + fc2(GELU(fc1(LayerNorm(x1)))) with x1 = x + proj(o): the whole Swin block behind the attention (swin_transformer.py:503-571,
// 662-706) for M rows.  o: attention output as bf16 hi/lo planes [M][ldo_in]; x: fp32 [M][ldx]; out: fp32 [M][ldc] (may be x).
// W: ONE device buffer of scp_swin_post_attn_weight_bytes() bytes holding the tiled planes (scp_split_weight_bf16 + scp_tile_weight_bf16)
// proj hi | fc1 hi | fc2 hi | proj lo | fc1 lo | fc2 lo with proj [256][256]; fc1 = (W1 diag(gamma))[:, rc_perm16] [1024][256]; fc2 =
// W2[:, rc_perm16] [256][1024]; rc_perm16: inside every group of 16 columns, columns 4 - 7 and 8 - 11 change places.  b1 = the fc1 bias
// + W1 beta (the LayerNorm affine folded in).
extern "C" SCP_API int scp_swin_post_attn(const void *Ohi, const void *Olo, int64_t ldo_in, const float *x, int64_t ldx, const void *W, const float *bp,
                                          const float *b1, const float *b2, float eps, float *out, int64_t ldc, int32_t M, const int32_t *tile_list,
                                          int32_t n_tiles, void *stream) {
    if (!Ohi || !Olo || !x || !W || !bp || !b1 || !b2 || !out || M <= 0 || (tile_list && (n_tiles <= 0 || n_tiles > (M + RC_ROWS - 1) / RC_ROWS)) || ldo_in < 256 || (ldo_in & 7) || ldx < 256 || (ldx & 3) || ldc < 256 ||
        (ldc & 3) || (((uintptr_t)Ohi | (uintptr_t)Olo | (uintptr_t)x | (uintptr_t)out | (uintptr_t)W) & 15) ||
        (int64_t)RC_ROWS * ldc * 4 > 0x7fffffffLL)
        return SCP_EINVAL;
    static bool configured = false;
    if (!configured) {
        HIP_TRY(hipFuncSetAttribute((const void *)rc_post_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS));
        HIP_TRY(hipFuncSetAttribute((const void *)rc_post_attn_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS));
        configured = true;
    }
    RcPostArgs a;
    a.Ohi = (const __bf16 *)Ohi; a.Olo = (const __bf16 *)Olo; a.ldo_in = ldo_in; a.x = x; a.ldx = ldx; a.W = W;
    a.bp = bp; a.b1 = b1; a.b2 = b2; a.out = out; a.ldc = ldc; a.M = M; a.eps = eps;
    a.dbg = g_rc_dbg;
    { const char *e = getenv("SCP_RC_DUMP"); a.dbg_mode = e ? atoi(e) : 0; }
    a.tile_list = tile_list; a.n_list = tile_list ? n_tiles : 0;
    const int ntiles = tile_list ? n_tiles : (M + RC_ROWS - 1) / RC_ROWS;
    const int ncu = rc_num_cu();
    // algorithmic work: the rows of the tiles processed (a tile list leaves out tiles of nothing but window padding)
    SCP_PROF(SCP_PROF_POST_ATTN, stream, 2.0 * (tile_list ? (double)n_tiles * RC_ROWS : (double)M) * (256.0 * 256.0 + 2.0 * 256.0 * 1024.0));
    // short launches: the wide kernel (a workgroup per 32 rows, the waves split the output channels; same bits) while its 32-row tiles fit the
    // chip (measured: 47 - 57 against 79 - 88 us up to 8 192 rows, 110 against 98 us at 16 384) - beyond that the chain kernel's 128-row tiles
    // fill the CUs and stream every weight byte once per 128 rows instead of 32
    if (rc_wide_mode() == 1 || (rc_wide_mode() < 0 && !g_rc_dbg && !a.dbg_mode && 4 * (int64_t)ntiles <= (int64_t)ncu)) {
        const int items = 4 * ntiles;
        hipLaunchKernelGGL(rc_post_attn_wide_kernel, dim3((unsigned)(items < 2 * ncu ? items : 2 * ncu)), dim3(256), RC_LDS, (hipStream_t)stream, a);
        LAUNCH_CHECK();
        return SCP_OK;
    }
    hipLaunchKernelGGL(rc_post_attn_kernel, dim3((unsigned)(ntiles < ncu ? ntiles : ncu)), dim3(256), RC_LDS, (hipStream_t)stream, a);
    LAUNCH_CHECK();
    return SCP_OK;
}

// size in bytes of the weight buffer of scp_swin_post_attn (six tiled planes, see RcPostArgs.W)
extern "C" SCP_API int64_t scp_swin_post_attn_weight_bytes(void) { return 2 * (int64_t)RC_W_PLANE; }

// out = edge_mlp2(cat(pos3, edge_mlp1(cat(pos1, pos2, pos3)))) for M points (see rc_edge_mlp_kernel): pos1 / pos2 / pos3 fp32 [M][64 | 128 | 256]
// with row strides ld1 / ld2 / ld3; W: scp_swin_post_attn_weight_bytes() bytes, the eight tiled matrices described there; bias [1408];
// out fp32 [M][ldo] (128 columns written).
extern "C" SCP_API int scp_geo_edge_mlps(const float *pos1, int64_t ld1, const float *pos2, int64_t ld2, const float *pos3, int64_t ld3, const void *W,
                                         const float *bias, float *out, int64_t ldo, int32_t M, void *stream) {
    if (!pos1 || !pos2 || !pos3 || !W || !bias || !out || M <= 0 || ld1 < 64 || ld2 < 128 || ld3 < 256 || ((ld1 | ld2 | ld3) & 3) || ldo < 128 || (ldo & 3) ||
        (((uintptr_t)pos1 | (uintptr_t)pos2 | (uintptr_t)pos3 | (uintptr_t)out | (uintptr_t)W) & 15) || (int64_t)RC_ROWS * ldo * 4 > 0x7fffffffLL)
        return SCP_EINVAL;
    static bool configured = false;
    if (!configured) {
        HIP_TRY(hipFuncSetAttribute((const void *)rc_edge_mlp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS));
        configured = true;
    }
    RcEdgeArgs a;
    a.p1 = pos1; a.p2 = pos2; a.p3 = pos3; a.ld1 = ld1; a.ld2 = ld2; a.ld3 = ld3; a.W = W; a.bias = bias; a.out = out; a.ldo = ldo; a.M = M;
    const int ntiles = (M + RC_ROWS - 1) / RC_ROWS, ncu = rc_num_cu();
    SCP_PROF(SCP_PROF_EDGE_MLP, stream, 2.0 * M * (448.0 * 256 + 2.0 * 256 * 256 + 512.0 * 256 + 256.0 * 256 + 256.0 * 128));
    hipLaunchKernelGGL(rc_edge_mlp_kernel, dim3((unsigned)(ntiles < ncu ? ntiles : ncu)), dim3(256), RC_LDS, (hipStream_t)stream, a);
    LAUNCH_CHECK();
    return SCP_OK;
}

// out[out_map ? out_map[m] : m][0 .. N) = W3 . leaky(W2 . leaky(W1 . x[in_map ? in_map[m] : m] + b1) + b2) + b3 for M rows (see rc_mlp3_kernel).
// x: fp32 [n_src][ldx] (256 channels), W: scp_swin_post_attn_weight_bytes() bytes with three tiled [256][256] matrices at m * 131072 bytes (lo planes
// RC_W_PLANE behind; layers 2, 3: columns in rc_perm16 order; unused rows / columns zero), bias [768], out: fp32 rows of ldo floats, N columns written
// (N % 4 == 0, N <= 256, 16-byte aligned rows); rows with out_map[m] < 0 are dropped.
extern "C" SCP_API int scp_mlp3_rows(const float *x, int64_t ldx, int64_t n_src, const int64_t *in_map, const void *W, const float *bias, float *out, int64_t ldo,
                                     const int64_t *out_map, int32_t M, int32_t N, void *stream) {
    if (!x || !W || !bias || !out || M <= 0 || n_src <= 0 || N <= 0 || N > 256 || (N & 3) || ldx < 256 || (ldx & 3) || ldo < N || (ldo & 3) ||
        (((uintptr_t)x | (uintptr_t)out | (uintptr_t)W) & 15))
        return SCP_EINVAL;
    static bool configured = false;
    if (!configured) {
        HIP_TRY(hipFuncSetAttribute((const void *)rc_mlp3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS));
        configured = true;
    }
    RcMlp3Args a;
    a.x = x; a.ldx = ldx; a.in_map = in_map; a.n_src = n_src; a.W = W; a.bias = bias; a.out = out; a.ldo = ldo; a.out_map = out_map; a.M = M; a.N = N;
    const int ntiles = (M + RC_ROWS - 1) / RC_ROWS, ncu = rc_num_cu();
    SCP_PROF(SCP_PROF_MLP3, stream, 2.0 * M * 3.0 * 256.0 * 256.0);
    hipLaunchKernelGGL(rc_mlp3_kernel, dim3((unsigned)(ntiles < ncu ? ntiles : ncu)), dim3(256), RC_LDS, (hipStream_t)stream, a);
    LAUNCH_CHECK();
    return SCP_OK;
}

// out[m] = LayerNorm_noaffine(cat(x[ia[m]], x[ib[m]])) . W'^T + wbeta: SwinPatchMerging for M merged rows (see rc_merge_kernel).
// x: fp32 [n_src][ldx] (256 channels; an index equal to n_src stands for a row of zeros); W: scp_swin_post_attn_weight_bytes() bytes, the
// tiled planes (scp_split_weight_bf16 + scp_tile_weight_bf16) of (W diag(gamma))[:, :256] at byte 0 and of (W diag(gamma))[:, 256:] at byte
// 131072, their lo planes at the same offsets behind the first half of the buffer; wbeta = W beta [256]; out fp32 [M][ldo].
extern "C" SCP_API int scp_swin_merge(const float *x, int64_t ldx, int64_t n_src, const int64_t *ia, const int64_t *ib, const void *W, const float *wbeta,
                                      float eps, float *out, int64_t ldo, int32_t M, void *stream) {
    if (!x || !ia || !ib || !W || !out || M <= 0 || n_src <= 0 || ldx < 256 || (ldx & 3) || ldo < 256 || (ldo & 3) ||
        (((uintptr_t)x | (uintptr_t)out | (uintptr_t)W) & 15) || (int64_t)RC_ROWS * ldo * 4 > 0x7fffffffLL)
        return SCP_EINVAL;
    static bool configured = false;
    if (!configured) {
        HIP_TRY(hipFuncSetAttribute((const void *)rc_merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS));
        configured = true;
    }
    RcMergeArgs a;
    a.x = x; a.ldx = ldx; a.n_src = n_src; a.ia = ia; a.ib = ib; a.W = W; a.wbeta = wbeta; a.out = out; a.ldo = ldo; a.M = M; a.eps = eps;
    const int ntiles = (M + RC_ROWS - 1) / RC_ROWS, ncu = rc_num_cu();
    SCP_PROF(SCP_PROF_MERGE, stream, 2.0 * M * 512.0 * 256.0);
    hipLaunchKernelGGL(rc_merge_kernel, dim3((unsigned)(ntiles < ncu ? ntiles : ncu)), dim3(256), RC_LDS, (hipStream_t)stream, a);
    LAUNCH_CHECK();
    return SCP_OK;
}
