// 1-D Swin window attention, flash-style, float32 MFMA (gfx950 / CDNA4, wave64).
//
// Replaces models/swin_transformer.py:443-501 (Attention.forward) together with the roll / window partition /
// -100 shift mask / relative-position-bias plumbing of SwinLayer (:603-652, :684-697):
//   scores[i][j] = q_i.k_j / 8 + table[i - j + 511][head] (+ -100 where the last shifted window mixes its two halves)
//   out_i        = softmax_j(scores) . v_j          window = 512 tokens, 4 heads x 64
// q/k/v come already projected, [B][Lp][>=256] with row strides ldq / ldkv (so they may be column slices of one fused
// QKV GEMM output); Lp % 512 == 0 (rows beyond the sequence hold the projection biases,
// exactly what the reference's zero padding AFTER LayerNorm produces).  The cyclic shift is index arithmetic.
//
// Work decomposition: one workgroup (4 waves) = 128 queries of one (window, head); a wave owns 32 queries.
// Both products run on v_mfma_f32_32x32x2_f32 with the QUERY on the MFMA column (= lane & 31):
//   S^T[key][q] = sum_d K[key][d] Q[q][d]      A = K tile from LDS, B = Q fragment held in 32 registers
//   O^T[d][q]   = sum_key V[key][d] P[q][key]  A = V tile from LDS, B = the S^T accumulator registers themselves
// so the softmax statistics and the O rescale are lane-local and P never leaves the register file.  The k index of
// each product is permuted consistently on both operands (sums are order-free): lane half h covers head dims
// 32h..32h+31 in QK^T and the keys its own accumulator rows hold in PV.
#include <stdlib.h>
#include "scp_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WIN 512
#define HD 64
#define NH 4
#define QT 128   // queries per workgroup
#define KT 64    // keys per staged tile
#define LDK 68   // K tile row stride (floats): conflict-free ds_read_b128, 16-B aligned rows

typedef __bf16 bf16x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_split4(__bf16 *hi, __bf16 *lo, float a, float b, float c, float d) {
    float f[4] = {a, b, c, d};
    // materialise the rounded fp32 values: without this the compiler may contract (o * inv) - hi into one FMA and the planes
    // would no longer be the split of the fp32 output
#pragma unroll
    for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(f[u]));
    bf16x4s vh, vl;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const __bf16 hh = (__bf16)f[u];
        vh[u] = hh;
        vl[u] = (__bf16)(f[u] - (float)hh);
    }
    *(bf16x4s *)hi = vh;
    *(bf16x4s *)lo = vl;
}

__global__ __launch_bounds__(256, 2) void swin_attn_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                          const float *__restrict__ v, const float *__restrict__ table,
                                                          int Lp, int shift, int ldq, int ldkv, float *__restrict__ out,
                                                          const int *__restrict__ wtab /* per 512-window: (sequence base row, sequence Lp) or NULL */,
                                                          __bf16 *__restrict__ ohi, __bf16 *__restrict__ olo, int64_t ldo) {
    __shared__ __attribute__((aligned(16))) float Ks[KT * LDK];
    __shared__ __attribute__((aligned(16))) float Vs[KT * HD];
    __shared__ float tab[2 * WIN - 1];

    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, col = lane & 31, h = lane >> 5;
    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs; the four query tiles of one (window, head) share its K/V,
    // so give every XCD a contiguous run of the grid (grid size is a multiple of 16) - K/V then come out of that XCD's L2.
    int bid = blockIdx.x;
    bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);
    const int qtile = bid & 3; bid >>= 2;
    const int head = bid & 3; bid >>= 2;
    // dense mode: B sequences of Lp rows each; packed mode: sequences of different (x512) lengths back to back, described per window
    size_t seq_row;
    int wnd, nW;
    if (wtab) { seq_row = (size_t)wtab[2 * bid]; Lp = wtab[2 * bid + 1]; nW = Lp / WIN; wnd = (int)(((size_t)bid * WIN - seq_row) / WIN); }
    else { nW = Lp / WIN; wnd = bid % nW; seq_row = (size_t)(bid / nW) * Lp; }
    const size_t base = seq_row * (NH * HD) + head * HD;        // output (dense rows of 256)
    const size_t qbase = seq_row * ldq + head * HD, kbase = seq_row * ldkv + head * HD;
    const bool masked = (shift > 0) && (wnd == nW - 1);

    for (int i = tid; i < 2 * WIN - 1; i += 256) tab[i] = table[i * NH + head];

    // Q fragment: query qi (position in window), head dims 32h .. 32h+31, pre-scaled by 1/8 (exact)
    const int qi = qtile * QT + w * 32 + col;
    const int qtok = (wnd * WIN + qi + shift) % Lp;
    float qf[32];
    {
        const float4 *src = (const float4 *)(q + qbase + (size_t)qtok * ldq + 32 * h);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 t = src[g];
            qf[4 * g] = t.x * 0.125f; qf[4 * g + 1] = t.y * 0.125f; qf[4 * g + 2] = t.z * 0.125f; qf[4 * g + 3] = t.w * 0.125f;
        }
    }
    f32x16 o0, o1;   // O^T rows d = 0..31 and 32..63 (row = (r&3) + 8*(r>>2) + 4h), column = query
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    const int qreg = qi >> 8;  // half of the window the query sits in (mask region)

    for (int kt = 0; kt < WIN / KT; ++kt) {
        __syncthreads();  // previous tile consumed (also orders the `tab` fill before first use)
        // stage K and V rows of keys kt*64 .. +63 : 64 rows x 16 float4 each
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int e = tid + it * 256;
            const int r = e >> 4, c4 = e & 15;
            const int ktok = (wnd * WIN + kt * KT + r + shift) % Lp;
            const size_t g = kbase + (size_t)ktok * ldkv + 4 * c4;
            const float4 kv = *(const float4 *)(k + g);
            const float4 vv = *(const float4 *)(v + g);
            *(float4 *)(Ks + r * LDK + 4 * c4) = kv;
            *(float4 *)(Vs + r * HD + 4 * c4) = vv;
        }
        __syncthreads();
        const float madd = (masked && ((kt * KT) >> 8) != qreg) ? -100.f : 0.f;

#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            // ---- S^T = K . Q^T for 32 keys -----------------------------------------------------------------
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const float *krow = Ks + (sub * 32 + col) * LDK + 32 * h;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float4 kk = *(const float4 *)(krow + 4 * g);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.x, qf[4 * g], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.y, qf[4 * g + 1], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.z, qf[4 * g + 2], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.w, qf[4 * g + 3], s, 0, 0, 0);
            }
            // ---- bias, mask, online softmax (lane-local: this lane's 16 keys + partner half's 16) -----------
            const int j0 = kt * KT + sub * 32 + 4 * h;  // key position of accumulator row 0 of this lane
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + (r & 3) + 8 * (r >> 2);
                s[r] = s[r] + tab[qi - j + (WIN - 1)] + madd;
                mx = fmaxf(mx, s[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __expf(m_run - m_new);  // exp(-inf) = 0 on the first tile
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = __expf(s[r] - m_new); ps += s[r]; }
            ps += __shfl_xor(ps, 32);
            l_run = l_run * alpha + ps;
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            // ---- O^T += V^T . P^T : step r pairs key (r&3)+8(r>>2)+4h of both operands -----------------------
            const float *vbase = Vs + (sub * 32 + 4 * h) * HD + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float *vr = vbase + ((r & 3) + 8 * (r >> 2)) * HD;
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[0], s[r], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[32], s[r], o1, 0, 0, 0);
            }
        }
    }
    // ---- normalise and store: lane = query, accumulator rows = head dims -----------------------------------
    const float inv = 1.0f / l_run;
    if (ohi) {   // hi/lo bf16 planes: the operand format of the projection GEMM (scp_linear_split)
        const size_t o = (seq_row + (size_t)qtok) * (size_t)ldo + head * HD;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 8 * g + 4 * h;
            store_split4(ohi + o + d, olo + o + d, o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            store_split4(ohi + o + 32 + d, olo + o + 32 + d, o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
        }
        return;
    }
    float *dst = out + base + (size_t)qtok * (NH * HD);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int d = 8 * g + 4 * h;
        *(float4 *)(dst + d) = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
        *(float4 *)(dst + 32 + d) = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
    }
}


// ================================================================================================================
// bf16x3 variant: the same algorithm on v_mfma_f32_32x32x16_bf16.  K, V and Q are split into bf16 (hi, lo) pairs while they
// are staged (x = hi + lo to 16 significant bits), the softmax probabilities are split in registers, and every product is
// a_hi.b_hi + a_hi.b_lo + a_lo.b_hi with fp32 accumulation: 24 bf16 MFMAs (768 cycles) per 32-key tile instead of 64 fp32
// MFMAs (4096 cycles).  End effect on the EHEM logits: see tests/test_gpu_model.py (well inside the 1e-3 tolerance).
//   S^T = K . Q^T : A = K tile [key][d] (8 consecutive d per lane), B = Q fragment in registers
//   O^T = V^T . P^T: A = V^T tile [d][key'] staged TRANSPOSED with the keys of each 16-group permuted (swap bits 2 and 3) so that
//                    the 8 keys a lane needs are contiguous AND are exactly the 8 keys its S^T accumulator registers 8c..8c+7
//                    hold (rows (r&3) + 8(r>>2) + 4h of the 32x32 tile); B = those registers, split to bf16 - P never moves.
// Measured and dropped (round 2): the scores of both 32-key halves as two interleaved accumulation chains with one softmax update per
// 64 keys.  A lone dependent chain issues one v_mfma_f32_32x32x16 per 45 - 52 cycles instead of 32 (tools/src/mb_mfma_chain.cpp), so
// the interleave helps a wave that is alone in its matrix phase - but the second score tile costs 16 VGPRs (168 -> 188) and with
// them the third wave per SIMD: 156 us against 146 us per 128 windows.  The register-neutral form of the same idea - probabilities of
// half 0 split to planes first (freeing the score registers), then the score products of half 1 issued between the output products of
// half 0, bit-identical - needs both halves' K and V fragments live: 212 VGPRs (151 us at two waves per SIMD), 180 us when forced
// into 168 with spills.
// Cycle stamps of a diagnostic build (round 2; per 64-key tile and wave, the instrumented build holds 172 registers = two waves per SIMD):
// barriers + staging 2 470, load issue 600, score products 1 150 (24 products of one chain: 48 cycles each), softmax 2 570, output
// products with the split of P 1 480 - 8 270 cycles where the matrix pipe needs 1 536; a wave's phases are strictly serial and two or
// three waves per SIMD overlap them only partly (1.85 of 2 wave slots occupied on average, HW_ID + s_memtime).  Tried on that basis,
// bit-identical, A/B on one box against this kernel (168 registers, three waves per SIMD): two sets of staged tiles with ONE barrier per
// tile (78 KB of LDS, two workgroups per CU), plus the scores of both halves as two alternating chains and the two output halves
// alternating - staging 1 870, scores 1 180 per tile, but 1 335 against 1 230 us per 1 152 windows: the third wave is worth more.
// Also measured and dropped: K and V handed over as bf16 hi/lo planes written by the q|k|v projection's epilogue (q columns fp32, k|v
// columns planes; bit-identical, staging becomes a copy and a 4 x 4 transposition of 16-bit elements).  A probe with free conversions
// promised 7 - 13 %; the real thing gives 1326 against 1408 us (no shift) / 1325 against 1351 us (shifted) per 1152 windows - the 8-byte
// plane loads and the 16-bit shuffles cost what the conversions did - while the projection pays 2 %: nothing per frame.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define LDB 72   // bf16 elements per LDS row (64 + 8): 144-byte rows -> conflict-free 16-byte fragment reads

// value of lane ^ 32 (the other half-lane of a query) by v_permlane32_swap - the half exchange of gfx950 - instead of ds_bpermute
__device__ __forceinline__ float attn_xchg32(float x, int h) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(h ? r[0] : r[1]);
}

// ---- the arithmetic of ONE 32-key tile, shared by the two bf16x3 kernels below (rows-fed, plane-fed): identical bits by construction ----
// kfrag(c, hi, lo) / vfrag(c, v0h, v0l, v1h, v1l) fetch the kernel's own LDS fragments.  The S accumulators START from the bias (+ the
// shift mask): the 16 bias adds of a tile become the products' C operand.
// FAST (round 4; the first sweep of every workgroup): scores against the FIXED reference 0 - P = exp2(S), l += sum P per half-lane - no
// maximum, no subtraction, no rescale of O, no half-lane exchange per tile: 105 instead of 165 vector instructions per tile and wave.
// fp32 carries 2^+-126, so this is exact arithmetic for any row whose largest score (log2 domain) lies within +-100; the caller
// checks the row sums behind the sweep (2^-100 < l < 2^100, O finite) and repeats the workgroup's sweep in the standard online-softmax form
// (FAST = false) otherwise.  A/B on one box, 1 152 windows: 1 094 -> 990 us.
template <bool FAST, class KF, class VF>
__device__ __forceinline__ void attn_tile(KF kfrag, VF vfrag, const bf16x8 (&qh)[4], const bf16x8 (&ql)[4], const float *tab, int b0 /* qi - j0 + WIN - 1 */,
                                          float madd, int h, f32x16 &o0, f32x16 &o1, float &m_run, float &l_run) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = tab[b0 - ((r & 3) + 8 * (r >> 2))];
    if (madd != 0.f) {                                          // uniform per workgroup and tile
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] += madd;
    }
    // the K fragments are requested two k-steps at a time and pinned (hipcc otherwise sinks every read down to its first use: ds_read ->
    // s_waitcnt lgkmcnt(0) -> one product, eight times per tile); all four at once costs a wave per SIMD (132 registers) and is slower
    bf16x8 kah[4], kal[4];
#pragma unroll
    for (int c0 = 0; c0 < 4; c0 += 2) {
#pragma unroll
        for (int c = c0; c < c0 + 2; ++c) kfrag(c, kah[c], kal[c]);
#pragma unroll
        for (int c = c0; c < c0 + 2; ++c) asm volatile("" : "+v"(kah[c]), "+v"(kal[c]));
#pragma unroll
        for (int c = c0; c < c0 + 2; ++c) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kal[c], qh[c], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kah[c], ql[c], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kah[c], qh[c], s, 0, 0, 0);
        }
    }
    if (FAST) {
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r]); ps += s[r]; }
        l_run += ps;                                            // this half-lane's share; the halves meet once, behind the sweep
    } else {
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, attn_xchg32(mx, h));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - m_new); ps += s[r]; }
        ps += attn_xchg32(ps, h);
        l_run = l_run * alpha + ps;
        m_run = m_new;
        if (__any(alpha != 1.f)) {   // the running maximum moved for some query of this wavefront (rare after the first tiles)
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        }
    }
    // O^T += V^T . P^T; chunk c consumes accumulator registers 8c..8c+7 = keys 16c + {0..3, 8..11} + 4h
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        bf16x8 ph, pl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = s[8 * c + j];
            const __bf16 hh = (__bf16)x;
            ph[j] = hh;
            pl[j] = (__bf16)(x - (float)hh);
        }
        bf16x8 v0h, v0l, v1h, v1l;
        vfrag(c, v0h, v0l, v1h, v1l);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0l, ph, o0, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0h, pl, o0, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0h, ph, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1l, ph, o1, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1h, pl, o1, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1h, ph, o1, 0, 0, 0);
    }
}

// behind a FAST sweep: the query's row sum (both half-lanes) and whether the fixed reference held for it
__device__ __forceinline__ bool attn_fast_ok(float &l_run, const f32x16 &o0, const f32x16 &o1, int h) {
    l_run += attn_xchg32(l_run, h);
    float am = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) am = fmaxf(am, fmaxf(fabsf(o0[r]), fabsf(o1[r])));
    return l_run > 0x1p-100f && l_run < 0x1p+100f && am < 0x1p+120f;       // false for NaN as well
}

struct AttnFast { static constexpr bool value = true; };
struct AttnStd { static constexpr bool value = false; };

// Q fragment (B operand of S^T): head dims 16c + 8h + j of the lane's query, pre-scaled by log2(e) / 8, split hi / lo
__device__ __forceinline__ void attn_load_q(const float *src, int h, bf16x8 (&qh)[4], bf16x8 (&ql)[4]) {
    constexpr float LOG2E = 1.4426950408889634f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float4 a = *(const float4 *)(src + 16 * c + 8 * h), b = *(const float4 *)(src + 16 * c + 8 * h + 4);
        const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = f[j] * (0.125f * LOG2E);
            const __bf16 hh = (__bf16)x;
            qh[c][j] = hh;
            ql[c][j] = (__bf16)(x - (float)hh);
        }
    }
}

__device__ __forceinline__ void attn_store_o(const f32x16 &o0, const f32x16 &o1, float inv, int h, float *dst /* fp32 row + head offset or null */,
                                             __bf16 *ohi, __bf16 *olo /* plane rows + head offset */) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int d = 8 * g + 4 * h;
        if (ohi) {   // hi/lo bf16 planes: the operand format of the projection that follows
            store_split4(ohi + d, olo + d, o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            store_split4(ohi + 32 + d, olo + 32 + d, o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
        } else {
            *(float4 *)(dst + d) = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            *(float4 *)(dst + 32 + d) = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
        }
    }
}

template <bool FASTFIRST>
__global__ __launch_bounds__(256, 2) void swin_attn_bf16x3_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                                 const float *__restrict__ v, const float *__restrict__ table,
                                                                 int Lp, int shift, int ldq, int ldkv, float *__restrict__ out,
                                                                 const int *__restrict__ wtab, __bf16 *__restrict__ ohi,
                                                                 __bf16 *__restrict__ olo, int64_t ldo) {
    __shared__ __attribute__((aligned(16))) __bf16 Kh[KT * LDB], Kl[KT * LDB];      // [key][d]
    __shared__ __attribute__((aligned(16))) __bf16 Vh[HD * LDB], Vl[HD * LDB];      // [d][permuted key]
    __shared__ float tab[2 * WIN - 1];

    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, col = lane & 31, h = lane >> 5;
    int bid = blockIdx.x;
    bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);
    const int qtile = bid & 3; bid >>= 2;
    const int head = bid & 3; bid >>= 2;
    size_t seq_row;
    int wnd, nW;
    if (wtab) { seq_row = (size_t)wtab[2 * bid]; Lp = wtab[2 * bid + 1]; nW = Lp / WIN; wnd = (int)(((size_t)bid * WIN - seq_row) / WIN); }
    else { nW = Lp / WIN; wnd = bid % nW; seq_row = (size_t)(bid / nW) * Lp; }
    const size_t base = seq_row * (NH * HD) + head * HD;
    const size_t qbase = seq_row * ldq + head * HD, kbase = seq_row * ldkv + head * HD;
    const bool masked = (shift > 0) && (wnd == nW - 1);

    // scores are kept in the log2 domain (q pre-scaled by log2(e) / 8, bias table and mask by log2(e)): exp() is then one v_exp_f32
    constexpr float LOG2E = 1.4426950408889634f;
    for (int i = tid; i < 2 * WIN - 1; i += 256) tab[i] = table[i * NH + head] * LOG2E;

    const int qi = qtile * QT + w * 32 + col;
    // cyclic shift: window base + offset < Lp and shift < WIN <= Lp, so one conditional subtract replaces the modulo (an integer
    // division by a run-time Lp costs ~10 instructions per staged row)
    auto wrap = [&](int t) { return t >= Lp ? t - Lp : t; };
    const int qtok = wrap(wnd * WIN + qi + shift);
    bf16x8 qh[4], ql[4];
    attn_load_q(q + qbase + (size_t)qtok * ldq, h, qh, ql);
    f32x16 o0, o1;
    float m_run, l_run;
    const int qreg = qi >> 8;

    // staging: 64 keys per tile, K as [key][d] planes, V transposed as [d][pi(key)] planes (pi swaps bits 2 and 3 of the key
    // index).  The global loads of tile kt + 1 are issued BEFORE the MFMAs of tile kt and converted / written to LDS after them,
    // so their latency hides behind the compute of the same workgroup.
    const int c4 = tid & 15, kg = tid >> 4;
    float4 kreg[4], vreg[4];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {   // K: rows kg + 16 it, head dims 4 c4 .. 4 c4 + 3
            const int ktok = wrap(wnd * WIN + kt * KT + kg + 16 * it + shift);
            kreg[it] = *(const float4 *)(k + kbase + (size_t)ktok * ldkv + 4 * c4);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {   // V: a 4-key x 4-dim block per thread (keys 4 kg .. 4 kg + 3: pi keeps them adjacent)
            const int ktok = wrap(wnd * WIN + kt * KT + 4 * kg + kk + shift);
            vreg[kk] = *(const float4 *)(v + kbase + (size_t)ktok * ldkv + 4 * c4);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = kg + 16 * it;
            const float kf[4] = {kreg[it].x, kreg[it].y, kreg[it].z, kreg[it].w};
            bf16x4 khi, klo;
#pragma unroll
            for (int u = 0; u < 4; ++u) { const __bf16 hh = (__bf16)kf[u]; khi[u] = hh; klo[u] = (__bf16)(kf[u] - (float)hh); }
            *(bf16x4 *)(Kh + r * LDB + 4 * c4) = khi;
            *(bf16x4 *)(Kl + r * LDB + 4 * c4) = klo;
        }
        // V transposed in registers: 8-byte stores of 4 consecutive permuted keys per head dim instead of sixteen 2-byte ones
        const float vf[4][4] = {{vreg[0].x, vreg[0].y, vreg[0].z, vreg[0].w}, {vreg[1].x, vreg[1].y, vreg[1].z, vreg[1].w},
                                {vreg[2].x, vreg[2].y, vreg[2].z, vreg[2].w}, {vreg[3].x, vreg[3].y, vreg[3].z, vreg[3].w}};
        const int pr = 4 * ((kg & ~3) | ((kg & 1) << 1) | ((kg >> 1) & 1));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bf16x4 vhi, vlo;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { const __bf16 hh = (__bf16)vf[kk][u]; vhi[kk] = hh; vlo[kk] = (__bf16)(vf[kk][u] - (float)hh); }
            *(bf16x4 *)(Vh + (4 * c4 + u) * LDB + pr) = vhi;
            *(bf16x4 *)(Vl + (4 * c4 + u) * LDB + pr) = vlo;
        }
    };

    auto sweep = [&](auto F) {
        constexpr bool FAST = decltype(F)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
        m_run = -INFINITY; l_run = 0.f;
        load_tile(0);
        for (int kt = 0; kt < WIN / KT; ++kt) {
            __syncthreads();            // everybody is done reading the previous tile
            store_tile();
            __syncthreads();
            if (kt + 1 < WIN / KT) load_tile(kt + 1);   // in flight during this tile's MFMAs
            const float madd = (masked && ((kt * KT) >> 8) != qreg) ? -100.f * LOG2E : 0.f;   // uniform per workgroup and tile
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int ko = (sub * 32 + col) * LDB + 8 * h;
                auto kfrag = [&](int c, bf16x8 &ah, bf16x8 &al) { ah = *(const bf16x8 *)(Kh + ko + 16 * c); al = *(const bf16x8 *)(Kl + ko + 16 * c); };
                auto vfrag = [&](int c, bf16x8 &v0h, bf16x8 &v0l, bf16x8 &v1h, bf16x8 &v1l) {
                    const int vo = col * LDB + sub * 32 + 16 * c + 8 * h;
                    v0h = *(const bf16x8 *)(Vh + vo); v0l = *(const bf16x8 *)(Vl + vo);
                    v1h = *(const bf16x8 *)(Vh + vo + 32 * LDB); v1l = *(const bf16x8 *)(Vl + vo + 32 * LDB);
                };
                attn_tile<FAST>(kfrag, vfrag, qh, ql, tab, qi - (kt * KT + sub * 32 + 4 * h) + (WIN - 1), madd, h, o0, o1, m_run, l_run);
            }
        }
    };
    if (FASTFIRST) {
        sweep(AttnFast{});
        if (__syncthreads_or(!attn_fast_ok(l_run, o0, o1, h))) sweep(AttnStd{});
    } else sweep(AttnStd{});
    const float inv = 1.0f / l_run;
    attn_store_o(o0, o1, inv, h, out ? out + base + (size_t)qtok * (NH * HD) : nullptr,
                 ohi ? ohi + (seq_row + (size_t)qtok) * (size_t)ldo + head * HD : nullptr, ohi ? olo + (seq_row + (size_t)qtok) * (size_t)ldo + head * HD : nullptr);
}

// ================================================================================================================
// The same bf16x3 attention with K and V arriving as bf16 hi / lo PLANES in the layout of its own LDS tiles, staged by LDS-DMA: no
// conversion, no staging registers, no ds_write in this kernel.  32-key tiles, two stages (4 planes x 4 KiB each), ONE barrier per tile:
//   K planes  [token][4 heads][64 d] bf16, the eight 16-byte chunks of a head's 128 bytes stored at position chunk ^ (token & 7);
//   V^T planes [32-token block][head][32 super-rows of 128 B]: head dim d, key chunk c (8 keys, 16 B) at super-row d >> 1, slot
//             ((d & 1) * 4 + c) ^ ((d >> 1) & 7); the 32 keys of a block in the order p = 16 c' + 8 h + j  <->  key 16 c' + 4 h + (j & 3) +
//             8 (j >> 2) (bits 2 and 3 of the key index swapped: the 8 keys of a lane's P fragment are contiguous).
// Both are linear images of the LDS tiles, so a DMA instruction copies 1 KiB as it lies.  Same products in the same order as
// swin_attn_bf16x3_kernel: identical bits.  (scp_swin_kv_planes writes the planes from fp32 k / v.)
// Measured and dropped (round 4, tools/pmc_lds.sh): with these swizzles two lanes of every 16-lane ds_read_b128 group ({0-3, 12-15, 20-27}, ...:
// MI355X_MICROARCH.md, LDS) meet in each bank - SQ_LDS_BANK_CONFLICT is 40 % of the kernel's SQ_LDS_IDX_ACTIVE (the row-chain and split-GEMM kernels:
// 0 - 5 %).  With (token >> 1) & 7 / (super-row >> 1) & 7 instead the counter reads 0 and the LDS-active cycles fall by 40 % - and the launch takes
// the same 1 030 - 1 060 us per 1 152 windows (the LDS is not what it waits for), while the projection's V^T bounce writes become two-way conflicted.
// Round 4: the stages are DYNAMIC shared memory.  For a static array hipcc (ROCm 7.2) knows that the LDS-DMA writes it and puts an
// s_waitcnt vmcnt(0) in front of the first ds_read of that array behind a DMA instruction - here the K reads of tile t, right behind the
// request for tile t + 1: the "prefetch" was drained every tile.  The barrier macro carries the wait that is needed.
// Measured and dropped (round 4, tools/src/mb_gap.cpp, mb_phase.cpp, profiles/r4k_attention_probes.md): a form with TWO query blocks per wave whose
// instruction stream is written gap by gap (every product followed by four to six vector instructions of the other block's softmax, every LDS read
// requested a phase ahead, K fragments shared by the blocks, 256-query workgroups: half the LDS-DMA traffic), bit-identical: 1 005 - 1 018 against
// 1 032 - 1 063 us per 1 152 windows - 3 %, for 250 lines.  Also: 8- / 16-wave workgroups (a half / a quarter of the DMA traffic): 1 218 / 1 080 us.
typedef __attribute__((address_space(3))) void *attn_lds_ptr_t;
typedef const __attribute__((address_space(1))) void *attn_glb_ptr_t;
#define PKT 32          // keys per tile
#define PST 16384       // bytes per stage: K hi | K lo | V hi | V lo, 4 KiB each
#define PNS 2           // stages.  3 (52 KiB of LDS: three workgroups per CU instead of four) measured twice - 1 282 against 1 227 us per 1 152 windows
                        // with static stages (round 3), 1 118 - 1 135 against 1 127 - 1 134 with dynamic ones - and not kept

template <bool FASTFIRST>      // A/B and test bracket (scp_debug.h: scp_set_attention_variant): false = the standard online softmax only
__global__ __launch_bounds__(256, 3) void swin_attn_planes_kernel(const float *__restrict__ q, const __bf16 *__restrict__ khi, const __bf16 *__restrict__ klo,
                                                                 const __bf16 *__restrict__ vthi, const __bf16 *__restrict__ vtlo,
                                                                 const float *__restrict__ table, int shift, int ldq, float *__restrict__ out,
                                                                 const int *__restrict__ wtab, __bf16 *__restrict__ ohi, __bf16 *__restrict__ olo, int64_t ldo,
                                                                 const float *__restrict__ valid) {
    extern __shared__ __attribute__((aligned(1024))) char stg[];            // PNS * PST bytes
    __shared__ float tab[2 * WIN - 1];
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);
    const int qtile = bid & 3; bid >>= 2;
    const int head = bid & 3; bid >>= 2;
    const size_t seq_row = (size_t)wtab[2 * bid];
    const int Lp = wtab[2 * bid + 1], nW = Lp / WIN;
    const int wnd = (int)(((size_t)bid * WIN - seq_row) / WIN);
    const size_t base = seq_row * (NH * HD) + head * HD;
    const size_t qbase = seq_row * ldq + head * HD;
    const bool masked = (shift > 0) && (wnd == nW - 1);
    auto wrap = [&](int t) { return t >= Lp ? t - Lp : t; };
    // A query tile of nothing but window padding (the real rows of a sequence are a prefix of its 512-aligned run, a tile is 128 aligned
    // rows - also under the cyclic shift of 256 -, so it is all padding iff its first row is): nobody reads its output (scp_swin_post_attn
    // skips the same tiles through its tile list), so the workgroup leaves.  4.7 % of the query tiles of a level-16 multi-level frame.
    if (valid && valid[seq_row + wrap(wnd * WIN + qtile * QT + shift)] == 0.f) return;
    constexpr float LOG2E = 1.4426950408889634f;
    for (int i = tid; i < 2 * WIN - 1; i += 256) tab[i] = table[i * NH + head] * LOG2E;

    const int qi = qtile * QT + w * 32 + col;
    const int qtok = wrap(wnd * WIN + qi + shift);
    bf16x8 qh[4], ql[4];
    attn_load_q(q + qbase + (size_t)qtok * ldq, h, qh, ql);
    f32x16 o0, o1;
    float m_run, l_run;
    const int qreg = qi >> 8;

    // wave w copies plane w of a tile (K hi, K lo, V hi, V lo): four 1 KiB DMA instructions
    const __bf16 *kpl = (w & 1) ? klo : khi, *vpl = (w & 1) ? vtlo : vthi;
    auto issue = [&](int t, int st) {
        const int tok0 = wrap(wnd * WIN + t * PKT + shift);                 // a multiple of 32: the tile is 32 consecutive tokens
        char *dst = stg + st * PST + w * 4096;
        if (w < 2) {
            const char *src = (const char *)(kpl + (seq_row + tok0 + (lane >> 3)) * (size_t)(NH * HD) + head * HD) + (lane & 7) * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((attn_glb_ptr_t)(src + (size_t)i * 8 * NH * HD * 2), (attn_lds_ptr_t)(dst + i * 1024), 16, 0, 0);
        } else {
            const char *src = (const char *)(vpl + (((seq_row + tok0) >> 5) * NH + head) * (size_t)2048) + lane * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((attn_glb_ptr_t)(src + i * 1024), (attn_lds_ptr_t)(dst + i * 1024), 16, 0, 0);
        }
    };
    __syncthreads();                                                        // the bias table
    auto sweep = [&](auto F) {
        constexpr bool FAST = decltype(F)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
        m_run = -INFINITY; l_run = 0.f;
        issue(0, 0);
        if (PNS > 2) issue(1, 1);
        for (int t = 0; t < WIN / PKT; ++t) {
            // tile t has landed (with three stages the 4 pieces of tile t + 1 stay in flight); everybody is done with the stage refilled next
            if (PNS > 2 && t + 1 < WIN / PKT) SCP_BARRIER_DMA(4); else SCP_BARRIER_DMA(0);
            if (t + PNS - 1 < WIN / PKT) issue(t + PNS - 1, (t + PNS - 1) % PNS);
            const char *S = stg + (t % PNS) * PST;
            const float madd = (masked && ((t * PKT) >> 8) != qreg) ? -100.f * LOG2E : 0.f;
            auto kfrag = [&](int c, bf16x8 &ah, bf16x8 &al) {
                const int ko = col * 128 + (((2 * c + h) ^ (col & 7)) << 4);
                ah = *(const bf16x8 *)(S + ko); al = *(const bf16x8 *)(S + 4096 + ko);
            };
            auto vfrag = [&](int c, bf16x8 &v0h, bf16x8 &v0l, bf16x8 &v1h, bf16x8 &v1l) {
                // head dim d = col (o0) / col + 32 (o1), key chunk 2 c + h: super-row d >> 1, slot ((d & 1) * 4 + chunk) ^ ((d >> 1) & 7)
                const int R0 = col >> 1, R1 = R0 + 16, sl = (col & 1) * 4 + 2 * c + h;
                const int v0 = R0 * 128 + ((sl ^ (R0 & 7)) << 4), v1 = R1 * 128 + ((sl ^ (R1 & 7)) << 4);
                v0h = *(const bf16x8 *)(S + 8192 + v0); v0l = *(const bf16x8 *)(S + 12288 + v0);
                v1h = *(const bf16x8 *)(S + 8192 + v1); v1l = *(const bf16x8 *)(S + 12288 + v1);
            };
            attn_tile<FAST>(kfrag, vfrag, qh, ql, tab, qi - (t * PKT + 4 * h) + (WIN - 1), madd, h, o0, o1, m_run, l_run);
        }
    };
    if (FASTFIRST) {
        sweep(AttnFast{});
        if (__syncthreads_or(!attn_fast_ok(l_run, o0, o1, h))) sweep(AttnStd{});      // (the barrier also frees the stages for the second sweep)
    } else sweep(AttnStd{});
    const float inv = 1.0f / l_run;
    attn_store_o(o0, o1, inv, h, out ? out + base + (size_t)qtok * (NH * HD) : nullptr,
                 ohi ? ohi + (seq_row + (size_t)qtok) * (size_t)ldo + head * HD : nullptr, ohi ? olo + (seq_row + (size_t)qtok) * (size_t)ldo + head * HD : nullptr);
}

// fp32 k, v [rows][ldkv] (4 heads x 64) -> the four planes of swin_attn_planes_kernel (rows % 32 == 0: whole 32-token blocks).  One
// thread per (row, head, 8 head dims): K as one swizzled 16-byte chunk per plane, V^T as eight 2-byte elements per plane.
__global__ __launch_bounds__(256) void kv_planes_kernel(const float *__restrict__ k, const float *__restrict__ v, int64_t ldkv, int64_t rows,
                                                       __bf16 *__restrict__ khi, __bf16 *__restrict__ klo, __bf16 *__restrict__ vthi,
                                                       __bf16 *__restrict__ vtlo) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= rows * 32) return;
    const int64_t row = g >> 5;
    const int head = (int)(g >> 3) & 3, ch = (int)g & 7;
    const float *ks = k + row * ldkv + head * HD + 8 * ch, *vs = v + row * ldkv + head * HD + 8 * ch;
    const float4 a = *(const float4 *)ks, b = *(const float4 *)(ks + 4), c = *(const float4 *)vs, d = *(const float4 *)(vs + 4);
    const float kf[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}, vf[8] = {c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    bf16x8 kh, kl;
#pragma unroll
    for (int u = 0; u < 8; ++u) { const __bf16 hh = (__bf16)kf[u]; kh[u] = hh; kl[u] = (__bf16)(kf[u] - (float)hh); }
    const int64_t ko = row * (NH * HD) + head * HD + ((ch ^ (int)(row & 7)) << 3);
    *(bf16x8 *)(khi + ko) = kh;
    *(bf16x8 *)(klo + ko) = kl;
    const int kk = (int)(row & 31), p = (kk & ~12) | ((kk & 4) << 1) | ((kk & 8) >> 1);       // bits 2 and 3 swapped
    const int64_t vb = ((row >> 5) * NH + head) * (int64_t)2048;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int dd = 8 * ch + u, R = dd >> 1, sl = (dd & 1) * 4 + (p >> 3);
        const int64_t vo = vb + R * 64 + ((sl ^ (R & 7)) << 3) + (p & 7);
        const __bf16 hh = (__bf16)vf[u];
        vthi[vo] = hh;
        vtlo[vo] = (__bf16)(vf[u] - (float)hh);
    }
}

extern "C" SCP_API int scp_swin_kv_planes(const float *k, const float *v, int64_t ldkv, int64_t rows, void *khi, void *klo, void *vthi, void *vtlo,
                                          void *stream) {
    if (!k || !v || !khi || !klo || !vthi || !vtlo || rows <= 0 || (rows & 31) || ldkv < NH * HD || (ldkv & 3) ||
        (((uintptr_t)k | (uintptr_t)v | (uintptr_t)khi | (uintptr_t)klo | (uintptr_t)vthi | (uintptr_t)vtlo) & 15))
        return SCP_EINVAL;
    hipLaunchKernelGGL(kv_planes_kernel, dim3((unsigned)cdiv64(rows * 32, 256)), dim3(256), 0, (hipStream_t)stream, k, v, ldkv, rows, (__bf16 *)khi,
                       (__bf16 *)klo, (__bf16 *)vthi, (__bf16 *)vtlo);
    LAUNCH_CHECK();
    return SCP_OK;
}

// process-wide test / A-B bracket (scp_debug.h).  It changes the logits' last bits, so the host writes a non-default value into the stream's
// numeric profile (native.numeric_profile: ",attnv=0") and a decoder running the other variant refuses the stream.
static std::atomic<int> g_attn_variant{1};
extern "C" SCP_API int scp_set_attention_variant(int32_t v) { g_attn_variant.store(v ? 1 : 0, std::memory_order_relaxed); return SCP_OK; }
extern "C" SCP_API int scp_get_attention_variant(void) { return g_attn_variant.load(std::memory_order_relaxed); }

// packed window attention on those planes (q fp32 [rows][ldq]); out fp32 [rows][256] or (ohi, olo) planes [rows][ldo]
extern "C" SCP_API int scp_swin_attention_packed_planes(const float *q, const void *khi, const void *klo, const void *vthi, const void *vtlo,
                                                        const float *bias_table, const int32_t *wtab, int32_t total_windows, int32_t shift, int32_t ldq,
                                                        float *out, void *ohi, void *olo, int64_t ldo, const float *valid, void *stream) {
    if (!q || !khi || !klo || !vthi || !vtlo || !bias_table || !wtab || (!out && !ohi) || total_windows <= 0 || (shift != 0 && shift != WIN / 2) ||
        ldq < NH * HD || (ldq & 3) || (((uintptr_t)q | (uintptr_t)out | (uintptr_t)khi | (uintptr_t)klo | (uintptr_t)vthi | (uintptr_t)vtlo) & 15) ||
        (ohi && (!olo || ldo < NH * HD || (ldo & 3) || (((uintptr_t)ohi | (uintptr_t)olo) & 7))))
        return SCP_EINVAL;
    SCP_PROF(SCP_PROF_ATTENTION, stream, (double)total_windows * WIN * 2.0 * 2.0 * WIN * NH * HD);
#define ATTN_GO(V) hipLaunchKernelGGL(swin_attn_planes_kernel<V>, dim3(total_windows * NH * (WIN / QT)), dim3(256), PNS * PST, (hipStream_t)stream, q, (const __bf16 *)khi, \
                                      (const __bf16 *)klo, (const __bf16 *)vthi, (const __bf16 *)vtlo, bias_table, shift, ldq, out, wtab, (__bf16 *)ohi, (__bf16 *)olo, ldo, valid)
    if (g_attn_variant.load(std::memory_order_relaxed) == 0) ATTN_GO(false); else ATTN_GO(true);
#undef ATTN_GO
    LAUNCH_CHECK();
    return SCP_OK;
}

static int g_attn_mode = -1;   // 0 = fp32 MFMA, 1 = bf16x3 (default); SCP_ATTN=f32 selects the former
static inline bool attn_bf16x3() {
    const int c = scp_ctx_attention_mode();      // the calling thread's current scp_ctx decides; without one, the process default
    if (c >= 0) return c == 1;
    if (g_attn_mode < 0) { const char *e = getenv("SCP_ATTN"); g_attn_mode = (e && e[0] == 'f') ? 0 : 1; }
    return g_attn_mode == 1;
}
extern "C" SCP_API int scp_set_attention_mode(int32_t bf16x3) { g_attn_mode = bf16x3 ? 1 : 0; return SCP_OK; }

// packed ("varlen") form: total_windows 512-row windows, wtab[2*w] = first row of the sequence that owns window w, wtab[2*w+1] = its padded length
static int attn_packed(const float *q, const float *k, const float *v, const float *bias_table, const int32_t *wtab, int32_t total_windows,
                       int32_t shift, int32_t ldq, int32_t ldkv, float *out, __bf16 *ohi, __bf16 *olo, int64_t ldo, void *stream) {
    if (!q || !k || !v || !bias_table || (!out && !ohi) || !wtab || total_windows <= 0 || (shift != 0 && shift != WIN / 2) || ldq < NH * HD ||
        ldkv < NH * HD || (ldq & 3) || (ldkv & 3) || (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) ||
        (ohi && (!olo || ldo < NH * HD || (ldo & 3) || (((uintptr_t)ohi | (uintptr_t)olo) & 7))))
        return SCP_EINVAL;
    SCP_PROF(SCP_PROF_ATTENTION, stream, (double)total_windows * WIN * 2.0 * 2.0 * WIN * NH * HD);
    if (attn_bf16x3() && g_attn_variant.load(std::memory_order_relaxed) == 0)
        hipLaunchKernelGGL(swin_attn_bf16x3_kernel<false>, dim3(total_windows * NH * (WIN / QT)), dim3(256), 0, (hipStream_t)stream, q, k, v, bias_table,
                           0, shift, ldq, ldkv, out, wtab, ohi, olo, ldo);
    else if (attn_bf16x3())
        hipLaunchKernelGGL(swin_attn_bf16x3_kernel<true>, dim3(total_windows * NH * (WIN / QT)), dim3(256), 0, (hipStream_t)stream, q, k, v, bias_table,
                           0, shift, ldq, ldkv, out, wtab, ohi, olo, ldo);
    else
        hipLaunchKernelGGL(swin_attn_kernel, dim3(total_windows * NH * (WIN / QT)), dim3(256), 0, (hipStream_t)stream, q, k, v, bias_table, 0, shift,
                           ldq, ldkv, out, wtab, ohi, olo, ldo);
    LAUNCH_CHECK();
    return SCP_OK;
}

extern "C" int scp_swin_attention_packed(const float *q, const float *k, const float *v, const float *bias_table, const int32_t *wtab,
                                         int32_t total_windows, int32_t shift, int32_t ldq, int32_t ldkv, float *out, void *stream) {
    if (!out) return SCP_EINVAL;
    return attn_packed(q, k, v, bias_table, wtab, total_windows, shift, ldq, ldkv, out, nullptr, nullptr, 0, stream);
}

// the same, output written as bf16 hi/lo planes [rows][ldo] (operand format of scp_linear_split)
extern "C" int scp_swin_attention_packed_split(const float *q, const float *k, const float *v, const float *bias_table, const int32_t *wtab,
                                               int32_t total_windows, int32_t shift, int32_t ldq, int32_t ldkv, void *ohi, void *olo,
                                               int64_t ldo, void *stream) {
    if (!ohi || !olo) return SCP_EINVAL;
    return attn_packed(q, k, v, bias_table, wtab, total_windows, shift, ldq, ldkv, nullptr, (__bf16 *)ohi, (__bf16 *)olo, ldo, stream);
}

extern "C" int scp_swin_attention(const float *q, const float *k, const float *v, const float *bias_table, int32_t B, int32_t Lp,
                                  int32_t shift, int32_t ldq, int32_t ldkv, float *out, void *stream) {
    if (!q || !k || !v || !bias_table || !out || B <= 0 || Lp <= 0 || (Lp % WIN) != 0 || (shift != 0 && shift != WIN / 2) ||
        ldq < NH * HD || ldkv < NH * HD || (ldq & 3) || (ldkv & 3) || (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15))
        return SCP_EINVAL;
    const int nblk = B * (Lp / WIN) * NH * (WIN / QT);
    SCP_PROF(SCP_PROF_ATTENTION, stream, (double)B * Lp * 2.0 * 2.0 * WIN * NH * HD);
    if (attn_bf16x3() && g_attn_variant.load(std::memory_order_relaxed) == 0)
        hipLaunchKernelGGL(swin_attn_bf16x3_kernel<false>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, q, k, v, bias_table, Lp, shift, ldq, ldkv, out,
                           (const int *)nullptr, (__bf16 *)nullptr, (__bf16 *)nullptr, (int64_t)0);
    else if (attn_bf16x3())
        hipLaunchKernelGGL(swin_attn_bf16x3_kernel<true>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, q, k, v, bias_table, Lp, shift, ldq, ldkv, out,
                           (const int *)nullptr, (__bf16 *)nullptr, (__bf16 *)nullptr, (int64_t)0);
    else
        hipLaunchKernelGGL(swin_attn_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, q, k, v, bias_table, Lp, shift, ldq, ldkv, out,
                           (const int *)nullptr, (__bf16 *)nullptr, (__bf16 *)nullptr, (int64_t)0);
    LAUNCH_CHECK();
    return SCP_OK;
}
