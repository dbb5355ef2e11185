// OctAttention input stage in one kernel (models/oct_attention.py:48-66 of the reference): the three embedding lookups for the four
// ancestors of a node, the absolute-position Linear(3 -> 12), the concatenation to 600 channels, the sqrt(D) scale, the sinusoidal
// position table - for BOTH streams (the "unknown" stream sees occ_enc[255] in place of the node's own occupancy) - and, in the same
// pass, the f16x3 operand of the first dense layers (power-of-two row scale + two IEEE-half planes, the arithmetic of
// split_rows_f16_kernel).  It replaces ~14 torch launches per forward (embedding x3, cat x3, stack, mul, add, clone, index_put, the
// K = 3 GEMM, the standalone split pass: ~5 ms of a 41 ms L14 frame) by one pass that writes every byte once.
// Compiled with -ffp-contract=off: (x * sqrt(D)) + pe must round twice, as torch does.
#include "scp_internal.h"

typedef float ef32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 ef16x4 __attribute__((ext_vector_type(4)));

struct OaEmbedArgs {
    const uint8_t *ctx;        // [n][12] = (occ, level, octant) x (ggp, gp, p, self)
    const float *pos;          // [n][4][3]
    const float *occ_enc, *level_enc, *octant_enc, *pos_w, *pos_b, *pe;   // [256|..][do], [..][dl], [9][dt], [dp][3], [dp], [c][D]
    float *E;                  // [2][n][D]
    _Float16 *hi, *lo;         // [2 n][ldp]
    float *sc, *isc;           // [2 n]
    int64_t n, ldp;
    int c, D, d_occ, d_lvl, d_oct, d_pos, cap, max_level;
    float scale;               // sqrt(D) as float32
};

__global__ __launch_bounds__(256) void oa_embed_kernel(OaEmbedArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.n) return;
    const int da = a.d_occ + a.d_lvl + a.d_oct + a.d_pos;          // channels per ancestor
    const uint8_t *cx = a.ctx + r * 12;
    int occ[4], lvl[4], oct[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { occ[k] = cx[3 * k]; lvl[k] = cx[3 * k + 1]; oct[k] = cx[3 * k + 2]; }
    // oct_attention.py:57-61: levels are shifted so that the node's own level does not exceed the cap, then clipped to the table
    const int sh = lvl[3] - a.cap > 0 ? lvl[3] - a.cap : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { int l = lvl[k] - sh; lvl[k] = l < 0 ? 0 : (l > a.max_level ? a.max_level : l); }
    const float *pp = a.pos + r * 12;
    const float *per = a.pe + (int64_t)(r % a.c) * a.D;
    float vk[3][4], vu[3][4];
    float mk = 0.f, mu = 0.f;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = 4 * lane + 256 * it + u;
            float x = 0.f, xu = 0.f;
            if (e < a.D) {
                const int an = e / da, j = e - an * da;
                float val, valu;
                if (j < a.d_occ) {
                    val = a.occ_enc[occ[an] * a.d_occ + j];
                    valu = (an == 3) ? a.occ_enc[255 * a.d_occ + j] : val;
                } else if (j < a.d_occ + a.d_lvl) {
                    val = valu = a.level_enc[lvl[an] * a.d_lvl + (j - a.d_occ)];
                } else if (j < a.d_occ + a.d_lvl + a.d_oct) {
                    val = valu = a.octant_enc[oct[an] * a.d_oct + (j - a.d_occ - a.d_lvl)];
                } else {
                    const int t = j - a.d_occ - a.d_lvl - a.d_oct;
                    // the k-ordered fp32 chain of scp_linear_f32 (K = 3), bias added to the finished sum
                    float s = __builtin_fmaf(pp[3 * an], a.pos_w[3 * t], 0.f);
                    s = __builtin_fmaf(pp[3 * an + 1], a.pos_w[3 * t + 1], s);
                    s = __builtin_fmaf(pp[3 * an + 2], a.pos_w[3 * t + 2], s);
                    val = valu = s + a.pos_b[t];
                }
                const float p = per[e];
                x = val * a.scale + p;            // two roundings (no contraction in this file)
                xu = valu * a.scale + p;
            }
            vk[it][u] = x; vu[it][u] = xu;
            mk = fmaxf(mk, fabsf(x)); mu = fmaxf(mu, fabsf(xu));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mk = fmaxf(mk, __shfl_xor(mk, o)); mu = fmaxf(mu, __shfl_xor(mu, o)); }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int64_t row = (int64_t)s * a.n + r;
        float sc, isc;
        scp_pow2_scale(s ? mu : mk, sc, isc);
        if (lane == 0) { a.sc[row] = sc; a.isc[row] = isc; }
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int k = 4 * lane + 256 * it;
            if (k >= a.ldp) continue;
            const float *v = s ? vu[it] : vk[it];
            if (k < a.D) *(ef32x4 *)(a.E + row * a.D + k) = (ef32x4){v[0], v[1], v[2], v[3]};
            ef16x4 h4, l4;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float x = v[u] * sc;
                const _Float16 hh = (_Float16)x;
                h4[u] = hh;
                l4[u] = (_Float16)(x - (float)hh);
            }
            *(ef16x4 *)(a.hi + row * a.ldp + k) = h4;
            *(ef16x4 *)(a.lo + row * a.ldp + k) = l4;
        }
    }
}

// ctx uint8 [n][12], pos fp32 [n][4][3] (n = B * c window rows, row r sits at position r % c of its window) -> E fp32 [2][n][D] (stream 0
// = known, 1 = unknown) and its f16x3 operand: planes [2 n][ldp] (ldp >= D rounded up to 32, % 8 == 0), scale / inv_scale [2 n].
// D = 4 (d_occ + d_lvl + d_oct + d_pos) <= 768, D % 4 == 0; pos_w [d_pos][3], pos_b [d_pos] (d_pos may be 0); pe [c][D].
extern "C" SCP_API int scp_octattn_embed(const uint8_t *ctx, const float *pos, int64_t n, int32_t c, const float *occ_enc, int32_t d_occ,
                                         const float *level_enc, int32_t d_lvl, int32_t max_level, const float *octant_enc, int32_t d_oct,
                                         const float *pos_w, const float *pos_b, int32_t d_pos, const float *pe, int32_t level_cap, float *E,
                                         void *hi, void *lo, int64_t ldp, float *scale, float *inv_scale, void *stream) {
    const int D = 4 * (d_occ + d_lvl + d_oct + d_pos);
    if (!ctx || !pos || !occ_enc || !level_enc || !octant_enc || !pe || !E || !hi || !lo || !scale || !inv_scale || n <= 0 || c <= 0 || d_occ <= 0 ||
        d_lvl <= 0 || d_oct <= 0 || d_pos < 0 || (d_pos > 0 && (!pos_w || !pos_b)) || D > 768 || (D & 3) || ldp < ((D + 31) & ~31) || (ldp & 7) ||
        ldp > 768 || max_level < 0 || (((uintptr_t)E | (uintptr_t)hi | (uintptr_t)lo) & 15))
        return SCP_EINVAL;
    OaEmbedArgs a;
    a.ctx = ctx; a.pos = pos; a.occ_enc = occ_enc; a.level_enc = level_enc; a.octant_enc = octant_enc; a.pos_w = pos_w; a.pos_b = pos_b; a.pe = pe;
    a.E = E; a.hi = (_Float16 *)hi; a.lo = (_Float16 *)lo; a.sc = scale; a.isc = inv_scale; a.n = n; a.ldp = ldp; a.c = c; a.D = D;
    a.d_occ = d_occ; a.d_lvl = d_lvl; a.d_oct = d_oct; a.d_pos = d_pos; a.cap = level_cap; a.max_level = max_level;
    a.scale = (float)sqrt((double)D);      // math.sqrt(D), cast to float32 when it meets the float32 tensor
    SCP_PROF(SCP_PROF_OTHER, stream, (double)n * (2.0 * D * 4 + 2.0 * ldp * 4));
    hipLaunchKernelGGL(oa_embed_kernel, dim3((unsigned)cdiv64(n, 4)), dim3(256), 0, (hipStream_t)stream, a);
    LAUNCH_CHECK();
    return SCP_OK;
}
