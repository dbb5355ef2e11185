// Token-wise glue of the packed EHEM forward, fused so that every activation is read and written once (gfx950).
//
//   scp_layernorm_rows : out[r] = valid[r] * LayerNorm(cat(x[ia[r]], x[ib[r]]))  -- LayerNorm (eps 1e-5) with an optional row
//                        gather of one or two 256-wide sources (Swin patch merging, swin_transformer.py:350-367, reads the even
//                        and odd token of a pair; index == n_rows means "zero row") and the zeroing of the rows a window pads
//                        AFTER LayerNorm (swin_transformer.py:638-641).  One wavefront per row, 16 B per lane, two-pass
//                        mean / variance in float32 with wave shuffles.
//   scp_gather_rows    : out[r][col0 : col0 + C] = src[idx[r]]  -- the stage gathers of concat_states (ehem.py:75-86) and the
//                        even/odd token split (ehem.py:113-114) written straight into their slot of the concatenated buffer.
#include "scp_internal.h"

typedef _Float16 sf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int NV>   // float4 per lane: C = 256 * NV
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float *__restrict__ x, int64_t ldx, int64_t n_src_rows,
                                                            const int64_t *__restrict__ ia, const int64_t *__restrict__ ib,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            const float *__restrict__ valid, float eps, float *__restrict__ out, int64_t ldo,
                                                            int64_t rows, __bf16 *__restrict__ ohi, __bf16 *__restrict__ olo) {
    // a wavefront takes RPW consecutive rows and issues all their loads before it reduces the first one: one row per wavefront
    // left only ~32 KiB in flight per CU, i.e. the kernel ran at the latency, not the bandwidth, of HBM
    constexpr int RPW = 4;
    constexpr int C = 256 * NV;
    const int lane = threadIdx.x & 63;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (r0 >= rows) return;
    f32x4 v[RPW][NV];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int64_t r = r0 + i < rows ? r0 + i : rows - 1;
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            // part p (256 channels) comes from x[ia[r]] (p == 0) or x[ib[r]] (p == 1); without a gather the row is x[r] itself
            int64_t src = r;
            if (ia) src = (p == 0) ? ia[r] : ib[r];
            const int64_t off = ia ? 0 : (int64_t)p * 256;
            v[i][p] = (src < n_src_rows) ? *(const f32x4 *)(x + src * ldx + off + 4 * lane) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    f32x4 g[NV], b[NV];
#pragma unroll
    for (int p = 0; p < NV; ++p) { g[p] = *(const f32x4 *)(gamma + p * 256 + 4 * lane); b[p] = *(const f32x4 *)(beta + p * 256 + 4 * lane); }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int64_t r = r0 + i;
        if (r >= rows) break;                            // wave-uniform
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < NV; ++p) s += (v[i][p][0] + v[i][p][1]) + (v[i][p][2] + v[i][p][3]);
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int p = 0; p < NV; ++p)
#pragma unroll
            for (int u = 0; u < 4; ++u) { const float d = v[i][p][u] - mean; q += d * d; }
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = rsqrtf(q * (1.0f / C) + eps);
        const float keep = valid ? valid[r] : 1.0f;
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            f32x4 y;
#pragma unroll
            for (int u = 0; u < 4; ++u) y[u] = ((v[i][p][u] - mean) * rstd * g[p][u] + b[p][u]) * keep;
            if (ohi) {   // hi/lo bf16 planes (operand format of scp_linear_split)
                bf16x4 vh, vl;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float yy = y[u];
                    asm volatile("" : "+v"(yy));   // the rounded fp32 value, not an FMA-contracted (.. * keep) - hi
                    const __bf16 hh = (__bf16)yy;
                    vh[u] = hh;
                    vl[u] = (__bf16)(yy - (float)hh);
                }
                *(bf16x4 *)(ohi + r * ldo + p * 256 + 4 * lane) = vh;
                *(bf16x4 *)(olo + r * ldo + p * 256 + 4 * lane) = vl;
            } else *(f32x4 *)(out + r * ldo + p * 256 + 4 * lane) = y;
        }
    }
}

static int ln_rows(const float *x, int64_t ldx, int64_t n_src_rows, const int64_t *ia, const int64_t *ib, int32_t C, const float *gamma,
                   const float *beta, const float *valid, float eps, float *out, __bf16 *ohi, __bf16 *olo, int64_t ldo, int64_t rows, void *stream) {
    if (!x || !gamma || !beta || (!out && !ohi) || rows < 0 || (C != 256 && C != 512) || (ldx & 3) || (ldo & 3) || ldo < C ||
        ((ia == nullptr) != (ib == nullptr) && C == 512) || (ia && C == 256 && ib) ||
        (((uintptr_t)x | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15) || (ohi && (!olo || (((uintptr_t)ohi | (uintptr_t)olo) & 7))))
        return SCP_EINVAL;
    if (rows == 0) return SCP_OK;
    const unsigned nb = (unsigned)cdiv64(rows, 16);   // 4 wavefronts x 4 rows per workgroup
    hipStream_t st = (hipStream_t)stream;
    SCP_PROF(SCP_PROF_LAYERNORM, st, 8.0 * rows * C);
    if (C == 256) hipLaunchKernelGGL(layernorm_rows_kernel<1>, dim3(nb), dim3(256), 0, st, x, ldx, n_src_rows, ia, ia, gamma, beta, valid, eps, out, ldo, rows, ohi, olo);
    else hipLaunchKernelGGL(layernorm_rows_kernel<2>, dim3(nb), dim3(256), 0, st, x, ldx, n_src_rows, ia, ib, gamma, beta, valid, eps, out, ldo, rows, ohi, olo);
    LAUNCH_CHECK();
    return SCP_OK;
}

extern "C" SCP_API int scp_layernorm_rows(const float *x, int64_t ldx, int64_t n_src_rows, const int64_t *ia, const int64_t *ib, int32_t C,
                                          const float *gamma, const float *beta, const float *valid, float eps, float *out, int64_t ldo,
                                          int64_t rows, void *stream) {
    if (!out) return SCP_EINVAL;
    return ln_rows(x, ldx, n_src_rows, ia, ib, C, gamma, beta, valid, eps, out, nullptr, nullptr, ldo, rows, stream);
}

// the same, output written as bf16 hi/lo planes [rows][ldo] (operand format of scp_linear_split)
extern "C" SCP_API int scp_layernorm_rows_split(const float *x, int64_t ldx, int64_t n_src_rows, const int64_t *ia, const int64_t *ib, int32_t C,
                                                const float *gamma, const float *beta, const float *valid, float eps, void *ohi, void *olo,
                                                int64_t ldo, int64_t rows, void *stream) {
    if (!ohi || !olo) return SCP_EINVAL;
    return ln_rows(x, ldx, n_src_rows, ia, ib, C, gamma, beta, valid, eps, nullptr, (__bf16 *)ohi, (__bf16 *)olo, ldo, rows, stream);
}

// LayerNorm(a + b) for any row width C = 4 * NQ (OctAttention: 600; attention_model.py:117,123 `norm(x + residual)`): one wavefront per
// row, the row's float4 pieces round-robin over the lanes, both operands read once, the sum never written.  b may be null.
// PLANES: the row also leaves as the f16x3 operand of the dense layers that read it (scp_linear_split_f16): power-of-two row scale of
// the OUTPUT row, its inverse, and the two IEEE-half planes of the scaled row (columns C .. Cp zero) - what scp_split_rows_f16 would
// make of `out` in a second pass over it (same arithmetic, same bits).
template <int NP, bool PLANES = false>   // NP: float4 pieces per lane (C <= 256 * NP)
__global__ __launch_bounds__(256) void layernorm_add_kernel(const float *__restrict__ a, const float *__restrict__ b, int64_t rows, int C,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                           float *__restrict__ out, _Float16 *__restrict__ phi = nullptr,
                                                           _Float16 *__restrict__ plo = nullptr, int64_t ldp = 0, float *__restrict__ psc = nullptr,
                                                           float *__restrict__ pisc = nullptr) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int nq = C >> 2;
    f32x4 v[NP];
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int q = lane + 64 * p;
        v[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (q < nq) {
            v[p] = *(const f32x4 *)(a + r * C + 4 * q);
            if (b) v[p] += *(const f32x4 *)(b + r * C + 4 * q);
            s += (v[p][0] + v[p][1]) + (v[p][2] + v[p][3]);
        }
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float qs = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p)
        if (lane + 64 * p < nq)
#pragma unroll
            for (int u = 0; u < 4; ++u) { const float d = v[p][u] - mean; qs += d * d; }
    for (int o = 32; o > 0; o >>= 1) qs += __shfl_xor(qs, o);
    const float rstd = rsqrtf(qs / (float)C + eps);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int q = lane + 64 * p;
        if (q < nq) {
            const f32x4 g = *(const f32x4 *)(gamma + 4 * q), bb = *(const f32x4 *)(beta + 4 * q);
            f32x4 y;
#pragma unroll
            for (int u = 0; u < 4; ++u) y[u] = (v[p][u] - mean) * rstd * g[u] + bb[u];
            *(f32x4 *)(out + r * C + 4 * q) = y;
            if (PLANES) v[p] = y;
        }
    }
    if (PLANES) {
        float mx = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p)
            if (lane + 64 * p < nq) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[p][0]), fabsf(v[p][1]))), fmaxf(fabsf(v[p][2]), fabsf(v[p][3])));
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sc, isc;
        scp_pow2_scale(mx, sc, isc);
        if (lane == 0) { psc[r] = sc; pisc[r] = isc; }
        const int nqp = ((C + 31) & ~31) >> 2;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int q = lane + 64 * p;
            if (q >= nqp) continue;
            sf16x4 h4, l4;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float x = (q < nq ? v[p][u] : 0.f) * sc;
                const _Float16 hh = (_Float16)x;
                h4[u] = hh;
                l4[u] = (_Float16)(x - (float)hh);
            }
            *(sf16x4 *)(phi + r * ldp + 4 * q) = h4;
            *(sf16x4 *)(plo + r * ldp + 4 * q) = l4;
        }
    }
}

extern "C" SCP_API int scp_layernorm_add(const float *a, const float *b, int64_t rows, int32_t C, const float *gamma, const float *beta, float eps,
                                         float *out, void *stream) {
    if (!a || !gamma || !beta || !out || rows < 0 || C <= 0 || (C & 3) || C > 1024 ||
        (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15))
        return SCP_EINVAL;
    if (rows == 0) return SCP_OK;
    const unsigned nb = (unsigned)cdiv64(rows, 4);
    hipStream_t st = (hipStream_t)stream;
    SCP_PROF(SCP_PROF_LAYERNORM, st, (b ? 12.0 : 8.0) * rows * C);
    if (C <= 256) hipLaunchKernelGGL(layernorm_add_kernel<1>, dim3(nb), dim3(256), 0, st, a, b, rows, C, gamma, beta, eps, out);
    else if (C <= 512) hipLaunchKernelGGL(layernorm_add_kernel<2>, dim3(nb), dim3(256), 0, st, a, b, rows, C, gamma, beta, eps, out);
    else if (C <= 768) hipLaunchKernelGGL(layernorm_add_kernel<3>, dim3(nb), dim3(256), 0, st, a, b, rows, C, gamma, beta, eps, out);
    else hipLaunchKernelGGL(layernorm_add_kernel<4>, dim3(nb), dim3(256), 0, st, a, b, rows, C, gamma, beta, eps, out);
    LAUNCH_CHECK();
    return SCP_OK;
}

// scp_layernorm_add + the f16x3 operand of its output in the same pass (see layernorm_add_kernel<.., true>): planes [rows][ldp] (ldp >= C
// rounded up to 32, ldp % 8 == 0), scale / inv_scale [rows].
extern "C" SCP_API int scp_layernorm_add_split_f16(const float *a, const float *b, int64_t rows, int32_t C, const float *gamma, const float *beta,
                                                   float eps, float *out, void *hi, void *lo, int64_t ldp, float *scale, float *inv_scale,
                                                   void *stream) {
    const int Cp = (C + 31) & ~31;
    if (!a || !gamma || !beta || !out || !hi || !lo || !scale || !inv_scale || rows < 0 || C <= 0 || (C & 3) || Cp > 1024 || ldp < Cp || (ldp & 7) ||
        (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)hi | (uintptr_t)lo) & 15))
        return SCP_EINVAL;
    if (rows == 0) return SCP_OK;
    const unsigned nb = (unsigned)cdiv64(rows, 4);
    hipStream_t st = (hipStream_t)stream;
    SCP_PROF(SCP_PROF_LAYERNORM, st, (b ? 16.0 : 12.0) * rows * C);
#define GOP(NP_) hipLaunchKernelGGL((layernorm_add_kernel<NP_, true>), dim3(nb), dim3(256), 0, st, a, b, rows, C, gamma, beta, eps, out, (_Float16 *)hi, \
                                    (_Float16 *)lo, ldp, scale, inv_scale)
    if (Cp <= 256) GOP(1); else if (Cp <= 512) GOP(2); else if (Cp <= 768) GOP(3); else GOP(4);
#undef GOP
    LAUNCH_CHECK();
    return SCP_OK;
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ src, int64_t lds, const int64_t *__restrict__ idx, int C4,
                                                         float *__restrict__ out, int64_t ldo, int64_t total /* rows * C4 */) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const int64_t r = g / C4;
    const int c = (int)(g - r * C4) * 4;
    *(f32x4 *)(out + r * ldo + c) = *(const f32x4 *)(src + idx[r] * lds + c);
}

extern "C" SCP_API int scp_gather_rows(const float *src, int64_t lds, const int64_t *idx, int32_t C, float *out, int64_t ldo, int64_t rows,
                                       void *stream) {
    if (!src || !idx || !out || rows < 0 || C <= 0 || (C & 3) || (lds & 3) || (ldo & 3) || (((uintptr_t)src | (uintptr_t)out) & 15)) return SCP_EINVAL;
    if (rows == 0) return SCP_OK;
    const int64_t total = rows * (C / 4);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, src, lds, idx, C / 4, out, ldo,
                       total);
    LAUNCH_CHECK();
    return SCP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Input stage of the packed EHEM forward (dgcnn.py:121-128 embeddings + the packed layout's input gather) in one kernel:
//   row r of the packed layout takes token t = inmap[r] of the frame arrays (t == n_tokens: the pad token (level 0, octant 0,
//   occ 255) x 4, position 0) and gets
//   x[r]   = [occ_enc[occ of ancestors 0..2] (3 x 16) | level_enc[level of rows 0..3] (4 x 4) | octant_enc[octant 0..3] (4 x 4)]
//   pos[r] = position of the token, occ_self[r] = its own occupancy symbol (int64, for the even-token embedding of phase 2).
// ctx is the compact context of stage G: uint8 [T][12] = 4 x (level, octant, occ).  One thread per 16-byte piece of x.
__global__ __launch_bounds__(256) void embed_gather_kernel(const unsigned char *__restrict__ ctx, const float *__restrict__ pos,
                                                          const int64_t *__restrict__ inmap, int64_t n_tokens, const float *__restrict__ occ_enc,
                                                          const float *__restrict__ level_enc, const float *__restrict__ octant_enc,
                                                          float *__restrict__ x, float *__restrict__ pos_out, int64_t *__restrict__ occ_self,
                                                          int64_t rows) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= rows * 20) return;
    const int64_t r = g / 20;
    const int p = (int)(g - r * 20);
    const int64_t t = inmap[r];
    const bool pad = t >= n_tokens;
    const unsigned char *c = ctx + (pad ? 0 : t) * 12;
    f32x4 v;
    if (p < 12) {            // occupancy embedding of ancestor a = p / 4 (context column 3 a + 2), 16 floats = 4 pieces
        const int a = p >> 2;
        const int occ = pad ? 255 : c[3 * a + 2];
        v = *(const f32x4 *)(occ_enc + occ * 16 + 4 * (p & 3));
    } else if (p < 16) {     // level embedding of context row a = p - 12
        const int lv = pad ? 0 : c[3 * (p - 12)];
        v = *(const f32x4 *)(level_enc + lv * 4);
    } else {                 // octant embedding of context row a = p - 16
        const int oc = pad ? 0 : c[3 * (p - 16) + 1];
        v = *(const f32x4 *)(octant_enc + oc * 4);
    }
    *(f32x4 *)(x + r * 80 + 4 * p) = v;
    if (p == 0) {
        occ_self[r] = pad ? 255 : c[11];
        pos_out[3 * r] = pad ? 0.f : pos[3 * t];
        pos_out[3 * r + 1] = pad ? 0.f : pos[3 * t + 1];
        pos_out[3 * r + 2] = pad ? 0.f : pos[3 * t + 2];
    }
}

extern "C" SCP_API int scp_embed_gather(const uint8_t *ctx, const float *pos, const int64_t *inmap, int64_t n_tokens, const float *occ_enc,
                                        const float *level_enc, const float *octant_enc, float *x, float *pos_out, int64_t *occ_self,
                                        int64_t rows, void *stream) {
    if (!ctx || !pos || !inmap || !occ_enc || !level_enc || !octant_enc || !x || !pos_out || !occ_self || rows < 0 || n_tokens <= 0 ||
        (((uintptr_t)occ_enc | (uintptr_t)level_enc | (uintptr_t)octant_enc | (uintptr_t)x) & 15))
        return SCP_EINVAL;
    if (rows == 0) return SCP_OK;
    hipLaunchKernelGGL(embed_gather_kernel, dim3((unsigned)cdiv64(rows * 20, 256)), dim3(256), 0, (hipStream_t)stream, ctx, pos, inmap, n_tokens,
                       occ_enc, level_enc, octant_enc, x, pos_out, occ_self, rows);
    LAUNCH_CHECK();
    return SCP_OK;
}
