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

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int NV>   // float4 per lane: C = 256 * NV
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float *__restrict__ x, int64_t ldx, int64_t n_src_rows,
                                                            const int64_t *__restrict__ ia, const int64_t *__restrict__ ib,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            const float *__restrict__ valid, float eps, float *__restrict__ out, int64_t ldo,
                                                            int64_t rows, __bf16 *__restrict__ ohi, __bf16 *__restrict__ olo) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    constexpr int C = 256 * NV;
    f32x4 v[NV];
#pragma unroll
    for (int p = 0; p < NV; ++p) {
        // part p (256 channels) comes from x[ia[r]] (p == 0) or x[ib[r]] (p == 1); without a gather the row is x[r] itself
        int64_t src = r;
        if (ia) src = (p == 0) ? ia[r] : ib[r];
        const int64_t off = ia ? 0 : (int64_t)p * 256;
        v[p] = (src < n_src_rows) ? *(const f32x4 *)(x + src * ldx + off + 4 * lane) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < NV; ++p) s += (v[p][0] + v[p][1]) + (v[p][2] + v[p][3]);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int p = 0; p < NV; ++p)
#pragma unroll
        for (int u = 0; u < 4; ++u) { const float d = v[p][u] - mean; q += d * d; }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q * (1.0f / C) + eps);
    const float keep = valid ? valid[r] : 1.0f;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
        const f32x4 g = *(const f32x4 *)(gamma + p * 256 + 4 * lane), b = *(const f32x4 *)(beta + p * 256 + 4 * lane);
        f32x4 y;
#pragma unroll
        for (int u = 0; u < 4; ++u) y[u] = ((v[p][u] - mean) * rstd * g[u] + b[u]) * keep;
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

static int ln_rows(const float *x, int64_t ldx, int64_t n_src_rows, const int64_t *ia, const int64_t *ib, int32_t C, const float *gamma,
                   const float *beta, const float *valid, float eps, float *out, __bf16 *ohi, __bf16 *olo, int64_t ldo, int64_t rows, void *stream) {
    if (!x || !gamma || !beta || (!out && !ohi) || rows < 0 || (C != 256 && C != 512) || (ldx & 3) || (ldo & 3) || ldo < C ||
        ((ia == nullptr) != (ib == nullptr) && C == 512) || (ia && C == 256 && ib) ||
        (((uintptr_t)x | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15) || (ohi && (!olo || (((uintptr_t)ohi | (uintptr_t)olo) & 7))))
        return SCP_EINVAL;
    if (rows == 0) return SCP_OK;
    const unsigned nb = (unsigned)cdiv64(rows, 4);
    hipStream_t st = (hipStream_t)stream;
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
