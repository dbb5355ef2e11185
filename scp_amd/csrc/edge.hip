// Edge-conv tail: neighbour gather + max, fused with BatchNorm(eval) and LeakyReLU(0.2) (gfx950).
//
// Reference (models/dgcnn.py:48-71,132-134): feature[i][j] = cat(f_j - f_i, f_i) for the k neighbours j, 1x1 conv W,
// BatchNorm, LeakyReLU(0.2), max over j.  With W = [W1 | W2]:  W.cat(f_j - f_i, f_i) = W1 f_j + (W2 - W1) f_i, so
// the conv is two [n x C] x [C x C'] GEMMs (u = F W1^T, v = F (W2-W1)^T, done once per point instead of k = 20
// times) and BN/LeakyReLU are monotone per channel, hence
//     max_j act(scale*(u_j + v_i) + shift) = act(scale * (sel_j u_j + v_i) + shift),  sel = max if scale >= 0 else min.
// This kernel is the gather: k rows of C' floats per point, 16 B per lane, rows served from L2 / Infinity Cache.
#include "scp_internal.h"

__global__ __launch_bounds__(256) void edge_gather_max_kernel(const float *__restrict__ u, const float *__restrict__ v,
                                                             const int *__restrict__ idx, const float *__restrict__ scale,
                                                             const float *__restrict__ shift, int n, int Cout, int k,
                                                             float *__restrict__ out, int out_stride, int64_t total /* B*n*Cout/4 */,
                                                             int64_t ldu, int64_t ldv) {
    // XCD-affine block order (round 5): workgroup b runs on XCD b & 7; give every XCD one contiguous eighth of the points, so that the u rows
    // a window's points gather (2 - 8 MB per 8192-point window) are fetched into ONE XCD's L2 instead of all eight (PMC before: 2.9 x the
    // unique bytes from HBM, at the achievable HBM ceiling).  The grid is a multiple of 8 (launcher), so the map is a bijection.
    const int64_t blk = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int64_t g = blk * 256 + threadIdx.x;
    if (g >= total) return;
    const int c4 = Cout >> 2;
    const int64_t pt = g / c4;          // global point index b*n + i
    const int c = (int)(g - pt * c4) * 4;
    const int64_t b = pt / n;
    const float *ub = u + b * (int64_t)n * ldu;
    const int *nb = idx + pt * k;
    float4 mx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    float4 mn = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
    auto acc = [&](const float4 a) {
        mx.x = fmaxf(mx.x, a.x); mx.y = fmaxf(mx.y, a.y); mx.z = fmaxf(mx.z, a.z); mx.w = fmaxf(mx.w, a.w);
        mn.x = fminf(mn.x, a.x); mn.y = fminf(mn.y, a.y); mn.z = fminf(mn.z, a.z); mn.w = fminf(mn.w, a.w);
    };
    if (k == 20 && (((uintptr_t)idx) & 15) == 0) {   // the reference's k: all 20 indices first (five 16-byte loads), then 20 independent row loads in flight
        int id[20];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int4 t = *(const int4 *)(nb + 4 * q);   // rows of 20 ints = 80 B: 16-byte aligned
            id[4 * q] = t.x; id[4 * q + 1] = t.y; id[4 * q + 2] = t.z; id[4 * q + 3] = t.w;
        }
        float4 a[20];
#pragma unroll
        for (int j = 0; j < 20; ++j) a[j] = *(const float4 *)(ub + (int64_t)id[j] * ldu + c);
#pragma unroll
        for (int j = 0; j < 20; ++j) acc(a[j]);
    } else {
        for (int j = 0; j < k; ++j) acc(*(const float4 *)(ub + (int64_t)nb[j] * ldu + c));
    }
    const float4 vi = *(const float4 *)(v + pt * ldv + c);
    const float4 sc = *(const float4 *)(scale + c), sh = *(const float4 *)(shift + c);
    float4 r;
#define FIN(f) { const float y = __fadd_rn(__fmul_rn(sc.f, __fadd_rn(sc.f >= 0.f ? mx.f : mn.f, vi.f)), sh.f); r.f = y > 0.f ? y : __fmul_rn(0.2f, y); }
    FIN(x) FIN(y) FIN(z) FIN(w)
#undef FIN
    *(float4 *)(out + pt * out_stride + c) = r;
}

// u, v with explicit row strides (e.g. the two halves of one [n][2 Cout] GEMM output, no copies)
extern "C" SCP_API int scp_edge_gather_max_ld(const float *u, int64_t ldu, const float *v, int64_t ldv, const int32_t *idx, const float *scale,
                                              const float *shift, int32_t B, int32_t n, int32_t Cout, int32_t k, float *out, int32_t out_stride,
                                              void *stream) {
    if (!u || !v || !idx || !scale || !shift || !out || B <= 0 || n <= 0 || Cout <= 0 || (Cout & 3) || k <= 0 || out_stride < Cout ||
        (out_stride & 3) || ldu < Cout || ldv < Cout || (ldu & 3) || (ldv & 3) || (((uintptr_t)out | (uintptr_t)u | (uintptr_t)v) & 15))
        return SCP_EINVAL;
    const int64_t total = (int64_t)B * n * (Cout / 4);
    // algorithmic HBM bytes: every u row, every v row and every output row once, and the index lists (the 20 gathered rows per point are
    // re-reads; PMC: the kernel pulls 2.9 x this through HBM at 6.4 TB/s, profiles/r4_pmc_traffic.json)
    SCP_PROF(SCP_PROF_EDGE_GATHER, stream, (double)B * n * (12.0 * Cout + 4.0 * k));
    hipLaunchKernelGGL(edge_gather_max_kernel, dim3((unsigned)(cdiv64(cdiv64(total, 256), 8) * 8)), dim3(256), 0, (hipStream_t)stream, u, v, idx, scale,
                       shift, n, Cout, k, out, out_stride, total, ldu, ldv);
    LAUNCH_CHECK();
    return SCP_OK;
}

extern "C" int scp_edge_gather_max(const float *u, const float *v, const int32_t *idx, const float *scale, const float *shift,
                                   int32_t B, int32_t n, int32_t Cout, int32_t k, float *out, int32_t out_stride, void *stream) {
    return scp_edge_gather_max_ld(u, Cout, v, Cout, idx, scale, shift, B, n, Cout, k, out, out_stride, stream);
}
