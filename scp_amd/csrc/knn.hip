// k-nearest-neighbour search in feature space, fused distance + top-k (gfx950, wave64, fp32 MFMA).
//
// Replaces models/dgcnn.py:10-45 (`knn`): the reference materialises the [n][n] matrix
//     pd = 2 * x_i.x_j - |x_j|^2 - |x_i|^2      (float32, evaluated as ((2*dot) - xx_j) - xx_i)
// and calls topk(20).  Here a workgroup owns 64 queries and streams 64-candidate tiles through LDS; the
// 64 x 64 distance tile comes out of v_mfma_f32_16x16x4_f32 (exact float32 k-ordered FMA chain, candidates on
// the MFMA rows, queries on the columns) so that every lane ends up with 16 candidates of ONE query and keeps
// a private sorted top-20 (value, index) list in registers.  Four lanes share a query; their lists are merged
// through LDS at the end.  Candidate tiles are visited outwards from the query's own tile: octree siblings are
// Morton neighbours, so the threshold tightens at once and later insertions are rare.
// Order: value descending, ties -> lower index.  Nothing of size n x n is ever stored.
#include <float.h>
#include <limits.h>
#include "scp_internal.h"

#define TK 20
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void topk_insert(float (&v)[TK], int (&id)[TK], float d, int j) {
#pragma unroll
    for (int t = 0; t < TK; ++t) {
        const bool better = (d > v[t]) || (d == v[t] && j < id[t]);
        const float nv = better ? d : v[t], od = better ? v[t] : d;
        const int ni = better ? j : id[t], oj = better ? id[t] : j;
        v[t] = nv; d = od; id[t] = ni; j = oj;
    }
}

__global__ __launch_bounds__(256) void knn_kernel(const float *__restrict__ x, int n, int C, int ldC, int k, int *__restrict__ idx) {
    extern __shared__ float smem[];
    float *Q = smem;                    // [64][ldC] query features
    float *Cd = Q + 64 * ldC;           // [64][ldC] candidate tile
    float *xxq = Cd + 64 * ldC;         // [64]
    float *xxc = xxq + 64;              // [64]
    float *mval = xxc + 64;             // [64][4][TK]
    int *midx = (int *)(mval + 64 * 4 * TK);

    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, col = lane & 15, kq = lane >> 4;
    const float *xb = x + (size_t)blockIdx.y * n * C;
    const int q0 = blockIdx.x * 64;
    const int Cpad = (C + 3) & ~3, KS = Cpad >> 2;
    const int nt = (n + 63) >> 6;

    for (int e = tid; e < 64 * Cpad; e += 256) {
        const int r = e / Cpad, c = e - r * Cpad;
        Q[r * ldC + c] = (c < C && q0 + r < n) ? xb[(size_t)(q0 + r) * C + c] : 0.f;
    }
    __syncthreads();
    if (tid < 64) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) { const float a = Q[tid * ldC + c]; s = __fadd_rn(s, __fmul_rn(a, a)); }
        xxq[tid] = s;
    }
    __syncthreads();
    const float xxi = xxq[w * 16 + col];

    float v[TK];
    int id[TK];
#pragma unroll
    for (int t = 0; t < TK; ++t) { v[t] = -INFINITY; id[t] = INT_MAX; }

    int lo = (int)blockIdx.x - 1, hi = (int)blockIdx.x + 1, tile = blockIdx.x;
    for (int s = 0; s < nt; ++s) {
        const int c0 = tile * 64;
        __syncthreads();  // previous tile fully consumed
        for (int e = tid; e < 64 * Cpad; e += 256) {
            const int r = e / Cpad, c = e - r * Cpad;
            Cd[r * ldC + c] = (c < C && c0 + r < n) ? xb[(size_t)(c0 + r) * C + c] : 0.f;
        }
        __syncthreads();
        if (tid < 64) {
            float sq = 0.f;
            for (int c = 0; c < C; ++c) { const float a = Cd[tid * ldC + c]; sq = __fadd_rn(sq, __fmul_rn(a, a)); }
            xxc[tid] = sq;
        }
        __syncthreads();

        f32x4 acc[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float *qrow = Q + (w * 16 + col) * ldC + kq;
        const float *crow = Cd + col * ldC + kq;
        for (int ks = 0; ks < KS; ++ks) {
            const float bq = qrow[4 * ks];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                const float a = crow[rt * 16 * ldC + 4 * ks];
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq, acc[rt], 0, 0, 0);
            }
        }
        // lane holds D[row = kq*4 + r][col] of each 16x16 tile: candidates rt*16 + kq*4 + r of query `col`
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cl = rt * 16 + kq * 4 + r;
                const int j = c0 + cl;
                const float d = __fsub_rn(__fsub_rn(__fmul_rn(2.f, acc[rt][r]), xxc[cl]), xxi);
                if (j < n && ((d > v[TK - 1]) || (d == v[TK - 1] && j < id[TK - 1]))) topk_insert(v, id, d, j);
            }
        }
        // next tile: alternate right / left of the own tile
        if ((s & 1) == 0) { if (hi < nt) tile = hi++; else tile = lo--; }
        else { if (lo >= 0) tile = lo--; else tile = hi++; }
    }

    // merge the four partial lists of every query
    {
        const int ql = w * 16 + col;
#pragma unroll
        for (int t = 0; t < TK; ++t) { mval[(ql * 4 + kq) * TK + t] = v[t]; midx[(ql * 4 + kq) * TK + t] = id[t]; }
    }
    __syncthreads();
    if (tid < 64 && q0 + tid < n) {
        int h[4] = {0, 0, 0, 0};
        int *out = idx + ((size_t)blockIdx.y * n + q0 + tid) * k;
        for (int o = 0; o < k; ++o) {
            int best = -1;
            float bv = 0.f;
            int bi = 0;
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                if (h[l] >= TK) continue;
                const float cv = mval[(tid * 4 + l) * TK + h[l]];
                const int ci = midx[(tid * 4 + l) * TK + h[l]];
                if (best < 0 || cv > bv || (cv == bv && ci < bi)) { best = l; bv = cv; bi = ci; }
            }
            out[o] = bi;
#pragma unroll
            for (int l = 0; l < 4; ++l) if (l == best) h[l]++;
        }
    }
}

extern "C" int scp_knn_topk(const float *x, int32_t B, int32_t n, int32_t C, int32_t k, int32_t *idx, void *stream) {
    if (!x || !idx || B <= 0 || n <= 0 || C <= 0 || k <= 0 || k > TK || k > n) return SCP_EINVAL;
    const int Cpad = (C + 3) & ~3;
    int ldC = Cpad;
    while ((ldC & 3) != 2) ++ldC;  // ldC/2 odd: the 16 candidate rows of a fragment read hit 16 distinct even banks
    const size_t lds = ((size_t)2 * 64 * ldC + 128 + (size_t)64 * 4 * TK * 2) * sizeof(float);
    if (lds > 160 * 1024) return SCP_EINVAL;
    static size_t configured = 0;
    if (lds > configured) {
        HIP_TRY(hipFuncSetAttribute((const void *)knn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
    }
    hipLaunchKernelGGL(knn_kernel, dim3((n + 63) / 64, B), dim3(256), lds, (hipStream_t)stream, x, n, C, ldC, k, idx);
    LAUNCH_CHECK();
    return SCP_OK;
}
