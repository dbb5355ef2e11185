// k-nearest-neighbour search in feature space, fused distance + top-k (gfx950, wave64, fp32 MFMA).
//
// Replaces models/dgcnn.py:10-45 (`knn`): the reference materialises the [n][n] matrix
//     pd = 2 * x_i.x_j - |x_j|^2 - |x_i|^2      (float32, evaluated as ((2*dot) - xx_j) - xx_i)
// and calls topk(20).  Nothing of size n x n is stored here.
//
// Numerics: PyTorch-CPU's matmul accumulates a dot product as one k-ordered float32 FMA chain (checked for K = 3..192)
// and v_mfma_f32_32x32x2_f32 is exactly such a chain, so as long as lane half h of the MFMA supplies feature 2*step + h
// the distance VALUES are bit-identical to the reference's; only exactly tied distances can be ordered differently
// (this kernel: lower index first; the reference: whatever its top-k implementation does).
//
// Layout (specialised kernel, K = C rounded up to 4, KS = K/2 MFMA steps):
//   workgroup = 4 waves = 128 queries, a wave owns 32 queries; B operand = the wave's query features, KS registers
//   per lane, loaded once; A operand = a 32-candidate tile staged in LDS de-interleaved as [cand][even k | odd k] so a
//   lane reads its KS values with ds_read_b128; tiles are double buffered through registers (issue the next tile's
//   global loads, run KS MFMAs on the current one, then write LDS: one barrier per tile).
//   D[cand][query]: a lane ends up with 16 candidates of ONE query and keeps a private sorted top-20 (value, index)
//   list in registers; the two lanes of a query share their pruning threshold and merge through LDS at the end.
//   Tiles are visited outwards from the queries' own position: octree siblings are Morton neighbours, so the
//   threshold tightens immediately and later insertions are rare.
#include <float.h>
#include <limits.h>
#include "scp_internal.h"

#define TK 20
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void topk_insert(float (&v)[TK], int (&id)[TK], float d, int j) {
#pragma unroll
    for (int t = 0; t < TK; ++t) {
        const bool better = (d > v[t]) || (d == v[t] && j < id[t]);
        const float nv = better ? d : v[t], od = better ? v[t] : d;
        const int ni = better ? j : id[t], oj = better ? id[t] : j;
        v[t] = nv; d = od; id[t] = ni; j = oj;
    }
}

// |x|^2 per point, summed in feature order with separate roundings (torch.sum(x**2, dim=1), dgcnn.py:19)
__global__ __launch_bounds__(256) void sqnorm_kernel(const float *__restrict__ x, int64_t npts, int C, float *__restrict__ xx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npts) return;
    const float *p = x + i * C;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s = s + p[c] * p[c];   // -ffp-contract=off: two roundings per term
    xx[i] = s;
}

// ------------------------------------------------------------------------------------------------ specialised kernel
template <int KS>   // MFMA k-steps = padded C / 2
__global__ __launch_bounds__(256, 2) void knn_mfma_kernel(const float *__restrict__ x, const float *__restrict__ xx, int n, int C, int k,
                                                         int *__restrict__ idx, const int *__restrict__ ctab /* packed mode: per 512-row chunk (seq base row, seq n) */) {
    constexpr int K = 2 * KS;                       // padded feature count (multiple of 4)
    constexpr int LD = (K % 8 == 0) ? K + 4 : K;    // LDS row stride in floats: LD/4 odd -> conflict-free ds_read_b128
    constexpr int F4 = K / 4;                       // float4 per candidate row
    constexpr int PER_T = (32 * F4 + 255) / 256;    // float4 global loads per thread per tile
    constexpr int TILE_F = 2 * 32 * LD, MERGE_F = 2 * 128 * 2 * TK;          // floats
    constexpr int POOL_F = TILE_F > MERGE_F ? TILE_F : MERGE_F;
    __shared__ __attribute__((aligned(16))) float pool[POOL_F];                // tiles during the sweep, merge lists after it
    __shared__ float txx[2][32];
    float (*tile)[32 * LD] = (float (*)[32 * LD])pool;
    float *mval = pool;
    int *midx = (int *)(pool + 128 * 2 * TK);

    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, col = lane & 31, h = lane >> 5;
    // dense mode: blockIdx.y = batch item with n points.  packed mode: sequences padded to x512 rows lie back to back; the chunk
    // table says which sequence (first row, real length) the 128 query rows of this workgroup belong to; indices come out GLOBAL
    // and always TK per row (rows of sequences shorter than TK repeat their nearest neighbour, harmless under the max-pool).
    size_t row0 = (size_t)blockIdx.y * n;
    int q0 = blockIdx.x * 128;
    if (ctab) {
        const int ch = (blockIdx.x * 128) >> 9;
        row0 = (size_t)ctab[2 * ch];
        n = ctab[2 * ch + 1];
        q0 = blockIdx.x * 128 - (int)row0;
        if (q0 >= n) return;   // workgroup entirely inside the padding of its sequence
    }
    const float *xb = x + row0 * C;
    const float *xxb = xx + row0;
    const int qi = q0 + w * 32 + col;
    const int nt = (n + 31) >> 5;

    // ---- query fragment: features 2*s + h, s = 0..KS-1 ------------------------------------------------------
    float qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int c = 2 * s + h;
        qf[s] = (qi < n && c < C) ? xb[(size_t)qi * C + c] : 0.f;
    }
    const float xxi = (qi < n) ? xxb[qi] : 0.f;

    float v[TK];
    int id[TK];
#pragma unroll
    for (int t = 0; t < TK; ++t) { v[t] = -INFINITY; id[t] = INT_MAX; }

    // ---- tile staging helpers ---------------------------------------------------------------------------------
    f32x4 pre[PER_T];
    float prexx = 0.f;
    auto issue = [&](int t) {   // global -> registers
        const int c0 = t * 32;
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int e = tid + i * 256;
            const int r = e / F4, g = e - r * F4;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (e < 32 * F4 && c0 + r < n) {
                const float *src = xb + (size_t)(c0 + r) * C + 4 * g;
                if ((C & 3) == 0) val = *(const f32x4 *)src;
                else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (4 * g + u < C) val[u] = src[u];
                }
            }
            pre[i] = val;
        }
        if (tid < 32) prexx = (c0 + tid < n) ? xxb[c0 + tid] : 0.f;
    };
    auto commit = [&](int buf) {   // registers -> LDS, de-interleaved: even features first, odd features second
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int e = tid + i * 256;
            const int r = e / F4, g = e - r * F4;
            if (e < 32 * F4) {
                float *row = tile[buf] + r * LD;
                *(float2 *)(row + 2 * g) = make_float2(pre[i][0], pre[i][2]);
                *(float2 *)(row + KS + 2 * g) = make_float2(pre[i][1], pre[i][3]);
            }
        }
        if (tid < 32) txx[buf][tid] = prexx;
    };

    // outward tile order starting at the first tile of this workgroup's own queries
    const int own = q0 >> 5;
    int lo = own - 1, hi = own + 1, cur = own < nt ? own : nt - 1;
    if (own >= nt) { lo = nt - 2; hi = nt; }
    issue(cur);
    commit(0);
    __syncthreads();

    for (int s = 0; s < nt; ++s) {
        const int buf = s & 1;
        // choose and prefetch the next tile
        int nxt = -1;
        if (s + 1 < nt) {
            if ((s & 1) == 0) { if (hi < nt) nxt = hi++; else nxt = lo--; }
            else { if (lo >= 0) nxt = lo--; else nxt = hi++; }
            issue(nxt);
        }
        // ---- distances of 32 candidates x 32 queries ----------------------------------------------------------
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float *arow = tile[buf] + col * LD + h * KS;
        if constexpr (KS % 4 == 0) {
#pragma unroll
            for (int g = 0; g < KS / 4; ++g) {
                const f32x4 a = *(const f32x4 *)(arow + 4 * g);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], qf[4 * g], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], qf[4 * g + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], qf[4 * g + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], qf[4 * g + 3], acc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int g = 0; g < KS / 2; ++g) {
                const float2 a = *(const float2 *)(arow + 2 * g);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qf[2 * g], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qf[2 * g + 1], acc, 0, 0, 0);
            }
        }
        // ---- selection: this lane holds candidates row = (r&3) + 8(r>>2) + 4h of the tile for query qi ------------
        // Pass 1 (cheap, branch-free): distances + a bit mask of the candidates that beat the current 20th best.
        // Pass 2: lanes pop their survivors one at a time, so the wavefront pays for max-over-lanes survivors (1-3 per
        // tile) instead of for all 16 slots as soon as any lane has a hit in each of them.
        const int c0 = cur * 32;
        const float thr = fmaxf(v[TK - 1], __shfl_xor(v[TK - 1], 32));   // both lanes of a query prune with the tighter bound
        float dd[16];
        unsigned pend = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cl = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int j = c0 + cl;
            dd[r] = (2.f * acc[r] - txx[buf][cl]) - xxi;
            const bool pass = j < n && dd[r] >= thr && ((dd[r] > v[TK - 1]) || (dd[r] == v[TK - 1] && j < id[TK - 1]));
            pend |= pass ? (1u << r) : 0u;
        }
        while (__any(pend != 0u)) {   // wave-uniform loop; lanes without a survivor insert a harmless (-inf, INT_MAX)
            const bool act = pend != 0u;
            const int r = act ? (__ffs(pend) - 1) : 0;
            pend &= pend - 1u;
            float d = dd[0];
#pragma unroll
            for (int rr = 1; rr < 16; ++rr) d = (rr == r) ? dd[rr] : d;
            d = act ? d : -INFINITY;
            const int j = act ? (c0 + (r & 3) + 8 * (r >> 2) + 4 * h) : INT_MAX;
            topk_insert(v, id, d, j);   // a survivor that no longer qualifies simply falls off the end
        }
        if (nxt >= 0) commit(buf ^ 1);
        cur = nxt;
        __syncthreads();
    }

    // ---- merge the two partial lists of every query -----------------------------------------------------------------
    {
        const int ql = w * 32 + col;
#pragma unroll
        for (int t = 0; t < TK; ++t) { mval[(ql * 2 + h) * TK + t] = v[t]; midx[(ql * 2 + h) * TK + t] = id[t]; }
    }
    __syncthreads();
    if (tid < 128 && q0 + tid < n) {
        int h0 = 0, h1 = 0;
        const int kk = ctab ? (n < TK ? n : TK) : k;                  // neighbours that exist
        const int kout = ctab ? TK : k;                               // entries written per row
        const int add = ctab ? (int)row0 : 0;
        int *out = idx + (row0 + q0 + tid) * (size_t)kout;
        const float *va = mval + (tid * 2) * TK, *vb = va + TK;
        const int *ia = midx + (tid * 2) * TK, *ib = ia + TK;
        int first = 0;
        for (int o = 0; o < kout; ++o) {
            if (o < kk) {
                const float a = h0 < TK ? va[h0] : -INFINITY, b = h1 < TK ? vb[h1] : -INFINITY;
                const int ja = h0 < TK ? ia[h0] : INT_MAX, jb = h1 < TK ? ib[h1] : INT_MAX;
                const bool takea = (a > b) || (a == b && ja < jb);
                const int j = (takea ? ja : jb) + add;
                if (o == 0) first = j;
                out[o] = j;
                if (takea) ++h0; else ++h1;
            } else out[o] = first;
        }
    }
}

// ------------------------------------------------------------------------------------------------ generic kernel (any C)
typedef float f32x4g __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void knn_generic_kernel(const float *__restrict__ x, const float *__restrict__ xx, int n, int C, int ldC,
                                                         int k, int *__restrict__ idx) {
    extern __shared__ float smem[];
    float *Q = smem;                    // [64][ldC] query features
    float *Cd = Q + 64 * ldC;           // [64][ldC] candidate tile
    float *xxc = Cd + 64 * ldC;         // [64]
    float *mval = xxc + 64;             // [64][4][TK]
    int *midx = (int *)(mval + 64 * 4 * TK);

    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, col = lane & 15, kq = lane >> 4;
    const float *xb = x + (size_t)blockIdx.y * n * C;
    const float *xxb = xx + (size_t)blockIdx.y * n;
    const int q0 = blockIdx.x * 64;
    const int Cpad = (C + 3) & ~3, KS = Cpad >> 2;
    const int nt = (n + 63) >> 6;

    for (int e = tid; e < 64 * Cpad; e += 256) {
        const int r = e / Cpad, c = e - r * Cpad;
        Q[r * ldC + c] = (c < C && q0 + r < n) ? xb[(size_t)(q0 + r) * C + c] : 0.f;
    }
    const int qi = q0 + w * 16 + col;
    const float xxi = qi < n ? xxb[qi] : 0.f;

    float v[TK];
    int id[TK];
#pragma unroll
    for (int t = 0; t < TK; ++t) { v[t] = -INFINITY; id[t] = INT_MAX; }

    int lo = (int)blockIdx.x - 1, hi = (int)blockIdx.x + 1, tile = blockIdx.x;
    for (int s = 0; s < nt; ++s) {
        const int c0 = tile * 64;
        __syncthreads();
        for (int e = tid; e < 64 * Cpad; e += 256) {
            const int r = e / Cpad, c = e - r * Cpad;
            Cd[r * ldC + c] = (c < C && c0 + r < n) ? xb[(size_t)(c0 + r) * C + c] : 0.f;
        }
        if (tid < 64) xxc[tid] = (c0 + tid < n) ? xxb[c0 + tid] : 0.f;
        __syncthreads();
        f32x4g acc[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = (f32x4g){0.f, 0.f, 0.f, 0.f};
        const float *qrow = Q + (w * 16 + col) * ldC + kq;
        const float *crow = Cd + col * ldC + kq;
        for (int ks = 0; ks < KS; ++ks) {
            const float bq = qrow[4 * ks];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(crow[rt * 16 * ldC + 4 * ks], bq, acc[rt], 0, 0, 0);
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cl = rt * 16 + kq * 4 + r;
                const int j = c0 + cl;
                const float d = (2.f * acc[rt][r] - xxc[cl]) - xxi;
                if (j < n && ((d > v[TK - 1]) || (d == v[TK - 1] && j < id[TK - 1]))) topk_insert(v, id, d, j);
            }
        }
        if ((s & 1) == 0) { if (hi < nt) tile = hi++; else tile = lo--; }
        else { if (lo >= 0) tile = lo--; else tile = hi++; }
    }
    {
        const int ql = w * 16 + col;
#pragma unroll
        for (int t = 0; t < TK; ++t) { mval[(ql * 4 + kq) * TK + t] = v[t]; midx[(ql * 4 + kq) * TK + t] = id[t]; }
    }
    __syncthreads();
    if (tid < 64 && q0 + tid < n) {
        int hh[4] = {0, 0, 0, 0};
        int *out = idx + ((size_t)blockIdx.y * n + q0 + tid) * k;
        for (int o = 0; o < k; ++o) {
            int best = -1, bi = 0;
            float bv = 0.f;
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                if (hh[l] >= TK) continue;
                const float cv = mval[(tid * 4 + l) * TK + hh[l]];
                const int ci = midx[(tid * 4 + l) * TK + hh[l]];
                if (best < 0 || cv > bv || (cv == bv && ci < bi)) { best = l; bv = cv; bi = ci; }
            }
            out[o] = bi;
#pragma unroll
            for (int l = 0; l < 4; ++l) if (l == best) hh[l]++;
        }
    }
}

static DevBuf g_xx;   // |x|^2 scratch, grown on demand

extern "C" int scp_knn_topk(const float *x, int32_t B, int32_t n, int32_t C, int32_t k, int32_t *idx, void *stream) {
    if (!x || !idx || B <= 0 || n <= 0 || C <= 0 || k <= 0 || k > TK || k > n) return SCP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int64_t npts = (int64_t)B * n;
    int rc = g_xx.reserve((size_t)npts * sizeof(float));
    if (rc) return rc;
    float *xx = g_xx.as<float>();
    hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)cdiv64(npts, 256)), dim3(256), 0, st, x, npts, C, xx);
    LAUNCH_CHECK();
    const dim3 grid2((n + 127) / 128, B);
    const int *none = nullptr;
    if (C <= 4) hipLaunchKernelGGL(knn_mfma_kernel<2>, grid2, dim3(256), 0, st, x, (const float *)xx, n, C, k, idx, none);
    else if (C == 144) hipLaunchKernelGGL(knn_mfma_kernel<72>, grid2, dim3(256), 0, st, x, (const float *)xx, n, C, k, idx, none);
    else if (C == 192) hipLaunchKernelGGL(knn_mfma_kernel<96>, grid2, dim3(256), 0, st, x, (const float *)xx, n, C, k, idx, none);
    else {
        const int Cpad = (C + 3) & ~3;
        int ldC = Cpad;
        while ((ldC & 3) != 2) ++ldC;   // ldC/2 odd: the 16 candidate rows of a fragment read hit 16 distinct even banks
        const size_t lds = ((size_t)2 * 64 * ldC + 64 + (size_t)64 * 4 * TK * 2) * sizeof(float);
        if (lds > 160 * 1024) return SCP_EINVAL;
        static size_t configured = 0;
        if (lds > configured) {
            HIP_TRY(hipFuncSetAttribute((const void *)knn_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            configured = lds;
        }
        hipLaunchKernelGGL(knn_generic_kernel, dim3((n + 63) / 64, B), dim3(256), lds, st, x, (const float *)xx, n, C, ldC, k, idx);
    }
    LAUNCH_CHECK();
    return SCP_OK;
}


// packed ("varlen") form: x [total_rows][C] with sequences padded to multiples of 512 rows, ctab[2*c] = first row of the sequence owning
// 512-row chunk c, ctab[2*c+1] = its real length; idx [total_rows][20] GLOBAL row indices (rows beyond a sequence's length are not written).
extern "C" SCP_API int scp_knn_topk_packed(const float *x, const int32_t *ctab, int32_t total_rows, int32_t C, int32_t *idx, void *stream) {
    if (!x || !ctab || !idx || total_rows <= 0 || (total_rows & 511) || (C != 144 && C != 192 && C > 4)) return SCP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int rc = g_xx.reserve((size_t)total_rows * sizeof(float));
    if (rc) return rc;
    float *xx = g_xx.as<float>();
    hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)cdiv64(total_rows, 256)), dim3(256), 0, st, x, (int64_t)total_rows, C, xx);
    LAUNCH_CHECK();
    const dim3 grid(total_rows / 128, 1);
    if (C <= 4) hipLaunchKernelGGL(knn_mfma_kernel<2>, grid, dim3(256), 0, st, x, (const float *)xx, 0, C, TK, idx, ctab);
    else if (C == 144) hipLaunchKernelGGL(knn_mfma_kernel<72>, grid, dim3(256), 0, st, x, (const float *)xx, 0, C, TK, idx, ctab);
    else hipLaunchKernelGGL(knn_mfma_kernel<96>, grid, dim3(256), 0, st, x, (const float *)xx, 0, C, TK, idx, ctab);
    LAUNCH_CHECK();
    return SCP_OK;
}
