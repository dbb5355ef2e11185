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
// Layout (specialised kernels, K = 4 / 144 / 192 features):
//   workgroup = 4 waves = 128 queries, a wave owns 32 queries; B operand = the wave's query features in registers, loaded
//   once; A operand = 32-candidate tiles staged in LDS by LDS-DMA, double buffered (the DMA of the next stage runs under the
//   MFMAs of the current one: one barrier per stage; a stage is 32 candidates, or 512 for K = 4).
//   D[cand][query]: a lane ends up with 16 candidates of ONE query and keeps a private sorted top-20 (value, index)
//   list in registers; the two lanes of a query share their pruning threshold and merge through LDS at the end.
//   Tiles are visited outwards from the queries' own position: octree siblings are Morton neighbours, so the
//   threshold tightens immediately and later insertions are rare.
#include <float.h>
#include <stdlib.h>
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

// ------------------------------------------------------------------------------------------------ specialised kernels
// Candidate tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no VALU, no branches around the
// loads) in their natural row order; the XOR swizzle that makes the fragment reads conflict-free is applied to the per-lane
// SOURCE address (the LDS image of one DMA instruction is lane-linear).
typedef __attribute__((address_space(3))) void *knn_lds_ptr_t;
typedef const __attribute__((address_space(1))) void *knn_glb_ptr_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---- sorted top-20 lists of the specialised kernels: ONE 64-bit key per entry, kept as a DOUBLE ----------------------------------
// key = the double whose high word is the bit pattern of the float d and whose low word is the index j (d < 0) or ~j (d >= 0).
// IEEE doubles are sign-magnitude like floats, and a float's exponent and mantissa occupy the high bits of the double's, so the
// order of the doubles is the order the lists want (larger d = larger key; equal d: lower j = larger key) for every float d:
// +-0, denormals (f64 denormals are never flushed) and +-inf, whose high word 0x7F800000 is a FINITE double; only a NaN d with
// mantissa bits 0x700000 set would be a double NaN, and d is never NaN for finite inputs.  An insertion step is then
// v_max_f64 + v_min_f64 (full rate on gfx950: 2 x 4 cycles) instead of v_cmp_gt_u64 + four v_cndmask (5 - 6 x 4 cycles): the
// selection is VALU-bound (two waves per SIMD spend ~60 % of their cycles issuing VALU work, most of it insertions).
// Empty slots hold KEY_EMPTY = (-inf, no index); KEY_NONE (the double -inf) is below every key, KEY_EMPTY included.
// Measured after this and dropped (A/B on one box, lists identical): pass 1 on v_pk_mul_f32 / v_pk_fma_f32 (two candidates per
// instruction) +2 %, its mask built with v_addc carry-in (3 instead of 4 instructions per candidate) +0 %, the per-lane acc[r] of
// pass 2 as a binary select tree (19 instead of 32 instructions) +6 % - fewer VALU instructions no longer buy time here.
typedef double kkey_t;
#define KEY_EMPTY __longlong_as_double((long long)0xFF800000FFFFFFFFull)
#define KEY_NONE __longlong_as_double((long long)0xFFF0000000000000ull)
__device__ __forceinline__ kkey_t knn_key(float d, int j) {
    const int hi = __float_as_int(d);
    return __hiloint2double(hi, j ^ ~(hi >> 31));
}
__device__ __forceinline__ float knn_key_val(kkey_t k) { return __int_as_float(__double2hiint(k)); }
__device__ __forceinline__ int knn_key_idx(kkey_t k) {
    const int hi = __double2hiint(k), lo = __double2loint(k);
    return (hi == (int)0xFF800000 && lo == -1) ? INT_MAX : (lo ^ ~(hi >> 31));
}

// (Measured and dropped, round 4: the list without the chain of 20 dependent v_min_f64 - new[t] = max(key[t], min(key[t - 1], kc)), every slot from the OLD
// neighbours, bottom up in place, identical lists - 4.62 / 5.69 against 4.60 - 4.64 / 5.67 - 5.68 ms per 70 windows of 8 192 at C = 144 / 192: not the insertion's latency either.)
__device__ __forceinline__ void key_insert(kkey_t (&key)[TK], kkey_t kc) {
#pragma unroll
    for (int t = 0; t < TK; ++t) {
        // inline asm: the compiler's fmax / fmin would canonicalise both operands first (two more instructions per step)
        // (the maximum is written in place: separate result registers made the compiler copy 40 registers back per insertion)
        kkey_t rest;
        asm("v_min_f64 %0, %1, %2" : "=v"(rest) : "v"(kc), "v"(key[t]));
        asm("v_max_f64 %0, %1, %0" : "+v"(key[t]) : "v"(kc));
        kc = rest;
    }
}

// ---- shared tail: merge the two partial lists of every query (lane halves h = 0 / 1) and write the indices -------------------
__device__ __forceinline__ void knn_merge_write(const kkey_t (&key)[TK], float *mval, int *midx, int tid, int w, int col, int h,
                                                int q0, int n, int k, size_t row0, const int *ctab, int *idx) {
    {
        const int ql = w * 32 + col;
#pragma unroll
        for (int t = 0; t < TK; ++t) { mval[(ql * 2 + h) * TK + t] = knn_key_val(key[t]); midx[(ql * 2 + h) * TK + t] = knn_key_idx(key[t]); }
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

// |x|^2 of a row beyond its sequence's length (the candidate tiles clamp such rows to the last real one): DMA'd in place of xx[c]
__device__ const float knn_inf = INFINITY;

// ---- selection step shared by both kernels: this lane holds candidates row = (r&3) + 8(r>>2) + 4h of a 32-tile for its query ---
// `acc[r]` = x_i . x_j (SCALED: times the two row scales; `usc2` = 2 / scale of the query, `sis` = 1 / scale of the candidates).
// With a = 2 x_i.x_j - |x_j|^2 the (negated squared) distance is d = fl(a - |x_i|^2).
// Pass 1 (branch-free, 4-5 VALU per candidate) never forms d: fl() is monotone, so "d < bound" is implied by "a < cut" for a
// cut placed a few ulps below bound + |x_i|^2 (derivation at `cut`); the pass builds a bit mask of the candidates at or above
// the cut and keeps the a of the last one.  Candidates the cut lets through although they do not qualify (a few ulps, or exact
// ties with the bound) are harmless: the insertion compares full (value, index) keys and they fall off the end of the list.
// Pass 2: lanes pop their survivors one at a time (the wavefront pays for max-over-lanes survivors, not for all 16 slots).
// value of the query's other lane (lane ^ 32) by one v_permlane32_swap (the half exchange of gfx950) instead of a ds_bpermute round trip
__device__ __forceinline__ unsigned knn_xchg32(unsigned x, int h) {
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return h ? r[0] : r[1];
}

template <bool SCALED>
__device__ __forceinline__ void knn_select(const f32x16 &acc, const float *sxx, const float *sis, float usc2, float xxi, int c0, int n, int h,
                                           kkey_t (&key)[TK], float thr0) {
    // Pruning bound.  The two lanes of a query (h = 0 / 1) keep separate sorted lists a, b over disjoint candidates.  a[i] and
    // b[18 - i] bound the 20th best of their union from below (i + 1 entries of a and 19 - i entries of b are at least
    // min(a[i], b[18 - i])), as do a[19] and b[19] alone; each lane evaluates four such pairs (values only: the sortable high
    // words) and the pair of lanes takes the maximum - about the 21st best seen so far instead of the ~38th that
    // max(a[19], b[19]) gives.  thr0 is the caller's a-priori bound (a distance that at least 20 candidates are known to beat;
    // -inf when there is none).
    static_assert(TK == 20, "pair table");
    float sb = knn_key_val(key[TK - 1]);
    {
        const float p3 = __uint_as_float(knn_xchg32(__float_as_uint(knn_key_val(key[3])), h)), p6 = __uint_as_float(knn_xchg32(__float_as_uint(knn_key_val(key[6])), h)),
                    p9 = __uint_as_float(knn_xchg32(__float_as_uint(knn_key_val(key[9])), h));
        sb = fmaxf(sb, fminf(knn_key_val(key[15]), p3));
        sb = fmaxf(sb, fminf(knn_key_val(key[12]), p6));
        sb = fmaxf(sb, fminf(knn_key_val(key[9]), p9));
        sb = fmaxf(sb, __uint_as_float(knn_xchg32(__float_as_uint(sb), h)));
    }
    const float thr = fmaxf(sb, thr0);
    // cut: with m = max(|thr|, |xxi|), s = fl(thr + xxi) and cut = fl(s - 2^-21 m) satisfy cut <= thr + xxi - 2^-22 m (both
    // roundings are at most 2^-23 m).  a < cut then gives a - xxi < thr - 2^-22 |thr|, which lies below the float preceding thr,
    // so fl(a - xxi) < thr: the candidate cannot be among the 20 best.  thr = -inf (list not full yet): cut = -inf, all pass.
    const float cut = (thr + xxi) - 0x1p-21f * fmaxf(fabsf(thr), fabsf(xxi));
    unsigned pend = 0;                                                // candidate r -> bit 15 - r
    float asel = 0.f;                                                 // a of the survivor with the lowest bit
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 xj = *(const f32x4 *)(sxx + 8 * g + 4 * h);      // candidates (r & 3) + 8 g + 4 h, r & 3 = 0..3
        f32x4 sj = {1.f, 1.f, 1.f, 1.f};
        if (SCALED) sj = *(const f32x4 *)(sis + 8 * g + 4 * h);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = 4 * g + u;
            // scales are powers of two: (acc * usc2) * sj is exact, one rounding in the fma
            const float a = SCALED ? fmaf(acc[r] * usc2, sj[u], -xj[u]) : fmaf(2.f, acc[r], -xj[u]);
            // rows beyond the sequence's length carry |x|^2 = +inf (knn_inf below): a = -inf, which only passes while the list is
            // not full (cut = -inf) and is then the lowest real key - no bounds test here
            const bool pass = a >= cut;
            pend = (pend << 1) | (pass ? 1u : 0u);
            asel = pass ? a : asel;
        }
    }
    if (!__any(pend != 0u)) return;
    {   // first survivor of every lane: its a is at hand
        const bool act = pend != 0u;
        const int r = act ? 16 - __ffs(pend) : 0;
        pend &= pend - 1u;
        const int j = c0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        key_insert(key, act ? knn_key(asel - xxi, j) : KEY_NONE);
    }
    while (__any(pend != 0u)) {   // further survivors (wave-uniform loop): recompute a of slot r, same arithmetic -> same value
        const bool act = pend != 0u;
        const int r = act ? 16 - __ffs(pend) : 0;
        pend &= pend - 1u;
        float t = acc[0];
#pragma unroll
        for (int rr = 1; rr < 16; ++rr) t = (rr == r) ? acc[rr] : t;
        const int cl = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float a = SCALED ? fmaf(t * usc2, sis[cl], -sxx[cl]) : fmaf(2.f, t, -sxx[cl]);
        key_insert(key, act ? knn_key(a - xxi, c0 + cl) : KEY_NONE);
    }
}

// ---- exact fp32 kernel: v_mfma_f32_32x32x2_f32, k-ordered chain (bit-identical distance values to PyTorch-CPU) ----------------
// rows of K floats; chunk q of row c at position q ^ (c & 15) (K = 192) / q ^ ((c >> 2) & 3) (K = 144); a lane reads four
// consecutive features and feeds two of them (k = 4g + h, 4g + 2 + h) to two MFMA steps: lane half h still supplies feature
// 2 * step + h, the k order of the chain is untouched.
// BOX (positions, K = 4): bbox[g] = (min x, min y, min z, max |p|^2, max x, max y, max z, -) of the 32 points of group g
// (bbox32_kernel).  Points arrive in Morton order, so most 32-candidate tiles of a window lie far from a wavefront's 32 queries: a tile
// whose box is farther from the queries' box than every lane's 20th best (plus a margin far above the rounding of the distances)
// cannot pass pass 1 for any lane and is skipped - products, selection and all; the lists do not change.  Position search of an L16
// frame 1.81 -> 1.33 ms, 1.17 ms with the registers held to four waves per SIMD.  (Measured and dropped: never requesting a whole
// 512-candidate stage whose box is that far from the box of the workgroup's 128 queries - boxes of 512 consecutive Morton points are
// too loose to separate: no gain.)
template <int KS, int NSUB, bool BOX = false>   // MFMA k-steps = C / 2; 32 * NSUB candidates per LDS stage
__global__ __launch_bounds__(256, BOX ? 4 : 2) void knn_mfma_kernel(const float *__restrict__ x, const float *__restrict__ xx, int n, int C, int k,
                                                         int *__restrict__ idx, const int *__restrict__ ctab /* packed mode: per 512-row chunk (seq base row, seq n) */,
                                                         const float *__restrict__ thr0, const float *__restrict__ bbox = nullptr) {
    constexpr int K = 2 * KS;                       // feature count (multiple of 4) = row length in floats, rows are 16-byte aligned
    constexpr int R = K / 4;                        // 16-byte chunks per row
    constexpr int SC = 32 * NSUB;                   // candidates per stage
    constexpr int STAGE_F = SC * K;                 // floats per stage
    constexpr int NDMA = (SC * R + 63) / 64;        // 1 KiB DMA instructions per stage
    constexpr int NDMA_W = (NDMA + 3) / 4;          // per wave
    constexpr int MERGE_F = 2 * 128 * 2 * TK;
    constexpr int TILES_F = 2 * STAGE_F + 2 * SC + (BOX ? 2 * NSUB * 8 : 0);   // two stages + their |x|^2 (+ their tiles' boxes)
    constexpr int POOL_F = TILES_F > MERGE_F ? TILES_F : MERGE_F;
    static_assert((SC * R) % 64 == 0, "stage is a whole number of DMA instructions");
    __shared__ __attribute__((aligned(1024))) float pool[POOL_F];              // stages during the sweep, merge lists after it
    float *txx = pool + 2 * STAGE_F;                // [2][SC]
    float *tbb = txx + 2 * SC;                      // BOX: [2][NSUB][8]

    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // dense mode: blockIdx.y = batch item with n points.  packed mode: sequences padded to x512 rows lie back to back; the chunk
    // table says which sequence (first row, real length) the 128 query rows of this workgroup belong to; indices come out GLOBAL
    // and always TK per row (rows of sequences shorter than TK repeat their nearest neighbour, harmless under the max-pool).
    size_t row0 = (size_t)blockIdx.y * n;
    const int bx = blockIdx.x;   // (an XCD-contiguous order was tried: the uneven window lengths cost more in balance than L2 locality gains)
    int q0 = bx * 128;
    if (ctab) {
        const int ch = (bx * 128) >> 9;
        row0 = (size_t)ctab[2 * ch];
        n = ctab[2 * ch + 1];
        q0 = bx * 128 - (int)row0;
        if (q0 >= n) return;   // workgroup entirely inside the padding of its sequence
    }
    const float *xb = x + row0 * K;
    const float *xxb = xx + row0;
    const int qi = q0 + w * 32 + col;
    const int nt = (n + SC - 1) / SC;               // stages
    // BOX: groups of 32 rows are numbered per sequence (packed: row0 is a multiple of 512) / per batch item (dense: (n + 31) / 32 each)
    const int ngrp = (n + 31) >> 5;
    const float *bbb = BOX ? bbox + (ctab ? (row0 >> 5) : (size_t)blockIdx.y * ngrp) * 8 : nullptr;
    f32x4 qb0 = {0.f, 0.f, 0.f, 0.f}, qb1 = qb0;
    if (BOX) {
        int g = (q0 >> 5) + w;
        g = g < ngrp ? g : ngrp - 1;
        qb0 = *(const f32x4 *)(bbb + 8 * g);
        qb1 = *(const f32x4 *)(bbb + 8 * g + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {               // wave-uniform: keep them in scalar registers
            qb0[c] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(qb0[c])));
            qb1[c] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(qb1[c])));
        }
    }

    float qf[KS];                                   // query fragment: features 2*s + h
    {
        const int qc = qi < n ? qi : n - 1;         // clamped row: lanes beyond n compute garbage that is never written
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[s] = xb[(size_t)qc * K + 2 * s + h];
    }
    const float xxi = (qi < n) ? xxb[qi] : 0.f;
    const float thr0v = (thr0 && qi < n) ? thr0[row0 + qi] : -INFINITY;

    kkey_t key[TK];
#pragma unroll
    for (int t = 0; t < TK; ++t) key[t] = KEY_EMPTY;

    auto issue = [&](int t, int buf) {   // candidates [t * SC, t * SC + SC) -> stage buf (rows beyond n: clamped, masked later)
        const int c0 = t * SC;
        char *sb = (char *)(pool + buf * STAGE_F);
        // opaque copy of the lane id: keeps the per-DMA address arithmetic INSIDE the loop - hoisted, those loop invariants get
        // spilled next to the query fragment, and every reload waits vmcnt(0), i.e. for the previous DMA to land
        int ln;
        asm volatile("v_mov_b32 %0, %1" : "=v"(ln) : "v"(lane));
#pragma unroll
        for (int j = 0; j < NDMA_W; ++j) {
            const int ii = w + 4 * j;
            if (ii < NDMA) {             // wave-uniform
                const int ci = ii * 64 + ln;      // this lane fills chunk ci of the stage = row ci / R, position ci % R
                const int r = ci / R, p = ci - r * R;
                int q = p;
                if (R % 16 == 0) q = p ^ (r & 15);
                else if (R % 4 == 0 && R > 4) q = p ^ ((r >> 2) & 3);
                int c = c0 + r;
                c = c < n ? c : n - 1;
                __builtin_amdgcn_global_load_lds((knn_glb_ptr_t)(xb + (size_t)c * K + 4 * q), (knn_lds_ptr_t)(sb + ii * 1024), 16, 0, 0);
            }
        }
        if (BOX && w == 3 && ln < 2 * NSUB) {   // the stage's NSUB boxes: 2 NSUB pieces of 16 bytes (groups beyond the sequence: the last one)
            int g = t * NSUB + (ln >> 1);
            g = g < ngrp ? g : ngrp - 1;
            __builtin_amdgcn_global_load_lds((knn_glb_ptr_t)(bbb + 8 * g + 4 * (ln & 1)), (knn_lds_ptr_t)((char *)(tbb + buf * NSUB * 8)), 16, 0, 0);
        }
#pragma unroll
        for (int e0 = 0; e0 < SC; e0 += 256) {
            if (e0 + w * 64 < SC) {      // wave-uniform
                const int e = e0 + w * 64 + ln;
                if (e < SC) {
                    const int c = c0 + e;
                    __builtin_amdgcn_global_load_lds((knn_glb_ptr_t)(c < n ? xxb + c : &knn_inf), (knn_lds_ptr_t)((char *)(txx + buf * SC) + (e0 + w * 64) * 4), 4, 0, 0);
                }
            }
        }
    };

    const int swz = (R % 16 == 0) ? (col & 15) : ((R % 4 == 0 && R > 4) ? ((col >> 2) & 3) : 0);

    // outward stage order starting at the stage that holds this workgroup's own queries
    const int own = q0 / SC;
    int lo = own - 1, hi = own + 1, cur = own < nt ? own : nt - 1;
    if (own >= nt) { lo = nt - 2; hi = nt; }
    int step = 0;
    auto next_stage = [&]() {            // -1 when every stage has been handed out
        int t = -1;
        if ((step & 1) == 0) { if (hi < nt) t = hi++; else if (lo >= 0) t = lo--; }
        else { if (lo >= 0) t = lo--; else if (hi < nt) t = hi++; }
        ++step;
        return t;
    };
    issue(cur, 0);
    float far2 = INFINITY;                           // BOX: this wavefront's bound (squared distance); +inf while a list is not full

    for (int s = 0; cur >= 0; ++s) {
        const int buf = s & 1;
        SCP_WAIT_DMA(0);
        __syncthreads();   // stage s has landed and every wave is done with the other stage
        const int nxt = next_stage();
        if (nxt >= 0) issue(nxt, buf ^ 1);
        const float *stage = pool + buf * STAGE_F;
        const float *sxx = txx + buf * SC;
#pragma unroll 1
        for (int sub = 0; sub < NSUB; ++sub) {
            if (BOX) {
                const f32x4 c0v = *(const f32x4 *)(tbb + (buf * NSUB + sub) * 8), c1v = *(const f32x4 *)(tbb + (buf * NSUB + sub) * 8 + 4);
                const float gx = fmaxf(fmaxf(c0v[0] - qb1[0], qb0[0] - c1v[0]), 0.f), gy = fmaxf(fmaxf(c0v[1] - qb1[1], qb0[1] - c1v[1]), 0.f),
                            gz = fmaxf(fmaxf(c0v[2] - qb1[2], qb0[2] - c1v[2]), 0.f);
                const float lb2 = gx * gx + gy * gy + gz * gz;
                if (lb2 > far2 + 0x1p-14f * (qb0[3] + c0v[3])) continue;     // the same value in every lane: a uniform branch
            }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float *arow = stage + (sub * 32 + col) * K;
#pragma unroll
            for (int g = 0; g < R; ++g) {
                const f32x4 a = *(const f32x4 *)(arow + 4 * (g ^ swz));
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a[1] : a[0], qf[2 * g], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a[3] : a[2], qf[2 * g + 1], acc, 0, 0, 0);
            }
            knn_select<false>(acc, sxx + sub * 32, nullptr, 1.f, xxi, cur * SC + sub * 32, n, h, key, thr0v);
        }
        if (BOX) {
            // the loosest bound of the wavefront: every lane's own 20th best (<= the pair's), refreshed once per stage (bounds only tighten)
            float tmin = knn_key_val(key[TK - 1]);
            for (int o = 32; o > 0; o >>= 1) tmin = fminf(tmin, __shfl_xor(tmin, o));
            // d = -|x - y|^2 is computed to within ~2^-20 (|x|^2 + |y|^2); 2^-14 of the larger norms is far above that and above pass 1's own
            // 2^-21 guard, and costs no skips (tile distances are orders of magnitude above it)
            far2 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(-tmin)));   // +inf while some list is not full: nothing is skipped
        }
        cur = nxt;
    }
    __syncthreads();   // everybody is done with the stages: the pool becomes the merge area
    knn_merge_write(key, pool, (int *)(pool + 128 * 2 * TK), tid, w, col, h, q0, n, k, row0, ctab, idx);
}

// positions (C <= 4 features) padded to rows of 4 floats for the K = 4 specialisation
__global__ __launch_bounds__(256) void pad4_kernel(const float *__restrict__ x, int64_t npts, int C, float *__restrict__ x4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npts) return;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) o[c] = x[i * C + c];
    *(f32x4 *)(x4 + 4 * i) = o;
}

// boxes of the groups of 32 consecutive points of every item (rows [b * n, b * n + n) of x4; the last group of an item may be short)
__global__ __launch_bounds__(256) void bbox32_kernel(const float *__restrict__ x4, int n, int B, float *__restrict__ bb) {
    const int ngrp = (n + 31) >> 5;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)ngrp * B) return;
    const int b = (int)(i / ngrp), g = (int)(i - (int64_t)b * ngrp);
    const int r1 = (32 * g + 32 < n) ? 32 * g + 32 : n;
    f32x4 lo = {INFINITY, INFINITY, INFINITY, 0.f}, hi = {-INFINITY, -INFINITY, -INFINITY, 0.f};
    for (int r = 32 * g; r < r1; ++r) {
        const f32x4 p = *(const f32x4 *)(x4 + ((int64_t)b * n + r) * 4);
#pragma unroll
        for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], p[c]); hi[c] = fmaxf(hi[c], p[c]); }
        lo[3] = fmaxf(lo[3], p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    }
    *(f32x4 *)(bb + 8 * i) = lo;
    *(f32x4 *)(bb + 8 * i + 4) = hi;
}

// ---- fp32-faithful kernel on f16 MFMA ("f16x3") -----------------------------------------------------------------------------
// split2_kernel scales every row by its own power of two (largest |feature| into [2^13, 2^14): exact, keeps f16 clear of
// overflow and of its subnormal range for everything that matters) and writes it as two f16 terms x ~= xa + xb (11 + 11
// significant bits, residual < 2^-22 |x|).  A dot product keeps the three partial products down to 2^-11 relative size,
//     x.y ~= xb.ya + xa.yb + xa.ya,          dropped: xb.yb < 2^-22 |x||y|,
// accumulated in fp32 by v_mfma_f32_32x32x16_f16, small terms first, then multiplied by the two rows' inverse scales (powers
// of two, exact).  Three f16 MFMAs cover 16 features in 96 cycles where the fp32 MFMA needs 512.  The distance values agree
// with the fp32 chain to about its own rounding error (worst case 7e-7 of sum |x_i y_i|; the k-ordered fp32 chain itself
// carries up to K 2^-24 of it), so the two kernels select the same neighbours except among candidates whose distances are
// closer than that - the same class of difference as the ordering of exact ties (measured: tests/test_gpu_model.py).
// SCP_KNN=f32 selects the exact kernel instead.
// Data: planes [row][2][K] f16 = rows of 4K bytes; swizzle as in the fp32 kernel (K = 192: q ^ (c & 15), K = 144: q ^ ((c >> 2) & 3)).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int K>
__global__ __launch_bounds__(256, 2) void knn_f16x3_kernel(const _Float16 *__restrict__ planes, const float *__restrict__ xx,
                                                          const float *__restrict__ inv_scale, int n, int k, int *__restrict__ idx,
                                                          const int *__restrict__ ctab, const float *__restrict__ thr0) {
    constexpr int RB = 4 * K;                       // bytes per row
    constexpr int R = RB / 16;                      // chunks per row (48 / 36)
    constexpr int NC = K / 16;                      // k-chunks of 16 features
    constexpr int STAGE_B = 32 * RB;                // bytes per stage (32 candidates)
    constexpr int NDMA = STAGE_B / 1024;            // 1 KiB DMA instructions per stage (24 / 18)
    constexpr int NDMA_W = (NDMA + 3) / 4;
    constexpr int MERGE_B = 2 * 128 * 2 * TK * 4;
    constexpr int NST = 3;                          // LDS stages: candidate tiles are requested two sweep steps ahead
    constexpr int TILES_B = NST * STAGE_B + NST * 64 * 4;
    constexpr int POOL_B = TILES_B > MERGE_B ? TILES_B : MERGE_B;
    static_assert(STAGE_B % 1024 == 0, "layout");
    __shared__ __attribute__((aligned(1024))) char pool[POOL_B];
    float *txx = (float *)(pool + NST * STAGE_B);   // [NST][64]: |x|^2 of the 32 candidates, then their inverse scales

    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    size_t row0 = (size_t)blockIdx.y * n;
    const int bx = blockIdx.x;   // (an XCD-contiguous order was tried: the uneven window lengths cost more in balance than L2 locality gains)
    int q0 = bx * 128;
    if (ctab) {
        const int ch = (bx * 128) >> 9;
        row0 = (size_t)ctab[2 * ch];
        n = ctab[2 * ch + 1];
        q0 = bx * 128 - (int)row0;
        if (q0 >= n) return;
    }
    const char *pb = (const char *)planes + row0 * RB;
    const float *xxb = xx + row0, *isb = inv_scale + row0;
    const int qi = q0 + w * 32 + col;
    const int nt = (n + 31) >> 5;

    // query fragments (B operand): plane p, chunk c: features 16c + 8h .. + 7
    f16x8 qa[NC], qb[NC];
    const int qc = qi < n ? qi : n - 1;
    {
        const char *src = pb + (size_t)qc * RB + 16 * h;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            qa[c] = *(const f16x8 *)(src + 32 * c);
            qb[c] = *(const f16x8 *)(src + 2 * K + 32 * c);
        }
    }
    const float xxi = (qi < n) ? xxb[qi] : 0.f;
    const float thr0v = (thr0 && qi < n) ? thr0[row0 + qi] : -INFINITY;
    const float isq = isb[qc];                      // 1 / scale of the query row (a power of two)

    kkey_t key[TK];
#pragma unroll
    for (int t = 0; t < TK; ++t) key[t] = KEY_EMPTY;

    auto issue = [&](int t, int buf) {
        const int c0 = t * 32;
        char *sb = pool + buf * STAGE_B;
        int ln;
        asm volatile("v_mov_b32 %0, %1" : "=v"(ln) : "v"(lane));   // see knn_mfma_kernel
#pragma unroll
        for (int j = 0; j < NDMA_W; ++j) {
            const int ii = w + 4 * j;
            if (ii < NDMA) {
                const int ci = ii * 64 + ln;
                const int r = ci / R, p = ci - r * R;
                const int q = (R % 16 == 0) ? (p ^ (r & 15)) : (p ^ ((r >> 2) & 3));
                int c = c0 + r;
                c = c < n ? c : n - 1;
                __builtin_amdgcn_global_load_lds((knn_glb_ptr_t)(pb + (size_t)c * RB + 16 * q), (knn_lds_ptr_t)(sb + ii * 1024), 16, 0, 0);
            }
        }
        if (w == 0) {   // lanes 0..31: |x|^2, lanes 32..63: inverse scale of candidate lane & 31
            int c = c0 + (ln & 31);
            const bool real = c < n;
            c = real ? c : n - 1;
            const float *src = ln < 32 ? (real ? xxb + c : &knn_inf) : isb + c;
            __builtin_amdgcn_global_load_lds((knn_glb_ptr_t)src, (knn_lds_ptr_t)((char *)(txx + buf * 64)), 4, 0, 0);
        }
    };

    const int swz = (R % 16 == 0) ? (col & 15) : ((col >> 2) & 3);
    // sweep order: own tile first, then alternately outwards (the pruning bound tightens fastest on nearby rows)
    const int own = q0 >> 5;
    int lo = own - 1, hi = own + 1, cur = own < nt ? own : nt - 1;
    if (own >= nt) { lo = nt - 2; hi = nt; }
    int step = 0;
    auto next_tile = [&]() {
        int t;
        if ((step & 1) == 0) { if (hi < nt) t = hi++; else t = lo--; }
        else { if (lo >= 0) t = lo--; else t = hi++; }
        ++step;
        return t;
    };
    // DMA instructions this wave issues per stage: a stage has landed when at most that many (the next stage's) are in flight
    const int per_stage = (NDMA - w + 3) / 4 + (w == 0 ? 1 : 0);
    issue(cur, 0);
    int cur1 = -1;
    if (nt > 1) { cur1 = next_tile(); issue(cur1, 1); }

    for (int s = 0; s < nt; ++s) {
        const int buf = s % NST;
        // stage s has landed and every wave is done with stage s - 1, whose buffer is refilled now (raw barrier: __syncthreads()
        // would drain the younger stage's DMAs too, scp_internal.h)
        if (s + 1 < nt) {
            if (per_stage >= 7) SCP_BARRIER_DMA(7); else if (per_stage == 6) SCP_BARRIER_DMA(6);
            else if (per_stage == 5) SCP_BARRIER_DMA(5); else SCP_BARRIER_DMA(4);
        } else SCP_BARRIER_DMA(0);
        int nxt = -1;
        if (s + 2 < nt) { nxt = next_tile(); issue(nxt, (s + 2) % NST); }
        const char *arow = pool + buf * STAGE_B + col * RB;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // candidate fragments (A operand): chunk index inside the row = plane * (K / 8) + 2c + h; the reads of chunk c + 1 are issued
        // in front of the products of chunk c (left to itself hipcc reads, waits, multiplies: the block then runs at the LDS latency)
        f16x8 ca = *(const f16x8 *)(arow + 16 * ((0 + h) ^ swz));
        f16x8 cb = *(const f16x8 *)(arow + 16 * ((K / 8 + h) ^ swz));
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            f16x8 na = ca, nb = cb;
            if (c + 1 < NC) {
                na = *(const f16x8 *)(arow + 16 * ((2 * (c + 1) + h) ^ swz));
                nb = *(const f16x8 *)(arow + 16 * ((K / 8 + 2 * (c + 1) + h) ^ swz));
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cb, qa[c], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ca, qb[c], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ca, qa[c], acc, 0, 0, 0);
            ca = na; cb = nb;
            // pin the interleave: the two fragment reads of the next chunk, then the three products of this one
            if (c + 1 < NC) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        }
        // un-scaling (acc / s_query / s_candidate, powers of two: exact) happens inside the selection
        const float *sxx = txx + buf * 64;
        knn_select<true>(acc, sxx, sxx + 32, 2.f * isq, xxi, cur * 32, n, h, key, thr0v);
        cur = cur1;
        cur1 = nxt;
    }
    __syncthreads();
    knn_merge_write(key, (float *)pool, (int *)(pool + 128 * 2 * TK * 4), tid, w, col, h, q0, n, k, row0, ctab, idx);
}

// ---- 256-query workgroups on an XCD-affine schedule (packed mode) ------------------------------------------------------------
// The 128-query kernel above streams a window's whole candidate set (8192 rows x 4K bytes = 6.3 MB at K = 192) through every one of
// its 64 workgroups, and consecutive workgroups land on different XCDs: every XCD's 4 MB L2 sees ~8 windows at once and the tiles
// come back from the Infinity Cache - 24 GB of fabric traffic per launch for 0.5 GB of data (profiles/r1z_pmc_traffic.json).
// Here (a) a workgroup is 8 waves = 256 queries sharing each candidate tile (half the tile fills per query), (b) workgroups are
// dealt to XCDs by a table (knn_sched_kernel): the blocks of a launch are cut into 8 runs of equal work, run x is executed by the
// blocks with blockIdx % 8 == x (observed round-robin placement: a speed assumption only), in window order, so an XCD's 32 CUs
// work on ONE window at a time, and (c) after the tiles holding its own queries a workgroup sweeps the window in ABSOLUTE tile
// order, so the co-resident workgroups of a window ask for the same tile at about the same time and L2 serves all but the first.
// (d) one workgroup barrier per GROUP of tiles instead of per tile (see the kernel).
// Cycle stamps (DBG build, K = 192, 256-tile sweeps; per tile and wave): product block 2,300 with the SIMD's other wave multiplying
// too, 2,100 alone on the SIMD (36 MFMAs = 1,152: the block waits for its LDS fragment reads, two reads then three MFMAs per
// 16 features at 256 VGPRs); selection 3,500 / 3,050; barrier + DMA issue 2,100 / 1,360.  I.e. a wave is latency-bound in every
// phase and two waves per SIMD nearly double the throughput.  Measured and dropped on this kernel: waves 4-7 half a step behind
// waves 0-3 (deferred selection; per-tile, per-group and shifted barriers): no gain; survivors appended to per-lane LDS buffers and
// inserted in batches (16 flushes and 184 insertion rounds per sweep instead of ~500 rounds): the per-tile mask / append code hipcc
// produces costs more than the rounds it saves (select 4,300 cycles per tile); a second accumulator for the odd chunks (no gain: the
// block is not paced by the accumulation chain; it also reorders the fp32 sum and 1 % of the lists change); without any fragment
// read the 36 products of a tile still take ~1,900 cycles for a wave alone on its SIMD.  What did pay: no bounds test per candidate
// (rows beyond a sequence's length get |x|^2 = +inf by DMA: three instructions per candidate less and 16 - 100 fewer VGPRs in every
// kernel that shares knn_select), the partner exchange by v_permlane32_swap instead of ds_bpermute: 7 - 9 % per search.  Queries dealt
// to the waves round-robin instead of in runs of 32 (to even out the waves at the barriers): barrier time -12 %, selection +22 % (a wave
// pays for its busiest lane, and neighbouring queries are busy together), 4 % slower overall.
// Distance arithmetic, selection and the order of insertions per lane are those of knn_f16x3_kernel: identical neighbour lists.
struct KnnWg { int32_t row0, n, q0, pad; };

// tab: [nslots][8] entries for (slot, blockIdx % 8), then `nspill` overflow entries that take whatever an XCD's run holds beyond
// nslots blocks (runs are cut by WORK, so a launch with many short windows can put more blocks into one run than any useful
// nslots; the spill region has room for every block of the launch, so nothing is ever dropped).  Entries left zero are empty.
// nsplit > 1 (short launches, round 5): every 256-query block becomes nsplit entries that sweep one nsplit-th of the candidate tiles each
// (KnnWg.pad = which); their partial lists are merged by knn_merge_split_kernel.
__global__ __launch_bounds__(1024) void knn_sched_kernel(const int *__restrict__ ctab, int nchunks, KnnWg *__restrict__ tab, int nslots, int nspill, int nsplit) {
    // sequences = chunks that start one (base == own first row); work of a 256-query block ~ n (its sweep length)
    __shared__ int seq_base[2048], seq_n[2048], seq_blk0[2049];
    __shared__ unsigned long long seq_w0[2049];
    __shared__ int nseq_s, first_blk[9], spill_s;
    __shared__ unsigned long long wtot_s;
    const int tid = threadIdx.x;
    if (tid == 0) { nseq_s = 0; spill_s = 0; }
    if (tid < 9) first_blk[tid] = INT_MAX;
    __syncthreads();
    if (tid < 64) {   // one wavefront compacts the sequence starts in chunk order
        int cnt = 0;
        for (int g = 0; g < nchunks; g += 64) {
            const int c = g + tid;
            const bool st = c < nchunks && ctab[2 * c] == c * 512 && ctab[2 * c + 1] > 0;
            const unsigned long long m = __ballot(st);
            if (st) {
                const int o = cnt + __popcll(m & ((1ull << tid) - 1ull));
                if (o < 2048) { seq_base[o] = ctab[2 * c]; seq_n[o] = ctab[2 * c + 1]; }
            }
            cnt += __popcll(m);
        }
        if (tid == 0) nseq_s = cnt < 2048 ? cnt : 2048;
    }
    __syncthreads();
    const int nseq = nseq_s;
    if (tid == 0) {   // prefix sums over <= ~130 windows: serial is a few microseconds
        int b = 0;
        unsigned long long wsum = 0;
        for (int q = 0; q < nseq; ++q) {
            seq_blk0[q] = b; seq_w0[q] = wsum;
            const int nb = ((seq_n[q] + 255) >> 8) * nsplit;
            b += nb; wsum += (unsigned long long)nb * (unsigned)(seq_n[q] / nsplit + 1);
        }
        seq_blk0[nseq] = b; seq_w0[nseq] = wsum; wtot_s = wsum ? wsum : 1ull;
    }
    __syncthreads();
    const int nblk = seq_blk0[nseq];
    const unsigned long long wtot = wtot_s;
    auto xcd_of = [&](int q, int j) { const unsigned long long cw = seq_w0[q] + (unsigned long long)j * (unsigned)(seq_n[q] / nsplit + 1); const int x = (int)(cw * 8ull / wtot); return x > 7 ? 7 : x; };
    for (int q = tid; q < nseq; q += 1024) {
        const int nb = seq_blk0[q + 1] - seq_blk0[q];
        for (int j = 0; j < nb; ++j) atomicMin(&first_blk[xcd_of(q, j)], seq_blk0[q] + j);
    }
    __syncthreads();
    for (int q = tid; q < nseq; q += 1024) {
        const int nb = seq_blk0[q + 1] - seq_blk0[q];
        for (int j = 0; j < nb; ++j) {
            const int x = xcd_of(q, j), slot = seq_blk0[q] + j - first_blk[x];
            KnnWg e; e.row0 = seq_base[q]; e.n = seq_n[q]; e.q0 = (j / nsplit) * 256; e.pad = j % nsplit;
            if (slot < nslots) tab[slot * 8 + x] = e;
            else { const int k = atomicAdd(&spill_s, 1); if (k < nspill) tab[nslots * 8 + k] = e; }
        }
    }
    (void)nblk;
}

// G = sweep steps between two workgroup barriers.  With a barrier per step every wave waits for the slowest selection of every tile
// (the survivors loop is data dependent: wave-state counters showed the kNN waves parked 42 % of their life); with a barrier per
// group of G tiles the waves drift inside a group and only the sums are compared.  Ring = 2 G tiles: the group being swept and the
// group in flight (requested right after the barrier, a whole group's sweep ahead of its use).
// Dropped (round 2, measured, lists identical): a producer / selector split of the 256-query workgroup - waves 0-3 (one per SIMD; waves w
// and w + 4 share a SIMD, read back from HW_ID) only multiply, 64 queries each as two alternating accumulation chains with the candidate
// fragments read once, and hand the 32 x 32 distance tiles to waves 4-7 through LDS, which only select, one step behind; one barrier per
// tile, ring of three tiles.  Stamps per tile step: 72 products 3585 cycles (50 per product although the chains are independent: the
// selecting wave's VALU stream competes for the issue port), selection of 64 queries 3746 (one wave alone is latency-bound where two
// in step share the VALU at 812 per wave and tile), barrier waits 2200: 5.60 / 6.49 ms against 4.89 / 5.98 ms.  Interleaving the two
// sets' selections in the selecting wave (both insertions every round) made it worse (6.57 / 7.97 ms): survivors are sparse, a round
// usually serves one set.  Also dropped: two accumulation chains over even / odd feature chunks in this kernel (+4 %; the waves of a
// SIMD run their product phases in step and already fill the matrix pipe between them).
// Dropped (round 2, measured, lists identical): 384-query workgroups of 12 waves - three waves per SIMD at 168 registers (the compiler
// spills 18 / 41 registers for K = 144 / 192; schedule table in 384-query blocks): 5.41 / 8.80 ms against 4.86 / 6.03 ms.
// Dropped (round 2, measured): pass 1 of tile s dealt between the products of tile s + 1 (one candidate behind every second product,
// order fenced with sched_barrier; +16 accumulator registers, no spills).  Identical lists; the fused block took 3130 cycles against
// 2575 for the products alone, the remaining selection 1820 against 2620: 5 % fewer busy cycles per wave, but the launch went from
// 6.31 / 7.05 ms to 6.56 / 7.46 ms (K = 144 / 192) - with two waves per SIMD the other wave already fills the pipe's idle slots.
template <int K, int G, bool DBG = false>
__global__ __launch_bounds__(512, 2) void knn_f16x3_wg256_kernel(const _Float16 *__restrict__ planes, const float *__restrict__ xx,
                                                                const float *__restrict__ inv_scale, const KnnWg *__restrict__ tab,
                                                                int *__restrict__ idx, const float *__restrict__ thr0, int order,
                                                                unsigned long long *__restrict__ dbg = nullptr, int nsplit = 1,
                                                                float *__restrict__ pval = nullptr, int *__restrict__ pidx = nullptr,
                                                                int64_t prows = 0) {
    constexpr int RB = 4 * K;                       // bytes per row
    constexpr int R = RB / 16;                      // chunks per row (48 / 36)
    constexpr int NC = K / 16;                      // k-chunks of 16 features
    constexpr int STAGE_B = 32 * RB;                // bytes per stage (32 candidates): 24 / 18 KiB
    constexpr int NDMA = STAGE_B / 1024;            // 1 KiB DMA instructions per stage (24 / 18)
    constexpr int NDMA_W = (NDMA + 7) / 8;
    constexpr int NST = 2 * G;
    constexpr int NXX = NST;
    constexpr int MERGE_B = 2 * 256 * 2 * TK * 4;
    constexpr int TILES_B = NST * STAGE_B + NXX * 64 * 4;
    constexpr int POOL_B = TILES_B > MERGE_B ? TILES_B : MERGE_B;
    static_assert(STAGE_B % 1024 == 0 && POOL_B <= 160 * 1024, "layout");
    __shared__ __attribute__((aligned(1024))) char pool[POOL_B];
    float *txx = (float *)(pool + NST * STAGE_B);   // [NXX][64]: |x|^2 of the 32 candidates, then their inverse scales

    const KnnWg wg = tab[blockIdx.x];
    const int n = wg.n;
    if (n <= 0) return;                             // slot beyond this XCD's run
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t row0 = (size_t)wg.row0;
    const int q0 = wg.q0;
    const char *pb = (const char *)planes + row0 * RB;
    const float *xxb = xx + row0, *isb = inv_scale + row0;
    const int qi = q0 + w * 32 + col;
    const int nt_all = (n + 31) >> 5;
    // candidate tiles of this workgroup: all of them, or split wg.pad of nsplit (short launches: the sweep is what a launch lasts)
    const int t_lo = nsplit > 1 ? (int)((int64_t)nt_all * wg.pad / nsplit) : 0;
    const int nt = nsplit > 1 ? (int)((int64_t)nt_all * (wg.pad + 1) / nsplit) - t_lo : nt_all;

    f16x8 qa[NC], qb[NC];
    const int qc = qi < n ? qi : n - 1;
    {
        const char *src = pb + (size_t)qc * RB + 16 * h;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            qa[c] = *(const f16x8 *)(src + 32 * c);
            qb[c] = *(const f16x8 *)(src + 2 * K + 32 * c);
        }
    }
    const float xxi = (qi < n) ? xxb[qi] : 0.f;
    const float thr0v = (thr0 && qi < n) ? thr0[row0 + qi] : -INFINITY;
    const float isq = isb[qc];

    kkey_t key[TK];
#pragma unroll
    for (int t = 0; t < TK; ++t) key[t] = KEY_EMPTY;

    auto issue = [&](int t, int slot, int xslot) {  // candidate tile t -> ring slot, its |x|^2 / scales -> xslot
        const int c0 = t * 32;
        char *sb = pool + slot * STAGE_B;
        int ln;
        asm volatile("v_mov_b32 %0, %1" : "=v"(ln) : "v"(lane));   // see knn_mfma_kernel
#pragma unroll
        for (int j = 0; j < NDMA_W; ++j) {
            const int ii = w + 8 * j;
            if (ii < NDMA) {
                const int ci = ii * 64 + ln;
                const int r = ci / R, p = ci - r * R;
                const int q = (R % 16 == 0) ? (p ^ (r & 15)) : (p ^ ((r >> 2) & 3));
                int c = c0 + r;
                c = c < n ? c : n - 1;
                __builtin_amdgcn_global_load_lds((knn_glb_ptr_t)(pb + (size_t)c * RB + 16 * q), (knn_lds_ptr_t)(sb + ii * 1024), 16, 0, 0);
            }
        }
        if (w == 7) {   // lanes 0..31: |x|^2, lanes 32..63: inverse scale of candidate lane & 31 (the wave with the fewest tile pieces)
            int c = c0 + (ln & 31);
            const bool real = c < n;
            c = real ? c : n - 1;
            const float *src = ln < 32 ? (real ? xxb + c : &knn_inf) : isb + c;
            __builtin_amdgcn_global_load_lds((knn_glb_ptr_t)src, (knn_lds_ptr_t)((char *)(txx + xslot * 64)), 4, 0, 0);
        }
    };
    // sweep order.  order 0: the tiles that hold this workgroup's own queries (every query meets its Morton neighbours first: the
    // pruning bound is useful from the start), then the window front to back - co-resident workgroups of a window then ask for
    // the same tile at about the same time; order 1: own tiles, then alternately outwards (the 128-query kernel's order)
    const int own0 = q0 >> 5;
    int own1 = own0 + 8;
    own1 = own1 < nt_all ? own1 : nt_all;
    const int nown = own1 - own0;
    auto tile_of = [&](int s) {
        if (nsplit > 1) return t_lo + s;                               // a split sweeps its run front to back
        if (s < nown) return own0 + s;
        const int r = s - nown;
        if (order == 0) return r < own0 ? r : r + nown;
        // outwards: alternately above own1 and below own0 while both sides last
        const int below = own0, above = nt_all - own1, m = below < above ? below : above;
        if (r < 2 * m) return (r & 1) ? own0 - 1 - (r >> 1) : own1 + (r >> 1);
        const int rr = r - 2 * m;
        return below > above ? own0 - 1 - m - rr : own1 + m + rr;
    };
    const int ngroups = (nt + G - 1) / G;
#pragma unroll
    for (int i = 0; i < G; ++i) if (i < nt) issue(tile_of(i), i, i);
    // DBG build (tools/mb_knn_wg.py stamps): shader cycles of this wave spent in [barrier + issue, MFMA block, selection]
    unsigned long long t_sync = 0, t_mfma = 0, t_sel = 0, t_prev = 0;
    if (DBG) t_prev = __builtin_amdgcn_s_memtime();
    auto stamp = [&](unsigned long long &acc_t) { if (DBG) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc_t += t - t_prev; t_prev = t; } };

    for (int g = 0; g < ngroups; ++g) {
        // group g has landed (its DMAs were issued a whole group's sweep ago) and every wave is done with group g - 1, whose half
        // of the ring is refilled now
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int half = (g & 1) * G, other = G - half;
#pragma unroll
        for (int i = 0; i < G; ++i) { const int s = (g + 1) * G + i; if (s < nt) issue(tile_of(s), other + i, other + i); }
        stamp(t_sync);
#pragma unroll 1
        for (int i = 0; i < G; ++i) {
            const int s = g * G + i;
            if (s >= nt) break;
            if (DBG && (order & 2) && w >= 4) continue;      // diagnostic: one computing wave per SIMD (results of waves 4-7 are garbage)
            const int cur = tile_of(s);
            const char *arow = pool + (half + i) * STAGE_B + col * RB;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const int swz = (R % 16 == 0) ? (col & 15) : ((col >> 2) & 3);
            f16x8 ca = *(const f16x8 *)(arow + 16 * ((0 + h) ^ swz));     // reads of chunk c + 1 in front of the products of chunk c
            f16x8 cb = *(const f16x8 *)(arow + 16 * ((K / 8 + h) ^ swz));
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                f16x8 na = ca, nb = cb;
                if (c + 1 < NC && !(DBG && (order & 4))) {       // diagnostic order & 4: no fragment reads (products of stale registers)
                    na = *(const f16x8 *)(arow + 16 * ((2 * (c + 1) + h) ^ swz));
                    nb = *(const f16x8 *)(arow + 16 * ((K / 8 + 2 * (c + 1) + h) ^ swz));
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cb, qa[c], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ca, qb[c], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ca, qa[c], acc, 0, 0, 0);
                ca = na; cb = nb;
                // pin the interleave: the two fragment reads of the next chunk, then the three products of this one
                if (c + 1 < NC) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            }
            if (DBG) { float keep; asm volatile("v_mov_b32 %0, %1" : "=v"(keep) : "v"(acc[15])); asm volatile("" :: "v"(keep)); stamp(t_mfma); }
            const float *sxx = txx + (half + i) * 64;
            knn_select<true>(acc, sxx, sxx + 32, 2.f * isq, xxi, cur * 32, n, h, key, thr0v);
            stamp(t_sel);
        }
    }
    if (DBG && dbg && lane == 0) {
        unsigned long long *o = dbg + ((size_t)blockIdx.x * 8 + w) * 8;
        o[0] = t_sync; o[1] = t_mfma; o[2] = t_sel; o[3] = (unsigned long long)nt;
    }
    __syncthreads();
    // merge the two half-lists of every query and write TK global indices per row
    {
        float *mval = (float *)pool;
        int *midx = (int *)(pool + 256 * 2 * TK * 4);
        const int ql = w * 32 + col;
#pragma unroll
        for (int t = 0; t < TK; ++t) { mval[(ql * 2 + h) * TK + t] = knn_key_val(key[t]); midx[(ql * 2 + h) * TK + t] = knn_key_idx(key[t]); }
        __syncthreads();
        if (nsplit > 1) {                                // the query's 20 best of THIS split (value, local index; empty slots: -inf, INT_MAX), in list order
            if (tid < 256 && q0 + tid < n) {
                int h0 = 0, h1 = 0;
                const size_t o0 = ((size_t)wg.pad * (size_t)prows + row0 + q0 + tid) * TK;
                const float *va = mval + (tid * 2) * TK, *vb = va + TK;
                const int *ia = midx + (tid * 2) * TK, *ib = ia + TK;
                for (int o = 0; o < TK; ++o) {
                    const float a = h0 < TK ? va[h0] : -INFINITY, b = h1 < TK ? vb[h1] : -INFINITY;
                    const int ja = h0 < TK ? ia[h0] : INT_MAX, jb = h1 < TK ? ib[h1] : INT_MAX;
                    const bool takea = (a > b) || (a == b && ja < jb);
                    pval[o0 + o] = takea ? a : b;
                    pidx[o0 + o] = takea ? ja : jb;
                    if (takea) ++h0; else ++h1;
                }
            }
            return;
        }
        if (tid < 256 && q0 + tid < n) {
            int h0 = 0, h1 = 0;
            const int kk = n < TK ? n : TK;
            int *out = idx + (row0 + q0 + tid) * (size_t)TK;
            const float *va = mval + (tid * 2) * TK, *vb = va + TK;
            const int *ia = midx + (tid * 2) * TK, *ib = ia + TK;
            int first = 0;
            for (int o = 0; o < TK; ++o) {
                if (o < kk) {
                    const float a = h0 < TK ? va[h0] : -INFINITY, b = h1 < TK ? vb[h1] : -INFINITY;
                    const int ja = h0 < TK ? ia[h0] : INT_MAX, jb = h1 < TK ? ib[h1] : INT_MAX;
                    const bool takea = (a > b) || (a == b && ja < jb);
                    const int j = (takea ? ja : jb) + (int)row0;
                    if (o == 0) first = j;
                    out[o] = j;
                    if (takea) ++h0; else ++h1;
                } else out[o] = first;
            }
        }
    }
}

// Merge of the partial lists of a split search: row r of sequence (base, n) takes the 20 best of its nsplit lists under the lists' own order
// (value descending, index ascending) - a total order, so the result is the single-sweep list whatever the split.
__global__ __launch_bounds__(256) void knn_merge_split_kernel(const int *__restrict__ ctab, const float *__restrict__ pval, const int *__restrict__ pidx,
                                                             int64_t prows, int nsplit, int *__restrict__ idx) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= prows) return;
    const int c = (int)(r >> 9), base = ctab[2 * c], n = ctab[2 * c + 1];
    if (n <= 0 || r - base >= n) return;
    int head[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) head[s] = 0;
    const int kk = n < TK ? n : TK;
    int first = 0;
    int *out = idx + r * TK;
    for (int o = 0; o < TK; ++o) {
        if (o < kk) {
            float bv = -INFINITY; int bj = INT_MAX, bs = 0;
            for (int s = 0; s < nsplit; ++s) {
                if (head[s] >= TK) continue;
                const size_t p = ((size_t)s * (size_t)prows + r) * TK + head[s];
                const float v = pval[p]; const int j = pidx[p];
                if (v > bv || (v == bv && j < bj)) { bv = v; bj = j; bs = s; }
            }
            ++head[bs];
            const int j = bj + base;
            if (o == 0) first = j;
            out[o] = j;
        } else out[o] = first;
    }
}

static unsigned long long *g_knn_dbg = nullptr;   // diagnostic build only: scp_knn_debug_buffer (tools/mb_knn_wg.py)
extern "C" SCP_API int scp_knn_debug_buffer(unsigned long long *dev_buf) { g_knn_dbg = dev_buf; return SCP_OK; }

// fp32 rows -> row scale 2^e (largest |x| into [2^13, 2^14)), two f16 planes of the scaled row ([row][2][K]), 1 / scale, and
// |x|^2 of the UNscaled row (same sequential summation as sqnorm_kernel).  A workgroup stages 64 rows in LDS (coalesced
// 16-byte loads; row stride K + 1 floats so that one thread per row can walk its row without bank conflicts), then one
// wavefront per row reduces the maximum and writes the planes.
__global__ __launch_bounds__(256) void split2_kernel(const float *__restrict__ x, int64_t npts, int K, _Float16 *__restrict__ planes,
                                                    float *__restrict__ xx, float *__restrict__ inv_scale) {
    __shared__ float tile[64 * 193];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int nr = (int)((npts - r0) < 64 ? (npts - r0) : 64);
    const int LDT = K + 1;
    const float *src = x + r0 * K;
    for (int e = tid; e < nr * (K / 4); e += 256) {
        const f32x4 val = *(const f32x4 *)(src + 4 * e);
        const int r = (4 * e) / K, c = 4 * e - r * K;
        float *d = tile + r * LDT + c;
        d[0] = val[0]; d[1] = val[1]; d[2] = val[2]; d[3] = val[3];
    }
    __syncthreads();
    if (tid < nr) {
        const float *p = tile + tid * LDT;
        float s = 0.f;
        for (int c = 0; c < K; ++c) s = s + p[c] * p[c];   // -ffp-contract=off: two roundings per term
        xx[r0 + tid] = s;
    }
    for (int rr = w; rr < nr; rr += 4) {
        const float *p = tile + rr * LDT;
        float a[3];
        float m = 0.f;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int c = lane + 64 * u;
            a[u] = c < K ? p[c] : 0.f;
            m = fmaxf(m, fabsf(a[u]));
        }
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        // scale = 2^(13 - floor(log2 m)), clamped so that scale and 1 / scale stay normal floats; m == 0 (or non-finite): scale 1
        int e = 0;
        if (m > 0.f && m < INFINITY) {
            e = 13 - (((int)(__float_as_uint(m) >> 23) & 255) - 127);
            e = e > 100 ? 100 : (e < -100 ? -100 : e);
        }
        const float sc = __uint_as_float((unsigned)(127 + e) << 23), isc = __uint_as_float((unsigned)(127 - e) << 23);
        _Float16 *o = planes + (r0 + rr) * 2 * K;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int c = lane + 64 * u;
            if (c < K) {
                const float t = a[u] * sc;                 // exact
                const _Float16 ta = (_Float16)t;
                o[c] = ta;
                o[K + c] = (_Float16)(t - (float)ta);
            }
        }
        if (lane == 0) inv_scale[r0 + rr] = isc;
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

// |x|^2 scratch + the padded / split copy of the input, grown on demand - ONE PER STREAM: two frames may run their kNN searches
// concurrently on two streams (FrameEncoder lanes); launches of one stream are ordered, so its buffer can be reused call after call
struct KnnScratch { void *stream; bool used; unsigned long long last; DevBuf buf; };
static KnnScratch g_xx_tab[8];
static unsigned long long g_xx_clock = 0;
// A ninth stream takes over the slot that has not been used for the longest time (streams of encoders that no longer exist, in
// practice): its buffer is released first - hipFree waits for everything in flight, so a kernel of the old stream that may still
// be reading the buffer finishes before the memory goes away.  Two live streams never share a buffer.
static DevBuf *knn_scratch(hipStream_t st) {
    ++g_xx_clock;
    for (auto &e : g_xx_tab) if (e.used && e.stream == (void *)st) { e.last = g_xx_clock; return &e.buf; }
    KnnScratch *pick = nullptr;
    for (auto &e : g_xx_tab) if (!e.used) { pick = &e; break; }
    if (!pick) {
        for (auto &e : g_xx_tab) if (!pick || e.last < pick->last) pick = &e;
        pick->buf.release();
    }
    pick->used = true; pick->stream = (void *)st; pick->last = g_xx_clock;
    return &pick->buf;
}

static int g_knn_mode = -1;
static int knn_mode() {   // 1 = f16x3 (default), 0 = exact fp32 MFMA (SCP_KNN=f32 or scp_set_knn_mode(0))
    const int c = scp_ctx_knn_mode();            // the calling thread's current scp_ctx decides; without one, the process default
    if (c >= 0) return c;
    if (g_knn_mode < 0) { const char *e = getenv("SCP_KNN"); g_knn_mode = (e && e[0] == 'f') ? 0 : 1; }
    return g_knn_mode;
}
extern "C" SCP_API int scp_set_knn_mode(int32_t f16x3) { g_knn_mode = f16x3 ? 1 : 0; return SCP_OK; }
// workgroup shape of the packed f16x3 search: 256 = 256-query workgroups on the XCD schedule, a barrier per group of 3 / 4 tiles (default);
// SCP_KNN_WG=257 / 258: groups of two tiles / one tile; +16: outward sweep order; 128: the 128-query kernel in launch order (all give
// identical neighbour lists)
static int knn_split_on() {   // SCP_KNN_SPLIT=0: A/B bracket of the split sweep of short launches (identical lists)
    static int on = -1;
    if (on < 0) { const char *e = getenv("SCP_KNN_SPLIT"); on = (e && e[0] == '0') ? 0 : 1; }
    return on;
}
static int g_knn_wg = -1;
static int knn_wg() { if (g_knn_wg < 0) { const char *e = getenv("SCP_KNN_WG"); g_knn_wg = e ? atoi(e) : 256; } return g_knn_wg; }
extern "C" SCP_API int scp_set_knn_workgroup(int32_t v) { g_knn_wg = v; return SCP_OK; }

static int knn_launch(const float *x, int64_t npts, int C, dim3 grid, int n, int k, int *idx, const int *ctab, hipStream_t st,
                      const float *thr0 = nullptr) {
    const size_t xx_bytes = ((size_t)npts * sizeof(float) + 1023) & ~(size_t)1023;
    const bool split = (C == 144 || C == 192) && knn_mode() == 1;
    const int RB = C * 4;
    DevBuf *sp = knn_scratch(st);
    if (!sp) return SCP_EINVAL;
    DevBuf &sbuf = *sp;
    // short launches (round 5): when the launch has too few 256-query blocks for the chip, every block's candidate sweep is cut into nsplit runs
    // (one workgroup each) and the partial lists are merged - the sweep is what a launch lasts (0.77 ms for one 8 192-point window on 32 CUs).
    // nsplit from the padded row count (the host does not know the sequences): blocks x nsplit <= 256, at least ~4 tiles per run, <= 16.
    int nsplit = 1;
    if (split && ctab && knn_wg() == 256 && !g_knn_dbg && knn_split_on()) {
        const int64_t nblk = npts / 256 > 0 ? npts / 256 : 1;
        while (nsplit < 16 && nblk * (nsplit * 2) <= 256 && npts / 32 >= 8 * (int64_t)nsplit) nsplit *= 2;
    }
    const size_t part_bytes = nsplit > 1 ? (size_t)nsplit * (size_t)npts * TK * 8 + 256 : 0;
    int rc = sbuf.reserve(xx_bytes * (split ? 2 : 1) + (C <= 4 ? (size_t)npts * 16 + ((size_t)npts / 32 + grid.y + 1) * 32 : (split ? (size_t)npts * RB + 256 + (((size_t)(npts / 256) * 2 + 8) * nsplit + 520) * sizeof(KnnWg) + part_bytes : 0)));
    if (rc) return rc;
    float *xx = sbuf.as<float>();
    void *aux = (char *)sbuf.p + xx_bytes;
    if (split) {
        float *isc = (float *)aux;
        _Float16 *pl = (_Float16 *)((char *)aux + xx_bytes);
        hipLaunchKernelGGL(split2_kernel, dim3((unsigned)cdiv64(npts, 64)), dim3(256), 0, st, x, npts, C, pl, xx, isc);
        if (ctab && knn_wg() >= 256 && npts / 512 <= 2048) {   // the schedule kernel lists at most 2048 sequences (one per 512-row chunk at worst)
            // XCD-affine schedule of 256-query workgroups (see knn_f16x3_wg256_kernel); the table lives behind the planes
            const int nchunks = (int)(npts / 512);
            const int nspill = (int)(npts / 256) * nsplit;          // every block of the launch fits
            const int nslots = nspill / 8 + 64;                      // an XCD's run: its share of the blocks + slack for short windows
            const int nblocks = nslots * 8 + nspill;
            KnnWg *tab = (KnnWg *)((char *)pl + (((size_t)npts * RB + 255) & ~(size_t)255));
            float *pval = (float *)((char *)tab + (((size_t)nblocks * sizeof(KnnWg) + 255) & ~(size_t)255));
            int *pidx = (int *)(pval + (size_t)nsplit * (size_t)npts * TK);
            HIP_TRY(hipMemsetAsync(tab, 0, (size_t)nblocks * sizeof(KnnWg), st));
            hipLaunchKernelGGL(knn_sched_kernel, dim3(1), dim3(1024), 0, st, ctab, nchunks, tab, nslots, nspill, nsplit);
            // knn_wg(): 256 = groups of 3 (K = 192) / 4 (K = 144) tiles, front-to-back order; 257: groups of 2; 258: one tile per barrier;
            // +16: outward order instead
            const int shape = knn_wg() & 15, order = (knn_wg() >> 4) & 7;
            SCP_PROF(SCP_PROF_KNN_FEAT, st, (double)C);
#define KNN_LAUNCH(KK, GG) hipLaunchKernelGGL((knn_f16x3_wg256_kernel<KK, GG>), dim3(nblocks), dim3(512), 0, st, (const _Float16 *)pl, (const float *)xx, \
                                              (const float *)isc, (const KnnWg *)tab, idx, thr0, order, (unsigned long long *)nullptr, nsplit, pval, pidx, (int64_t)npts)
            if (C == 144) { if (shape == 2) KNN_LAUNCH(144, 1); else if (shape == 1) KNN_LAUNCH(144, 2); else KNN_LAUNCH(144, 4); }
            else if (g_knn_dbg && shape == 0)
                hipLaunchKernelGGL((knn_f16x3_wg256_kernel<192, 3, true>), dim3(nblocks), dim3(512), 0, st, (const _Float16 *)pl, (const float *)xx,
                                   (const float *)isc, (const KnnWg *)tab, idx, thr0, order, g_knn_dbg);
            else { if (shape == 2) KNN_LAUNCH(192, 1); else if (shape == 1) KNN_LAUNCH(192, 2); else KNN_LAUNCH(192, 3); }
#undef KNN_LAUNCH
            LAUNCH_CHECK();
            if (nsplit > 1) {
                hipLaunchKernelGGL(knn_merge_split_kernel, dim3((unsigned)cdiv64(npts, 256)), dim3(256), 0, st, ctab, (const float *)pval, (const int *)pidx, (int64_t)npts, nsplit, idx);
                LAUNCH_CHECK();
            }
            return SCP_OK;
        }
        SCP_PROF(SCP_PROF_KNN_FEAT, st, (double)C);
        if (C == 144) hipLaunchKernelGGL(knn_f16x3_kernel<144>, grid, dim3(256), 0, st, (const _Float16 *)pl, (const float *)xx, (const float *)isc, n, k, idx, ctab, thr0);
        else hipLaunchKernelGGL(knn_f16x3_kernel<192>, grid, dim3(256), 0, st, (const _Float16 *)pl, (const float *)xx, (const float *)isc, n, k, idx, ctab, thr0);
        LAUNCH_CHECK();
        return SCP_OK;
    }
    hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)cdiv64(npts, 256)), dim3(256), 0, st, x, npts, C, xx);
    if (C <= 4) {
        hipLaunchKernelGGL(pad4_kernel, dim3((unsigned)cdiv64(npts, 256)), dim3(256), 0, st, x, npts, C, (float *)aux);
        // tile boxes behind the padded points: per sequence in packed mode (one item of npts rows, sequences start at multiples of 512),
        // per batch item otherwise
        float *bb = (float *)((char *)aux + (size_t)npts * 16);
        const int bn = ctab ? (int)npts : n, bB = ctab ? 1 : (int)grid.y;
        hipLaunchKernelGGL(bbox32_kernel, dim3((unsigned)cdiv64((int64_t)((bn + 31) / 32) * bB, 256)), dim3(256), 0, st, (const float *)aux, bn, bB, bb);
        SCP_PROF(SCP_PROF_KNN_POS, st, 4.0);
        hipLaunchKernelGGL((knn_mfma_kernel<2, 16, true>), grid, dim3(256), 0, st, (const float *)aux, (const float *)xx, n, 4, k, idx, ctab, thr0,
                           (const float *)bb);
    } else {
        SCP_PROF(SCP_PROF_KNN_FEAT, st, (double)C);
        if (C == 144) hipLaunchKernelGGL((knn_mfma_kernel<72, 1>), grid, dim3(256), 0, st, x, (const float *)xx, n, C, k, idx, ctab, thr0);
        else hipLaunchKernelGGL((knn_mfma_kernel<96, 1>), grid, dim3(256), 0, st, x, (const float *)xx, n, C, k, idx, ctab, thr0);
    }
    LAUNCH_CHECK();
    return SCP_OK;
}

extern "C" int scp_knn_topk(const float *x, int32_t B, int32_t n, int32_t C, int32_t k, int32_t *idx, void *stream) {
    if (!x || !idx || B <= 0 || n <= 0 || C <= 0 || k <= 0 || k > TK || k > n) return SCP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int64_t npts = (int64_t)B * n;
    if (C <= 4 || ((C == 144 || C == 192) && ((uintptr_t)x & 15) == 0))
        return knn_launch(x, npts, C, dim3((n + 127) / 128, B), n, k, idx, nullptr, st);
    DevBuf *sp = knn_scratch(st);
    if (!sp) return SCP_EINVAL;
    DevBuf &sbuf = *sp;
    int rc = sbuf.reserve((size_t)npts * sizeof(float));
    if (rc) return rc;
    float *xx = sbuf.as<float>();
    hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)cdiv64(npts, 256)), dim3(256), 0, st, x, npts, C, xx);
    LAUNCH_CHECK();
    {
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
    if (!x || !ctab || !idx || total_rows <= 0 || (total_rows & 511) || (C != 144 && C != 192 && C > 4) || (C > 4 && ((uintptr_t)x & 15)))
        return SCP_EINVAL;
    return knn_launch(x, (int64_t)total_rows, C, dim3(total_rows / 128, 1), 0, TK, idx, ctab, (hipStream_t)stream);
}

// the same with an a-priori pruning bound per row (thr0[row], in the kernel's distance convention 2 x.y - |x|^2 - |y|^2, i.e.
// minus the squared distance): a value that at least 20 candidates of the row's sequence are known to reach.  Rows prune with
// it from the first tile on instead of building their bound from scratch; the result is unchanged.
extern "C" SCP_API int scp_knn_topk_packed_bounded(const float *x, const int32_t *ctab, int32_t total_rows, int32_t C, const float *thr0,
                                                   int32_t *idx, void *stream) {
    if (!x || !ctab || !idx || total_rows <= 0 || (total_rows & 511) || (C != 144 && C != 192 && C > 4) || (C > 4 && ((uintptr_t)x & 15)))
        return SCP_EINVAL;
    return knn_launch(x, (int64_t)total_rows, C, dim3(total_rows / 128, 1), 0, TK, idx, ctab, (hipStream_t)stream, thr0);
}
