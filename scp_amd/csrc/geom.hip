// Stage G of the SCP encode path on gfx950: quantiser, Morton keys, octree serialisation, context tables.
//
// Algorithm (sorted-prefix formulation; independent of the oracle's top-down partition):
//   key   = segment << 58 | Morton(x,y,z) with x the most significant axis of every 3-bit digit
//   sort  = stable LSD radix (sort_u64.hip)
//   a sorted key i is the FIRST key ("head") of a tree node at level L (root = 1, leaves = D+1) iff it shares
//   fewer than L-1 leading digits with key i-1; the node's BFS rank inside its level is the number of heads
//   before it, obtained from 64-bit wavefront ballots + popcounts and one scan over [level][block] counters;
//   node (L, r) knows its first child = rank of the same key at level L+1, so the occupancy byte is the OR of
//   1 << digit over a contiguous run of <= 8 children: no atomics, no per-node point lists.
#include <math.h>
#include <string.h>
#include <algorithm>
#include <new>
#include <vector>
#include "scp_internal.h"

#define NLV (SCP_MAX_DEPTH + 2)  // level slots 1..D+1 (slot 0 unused)
#define SENTINEL_SEG 63
#define SEG_SHIFT 58
#define WG 256

// ------------------------------------------------------------------------------------------------ device tables
struct SegTab {                 // one per segment, lives in device memory
    int64_t pt_begin, pt_count; // slice of q
    int64_t key_begin;          // slice of the key array before sorting (== prefix of pt_count)
    int32_t path_len, path_bits, drop_last, depth;
    int64_t node_base, leaf_base, n_nodes, n_leaves;
    int64_t level_off[NLV + 1]; // [L] = first node of level L relative to node_base (L = 1..D), [D+1] = n_nodes
    int64_t rank0[NLV];         // global rank (per level) of the segment's first head
};

__device__ __forceinline__ uint64_t spread3(uint64_t v) {
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}
__device__ __forceinline__ uint32_t compact3(uint64_t v) {
    v &= 0x1249249249249249ull;
    v = (v ^ (v >> 2)) & 0x10c30c30c30c30c3ull;
    v = (v ^ (v >> 4)) & 0x100f00f00f00f00full;
    v = (v ^ (v >> 8)) & 0x1f0000ff0000ffull;
    v = (v ^ (v >> 16)) & 0x1f00000000ffffull;
    v = (v ^ (v >> 32)) & 0x1fffffull;
    return (uint32_t)v;
}

// ------------------------------------------------------------------------------------------------ G1: quantiser
// float <-> order-preserving uint (for atomicMax / atomicMin on floats of either sign)
__device__ __forceinline__ uint32_t f2ord(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
static inline float ord2f_host(uint32_t u) { uint32_t v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u; float f; memcpy(&f, &v, 4); return f; }

// correctly rounded float32 square root: hipcc folds (float)sqrt((double)s) back to the 1-ulp v_sqrt_f32 sequence, so
// refine in float64 by hand (two Newton steps from the hardware estimate, IEEE float64 division) and round once.
__device__ __forceinline__ float sqrt_cr(float s) {
    if (!(s > 0.f)) return s == 0.f ? 0.f : __builtin_nanf("");
    const double d = (double)s;
    double r = (double)__builtin_sqrtf(s);
    r = 0.5 * (r + d / r);
    r = 0.5 * (r + d / r);
    return (float)r;
}

// data_preprocess.py:200-207 / :171-177, float32 arithmetic in numpy's evaluation order (no FMA contraction);
// atan2/acos are evaluated in float64 and rounded once (numpy's SIMD float32 routines are not reproducible
// across CPUs - DESIGN.md "float -> integer boundary").
__global__ __launch_bounds__(WG) void transform_kernel(const float *__restrict__ xyz, int64_t n, int mode, float *__restrict__ tr,
                                                      uint32_t *__restrict__ red /* [0]=max rho, [1]=min z (ordered) */) {
    const int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x;
    float a = 0.f, b = 0.f, c = 0.f;
    uint32_t omax = 0u, omin = 0xffffffffu;
    if (i < n) {
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        if (mode == SCP_CART) {
            a = x; b = y; c = z;
        } else {
            const float xx = __fmul_rn(x, x), yy = __fmul_rn(y, y);
            float s = __fadd_rn(xx, yy);
            if (mode == SCP_SPHER) s = __fadd_rn(s, __fmul_rn(z, z));
            a = sqrt_cr(s);
            const float xe = __fadd_rn(x, 1e-9f);
            float phi = (float)atan2((double)y, (double)xe);
            if (phi < 0.f) phi = __fadd_rn(phi, 6.2831855f);
            b = phi;
            c = (mode == SCP_SPHER) ? (float)acos((double)(float)((double)z / (double)a)) : z;
        }
        tr[3 * i] = a; tr[3 * i + 1] = b; tr[3 * i + 2] = c;
        omax = f2ord(a);
        omin = f2ord(c);
    }
    // wave reduce, then LDS across the four waves, then ONE atomic pair per workgroup (thousands of same-address atomics
    // serialise at ~10 ns each and used to dominate this kernel)
    __shared__ uint32_t smax[4], smin[4];
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t o1 = __shfl_xor(omax, off), o2 = __shfl_xor(omin, off);
        omax = omax > o1 ? omax : o1;
        omin = omin < o2 ? omin : o2;
    }
    if ((threadIdx.x & 63) == 0) { smax[threadIdx.x >> 6] = omax; smin[threadIdx.x >> 6] = omin; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = max(max(smax[0], smax[1]), max(smax[2], smax[3])), b = min(min(smin[0], smin[1]), min(smin[2], smin[3]));
        atomicMax(&red[0], a);
        atomicMin(&red[1], b);
    }
}

struct QuantParams { double qs[3]; double off[3]; float qsf; float offf; int cart_f32; };

// data_preprocess.py:56,68: spher/cylin divide in float64 ((float64)f32 - off) / qs; the Cartesian branch stays in
// float32 under numpy>=2 promotion ((x - (-200)) / qs with python scalars): verified against the oracle.
__global__ __launch_bounds__(WG) void quantize_kernel(const float *__restrict__ tr, int64_t n, QuantParams p, int32_t *__restrict__ q,
                                                     int32_t *__restrict__ red /* [0]=max, [1]=min */) {
    const int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x;
    int32_t mx = INT32_MIN, mn = INT32_MAX;
    if (i < n) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float t = tr[3 * i + k];
            double r;
            if (p.cart_f32) r = (double)rintf((float)((double)__fsub_rn(t, p.offf) / (double)p.qsf));
            else r = rint(((double)t - p.off[k]) / p.qs[k]);
            const int32_t v = (int32_t)r;
            q[3 * i + k] = v;
            mx = v > mx ? v : mx;
            mn = v < mn ? v : mn;
        }
    }
    __shared__ int32_t smx[4], smn[4];
    for (int off = 32; off > 0; off >>= 1) {
        int32_t o1 = __shfl_xor(mx, off), o2 = __shfl_xor(mn, off);
        mx = mx > o1 ? mx : o1;
        mn = mn < o2 ? mn : o2;
    }
    if ((threadIdx.x & 63) == 0) { smx[threadIdx.x >> 6] = mx; smn[threadIdx.x >> 6] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(&red[0], max(max(smx[0], smx[1]), max(smx[2], smx[3])));
        atomicMin(&red[1], min(min(smn[0], smn[1]), min(smn[2], smn[3])));
    }
}

static DevBuf g_qtmp;  // transform scratch for scp_quantize when the caller does not ask for tr_out

extern "C" int scp_quantize(const float *xyz, int64_t n, int32_t mode, double qs, double cart_offset, int32_t *q_out,
                            float *tr_out, scp_quant_info *info, void *stream) {
    if (!xyz || !q_out || !info || n <= 0 || mode < 0 || mode > 2 || !(qs > 0)) return SCP_EINVAL;
    scp_d2h_abort();
    hipStream_t st = (hipStream_t)stream;
    int rc = g_qtmp.reserve((size_t)n * 12 + 64);
    if (rc) return rc;
    // scratch layout: [4 reduction words, padded to 64 B][transformed coordinates].  (The reduction words used to sit at
    // `cap - 32`; the capacity is not a multiple of 4 for odd n, and a misaligned atomic faults.)
    uint32_t *red = g_qtmp.as<uint32_t>();
    float *tr = tr_out ? tr_out : (float *)((char *)g_qtmp.p + 64);
    uint32_t init[4] = {0u, 0xffffffffu, (uint32_t)INT32_MIN, (uint32_t)INT32_MAX};
    HIP_TRY(hipMemcpyAsync(red, init, sizeof(init), hipMemcpyHostToDevice, st));
    const int nb = (int)cdiv64(n, WG);
    hipLaunchKernelGGL(transform_kernel, dim3(nb), dim3(WG), 0, st, xyz, n, mode, tr, red);
    LAUNCH_CHECK();
    uint32_t h[2];
    SCP_D2H_TRY(scp_d2h_async(h, red, 8, st));
    SCP_D2H_TRY(scp_stream_wait(st));
    const float rho_max = ord2f_host(h[0]), z_min = ord2f_host(h[1]);

    QuantParams p;
    memset(&p, 0, sizeof(p));
    memset(info, 0, sizeof(*info));
    if (mode == SCP_CART) {
        p.cart_f32 = 1; p.qsf = (float)qs; p.offf = (float)cart_offset;
        for (int k = 0; k < 3; ++k) { info->qs[k] = qs; info->offset[k] = cart_offset; }
    } else {
        // bin_num = np.round(rho.max() / qs) + 1 : float32 arithmetic (numpy>=2 keeps float32 with a python scalar)
        const float binf = rintf(rho_max / (float)qs) + 1.0f;
        const float q_phi = 6.2831855f / (binf - 1.0f);  // 2*math.pi / (bin_num-1) evaluated in float32
        const float q_th = 3.1415927f / (binf - 1.0f);
        info->bin_num = (double)binf;
        p.qs[0] = qs; p.qs[1] = (double)q_phi; p.qs[2] = (mode == SCP_SPHER) ? (double)q_th : qs;
        p.off[2] = (mode == SCP_CYLIN) ? (double)z_min : 0.0;
        for (int k = 0; k < 3; ++k) { info->qs[k] = p.qs[k]; info->offset[k] = p.off[k]; }
    }
    hipLaunchKernelGGL(quantize_kernel, dim3(nb), dim3(WG), 0, st, (const float *)tr, n, p, q_out, (int32_t *)(red + 2));
    LAUNCH_CHECK();
    int32_t hm[2];
    SCP_D2H_TRY(scp_d2h_async(hm, red + 2, 8, st));
    SCP_D2H_TRY(scp_stream_wait(st));
    info->max_coord = hm[0];
    info->min_coord = hm[1];
    return hm[1] < 0 ? SCP_EINVAL : SCP_OK;
}

// ------------------------------------------------------------------------------------------------ G2: octree
struct scp_geom {
    DevBuf segtab, keys_a, keys_b, red, blkcnt, leafkey, leafoct;
    DevBuf occ, level, octant, parent, pos, fchild, posmm;
    DevBuf tr, front;             // scp_geom_build_xyz: transformed coordinates of every frame, the frame / shell tables of its two kernels
    DevBuf rowtab;                // scp_geom_context_ehem_all: per-segment row bases
    RadixWorkspace radix;
    std::vector<SegTab> segs;     // host mirror
    std::vector<int32_t> mm_host; // initial (min, max) words of a build: a member, so the copy that reads it never outlives it
    std::vector<int64_t> rowtab_host; // [0, nseg): first context row of every segment, [nseg, 2 nseg): first (min, max) row (scp_geom_context_ehem_all)
    hipEvent_t h2d_done = nullptr; // recorded behind the last host -> device table copy of a build (see geom_take_segments)
    uint64_t *sorted = nullptr;
    int64_t n_keys = 0, total_nodes = 0, total_leaves = 0;
    int nseg = 0, lmax = 0, ntiles = 0;
    bool built = false;
};

__global__ __launch_bounds__(WG) void seg_minmax_kernel(const int32_t *__restrict__ q, const SegTab *__restrict__ tab,
                                                       int32_t *__restrict__ red /* [seg][2] */) {
    const SegTab &s = tab[blockIdx.y];
    int32_t mx = INT32_MIN, mn = INT32_MAX;
    for (int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x; i < s.pt_count * 3; i += (int64_t)gridDim.x * WG) {
        const int32_t v = q[s.pt_begin * 3 + i];
        mx = v > mx ? v : mx;
        mn = v < mn ? v : mn;
    }
    __shared__ int32_t smx[4], smn[4];
    for (int off = 32; off > 0; off >>= 1) {
        int32_t o1 = __shfl_xor(mx, off), o2 = __shfl_xor(mn, off);
        mx = mx > o1 ? mx : o1;
        mn = mn < o2 ? mn : o2;
    }
    if ((threadIdx.x & 63) == 0) { smx[threadIdx.x >> 6] = mx; smn[threadIdx.x >> 6] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(&red[2 * blockIdx.y], max(max(smx[0], smx[1]), max(smx[2], smx[3])));
        atomicMin(&red[2 * blockIdx.y + 1], min(min(smn[0], smn[1]), min(smn[2], smn[3])));
    }
}

__global__ __launch_bounds__(WG) void morton_key_kernel(const int32_t *__restrict__ q, const SegTab *__restrict__ tab,
                                                       uint64_t *__restrict__ keys) {
    const SegTab &s = tab[blockIdx.y];
    const int D = s.depth;
    for (int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x; i < s.pt_count; i += (int64_t)gridDim.x * WG) {
        const int64_t g = s.pt_begin + i;
        const uint32_t x = (uint32_t)q[3 * g], y = (uint32_t)q[3 * g + 1], z = (uint32_t)q[3 * g + 2];
        uint64_t key = ((uint64_t)blockIdx.y << SEG_SHIFT) | (spread3(x) << 2) | (spread3(y) << 1) | spread3(z);
        // rho-shell filter, Octree.py:188: the top path_len bits of the x axis must equal the path
        if (s.path_len > 0 && (int)(x >> (D - s.path_len)) != s.path_bits) key = ~0ull;
        keys[s.key_begin + i] = key;
    }
}

// number of leading 3-bit digits two keys of the same segment share (D = all digits -> duplicate point)
__device__ __forceinline__ int shared_digits(uint64_t a, uint64_t b, int D) {
    const uint64_t x = (a ^ b) & ((1ull << SEG_SHIFT) - 1ull);
    if (x == 0) return D;
    const int hb = 63 - __clzll((long long)x);  // highest differing bit, < 3D
    return D - 1 - hb / 3;
}

// lowest tree level at which sorted key i starts a new node (1 => segment start, D+2 => duplicate: never)
__device__ __forceinline__ int head_level(const uint64_t *__restrict__ keys, int64_t i, int64_t n, const SegTab *__restrict__ tab,
                                          int &seg, int &D, uint64_t &key) {
    seg = SENTINEL_SEG; D = 0; key = ~0ull;
    if (i >= n) return 1 << 20;
    key = keys[i];
    seg = (int)(key >> SEG_SHIFT);
    if (seg == SENTINEL_SEG) return 1 << 20;
    D = tab[seg].depth;
    if (i == 0) return 1;
    const uint64_t prev = keys[i - 1];
    if ((int)(prev >> SEG_SHIFT) != seg) return 1;
    return shared_digits(key, prev, D) + 2;
}

// T1: per-block head counts for every level, laid out [level][block]
__global__ __launch_bounds__(WG) void tree_count_kernel(const uint64_t *__restrict__ keys, int64_t n, const SegTab *__restrict__ tab,
                                                       int lmax, uint32_t *__restrict__ blkcnt, int nblk) {
    __shared__ uint32_t cnt[NLV];
    if (threadIdx.x < NLV) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x;
    int seg, D; uint64_t key;
    const int hl = head_level(keys, i, n, tab, seg, D, key);
    for (int L = 1; L <= lmax; ++L) {
        const uint64_t b = __ballot(hl <= L && L <= D + 1);
        if ((threadIdx.x & 63) == 0 && b) atomicAdd(&cnt[L], (uint32_t)__popcll(b));
    }
    __syncthreads();
    // rows 0 and lmax+1 of the [lmax+2][nblk] table stay zero: after the exclusive scan, row L+1 column 0 closes level L
    if (threadIdx.x <= lmax + 1)
        blkcnt[(int64_t)threadIdx.x * nblk + blockIdx.x] = (threadIdx.x >= 1 && threadIdx.x <= lmax) ? cnt[threadIdx.x] : 0u;
}

// exclusive ranks of this thread's key at every level (global, level-major numbering from the scanned table)
__device__ __forceinline__ void level_ranks(int hl, int D, int lmax, const uint32_t *__restrict__ blkscan, int nblk,
                                            uint32_t (*wc)[4], uint32_t *rank /* [NLV] */) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int L = 1; L <= lmax; ++L) {
        const uint64_t b = __ballot(hl <= L && L <= D + 1);
        if (lane == 0) wc[L][w] = (uint32_t)__popcll(b);
        rank[L] = (uint32_t)__popcll(b & lt);
    }
    __syncthreads();
    for (int L = 1; L <= lmax; ++L) {
        uint32_t base = blkscan[(int64_t)L * nblk + blockIdx.x];
        for (int ww = 0; ww < w; ++ww) base += wc[L][ww];
        rank[L] += base;
    }
}

// T2: ranks of every segment's first key -> tab[s].rank0[L]
__global__ __launch_bounds__(WG) void tree_segrank_kernel(const uint64_t *__restrict__ keys, int64_t n, SegTab *__restrict__ tab, int lmax,
                                                         const uint32_t *__restrict__ blkscan, int nblk) {
    __shared__ uint32_t wc[NLV][4];
    const int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x;
    int seg, D; uint64_t key;
    const int hl = head_level(keys, i, n, tab, seg, D, key);
    uint32_t rank[NLV];
    level_ranks(hl, D, lmax, blkscan, nblk, wc, rank);
    if (hl == 1)
        for (int L = 1; L <= lmax; ++L) tab[seg].rank0[L] = rank[L];
}

// T3: every head writes its node; heads of the leaf pseudo-level D+1 write the leaf key / last digit
__global__ __launch_bounds__(WG) void tree_write_kernel(const uint64_t *__restrict__ keys, int64_t n, const SegTab *__restrict__ tab, int lmax,
                                                       const uint32_t *__restrict__ blkscan, int nblk, uint8_t *__restrict__ level,
                                                       uint8_t *__restrict__ octant, int32_t *__restrict__ parent, int32_t *__restrict__ pos,
                                                       int32_t *__restrict__ fchild, uint64_t *__restrict__ leafkey, uint8_t *__restrict__ leafoct) {
    __shared__ uint32_t wc[NLV][4];
    const int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x;
    int seg, D; uint64_t key;
    const int hl = head_level(keys, i, n, tab, seg, D, key);
    uint32_t rank[NLV];
    level_ranks(hl, D, lmax, blkscan, nblk, wc, rank);
    if (seg == SENTINEL_SEG || hl > D + 1) return;
    const SegTab &s = tab[seg];
    const uint64_t m = key & ((1ull << SEG_SHIFT) - 1ull);
    const uint32_t x = compact3(m >> 2), y = compact3(m >> 1), z = compact3(m);
    for (int L = hl; L <= D; ++L) {
        const int64_t nd = s.node_base + s.level_off[L] + ((int64_t)rank[L] - s.rank0[L]);
        const int sh = D - L + 1;  // bits below this node
        level[nd] = (uint8_t)L;
        octant[nd] = (L == 1) ? (uint8_t)1 : (uint8_t)(((m >> (3 * sh)) & 7ull) + 1);
        const uint32_t keep = ~((1u << sh) - 1u);
        pos[3 * nd] = (int32_t)(x & keep); pos[3 * nd + 1] = (int32_t)(y & keep); pos[3 * nd + 2] = (int32_t)(z & keep);
        // parent = node holding this key one level up; the key is not its head unless hl <= L-1
        if (L == 1) parent[nd] = -1;
        else {
            const int64_t pr = (int64_t)rank[L - 1] + ((hl <= L - 1) ? 0 : -1);
            parent[nd] = (int32_t)(s.node_base + s.level_off[L - 1] + (pr - s.rank0[L - 1]));
        }
        if (L < D) fchild[nd] = (int32_t)(s.node_base + s.level_off[L + 1] + ((int64_t)rank[L + 1] - s.rank0[L + 1]));
        else fchild[nd] = (int32_t)(s.leaf_base + ((int64_t)rank[D + 1] - s.rank0[D + 1]));
    }
    {   // leaf pseudo-level
        const int64_t lf = s.leaf_base + ((int64_t)rank[D + 1] - s.rank0[D + 1]);
        leafkey[lf] = key;
        leafoct[lf] = (uint8_t)((m & 7ull) + 1);
    }
}

// T4: occupancy byte = OR over the contiguous child run; per-level min/max of the node origins
__global__ __launch_bounds__(WG) void tree_occ_kernel(const SegTab *__restrict__ tab, const uint8_t *__restrict__ level,
                                                     const uint8_t *__restrict__ octant, const int32_t *__restrict__ fchild,
                                                     const uint8_t *__restrict__ leafoct, const int32_t *__restrict__ pos,
                                                     uint8_t *__restrict__ occ, int32_t *__restrict__ posmm /* [seg][NLV][2] */) {
    __shared__ int32_t smn[NLV], smx[NLV];
    const SegTab &s = tab[blockIdx.y];
    if (threadIdx.x < NLV) { smn[threadIdx.x] = INT32_MAX; smx[threadIdx.x] = INT32_MIN; }
    __syncthreads();
    const int D = s.depth;
    // a workgroup owns a CONTIGUOUS run of nodes (BFS order: one or two levels), at most 256 workgroups per segment: a level's two
    // (min, max) words then take a few dozen same-address atomics instead of one per 256 nodes (~400 for a deep level of an L16-m frame -
    // serialised, they were most of this kernel's 43 us).  (A look-before-atomic with non-temporal loads made it 146 us.)
    const int64_t per = ((s.n_nodes + gridDim.x - 1) / gridDim.x + WG - 1) / WG * WG;
    const int64_t r_end = ((int64_t)blockIdx.x + 1) * per < s.n_nodes ? ((int64_t)blockIdx.x + 1) * per : s.n_nodes;
    for (int64_t r = (int64_t)blockIdx.x * per + threadIdx.x; r < r_end; r += WG) {
        const int64_t nd = s.node_base + r;
        const int L = level[nd];
        const int64_t lvl_end = s.level_off[L + 1];  // relative end of this level
        int64_t c0 = fchild[nd], c1;
        if (r + 1 < lvl_end) c1 = fchild[nd + 1];
        else c1 = (L < D) ? s.node_base + s.level_off[L + 2] : s.leaf_base + s.n_leaves;
        const uint8_t *src = (L < D) ? octant : leafoct;
        uint32_t o = 0;
        for (int64_t c = c0; c < c1; ++c) o |= 1u << (src[c] - 1);
        occ[nd] = (uint8_t)o;
        // per-level (min, max) of the node origins.  Nodes are in BFS order, so the 64 nodes of a wavefront nearly always share one level:
        // then the wave reduces by shuffles and ONE lane touches the LDS slot (256 same-address LDS atomics per workgroup pass used to
        // serialise: 44 us for the 577 k nodes of an L16-m frame, the slowest kernel of the stage)
        const bool counts = !(s.drop_last && r == s.n_nodes - 1);
        int32_t mn = INT32_MAX, mx = INT32_MIN;
        if (counts) {
            const int32_t a = pos[3 * nd], b = pos[3 * nd + 1], cc = pos[3 * nd + 2];
            mn = min(a, min(b, cc)); mx = max(a, max(b, cc));
        }
        const int L0 = __builtin_amdgcn_readfirstlane(L);
        if (__ballot(L != L0) == 0ull && __ballot(1) == ~0ull) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { mn = min(mn, __shfl_xor(mn, o)); mx = max(mx, __shfl_xor(mx, o)); }
            if ((threadIdx.x & 63) == 0 && mn != INT32_MAX) { atomicMin(&smn[L0], mn); atomicMax(&smx[L0], mx); }
        } else if (counts) {
            atomicMin(&smn[L], mn);
            atomicMax(&smx[L], mx);
        }
    }
    __syncthreads();
    if (threadIdx.x >= 1 && threadIdx.x <= D && smn[threadIdx.x] != INT32_MAX) {
        atomicMin(&posmm[((int64_t)blockIdx.y * NLV + threadIdx.x) * 2], smn[threadIdx.x]);
        atomicMax(&posmm[((int64_t)blockIdx.y * NLV + threadIdx.x) * 2 + 1], smx[threadIdx.x]);
    }
}

extern "C" int scp_geom_create(scp_geom **out) {
    if (!out) return SCP_EINVAL;
    *out = new (std::nothrow) scp_geom();
    return *out ? SCP_OK : SCP_ENOMEM;
}

extern "C" int scp_geom_destroy(scp_geom *g) {
    if (!g) return SCP_EINVAL;
    DevBuf *all[] = {&g->segtab, &g->keys_a, &g->keys_b, &g->red, &g->blkcnt, &g->leafkey, &g->leafoct, &g->occ, &g->level,
                     &g->octant, &g->parent, &g->pos, &g->fchild, &g->posmm, &g->radix.counts, &g->tr, &g->front, &g->rowtab};
    for (DevBuf *b : all) b->release();
    if (g->h2d_done) (void)hipEventDestroy(g->h2d_done);
    delete g;
    return SCP_OK;
}

static inline int grid_for(int64_t n) { int64_t b = cdiv64(n > 0 ? n : 1, WG); return (int)(b > 2048 ? 2048 : b); }


void scp_launch_scan_u32(uint32_t *counts, int64_t m, hipStream_t st);  // sort_u64.hip

// everything behind the keys: sort, head counts, node layout (the build's one size hand-back to the host), node tables.
// g->segs[s] holds pt_count / key_begin / path / drop_last / depth, the device SegTab is current, keys_a holds the unsorted keys.
static int geom_build_sorted(scp_geom *g, scp_segment_info *info, int dmax, hipStream_t st, bool first_hist_done) {
    const int nseg = g->nseg;
    const int64_t nk = g->n_keys;
    SegTab *dtab = g->segtab.as<SegTab>();
    g->lmax = dmax + 1;
    const int lmax = g->lmax;
    int rc;
    int plo[16], pnb[16], np = 0;
    for (int b = 0; b < 3 * dmax; b += 8) { plo[np] = b; pnb[np] = (3 * dmax - b) < 8 ? (3 * dmax - b) : 8; ++np; }
    plo[np] = SEG_SHIFT; pnb[np] = 6; ++np;  // segment id (and the all-ones sentinel of filtered points) last
    rc = scp_radix_sort_u64(g->keys_a.as<uint64_t>(), g->keys_b.as<uint64_t>(), nk, plo, pnb, np, &g->radix, st, &g->sorted, first_hist_done);
    if (rc) return rc;

    // --- head counts per level, scanned ---------------------------------------------------------------------
    const int nblk = (int)cdiv64(nk, WG);
    g->ntiles = nblk;
    if ((rc = g->blkcnt.reserve(sizeof(uint32_t) * (size_t)(lmax + 2) * nblk + 64))) return rc;
    uint32_t *blk = g->blkcnt.as<uint32_t>();
    hipLaunchKernelGGL(tree_count_kernel, dim3(nblk), dim3(WG), 0, st, (const uint64_t *)g->sorted, nk, (const SegTab *)dtab, lmax, blk, nblk);
    LAUNCH_CHECK();
    scp_launch_scan_u32(blk, (int64_t)(lmax + 2) * nblk, st);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(tree_segrank_kernel, dim3(nblk), dim3(WG), 0, st, (const uint64_t *)g->sorted, nk, dtab, lmax, (const uint32_t *)blk, nblk);
    LAUNCH_CHECK();
    std::vector<SegTab> back(nseg);
    std::vector<uint32_t> lvl_first(lmax + 2, 0u);
    SCP_D2H_TRY(scp_d2h_async(back.data(), dtab, sizeof(SegTab) * nseg, st));
    SCP_D2H_TRY(scp_d2h_2d_async(lvl_first.data(), blk, (size_t)nblk * 4, 4, lmax + 2, st));
    SCP_D2H_TRY(scp_stream_wait(st));

    // --- per-segment level tables (host, tiny) --------------------------------------------------------------
    int64_t node_base = 0, leaf_base = 0;
    for (int s = 0; s < nseg; ++s) {
        SegTab &t = g->segs[s];
        if (back[s].rank0[1] < 0) return SCP_EINVAL;  // every point filtered out: the reference raises too
        int nxt = s + 1;
        const int D = t.depth;
        int64_t off = 0;
        t.level_off[0] = 0;
        for (int L = 1; L <= lmax; ++L) {
            t.rank0[L] = back[s].rank0[L];
            const int64_t end = (nxt < nseg) ? back[nxt].rank0[L] : (int64_t)lvl_first[L + 1];
            const int64_t cnt = end - back[s].rank0[L];
            if (L <= D) { t.level_off[L] = off; off += cnt; info[s].level_count[L - 1] = cnt; }
            if (L == D + 1) { t.level_off[L] = off; t.n_nodes = off; t.n_leaves = cnt; t.level_off[L + 1] = off + cnt; }
        }
        t.node_base = node_base; t.leaf_base = leaf_base;
        info[s].n_nodes = t.n_nodes; info[s].n_leaves = t.n_leaves; info[s].node_base = node_base;
        node_base += t.n_nodes; leaf_base += t.n_leaves;
    }
    g->total_nodes = node_base; g->total_leaves = leaf_base;
    if (node_base > 0x7fffff00ll) return SCP_EINVAL;

    // --- node tables ----------------------------------------------------------------------------------------
    const size_t N = (size_t)node_base, U = (size_t)leaf_base;
    if ((rc = g->occ.reserve(N)) || (rc = g->level.reserve(N)) || (rc = g->octant.reserve(N)) || (rc = g->parent.reserve(N * 4)) ||
        (rc = g->pos.reserve(N * 12)) || (rc = g->fchild.reserve(N * 4 + 4)) || (rc = g->leafkey.reserve(U * 8)) ||
        (rc = g->leafoct.reserve(U)) || (rc = g->posmm.reserve(sizeof(int32_t) * 2 * NLV * nseg)))
        return rc;
    std::vector<int32_t> &mm = g->mm_host;
    mm.resize((size_t)2 * NLV * nseg);
    for (size_t k = 0; k < mm.size(); k += 2) { mm[k] = INT32_MAX; mm[k + 1] = INT32_MIN; }
    HIP_TRY(hipMemcpyAsync(g->posmm.p, mm.data(), mm.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dtab, g->segs.data(), sizeof(SegTab) * nseg, hipMemcpyHostToDevice, st));
    {
        std::vector<int64_t> &rt = g->rowtab_host;
        rt.resize(2 * (size_t)nseg);
        int64_t rows = 0, mmrows = 0;
        for (int s = 0; s < nseg; ++s) {
            rt[s] = rows; rt[nseg + s] = mmrows;
            rows += g->segs[s].n_nodes - g->segs[s].drop_last; mmrows += g->segs[s].depth;
        }
        if ((rc = g->rowtab.reserve(sizeof(int64_t) * 2 * nseg))) return rc;
        HIP_TRY(hipMemcpyAsync(g->rowtab.p, rt.data(), rt.size() * 8, hipMemcpyHostToDevice, st));
    }
    if (!g->h2d_done) HIP_TRY(hipEventCreateWithFlags(&g->h2d_done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(g->h2d_done, st));
    hipLaunchKernelGGL(tree_write_kernel, dim3(nblk), dim3(WG), 0, st, (const uint64_t *)g->sorted, nk, (const SegTab *)dtab, lmax,
                       (const uint32_t *)blk, nblk, g->level.as<uint8_t>(), g->octant.as<uint8_t>(), g->parent.as<int32_t>(),
                       g->pos.as<int32_t>(), g->fchild.as<int32_t>(), g->leafkey.as<uint64_t>(), g->leafoct.as<uint8_t>());
    LAUNCH_CHECK();
    int64_t maxnodes = 0;
    for (int s = 0; s < nseg; ++s) if (g->segs[s].n_nodes > maxnodes) maxnodes = g->segs[s].n_nodes;
    hipLaunchKernelGGL(tree_occ_kernel, dim3(std::min(grid_for(maxnodes), 256), nseg), dim3(WG), 0, st, (const SegTab *)dtab, g->level.as<uint8_t>(),
                       g->octant.as<uint8_t>(), g->fchild.as<int32_t>(), g->leafoct.as<uint8_t>(), g->pos.as<int32_t>(), g->occ.as<uint8_t>(),
                       g->posmm.as<int32_t>());
    LAUNCH_CHECK();
    // no trailing synchronisation: the two host tables copied above are members of the handle and stay untouched until the next build,
    // which first waits for h2d_done (long signalled by then); the node tables are ordered behind this stream like any kernel output
    g->built = true;
    return SCP_OK;
}

static int geom_take_segments(scp_geom *g, const scp_segment *segs, int32_t nseg, int64_t n, int64_t *maxcount) {
    if (g->h2d_done) HIP_TRY(hipEventSynchronize(g->h2d_done));   // the previous build's table copies have read g->segs / g->mm_host
    g->built = false;
    g->nseg = nseg;
    g->segs.assign(nseg, SegTab());
    int64_t nk = 0;
    *maxcount = 0;
    for (int s = 0; s < nseg; ++s) {
        SegTab &t = g->segs[s];
        memset(&t, 0, sizeof(t));
        if (segs[s].point_begin < 0 || segs[s].point_count <= 0 || segs[s].point_begin + segs[s].point_count > n ||
            segs[s].path_len < 0 || segs[s].path_len > 8)
            return SCP_EINVAL;
        t.pt_begin = segs[s].point_begin; t.pt_count = segs[s].point_count; t.key_begin = nk;
        t.path_len = segs[s].path_len; t.path_bits = segs[s].path_bits; t.drop_last = segs[s].drop_last ? 1 : 0;
        for (int L = 0; L < NLV; ++L) t.rank0[L] = -1;  // marker: "segment owns no key"
        nk += t.pt_count;
        if (t.pt_count > *maxcount) *maxcount = t.pt_count;
    }
    g->n_keys = nk;
    int rc;
    if ((rc = g->segtab.reserve(sizeof(SegTab) * nseg))) return rc;
    if ((rc = g->red.reserve(sizeof(int32_t) * 8 * (nseg + 1) + 64))) return rc;
    if ((rc = g->keys_a.reserve(sizeof(uint64_t) * nk))) return rc;
    if ((rc = g->keys_b.reserve(sizeof(uint64_t) * nk))) return rc;
    return SCP_OK;
}

static inline int depth_of(int32_t mx) {
    int d = 0;
    while (((int64_t)1 << d) < (int64_t)mx + 1) ++d;
    return d;
}

extern "C" int scp_geom_build(scp_geom *g, const int32_t *q, int64_t n, const scp_segment *segs, int32_t nseg,
                              scp_segment_info *info, void *stream) {
    if (!g || !q || !segs || !info || n <= 0 || nseg <= 0 || nseg > SCP_MAX_SEGMENTS) return SCP_EINVAL;
    scp_d2h_abort();                           // nothing a failed earlier call of this thread queued may land in this call's buffers
    hipStream_t st = (hipStream_t)stream;
    int64_t maxcount = 0;
    int rc = geom_take_segments(g, segs, nseg, n, &maxcount);
    if (rc) return rc;
    SegTab *dtab = g->segtab.as<SegTab>();

    // --- depth of every segment (Octree.py:58) --------------------------------------------------------------
    std::vector<int32_t> red(2 * nseg);
    for (int s = 0; s < nseg; ++s) { red[2 * s] = INT32_MIN; red[2 * s + 1] = INT32_MAX; }
    HIP_TRY(hipMemcpyAsync(g->red.p, red.data(), red.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dtab, g->segs.data(), sizeof(SegTab) * nseg, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(seg_minmax_kernel, dim3(std::min(grid_for(maxcount * 3), 64), nseg), dim3(WG), 0, st, q, (const SegTab *)dtab, g->red.as<int32_t>());
    LAUNCH_CHECK();
    SCP_D2H_TRY(scp_d2h_async(red.data(), g->red.p, red.size() * 4, st));
    SCP_D2H_TRY(scp_stream_wait(st));
    int dmax = 0;
    for (int s = 0; s < nseg; ++s) {
        const int32_t mx = red[2 * s], mn = red[2 * s + 1];
        if (mn < 0) return SCP_EINVAL;
        const int d = depth_of(mx);
        // depth 0 (all-zero cloud) aborts inside the reference as well
        if (d == 0 || d > SCP_MAX_DEPTH - 2 || g->segs[s].path_len > d) return SCP_EINVAL;
        g->segs[s].depth = d;
        if (d > dmax) dmax = d;
        memset(&info[s], 0, sizeof(info[s]));
        info[s].depth = d;
        info[s].max_coord = mx;
    }

    // --- keys -----------------------------------------------------------------------------------------------
    HIP_TRY(hipMemcpyAsync(dtab, g->segs.data(), sizeof(SegTab) * nseg, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(morton_key_kernel, dim3(grid_for(maxcount), nseg), dim3(WG), 0, st, q, (const SegTab *)dtab, g->keys_a.as<uint64_t>());
    LAUNCH_CHECK();
    return geom_build_sorted(g, info, dmax, st, false);
}

// ------------------------------------------------------------------------------------------------ G1 + G2 in one launch sequence
// scp_geom_build_xyz: float frames in, node tables out.  Two kernels in front of the sort instead of (transform, quantise) per shell +
// concatenation + min / max + key kernel, every point transformed ONCE (float64 atan2 / acos are the expensive part of stage G):
//   front_transform_kernel  xyz -> (rho, phi, theta | z) float32 for every frame, per-frame max / min of every axis (6 ordered words);
//                           ONE read-back: the maxima fix bin_num, the steps, and - the quantiser being monotone - the largest integer of
//                           every shell, hence every tree's depth, without another pass over the points;
//   front_key_kernel        one workgroup per 4096-key tile of the (frame, shell)-major key array: quantise the point for ITS shell, apply
//                           the rho-shell filter, write the 64-bit key - and the sort's first digit histogram of the tile, which
//                           radix_hist_kernel would otherwise re-read the keys for.
struct FrontFrame { const float *xyz; int64_t n; int64_t tr_begin; };
struct FrontSeg { int64_t key_begin, pt_count, tr_begin; QuantParams qp; int32_t depth, path_len, path_bits, pad; };

__global__ __launch_bounds__(WG) void front_transform_kernel(const FrontFrame *__restrict__ frames, int mode, float *__restrict__ tr,
                                                            uint32_t *__restrict__ red /* [frame][8]: max a, b, c, min a, b, c (ordered) */) {
    const FrontFrame f = frames[blockIdx.y];
    uint32_t omax[3] = {0u, 0u, 0u}, omin[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
    for (int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x; i < f.n; i += (int64_t)gridDim.x * WG) {
        const float x = f.xyz[3 * i], y = f.xyz[3 * i + 1], z = f.xyz[3 * i + 2];
        float a, b, c;
        if (mode == SCP_CART) {
            a = x; b = y; c = z;
        } else {       // the arithmetic of transform_kernel, line for line
            const float xx = __fmul_rn(x, x), yy = __fmul_rn(y, y);
            float s = __fadd_rn(xx, yy);
            if (mode == SCP_SPHER) s = __fadd_rn(s, __fmul_rn(z, z));
            a = sqrt_cr(s);
            const float xe = __fadd_rn(x, 1e-9f);
            float phi = (float)atan2((double)y, (double)xe);
            if (phi < 0.f) phi = __fadd_rn(phi, 6.2831855f);
            b = phi;
            c = (mode == SCP_SPHER) ? (float)acos((double)(float)((double)z / (double)a)) : z;
        }
        float *o = tr + 3 * (f.tr_begin + i);
        o[0] = a; o[1] = b; o[2] = c;
        const uint32_t oa = f2ord(a), ob = f2ord(b), oc = f2ord(c);
        omax[0] = max(omax[0], oa); omax[1] = max(omax[1], ob); omax[2] = max(omax[2], oc);
        omin[0] = min(omin[0], oa); omin[1] = min(omin[1], ob); omin[2] = min(omin[2], oc);
    }
    __shared__ uint32_t sm[4][6];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        for (int off = 32; off > 0; off >>= 1) {
            omax[k] = max(omax[k], (uint32_t)__shfl_xor(omax[k], off));
            omin[k] = min(omin[k], (uint32_t)__shfl_xor(omin[k], off));
        }
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; ++k) { sm[threadIdx.x >> 6][k] = omax[k]; sm[threadIdx.x >> 6][3 + k] = omin[k]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        uint32_t v = sm[0][k];
        for (int w = 1; w < 4; ++w) v = k < 3 ? max(v, sm[w][k]) : min(v, sm[w][k]);
        if (k < 3) atomicMax(&red[8 * blockIdx.y + k], v); else atomicMin(&red[8 * blockIdx.y + k], v);
    }
}

// the quantiser of quantize_kernel for one coordinate (same expressions: identical integers)
__device__ __forceinline__ int32_t quant1(float t, const QuantParams &p, int k) {
    double r;
    if (p.cart_f32) r = (double)rintf((float)((double)__fsub_rn(t, p.offf) / (double)p.qsf));
    else r = rint(((double)t - p.off[k]) / p.qs[k]);
    return (int32_t)r;
}
static inline int32_t quant1_host(float t, const QuantParams &p, int k) {      // the same arithmetic on the host (IEEE, no contraction: one operation per statement)
    if (p.cart_f32) {
        const float d = t - p.offf;
        const float qf = (float)((double)d / (double)p.qsf);
        return (int32_t)(double)rintf(qf);
    }
    const double d = (double)t - p.off[k];
    const double qd = d / p.qs[k];
    return (int32_t)rint(qd);
}

__global__ __launch_bounds__(WG) void front_key_kernel(const float *__restrict__ tr, const FrontSeg *__restrict__ segs, int nseg, int64_t nk,
                                                      uint64_t *__restrict__ keys, int32_t *__restrict__ q_out /* optional [nk][3] */,
                                                      uint32_t mask0, uint32_t *__restrict__ counts, int ntiles) {
    __shared__ uint32_t hist[256];
    __shared__ FrontSeg ss[2];           // a 4096-key tile touches at most two segments unless segments are tiny: others are fetched from memory
    hist[threadIdx.x] = 0;
    const int64_t base = (int64_t)blockIdx.x * SCP_RADIX_TILE;
    // first segment of the tile (segments are in key order): the last one whose key_begin <= base
    int s0 = 0;
    for (int s = 1; s < nseg; ++s) if (segs[s].key_begin <= base) s0 = s;
    if (threadIdx.x < 2 && s0 + (int)threadIdx.x < nseg) ss[threadIdx.x] = segs[s0 + threadIdx.x];
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < SCP_RADIX_TILE / WG; ++r) {
        const int64_t k = base + r * WG + threadIdx.x;
        if (k >= nk) break;
        int si = 0;
        const FrontSeg *sp = &ss[0];
        if (s0 + 1 < nseg && k >= ss[1].key_begin) {
            si = 1; sp = &ss[1];
            if (k >= ss[1].key_begin + ss[1].pt_count) {        // tiny segments: walk on in memory
                int s = s0 + 2;
                while (s + 1 < nseg && k >= segs[s + 1].key_begin) ++s;
                si = s - s0; sp = &segs[s];
            }
        }
        const FrontSeg &sg = *sp;
        const int64_t i = k - sg.key_begin;
        const float *t = tr + 3 * (sg.tr_begin + i);
        const int32_t qx = quant1(t[0], sg.qp, 0), qy = quant1(t[1], sg.qp, 1), qz = quant1(t[2], sg.qp, 2);
        if (q_out) { q_out[3 * k] = qx; q_out[3 * k + 1] = qy; q_out[3 * k + 2] = qz; }
        uint64_t key = ((uint64_t)(s0 + si) << SEG_SHIFT) | (spread3((uint32_t)qx) << 2) | (spread3((uint32_t)qy) << 1) | spread3((uint32_t)qz);
        if (sg.path_len > 0 && (int)((uint32_t)qx >> (sg.depth - sg.path_len)) != sg.path_bits) key = ~0ull;
        keys[k] = key;
        atomicAdd(&hist[(uint32_t)key & mask0], 1u);
    }
    __syncthreads();
    counts[(int64_t)blockIdx.x * 256 + threadIdx.x] = hist[threadIdx.x];       // [tile][digit], the layout of radix_hist_kernel
}

// frames: nframes device arrays float32 [n_points[f]][3]; every frame is cut into nshell trees (shell s: step qs[s], rho-shell path of
// shells[s], drop_last of shells[s]; point_begin / point_count of shells[] are ignored), segment index = f * nshell + s.
// qinfo / info: host arrays [nframes * nshell].  q_out (optional): device int32 [nframes * nshell * n][3] = the integers of every segment.
extern "C" int scp_geom_build_xyz(scp_geom *g, const float *const *frames, const int64_t *n_points, int32_t nframes, int32_t mode,
                                  const double *qs, int32_t nshell, double cart_offset, const scp_segment *shells, int32_t *q_out,
                                  scp_quant_info *qinfo, scp_segment_info *info, void *stream) {
    if (!g || !frames || !n_points || !qs || !shells || !qinfo || !info || nframes <= 0 || nshell <= 0 || mode < 0 || mode > 2 ||
        (int64_t)nframes * nshell > SCP_MAX_SEGMENTS)
        return SCP_EINVAL;
    scp_d2h_abort();
    const int nseg = nframes * nshell;
    hipStream_t st = (hipStream_t)stream;
    int64_t npts = 0, maxn = 0;
    std::vector<scp_segment> segv(nseg);
    std::vector<FrontFrame> ff(nframes);
    for (int f = 0; f < nframes; ++f) {
        if (!frames[f] || n_points[f] <= 0 || ((uintptr_t)frames[f] & 3)) return SCP_EINVAL;
        ff[f].xyz = frames[f]; ff[f].n = n_points[f]; ff[f].tr_begin = npts;
        for (int s = 0; s < nshell; ++s) {
            if (!(qs[s] > 0)) return SCP_EINVAL;
            scp_segment &sg = segv[f * nshell + s];
            sg = shells[s];
            sg.point_begin = 0; sg.point_count = n_points[f];
        }
        npts += n_points[f];
        if (n_points[f] > maxn) maxn = n_points[f];
    }
    int64_t maxcount = 0;
    int rc = geom_take_segments(g, segv.data(), nseg, maxn, &maxcount);
    if (rc) return rc;
    if ((rc = g->tr.reserve((size_t)npts * 12 + 64))) return rc;
    if ((rc = g->front.reserve(sizeof(FrontFrame) * nframes + sizeof(FrontSeg) * nseg + 256))) return rc;
    FrontFrame *dff = g->front.as<FrontFrame>();
    FrontSeg *dfs = (FrontSeg *)((char *)g->front.p + ((sizeof(FrontFrame) * nframes + 63) & ~(size_t)63));
    uint32_t *dred = g->red.as<uint32_t>();
    std::vector<uint32_t> red((size_t)8 * nframes);
    for (int f = 0; f < nframes; ++f)
        for (int k = 0; k < 8; ++k) red[8 * f + k] = (k < 3) ? 0u : 0xffffffffu;
    HIP_TRY(hipMemcpyAsync(dred, red.data(), red.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dff, ff.data(), sizeof(FrontFrame) * nframes, hipMemcpyHostToDevice, st));
    {
        SCP_PROF(SCP_PROF_GEOM, st, 24.0 * npts);
        hipLaunchKernelGGL(front_transform_kernel, dim3(std::min(grid_for(maxn), 512), nframes), dim3(WG), 0, st, (const FrontFrame *)dff, mode,
                           g->tr.as<float>(), dred);
    }
    LAUNCH_CHECK();
    SCP_D2H_TRY(scp_d2h_async(red.data(), dred, red.size() * 4, st));
    SCP_D2H_TRY(scp_stream_wait(st));                 // read-back 1 of 2: the extrema fix the steps and the depths

    std::vector<FrontSeg> fs(nseg);
    int dmax = 0;
    for (int f = 0; f < nframes; ++f) {
        float tmax[3], tmin[3];
        for (int k = 0; k < 3; ++k) { tmax[k] = ord2f_host(red[8 * f + k]); tmin[k] = ord2f_host(red[8 * f + 3 + k]); }
        for (int s = 0; s < nshell; ++s) {
            const int sidx = f * nshell + s;
            QuantParams p;
            memset(&p, 0, sizeof(p));
            scp_quant_info &qi = qinfo[sidx];
            memset(&qi, 0, sizeof(qi));
            if (mode == SCP_CART) {
                p.cart_f32 = 1; p.qsf = (float)qs[s]; p.offf = (float)cart_offset;
                for (int k = 0; k < 3; ++k) { qi.qs[k] = qs[s]; qi.offset[k] = cart_offset; }
            } else {               // scp_quantize's float32 expressions
                const float binf = rintf(tmax[0] / (float)qs[s]) + 1.0f;
                const float q_phi = 6.2831855f / (binf - 1.0f);
                const float q_th = 3.1415927f / (binf - 1.0f);
                qi.bin_num = (double)binf;
                p.qs[0] = qs[s]; p.qs[1] = (double)q_phi; p.qs[2] = (mode == SCP_SPHER) ? (double)q_th : qs[s];
                p.off[2] = (mode == SCP_CYLIN) ? (double)tmin[2] : 0.0;
                for (int k = 0; k < 3; ++k) { qi.qs[k] = p.qs[k]; qi.offset[k] = p.off[k]; }
            }
            // the quantiser is monotone in every coordinate: the extreme integers are the integers of the extreme coordinates
            int32_t mx = INT32_MIN, mn = INT32_MAX;
            for (int k = 0; k < 3; ++k) {
                const int32_t hi = quant1_host(tmax[k], p, k), lo = quant1_host(tmin[k], p, k);
                mx = std::max(mx, hi); mn = std::min(mn, lo);
            }
            qi.max_coord = mx; qi.min_coord = mn;
            if (mn < 0) return SCP_EINVAL;
            const int d = depth_of(mx);
            if (d == 0 || d > SCP_MAX_DEPTH - 2 || g->segs[sidx].path_len > d) return SCP_EINVAL;
            g->segs[sidx].depth = d;
            if (d > dmax) dmax = d;
            memset(&info[sidx], 0, sizeof(info[sidx]));
            info[sidx].depth = d;
            info[sidx].max_coord = mx;
            FrontSeg &o = fs[sidx];
            o.key_begin = g->segs[sidx].key_begin; o.pt_count = n_points[f]; o.tr_begin = ff[f].tr_begin; o.qp = p;
            o.depth = d; o.path_len = g->segs[sidx].path_len; o.path_bits = g->segs[sidx].path_bits; o.pad = 0;
        }
    }
    HIP_TRY(hipMemcpyAsync(dfs, fs.data(), sizeof(FrontSeg) * nseg, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(g->segtab.p, g->segs.data(), sizeof(SegTab) * nseg, hipMemcpyHostToDevice, st));
    int ntiles = 0;
    uint32_t *counts = scp_radix_counts(&g->radix, g->n_keys, &ntiles);
    if (!counts) return SCP_ENOMEM;
    const int nb0 = 3 * dmax < 8 ? 3 * dmax : 8;
    {
        SCP_PROF(SCP_PROF_GEOM, st, 12.0 * g->n_keys + 8.0 * g->n_keys);
        hipLaunchKernelGGL(front_key_kernel, dim3(ntiles), dim3(WG), 0, st, g->tr.as<float>(), (const FrontSeg *)dfs, nseg, g->n_keys,
                           g->keys_a.as<uint64_t>(), q_out, (1u << nb0) - 1u, counts, ntiles);
    }
    LAUNCH_CHECK();
    return geom_build_sorted(g, info, dmax, st, g->n_keys > 1);
}

#define REQUIRE_BUILT(g) do { if (!(g)) return SCP_EINVAL; if (!(g)->built) return SCP_ESTATE; } while (0)

extern "C" int scp_geom_emit_nodes(scp_geom *g, uint8_t *occ, uint8_t *level, uint8_t *octant, int32_t *parent, int32_t *pos,
                                   void *stream) {
    REQUIRE_BUILT(g);
    hipStream_t st = (hipStream_t)stream;
    const size_t N = (size_t)g->total_nodes;
    if (occ) HIP_TRY(hipMemcpyAsync(occ, g->occ.p, N, hipMemcpyDeviceToDevice, st));
    if (level) HIP_TRY(hipMemcpyAsync(level, g->level.p, N, hipMemcpyDeviceToDevice, st));
    if (octant) HIP_TRY(hipMemcpyAsync(octant, g->octant.p, N, hipMemcpyDeviceToDevice, st));
    if (parent) HIP_TRY(hipMemcpyAsync(parent, g->parent.p, N * 4, hipMemcpyDeviceToDevice, st));
    if (pos) HIP_TRY(hipMemcpyAsync(pos, g->pos.p, N * 12, hipMemcpyDeviceToDevice, st));
    return SCP_OK;
}

__global__ __launch_bounds__(WG) void leaves_kernel(const uint64_t *__restrict__ leafkey, int64_t base, int64_t n, int32_t *__restrict__ pts) {
    const int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x;
    if (i >= n) return;
    const uint64_t m = leafkey[base + i] & ((1ull << SEG_SHIFT) - 1ull);
    pts[3 * i] = (int32_t)compact3(m >> 2); pts[3 * i + 1] = (int32_t)compact3(m >> 1); pts[3 * i + 2] = (int32_t)compact3(m);
}

extern "C" int scp_geom_emit_leaves(scp_geom *g, int32_t seg, int32_t *pts, void *stream) {
    REQUIRE_BUILT(g);
    if (seg < 0 || seg >= g->nseg || !pts) return SCP_EINVAL;
    const SegTab &t = g->segs[seg];
    hipLaunchKernelGGL(leaves_kernel, dim3((int)cdiv64(t.n_leaves, WG)), dim3(WG), 0, (hipStream_t)stream, g->leafkey.as<uint64_t>(),
                       t.leaf_base, t.n_leaves, pts);
    LAUNCH_CHECK();
    return SCP_OK;
}

// ------------------------------------------------------------------------------------------------ G3: K=4 ancestor gathers
// ancestors of node nd inside its segment: a[3] = self, a[2] = parent, a[1] = grandparent, a[0] = great-grandparent (-1 = none)
__device__ __forceinline__ void ancestors(const int32_t *__restrict__ parent, int64_t nd, int64_t a[4]) {
    a[3] = nd;
    a[2] = parent[nd];
    a[1] = a[2] >= 0 ? parent[a[2]] : -1;
    a[0] = a[1] >= 0 ? parent[a[1]] : -1;
}

// reference record, data_preprocess.py:74: int64 [rows][4][6] = (occ, level, octant, x, y, z); pad rows (256,0,0,0,0,0)
__global__ __launch_bounds__(WG) void krecords_kernel(const uint8_t *__restrict__ occ, const uint8_t *__restrict__ level,
                                                     const uint8_t *__restrict__ octant, const int32_t *__restrict__ parent,
                                                     const int32_t *__restrict__ pos, int64_t base, int64_t rows, int64_t *__restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * WG + threadIdx.x;
    if (r >= rows) return;
    int64_t a[4];
    ancestors(parent, base + r, a);
    int64_t *o = out + r * 24;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t nd = a[k];
        if (nd < 0) { o[6 * k] = 256; o[6 * k + 1] = o[6 * k + 2] = o[6 * k + 3] = o[6 * k + 4] = o[6 * k + 5] = 0; }
        else {
            o[6 * k] = occ[nd]; o[6 * k + 1] = level[nd]; o[6 * k + 2] = octant[nd];
            o[6 * k + 3] = pos[3 * nd]; o[6 * k + 4] = pos[3 * nd + 1]; o[6 * k + 5] = pos[3 * nd + 2];
        }
    }
}

extern "C" int scp_geom_krecords_i64(scp_geom *g, int32_t seg, int64_t *out, void *stream) {
    REQUIRE_BUILT(g);
    if (seg < 0 || seg >= g->nseg || !out) return SCP_EINVAL;
    const SegTab &t = g->segs[seg];
    const int64_t rows = t.n_nodes - t.drop_last;
    if (rows <= 0) return SCP_OK;
    hipLaunchKernelGGL(krecords_kernel, dim3((int)cdiv64(rows, WG)), dim3(WG), 0, (hipStream_t)stream, g->occ.as<uint8_t>(),
                       g->level.as<uint8_t>(), g->octant.as<uint8_t>(), g->parent.as<int32_t>(), g->pos.as<int32_t>(), t.node_base, rows, out);
    LAUNCH_CHECK();
    return SCP_OK;
}

// EHEM context, encode_dataset_ehem.py:52-105: ctx = (level, octant, occ-1) x 4 rows, self position normalised per level
__global__ __launch_bounds__(WG) void ctx_ehem_kernel(const uint8_t *__restrict__ occ, const uint8_t *__restrict__ level,
                                                     const uint8_t *__restrict__ octant, const int32_t *__restrict__ parent,
                                                     const int32_t *__restrict__ pos, const int32_t *__restrict__ posmm /* [NLV][2] of this segment */,
                                                     int64_t base, int64_t rows, int depth, int pos_mode, int lidar_level,
                                                     uint8_t *__restrict__ ctx, float *__restrict__ posn, uint8_t *__restrict__ sym) {
    const int64_t r = (int64_t)blockIdx.x * WG + threadIdx.x;
    if (r >= rows) return;
    int64_t a[4];
    ancestors(parent, base + r, a);
    const int L = level[base + r];
    const bool last = (L == depth);
    uint8_t c[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t nd = a[k];
        if (nd < 0) { c[3 * k] = 0; c[3 * k + 1] = 0; c[3 * k + 2] = 255; }
        else {
            int lv = level[nd];
            if (last && lv > lidar_level) lv = lidar_level;  // ehem:86 clips the whole last chunk
            c[3 * k] = (uint8_t)lv; c[3 * k + 1] = octant[nd]; c[3 * k + 2] = (uint8_t)(occ[nd] - 1);
        }
    }
    uint32_t *cw = (uint32_t *)(ctx + r * 12);
    cw[0] = c[0] | (c[1] << 8) | (c[2] << 16) | ((uint32_t)c[3] << 24);
    cw[1] = c[4] | (c[5] << 8) | (c[6] << 16) | ((uint32_t)c[7] << 24);
    cw[2] = c[8] | (c[9] << 8) | (c[10] << 16) | ((uint32_t)c[11] << 24);
    if (sym) sym[r] = (uint8_t)(occ[base + r] - 1);
    if (posn) {
        const int64_t nd = base + r;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double p = (double)pos[3 * nd + k];
            double v;
            if (pos_mode == SCP_POS_POW2) v = p / (double)(1ll << depth);
            else {
                const double mn = (double)posmm[2 * L], mx = (double)posmm[2 * L + 1];
                const double eps = (pos_mode == SCP_POS_MINMAX_MUL && last) ? 0.0 : 1e-9;
                v = (p - mn) / (mx - mn + eps);
            }
            posn[3 * r + k] = (float)v;
        }
    }
}

__global__ void posmm_i64_kernel(const int32_t *__restrict__ posmm, int depth, int64_t *__restrict__ out) {
    const int L = threadIdx.x + 1;
    if (L <= depth) { out[2 * (L - 1)] = posmm[2 * L]; out[2 * (L - 1) + 1] = posmm[2 * L + 1]; }
}

extern "C" int scp_geom_context_ehem(scp_geom *g, int32_t seg, int32_t pos_mode, int32_t lidar_level, uint8_t *ctx, float *pos,
                                     uint8_t *sym, int64_t *pos_mm, void *stream) {
    REQUIRE_BUILT(g);
    if (seg < 0 || seg >= g->nseg || !ctx || pos_mode < 0 || pos_mode > 2) return SCP_EINVAL;
    if (((uintptr_t)ctx & 3) != 0) return SCP_EINVAL;
    const SegTab &t = g->segs[seg];
    const int64_t rows = t.n_nodes - t.drop_last;
    hipStream_t st = (hipStream_t)stream;
    const int32_t *mm = g->posmm.as<int32_t>() + (size_t)seg * NLV * 2;
    if (rows > 0) {
        hipLaunchKernelGGL(ctx_ehem_kernel, dim3((int)cdiv64(rows, WG)), dim3(WG), 0, st, g->occ.as<uint8_t>(), g->level.as<uint8_t>(),
                           g->octant.as<uint8_t>(), g->parent.as<int32_t>(), g->pos.as<int32_t>(), mm, t.node_base, rows, t.depth, pos_mode,
                           lidar_level, ctx, pos, sym);
        LAUNCH_CHECK();
    }
    if (pos_mm) {
        hipLaunchKernelGGL(posmm_i64_kernel, dim3(1), dim3(64), 0, st, mm, t.depth, pos_mm);
        LAUNCH_CHECK();
    }
    return SCP_OK;
}

// The same for EVERY segment of the build in one launch, rows of all segments back to back (segment s starts at row_base[s] = the sum of
// the earlier segments' n_nodes - drop_last), and with the coded symbols written straight in CODING ORDER (encode.py:109-136: the rows
// of a level are cut into windows of context_size, inside a window all even positions come first, then the odd ones): the encoder needs
// no per-segment launches, no concatenation and no gather through a coding-order index.
__global__ __launch_bounds__(WG) void ctx_ehem_all_kernel(const SegTab *__restrict__ tab, const int64_t *__restrict__ row_base, const uint8_t *__restrict__ occ,
                                                         const uint8_t *__restrict__ level, const uint8_t *__restrict__ octant,
                                                         const int32_t *__restrict__ parent, const int32_t *__restrict__ pos,
                                                         const int32_t *__restrict__ posmm, int pos_mode, int lidar_level, int cs,
                                                         uint8_t *__restrict__ ctx, float *__restrict__ posn, uint8_t *__restrict__ sym_coded,
                                                         int64_t *__restrict__ pos_mm_out, const int64_t *__restrict__ mm_base) {
    const SegTab &s = tab[blockIdx.y];
    const int64_t rows = s.n_nodes - s.drop_last, rb = row_base[blockIdx.y];
    const int depth = s.depth;
    const int32_t *mm = posmm + (size_t)blockIdx.y * NLV * 2;
    if (pos_mm_out && blockIdx.x == 0 && (int)threadIdx.x < depth) {
        const int L = threadIdx.x + 1;
        pos_mm_out[2 * (mm_base[blockIdx.y] + L - 1)] = mm[2 * L];
        pos_mm_out[2 * (mm_base[blockIdx.y] + L - 1) + 1] = mm[2 * L + 1];
    }
    for (int64_t r = (int64_t)blockIdx.x * WG + threadIdx.x; r < rows; r += (int64_t)gridDim.x * WG) {
        const int64_t nd = s.node_base + r;
        int64_t a[4];
        ancestors(parent, nd, a);
        const int L = level[nd];
        const bool last = (L == depth);
        uint8_t c[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t an = a[k];
            if (an < 0) { c[3 * k] = 0; c[3 * k + 1] = 0; c[3 * k + 2] = 255; }
            else {
                int lv = level[an];
                if (last && lv > lidar_level) lv = lidar_level;
                c[3 * k] = (uint8_t)lv; c[3 * k + 1] = octant[an]; c[3 * k + 2] = (uint8_t)(occ[an] - 1);
            }
        }
        uint32_t *cw = (uint32_t *)(ctx + (rb + r) * 12);
        cw[0] = c[0] | (c[1] << 8) | (c[2] << 16) | ((uint32_t)c[3] << 24);
        cw[1] = c[4] | (c[5] << 8) | (c[6] << 16) | ((uint32_t)c[7] << 24);
        cw[2] = c[8] | (c[9] << 8) | (c[10] << 16) | ((uint32_t)c[11] << 24);
        if (sym_coded) {
            const int64_t l0 = s.level_off[L];
            const int64_t nl = s.level_off[L + 1] - l0 - ((last && s.drop_last) ? 1 : 0);      // coded rows of this level
            const int64_t i = r - l0, w0 = (i / cs) * cs, p = i - w0;
            const int64_t cw_len = (nl - w0) < cs ? (nl - w0) : cs, ne = (cw_len + 1) >> 1;
            sym_coded[rb + l0 + w0 + ((p & 1) ? ne + (p >> 1) : (p >> 1))] = (uint8_t)(occ[nd] - 1);
        }
        if (posn) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double pv = (double)pos[3 * nd + k];
                double v;
                if (pos_mode == SCP_POS_POW2) v = pv / (double)(1ll << depth);
                else {
                    const double mn = (double)mm[2 * L], mx = (double)mm[2 * L + 1];
                    const double eps = (pos_mode == SCP_POS_MINMAX_MUL && last) ? 0.0 : 1e-9;
                    v = (pv - mn) / (mx - mn + eps);
                }
                posn[3 * (rb + r) + k] = (float)v;
            }
        }
    }
}

extern "C" int scp_geom_context_ehem_all(scp_geom *g, int32_t pos_mode, int32_t lidar_level, int32_t context_size, uint8_t *ctx, float *pos,
                                         uint8_t *sym_coded, int64_t *pos_mm, void *stream) {
    REQUIRE_BUILT(g);
    if (!ctx || pos_mode < 0 || pos_mode > 2 || context_size <= 0 || ((uintptr_t)ctx & 3)) return SCP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nseg = g->nseg;
    int64_t rows = 0, maxrows = 1;
    for (int s = 0; s < nseg; ++s) {
        const int64_t r = g->segs[s].n_nodes - g->segs[s].drop_last;
        rows += r;
        if (r > maxrows) maxrows = r;
    }
    SCP_PROF(SCP_PROF_GEOM, st, 25.0 * rows);
    hipLaunchKernelGGL(ctx_ehem_all_kernel, dim3(grid_for(maxrows), nseg), dim3(WG), 0, st, (const SegTab *)g->segtab.p,
                       (const int64_t *)g->rowtab.p, g->occ.as<uint8_t>(), g->level.as<uint8_t>(), g->octant.as<uint8_t>(), g->parent.as<int32_t>(),
                       g->pos.as<int32_t>(), g->posmm.as<int32_t>(), pos_mode, lidar_level, context_size, ctx, pos, sym_coded, pos_mm,
                       (const int64_t *)g->rowtab.p + nseg);
    LAUNCH_CHECK();
    return SCP_OK;
}

// OctAttention context, encode_dataset.py:32-55: ctx = (occ-1, level, octant) x 4; pos = xyz / 2^D for all four rows
__global__ __launch_bounds__(WG) void ctx_octattn_kernel(const uint8_t *__restrict__ occ, const uint8_t *__restrict__ level,
                                                        const uint8_t *__restrict__ octant, const int32_t *__restrict__ parent,
                                                        const int32_t *__restrict__ pos, int64_t base, int64_t rows, int depth,
                                                        uint8_t *__restrict__ ctx, float *__restrict__ posn, uint8_t *__restrict__ sym) {
    const int64_t r = (int64_t)blockIdx.x * WG + threadIdx.x;
    if (r >= rows) return;
    int64_t a[4];
    ancestors(parent, base + r, a);
    const double scale = (double)(1ll << depth);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t nd = a[k];
        uint8_t *c = ctx + r * 12 + 3 * k;
        float *p = posn ? posn + r * 12 + 3 * k : nullptr;
        if (nd < 0) {
            c[0] = 255; c[1] = 0; c[2] = 0;
            if (p) { p[0] = 0.f; p[1] = 0.f; p[2] = 0.f; }
        } else {
            c[0] = (uint8_t)(occ[nd] - 1); c[1] = level[nd]; c[2] = octant[nd];
            if (p) { p[0] = (float)((double)pos[3 * nd] / scale); p[1] = (float)((double)pos[3 * nd + 1] / scale); p[2] = (float)((double)pos[3 * nd + 2] / scale); }
        }
    }
    if (sym) sym[r] = (uint8_t)(occ[base + r] - 1);
}

extern "C" int scp_geom_context_octattn(scp_geom *g, int32_t seg, uint8_t *ctx, float *pos, uint8_t *sym, void *stream) {
    REQUIRE_BUILT(g);
    if (seg < 0 || seg >= g->nseg || !ctx) return SCP_EINVAL;
    const SegTab &t = g->segs[seg];
    const int64_t rows = t.n_nodes - t.drop_last;
    if (rows <= 0) return SCP_OK;
    hipLaunchKernelGGL(ctx_octattn_kernel, dim3((int)cdiv64(rows, WG)), dim3(WG), 0, (hipStream_t)stream, g->occ.as<uint8_t>(),
                       g->level.as<uint8_t>(), g->octant.as<uint8_t>(), g->parent.as<int32_t>(), g->pos.as<int32_t>(), t.node_base, rows,
                       t.depth, ctx, pos, sym);
    LAUNCH_CHECK();
    return SCP_OK;
}
