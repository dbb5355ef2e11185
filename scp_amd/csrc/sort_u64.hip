// Stable LSD radix sort of 64-bit Morton keys (gfx950, wave64).
//
// One pass = histogram -> scatter (which derives its bucket starts from the raw [tile][digit] table itself).  A tile is 4096
// consecutive keys owned by one 256-thread workgroup; wave w of the workgroup owns the contiguous
// quarter [w*1024, (w+1)*1024) and walks it in 16 rounds of 64 keys, so a key's stable rank inside its
// (tile, digit) bucket is   sum over lower waves of their digit count  +  its rank inside its wave,
// and the rank inside a wave comes from a wavefront match-any built out of `__ballot` (64-bit masks)
// and per-wave LDS digit counters.  No inter-workgroup communication inside a launch.
#include "scp_internal.h"

#define TILE 4096
#define WG 256
#define ROUNDS 16  // TILE / WG * (WG/64) / 4 waves = 1024 keys per wave / 64

__global__ __launch_bounds__(WG) void radix_hist_kernel(const uint64_t *__restrict__ keys, int64_t n, int lo,
                                                       uint32_t mask, uint32_t *__restrict__ counts, int ntiles) {
    __shared__ uint32_t hist[256];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * TILE;
#pragma unroll 4
    for (int r = 0; r < TILE / WG; ++r) {
        int64_t i = base + r * WG + threadIdx.x;
        if (i < n) atomicAdd(&hist[(uint32_t)(keys[i] >> lo) & mask], 1u);
    }
    __syncthreads();
    counts[(int64_t)blockIdx.x * 256 + threadIdx.x] = hist[threadIdx.x];     // [tile][digit]: coalesced here and in the scatter's column sums
}

// exclusive scan of `m` uint32 counters (16-byte aligned), single workgroup of 1024 threads.  A thread owns a run of
// counters whose length is a multiple of 4 and walks it with 16-byte loads / stores, all loads of a pass in flight together (the scalar
// walk - one dependent 4-byte access after the other - took 38 us for the 30 k counters of a four-frame build, 8 us for one frame's).
// VEC = false: any alignment, one counter at a time.
template <bool VEC>
__global__ __launch_bounds__(1024) void radix_scan_kernel(uint32_t *__restrict__ counts, int64_t m) {
    __shared__ uint32_t part[1024];
    const int t = threadIdx.x;
    const int64_t chunk = VEC ? ((((m + 1023) / 1024) + 3) & ~(int64_t)3) : (m + 1023) / 1024;
    const int64_t b = (int64_t)t * chunk, e = (b + chunk < m) ? b + chunk : m;
    uint4 *c4 = (uint4 *)counts;
    uint32_t s = 0;
    const int64_t ev = VEC ? b + ((e - b) & ~(int64_t)3) : b;      // whole groups of four; the (last thread's) remainder one by one
    for (int64_t i = b; i < ev; i += 4) { const uint4 v = c4[i >> 2]; s += (v.x + v.y) + (v.z + v.w); }
    for (int64_t i = ev > b ? ev : b; i < e; ++i) s += counts[i];
    part[t] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partials
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = (t >= off) ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - s;
    for (int64_t i = b; i < ev; i += 4) {
        const uint4 v = c4[i >> 2];
        uint4 o;
        o.x = run; o.y = run + v.x; o.z = o.y + v.y; o.w = o.z + v.z;
        run = o.w + v.w;
        c4[i >> 2] = o;
    }
    for (int64_t i = ev > b ? ev : b; i < e; ++i) { const uint32_t c = counts[i]; counts[i] = run; run += c; }
}

__global__ __launch_bounds__(WG) void radix_scatter_kernel(const uint64_t *__restrict__ in, uint64_t *__restrict__ out,
                                                          int64_t n, int lo, int nb, const uint32_t *__restrict__ counts,
                                                          int ntiles) {
    __shared__ uint32_t wcnt[4][256];
    __shared__ uint32_t wtot[4];
    const int t = threadIdx.x, w = t >> 6, lane = t & 63;
    const uint32_t mask = (1u << nb) - 1u;
#pragma unroll
    for (int i = 0; i < 4; ++i) wcnt[i][t] = 0;
    __syncthreads();

    const int64_t wbase = (int64_t)blockIdx.x * TILE + (int64_t)w * (TILE / 4);
    uint64_t key[ROUNDS];
    uint32_t rank[ROUNDS];
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int64_t i = wbase + r * 64 + lane;
        const bool valid = i < n;
        key[r] = valid ? in[i] : 0ull;
        const uint32_t d = (uint32_t)(key[r] >> lo) & mask;
        uint64_t peers = __ballot(valid);
        for (int b = 0; b < nb; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t vote = __ballot(bit);
            peers &= bit ? vote : ~vote;
        }
        const uint32_t before = (uint32_t)__popcll(peers & lt);
        const uint32_t cnt = (uint32_t)__popcll(peers);
        uint32_t prev = 0;
        if (valid) prev = ((volatile uint32_t *)wcnt[w])[d];
        __builtin_amdgcn_wave_barrier();
        if (valid && before == 0) ((volatile uint32_t *)wcnt[w])[d] = prev + cnt;
        __builtin_amdgcn_wave_barrier();
        rank[r] = prev + before;
    }
    __syncthreads();
    // digit t: global start of this tile's (digit t) bucket, straight from the RAW [tile][digit] counts (round 4: no scan kernel between
    // the histogram and the scatter - a single-workgroup pass over 256 x tiles counters that cost 5 - 15 us per sort pass, a third of a
    // build's kernel time; here every workgroup re-reads the table out of L2, 120 KB for a four-frame build):
    //   start = sum over digits d' < t of (all tiles' counts of d')  +  sum over tiles t' < this one of counts[t'][t]
    {
        uint32_t tot = 0, pre = 0;
        const int me = (int)blockIdx.x;
#pragma unroll 8
        for (int j = 0; j < ntiles; ++j) {
            const uint32_t v = counts[(int64_t)j * 256 + t];      // [tile][digit]: the workgroup reads 1 KiB rows, all loads independent
            tot += v;
            pre += j < me ? v : 0u;
        }
        // exclusive scan of `tot` over the 256 digits: inside a wave by shuffles, across the four waves through LDS
        uint32_t inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        if (lane == 63) wtot[w] = inc;
        __syncthreads();
        uint32_t g = inc - tot + pre;
        for (int i = 0; i < w; ++i) g += wtot[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t c = wcnt[i][t];
            wcnt[i][t] = g;
            g += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int64_t i = wbase + r * 64 + lane;
        if (i < n) {
            const uint32_t d = (uint32_t)(key[r] >> lo) & mask;
            out[(int64_t)wcnt[w][d] + rank[r]] = key[r];
        }
    }
}

void scp_launch_scan_u32(uint32_t *counts, int64_t m, hipStream_t st) {
    if ((uintptr_t)counts & 15) hipLaunchKernelGGL(radix_scan_kernel<false>, dim3(1), dim3(1024), 0, st, counts, m);
    else hipLaunchKernelGGL(radix_scan_kernel<true>, dim3(1), dim3(1024), 0, st, counts, m);
}

// first_hist_done: the producer of keys_a has already written pass 0's [tile][digit] table into ws->counts (scp_radix_counts; the fused
// key kernel of geom.hip does, one workgroup per 4096-key tile like radix_hist_kernel): pass 0 starts at its scan.
uint32_t *scp_radix_counts(RadixWorkspace *ws, int64_t n, int *ntiles_out) {
    const int ntiles = (int)cdiv64(n, TILE);
    if (ws->counts.reserve((size_t)256 * ntiles * sizeof(uint32_t))) return nullptr;
    if (ntiles_out) *ntiles_out = ntiles;
    return ws->counts.as<uint32_t>();
}

int scp_radix_sort_u64(uint64_t *keys_a, uint64_t *keys_b, int64_t n, const int *pass_lo, const int *pass_bits,
                       int npass, RadixWorkspace *ws, hipStream_t st, uint64_t **result, bool first_hist_done) {
    *result = keys_a;
    if (n <= 1 || npass == 0) return SCP_OK;
    const int ntiles = (int)cdiv64(n, TILE);
    int rc = ws->counts.reserve((size_t)256 * ntiles * sizeof(uint32_t));
    if (rc) return rc;
    uint32_t *counts = ws->counts.as<uint32_t>();
    uint64_t *src = keys_a, *dst = keys_b;
    for (int p = 0; p < npass; ++p) {
        const int lo = pass_lo[p], nb = pass_bits[p];
        if (nb < 1 || nb > 8) return SCP_EINVAL;
        if (!(p == 0 && first_hist_done))
            hipLaunchKernelGGL(radix_hist_kernel, dim3(ntiles), dim3(WG), 0, st, src, n, lo, (1u << nb) - 1u, counts, ntiles);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(ntiles), dim3(WG), 0, st, src, dst, n, lo, nb, counts, ntiles);
        LAUNCH_CHECK();
        uint64_t *tmp = src; src = dst; dst = tmp;
    }
    *result = src;
    return SCP_OK;
}
