// OctAttention dual-stream causal attention on f16 MFMA with fp32-class accuracy ("f16x3", gfx950 / CDNA4), head width 150.
//
// Same mathematics as octattn_mfma_kernel (octattn.hip; replaces models/attention_model.py:58-95): one flash pass over the strictly
// lower triangle gives (M, L, A) per query, the two streams differ in their diagonal term only.  The difference is the arithmetic
// of the two big products: every operand is a pair of IEEE-half planes hi = f16(x s), lo = f16(x s - hi) (22 significant bits)
// and a product is three v_mfma_f32_32x32x16_f16 (lo.hi + hi.lo + hi.hi, fp32 accumulate) - 60 MFMAs of 32 cycles per 32-key
// tile instead of 156 fp32 MFMAs of 64 cycles.  Scales s are powers of two, undone exactly:
//   q, k : one per (token, head), largest magnitude -> [2^13, 2^14)   (score = acc / (s_q s_k))
//   v    : one for the whole launch (scp_octattn_attention_f16x3 reduces max |v| first): o = acc / s_v
//   p    : none (p <= 1)
// A preparation kernel writes the planes in the layouts the attention kernel consumes, so that a key tile goes global -> LDS by
// LDS-DMA as a linear copy:
//   Q planes  [b][token][head][2][160]                                  (read once per query block, straight into registers)
//   K image   [b][tile][head] : planes [2][32 keys][168] (336-byte rows: conflict-free ds_read_b128) + 32 inverse scales
//   V image   [b][tile][head] : planes [2][160 dims][40] = V^T, key order permuted (bits 2 and 3 of the key index swapped: the
//             order in which the S^T accumulator registers hold the keys), 80-byte rows
// The preparation kernel also evaluates the two diagonal terms q_i.k_i and q_i.ku_i (plain fp32) - the attention kernel reads
// neither k nor k_u.
// Workgroup = 4 waves = 128 queries of one (batch, head); a wave owns 32 queries.  Per key tile: wait K -> barrier -> start the
// V DMA -> S^T = K . Q^T, softmax -> wait V -> barrier -> start the next K DMA -> O^T += V^T . P^T.  The epilogue goes through a
// wave-private LDS tile so that v / v_u loads and the stores of out / out_u move 128 contiguous bytes per row.
#include "scp_internal.h"

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16h __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void *oa_lds_ptr_t;
typedef const __attribute__((address_space(1))) void *oa_glb_ptr_t;

#define FHD 150
#define FHP 160                     // head dims padded to ten k-chunks of 16
#define FKLD 168                    // f16 per K row in the tile image
#define FVLD 40                     // f16 per V^T row in the tile image
#define FK_PLANE (32 * FKLD)        // f16 per K plane
#define FV_PLANE (FHP * FVLD)       // f16 per V plane
#define FK_ISK (2 * FK_PLANE * 2)   // byte offset of the 32 inverse key scales inside the K image
#define FK_IMG 22528                // bytes: 2 planes + 128 B of scales, rounded up to 1 KiB
#define FV_IMG 25600                // bytes: 2 planes (25 KiB)
static_assert(FK_ISK + 128 <= FK_IMG && 2 * FV_PLANE * 2 == FV_IMG, "image sizes");

__device__ __forceinline__ void oa_pow2_scale(float mx, float &sc, float &isc) {
    int e = 0;
    if (mx > 0.f && mx < INFINITY) {
        e = 140 - (int)((__float_as_uint(mx) >> 23) & 0xffu);   // mx 2^e in [2^13, 2^14)
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
    }
    sc = __uint_as_float((unsigned)(127 + e) << 23);
    isc = __uint_as_float((unsigned)(127 - e) << 23);
}

// max |v| over the launch: one partial maximum per wave (finite, non-negative), reduced by every workgroup of oa_prep_kernel.  (Until round 5
// the waves met in an atomicMax on a word that a 4-byte hipMemsetAsync had cleared: that memset is a runtime blit kernel whose system-scope
// release writes the whole L2 back - 100 us on average and up to 1.25 ms behind the key | value GEMM, 1.2 ms per L14 frame,
// profiles/r5_octattn_L14_frame_kernel_stats.csv: __amd_rocclr_fillBufferAligned.)
#define OA_PART_MAX 4096            // 1024 workgroups x 4 waves
#define OA_HDR (1024 + 4 * OA_PART_MAX)
__global__ __launch_bounds__(256) void oa_absmax_kernel(const float *__restrict__ v, int64_t n4, int d4 /* float4 per row */, int64_t ld,
                                                       float *__restrict__ part) {
    float mx = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / d4;
        const float4 x = *(const float4 *)(v + r * ld + 4 * (i - r * d4));
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(x.x), fabsf(x.y))), fmaxf(fabsf(x.z), fabsf(x.w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) part[blockIdx.x * 4 + (threadIdx.x >> 6)] = mx == mx ? fminf(mx, 3.0e38f) : 0.f;
}

// workgroup = one 32-token tile of one (batch, head): thread (t = tid >> 3, s = tid & 7) walks float2 pieces s, s + 8, ... of row t.
// Round 5: the Q rows and the K image are assembled in LDS and leave as 16-byte stores of whole rows / one linear 22 KiB block.  (Rounds 1 - 4
// stored every converted pair straight from the registers: 40 four-byte stores per thread, 32-byte segments of eight different rows per
// wave instruction - the kernel moved its 5.3 GB per L14 layer at 0.85 TB/s, 81 % of its wave cycles issue-stalled, profiles/r4i_octattn_L14_*.)
// Same arithmetic, same bits.
__global__ __launch_bounds__(256) void oa_prep_kernel(const float *__restrict__ q_u, const float *__restrict__ k, const float *__restrict__ k_u,
                                                     const float *__restrict__ v, int64_t ldkv, int c, int H, int nt, const float *__restrict__ vpart, int npart,
                                                     unsigned *__restrict__ vmax_out, _Float16 *__restrict__ qp, float *__restrict__ isq, float2 *__restrict__ diag,
                                                     char *__restrict__ kimg, char *__restrict__ vimg) {
    // Round 6: ONE 22 KiB LDS buffer, used three times in turn - the scaled V rows (19 KiB), the 32 Q rows (20 KiB), the K image (22 KiB) -
    // instead of three buffers side by side (61 KiB: two workgroups per CU, two waves per SIMD, and the kernel is latency-bound: 61 % of its
    // wave cycles instruction-stalled, profiles/r5_octattn_L14_frame_sq_wave_states.txt).  The converted q / k values wait in registers (they did
    // before, too); three more barriers per workgroup buy twice the resident waves.  Same arithmetic, same bits.
    __shared__ __attribute__((aligned(16))) char stage[FK_IMG];
    __shared__ float vred[4];
    float (*vt)[FHD + 1] = (float (*)[FHD + 1])stage;                      // phase V: the 32 scaled V rows
    _Float16 (*qst)[2 * FHP] = (_Float16 (*)[2 * FHP])stage;               // phase Q: the 32 Q rows (hi plane | lo plane)
    char *kst = stage;                                                      // phase K: the K image of this (tile, head), as it will lie in memory
    static_assert(32 * (FHD + 1) * 4 <= FK_IMG && 32 * 2 * FHP * 2 <= FK_IMG, "one buffer holds every phase");
    const int tid = threadIdx.x, t = tid >> 3, s = tid & 7;
    const int tile = blockIdx.x % nt, b = blockIdx.x / nt, head = blockIdx.y;
    const int D = H * FHD, cpad = nt * 32;
    const int tok = tile * 32 + t;
    const bool real = tok < c;
    const size_t row = ((size_t)b * c + (real ? tok : c - 1)) * D + (size_t)head * FHD;            // q_u: dense rows
    const size_t rowk = ((size_t)b * c + (real ? tok : c - 1)) * (size_t)ldkv + (size_t)head * FHD;   // k, k_u, v: rows ldkv floats apart
    float pm = 0.f;
    for (int i = tid; i < npart; i += 256) pm = fmaxf(pm, vpart[i]);

    float2 qv[10], kv[10];
    float qm = 0.f, km = 0.f, sii = 0.f, dz = 0.f;                   // the two diagonal terms q_i.k_i, q_i.ku_i: plain fp32
    // all 40 loads of the thread first (80 registers): written as one loop with their use, the compiler keeps only one iteration's four
    // loads in flight (it waits for an iteration's data before it issues the next-but-one's) and the kernel is latency-bound
    float2 ld_q[10], ld_k[10], ld_v[10], ld_u[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const int p = s + 8 * i;
        const int pc = p < 75 ? p : 74;
        ld_q[i] = *(const float2 *)(q_u + row + 2 * pc); ld_k[i] = *(const float2 *)(k + rowk + 2 * pc);
        ld_v[i] = *(const float2 *)(v + rowk + 2 * pc); ld_u[i] = *(const float2 *)(k_u + rowk + 2 * pc);
    }
    __builtin_amdgcn_sched_barrier(0);
    // max |v| of the launch (every workgroup reduces the partial maxima itself; workgroup (0, 0) leaves the value for the attention kernel)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pm = fmaxf(pm, __shfl_xor(pm, o));
    if ((tid & 63) == 0) vred[tid >> 6] = pm;
    __syncthreads();
    const float vmaxv = fmaxf(fmaxf(vred[0], vred[1]), fmaxf(vred[2], vred[3]));
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *vmax_out = __float_as_uint(vmaxv);
    float vs, ivs;
    oa_pow2_scale(vmaxv, vs, ivs);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const int p = s + 8 * i;
        const bool ok = real && p < 75;
        const float2 a = ld_q[i], bb = ld_k[i], cc = ld_v[i], uu = ld_u[i];
        qv[i] = ok ? a : make_float2(0.f, 0.f);
        kv[i] = ok ? bb : make_float2(0.f, 0.f);
        qm = fmaxf(qm, fmaxf(fabsf(qv[i].x), fabsf(qv[i].y)));
        km = fmaxf(km, fmaxf(fabsf(kv[i].x), fabsf(kv[i].y)));
        sii = fmaf(qv[i].x, kv[i].x, fmaf(qv[i].y, kv[i].y, sii));
        dz = fmaf(qv[i].x, uu.x, fmaf(qv[i].y, uu.y, dz));
        if (p < 75) { vt[t][2 * p] = ok ? cc.x * vs : 0.f; vt[t][2 * p + 1] = ok ? cc.y * vs : 0.f; }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        qm = fmaxf(qm, __shfl_xor(qm, o)); km = fmaxf(km, __shfl_xor(km, o));
        sii += __shfl_xor(sii, o); dz += __shfl_xor(dz, o);
    }
    float qs, iqs, ks, iks;
    oa_pow2_scale(qm, qs, iqs);
    oa_pow2_scale(km, ks, iks);
    const size_t th = ((size_t)b * cpad + tok) * H + head;
    if (s == 0) { isq[th] = iqs; diag[th] = make_float2(sii, dz); }
    __syncthreads();
    // ---- phase V.  V^T: thread = (dim d, group g of 8 permuted key positions): position 8 g + j holds key 16 (g >> 1) + 4 (g & 1) + (j & 3) + 8 (j >> 2)
    {
        _Float16 *vbase = (_Float16 *)(vimg + (((size_t)b * nt + tile) * H + head) * FV_IMG);
        for (int e = tid; e < FHP * 4; e += 256) {
            const int d = e >> 2, g = e & 3;
            h16x8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int key = 16 * (g >> 1) + 4 * (g & 1) + (j & 3) + 8 * (j >> 2);
                const float x = d < FHD ? vt[key][d] : 0.f;
                hi[j] = (_Float16)x;
                lo[j] = (_Float16)(x - (float)hi[j]);
            }
            *(h16x8 *)(vbase + d * FVLD + 8 * g) = hi;
            *(h16x8 *)(vbase + FV_PLANE + d * FVLD + 8 * g) = lo;
        }
    }
    __syncthreads();
    // ---- phase Q: the 32 Q rows assembled in LDS, out as 16-byte stores of whole rows (640 contiguous bytes per token, H x 640 bytes apart)
    {
        _Float16 *qrow = qst[t];
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int p = s + 8 * i;                  // float2 piece; pieces 75..79 are the zero padding (dims 150..159)
            h16x2 qh, ql;
            const float x[2] = {qv[i].x * qs, qv[i].y * qs};
            qh[0] = (_Float16)x[0]; ql[0] = (_Float16)(x[0] - (float)qh[0]);
            qh[1] = (_Float16)x[1]; ql[1] = (_Float16)(x[1] - (float)qh[1]);
            *(h16x2 *)(qrow + 2 * p) = qh;
            *(h16x2 *)(qrow + FHP + 2 * p) = ql;
        }
    }
    __syncthreads();
    {
        constexpr int QV = 2 * FHP * 2 / 16;          // 16-byte pieces per Q row (40)
        for (int e = tid; e < 32 * QV; e += 256) {
            const int r = e / QV, pce = e - r * QV;
            const size_t thr = ((size_t)b * cpad + tile * 32 + r) * H + head;
            *(uint4 *)((char *)(qp + thr * (2 * FHP)) + 16 * pce) = *(const uint4 *)((const char *)qst[r] + 16 * pce);
        }
    }
    __syncthreads();
    // ---- phase K: the K image (planes [2][32 keys][168] + 32 inverse scales), one linear 22 KiB block
    {
        _Float16 *krow = (_Float16 *)kst + t * FKLD;
        if (s == 0) ((float *)(kst + FK_ISK))[t] = iks;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int p = s + 8 * i;
            h16x2 kh, kl;
            const float x[2] = {kv[i].x * ks, kv[i].y * ks};
            kh[0] = (_Float16)x[0]; kl[0] = (_Float16)(x[0] - (float)kh[0]);
            kh[1] = (_Float16)x[1]; kl[1] = (_Float16)(x[1] - (float)kh[1]);
            *(h16x2 *)(krow + 2 * p) = kh;
            *(h16x2 *)(krow + FK_PLANE + 2 * p) = kl;
        }
        // the bytes of the K image no row writes (dims 160 .. 167 of every row, the tail behind the scales) go out as zeros: the attention kernel
        // copies the image as a whole and never reads them, but the workspace stays deterministic
        if (s == 0) {
            *(uint4 *)(krow + FHP) = make_uint4(0, 0, 0, 0);
            *(uint4 *)(krow + FK_PLANE + FHP) = make_uint4(0, 0, 0, 0);
        }
        for (int e = FK_ISK + 128 + 16 * tid; e < FK_IMG; e += 16 * 256) *(uint4 *)(kst + e) = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    {
        char *kbase = kimg + (((size_t)b * nt + tile) * H + head) * FK_IMG;
        for (int e = tid; e < FK_IMG / 16; e += 256) *(uint4 *)(kbase + 16 * e) = *(const uint4 *)(kst + 16 * e);
    }
}

__global__ __launch_bounds__(256, 2) void oa_attn_f16x3_kernel(const float *__restrict__ v,
                                                              const float *__restrict__ v_u, int64_t ldkv, int c, int H, int nt,
                                                              const unsigned *__restrict__ vmax_bits, const _Float16 *__restrict__ qp,
                                                              const float *__restrict__ isq, const float2 *__restrict__ diag,
                                                              const char *__restrict__ kimg,
                                                              const char *__restrict__ vimg, float *__restrict__ out,
                                                              float *__restrict__ out_u) {
    __shared__ __attribute__((aligned(1024))) char Ks[FK_IMG];
    __shared__ __attribute__((aligned(1024))) char Vs[FV_IMG];
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qblocks = (c + 127) / 128;
    int bid = blockIdx.x;
    // workgroups are dealt round-robin over the 8 XCDs: give every XCD a contiguous run, so that the query blocks of one
    // (batch, head) - which stream the same key tiles - share one L2
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);
    const int qb = qblocks - 1 - bid % qblocks; bid /= qblocks;      // heavy query blocks (late rows see more keys) first
    const int head = bid % H, b = bid / H;
    const int D = H * FHD, cpad = nt * 32;
    const size_t base = (size_t)b * c * D + (size_t)head * FHD;
    const size_t basev = (size_t)b * c * (size_t)ldkv + (size_t)head * FHD;
    const int q0 = qb * 128, qi = q0 + w * 32 + col;
    const int qc = qi < c ? qi : c - 1;
    constexpr float LOG2E = 1.4426950408889634f;

    // Q fragments (B operand): query qc, dims 16 ch + 8 h .. + 7, both planes
    h16x8 qh[10], ql[10];
    float qmul;                                                       // 1 / s_q, 1 / sqrt(150) and log2(e) in one factor
    {
        const size_t th = ((size_t)b * cpad + qc) * H + head;
        const _Float16 *src = qp + th * (2 * FHP) + 8 * h;
#pragma unroll
        for (int ch = 0; ch < 10; ++ch) { qh[ch] = *(const h16x8 *)(src + 16 * ch); ql[ch] = *(const h16x8 *)(src + FHP + 16 * ch); }
        qmul = isq[th] * (LOG2E / sqrtf((float)FHD));
    }
    float sii, dz;                                                    // diagonal terms (oa_prep_kernel), log2 domain
    {
        const float2 dd = diag[((size_t)b * cpad + qc) * H + head];
        const float sc2 = LOG2E / sqrtf((float)FHD);
        sii = dd.x * sc2;
        dz = dd.y * sc2;
    }
    f32x16h o[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;                             // log2 domain

    const int last_q = (q0 + 127 < c ? q0 + 127 : c - 1);
    const int nkt = (last_q + 31) / 32;                               // tiles holding keys j < last_q
    const int wave_last = q0 + w * 32 + 31;                           // this wave needs keys j < wave_last
    const char *kt_base = kimg + ((size_t)b * nt * H + head) * FK_IMG, *vt_base = vimg + ((size_t)b * nt * H + head) * FV_IMG;
    auto issue_k = [&](int kt) {
        int ln;
        asm volatile("v_mov_b32 %0, %1" : "=v"(ln) : "v"(lane));     // keep the address arithmetic next to the DMA (no hoisting, no spills)
        const char *src = kt_base + (size_t)kt * H * FK_IMG + 16 * ln;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int i = w + 4 * j;
            if (i < FK_IMG / 1024) __builtin_amdgcn_global_load_lds((oa_glb_ptr_t)(src + i * 1024), (oa_lds_ptr_t)(Ks + i * 1024), 16, 0, 0);
        }
    };
    auto issue_v = [&](int kt) {
        int ln;
        asm volatile("v_mov_b32 %0, %1" : "=v"(ln) : "v"(lane));
        const char *src = vt_base + (size_t)kt * H * FV_IMG + 16 * ln;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int i = w + 4 * j;
            if (i < FV_IMG / 1024) __builtin_amdgcn_global_load_lds((oa_glb_ptr_t)(src + i * 1024), (oa_lds_ptr_t)(Vs + i * 1024), 16, 0, 0);
        }
    };

    if (nkt > 0) issue_k(0);
    for (int kt = 0; kt < nkt; ++kt) {
        SCP_WAIT_DMA(0);
        __syncthreads();                                              // K tile kt has landed; every wave is done with V tile kt - 1
        issue_v(kt);
        const bool active = kt * 32 < wave_last;                      // wave-uniform: otherwise the tile is above this wave's diagonal
        f32x16h sc;
        if (active) {
            // ---- S^T = K . Q^T ------------------------------------------------------------------------------------------
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = 0.f;
            const _Float16 *krow = (const _Float16 *)Ks + col * FKLD + 8 * h;
#pragma unroll
            for (int ch = 0; ch < 10; ++ch) {
                const h16x8 kh = *(const h16x8 *)(krow + 16 * ch), kl = *(const h16x8 *)(krow + FK_PLANE + 16 * ch);
                sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ch], sc, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ch], sc, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ch], sc, 0, 0, 0);
            }
            // ---- un-scale, strict causal mask (j < i), online softmax (log2 domain) ---------------------------------------
            const float *isk = (const float *)(Ks + FK_ISK);
            const int j0 = kt * 32 + 4 * h;
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jl = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float sv = (sc[r] * isk[jl]) * qmul;
                sc[r] = (j0 - 4 * h + jl < qi) ? sv : -INFINITY;
                mx = fmaxf(mx, sc[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            // a query with no admissible key so far keeps (m, l, o) = (-inf, 0, 0): exp(-inf - (-inf)) must not produce NaN
            const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new);
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sc[r] = (sc[r] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(sc[r] - m_new); ps += sc[r]; }
            ps += __shfl_xor(ps, 32);
            l_run = l_run * alpha + ps;
            m_run = m_new;
            if (__any(alpha != 1.f)) {
#pragma unroll
                for (int t = 0; t < 5; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
            }
        }
        SCP_WAIT_DMA(0);
        __syncthreads();                                              // V tile kt has landed; every wave is done with K tile kt
        if (kt + 1 < nkt) issue_k(kt + 1);
        if (active) {
            // ---- O^T += V^T . P^T: chunk cc pairs accumulator registers 8 cc .. 8 cc + 7 with V^T columns 16 cc + 8 h + j ----
            const _Float16 *vrow = (const _Float16 *)Vs + col * FVLD + 8 * h;
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                h16x8 ph, pl;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x = sc[8 * cc + j];
                    const _Float16 hh = (_Float16)x;
                    ph[j] = hh;
                    pl[j] = (_Float16)(x - (float)hh);
                }
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    const h16x8 vh = *(const h16x8 *)(vrow + 32 * t * FVLD + 16 * cc), vl = *(const h16x8 *)(vrow + FV_PLANE + 32 * t * FVLD + 16 * cc);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o[t], 0, 0, 0);
                }
            }
        }
    }
    // ---- epilogue: out_i = (A e^{M-m} + e^{s_ii-m} v_i) / (L e^{M-m} + e^{s_ii-m}), out_u likewise with dz_i, vu_i.  The accumulators
    //      hold one query per lane (rows 2400 B apart): each 32-dim slab goes through a wave-private LDS tile so that loads of
    //      v / v_u and the stores move 128 contiguous bytes per row (16 lanes x float2, four rows per instruction)
    __syncthreads();                                                  // every wave is done with the operand tiles
    float vs, ivs;
    oa_pow2_scale(__uint_as_float(*vmax_bits), vs, ivs);
    const float m1 = fmaxf(m_run, sii), m2 = fmaxf(m_run, dz);
    const float eo1 = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m1), ed1 = __builtin_amdgcn_exp2f(sii - m1);
    const float eo2 = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m2), ed2 = __builtin_amdgcn_exp2f(dz - m2);
    const float inv1 = 1.f / (l_run * eo1 + ed1), inv2 = 1.f / (l_run * eo2 + ed2);
    constexpr int SLD = 34;                                           // floats per staged row (even: float2 reads stay aligned)
    float *stg = (float *)Ks + w * (32 * SLD);
    float *coef = (float *)Vs + w * 128;
    if (h == 0) *(float4 *)(coef + 4 * col) = make_float4(eo1 * inv1 * ivs, ed1 * inv1, eo2 * inv2 * ivs, ed2 * inv2);
    const int rsub = lane >> 4, pp = lane & 15;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) stg[col * SLD + (r & 3) + 8 * (r >> 2) + 4 * h] = o[t][r];
        __builtin_amdgcn_wave_barrier();
        const int d = 32 * t + 2 * pp;
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int ql_ = pass * 4 + rsub, qr_ = q0 + w * 32 + ql_;
            if (qr_ < c && d < FHD) {
                const float2 ov = *(const float2 *)(stg + ql_ * SLD + 2 * pp);
                const float4 cf = *(const float4 *)(coef + 4 * ql_);
                const size_t ro = base + (size_t)qr_ * D + d, rv = basev + (size_t)qr_ * (size_t)ldkv + d;
                const float2 vv = *(const float2 *)(v + rv), vu = *(const float2 *)(v_u + rv);
                *(float2 *)(out + ro) = make_float2(fmaf(cf.y, vv.x, cf.x * ov.x), fmaf(cf.y, vv.y, cf.x * ov.y));
                *(float2 *)(out_u + ro) = make_float2(fmaf(cf.w, vu.x, cf.z * ov.x), fmaf(cf.w, vu.y, cf.z * ov.y));
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

static inline size_t oa_align(size_t x) { return (x + 1023) & ~(size_t)1023; }

extern "C" SCP_API int64_t scp_octattn_f16x3_ws_bytes(int32_t B, int32_t c, int32_t H) {
    if (B <= 0 || c <= 0 || H <= 0) return SCP_EINVAL;
    const size_t nt = (size_t)(c + 31) / 32, rows = (size_t)B * nt * 32 * H;
    return (int64_t)(OA_HDR + oa_align(rows * 4) + oa_align(rows * 8) + oa_align(rows * 2 * FHP * 2) + (size_t)B * nt * H * (FK_IMG + FV_IMG));
}

static int oa_attention_impl(const float *q_u, const float *k, const float *k_u, const float *v, const float *v_u, int64_t ldkv, int32_t B, int32_t c, int32_t H,
                             int32_t hd, float *out, float *out_u, void *workspace, int64_t ws_bytes, const uint32_t *vmax_given, void *stream) {
    if (!q_u || !k || !k_u || !v || !v_u || !out || !out_u || !workspace || B <= 0 || c <= 0 || c > 1024 || H <= 0 || hd != FHD || ((H * FHD) & 3) ||
        ldkv < H * FHD || (ldkv & 3) || ((((uintptr_t)k_u | (uintptr_t)v_u) & 15) != 0) ||
        ((((uintptr_t)q_u | (uintptr_t)k | (uintptr_t)v) & 15) != 0) || ((uintptr_t)workspace & 1023) ||
        ws_bytes < scp_octattn_f16x3_ws_bytes(B, c, H))
        return SCP_EINVAL;
    const int nt = (c + 31) / 32;
    const size_t rows = (size_t)B * nt * 32 * H;
    char *ws = (char *)workspace;
    unsigned *vmax = (unsigned *)ws;
    float *vpart = (float *)(ws + 1024);
    float *isq = (float *)(ws + OA_HDR);
    float2 *diag = (float2 *)(ws + OA_HDR + oa_align(rows * 4));
    _Float16 *qp = (_Float16 *)((char *)diag + oa_align(rows * 8));
    char *kimg = (char *)qp + oa_align(rows * 2 * FHP * 2);
    char *vimg = kimg + (size_t)B * nt * H * FK_IMG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n4 = (int64_t)B * c * H * FHD / 4;                 // 600 floats per token: a multiple of 4
    {
        SCP_PROF(SCP_PROF_OTHER, st, 0.0);             // operand preparation (planes, scales, diagonal terms)
        const unsigned nblk = (unsigned)(n4 < 256 * 1024 ? (n4 + 255) / 256 : 1024);
        // max |v|: given by the caller (the epilogue of the projection that wrote v took it: one word, the bit pattern of a finite non-negative float),
        // or one pass over v here
        if (!vmax_given) hipLaunchKernelGGL(oa_absmax_kernel, dim3(nblk), dim3(256), 0, st, v, n4, H * FHD / 4, ldkv, vpart);
        hipLaunchKernelGGL(oa_prep_kernel, dim3(B * nt, H), dim3(256), 0, st, q_u, k, k_u, v, ldkv, c, H, nt, vmax_given ? (const float *)vmax_given : (const float *)vpart,
                           vmax_given ? 1 : (int)(4 * nblk), vmax, qp, isq, diag, kimg, vimg);
    }
    SCP_PROF(SCP_PROF_OA_ATTENTION, st, (double)B * 3.0 * 2.0 * c * (double)c * H * FHD);
    hipLaunchKernelGGL(oa_attn_f16x3_kernel, dim3(B * H * ((c + 127) / 128)), dim3(256), 0, st, v, v_u, ldkv, c, H, nt, vmax, qp, isq, diag,
                       kimg, vimg, out, out_u);
    LAUNCH_CHECK();
    return SCP_OK;
}

extern "C" SCP_API int scp_octattn_attention_f16x3(const float *q_u, const float *k, const float *k_u, const float *v, const float *v_u,
                                                   int64_t ldkv, int32_t B, int32_t c, int32_t H, int32_t hd, float *out, float *out_u, void *workspace,
                                                   int64_t ws_bytes, void *stream) {
    return oa_attention_impl(q_u, k, k_u, v, v_u, ldkv, B, c, H, hd, out, out_u, workspace, ws_bytes, nullptr, stream);
}

// the same with max |v| (over the B * c rows of v) given as the bit pattern of a float in device memory - e.g. the col_max word of the scp_linear_split_f16_max
// call that wrote v: no pass over v
extern "C" SCP_API int scp_octattn_attention_f16x3_vmax(const float *q_u, const float *k, const float *k_u, const float *v, const float *v_u,
                                                        int64_t ldkv, int32_t B, int32_t c, int32_t H, int32_t hd, float *out, float *out_u, void *workspace,
                                                        int64_t ws_bytes, const uint32_t *vmax_bits, void *stream) {
    if (!vmax_bits || ((uintptr_t)vmax_bits & 3)) return SCP_EINVAL;
    return oa_attention_impl(q_u, k, k_u, v, v_u, ldkv, B, c, H, hd, out, out_u, workspace, ws_bytes, vmax_bits, stream);
}
