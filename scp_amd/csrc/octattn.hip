// OctAttention dual-stream causal attention (gfx950).
//
// Replaces models/attention_model.py:58-95 (SelfMultiheadAttention.forward).  Two softmaxes share one score row:
//   known stream   : s_ij = q_u[i].k[j]/sqrt(hd), j <= i                         -> out   = softmax(s) . v
//   unknown stream : same row with the diagonal replaced by q_u[i].k_u[i]/sqrt(hd) -> out_u = sum_{j<i} p'_j v[j] + p'_i v_u[i]
// The reference builds both with (1 - I) masks over [B,H,c,c] tensors; here one wavefront owns one query row,
// keeps the row of scores in LDS, and emits both outputs in a single pass (no c x c tensor, no eye()).
// octattn_kernel: generic VALU version (any head width <= 192).  octattn_mfma_kernel (below): head width 150, the reference
// configuration, on fp32 MFMA.
#include "scp_internal.h"

#define MAXC 1024

__global__ __launch_bounds__(256) void octattn_kernel(const float *__restrict__ q_u, const float *__restrict__ k,
                                                     const float *__restrict__ k_u, const float *__restrict__ v,
                                                     const float *__restrict__ v_u, int c, int H, int hd,
                                                     float *__restrict__ out, float *__restrict__ out_u) {
    __shared__ float sc[4][MAXC];
    __shared__ float qs[4][256];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rows_per_bh = (c + 3) / 4;
    int bid = blockIdx.x;
    const int rblk = bid % rows_per_bh; bid /= rows_per_bh;
    const int head = bid % H, b = bid / H;
    const int i = rblk * 4 + w;
    const int D = H * hd;
    const float scale = 1.0f / sqrtf((float)hd);
    if (i >= c) return;  // whole wave exits together (no block-level barrier below)
    const float *qrow = q_u + ((size_t)b * c + i) * D + head * hd;
    for (int d = lane; d < hd; d += 64) qs[w][d] = qrow[d];
    __builtin_amdgcn_wave_barrier();
    // scores for j <= i
    float mx = -INFINITY;
    for (int j = lane; j <= i; j += 64) {
        const float *kr = k + ((size_t)b * c + j) * D + head * hd;
        float acc = 0.f;
        for (int d = 0; d < hd; ++d) acc = fmaf(qs[w][d], kr[d], acc);
        acc *= scale;
        sc[w][j] = acc;
        mx = fmaxf(mx, acc);
    }
    // diagonal of the unknown stream
    float dz = 0.f;
    {
        const float *kr = k_u + ((size_t)b * c + i) * D + head * hd;
        for (int d = lane; d < hd; d += 64) dz = fmaf(qs[w][d], kr[d], dz);
        for (int off = 32; off > 0; off >>= 1) dz += __shfl_xor(dz, off);
        dz *= scale;
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    __builtin_amdgcn_wave_barrier();
    const float sii = sc[w][i];
    // known stream: max = mx ; unknown stream: max over {s_ij (j<i), dz}
    float mxu = dz;
    for (int j = lane; j < i; j += 64) mxu = fmaxf(mxu, sc[w][j]);
    for (int off = 32; off > 0; off >>= 1) mxu = fmaxf(mxu, __shfl_xor(mxu, off));
    float sum = 0.f, sumu = 0.f;
    for (int j = lane; j < i; j += 64) { const float s = sc[w][j]; sum += __expf(s - mx); sumu += __expf(s - mxu); }
    for (int off = 32; off > 0; off >>= 1) { sum += __shfl_xor(sum, off); sumu += __shfl_xor(sumu, off); }
    const float pii = __expf(sii - mx), piu = __expf(dz - mxu);
    sum += pii; sumu += piu;
    const float inv = 1.f / sum, invu = 1.f / sumu;
    // outputs: lane owns head dims lane, lane+64, lane+128
    float o[3] = {0.f, 0.f, 0.f}, ou[3] = {0.f, 0.f, 0.f};
    for (int j = 0; j < i; ++j) {
        const float s = sc[w][j];
        const float p = __expf(s - mx) * inv, pu = __expf(s - mxu) * invu;
        const float *vr = v + ((size_t)b * c + j) * D + head * hd;
#pragma unroll
        for (int t = 0; t < 3; ++t) { const int d = lane + 64 * t; if (d < hd) { const float vv = vr[d]; o[t] = fmaf(p, vv, o[t]); ou[t] = fmaf(pu, vv, ou[t]); } }
    }
    {
        const float *vr = v + ((size_t)b * c + i) * D + head * hd, *vur = v_u + ((size_t)b * c + i) * D + head * hd;
        float *orow = out + ((size_t)b * c + i) * D + head * hd, *ourow = out_u + ((size_t)b * c + i) * D + head * hd;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int d = lane + 64 * t;
            if (d < hd) { orow[d] = fmaf(pii * inv, vr[d], o[t]); ourow[d] = fmaf(piu * invu, vur[d], ou[t]); }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// MFMA version (head width 150 = the reference configuration, 600 / 4 heads), fp32 on v_mfma_f32_32x32x2_f32.
// Both streams share every off-diagonal score, so ONE flash-attention pass over the strictly lower triangle gives, per query i,
//     M = max_{j<i} s_ij,   L = sum_{j<i} e^{s_ij - M},   A = sum_{j<i} e^{s_ij - M} v_j
// and the two outputs differ only in their diagonal term (s_ii with v_i, or dz_i = q_i.ku_i with vu_i):
//     out_i = (A e^{M-m} + e^{s_ii-m} v_i) / (L e^{M-m} + e^{s_ii-m}),  m = max(M, s_ii)      (out_u likewise with dz_i, vu_i).
// Workgroup = 4 waves = 128 queries of one (batch, head); a wave owns 32 queries (its q fragment lives in registers: lane half h
// holds dims 75 h .. 75 h + 74).  Key tiles of 32 rows go through LDS: K as [key][dims 0..74 | pad | dims 75..149 | pad] (row
// stride 156 floats: conflict-free ds_read_b128), V as [key][160] (zero padded).  S^T = K . Q^T, O^T += V^T . P^T with P kept in
// the accumulator registers (the same register <-> key pairing as the Swin kernel).  Causality skips the tiles above the
// diagonal; the diagonal tile is masked per element.
typedef float f32x4o __attribute__((ext_vector_type(4)));
typedef float f32x16o __attribute__((ext_vector_type(16)));
#define OHD 150
#define OKS 76          // MFMA k-steps per half (75 dims + 1 zero pad)
#define OLDK 156
#define OLDV 160
#define ODT 5           // 32-wide tiles of the head dims (160 padded)

__global__ __launch_bounds__(256, 2) void octattn_mfma_kernel(const float *__restrict__ q_u, const float *__restrict__ k,
                                                             const float *__restrict__ k_u, const float *__restrict__ v,
                                                             const float *__restrict__ v_u, int c, int H, float *__restrict__ out,
                                                             float *__restrict__ out_u) {
    __shared__ __attribute__((aligned(16))) float Ks[32 * OLDK];
    __shared__ __attribute__((aligned(16))) float Vs[32 * OLDV];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, col = lane & 31, h = lane >> 5;
    const int qblocks = (c + 127) / 128;
    int bid = blockIdx.x;
    // heavy query blocks (late rows attend to more keys) first
    const int qb = qblocks - 1 - bid % qblocks; bid /= qblocks;
    const int head = bid % H, b = bid / H;
    const int D = H * OHD;
    const size_t base = (size_t)b * c * D + (size_t)head * OHD;
    const float scale = 1.0f / sqrtf((float)OHD);
    const int q0 = qb * 128, qi = q0 + w * 32 + col;
    const int qc = qi < c ? qi : c - 1;

    float qf[OKS];
    {
        const float *src = q_u + base + (size_t)qc * D + 75 * h;
#pragma unroll
        for (int s = 0; s < 75; ++s) qf[s] = src[s];
        qf[75] = 0.f;
    }
    f32x16o o[ODT];
#pragma unroll
    for (int t = 0; t < ODT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // zero the padding once: K columns 75, 151..155 and V columns 150..159 are never written by the staging loop
    for (int e = tid; e < 32 * OLDK; e += 256) Ks[e] = 0.f;
    for (int e = tid; e < 32 * OLDV; e += 256) Vs[e] = 0.f;

    const int last_q = (q0 + 127 < c ? q0 + 127 : c - 1);
    const int nkt = (last_q + 31) / 32;                 // tiles holding keys j < last_q
    const int wave_last = q0 + w * 32 + 31;             // this wave needs keys j < wave_last
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        // stage keys kt*32 .. +31: 75 float2 per row and operand (rows are 8-byte aligned: head offset 600 B)
        for (int e = tid; e < 32 * 75; e += 256) {
            const int r = e / 75, p = e - r * 75;
            int j = kt * 32 + r;
            j = j < c ? j : c - 1;
            const size_t g = base + (size_t)j * D + 2 * p;
            const float2 kv = *(const float2 *)(k + g);
            const float2 vv = *(const float2 *)(v + g);
            const int d = 2 * p;                         // dims d, d + 1 -> column d (< 75) or 76 + (d - 75)
            Ks[r * OLDK + (d < 75 ? d : d + 1)] = kv.x;
            Ks[r * OLDK + (d + 1 < 75 ? d + 1 : d + 2)] = kv.y;
            *(float2 *)(Vs + r * OLDV + d) = vv;
        }
        __syncthreads();
        if (kt * 32 >= wave_last) continue;              // wave-uniform: the whole tile is above this wave's diagonal
        // ---- S^T = K . Q^T --------------------------------------------------------------------------------------
        f32x16o sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
        const float *krow = Ks + col * OLDK + 76 * h;
#pragma unroll
        for (int g = 0; g < OKS / 4; ++g) {
            const f32x4o kk = *(const f32x4o *)(krow + 4 * g);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kk[0], qf[4 * g], sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kk[1], qf[4 * g + 1], sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kk[2], qf[4 * g + 2], sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kk[3], qf[4 * g + 3], sc, 0, 0, 0);
        }
        // ---- strict causal mask (j < i), online softmax ------------------------------------------------------------
        const int j0 = kt * 32 + 4 * h;
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + (r & 3) + 8 * (r >> 2);
            sc[r] = (j < qi) ? sc[r] * scale : -INFINITY;
            mx = fmaxf(mx, sc[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        // a query with no admissible key so far keeps (m, l, o) = (-inf, 0, 0): exp(-inf - (-inf)) must not produce NaN
        const float alpha = (m_new == -INFINITY) ? 1.f : __expf(m_run - m_new);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sc[r] = (sc[r] == -INFINITY) ? 0.f : __expf(sc[r] - m_new); ps += sc[r]; }
        ps += __shfl_xor(ps, 32);
        l_run = l_run * alpha + ps;
        m_run = m_new;
        if (__any(alpha != 1.f)) {
#pragma unroll
            for (int t = 0; t < ODT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        }
        // ---- O^T += V^T . P^T : step r pairs key (r&3)+8(r>>2)+4h of both operands -----------------------------------
        const float *vbase = Vs + (4 * h) * OLDV + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float *vr = vbase + ((r & 3) + 8 * (r >> 2)) * OLDV;
#pragma unroll
            for (int t = 0; t < ODT; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[32 * t], sc[r], o[t], 0, 0, 0);
        }
    }
    // ---- the two diagonal terms -------------------------------------------------------------------------------------
    float sii = 0.f, dz = 0.f;
    {
        const float *kr = k + base + (size_t)qc * D + 75 * h, *kur = k_u + base + (size_t)qc * D + 75 * h;
#pragma unroll
        for (int s = 0; s < 75; ++s) { sii = fmaf(qf[s], kr[s], sii); dz = fmaf(qf[s], kur[s], dz); }
        sii += __shfl_xor(sii, 32);
        dz += __shfl_xor(dz, 32);
        sii *= scale; dz *= scale;
    }
    if (qi >= c) return;
    const float m1 = fmaxf(m_run, sii), m2 = fmaxf(m_run, dz);
    const float eo1 = (m_run == -INFINITY) ? 0.f : __expf(m_run - m1), ed1 = __expf(sii - m1);
    const float eo2 = (m_run == -INFINITY) ? 0.f : __expf(m_run - m2), ed2 = __expf(dz - m2);
    const float inv1 = 1.f / (l_run * eo1 + ed1), inv2 = 1.f / (l_run * eo2 + ed2);
    const float a1 = eo1 * inv1, b1 = ed1 * inv1, a2 = eo2 * inv2, b2 = ed2 * inv2;
    const float *vr = v + base + (size_t)qi * D, *vur = v_u + base + (size_t)qi * D;
    float *orow = out + base + (size_t)qi * D, *ourow = out_u + base + (size_t)qi * D;
#pragma unroll
    for (int t = 0; t < ODT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (d < OHD) {
                orow[d] = fmaf(b1, vr[d], a1 * o[t][r]);
                ourow[d] = fmaf(b2, vur[d], a2 * o[t][r]);
            }
        }
}

extern "C" int scp_octattn_attention(const float *q_u, const float *k, const float *k_u, const float *v, const float *v_u, int32_t B,
                                     int32_t c, int32_t H, int32_t hd, float *out, float *out_u, void *stream) {
    if (!q_u || !k || !k_u || !v || !v_u || !out || !out_u || B <= 0 || c <= 0 || c > MAXC || H <= 0 || hd <= 0 || hd > 192)
        return SCP_EINVAL;
    if (hd == OHD && H * hd % 2 == 0 && ((((uintptr_t)k | (uintptr_t)v) & 7) == 0)) {   // the reference configuration: MFMA kernel
        hipLaunchKernelGGL(octattn_mfma_kernel, dim3(B * H * ((c + 127) / 128)), dim3(256), 0, (hipStream_t)stream, q_u, k, k_u, v, v_u, c, H, out,
                           out_u);
        LAUNCH_CHECK();
        return SCP_OK;
    }
    const int nblk = B * H * ((c + 3) / 4);
    hipLaunchKernelGGL(octattn_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, q_u, k, k_u, v, v_u, c, H, hd, out, out_u);
    LAUNCH_CHECK();
    return SCP_OK;
}
