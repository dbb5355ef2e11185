// OctAttention dual-stream causal attention (gfx950).
//
// Replaces models/attention_model.py:58-95 (SelfMultiheadAttention.forward).  Two softmaxes share one score row:
//   known stream   : s_ij = q_u[i].k[j]/sqrt(hd), j <= i                         -> out   = softmax(s) . v
//   unknown stream : same row with the diagonal replaced by q_u[i].k_u[i]/sqrt(hd) -> out_u = sum_{j<i} p'_j v[j] + p'_i v_u[i]
// The reference builds both with (1 - I) masks over [B,H,c,c] tensors; here one wavefront owns one query row,
// keeps the row of scores in LDS, and emits both outputs in a single pass (no c x c tensor, no eye()).
// Round-1 version: VALU dot products (head width 150 is not an MFMA-friendly K); the MFMA variant is future work
//  - this path only serves the OctAttention configs, not the headline metric.
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

extern "C" int scp_octattn_attention(const float *q_u, const float *k, const float *k_u, const float *v, const float *v_u, int32_t B,
                                     int32_t c, int32_t H, int32_t hd, float *out, float *out_u, void *stream) {
    if (!q_u || !k || !k_u || !v || !v_u || !out || !out_u || B <= 0 || c <= 0 || c > MAXC || H <= 0 || hd <= 0 || hd > 192)
        return SCP_EINVAL;
    const int nblk = B * H * ((c + 3) / 4);
    hipLaunchKernelGGL(octattn_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, q_u, k, k_u, v, v_u, c, H, hd, out, out_u);
    LAUNCH_CHECK();
    return SCP_OK;
}
