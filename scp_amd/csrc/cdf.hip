// Stage C: softmax -> numpyAc integer CDF on device (gfx950).
//
// numpyAc defines the integer CDF through a SERIAL float32 cumsum (numpyAc/numpyAc.py:111), so the prefix
// sums cannot be re-associated: one lane walks one row.  A workgroup stages 64 rows in LDS with coalesced
// loads (row stride 255 dwords is odd -> conflict-free for ds_read_b32), the serial passes run out of LDS.  Only (cdf[sym], cdf[sym+1]) leave the chip on the encode path: 4 B/node instead of the
// reference's 1020 B PMF row + 512 B CDF row over PCIe.
#include "scp_internal.h"

#define ROWS 64
#define MAXSYM 255

// Workgroup = 4 wavefronts on 64 rows.  Everything that is not order-sensitive - stage-in, row maximum, exp, the division by
// the row sum, the stores - is spread over the four wavefronts (wavefront q owns columns 64 q .. 64 q + 63 of every row: lane =
// row, so the odd row stride keeps ds_read_b32 conflict-free); the two serial float32 passes (row sum, cumsum - both in column
// order: the PMF this library defines and numpyAc's prefix sums) are walked by one lane per row on wavefront 0.  The first
// version ran one wavefront per 65 KB tile, i.e. two wavefronts per CU doing 255 exp and 255 float64 divisions per lane:
// 1.35 ms per 577k-row frame, 0.44 TB/s.  (e / sum: a correctly rounded float32 division - HIP's default, and what the float64
// division rounded to float32 of the first version produced: 53 >= 2 * 24 + 2 bits make that double rounding innocuous.)
template <bool FROM_LOGITS>
__global__ __launch_bounds__(256) void cdf_kernel(const float *__restrict__ in, int64_t ld, int64_t n, int nsym,
                                                  const uint8_t *__restrict__ sym, float *__restrict__ pmf_out,
                                                  uint32_t *__restrict__ lohi, uint16_t *__restrict__ cdf_full) {
    __shared__ float tile[ROWS * MAXSYM];
    __shared__ float part[4 * ROWS];     // per (column quarter, row): partial maxima, then [0 .. 63] the row sums
    const int tid = threadIdx.x, lane = tid & 63, wq = tid >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * ROWS;
    const int nrow = (int)((n - row0) < ROWS ? (n - row0) : ROWS);
    const int total = nrow * nsym;
    // coalesced stage-in
    if (ld == nsym) {
        const float *src = in + row0 * ld;
        for (int i = tid; i < total; i += 256) tile[i] = src[i];
    } else if (ld == 256 && nsym <= 256 && (((uintptr_t)in) & 15) == 0) {
        // 16-byte aligned rows of 256 floats (the coding-order table the probability heads write into): one 16-byte load per lane and row
        for (int r = wq; r < nrow; r += 4) {
            const float4 v = *(const float4 *)(in + (row0 + r) * 256 + 4 * lane);
            const int c = 4 * lane;
            float *d = tile + r * nsym + c;
            if (c < nsym) d[0] = v.x;
            if (c + 1 < nsym) d[1] = v.y;
            if (c + 2 < nsym) d[2] = v.z;
            if (c + 3 < nsym) d[3] = v.w;
        }
    } else {
        for (int i = tid; i < total; i += 256) { const int r = i / nsym, c = i - r * nsym; tile[i] = in[(row0 + r) * ld + c]; }
    }
    __syncthreads();
    float *row = tile + lane * nsym;
    const int c0 = 64 * wq, c1 = (c0 + 64 < nsym) ? c0 + 64 : nsym;      // this wavefront's columns of every row
    if (FROM_LOGITS) {
        float m = -INFINITY;
        if (lane < nrow) for (int j = c0; j < c1; ++j) m = fmaxf(m, row[j]);
        part[wq * ROWS + lane] = m;
        __syncthreads();
        m = fmaxf(fmaxf(part[lane], part[ROWS + lane]), fmaxf(part[2 * ROWS + lane], part[3 * ROWS + lane]));
        if (lane < nrow) for (int j = c0; j < c1; ++j) row[j] = expf(row[j] - m);
        __syncthreads();
        if (wq == 0 && lane < nrow) {   // serial float32 sum in column order; eight LDS reads in flight, then their eight dependent additions
            float sum = 0.f;
            int j = 0;
            for (; j + 8 <= nsym; j += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = row[j + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) sum = __fadd_rn(sum, v[u]);
            }
            for (; j < nsym; ++j) sum = __fadd_rn(sum, row[j]);
            part[lane] = sum;
        }
        __syncthreads();
        if (lane < nrow) {
            const float sum = part[lane];
            for (int j = c0; j < c1; ++j) row[j] = row[j] / sum;          // correctly rounded: the PMF this library defines (float32)
        }
        __syncthreads();
        if (pmf_out) {
            float *dst = pmf_out + row0 * nsym;
            for (int i = tid; i < total; i += 256) dst[i] = tile[i];
            __syncthreads();
        }
    }
    if (wq == 0 && lane < nrow) {
        const int s = sym ? (int)sym[row0 + lane] : 0;
        // numpyAc.py:111 serial float32 cumsum; keep F[s] and F[s+1] (F[0] = 0, F[k] = c[k-1])
        // (round 6: the prefix sums go back into the row - eight reads in flight, eight dependent additions, eight writes - and F[s], F[s + 1] are
        // read from it afterwards: two instructions per element instead of the add + two compare / select pairs of the first version)
        float c = 0.f;
        int j = 0;
        for (; j + 8 <= nsym; j += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = row[j + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) { c = __fadd_rn(c, v[u]); v[u] = c; }
#pragma unroll
            for (int u = 0; u < 8; ++u) row[j + u] = v[u];
        }
        for (; j < nsym; ++j) { c = __fadd_rn(c, row[j]); row[j] = c; }
        const float c_last = c;
        const float c_lo = s > 0 ? row[s - 1] : 0.f, c_hi = row[s < nsym ? s : nsym - 1];
        if (lohi) {
            // :112 c / c[-1] in float32, :113 -> float64, :101-103 * 65281, rint, wrap, :106 + arange
            const double scale = (double)(65536 - nsym);
            const uint32_t lo = (s == 0) ? 0u : ((uint32_t)(int64_t)rint((double)(float)((double)c_lo / (double)c_last) * scale) + (uint32_t)s) & 0xFFFFu;
            uint32_t hi = ((uint32_t)(int64_t)rint((double)(float)((double)c_hi / (double)c_last) * scale) + (uint32_t)(s + 1)) & 0xFFFFu;
            if (s == nsym - 1) hi = 0u;  // 0 encodes 0x10000 (numpyAc_backend.cpp:277)
            lohi[row0 + lane] = lo | (hi << 16);
        }
    }
    if (cdf_full) {
        __syncthreads();
        const double scale = (double)(65536 - nsym);
        const int Lp = nsym + 1;
        for (int i = tid; i < nrow * Lp; i += 256) {
            const int r = i / Lp, k = i - r * Lp;
            uint32_t v = 0;
            if (k > 0) {
                const float last = tile[r * nsym + nsym - 1];
                v = ((uint32_t)(int64_t)rint((double)(float)((double)tile[r * nsym + k - 1] / (double)last) * scale) + (uint32_t)k) & 0xFFFFu;
            }
            cdf_full[(row0 + r) * Lp + k] = (uint16_t)v;
        }
    }
}

static int launch(bool from_logits, const float *in, int64_t ld, int64_t n, int32_t nsym, const uint8_t *sym, float *pmf,
                  uint32_t *lohi, uint16_t *cdf_full, void *stream) {
    if (n == 0) return SCP_OK;
    if (!in || n < 0 || nsym < 2 || nsym > MAXSYM || ld < nsym) return SCP_EINVAL;
    if (lohi && !sym) return SCP_EINVAL;
    const int nb = (int)cdiv64(n, ROWS);
    hipStream_t st = (hipStream_t)stream;
    SCP_PROF(SCP_PROF_CDF, st, (double)n * (4.0 * nsym + 4.0));
    if (from_logits) hipLaunchKernelGGL(cdf_kernel<true>, dim3(nb), dim3(256), 0, st, in, ld, n, nsym, sym, pmf, lohi, cdf_full);
    else hipLaunchKernelGGL(cdf_kernel<false>, dim3(nb), dim3(256), 0, st, in, ld, n, nsym, sym, pmf, lohi, cdf_full);
    LAUNCH_CHECK();
    return SCP_OK;
}

extern "C" int scp_softmax_cdf(const float *logits, int64_t ld, int64_t n, int32_t nsym, const uint8_t *sym, float *pmf,
                               uint32_t *lohi, uint16_t *cdf_full, void *stream) {
    return launch(true, logits, ld, n, nsym, sym, pmf, lohi, cdf_full, stream);
}

extern "C" int scp_pmf_cdf(const float *pmf, int64_t n, int32_t nsym, const uint8_t *sym, uint32_t *lohi, uint16_t *cdf_full,
                           void *stream) {
    return launch(false, pmf, nsym, n, nsym, sym, nullptr, lohi, cdf_full, stream);
}
