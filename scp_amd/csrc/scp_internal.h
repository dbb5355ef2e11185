// Internal helpers shared by the libscp_hip translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include "../../include/scp.h"
#include "../../include/scp_debug.h"

#define SCP_WAVE 64

extern int g_scp_last_hip_error;

#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) {                        \
            g_scp_last_hip_error = (int)_e;            \
            return SCP_EHIP;                           \
        }                                              \
    } while (0)

#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())

// hipStreamSynchronize without the spin: the runtime busy-waits, and the stage-G read-backs of the pipelined encoder wait for tens of
// milliseconds per frame (their small kernels queue behind the whole-GPU model kernels of the frames in flight: 33 - 43 ms of one core per
// frame).  An event per host thread, polled every 50 us.  (Defined in api.cpp.)
int scp_stream_wait(hipStream_t st);
// Small device -> host read-backs in front of such a wait: hipMemcpyAsync into PAGEABLE host memory does not return before the copy has run
// (the runtime spins there instead), so the bytes go through a pinned staging buffer of the calling thread and scp_stream_wait hands them to
// their destinations.  2-D form: `height` elements of `width` bytes, source pitch `spitch`, packed at the destination.
int scp_d2h_async(void *dst, const void *src, size_t bytes, hipStream_t st);
int scp_d2h_2d_async(void *dst, const void *src, size_t spitch, size_t width, size_t height, hipStream_t st);
// Drop what this thread has queued and not yet waited for.  Every early return between a read-back and its scp_stream_wait goes through
// it (SCP_D2H_TRY), and the builds call it on entry: a queued destination is a stack / local-vector address of the function that queued it.
void scp_d2h_abort();
#define SCP_D2H_TRY(expr)                                   \
    do {                                                    \
        const int _rc = (expr);                             \
        if (_rc) { scp_d2h_abort(); return _rc; }           \
    } while (0)

// scp_debug.h launch brackets (api.cpp): `SCP_PROF(tag, stream, work);` right in front of a launch records a hipEvent on the stream now
// and another when the enclosing scope ends (i.e. after the launch).  One relaxed load when profiling is off.
extern std::atomic<int> g_scp_prof_on;
struct ScpProfScope {
    int slot;
    unsigned gen;
    hipStream_t st;
    ScpProfScope(int tag, hipStream_t s, double work) : slot(-1), gen(0), st(s) { if (__builtin_expect(g_scp_prof_on.load(std::memory_order_relaxed), 0)) begin(tag, work); }
    ~ScpProfScope() { if (__builtin_expect(slot >= 0, 0)) end(); }
    void begin(int tag, double work);
    void end();
};
#define SCP_PROF(tag, stream, work) ScpProfScope _scp_prof((tag), (hipStream_t)(stream), (double)(work))

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// numeric profile of the calling thread's current scp_ctx (api.cpp): 0 / 1, or -1 = no context current (process default)
int scp_ctx_knn_mode();
int scp_ctx_attention_mode();

// LDS-DMA (global_load_lds_*) completes on the VM counter.  hipcc usually drains it in front of a __syncthreads(), but whether it
// does depends on the surrounding code (observed: adding an unrelated, never-executed DMA to a loop removed the wait) - so every
// barrier that publishes DMA data is preceded by an explicit wait.  N = DMA instructions of THIS wave that may stay in flight.
#define SCP_WAIT_DMA(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
// Workgroup barrier that lets N LDS-DMA instructions of this wave stay in flight.  __syncthreads() cannot do that: its release
// fence makes hipcc emit s_waitcnt vmcnt(0) in front of s_barrier, which drains every prefetch deeper than one step (seen in the
// ISA of the first three-stage pipelines: the counted wait was there, followed by the compiler's vmcnt(0)).  LDS traffic of this
// wave (ds_read / ds_write) is drained by lgkmcnt(0); global stores that other waves must see still need __syncthreads().
#define SCP_BARRIER_DMA(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory")

// f16x3 operands (gemm.hip, gemm_split.hip, fused.hip): the power of two that maps a row maximum mx into [2^13, 2^14) and its inverse
// (1 for an all-zero or non-finite row; exponent clamped to +-100)
__device__ __forceinline__ void scp_pow2_scale(float mx, float &sc, float &isc) {
    int e = 0;
    if (mx > 0.f && mx < INFINITY) {
        e = 140 - (int)((__float_as_uint(mx) >> 23) & 0xffu);
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
    }
    sc = __uint_as_float((unsigned)(127 + e) << 23);
    isc = __uint_as_float((unsigned)(127 - e) << 23);
}

// GELU, 0.5 y (1 + erf(y / sqrt 2)) (swin_transformer.py:543-571: HF ACT2FN["gelu"]), round 5 form, ten vector instructions:
//     GELU(y) = max(y, 0) - T(|y|),   T(a) = a Phi(-a) ~= a exp(-beta a^2) / P4(a)
// P4 = a degree-4 minimax fit on [0, 6.5] with the exponent's scale beta free (0.50920: tools/fit_gelu.py), positive on the whole axis;
// beyond 6.5 both T and its stand-in are < 1e-8.  Everything is written in y' = s y with s = sqrt(beta log2 e), so that the exponential
// is v_exp_f32 of -y'^2: one multiplication.  Maximum absolute error of the activation 4.4e-7 (5.8e-7 with float32 rounding,
// at |y| ~ 2.4 where an ulp is 2.4e-7) on |y| <= 20 - the degree-12 erf polynomial of rounds 1 - 4 (24 instructions) had 4.8e-6 in its
// clamped tails and 8.1e-7 inside.  scp_gelu_scaled maps y' to s GELU(y'/ s): the row-chain kernel has s folded into fc1 (weights and
// bias) and 1/s into fc2 on the host (scp_gelu_prescale); the tile epilogues use scp_gelu.
#define SCP_GELU_S 0.857099074f
#define SCP_GELU_INV_S 1.166726264f
#define SCP_GELU_C0 2.000079870223999f
#define SCP_GELU_C1 1.8606146574020386f
#define SCP_GELU_C2 0.35269710421562195f
#define SCP_GELU_C3 -0.11188109964132309f
#define SCP_GELU_C4 0.008278189226984978f
__device__ __forceinline__ float scp_gelu_scaled(float yp) {
    const float a = __builtin_fabsf(yp);
    const float e = __builtin_amdgcn_exp2f(-yp * yp);
    float p = fmaf(SCP_GELU_C4, a, SCP_GELU_C3);
    p = fmaf(p, a, SCP_GELU_C2);
    p = fmaf(p, a, SCP_GELU_C1);
    p = fmaf(p, a, SCP_GELU_C0);
    return fmaf(-(a * e), __builtin_amdgcn_rcpf(p), fmaxf(yp, 0.f));
}
__device__ __forceinline__ float scp_gelu(float y) { return scp_gelu_scaled(y * SCP_GELU_S) * SCP_GELU_INV_S; }

// growable device buffer owned by a handle
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return SCP_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { g_scp_last_hip_error = (int)e; p = nullptr; return SCP_ENOMEM; }
        cap = want;
        return SCP_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return (T *)p; }
};

// sort_u64.hip: stable LSD radix sort of 64-bit keys on the bit ranges [lo, lo+nb) listed in passes.
// keys_in is overwritten (ping-pong); *result points at whichever buffer holds the sorted keys.
struct RadixWorkspace { DevBuf counts; };
int scp_radix_sort_u64(uint64_t *keys_a, uint64_t *keys_b, int64_t n, const int *pass_lo, const int *pass_bits,
                       int npass, RadixWorkspace *ws, hipStream_t st, uint64_t **result, bool first_hist_done = false);
// the [ntiles][256] digit table of the sort's first pass (tile = 4096 consecutive keys), for a producer that fills it itself
uint32_t *scp_radix_counts(RadixWorkspace *ws, int64_t n, int *ntiles_out);
#define SCP_RADIX_TILE 4096
