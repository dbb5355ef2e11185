// Internal helpers shared by the libscp_hip translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
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

// scp_debug.h launch brackets (api.cpp): `SCP_PROF(tag, stream, work);` right in front of a launch records a hipEvent on the stream now
// and another when the enclosing scope ends (i.e. after the launch).  One relaxed load when profiling is off.
extern int g_scp_prof_on;
struct ScpProfScope {
    int slot;
    hipStream_t st;
    ScpProfScope(int tag, hipStream_t s, double work) : slot(-1), st(s) { if (__builtin_expect(g_scp_prof_on, 0)) begin(tag, work); }
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
