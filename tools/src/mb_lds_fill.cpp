// L2 -> LDS fill rate of one CU on gfx950: 8 waves stream an L2-resident buffer into two LDS stages, one step ahead, a barrier per step
// (the operand pipeline of mlp_fused.hip / gemm_split.hip without the products).  Variants: LDS-DMA (global_load_lds_dwordx4) vs
// global_load_dwordx4 + ds_write_b128 through registers; step = 16 / 32 / 48 / 64 KiB; all CUs busy or a single workgroup.
//   hipcc --offload-arch=gfx950 -O3 -o mb_lds_fill mb_lds_fill.cpp && ./mb_lds_fill
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ROWS variant: the access pattern of mlp_fused.hip / gemm_split.hip - one DMA instruction reads 16 rows x 64 B (a 32-element k-slab
// of 16 rows of a [N][256] bf16 plane: row stride 512 B) instead of 1 KiB of consecutive bytes
template <int STEP_KB>
__global__ __launch_bounds__(512, 2) void fill_rows_kernel(const char *__restrict__ src, size_t src_bytes, int steps, unsigned long long *cyc, float *sink, int desync) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int STEP = STEP_KB * 1024;
    constexpr int PER_WAVE = STEP / 8 / 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float acc = 0.f;
    // the "matrix": rows of 512 B; a step = k-slab ks (64 B of every row) of 16 * 8 * PER_WAVE rows; slabs 0..7 then the next row block
    const int rows_per_step = 16 * 8 * PER_WAVE;
    const size_t nrows = src_bytes / 512;
    const int phase = desync ? (int)((blockIdx.x * 37u) % 61u) : 0;       // workgroups at different positions of the weight stream
    auto issue = [&](int s) {
        const int sp = s + phase;
        const int ks = sp & 7;
        const size_t row0 = ((size_t)(sp >> 3) * rows_per_step) % (nrows - rows_per_step);
        char *l = smem + (s & 1) * STEP;
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) {
            const int piece = w * PER_WAVE + j;
            const size_t row = row0 + piece * 16 + (lane >> 2);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + row * 512 + ks * 64 + (lane & 3) * 16), (lds_ptr_t)(l + piece * 1024), 16, 0, 0);
        }
    };
    issue(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        issue(s + 1);
        acc += *(const float *)(smem + (s & 1) * STEP + tid * 4);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (lane == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
    if (acc == 123.456f) sink[0] = acc;
}

template <int STEP_KB>
static void run_rows(const char *src, size_t bytes, int blocks, unsigned long long *cyc, float *sink, int desync) {
    const int steps = 400;
    const int lds = 2 * STEP_KB * 1024;
    (void)hipFuncSetAttribute((const void *)fill_rows_kernel<STEP_KB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((fill_rows_kernel<STEP_KB>), dim3(blocks), dim3(512), lds, 0, src, bytes, steps, cyc, sink, desync);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[256 * 8];
    (void)hipMemcpy(h, cyc, sizeof(unsigned long long) * blocks * 8, hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < blocks * 8; ++i) m += (double)h[i];
    m /= blocks * 8;
    const double bytes_per_cu = (double)steps * STEP_KB * 1024;
    printf("rows 16x64B%s step %2d KiB, %3d workgroups: %7.0f cycles per step, %6.1f GB/s per CU by wall clock, %6.2f TB/s aggregate\n", desync ? " desync" : "       ",
           STEP_KB, blocks, m / steps, bytes_per_cu / (ms * 1e-3) / 1e9, bytes_per_cu * blocks / (ms * 1e-3) / 1e12);
}

template <int STEP_KB, bool DMA>
__global__ __launch_bounds__(512, 2) void fill_kernel(const char *__restrict__ src, size_t src_bytes, int steps, unsigned long long *cyc, float *sink) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int STEP = STEP_KB * 1024;
    constexpr int PER_WAVE = STEP / 8 / 1024;           // 1 KiB instructions per wave and step
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t wg_off = ((size_t)blockIdx.x * 7919 * 1024) % (src_bytes - (size_t)STEP * 4);
    f32x4 regs[PER_WAVE];
    float acc = 0.f;
    auto issue = [&](int s) {
        const char *g = src + (wg_off + (size_t)s * STEP) % (src_bytes - STEP);
        char *l = smem + (s & 1) * STEP;
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) {
            const int piece = w * PER_WAVE + j;
            if (DMA) __builtin_amdgcn_global_load_lds((glb_ptr_t)(g + piece * 1024 + lane * 16), (lds_ptr_t)(l + piece * 1024), 16, 0, 0);
            else regs[j] = *(const f32x4 *)(g + piece * 1024 + lane * 16);
        }
    };
    auto commit = [&](int s) {
        if (DMA) return;
        char *l = smem + (s & 1) * STEP;
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) *(f32x4 *)(l + (w * PER_WAVE + j) * 1024 + lane * 16) = regs[j];
    };
    issue(0);
    commit(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        issue(s + 1);
        // touch the landed stage (one read per lane) so that nothing is optimised away
        acc += *(const float *)(smem + (s & 1) * STEP + tid * 4);
        if (!DMA) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); commit(s + 1); }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (lane == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
    if (acc == 123.456f) sink[0] = acc;
}

template <int STEP_KB, bool DMA>
static void run(const char *src, size_t bytes, int blocks, unsigned long long *cyc, float *sink) {
    const int steps = 400;
    const int lds = 2 * STEP_KB * 1024;
    hipFuncSetAttribute((const void *)fill_kernel<STEP_KB, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((fill_kernel<STEP_KB, DMA>), dim3(blocks), dim3(512), lds, 0, src, bytes, steps, cyc, sink);
        hipEventRecord(e1);
        hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256 * 8];
    hipMemcpy(h, cyc, sizeof(unsigned long long) * blocks * 8, hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < blocks * 8; ++i) m += (double)h[i];
    m /= blocks * 8;
    const double bytes_per_cu = (double)steps * STEP_KB * 1024;
    printf("%-9s step %2d KiB, %3d workgroups: %7.0f cycles per step (s_memtime units), %6.1f GB/s per CU by wall clock, %6.2f TB/s aggregate\n", DMA ? "LDS-DMA" : "registers",
           STEP_KB, blocks, m / steps, bytes_per_cu / (ms * 1e-3) / 1e9, bytes_per_cu * blocks / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t bytes = 2u << 20;                      // 2 MiB: resident in every XCD's L2 after the first pass
    char *src; unsigned long long *cyc; float *sink;
    hipMalloc(&src, bytes); hipMemset(src, 1, bytes); hipMalloc(&cyc, 8 * 256 * 8); hipMalloc(&sink, 4);
    for (int d : {0, 1}) { run_rows<32>(src, bytes, 256, cyc, sink, d); run_rows<48>(src, bytes, 256, cyc, sink, d); }
    run_rows<48>(src, bytes, 1, cyc, sink, 0);
    for (int blocks : {256, 1}) {
        run<16, true>(src, bytes, blocks, cyc, sink);  run<16, false>(src, bytes, blocks, cyc, sink);
        run<32, true>(src, bytes, blocks, cyc, sink);  run<32, false>(src, bytes, blocks, cyc, sink);
        run<48, true>(src, bytes, blocks, cyc, sink);  run<48, false>(src, bytes, blocks, cyc, sink);
        run<64, true>(src, bytes, blocks, cyc, sink);  run<64, false>(src, bytes, blocks, cyc, sink);
    }
    return 0;
}
