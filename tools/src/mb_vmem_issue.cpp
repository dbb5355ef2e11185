// What one VMEM / LDS instruction costs a wave that is alone on its SIMD and otherwise issues MFMAs back to back (gfx950).
// A slice = 6 x v_mfma_f32_32x32x16_bf16 on two alternating chains (192 cycles of matrix pipe) + the instruction(s) under test;
// every workgroup = 4 waves (one per SIMD), one workgroup per CU, all CUs; the DMA / load source is a 64 KiB buffer (L2 resident).
//   hipcc -O3 --offload-arch=gfx950 tools/src/mb_vmem_issue.cpp -o /tmp/mb_vmem_issue && /tmp/mb_vmem_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_t;
typedef const __attribute__((address_space(1))) void *glb_t;

template <int V>
__global__ __launch_bounds__(256, 1) void k(const char *src, float *out, unsigned long long *cyc, int iters) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (lane + i)); b[i] = (__bf16)(0.02f * (lane - i)); }
    f32x16 c0, c1;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
    const char *g = src + lane * 16;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 65536, 0x00020000);
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, 0, 0x00020000);   // stores dropped (range check)
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, 1 << 30, 0x00020000);
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    const unsigned la = (unsigned)(uintptr_t)(lds_t)(smem) + lane * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int slot = (it & 7) * 4096 + w * 1024;
        if (V == 1 || V == 2 || V == 7) __builtin_amdgcn_global_load_lds((glb_t)(g + (it & 31) * 1024), (lds_t)(smem + slot), 16, 0, 0);
        if (V == 2) __builtin_amdgcn_global_load_lds((glb_t)(g + ((it + 7) & 31) * 1024), (lds_t)(smem + 32768 + slot), 16, 0, 0);
        if (V == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_t)(smem + slot), 16, lane * 16, (it & 31) * 1024, 0, 0);
        if (V == 4) { const f32x4 t = *(const f32x4 *)(g + (it & 31) * 1024); acc4 += t; }
        if (V == 5) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, acc4), ro, lane * 16, 0, 0);
        if (V == 8) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, acc4), rw, lane * 16 + (blockIdx.x * 4 + w) * 65536 + (it & 63) * 1024, 0, 0);
        if (V == 6 || V == 7) {
            bf16x8 t0_, t1_, t2_, t3_;
            asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(t0_) : "v"(la) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(t1_) : "v"(la) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(t2_) : "v"(la) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(t3_) : "v"(la) : "memory");
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            asm volatile("" :: "v"(t0_), "v"(t1_), "v"(t2_), "v"(t3_));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (V >= 10) {   // the same instructions dealt into the gaps BETWEEN the MFMAs
            bf16x8 t0_, t1_, t2_, t3_;
#define SB __builtin_amdgcn_sched_barrier(0)
            if (V == 10 || V == 13) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SB; c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); SB;
            if (V == 10 || V == 13) asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(t0_) : "v"(la) : "memory");
            if (V == 11 || V == 12 || V == 13) __builtin_amdgcn_global_load_lds((glb_t)(g + (it & 31) * 1024), (lds_t)(smem + slot), 16, 0, 0);
            SB; c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0); SB;
            if (V == 10 || V == 13) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(t1_) : "v"(la) : "memory");
            SB; c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); SB;
            if (V == 10 || V == 13) asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(t2_) : "v"(la) : "memory");
            SB; c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0); SB;
            if (V == 10 || V == 13) asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(t3_) : "v"(la) : "memory");
            if (V == 12) __builtin_amdgcn_global_load_lds((glb_t)(g + ((it + 7) & 31) * 1024), (lds_t)(smem + 32768 + slot), 16, 0, 0);
            SB; c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); SB;
            SB; c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0); SB;
            if (V == 10 || V == 13) asm volatile("" :: "v"(t0_), "v"(t1_), "v"(t2_), "v"(t3_));
            if ((V == 11 || V == 12 || V == 13) && (it & 7) == 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            continue;
        }
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if ((V == 1 || V == 2 || V == 3 || V == 7) && (it & 7) == 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = acc4[0];
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    if (s == 123.456f) out[threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + w] = t1 - t0;
}

template <int V>
void run(const char *name, const char *src, float *out, unsigned long long *cyc, int ncu) {
    const int iters = 4096;
    hipFuncSetAttribute((const void *)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<V>, dim3(ncu), dim3(256), 65536, 0, src, out, cyc, iters);
        hipDeviceSynchronize();
    }
    unsigned long long h[1024];
    hipMemcpy(h, cyc, sizeof(unsigned long long) * ncu * 4, hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < ncu * 4; ++i) sum += (double)h[i];
    printf("%-44s %7.1f cycles per slice (6 MFMAs = 192)\n", name, sum / (ncu * 4) / iters);
}

int main() {
    char *src; float *out; unsigned long long *cyc;
    hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20);
    hipMalloc(&out, (size_t)1 << 30);
    hipMalloc(&cyc, 8 * 1024);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    run<0>("MFMAs only", src, out, cyc, ncu);
    run<1>("+ 1 global_load_lds_dwordx4", src, out, cyc, ncu);
    run<2>("+ 2 global_load_lds_dwordx4", src, out, cyc, ncu);
    run<3>("+ 1 buffer_load_dwordx4 ... lds", src, out, cyc, ncu);
    run<4>("+ 1 global_load_dwordx4 (to registers)", src, out, cyc, ncu);
    run<5>("+ 1 buffer_store_dwordx4 (out of range)", src, out, cyc, ncu);
    run<8>("+ 1 buffer_store_dwordx4 (1 KiB, real)", src, out, cyc, ncu);
    run<6>("+ 4 ds_read_b128", src, out, cyc, ncu);
    run<7>("+ 1 global_load_lds + 4 ds_read_b128", src, out, cyc, ncu);
    run<10>("4 ds_read_b128, one per MFMA gap", src, out, cyc, ncu);
    run<11>("1 global_load_lds in an MFMA gap", src, out, cyc, ncu);
    run<12>("2 global_load_lds in two MFMA gaps", src, out, cyc, ncu);
    run<13>("4 ds_read_b128 + 1 global_load_lds, in gaps", src, out, cyc, ncu);
    return 0;
}
