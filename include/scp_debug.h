/*
 * scp_debug.h - test and measurement hooks of libscp_hip.so.  NOT part of the drop-in boundary (include/scp.h): nothing a product
 * caller needs is here, and nothing here is stable.
 *   - process-default setters of the numeric profile (an encoder / decoder uses an scp_ctx, scp.h): they change what calls WITHOUT a
 *     current context do, for the whole process, and are not thread-safe;
 *   - performance brackets and cycle-stamp buffers of the diagnostic kernel builds (tools/mb_*.py).
 */
#ifndef SCP_DEBUG_H
#define SCP_DEBUG_H
#include "scp.h"

#ifdef __cplusplus
extern "C" {
#endif

/* process defaults of the numeric profile (what SCP_KNN / SCP_ATTN select at load): see scp_ctx in scp.h */
SCP_API int scp_set_knn_mode(int32_t f16x3);
SCP_API int scp_set_attention_mode(int32_t bf16x3);

/* workgroup shape of the packed f16x3 search (identical neighbour lists, a performance bracket for microbenchmarks): 256 (default)
 * = 256-query workgroups on the XCD-affine schedule, one barrier per group of 3 / 4 candidate tiles; 257 / 258 = groups of 2 / 1;
 * +16 = outward sweep order; 128 = 128-query workgroups in launch order, one barrier per tile. */
SCP_API int scp_set_knn_workgroup(int32_t shape);
/* with a device buffer of (blocks * 8 * 4) u64 set, the K = 192 search of shape 256 runs its cycle-stamped build and writes per
 * wave [cycles at barrier + DMA issue, in the MFMA block, in the selection, tiles]; NULL (default) = the product kernel */
SCP_API int scp_knn_debug_buffer(unsigned long long *dev_buf);
/* the same for the row-chain kernels (scp_swin_ln_linear / scp_swin_post_attn): (workgroups * 4 * 8) u64, per wave the cycle sums of
 * the kernel's phases and its tile count (tools/mb_rowchain_probe.py, tools/mb_postattn.py) */
SCP_API int scp_rc_debug_buffer(unsigned long long *dev_buf);
/* test / A/B bracket of the bf16x3 window-attention kernels (tools/mb_attn_planes.py, tests/test_gpu_model.py): 1 (default) = every workgroup
 * sweeps its keys against the fixed reference 0 first (P = exp2(S), no running maximum) and falls back to the standard online softmax only if a
 * row sum left [2^-100, 2^100]; 0 = the standard form only (what the fallback computes).  The two differ in the last bits. */
SCP_API int scp_set_attention_variant(int32_t v);
SCP_API int scp_get_attention_variant(void);   /* 1 = default; anything else is written into the stream's numeric profile (attnv=) */
/* persistent workgroups of the row-chain launches (0 = one per CU of the device): for launches on a stream created with a CU mask
 * (tools/mb_cumask.py), whose CU set is smaller than the device's */
SCP_API int scp_rc_set_grid(int32_t workgroups);
/* which kernel scp_swin_post_attn launches: -1 (default) = by launch size (the wide kernel - 32 rows per workgroup, the waves split the output
 * channels - while its tiles fit the chip twice, the chain kernel beyond), 0 = the chain kernel always, 1 = the wide kernel always.  The two give
 * identical bits per row; the switch is the bracket of the test that asserts it and of timing runs (environment: SCP_RC_WIDE). */
SCP_API int scp_rc_set_wide(int32_t mode);

/* ---- launch brackets: HIP events recorded INSIDE the C ABI, directly around a kernel launch, on the stream it is launched on ----
 * While enabled, every bracketed entry point records one hipEvent immediately before and one immediately after its main kernel
 * launch (no host code of the caller lies between the two records: what the pair measures is the kernel, plus the few microseconds
 * between the two host-side enqueues when the stream's queue is empty).  bench.py builds its live `roofline` from these records; the
 * rocprofv3 --kernel-trace summary of the same command must agree (profiles/).  A record carries the ALGORITHMIC work of the launch
 * as DESIGN.md prices it: flop for the MFMA-bound kernels (2 M N K of the fp32 product a split kernel stands for), bytes for the
 * HBM-bound ones; the kNN searches record their feature count C (the pair count - the sum over 512-row chunks of n x 512 - lives in a
 * device table only the caller can price: flop = 2 C pairs).
 * Not thread-safe against concurrent enable / read; launches of any thread are recorded while enabled.  At most 65536 records. */
enum {
    SCP_PROF_POST_ATTN = 1,    /* rc_post_attn_kernel: 2 M (256*256 + 2*256*1024) flop                                  */
    SCP_PROF_LN_LINEAR = 2,    /* rc_ln_linear_kernel (incl. <KV>): 2 M 256 N flop                                      */
    SCP_PROF_ATTENTION = 3,    /* swin_attn_*_kernel: rows x 2 x 2 x 512 x 256 flop (QK^T and PV over the window)       */
    SCP_PROF_KNN_FEAT = 4,     /* knn_f16x3_* / knn_mfma_kernel<72|96>: work = C (flop = 2 C x pairs)                   */
    SCP_PROF_KNN_POS = 5,      /* knn_mfma_kernel<2,16>: work = 4 (padded feature count)                               */
    SCP_PROF_GEMM_SPLIT = 6,   /* gemm_split_kernel, all variants: 2 M N K flop                                         */
    SCP_PROF_EDGE_MLP = 7,     /* rc_edge_mlp_kernel: 2 M (448*256 + 2*256*256 + 512*256 + 256*256 + 256*128) flop      */
    SCP_PROF_MERGE = 8,        /* rc_merge_kernel: 2 M 512 256 flop                                                     */
    SCP_PROF_EDGE_GATHER = 9,  /* edge_gather_max_kernel: n (3 C' 4 + 4 k) bytes: u, v, out rows once + the index lists        */
    SCP_PROF_CDF = 10,         /* cdf_kernel: n (4 nsym + 4) bytes                                                      */
    SCP_PROF_GEMM_F32 = 11,    /* gemm_f32_kernel: 2 M N K flop                                                         */
    SCP_PROF_GEMM_ROWS = 12,   /* gemm_bf16x3_kernel (fp32 activation rows; F16 form too): 2 M N K flop                 */
    SCP_PROF_SPLIT_ROWS = 13,  /* split_rows_kernel / split_rows_f16: 8 bytes per element                               */
    SCP_PROF_LAYERNORM = 14,   /* layernorm_rows / layernorm_add kernels: 8 bytes per element (+ planes)                */
    SCP_PROF_OA_ATTENTION = 15,/* oa_attn_f16x3_kernel: B x 3 x 2 c^2 D flop; its preparation: OTHER     */
    SCP_PROF_GEOM = 16,        /* scp_quantize / scp_geom_build / context kernels: the call's algorithmic bytes         */
    SCP_PROF_OTHER = 17,
    SCP_PROF_MLP3 = 18,        /* rc_mlp3_kernel (the 256-wide heads, one launch each): 2 M 3 x 256 x 256 flop                    */
    SCP_PROF_NTAGS = 19
};
SCP_API int scp_prof_enable(int32_t on);     /* 1: drop old records and start recording; 0: stop (records stay readable)   */
SCP_API int scp_prof_count(void);            /* records taken since the last enable(1)                                     */
/* waits for every recorded launch, then writes up to cap records in launch order: tag, milliseconds between the two events, work;
 * returns the number written or a negative SCP_E* code */
SCP_API int scp_prof_read(int32_t cap, int32_t *tags, float *ms, double *work);

#ifdef __cplusplus
}
#endif
#endif
