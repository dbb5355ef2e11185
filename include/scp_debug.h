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
/* the same for scp_mlp_split_fused: (workgroups * 8 * 8) u64, per wave [cycles at barriers, phase-1 products, GELU + split, phase-2
 * products, epilogue, row tiles, -, -] */
SCP_API int scp_mlp_debug_buffer(unsigned long long *dev_buf);
/* the same for the row-chain kernels (scp_swin_ln_linear / scp_swin_post_attn): (workgroups * 4 * 8) u64, per wave the cycle sums of
 * the kernel's phases and its tile count (tools/mb_rowchain_probe.py, tools/mb_postattn.py) */
SCP_API int scp_rc_debug_buffer(unsigned long long *dev_buf);
/* persistent workgroups of the row-chain launches (0 = one per CU of the device): for launches on a stream created with a CU mask
 * (tools/mb_cumask.py), whose CU set is smaller than the device's */
SCP_API int scp_rc_set_grid(int32_t workgroups);

#ifdef __cplusplus
}
#endif
#endif
