/*
 * scp.h - C ABI of libscp_hip.so, the MI355X-native SCP encode hot path.
 *
 * Every entry point is plain C (pointers + sizes, no torch / pybind types).  Device pointers are
 * HIP device addresses; `stream` is a hipStream_t passed as void* (NULL = default stream).
 * All functions return 0 on success or a negative SCP_E* code; nothing aborts the process.
 *
 * Each block names the reference interface it replaces (paths relative to luoao-kddi/SCP).
 * INTEGRATION.md shows the binding a reference maintainer would add for each.
 */
#ifndef SCP_H
#define SCP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCP_API __attribute__((visibility("default")))

#define SCP_OK 0
#define SCP_EINVAL (-1)   /* bad argument (NULL, negative size, depth 0, coordinate < 0 ...) */
#define SCP_ENOMEM (-2)   /* host or device allocation failed                               */
#define SCP_ESMALL (-3)   /* caller-provided output buffer too small                        */
#define SCP_EHIP (-4)     /* a HIP runtime call failed (scp_last_hip_error() has the code)   */
#define SCP_ESTATE (-5)   /* calls made out of order on a handle                            */

#define SCP_MAX_DEPTH 21      /* 3*21 = 63 Morton bits                     */
#define SCP_MAX_SEGMENTS 62   /* trees built by one scp_geom_build call    */

/* SCP_ABI_VERSION is what this header describes, scp_version() what the loaded library implements; a caller compares the two once
 * after loading (scp_amd/native.py does) instead of finding out through wrong numbers.  History of incompatible changes:
 *   100  rounds 1 - 2a
 *   200  WEIGHT LAYOUT: every dense entry point (scp_linear_bf16x3, scp_linear_f16x3*, scp_linear_split*, scp_mlp_split_fused and the
 *        row-chain entry points) reads its weight planes in the TILED layout of scp_tile_weight_bf16; planes straight out of
 *        scp_split_weight_bf16 / scp_split_weight_f16 (the version-100 contract) give SCP_OK and wrong products.  Pass every plane
 *        through scp_tile_weight_bf16 once, or load the library with SCP_WTILE=0 in the environment for the row-major contract;
 *        the process-wide numeric-profile setters and the debug hooks moved to scp_debug.h, scp_ctx replaces the setters.
 *   210  scp_mlp_split_fused is gone (round 4: the Swin blocks run on scp_swin_ln_linear / scp_swin_post_attn, which supersede it; two
 *        scp_linear_split calls give its bits); scp_geom_build takes float xyz through scp_geom_build_xyz as well (stage G in one
 *        launch sequence); scp_debug.h gained the launch brackets scp_prof_*.
 *   220  GELU (round 5, numeric profile ehem/5): max(y, 0) - |y| exp(-beta y^2) / P4(|y|) instead of the degree-12 erf polynomial, in every
 *        kernel that applies it; scp_swin_post_attn expects fc1 scaled by scp_gelu_prescale() and fc2 by its inverse.
 *        (additive, no new version: scp_decode_expand, scp_linear_split_f16_max, scp_row_scale_from_max, scp_octattn_attention_f16x3_vmax;
 *        round 6: scp_linear_split_hier2, scp_mlp3_rows - existing entry points keep their bits, the EHEM model's numeric profile moved to ehem/6
 *        because models/packed.py now calls the two new ones.) */
#define SCP_ABI_VERSION 220
SCP_API int scp_version(void);
SCP_API int scp_last_hip_error(void);
/* number of HIP devices visible / name of device 0 (for bench reports) */
SCP_API int scp_device_count(void);
SCP_API int scp_device_name(char *buf, int cap);

/* ------------------------------------------------------------------------------------------------
 * Numeric profile - a property of an encoder / decoder HANDLE, not of the process.
 * Two kernels offer a choice of arithmetic: the 144- / 192-feature kNN searches (dgcnn.py:10-45) run either "f16x3" (rows
 * scaled by a power of two, two f16 terms, three f16 MFMA products, fp32 accumulate: distances within ~1e-6 relative of the
 * fp32 chain; default) or the exact k-ordered fp32 MFMA chain (bit-identical distance values to PyTorch-CPU; the 3-feature
 * position search always is), and the two products inside window attention (swin_transformer.py:443-501) run either as bf16x3
 * splits on bf16 MFMA (default) or on plain fp32 MFMA.  The choice decides the last bits of the logits, i.e. the integer CDFs a
 * decoder must reproduce: an encoder and the decoder of its streams must use one profile (the stream's side-info file names it).
 * A context holds the choice; the calls of a host thread follow that thread's current context (NULL = the process default:
 * f16x3 / bf16x3 unless SCP_KNN=f32 / SCP_ATTN=f32 are set when the library is loaded).  Contexts of different threads, or
 * different contexts made current one after the other, never influence each other.
 * ---------------------------------------------------------------------------------------------- */
typedef struct scp_ctx scp_ctx;
enum { SCP_CTX_KNN_F16X3 = 1, SCP_CTX_ATTENTION_BF16X3 = 2 };
SCP_API int scp_ctx_create(scp_ctx **out);                          /* both keys 1 */
SCP_API int scp_ctx_destroy(scp_ctx *c);
SCP_API int scp_ctx_set(scp_ctx *c, int32_t key, int32_t value);    /* value 0 / 1 */
SCP_API int scp_ctx_get(const scp_ctx *c, int32_t key);             /* 0 / 1, or SCP_EINVAL */
SCP_API int scp_ctx_make_current(const scp_ctx *c);                 /* for the calling thread; NULL = process default */

/* ------------------------------------------------------------------------------------------------
 * Stage G1 - coordinate transform + quantiser
 * replaces: data_preproc/data_preprocess.py:40-68 (proc_pc) / :107-137 (mul_proc_pc),
 *           cart2spher :200-207, cart2cylin :171-177
 * ---------------------------------------------------------------------------------------------- */
enum { SCP_CART = 0, SCP_SPHER = 1, SCP_CYLIN = 2 };

typedef struct scp_quant_info {
    double bin_num;      /* round(rho_max/qs)+1 (0 for SCP_CART)            */
    double qs[3];        /* per-axis step actually used                      */
    double offset[3];    /* per-axis offset subtracted before dividing       */
    int32_t max_coord;   /* max over all points and axes of the integers     */
    int32_t min_coord;   /* min (negative => the cloud does not fit: error)  */
} scp_quant_info;

/* xyz: device float32 [n][3].  q_out: device int32 [n][3].  tr_out (optional, may be NULL): device
 * float32 [n][3] transformed coordinates (rho,phi,theta | rho,phi,z).  Synchronises `stream`
 * internally once (rho_max feeds the step sizes) and fills *info on the host. */
SCP_API int scp_quantize(const float *xyz, int64_t n, int32_t mode, double qs, double cart_offset,
                 int32_t *q_out, float *tr_out, scp_quant_info *info, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Stage G2 - octree serialisation (Morton keys -> radix sort -> per-level nodes -> occupancy)
 * replaces: data_preproc/OctreeCPP/Octree_python_lib.so `genOctreeInterface`
 *           (Octreewarpper.py:17-39,67-71), data_preproc/Octree.py:148-181 (GenOctree),
 *           :184-221 (mullevel_gen_octree), :102-137 / :224-272 (gen_K_parent_seq[_mullevel])
 * ---------------------------------------------------------------------------------------------- */
typedef struct scp_geom scp_geom; /* opaque: owns device workspace, reusable across frames */

typedef struct scp_segment {
    int64_t point_begin;  /* slice [begin, begin+count) of the q array                        */
    int64_t point_count;
    int32_t path_len;     /* rho-shell filter (Octree.py:188): 0 = keep all                   */
    int32_t path_bits;    /* path[0] is the MOST significant of the path_len low bits         */
    int32_t drop_last;    /* 1: the context tables omit the last BFS node (Octree.py:259-262) */
    int32_t reserved;
} scp_segment;

typedef struct scp_segment_info {
    int32_t depth;        /* D = ceil(log2(max+1)) over the unfiltered slice (Octree.py:58)   */
    int32_t max_coord;
    int64_t n_leaves;     /* distinct points kept                                             */
    int64_t n_nodes;      /* all tree nodes (= length of the occupancy code list)             */
    int64_t node_base;    /* first node of this segment in the concatenated tables            */
    int64_t level_count[SCP_MAX_DEPTH + 1]; /* [l] = nodes at tree level l+1                   */
} scp_segment_info;

SCP_API int scp_geom_create(scp_geom **out);
SCP_API int scp_geom_destroy(scp_geom *g);

/* q: device int32 [n][3], non-negative.  Builds every segment's tree structure (sort + counting)
 * and reports sizes; synchronises `stream` internally.  info: host array [nseg]. */
SCP_API int scp_geom_build(scp_geom *g, const int32_t *q, int64_t n, const scp_segment *segs, int32_t nseg,
                   scp_segment_info *info, void *stream);

/* The float front end and the build in ONE launch sequence (stage G1 + G2; two host read-backs instead of two per shell + three):
 * frames: nframes device arrays float32 [n_points[f]][3]; every frame is cut into nshell trees - shell s with step qs[s], the rho-shell
 * path and drop_last of shells[s] (their point_begin / point_count are ignored) - segment index = f * nshell + s.  Every point is
 * transformed once (front_transform_kernel), one kernel then quantises it per shell, applies the shell filter and writes the Morton keys
 * together with the sort's first digit histogram (front_key_kernel).  Same integers as scp_quantize, same trees as scp_geom_build.
 * qinfo / info: host arrays [nframes * nshell]; q_out (optional, may be NULL): device int32 [sum over segments of n_points][3]. */
SCP_API int scp_geom_build_xyz(scp_geom *g, const float *const *frames, const int64_t *n_points, int32_t nframes, int32_t mode,
                               const double *qs, int32_t nshell, double cart_offset, const scp_segment *shells, int32_t *q_out,
                               scp_quant_info *qinfo, scp_segment_info *info, void *stream);

/* Node tables of the last build, all segments concatenated (segment s starts at node_base[s];
 * inside a segment nodes are in BFS order = level by level, Morton order inside a level).
 * Every pointer is a device buffer of total_nodes elements and may be NULL to skip that column.
 *   occ    uint8  1..255  child-occupancy byte (= the code stream, Octreewarpper.py:71)
 *   level  uint8  1..D
 *   octant uint8  1..8
 *   parent int32  index of the parent node in the concatenated table, -1 for a root
 *   pos    int32 [total][3] node origin in leaf units (Octree.py:140-145)                      */
SCP_API int scp_geom_emit_nodes(scp_geom *g, uint8_t *occ, uint8_t *level, uint8_t *octant, int32_t *parent,
                        int32_t *pos, void *stream);

/* leaves (distinct quantised points) of segment `seg` in Morton order: device int32 [n_leaves][3] */
SCP_API int scp_geom_emit_leaves(scp_geom *g, int32_t seg, int32_t *pts, void *stream);

/* The reference's on-disk record (data_preprocess.py:74,81): int64 [rows][4][6] with channels
 * (occ 1..256, level, octant, x, y, z) for (great-grandparent, grandparent, parent, self).
 * rows = n_nodes - drop_last of segment `seg`.  out: device int64. */
SCP_API int scp_geom_krecords_i64(scp_geom *g, int32_t seg, int64_t *out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Stage G3 - per-node context for the models
 * replaces: dataloaders/encode_dataset_ehem.py:52-105, encode_dataset_ehem_mullevel.py:47-85 (EHEM),
 *           dataloaders/encode_dataset.py:32-55 (OctAttention)
 * ---------------------------------------------------------------------------------------------- */
enum { SCP_POS_MINMAX = 0,       /* (p-min)/(max-min+1e-9), scalar min/max per level (spher/cylin) */
       SCP_POS_MINMAX_MUL = 1,   /* same, but the LAST level divides by (max-min) (mul:80)        */
       SCP_POS_POW2 = 2 };       /* p / 2^max_level (Cartesian, ehem:74)                           */

/* For segment `seg`, rows = n_nodes - drop_last:
 *   ctx    uint8 [rows][12]  (level, octant, occ-1 | 255) x (ggp, gp, p, self); levels of the LAST
 *                            tree level are clipped to lidar_level (ehem:86)
 *   pos    float32 [rows][3] normalised self position (row-major; the model API transposes)
 *   sym    uint8 [rows]      occ-1 = the symbol the arithmetic coder encodes
 *   pos_mm int64 [D][2]      (min,max) per level - DEVICE pointer, may be NULL                   */
SCP_API int scp_geom_context_ehem(scp_geom *g, int32_t seg, int32_t pos_mode, int32_t lidar_level,
                          uint8_t *ctx, float *pos, uint8_t *sym, int64_t *pos_mm, void *stream);

/* scp_geom_context_ehem for EVERY segment of the build in one launch: rows of all segments back to back (segment s starts at the sum of
 * the earlier segments' n_nodes - drop_last), pos_mm int64 [sum of depths][2], and the coded symbols in CODING ORDER (encode.py:109-136:
 * every level cut into windows of context_size rows; inside a window the even positions first, then the odd ones) - sym_coded[k] is
 * the symbol the range coder takes k-th, no coding-order index needed.  Any of pos / sym_coded / pos_mm may be NULL. */
SCP_API int scp_geom_context_ehem_all(scp_geom *g, int32_t pos_mode, int32_t lidar_level, int32_t context_size, uint8_t *ctx, float *pos,
                                      uint8_t *sym_coded, int64_t *pos_mm, void *stream);

/* OctAttention: ctx uint8 [rows][12] = (occ-1|255, level, octant) x 4; pos float32 [rows][4][3] =
 * xyz / 2^D for all four rows (no front padding - the window kernel pads on the fly). */
SCP_API int scp_geom_context_octattn(scp_geom *g, int32_t seg, uint8_t *ctx, float *pos, uint8_t *sym, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Stage M - context-model kernels (fp32).  Layouts are row-major, "tokens x channels".
 * replaces the op sequences of models/dgcnn.py:10-71, models/swin_transformer.py:443-501,625-697,
 * models/attention_model.py:58-95 (see DESIGN.md for the mapping)
 * ---------------------------------------------------------------------------------------------- */
/* k nearest neighbours in feature space, per batch item: x [B][n][C] -> idx int32 [B][n][k],
 * the k largest of  2*xi.xj - |xj|^2 - |xi|^2  (dgcnn.py:18-20), ties -> lower index. */
SCP_API int scp_knn_topk(const float *x, int32_t B, int32_t n, int32_t C, int32_t k, int32_t *idx, void *stream);

/* packed ("varlen") form for many windows in one launch: x [total_rows][C], every sequence padded to a multiple of 512 rows;
 * ctab[2c] = first row of the sequence owning 512-row chunk c, ctab[2c+1] = its real length; idx [total_rows][20] holds GLOBAL rows */
SCP_API int scp_knn_topk_packed(const float *x, const int32_t *ctab, int32_t total_rows, int32_t C, int32_t *idx, void *stream);
/* scp_knn_topk_packed with an a-priori pruning bound per row: thr0[row] = a value of (2 x.y - |x|^2 - |y|^2) that at least 20
 * candidates of the row's sequence are known to reach (e.g. the 20th best over last layer's neighbours); same result, fewer
 * list insertions.  thr0 may be NULL. */
SCP_API int scp_knn_topk_packed_bounded(const float *x, const int32_t *ctab, int32_t total_rows, int32_t C, const float *thr0,
                                int32_t *idx, void *stream);

/* edge-conv tail: out[b][i][c] = lrelu_0.2( scale[c] * (sel_j u[b][idx[b][i][j]][c] + v[b][i][c]) + shift[c] ),
 * sel = max when scale[c] >= 0 else min  (== max over j of BN(conv(edge feature)), dgcnn.py:62-71,132-134) */
SCP_API int scp_edge_gather_max(const float *u, const float *v, const int32_t *idx, const float *scale, const float *shift,
                        int32_t B, int32_t n, int32_t Cout, int32_t k, float *out, int32_t out_stride, void *stream);
/* the same with explicit row strides of u and v (floats): the two halves of one [n][2 Cout] product need no copies */
SCP_API int scp_edge_gather_max_ld(const float *u, int64_t ldu, const float *v, int64_t ldv, const int32_t *idx, const float *scale,
                           const float *shift, int32_t B, int32_t n, int32_t Cout, int32_t k, float *out, int32_t out_stride,
                           void *stream);

/* 1-D Swin window attention (window 512, 4 heads x 64): q [B][Lp][ldq], k,v [B][Lp][ldkv] already projected
 * (row strides in floats, >= 256: the operands may be column slices of a fused QKV buffer), Lp a
 * multiple of 512 (zero rows beyond L were padded AFTER LayerNorm, so their q/k/v equal the biases);
 * `shift` in {0,256}: the kernel applies the cyclic roll, the -100 mask of the last window and the
 * relative-position bias table [1023][4]; out [B][Lp][256] in un-rolled token order. */
SCP_API int scp_swin_attention(const float *q, const float *k, const float *v, const float *bias_table,
                       int32_t B, int32_t Lp, int32_t shift, int32_t ldq, int32_t ldkv, float *out, void *stream);

/* Dense layer C = epilogue(A . W^T) on bf16 MFMA with fp32-class accuracy ("bf16x3": x = hi + lo, three products, fp32
 * accumulate; replaces the nn.Linear calls of models/ehem.py / swin_transformer.py:448-452,511,559,571).
 *   scp_split_weight_bf16: W fp32 [N][K] -> bf16 planes hi/lo [Npad][Kpad] (Npad % 256 == 0, Kpad % 32 == 0, zero padded)
 *   scp_linear_bf16x3    : A fp32 [M][lda] (K % 4 == 0), the planes from above passed through scp_tile_weight_bf16 (every dense entry point
 *                          of this library streams TILED weight planes; SCP_WTILE=0: row-major), optional bias[N], residual[M][ldr];
 *                          act: 0 none, 1 LeakyReLU(0.01), 2 GELU(erf), 3 ReLU;  C fp32 [M][ldc]                        */
SCP_API int scp_split_weight_bf16(const float *W, int32_t N, int32_t K, int32_t Npad, int32_t Kpad, void *hi, void *lo, void *stream);
SCP_API int scp_linear_bf16x3(const float *A, int64_t lda, const void *Whi, const void *Wlo, int32_t Kpad, const float *bias,
                      const float *residual, int64_t ldr, float *C, int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t act,
                      void *stream);

/* "f16x3" form of scp_linear_bf16x3: IEEE-half planes (22 significant bits per operand instead of 16) with one power-of-two scale
 * per row of A and per row of W (largest magnitude -> [2^13, 2^14)), undone exactly in the epilogue: the accuracy class of an
 * fp32 FMA chain at the MFMA rate of the bf16 form.  Replaces the nn.Linear calls of models/oct_attention.py:48-83 and
 * models/attention_model.py:58-125 (embeddings scaled by sqrt(600): bf16x3 misses the 1e-3 logit tolerance there).
 *   scp_split_weight_f16: W fp32 [N][K] -> scaled f16 planes hi/lo [Npad][Kpad] (Npad % 128 == 0, Kpad % 32 == 0) and
 *                         inv_scale[Npad] (1 / row scale; 1 on padding rows)
 *   scp_linear_f16x3    : as scp_linear_bf16x3; row_scale_ws = device workspace of 2 M floats (row scales of A, written here) */
SCP_API int scp_split_weight_f16(const float *W, int32_t N, int32_t K, int32_t Npad, int32_t Kpad, void *hi, void *lo,
                                 float *inv_scale, void *stream);
SCP_API int scp_linear_f16x3(const float *A, int64_t lda, const void *Whi, const void *Wlo, const float *w_inv_scale, int32_t Kpad,
                             const float *bias, const float *residual, int64_t ldr, float *C, int64_t ldc, int32_t M, int32_t N,
                             int32_t K, int32_t act, float *row_scale_ws, void *stream);
/* row scales of an activation on their own, and scp_linear_f16x3 with them given: layers that read the same rows share one pass */
SCP_API int scp_row_scale_f16(const float *A, int64_t lda, int32_t M, int32_t K, float *scale, float *inv_scale, void *stream);
SCP_API int scp_linear_f16x3_scaled(const float *A, int64_t lda, const void *Whi, const void *Wlo, const float *w_inv_scale, int32_t Kpad,
                                    const float *bias, const float *residual, int64_t ldr, float *C, int64_t ldc, int32_t M, int32_t N,
                                    int32_t K, int32_t act, const float *scale, const float *inv_scale, void *stream);

/* weight plane [Npad][Kpad] (row-major bf16, scp_split_weight_bf16) -> tiled: 1 KiB blocks ordered [16-row group][32-element k-slab],
 * each block the LDS image of one LDS-DMA instruction (row r at bytes 64 r, its 16-byte chunk q at position q ^ ((r >> 2) & 3)). */
SCP_API int scp_tile_weight_bf16(const void *plane, int32_t Npad, int32_t Kpad, void *tiled, void *stream);
/* OctAttention's dense layers on PRE-SPLIT f16 planes (oct_attention.py:48-83, attention_model.py:97-125): scp_split_rows_f16 writes, once
 * per activation tensor, what scp_linear_f16x3 converts in every tile that reads a row - the power-of-two row scale (largest magnitude
 * into [2^13, 2^14)), its inverse, and the two IEEE-half planes of the scaled row ([M][ldp], ldp >= K rounded up to 32, ldp % 8 == 0,
 * padding columns zero; K % 4 == 0, K <= 1024); scp_linear_split_f16 streams those planes by LDS-DMA (weights: scp_split_weight_f16 with
 * Npad % 256 == 0, then scp_tile_weight_bf16 per plane; act 0 none / 3 ReLU; C = act(A W^T + bias) + residual).  Bit-identical to
 * scp_linear_f16x3_scaled on the same fp32 rows. */
SCP_API int scp_split_rows_f16(const float *A, int64_t lda, int32_t M, int32_t K, void *hi, void *lo, int64_t ldp, float *scale, float *inv_scale,
                               void *stream);
/* scp_layernorm_add (below) that also writes the f16x3 operand of its output in the same pass - what scp_split_rows_f16 would make of `out` */
SCP_API int scp_layernorm_add_split_f16(const float *a, const float *b, int64_t rows, int32_t C, const float *gamma, const float *beta, float eps,
                                        float *out, void *hi, void *lo, int64_t ldp, float *scale, float *inv_scale, void *stream);
SCP_API int scp_linear_split_f16(const void *Ahi, const void *Alo, int64_t lda, const float *a_inv_scale, const void *Whi, const void *Wlo,
                                 const float *w_inv_scale, int32_t Npad, int32_t Kpad, const float *bias, const float *residual, int64_t ldr,
                                 float *C, int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t act, int32_t cfg, void *stream);
/* scp_linear_split_f16 that also takes maxima of its OUTPUT in the epilogue (round 5; atomicMax on the bit patterns of |C|, the caller zeroes the
 * words before the call): row_max[M] = max |C[m][:]| (may be NULL), col_max (one word, may be NULL) = max |C[m][n]| over m < col_rows,
 * col_lo <= n < col_hi.  The power-of-two scales of the next f16x3 layer come from them without a pass over C: scp_row_scale_from_max gives what
 * scp_row_scale_f16 gives on C (attention_model.py:97-125: linear1 -> ReLU -> linear2), scp_octattn_attention_f16x3_vmax takes col_max of the
 * key | value projection for max |v|.  C is the same, bit for bit. */
SCP_API int scp_linear_split_f16_max(const void *Ahi, const void *Alo, int64_t lda, const float *a_inv_scale, const void *Whi, const void *Wlo,
                                     const float *w_inv_scale, int32_t Npad, int32_t Kpad, const float *bias, const float *residual, int64_t ldr,
                                     float *C, int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t act, int32_t cfg, uint32_t *row_max,
                                     uint32_t *col_max, int32_t col_lo, int32_t col_hi, int32_t col_rows, void *stream);
SCP_API int scp_row_scale_from_max(const uint32_t *row_max, int32_t M, float *scale, float *inv_scale, void *stream);
/* The same dense layer with the ACTIVATION pre-split too: A arrives as bf16 planes hi/lo [M][lda] (lda % 8 == 0, lda >= Kpad,
 * columns K..Kpad zero) written by the producing kernel (scp_split_rows, scp_layernorm_rows_split, the attention kernels, or this
 * function's own split output), so operand tiles go global -> LDS by LDS-DMA with no conversion.  Outputs: C fp32 [M][ldc]
 * and/or planes Ohi/Olo [M][ldo] (columns N..round32(N) zero filled).  Weight planes must be padded to Npad % 256 == 0 and are
 * expected in the TILED layout of scp_tile_weight_bf16 (all scp_linear_split* entry points and the row-chain entry points; the environment
 * variable SCP_WTILE=0 switches the library to row-major [Npad][Kpad] planes, for A/B measurements).
 * cfg: 0 automatic, 1 = 256 x 256 tile, 2 = 256 x 128 tile.  Results are bit-identical to scp_linear_bf16x3 on the same values.
 *   scp_split_rows: fp32 rows -> planes, optionally gathered: out[r] = split(src[idx ? idx[r] : r]); idx == n_src -> zero row */
SCP_API int scp_linear_split(const void *Ahi, const void *Alo, int64_t lda, const void *Whi, const void *Wlo, int32_t Npad, int32_t Kpad,
                     const float *bias, const float *residual, int64_t ldr, float *C, int64_t ldc, void *Ohi, void *Olo, int64_t ldo,
                     int32_t M, int32_t N, int32_t K, int32_t act, int32_t cfg, void *stream);

/* scp_linear_split with a GATHERED residual added BEFORE the activation: out[m] = act(A[m].W^T + bias + residual[res_map[m]])
 * (res_map may be NULL = identity).  Used to evaluate a layer over concat_states (ehem.py:75-86) as one product per Swin stage. */
SCP_API int scp_linear_split_gather(const void *Ahi, const void *Alo, int64_t lda, const void *Whi, const void *Wlo, int32_t Npad, int32_t Kpad,
                            const float *bias, const float *residual, int64_t ldr, const int64_t *res_map, float *C, int64_t ldc, void *Ohi,
                            void *Olo, int64_t ldo, int32_t M, int32_t N, int32_t K, int32_t act, int32_t cfg, void *stream);
/* round 6: the two FINEST stages of such a layer in one launch (models/ehem.py:75-86,100-136: concat_states feeding ancient_mlp / prob_pred_mlp2):
 *   out[m] = act(A0[m] . W0^T + A1[parent[m]] . W1^T + bias + res[res_map[m]])  as split planes,
 * parent[m] = the stage-1 row of stage-0 row m (token t -> token t >> 1 of the same window; consecutive inside a 256-row tile), A1 with 256 columns,
 * M % 256 == 0, N % 4 == 0, weights as tiled planes.  The stage-1 product never exists in memory (it is computed per tile and stays in the
 * accumulators); summation order of an element: stage-1 k ascending, stage-0 k ascending, + bias, + residual. */
SCP_API int scp_linear_split_hier2(const void *A0hi, const void *A0lo, int64_t lda0, int32_t K0pad, const void *A1hi, const void *A1lo, int64_t lda1,
                                   int64_t M1, const void *W0hi, const void *W0lo, const void *W1hi, const void *W1lo, int32_t Npad,
                                   const int64_t *parent, const float *bias, const float *res, int64_t ldr, const int64_t *res_map, void *Ohi,
                                   void *Olo, int64_t ldo, int32_t M, int32_t N, int32_t act, void *stream);
/* scp_linear_split with SCATTERED fp32 output rows: row m goes to C row out_map[m] (negative: dropped). */
SCP_API int scp_linear_split_scatter(const void *Ahi, const void *Alo, int64_t lda, const void *Whi, const void *Wlo, int32_t Npad, int32_t Kpad,
                             const float *bias, const int64_t *out_map, float *C, int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t act,
                             int32_t cfg, void *stream);
SCP_API int scp_split_rows(const float *src, int64_t ld_src, int64_t n_src, const int64_t *idx, int32_t C, void *hi, void *lo, int64_t ldo,
                   int64_t rows, void *stream);
/* producers that write the split format directly (same arithmetic as their fp32 forms, then hi = bf16(y), lo = bf16(y - hi)) */
SCP_API int scp_layernorm_rows_split(const float *x, int64_t ldx, int64_t n_src_rows, const int64_t *ia, const int64_t *ib, int32_t C,
                             const float *gamma, const float *beta, const float *valid, float eps, void *ohi, void *olo, int64_t ldo,
                             int64_t rows, void *stream);
/* out = LayerNorm(a + b) over rows of C floats (C % 4 == 0, C <= 1024; b may be NULL): the `norm(x + residual)` of
 * models/attention_model.py:117,123 (OctAttention, C = 600) in one pass. */
SCP_API int scp_layernorm_add(const float *a, const float *b, int64_t rows, int32_t C, const float *gamma, const float *beta, float eps,
                              float *out, void *stream);
SCP_API int scp_swin_attention_packed_split(const float *q, const float *k, const float *v, const float *bias_table, const int32_t *wtab,
                                    int32_t total_windows, int32_t shift, int32_t ldq, int32_t ldkv, void *ohi, void *olo, int64_t ldo,
                                    void *stream);


/* packed form: total_windows windows of 512 rows; wtab[2w] = first row of the sequence owning window w, wtab[2w+1] = its padded length */
SCP_API int scp_swin_attention_packed(const float *q, const float *k, const float *v, const float *bias_table, const int32_t *wtab,
                              int32_t total_windows, int32_t shift, int32_t ldq, int32_t ldkv, float *out, void *stream);

/* Token-wise glue of the packed forward (csrc/fused.hip):
 *   scp_layernorm_rows: out[r] = valid[r] * LayerNorm_C(row), C in {256, 512}, eps as given; row = x[r] (ia == NULL), x[ia[r]] (C 256)
 *                       or cat(x[ia[r]], x[ib[r]]) (C 512, patch merging swin_transformer.py:350-367); an index == n_src_rows is a zero row;
 *                       valid (nullable) zeroes the rows a window pads after LayerNorm (swin_transformer.py:638-641)
 *   scp_gather_rows   : out[r][0:C] = src[idx[r]][0:C] with independent row strides (concat_states ehem.py:75-86, even/odd split :113)  */
SCP_API int scp_layernorm_rows(const float *x, int64_t ldx, int64_t n_src_rows, const int64_t *ia, const int64_t *ib, int32_t C,
                       const float *gamma, const float *beta, const float *valid, float eps, float *out, int64_t ldo, int64_t rows,
                       void *stream);
SCP_API int scp_gather_rows(const float *src, int64_t lds, const int64_t *idx, int32_t C, float *out, int64_t ldo, int64_t rows, void *stream);

/* exact fp32 dense layer (k-ordered FMA chain per output, independent of the number of rows in the launch): C = act(A . W^T + bias),
 * any K / N; used for the small-K layers and for every feature that feeds a kNN search */
SCP_API int scp_linear_f32(const float *A, int64_t lda, const float *W, const float *bias, float *C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                   int32_t act, void *stream);

/* OctAttention dual-stream causal attention (attention_model.py:58-95): heads of width hd,
 * q_u,k,k_u,v,v_u [B][c][H*hd] -> out, out_u [B][c][H*hd] */
SCP_API int scp_octattn_attention(const float *q_u, const float *k, const float *k_u, const float *v, const float *v_u,
                          int32_t B, int32_t c, int32_t H, int32_t hd, float *out, float *out_u, void *stream);

/* The same attention for head width 150 on f16 MFMA with fp32-class accuracy ("f16x3": every operand as two IEEE-half planes
 * with power-of-two scales - per (token, head) for q and k, one per launch for v - three MFMA products per product, fp32
 * accumulate; csrc/octattn_f16.hip).  A preparation kernel lays the planes out as LDS-DMA-ready key tiles in `workspace`
 * (device memory, 1 KiB aligned, scp_octattn_f16x3_ws_bytes(B, c, H) bytes).  q_u, k, v must be 16-byte aligned. */
SCP_API int64_t scp_octattn_f16x3_ws_bytes(int32_t B, int32_t c, int32_t H);
SCP_API int scp_octattn_attention_f16x3(const float *q_u, const float *k, const float *k_u, const float *v, const float *v_u,
                                        int64_t ldkv, int32_t B, int32_t c, int32_t H, int32_t hd, float *out, float *out_u, void *workspace,
                                        int64_t ws_bytes, void *stream);   /* ldkv: row stride (floats) of k, k_u, v, v_u - H * hd for dense
                                        rows, 1280 when they are column slices of one key | value projection (round 4); q_u, out, out_u dense */
/* the same with max |v| over the B * c rows of v GIVEN (device word: the bit pattern of a finite non-negative float, e.g. col_max of the
 * scp_linear_split_f16_max call that wrote v): no pass over v */
SCP_API int scp_octattn_attention_f16x3_vmax(const float *q_u, const float *k, const float *k_u, const float *v, const float *v_u,
                                             int64_t ldkv, int32_t B, int32_t c, int32_t H, int32_t hd, float *out, float *out_u, void *workspace,
                                             int64_t ws_bytes, const uint32_t *vmax_bits, void *stream);
/* OctAttention's input stage in one launch (oct_attention.py:48-66): embeddings of the four ancestors + Linear(3 -> d_pos) of their
 * positions, concatenated to D = 4 (d_occ + d_lvl + d_oct + d_pos) <= 768 channels, scaled by sqrt(D), plus the position table pe [c][D];
 * both streams (1 = "unknown": occ_enc[255] for the node's own occupancy).  ctx uint8 [n][12] = (occ, level, octant) x 4, pos fp32
 * [n][4][3], row r is position r % c of its window; levels are shifted / clipped as the reference does (level_cap 12, or 10 for obj).
 * Out: E fp32 [2][n][D] and E's f16x3 operand (planes [2 n][ldp] + row scales: what scp_split_rows_f16 makes of E). */
SCP_API int scp_octattn_embed(const uint8_t *ctx, const float *pos, int64_t n, int32_t c, const float *occ_enc, int32_t d_occ,
                              const float *level_enc, int32_t d_lvl, int32_t max_level, const float *octant_enc, int32_t d_oct,
                              const float *pos_w, const float *pos_b, int32_t d_pos, const float *pe, int32_t level_cap, float *E,
                              void *hi, void *lo, int64_t ldp, float *scale, float *inv_scale, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Stage C - softmax -> integer CDF, on device
 * replaces: torch.softmax at encode.py:126-127 + numpyAc/numpyAc.py:109-114,80-107
 * ---------------------------------------------------------------------------------------------- */
/* logits [n][nsym] (row stride `ld` floats) -> pmf [n][nsym] (may be NULL) and, if sym != NULL,
 * lohi [n]: low 16 bits = cdf[sym], high 16 bits = cdf[sym+1] (0 encodes 65536 for the top symbol).
 * cdf_full (may be NULL): uint16 [n][nsym+1] complete table (decoder / tests). */
SCP_API int scp_softmax_cdf(const float *logits, int64_t ld, int64_t n, int32_t nsym, const uint8_t *sym,
                    float *pmf, uint32_t *lohi, uint16_t *cdf_full, void *stream);
/* same, starting from a float32 PMF table (bit-exact numpyAc integer CDF) */
SCP_API int scp_pmf_cdf(const float *pmf, int64_t n, int32_t nsym, const uint8_t *sym, uint32_t *lohi,
                uint16_t *cdf_full, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Range coder (host, serial) - replaces numpyAc/backend/numpyAc_backend.cpp
 *   encode_cdf :327-334 / encode :245-323  ->  scp_ac_encode_cdf, scp_ac_encode_lohi
 *   class decode :134-217                  ->  scp_ac_dec_*
 * ---------------------------------------------------------------------------------------------- */
SCP_API int scp_ac_encode_cdf(const uint16_t *cdf, const int16_t *sym, int64_t n, int32_t Lp, uint8_t *out, size_t cap,
                      size_t *out_len);
SCP_API int scp_ac_encode_lohi(const uint32_t *lohi, int64_t n, uint8_t *out, size_t cap, size_t *out_len);
typedef struct scp_ac_dec scp_ac_dec;
SCP_API int scp_ac_dec_new(scp_ac_dec **d, const uint8_t *stream, size_t len, int32_t Lp);
SCP_API int scp_ac_dec_next(scp_ac_dec *d, const uint16_t *cdf_row); /* returns the symbol (>= 0) */
SCP_API int scp_ac_dec_run(scp_ac_dec *d, const uint16_t *cdf, int64_t n, int16_t *out); /* n symbols, row i = CDF of symbol i */
SCP_API int scp_ac_dec_free(scp_ac_dec *d);

/* ------------------------------------------------------------------------------------------------
 * Legacy octree ABI - the ten symbols data_preproc/OctreeCPP/Octreewarpper.py:17-39 binds, so the
 * reference's own wrapper can load libscp_hip.so in place of Octree_python_lib.so.
 * ---------------------------------------------------------------------------------------------- */
typedef struct scp_legacy_node { uint32_t nodeid, octant, parent; uint8_t oct; uint32_t pos[3]; } scp_legacy_node;
SCP_API void *new_vector(void);
SCP_API void delete_vector(void *v);
SCP_API int vector_size(void *v);
SCP_API void *vector_get(void *v, int level);
SCP_API void vector_push_back(void *v, int i);
SCP_API void *genOctreeInterface(void *v, const double *xyz, int n);
SCP_API int Nodes_size(void *level);
SCP_API scp_legacy_node *Nodes_get(void *level, int i);
SCP_API int int_size(void *codes);
SCP_API int int_get(void *codes, int i);

/* ---- distortion metrics of the quantiser -------------------------------------------------------------
 * d2[i] = min_j |a_i - b_j|^2 in float64 (device pointers, row-major [n][3]).  Replaces the KD-tree nearest-neighbour queries
 * of data_preproc/pt.py:88-95 (distChamfer) and of the MPEG pc_error tool behind the D1 PSNR (pt.py:13-85,
 * utils/__init__.py:3-15); the host side (scp_amd/metrics.py) forms chamfer = max(mean sqrt d2_ab, mean sqrt d2_ba) and
 * PSNR = 10 log10(3 peak^2 / max(mean d2_ab, mean d2_ba)). */
SCP_API int scp_nn_sqdist_f64(const double *a, int64_t na, const double *b, int64_t nb, double *d2, void *stream);

/* Input stage of the packed EHEM forward: embeddings of dgcnn.py:121-128 fused with the packed layout's input gather.
 * ctx u8 [T][12] = 4 x (level, octant, occ) (scp_geom_context_ehem), pos f32 [T][3], inmap i64 [rows] (== n_tokens: pad token);
 * tables occ_enc [256][16], level_enc [*][4], octant_enc [*][4]; out x f32 [rows][80], pos_out f32 [rows][3], occ_self i64 [rows]. */
SCP_API int scp_embed_gather(const uint8_t *ctx, const float *pos, const int64_t *inmap, int64_t n_tokens, const float *occ_enc,
                     const float *level_enc, const float *octant_enc, float *x, float *pos_out, int64_t *occ_self, int64_t rows,
                     void *stream);

/* ---- index maps of the packed ("varlen") EHEM forward ------------------------------------------------------------------
 * lengths[W] (host): the window lengths of one packed chunk (encode.py:109-136 cuts every level into windows of <= 8192 nodes).
 * scp_packed_plan_sizes: rows of the 11 layouts (self stages 0..4, cross stages 0..3, even outputs, odd outputs; every window
 * padded to x512 rows per Swin stage).  scp_packed_plan: writes all 56 maps in one launch; outs[] = device buffers in this order:
 *   inmap i64[rows0]; a1map, a2map i64[rowsX0]; even_rows i64[E]; odd_rows i64[O]; even_dst i64[E]; odd_dst i64[O];
 *   self merge (even, odd) i64[rows s+1] for s = 0..3; cross merge (even, odd) for s = 0..2; self concat i64[rows0] for s = 1..4;
 *   cross concat i64[rowsX0] for s = 1..3; window tables i32[rows/512][2] (base, padded length) of the 9 stage layouts;
 *   kNN table i32[rows0/512][2] (base, real length); valid f32[rows] of the 9 stage layouts; parent-row maps i64[rows s] (stage s
 *   token t -> stage s+1 token t >> 1) for self s = 0..3 and cross s = 0..2; coded positions i64[rowsX0] of the even / odd token
 *   of every cross-stage-0 row (-1 for padding rows).
 * Replaces the per-window Python bookkeeping of models/ehem.py:88-136 / swin_transformer.py:350-367,638-641 in the packed forward;
 * tables_dev: int64 scratch of 36 * W entries. */
SCP_API int scp_packed_plan_sizes(const int64_t *lengths, int32_t W, int64_t *rows_out);
SCP_API int scp_packed_plan(const int64_t *lengths, int32_t W, int64_t *tables_dev, void *const *outs, int32_t n_outs, void *stream);

/* ---- decoder: the children of one decoded octree level + the model inputs of the level they form, one launch ------------------
 * replaces: decode_ehem_mullevel.py:100-130 (decode_ehem.py likewise): child occupancy bits -> (parent, digit) pairs in breadth-first
 * order, the ancestor window shifted by one, cal_pos_ary (:41-53), the next level's context rows and normalised positions.
 *   sym i64[n]: decoded symbols of the parents (occupancy - 1; -1 = unknown: no children); cum i64[n]: INCLUSIVE scan of popcount(sym + 1);
 *   pos i32[n][3] node origins; anc u8[n][9] = (level, octant, symbol) of (ggp, gp, p), 255 = pad; octant u8[n]; L = the parents' level;
 *   shift = depth - L.  Outputs for the m = cum[n - 1] children: cpos i32[m][3], canc u8[m][9], coct u8[m], cctx u8[m][12] (ancestor
 *   levels clamped to lv_clamp, own entry (lv_next, octant, 255)), cposn f32[m][3] = polar ? (pos - mn) / den : pos / den (double
 *   arithmetic, rounded once), occ8 u8[n] = the parents' occupancy codes. */
SCP_API int scp_decode_expand(const int64_t *sym, const int64_t *cum, const int32_t *pos, const uint8_t *anc, const uint8_t *octant, int64_t n,
                              int32_t L, int32_t shift, int32_t lv_next, int32_t lv_clamp, int32_t polar, double mn, double den, int32_t *cpos,
                              uint8_t *canc, uint8_t *coct, uint8_t *cctx, float *cposn, uint8_t *occ8, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Swin blocks on the row-chain kernels (csrc/rowchain.hip): a workgroup keeps 128 token rows in registers as the B operand of
 * every product, weights stream through LDS, and the accumulator of one product is the B fragment of the next.
 * replaces: models/swin_transformer.py:654-706 (SwinLayer.forward) around the attention kernel -
 *   scp_swin_ln_linear : layernorm_before + query|key|value (:443-501,654-660), or layernorm(query) + query of a cross layer:
 *                        out[m] = valid[m] * n(x[m]) . W'^T + bias + valid[m] * wbeta, n = the row normalised WITHOUT affine;
 *                        the caller folds the LayerNorm affine into the weight: W' = W diag(gamma), wbeta = W beta.  valid (may be
 *                        NULL) zeroes the rows a window pads AFTER LayerNorm (:638-641): they come out as the bias alone.
 *                        x fp32 [M][ldx] (256 channels); W' as scp_split_weight_bf16 + scp_tile_weight_bf16 planes [Npad][256];
 *                        N % 128 == 0, N <= 1024; out fp32 [M][ldo].
 *   scp_swin_post_attn : attention.output.dense + residual, layernorm_after, intermediate.dense + GELU, output.dense + residual
 *                        (:503-571,662-706) in ONE launch: x2 = x1 + fc2(GELU(fc1(LN(x1)))), x1 = x + proj(o).  o: the attention
 *                        output as bf16 hi/lo planes [M][ldo_in]; x fp32 [M][ldx]; out fp32 [M][ldc] (may be x).  W: ONE buffer of
 *                        scp_swin_post_attn_weight_bytes() bytes = tiled planes proj hi | fc1 hi | fc2 hi | proj lo | fc1 lo | fc2 lo,
 *                        proj [256][256], fc1 = (W1 diag(gamma))[:, P] [1024][256], fc2 = W2[:, P] [256][1024], P = inside every
 *                        16 columns, columns 4-7 and 8-11 change places (the order in which an MFMA accumulator holds a row's
 *                        channels); b1 = fc1 bias + W1 beta.  Since version 220 the kernel evaluates GELU in the variable s y
 *                        (s = scp_gelu_prescale()): the caller multiplies fc1 (weight and b1) by s and divides fc2's weight by s.
 * Results are per row: independent of M, of the row's position and of what else is in the launch.
 * ----------------------------------------------------------------------------------------------  * tile_list (device int32 [n_tiles], ascending; may be NULL = every tile): the 128-row tiles to process.  The packed forward passes the
 * tiles that hold at least one real row: tiles of nothing but window padding (4.7 % of a level-16 multi-level frame) keep their old
 * contents - run it in place (out == x) so that these stay finite. */
SCP_API int scp_swin_ln_linear(const float *x, int64_t ldx, const float *valid, const void *Whi, const void *Wlo, const float *bias,
                               const float *wbeta, float eps, float *out, int64_t ldo, int32_t M, int32_t N, void *stream);
SCP_API int scp_swin_post_attn(const void *Ohi, const void *Olo, int64_t ldo_in, const float *x, int64_t ldx, const void *W, const float *bp,
                               const float *b1, const float *b2, float eps, float *out, int64_t ldc, int32_t M, const int32_t *tile_list,
                               int32_t n_tiles, void *stream);
SCP_API int64_t scp_swin_post_attn_weight_bytes(void);
SCP_API double scp_gelu_prescale(void);

/* SwinPatchMerging (swin_transformer.py:350-384) in one launch: out[m] = LayerNorm(cat(x[ia[m]], x[ib[m]])) . W^T for M merged rows (the
 * reduction has no bias).  x: fp32 [n_src][ldx] (256 channels; an index equal to n_src stands for a row of zeros - the pad of an odd
 * window); W: a buffer of scp_swin_post_attn_weight_bytes() bytes holding the tiled planes (scp_split_weight_bf16 + scp_tile_weight_bf16) of
 * (W diag(gamma))[:, 0:256] at byte 0 and of (W diag(gamma))[:, 256:512] at byte 131072, their lo planes at the same offsets behind the
 * first half of the buffer; wbeta = W beta [256]; out fp32 [M][ldo]. */
SCP_API int scp_swin_merge(const float *x, int64_t ldx, int64_t n_src, const int64_t *ia, const int64_t *ib, const void *W, const float *wbeta,
                           float eps, float *out, int64_t ldo, int32_t M, void *stream);

/* The two edge MLPs of the geometry feature generator (dgcnn.py:121-151: edge_mlp1 448 -> 256 -> 256 -> 256 on cat(pos1, pos2, pos3), edge_mlp2
 * 512 -> 256 -> 256 -> 128 on cat(pos3, edge_mlp1(...)), LeakyReLU(0.01) between the layers) for M points in one launch: six dense layers
 * chained through the MFMA accumulators.  pos1 / pos2 / pos3: fp32 [M][64 | 128 | 256] (row strides ld1 / ld2 / ld3); W: a buffer of
 * scp_swin_post_attn_weight_bytes() bytes with eight tiled [256][256] matrices (hi planes at m * 131072 bytes, lo planes at the same
 * offsets behind the first half of the buffer): 0 / 1 = edge_mlp1 layer 1, input columns [0, 256) / [256, 448) + 64 zero columns;
 * 2, 3 = its layers 2, 3; 4 / 5 = edge_mlp2 layer 1, input columns [256, 512) / [0, 256); 6, 7 = its layers 2, 3 (7: 128 rows); the
 * input columns of matrices 2, 3, 4, 6, 7 in accumulator order (inside every group of 16: columns 4 - 7 and 8 - 11 exchanged);
 * bias: the six bias vectors back to back (1408 floats); out: fp32 [M][ldo], 128 columns written. */
SCP_API int scp_geo_edge_mlps(const float *pos1, int64_t ld1, const float *pos2, int64_t ld2, const float *pos3, int64_t ld3, const void *W,
                              const float *bias, float *out, int64_t ldo, int32_t M, void *stream);
/* round 6: a three-layer head Sequential(Linear, LeakyReLU(0.01), Linear, LeakyReLU, Linear) on 256-channel rows in ONE launch (models/ehem.py:113-121:
 * prob_pred_mlp1 256 -> 256 -> 256 -> 255 and pre_attn_mlp 256 -> 256 -> 240 -> 240): out[out_map ? out_map[m] : m][0 .. N) = head(x[in_map ? in_map[m] : m]),
 * rows with out_map[m] < 0 dropped.  x: fp32 [n_src][ldx]; W: scp_swin_post_attn_weight_bytes() bytes with three tiled [256][256] matrices (hi planes at
 * m * 131072 bytes, lo planes at the same offsets in the second half; layers 2, 3: columns in accumulator order; unused rows / columns zero); bias [768];
 * N % 4 == 0, N <= 256, 16-byte aligned output rows.  Every row's result is independent of what else is in the launch. */
SCP_API int scp_mlp3_rows(const float *x, int64_t ldx, int64_t n_src, const int64_t *in_map, const void *W, const float *bias, float *out, int64_t ldo,
                          const int64_t *out_map, int32_t M, int32_t N, void *stream);

/* Keys and values handed from the projection to the window attention as bf16 hi / lo PLANES in the layout of the attention kernel's own
 * LDS tiles (swin_transformer.py:443-501; round 3).  planes: [4][Tp][256] bf16 = K hi, K lo, V^T hi, V^T lo for Tp rows of the packed
 * layout (Tp % 32 == 0; windows of 512 rows):
 *   K   [row][4 heads][64 d], the eight 16-byte chunks of a head's 128 bytes at position chunk ^ (row & 7);
 *   V^T [32-row block][head][32 super-rows of 128 B]: head dim d, 8-key chunk c at super-row d >> 1, slot ((d & 1) * 4 + c) ^ ((d >> 1) & 7),
 *       the block's keys in the order p = 16 c' + 8 h + j  <->  key 16 c' + 4 h + (j & 3) + 8 (j >> 2).
 * The attention kernel copies whole 1 KiB pieces of them into LDS by LDS-DMA: no conversion, no staging registers.
 *   scp_swin_kv_planes              : fp32 k, v [rows][ldkv] -> planes (a plain conversion pass; rows % 32 == 0)
 *   scp_swin_ln_qkv                 : scp_swin_ln_linear for N = 768 (query | key | value; q fp32 [M][ldq]) or N = 512 (key | value, q NULL)
 *                                     writing the key / value heads as planes straight from the accumulators (M % 128 == 0, Tp >= M)
 *   scp_swin_attention_packed_planes: scp_swin_attention_packed(_split) on (q fp32, planes); out fp32 [rows][256] or ohi / olo planes
 * All three give the bits of the fp32 hand-over (scp_swin_ln_linear + scp_swin_attention_packed).  * valid (device fp32 [rows], may be NULL): 1 for real rows, 0 for window padding - a query tile (128 aligned rows) whose first row is
 * padding is skipped and its output rows are not written. */
SCP_API int scp_swin_kv_planes(const float *k, const float *v, int64_t ldkv, int64_t rows, void *khi, void *klo, void *vthi, void *vtlo, void *stream);
SCP_API int scp_swin_ln_qkv(const float *x, int64_t ldx, const float *valid, const void *Whi, const void *Wlo, const float *bias, const float *wbeta,
                            float eps, float *q, int64_t ldq, void *planes, int64_t Tp, int32_t M, int32_t N, void *stream);
SCP_API int scp_swin_attention_packed_planes(const float *q, const void *khi, const void *klo, const void *vthi, const void *vtlo,
                                             const float *bias_table, const int32_t *wtab, int32_t total_windows, int32_t shift, int32_t ldq,
                                             float *out, void *ohi, void *olo, int64_t ldo, const float *valid, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SCP_H */
