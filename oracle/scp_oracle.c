/*
 * scp_oracle.c - CPU restatement of the integer/byte parts of the SCP encode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / the reported CPU baseline.  The product path (scp_amd/) never calls it.
 *
 * Parity status: PINNED.  Every function here is checked in tests/test_oracle.py against
 * golden vectors produced by running the reference itself (tests/golden/make_golden.py):
 *   octree build      <- data_preproc/OctreeCPP/Octree_python_lib.so (genOctreeInterface) and its
 *                        source-form twin data_preproc/Octree.py:148-181 (GenOctree)
 *   shell filter      <- data_preproc/Octree.py:184-221 (mullevel_gen_octree)
 *   K=4 records       <- data_preproc/Octree.py:102-137 / :224-272 + data_preprocess.py:74
 *   de-octree         <- data_preproc/Octree.py:68-99 (DeOctree)
 *   PMF -> int CDF    <- numpyAc/numpyAc.py:109-114 and :80-107
 *   range coder       <- numpyAc/backend/numpyAc_backend.cpp:245-323 (encode), :134-217 (decode)
 *
 * The octree builder deliberately follows the reference's own algorithm shape (top-down,
 * breadth-first, per-node partition of the point list by child digit) and NOT the sorted-prefix
 * formulation the HIP kernels use, so that the two are independent derivations.
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off -o liboracle.so scp_oracle.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_EINVAL (-1)
#define ORC_ENOMEM (-2)
#define ORC_ESMALL (-3)

typedef struct {
    int64_t n_nodes;
    int32_t depth;
    uint8_t *level;   /* 1..depth                                     */
    uint8_t *octant;  /* 1..8 (root: 1)                               */
    uint8_t *occ;     /* 1..255 = sum over children d of 1<<d          */
    int32_t *parent;  /* 1-based BFS id of the parent, root: 0        */
    int32_t *pos;     /* [n_nodes][3] node origin in leaf units       */
    int64_t *level_off; /* [depth+1] first node of each level         */
} orc_tree;

void orc_tree_free(orc_tree *t) {
    if (!t) return;
    free(t->level); free(t->octant); free(t->occ); free(t->parent); free(t->pos); free(t->level_off);
    free(t);
}

/* Octree.py:58  n = ceil(log2(max(A)+1)) over all three axes */
static int depth_of(const int64_t *pts, int64_t n) {
    int64_t mx = 0;
    for (int64_t i = 0; i < 3 * n; ++i) if (pts[i] > mx) mx = pts[i];
    int d = 0;
    while (((int64_t)1 << d) < mx + 1) ++d;
    return d;
}

/* child digit of point p at tree level L (1-based), Octree.py:56-65 + :156-158:
 * bits are taken MSB first; digit = 4*x_bit + 2*y_bit + z_bit                      */
static inline int digit_at(const int64_t *p, int depth, int L) {
    int sh = depth - L;
    return (int)((((p[0] >> sh) & 1) << 2) | (((p[1] >> sh) & 1) << 1) | ((p[2] >> sh) & 1));
}

/*
 * Build the octree of integer points (duplicates are removed first, as proc_pc does with
 * np.unique at data_preprocess.py:69).  If path_len > 0 only points whose top path_len x-axis
 * bits equal path[] are kept (Octree.py:188); the depth always comes from the unfiltered cloud.
 * Returns NULL on error (*err set).  A cloud whose maximum coordinate is 0 has depth 0: the
 * reference aborts on it (std::out_of_range) -> ORC_EINVAL here.
 */
orc_tree *orc_octree_build(const int64_t *pts_in, int64_t n, const int32_t *path, int32_t path_len, int32_t *err) {
    *err = ORC_OK;
    if (n <= 0) { *err = ORC_EINVAL; return NULL; }
    for (int64_t i = 0; i < 3 * n; ++i) if (pts_in[i] < 0) { *err = ORC_EINVAL; return NULL; }
    int depth = depth_of(pts_in, n);
    if (depth == 0 || depth > 21) { *err = ORC_EINVAL; return NULL; }

    /* shell filter, then order-preserving removal of duplicates by brute sort of indices */
    int64_t *pts = (int64_t *)malloc(sizeof(int64_t) * 3 * n);
    if (!pts) { *err = ORC_ENOMEM; return NULL; }
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t *p = pts_in + 3 * i;
        int keep = 1;
        for (int k = 0; k < path_len; ++k)
            if (((p[0] >> (depth - 1 - k)) & 1) != path[k]) { keep = 0; break; }
        if (keep) { memcpy(pts + 3 * m, p, 3 * sizeof(int64_t)); ++m; }
    }
    if (m == 0) { free(pts); *err = ORC_EINVAL; return NULL; }

    /* idx = the reference's per-node point-id lists, kept as ranges of one array */
    int64_t *idx = (int64_t *)malloc(sizeof(int64_t) * m), *tmp = (int64_t *)malloc(sizeof(int64_t) * m);
    for (int64_t i = 0; i < m; ++i) idx[i] = i;

    int64_t cap = m * (depth < 4 ? depth : 4) + 64, N = 0;
    orc_tree *t = (orc_tree *)calloc(1, sizeof(orc_tree));
    t->depth = depth;
    t->level_off = (int64_t *)calloc(depth + 1, sizeof(int64_t));
#define GROW() do { if (N >= cap) { cap *= 2; \
        t->level = realloc(t->level, cap); t->octant = realloc(t->octant, cap); t->occ = realloc(t->occ, cap); \
        t->parent = realloc(t->parent, cap * 4); t->pos = realloc(t->pos, cap * 12); \
        rs = realloc(rs, cap * 8); re = realloc(re, cap * 8); } } while (0)
    t->level = malloc(cap); t->octant = malloc(cap); t->occ = malloc(cap);
    t->parent = malloc(cap * 4); t->pos = malloc(cap * 12);
    int64_t *rs = malloc(cap * 8), *re = malloc(cap * 8); /* point range of every node */

    /* level 1: the single root node holds every point (Octree.py:161-163) */
    t->level[0] = 1; t->octant[0] = 1; t->parent[0] = 0; t->pos[0] = t->pos[1] = t->pos[2] = 0;
    rs[0] = 0; re[0] = m; N = 1; t->level_off[0] = 0;

    int64_t lvl_begin = 0;
    for (int L = 1; L <= depth; ++L) {
        int64_t lvl_end = N;
        if (L < depth + 1) t->level_off[L] = lvl_end; /* start of level L+1 */
        for (int64_t nd = lvl_begin; nd < lvl_end; ++nd) {
            /* stable partition of this node's points by digit at level L (Octree.py:170-173) */
            int64_t cnt[8] = {0}, off[8];
            for (int64_t i = rs[nd]; i < re[nd]; ++i) cnt[digit_at(pts + 3 * idx[i], depth, L)]++;
            int64_t a = rs[nd];
            int occ = 0;
            for (int d = 0; d < 8; ++d) { off[d] = a; a += cnt[d]; if (cnt[d]) occ |= 1 << d; }
            t->occ[nd] = (uint8_t)occ; /* Octree.py:174-176 */
            if (L == depth) {
                /* leaves: several identical points in one child are duplicates -> collapse */
                continue;
            }
            int64_t o2[8];
            memcpy(o2, off, sizeof(off));
            for (int64_t i = rs[nd]; i < re[nd]; ++i) { int d = digit_at(pts + 3 * idx[i], depth, L); tmp[o2[d]++] = idx[i]; }
            memcpy(idx + rs[nd], tmp + rs[nd], (re[nd] - rs[nd]) * sizeof(int64_t));
            for (int d = 0; d < 8; ++d) {
                if (!cnt[d]) continue;
                GROW();
                t->level[N] = (uint8_t)(L + 1);
                t->octant[N] = (uint8_t)(d + 1);
                t->parent[N] = (int32_t)(nd + 1);
                int sh = depth - L; /* Octree.py:140-145 get_pos: digit i contributes bit << (depth-1-i) */
                t->pos[3 * N + 0] = t->pos[3 * nd + 0] + (((d >> 2) & 1) << sh);
                t->pos[3 * N + 1] = t->pos[3 * nd + 1] + (((d >> 1) & 1) << sh);
                t->pos[3 * N + 2] = t->pos[3 * nd + 2] + ((d & 1) << sh);
                rs[N] = off[d]; re[N] = off[d] + cnt[d];
                ++N;
            }
        }
        lvl_begin = lvl_end;
    }
    t->n_nodes = N;
    free(rs); free(re); free(idx); free(tmp); free(pts);
    return t;
}

/* plain accessors so that ctypes does not need the struct layout */
int64_t orc_tree_nodes(const orc_tree *t) { return t->n_nodes; }
int32_t orc_tree_depth(const orc_tree *t) { return t->depth; }
void orc_tree_export(const orc_tree *t, uint8_t *level, uint8_t *octant, uint8_t *occ, int32_t *parent, int32_t *pos,
                     int64_t *level_off) {
    memcpy(level, t->level, t->n_nodes); memcpy(octant, t->octant, t->n_nodes); memcpy(occ, t->occ, t->n_nodes);
    memcpy(parent, t->parent, t->n_nodes * 4); memcpy(pos, t->pos, t->n_nodes * 12);
    memcpy(level_off, t->level_off, (t->depth + 1) * 8);
    level_off[t->depth] = t->n_nodes;
}

/*
 * K=4 ancestor records, int64 [n_out][4][6] = (occ, level, octant, x, y, z) for
 * (great-grandparent, grandparent, parent, self); missing ancestors: occ 256, rest 0.
 * Octree.py:102-137; channel order from data_preprocess.py:74.  drop_last=1 reproduces the
 * mullevel variant that returns rows 1:n (Octree.py:259-262).  Returns the row count.
 */
int64_t orc_krecords(const orc_tree *t, int32_t drop_last, int64_t *out) {
    int64_t N = t->n_nodes;
    for (int64_t n = 0; n < N; ++n) {
        int64_t *r = out + n * 24;
        if (n == 0) {
            for (int k = 0; k < 3; ++k) { r[k * 6] = 256; for (int c = 1; c < 6; ++c) r[k * 6 + c] = 0; }
        } else {
            const int64_t *p = out + (int64_t)(t->parent[n] - 1) * 24;
            memcpy(r, p + 6, 18 * sizeof(int64_t)); /* rows 0..2 <- parent's rows 1..3 */
        }
        r[18] = t->occ[n]; r[19] = t->level[n]; r[20] = t->octant[n];
        r[21] = t->pos[3 * n]; r[22] = t->pos[3 * n + 1]; r[23] = t->pos[3 * n + 2];
    }
    return drop_last ? N - 1 : N;
}

/* Octree.py:68-99 DeOctree: occupancy codes (BFS) -> leaf coordinates, in BFS/Morton order.
 * Returns number of points written (<= cap) or a negative error. */
int64_t orc_deoctree(const uint8_t *codes, int64_t n_codes, int32_t *pts_out, int64_t cap) {
    /* first pass: depth from the level sizes */
    int64_t counts[64]; int Lmax = 0; int64_t cal = 0, cur = 1;
    while (cal + cur <= n_codes && Lmax < 62) {
        int64_t nxt = 0;
        for (int64_t i = cal; i < cal + cur; ++i) nxt += __builtin_popcount(codes[i]);
        counts[Lmax++] = cur; cal += cur; cur = nxt;
        if (cur == 0) break;
    }
    int64_t maxn = 1;
    for (int i = 0; i < Lmax; ++i) if (counts[i] > maxn) maxn = counts[i];
    int64_t last = cur > maxn ? cur : maxn;
    int32_t *a = calloc(3 * last, 4), *b = calloc(3 * last, 4);
    if (!a || !b) { free(a); free(b); return ORC_ENOMEM; }
    int64_t na = 1, ci = 0;
    for (int L = 1; L <= Lmax; ++L) {
        int64_t nb = 0;
        for (int64_t i = 0; i < na; ++i) {
            int code = codes[ci++];
            for (int d = 0; d < 8; ++d) if (code & (1 << d)) {
                b[3 * nb] = a[3 * i] + (((d >> 2) & 1) << (Lmax - L));
                b[3 * nb + 1] = a[3 * i + 1] + (((d >> 1) & 1) << (Lmax - L));
                b[3 * nb + 2] = a[3 * i + 2] + ((d & 1) << (Lmax - L));
                ++nb;
            }
        }
        int32_t *s = a; a = b; b = s; na = nb;
    }
    if (na > cap) { free(a); free(b); return ORC_ESMALL; }
    memcpy(pts_out, a, na * 12);
    free(a); free(b);
    return na;
}

/* ------------------------------------------------------------------ PMF -> integer CDF
 * numpyAc.py:109-114: c = cumsum_f32(p) (serial), c /= c[-1] (f32), F = [0, c...] as f64;
 * numpyAc.py:99-106: q = rint(F * (65536 - (Lp-1))), int16 wrap, + arange(Lp).           */
void orc_pmf_to_cdf(const float *pmf, int64_t n, int32_t nsym, uint16_t *cdf) {
    int Lp = nsym + 1;
    double scale = (double)(65536 - (Lp - 1));
    float *c = (float *)malloc(sizeof(float) * nsym);
    for (int64_t r = 0; r < n; ++r) {
        const float *p = pmf + r * nsym;
        volatile float acc = 0.0f;
        for (int j = 0; j < nsym; ++j) { acc = acc + p[j]; c[j] = acc; }
        float last = c[nsym - 1];
        uint16_t *o = cdf + r * Lp;
        o[0] = 0;
        for (int j = 0; j < nsym; ++j) {
            float f = c[j] / last;
            double q = nearbyint((double)f * scale); /* default rounding mode = half-to-even, as np.round */
            o[j + 1] = (uint16_t)(((int64_t)q + (j + 1)) & 0xFFFF);
        }
    }
    free(c);
}

/* ------------------------------------------------------------------ range coder
 * numpyAc_backend.cpp:245-323.  cdf: uint16 [n][Lp]; sym: int16 [n].  Returns byte count or
 * ORC_ESMALL if cap is too small. */
typedef struct { uint8_t *out; int64_t cap, len; uint8_t cache, count; int overflow; } bitsink;
static inline void put_bit(bitsink *s, int bit) {
    s->cache = (uint8_t)((s->cache << 1) | bit);
    if (++s->count == 8) {
        if (s->len < s->cap) s->out[s->len] = s->cache; else s->overflow = 1;
        s->len++; s->count = 0;
    }
}
static inline void put_bit_pending(bitsink *s, int bit, uint64_t *pending) {
    put_bit(s, bit);
    while (*pending) { put_bit(s, !bit); --*pending; }
}

int64_t orc_ac_encode(const uint16_t *cdf, const int16_t *sym, int64_t n, int32_t Lp, uint8_t *out, int64_t cap) {
    bitsink s = {out, cap, 0, 0, 0, 0};
    uint32_t low = 0, high = 0xFFFFFFFFu;
    uint64_t pending = 0;
    const int max_symbol = Lp - 2;
    for (int64_t i = 0; i < n; ++i) {
        int sy = sym[i];
        uint64_t span = (uint64_t)high - (uint64_t)low + 1;
        uint32_t c_low = cdf[i * Lp + sy];
        uint32_t c_high = sy == max_symbol ? 0x10000u : cdf[i * Lp + sy + 1];
        high = (low - 1) + (uint32_t)((span * (uint64_t)c_high) >> 16);
        low = low + (uint32_t)((span * (uint64_t)c_low) >> 16);
        for (;;) {
            if (high < 0x80000000u) { put_bit_pending(&s, 0, &pending); low <<= 1; high = (high << 1) | 1; }
            else if (low >= 0x80000000u) { put_bit_pending(&s, 1, &pending); low <<= 1; high = (high << 1) | 1; }
            else if (low >= 0x40000000u && high < 0xC0000000u) {
                pending++; low = (low << 1) & 0x7FFFFFFFu; high = (high << 1) | 0x80000001u;
            } else break;
        }
    }
    pending += 1;
    put_bit_pending(&s, low < 0x40000000u ? 0 : 1, &pending);
    while (s.count) put_bit(&s, 0);
    return s.overflow ? ORC_ESMALL : s.len;
}

/* decoder, numpyAc_backend.cpp:70-217 */
typedef struct {
    const uint8_t *in; int64_t len, ptr; uint8_t cache, cached_bits;
    uint32_t low, high, value; int32_t Lp;
} orc_dec;

static inline void get_bit(orc_dec *d) {
    if (d->cached_bits == 0) {
        if (d->ptr == d->len) { d->value <<= 1; return; }
        d->cache = d->in[d->ptr++]; d->cached_bits = 8;
    }
    d->value = (d->value << 1) | ((d->cache >> (d->cached_bits - 1)) & 1);
    d->cached_bits--;
}

orc_dec *orc_ac_dec_new(const uint8_t *in, int64_t len, int32_t Lp) {
    orc_dec *d = (orc_dec *)calloc(1, sizeof(orc_dec));
    uint8_t *copy = (uint8_t *)malloc(len > 0 ? len : 1);
    memcpy(copy, in, len);
    d->in = copy; d->len = len; d->high = 0xFFFFFFFFu; d->Lp = Lp;
    for (int i = 0; i < 32; ++i) get_bit(d);
    return d;
}
void orc_ac_dec_free(orc_dec *d) { if (d) { free((void *)d->in); free(d); } }

int32_t orc_ac_dec_next(orc_dec *d, const uint16_t *cdf_row) {
    const int max_symbol = d->Lp - 2;
    uint64_t span = (uint64_t)d->high - (uint64_t)d->low + 1;
    uint16_t count = (uint16_t)((((uint64_t)d->value - (uint64_t)d->low + 1) * 0x10000u - 1) / span);
    /* binsearch, numpyAc_backend.cpp:110-131 */
    uint16_t left = 0, right = (uint16_t)(max_symbol + 1);
    int found = -1;
    while (left + 1 < right) {
        uint16_t mid = (uint16_t)((left + right) / 2);
        uint16_t v = cdf_row[mid];
        if (v < count) left = mid; else if (v > count) right = mid; else { found = mid; break; }
    }
    int sy = found >= 0 ? found : left;
    uint32_t c_low = cdf_row[sy];
    uint32_t c_high = sy == max_symbol ? 0x10000u : cdf_row[sy + 1];
    d->high = (d->low - 1) + (uint32_t)((span * (uint64_t)c_high) >> 16);
    d->low = d->low + (uint32_t)((span * (uint64_t)c_low) >> 16);
    for (;;) {
        if (d->low >= 0x80000000u || d->high < 0x80000000u) {
            d->low <<= 1; d->high = (d->high << 1) | 1; get_bit(d);
        } else if (d->low >= 0x40000000u && d->high < 0xC0000000u) {
            d->low = (d->low << 1) & 0x7FFFFFFFu; d->high = (d->high << 1) | 0x80000001u;
            d->value -= 0x40000000u; get_bit(d);
        } else break;
    }
    return sy;
}
