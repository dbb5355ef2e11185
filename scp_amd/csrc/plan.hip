// Index maps of the packed ("varlen") EHEM forward, all in ONE launch (gfx950).
//
// The packed forward (scp_amd/models/packed.py) runs every window of a frame through one launch sequence; windows own runs of
// rows padded to a multiple of 512 at every Swin stage, and the stage transitions (patch merging swin_transformer.py:350-367,
// concat_states ehem.py:75-86, the even / odd token split ehem.py:113-114, the scatter into coding order encode.py:126-131)
// are row gathers whose index maps depend only on the list of window lengths.  Built with torch index ops this is ~600 tiny
// kernels per frame - launch-bound, and on a side stream every one of them waits for a slot between the model's persistent
// GEMM workgroups.  Here the host lays out the per-window tables (a few hundred integers) and one kernel writes every entry of
// every map: entry -> (job, row) -> window by binary search on the stage's base rows -> value.
#include <vector>
#include "scp_internal.h"

enum PlanKind : int32_t {
    PK_INMAP = 0,      // [rows(self 0)]  i64: token row of the frame arrays, or n_tokens (pad token)
    PK_A1 = 1,         // [rows(cross 0)] i64: self-0 row of the even token 2t
    PK_A2 = 2,         // [rows(cross 0)] i64: self-0 row of the odd token 2t + 1
    PK_EVEN_ROWS = 3,  // [sum ne] i64: cross-0 row of output position t of the window's even half
    PK_ODD_ROWS = 4,   // [sum no] i64
    PK_EVEN_DST = 5,   // [sum ne] i64: coded[w] + t
    PK_ODD_DST = 6,    // [sum no] i64: coded[w] + ne[w] + t
    PK_MERGE_EVEN = 7, // [rows(next)] i64: row of token 2t in the current stage, or rows(cur) (zero row)
    PK_MERGE_ODD = 8,  // [rows(next)] i64: row of token 2t + 1, or rows(cur)
    PK_CONCAT = 9,     // [rows(stage 0)] i64: stage-s row of token t >> s
    PK_TAB = 10,       // [rows/512][2] i32: (base, padded length) of the window owning the chunk
    PK_TAB_REAL = 11,  // [rows/512][2] i32: (base, real length)
    PK_VALID = 12,     // [rows] f32: 1 for real tokens
    PK_EVEN_OUT = 13,  // [rows(cross 0)] i64: coded position of the window's even token t (cstart + t), -1 beyond ceil(c/2)
    PK_ODD_OUT = 14    // [rows(cross 0)] i64: coded position of the odd token t (cstart + ne + t), -1 beyond floor(c/2)
};

struct PlanJob {
    int32_t kind;
    int32_t lay;       // layout the entries are indexed by (0..4 self stages, 5..8 cross stages, 9 = even outputs, 10 = odd outputs)
    int32_t src;       // second layout (merge: current stage; concat: stage s)
    int32_t shift;     // concat: s
    int64_t n;         // entries (rows, chunks or outputs)
    int64_t first;     // first global entry of the job
    void *out;
};

#define PLAN_MAX_JOBS 64
#define PLAN_LAYOUTS 11

struct PlanArgs {
    PlanJob job[PLAN_MAX_JOBS];
    int32_t njobs, W;
    int64_t n_tokens;
    // per layout: W-long tables in one device buffer (int64): base (start row / start output), L (real length), Lp (padded)
    const int64_t *base[PLAN_LAYOUTS], *len[PLAN_LAYOUTS], *lenp[PLAN_LAYOUTS];
    int64_t rows[PLAN_LAYOUTS];
    const int64_t *c, *cstart, *ne;   // window length, first token row in the frame arrays (= coded offset), even count
};

__device__ __forceinline__ int plan_find(const int64_t *__restrict__ base, int W, int64_t r) {   // last w with base[w] <= r
    int lo = 0, hi = W - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (base[mid] <= r) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void plan_kernel(const PlanArgs a, int64_t total) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    int j = 0;
    while (j + 1 < a.njobs && a.job[j + 1].first <= g) ++j;
    const PlanJob &jb = a.job[j];
    const int64_t e = g - jb.first;
    const int lay = jb.lay;
    if (jb.kind == PK_TAB || jb.kind == PK_TAB_REAL) {
        const int64_t r = e * 512;
        // windows of length 0 share their base with the next one: take the last window whose base <= r that has rows
        int w = plan_find(a.base[lay], a.W, r);
        int32_t *o = (int32_t *)jb.out + 2 * e;
        o[0] = (int32_t)a.base[lay][w];
        o[1] = (int32_t)(jb.kind == PK_TAB ? a.lenp[lay][w] : a.len[lay][w]);
        return;
    }
    const int w = plan_find(a.base[lay], a.W, e);
    const int64_t t = e - a.base[lay][w];
    const bool real = t < a.len[lay][w];
    int64_t v = 0;
    switch (jb.kind) {
    case PK_INMAP: v = t < a.c[w] ? a.cstart[w] + t : a.n_tokens; break;
    case PK_A1: v = real ? a.base[0][w] + 2 * t : 0; break;
    case PK_A2: v = real ? a.base[0][w] + 2 * t + 1 : 0; break;
    case PK_EVEN_ROWS: case PK_ODD_ROWS: v = a.base[5][w] + t; break;
    case PK_EVEN_DST: v = a.cstart[w] + t; break;
    case PK_ODD_DST: v = a.cstart[w] + a.ne[w] + t; break;
    case PK_MERGE_EVEN: v = real ? a.base[jb.src][w] + 2 * t : a.rows[jb.src]; break;
    case PK_MERGE_ODD: v = (real && 2 * t + 1 < a.len[jb.src][w]) ? a.base[jb.src][w] + 2 * t + 1 : a.rows[jb.src]; break;
    case PK_CONCAT: v = real ? a.base[jb.src][w] + (t >> jb.shift) : 0; break;
    case PK_VALID: ((float *)jb.out)[e] = real ? 1.f : 0.f; return;
    case PK_EVEN_OUT: v = t < a.ne[w] ? a.cstart[w] + t : -1; break;
    case PK_ODD_OUT: v = t < a.c[w] - a.ne[w] ? a.cstart[w] + a.ne[w] + t : -1; break;
    default: break;
    }
    ((int64_t *)jb.out)[e] = v;
}

/* lengths[W] (host): window lengths of one packed chunk.  Layout sizes are returned in rows_out[PLAN_LAYOUTS] by
 * scp_packed_plan_sizes so that the caller can allocate; outs[] of scp_packed_plan are device pointers in the fixed job order
 * documented in scp_amd/native.py: packed_plan(). */
static void plan_layouts(const int64_t *c, int W, std::vector<int64_t> (&base)[PLAN_LAYOUTS], std::vector<int64_t> (&len)[PLAN_LAYOUTS],
                         std::vector<int64_t> (&lenp)[PLAN_LAYOUTS], int64_t (&rows)[PLAN_LAYOUTS], std::vector<int64_t> &cstart,
                         std::vector<int64_t> &ne, int64_t &n_tokens) {
    for (int l = 0; l < PLAN_LAYOUTS; ++l) { base[l].assign(W, 0); len[l].assign(W, 0); lenp[l].assign(W, 0); }
    cstart.assign(W, 0); ne.assign(W, 0);
    n_tokens = 0;
    for (int w = 0; w < W; ++w) {
        cstart[w] = n_tokens; n_tokens += c[w];
        ne[w] = (c[w] + 1) / 2;
        const int64_t e = c[w] + (c[w] & 1);
        len[0][w] = e;
        for (int s = 1; s < 5; ++s) len[s][w] = (len[s - 1][w] + 1) / 2;
        len[5][w] = e / 2;
        for (int s = 6; s < 9; ++s) len[s][w] = (len[s - 1][w] + 1) / 2;
        len[9][w] = ne[w];            // even outputs
        len[10][w] = c[w] / 2;        // odd outputs
    }
    for (int l = 0; l < PLAN_LAYOUTS; ++l) {
        int64_t r = 0;
        for (int w = 0; w < W; ++w) {
            lenp[l][w] = l < 9 ? (len[l][w] + 511) / 512 * 512 : len[l][w];
            base[l][w] = r;
            r += lenp[l][w];
        }
        rows[l] = r;
    }
}

extern "C" SCP_API int scp_packed_plan_sizes(const int64_t *lengths, int32_t W, int64_t *rows_out /* [11] */) {
    if (!lengths || !rows_out || W <= 0) return SCP_EINVAL;
    std::vector<int64_t> base[PLAN_LAYOUTS], len[PLAN_LAYOUTS], lenp[PLAN_LAYOUTS], cstart, ne;
    int64_t rows[PLAN_LAYOUTS], nt;
    plan_layouts(lengths, W, base, len, lenp, rows, cstart, ne, nt);
    for (int l = 0; l < PLAN_LAYOUTS; ++l) rows_out[l] = rows[l];
    return SCP_OK;
}

extern "C" SCP_API int scp_packed_plan(const int64_t *lengths, int32_t W, int64_t *tables_dev /* [(3 * 11 + 3) * W] scratch */,
                                       void *const *outs, int32_t n_outs, void *stream) {
    if (!lengths || !tables_dev || !outs || W <= 0) return SCP_EINVAL;
    std::vector<int64_t> base[PLAN_LAYOUTS], len[PLAN_LAYOUTS], lenp[PLAN_LAYOUTS], cstart, ne;
    int64_t rows[PLAN_LAYOUTS], nt;
    plan_layouts(lengths, W, base, len, lenp, rows, cstart, ne, nt);
    // host image of the tables
    std::vector<int64_t> img((size_t)(3 * PLAN_LAYOUTS + 3) * W);
    PlanArgs a;
    a.W = W; a.n_tokens = nt;
    size_t o = 0;
    auto put = [&](const std::vector<int64_t> &v) { const int64_t *p = tables_dev + o; for (int w = 0; w < W; ++w) img[o + w] = v[w]; o += W; return p; };
    for (int l = 0; l < PLAN_LAYOUTS; ++l) { a.base[l] = put(base[l]); a.len[l] = put(len[l]); a.lenp[l] = put(lenp[l]); a.rows[l] = rows[l]; }
    std::vector<int64_t> cv(lengths, lengths + W);
    a.c = put(cv); a.cstart = put(cstart); a.ne = put(ne);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(tables_dev, img.data(), img.size() * 8, hipMemcpyHostToDevice, st));
    // jobs in the fixed order of native.packed_plan()
    int nj = 0;
    int64_t first = 0;
    auto add = [&](int kind, int lay, int src, int shift, int64_t n) {
        if (nj >= PLAN_MAX_JOBS || nj >= n_outs) return;
        PlanJob &j = a.job[nj];
        j.kind = kind; j.lay = lay; j.src = src; j.shift = shift; j.n = n; j.first = first; j.out = outs[nj];
        first += n; ++nj;
    };
    add(PK_INMAP, 0, 0, 0, rows[0]);
    add(PK_A1, 5, 0, 0, rows[5]);
    add(PK_A2, 5, 0, 0, rows[5]);
    add(PK_EVEN_ROWS, 9, 0, 0, rows[9]);
    add(PK_ODD_ROWS, 10, 0, 0, rows[10]);
    add(PK_EVEN_DST, 9, 0, 0, rows[9]);
    add(PK_ODD_DST, 10, 0, 0, rows[10]);
    for (int s = 0; s < 4; ++s) { add(PK_MERGE_EVEN, s + 1, s, 0, rows[s + 1]); add(PK_MERGE_ODD, s + 1, s, 0, rows[s + 1]); }
    for (int s = 5; s < 8; ++s) { add(PK_MERGE_EVEN, s + 1, s, 0, rows[s + 1]); add(PK_MERGE_ODD, s + 1, s, 0, rows[s + 1]); }
    for (int s = 1; s < 5; ++s) add(PK_CONCAT, 0, s, s, rows[0]);
    for (int s = 1; s < 4; ++s) add(PK_CONCAT, 5, 5 + s, s, rows[5]);
    for (int l = 0; l < 9; ++l) add(PK_TAB, l, 0, 0, rows[l] / 512);
    add(PK_TAB_REAL, 0, 0, 0, rows[0] / 512);
    for (int l = 0; l < 9; ++l) add(PK_VALID, l, 0, 0, rows[l]);
    for (int s = 0; s < 4; ++s) add(PK_CONCAT, s, s + 1, 1, rows[s]);        // parent rows: stage s token t -> stage s + 1 token t >> 1
    for (int s = 5; s < 8; ++s) add(PK_CONCAT, s, s + 1, 1, rows[s]);
    add(PK_EVEN_OUT, 5, 0, 0, rows[5]);
    add(PK_ODD_OUT, 5, 0, 0, rows[5]);
    if (nj != n_outs) return SCP_EINVAL;
    a.njobs = nj;
    // the host image must outlive the async copy: pageable-memory hipMemcpyAsync returns after staging, but be explicit
    HIP_TRY(hipStreamSynchronize(st));
    if (first > 0) hipLaunchKernelGGL(plan_kernel, dim3((unsigned)cdiv64(first, 256)), dim3(256), 0, st, a, first);
    LAUNCH_CHECK();
    return SCP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Decoder: the children of one decoded octree level (decode_ehem_mullevel.py:100-130 / decode_ehem.py: the breadth-first
// regeneration - children in (parent order, child digit) order, the ancestor window shifted by one, cal_pos_ary) AND the
// model inputs of the level they form (context rows (level, octant, occupancy) x (ggp, gp, p, self) as uint8, positions
// normalised with the level's (min, max) pair in double precision), in ONE launch.  Until round 5 this was ~25 torch index
// launches per level behind the last window's symbols, with the GPU idle.
//   sym[i]   decoded symbol of parent i (occupancy - 1; -1 = the multi-level shell's dropped last node: no children)
//   cum[i]   inclusive scan of popcount(sym + 1)
//   anc      uint8 [n][9]: (level, octant, symbol) of (ggp, gp, p) of every parent, 255 = unknown symbol / pad
//   child c of parent i: pos = pos_i + digit bits << shift; anc = (anc_i[3:9], (L, octant_i, sym_i)); octant = digit + 1
//   ctx row = (anc with levels clamped to lv_clamp, (lv_next, octant, 255)); posn = polar ? (pos - mn) / (mx - mn + eps) : pos / div
struct ExpandArgs {
    const int64_t *sym, *cum;
    const int32_t *pos;
    const uint8_t *anc, *octant;
    int64_t n;
    int32_t L, shift, lv_next, lv_clamp, polar;
    double mn, den;
    int32_t *cpos;
    uint8_t *canc, *coct, *cctx, *occ8;
    float *cposn;
};

__global__ __launch_bounds__(256) void decode_expand_kernel(const ExpandArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const int64_t s = a.sym[i];
    const unsigned occ = (unsigned)((s + 1) & 0xff);
    a.occ8[i] = (uint8_t)occ;
    if (!occ) return;
    int64_t c = a.cum[i] - __popc(occ);
    const int px = a.pos[3 * i], py = a.pos[3 * i + 1], pz = a.pos[3 * i + 2];
    uint8_t an[9];
#pragma unroll
    for (int j = 0; j < 6; ++j) an[j] = a.anc[9 * i + 3 + j];
    an[6] = (uint8_t)a.L; an[7] = a.octant[i]; an[8] = (uint8_t)s;
    for (int d = 0; d < 8; ++d) {
        if (!((occ >> d) & 1)) continue;
        const int x = px + (((d >> 2) & 1) << a.shift), y = py + (((d >> 1) & 1) << a.shift), z = pz + ((d & 1) << a.shift);
        a.cpos[3 * c] = x; a.cpos[3 * c + 1] = y; a.cpos[3 * c + 2] = z;
        a.coct[c] = (uint8_t)(d + 1);
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            a.canc[9 * c + j] = an[j];
            a.cctx[12 * c + j] = (j % 3 == 0 && an[j] > a.lv_clamp) ? (uint8_t)a.lv_clamp : an[j];
        }
        a.cctx[12 * c + 9] = (uint8_t)a.lv_next; a.cctx[12 * c + 10] = (uint8_t)(d + 1); a.cctx[12 * c + 11] = 255;
        if (a.polar) {
            a.cposn[3 * c] = (float)(((double)x - a.mn) / a.den); a.cposn[3 * c + 1] = (float)(((double)y - a.mn) / a.den);
            a.cposn[3 * c + 2] = (float)(((double)z - a.mn) / a.den);
        } else {
            a.cposn[3 * c] = (float)((double)x / a.den); a.cposn[3 * c + 1] = (float)((double)y / a.den); a.cposn[3 * c + 2] = (float)((double)z / a.den);
        }
        ++c;
    }
}

extern "C" SCP_API int scp_decode_expand(const int64_t *sym, const int64_t *cum, const int32_t *pos, const uint8_t *anc, const uint8_t *octant, int64_t n,
                                         int32_t L, int32_t shift, int32_t lv_next, int32_t lv_clamp, int32_t polar, double mn, double den, int32_t *cpos,
                                         uint8_t *canc, uint8_t *coct, uint8_t *cctx, float *cposn, uint8_t *occ8, void *stream) {
    if (!sym || !cum || !pos || !anc || !octant || !cpos || !canc || !coct || !cctx || !cposn || !occ8 || n <= 0 || L < 1 || L > 254 || shift < 0 ||
        shift > 30 || lv_next < 0 || lv_next > 255 || lv_clamp < 0 || lv_clamp > 255 || !(den == den) || den == 0.0)
        return SCP_EINVAL;
    ExpandArgs a;
    a.sym = sym; a.cum = cum; a.pos = pos; a.anc = anc; a.octant = octant; a.n = n; a.L = L; a.shift = shift; a.lv_next = lv_next;
    a.lv_clamp = lv_clamp; a.polar = polar; a.mn = mn; a.den = den; a.cpos = cpos; a.canc = canc; a.coct = coct; a.cctx = cctx; a.cposn = cposn; a.occ8 = occ8;
    hipLaunchKernelGGL(decode_expand_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, a);
    LAUNCH_CHECK();
    return SCP_OK;
}
