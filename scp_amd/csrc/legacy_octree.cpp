// The ten C symbols data_preproc/OctreeCPP/Octreewarpper.py:17-39 binds, served by the HIP octree builder, so the
// reference's own ctypes wrapper can load libscp_hip.so in place of Octree_python_lib.so.
// Ownership mirrors the wrapper's expectations: the tree vector is freed by delete_vector; the code vector
// returned by genOctreeInterface is owned by the tree vector here (the reference leaks it); Nodes_get returns a
// pointer into the level array (the reference leaks a 28-byte copy per call).
#include <hip/hip_runtime.h>
#include <math.h>
#include <new>
#include <vector>
#include "../../include/scp.h"

namespace {
struct LegacyTree {
    std::vector<std::vector<scp_legacy_node>> levels;
    std::vector<int> codes;
};
}  // namespace

extern "C" void *new_vector(void) { return new (std::nothrow) LegacyTree(); }
extern "C" void delete_vector(void *v) { delete (LegacyTree *)v; }
extern "C" int vector_size(void *v) { return v ? (int)((LegacyTree *)v)->levels.size() : 0; }
extern "C" void *vector_get(void *v, int level) {
    LegacyTree *t = (LegacyTree *)v;
    if (!t || level < 0 || level >= (int)t->levels.size()) return nullptr;
    return &t->levels[level];
}
extern "C" void vector_push_back(void *v, int) {
    if (v) ((LegacyTree *)v)->levels.emplace_back();
}
extern "C" int Nodes_size(void *level) { return level ? (int)((std::vector<scp_legacy_node> *)level)->size() : 0; }
extern "C" scp_legacy_node *Nodes_get(void *level, int i) {
    auto *l = (std::vector<scp_legacy_node> *)level;
    if (!l || i < 0 || i >= (int)l->size()) return nullptr;
    return &(*l)[i];
}
extern "C" int int_size(void *codes) { return codes ? (int)((std::vector<int> *)codes)->size() : 0; }
extern "C" int int_get(void *codes, int i) {
    auto *c = (std::vector<int> *)codes;
    return (c && i >= 0 && i < (int)c->size()) ? (*c)[i] : -1;
}

// xyz: C-contiguous float64 [n][3] holding integer values (Octreewarpper.py:69-70).  Returns the code vector, or
// NULL on any error (the reference aborts the process instead).
extern "C" void *genOctreeInterface(void *v, const double *xyz, int n) {
    LegacyTree *t = (LegacyTree *)v;
    if (!t || !xyz || n <= 0) return nullptr;
    std::vector<int32_t> q((size_t)n * 3);
    for (size_t i = 0; i < q.size(); ++i) q[i] = (int32_t)llround(xyz[i]);
    int32_t *dq = nullptr;
    scp_geom *g = nullptr;
    void *result = nullptr;
    uint8_t *docc = nullptr;
    if (hipMalloc((void **)&dq, q.size() * 4) != hipSuccess) return nullptr;
    do {
        if (hipMemcpy(dq, q.data(), q.size() * 4, hipMemcpyHostToDevice) != hipSuccess) break;
        if (scp_geom_create(&g) != SCP_OK) break;
        scp_segment seg = {0, n, 0, 0, 0, 0};
        scp_segment_info info;
        if (scp_geom_build(g, dq, n, &seg, 1, &info, nullptr) != SCP_OK) break;
        const size_t N = (size_t)info.n_nodes;
        if (hipMalloc((void **)&docc, 2 * N) != hipSuccess) break;      // occupancy bytes, then octants
        uint8_t *doct = docc + N;
        int32_t *dparent = nullptr, *dpos = nullptr;                    // the int tables: separate (aligned) allocations
        if (hipMalloc((void **)&dparent, N * 4) != hipSuccess) break;
        if (hipMalloc((void **)&dpos, N * 12) != hipSuccess) { (void)hipFree(dparent); break; }
        int rc = scp_geom_emit_nodes(g, docc, nullptr, doct, dparent, dpos, nullptr);
        std::vector<uint8_t> occ(N), oct(N);
        std::vector<int32_t> parent(N), pos(3 * N);
        if (rc == SCP_OK && hipMemcpy(occ.data(), docc, N, hipMemcpyDeviceToHost) == hipSuccess &&
            hipMemcpy(oct.data(), doct, N, hipMemcpyDeviceToHost) == hipSuccess &&
            hipMemcpy(parent.data(), dparent, N * 4, hipMemcpyDeviceToHost) == hipSuccess &&
            hipMemcpy(pos.data(), dpos, N * 12, hipMemcpyDeviceToHost) == hipSuccess) {
            t->levels.assign(info.depth, {});
            t->codes.resize(N);
            size_t nd = 0;
            for (int L = 0; L < info.depth; ++L) {
                auto &lv = t->levels[L];
                lv.resize((size_t)info.level_count[L]);
                for (auto &e : lv) {
                    e.nodeid = (uint32_t)(nd + 1);
                    e.octant = oct[nd];
                    e.parent = parent[nd] < 0 ? 1u : (uint32_t)(parent[nd] + 1);  // the binary reports 1 for the root
                    e.oct = occ[nd];
                    e.pos[0] = (uint32_t)pos[3 * nd]; e.pos[1] = (uint32_t)pos[3 * nd + 1]; e.pos[2] = (uint32_t)pos[3 * nd + 2];
                    t->codes[nd] = occ[nd];
                    ++nd;
                }
            }
            result = &t->codes;
        }
        (void)hipFree(dparent);
        (void)hipFree(dpos);
    } while (0);
    if (docc) (void)hipFree(docc);
    if (g) scp_geom_destroy(g);
    (void)hipFree(dq);
    return result;
}
