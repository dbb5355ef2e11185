// Library-level entry points of libscp_hip.so.
#include <string.h>
#include <atomic>
#include <mutex>
#include <new>
#include <vector>
#include "scp_internal.h"

int g_scp_last_hip_error = 0;

extern "C" int scp_version(void) { return SCP_ABI_VERSION; }
extern "C" int scp_last_hip_error(void) { return g_scp_last_hip_error; }

extern "C" int scp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int scp_device_name(char *buf, int cap) {
    if (!buf || cap <= 0) return SCP_EINVAL;
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, 0));
    strncpy(buf, p.name, cap - 1);
    buf[cap - 1] = 0;
    return SCP_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Numeric profile of one encoder / decoder: which arithmetic the kernels with a choice use.  The choice decides the last bits of the
// logits, hence the integer CDFs a decoder must reproduce, so it belongs to a handle, not to the process: every host thread has a
// current profile (scp_ctx_make_current; NULL = the process default, which SCP_KNN / SCP_ATTN select at load).
struct scp_ctx { int knn_f16x3; int attn_bf16x3; };
static thread_local const scp_ctx *t_ctx = nullptr;

extern "C" int scp_ctx_create(scp_ctx **out) {
    if (!out) return SCP_EINVAL;
    scp_ctx *c = new (std::nothrow) scp_ctx;
    if (!c) return SCP_ENOMEM;
    c->knn_f16x3 = 1; c->attn_bf16x3 = 1;
    *out = c;
    return SCP_OK;
}
extern "C" int scp_ctx_destroy(scp_ctx *c) {
    if (!c) return SCP_EINVAL;
    if (t_ctx == c) t_ctx = nullptr;
    delete c;
    return SCP_OK;
}
extern "C" int scp_ctx_set(scp_ctx *c, int32_t key, int32_t value) {
    if (!c || (value != 0 && value != 1)) return SCP_EINVAL;
    if (key == SCP_CTX_KNN_F16X3) c->knn_f16x3 = value;
    else if (key == SCP_CTX_ATTENTION_BF16X3) c->attn_bf16x3 = value;
    else return SCP_EINVAL;
    return SCP_OK;
}
extern "C" int scp_ctx_get(const scp_ctx *c, int32_t key) {
    if (!c) return SCP_EINVAL;
    if (key == SCP_CTX_KNN_F16X3) return c->knn_f16x3;
    if (key == SCP_CTX_ATTENTION_BF16X3) return c->attn_bf16x3;
    return SCP_EINVAL;
}
extern "C" int scp_ctx_make_current(const scp_ctx *c) { t_ctx = c; return SCP_OK; }
int scp_ctx_knn_mode() { return t_ctx ? t_ctx->knn_f16x3 : -1; }
int scp_ctx_attention_mode() { return t_ctx ? t_ctx->attn_bf16x3 : -1; }

// ------------------------------------------------------------------------------------------------------------------------------
// Launch brackets (scp_debug.h): two hipEvents per bracketed launch, recorded inside the C ABI around the hipLaunchKernelGGL itself.
std::atomic<int> g_scp_prof_on{0};
namespace {
unsigned g_prof_gen = 0;             // bumped whenever scp_prof_enable(1) clears the records: a scope that began before the clear ends nowhere
struct ProfRec { int tag; double work; hipEvent_t e0, e1; bool ended; };
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_prof_pool;
hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

void ScpProfScope::begin(int tag, double work) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_scp_prof_on.load(std::memory_order_relaxed) || g_prof.size() >= 65536) return;
    ProfRec r{tag, work, prof_event(), prof_event(), false};
    if (!r.e0 || !r.e1 || hipEventRecord(r.e0, st) != hipSuccess) {
        if (r.e0) g_prof_pool.push_back(r.e0);
        if (r.e1) g_prof_pool.push_back(r.e1);
        return;
    }
    slot = (int)g_prof.size();
    gen = g_prof_gen;
    g_prof.push_back(r);
}
void ScpProfScope::end() {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (gen != g_prof_gen) return;      // the records this scope belongs to were cleared meanwhile (another thread's scp_prof_enable(1))
    if (slot < (int)g_prof.size() && !g_prof[slot].ended && hipEventRecord(g_prof[slot].e1, st) == hipSuccess) g_prof[slot].ended = true;
}

extern "C" int scp_prof_enable(int32_t on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (on) {
        for (auto &r : g_prof) { g_prof_pool.push_back(r.e0); g_prof_pool.push_back(r.e1); }
        g_prof.clear();
        ++g_prof_gen;
    }
    g_scp_prof_on.store(on ? 1 : 0, std::memory_order_relaxed);
    return SCP_OK;
}
extern "C" int scp_prof_count(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    return (int)g_prof.size();
}
extern "C" int scp_prof_read(int32_t cap, int32_t *tags, float *ms, double *work) {
    if (cap < 0 || (cap > 0 && (!tags || !ms || !work))) return SCP_EINVAL;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = 0;
    for (auto &r : g_prof) {
        if (n >= cap) break;
        float t = -1.f;
        if (r.ended) {
            HIP_TRY(hipEventSynchronize(r.e1));
            HIP_TRY(hipEventElapsedTime(&t, r.e0, r.e1));
        }
        tags[n] = r.tag; ms[n] = t; work[n] = r.work;
        ++n;
    }
    return n;
}

// ---- scp_stream_wait, scp_d2h_async (scp_internal.h) -----------------------------------------------------------------------------------------
#include <time.h>
#include <string.h>
#include <vector>
namespace {
struct PinStage {
    char *p = nullptr;
    size_t cap = 0, used = 0;
    struct Item { void *dst; size_t off, bytes; };
    std::vector<Item> items;
};
thread_local PinStage g_pin;
constexpr size_t PIN_BYTES = 1 << 20;
int pin_take(size_t bytes, void *dst, char **at) {
    PinStage &s = g_pin;
    if (!s.p) {
        if (hipHostMalloc((void **)&s.p, PIN_BYTES, hipHostMallocDefault) != hipSuccess) return SCP_ENOMEM;
        s.cap = PIN_BYTES;
    }
    const size_t off = (s.used + 63) & ~(size_t)63;
    if (off + bytes > s.cap) return SCP_ESMALL;        // (a megabyte: the read-backs of a build are a few kilobytes)
    s.items.push_back({dst, off, bytes});
    s.used = off + bytes;
    *at = s.p + off;
    return SCP_OK;
}
}  // namespace

// The staging is transactional: an item is queued only once its copy has been enqueued (a failed copy takes its item back), and a caller that
// leaves between a read-back and its scp_stream_wait drops what it queued with scp_d2h_abort (SCP_D2H_TRY in geom.hip) - the destinations are
// stack and local-vector addresses, which the NEXT wait of this thread must never write to.
static int pin_untake(int rc_hip) {
    PinStage &s = g_pin;
    s.used = s.items.back().off;
    s.items.pop_back();
    g_scp_last_hip_error = rc_hip;
    return SCP_EHIP;
}

int scp_d2h_async(void *dst, const void *src, size_t bytes, hipStream_t st) {
    char *at;
    const int rc = pin_take(bytes, dst, &at);
    if (rc) return rc;
    const hipError_t e = hipMemcpyAsync(at, src, bytes, hipMemcpyDeviceToHost, st);
    return e == hipSuccess ? SCP_OK : pin_untake((int)e);
}

int scp_d2h_2d_async(void *dst, const void *src, size_t spitch, size_t width, size_t height, hipStream_t st) {
    char *at;
    const int rc = pin_take(width * height, dst, &at);
    if (rc) return rc;
    const hipError_t e = hipMemcpy2DAsync(at, width, src, spitch, width, height, hipMemcpyDeviceToHost, st);
    return e == hipSuccess ? SCP_OK : pin_untake((int)e);
}

void scp_d2h_abort() {
    PinStage &s = g_pin;
    s.items.clear();
    s.used = 0;
}

static int stream_wait_event(hipStream_t st) {
    static thread_local hipEvent_t ev = nullptr;
    static thread_local int ev_dev = -1;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (!ev || ev_dev != dev) {           // (an event belongs to the device that was current when it was made)
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        ev_dev = dev;
    }
    HIP_TRY(hipEventRecord(ev, st));
    for (int spins = 0;; ++spins) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) { (void)hipGetLastError(); return SCP_OK; }      // (drop a recorded hipErrorNotReady: the next LAUNCH_CHECK must not see it)
        if (q != hipErrorNotReady) { HIP_TRY(q); }
        if (spins < 20) continue;          // a wait that is over within microseconds (the GPU was idle) costs no sleep
        struct timespec ts = {0, 50000};
        nanosleep(&ts, nullptr);
    }
}

int scp_stream_wait(hipStream_t st) {
    const int rc = stream_wait_event(st);
    PinStage &s = g_pin;
    if (!rc)
        for (const auto &it : s.items) memcpy(it.dst, s.p + it.off, it.bytes);
    s.items.clear();
    s.used = 0;
    return rc;
}
