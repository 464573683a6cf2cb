// kao_session.cpp -- host side of libkao.so, part 2 of 3 (see kao_host.h): the runtime (device selection, error text, arena
// and stream pools), K-eval plans, the canonical tie-break, and sessions -- the resident state of a batch of topics on one
// device with its K-search / K-eval steps, K-bound launches and search prices.
//
// Everything that computes runs in the gfx950 kernels (kao_kernels.hip, kao_bound.hip, kao_cycle.hip); this file only prepares
// instances, owns the device pools, launches, and reads results back.  There is deliberately no CPU evaluation or search
// path here: if the HIP device is missing every compute entry point fails with KAO_ERR_NO_DEVICE.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "kao_host.h"

namespace kao {

thread_local int t_device = -1;
thread_local double g_timing[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
thread_local double g_profile[8] = {0, 0, 0, 0, 0, 0, 0, 0};
thread_local double g_lp[8] = {0, 0, 0, 0, 0, 0, 0, 0};

namespace {
thread_local std::string g_err;
int g_device = -1;           // the process default (kao_init)
bool g_init = false;
int g_num_cu_of[kMaxDevices] = {0};
}  // namespace

int cur_device() { return t_device >= 0 ? t_device : g_device; }
int num_cu(int device) {
    if (device < 0 || device >= kMaxDevices) return 256;
    if (!g_num_cu_of[device]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || v <= 0) v = 256;
        g_num_cu_of[device] = v;
    }
    return g_num_cu_of[device];
}

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

double now_s() {
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

bool is_init() { return g_init; }
int require_init() {
    if (!g_init) {
        int rc = kao_init(g_device < 0 ? 0 : g_device);
        if (rc) return rc;
    }
    HIP_TRY(hipSetDevice(cur_device()));
    return KAO_OK;
}

}  // namespace kao

namespace {

template <typename T>
int dev_alloc_copy(T **dst, const std::vector<T> &src) {
    *dst = nullptr;
    const size_t n = std::max<size_t>(src.size(), 1);
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(dst), n * sizeof(T)));
    if (!src.empty()) HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return KAO_OK;
}

// Physical workgroup order: the dispatcher places workgroup b on XCD b % 8 (observed, used for L2
// affinity only), so the workgroups of one topic -- which read the same tables -- are dealt to one XCD.
template <typename Item>
std::vector<Item> xcd_order(const std::vector<Item> &items, const std::vector<int> &topic_of) {
    const size_t n = items.size();
    std::vector<std::vector<size_t>> q(8);
    for (size_t i = 0; i < n; ++i) q[(size_t)topic_of[i] % 8].push_back(i);
    std::vector<size_t> head(8, 0);
    std::vector<Item> out;
    out.reserve(n);
    for (size_t b = 0; b < n; ++b) {
        size_t x = b % 8;
        if (head[x] >= q[x].size()) {  // this XCD's queue ran dry: steal from the longest remaining one
            size_t bestx = 0, bestlen = 0;
            for (size_t y = 0; y < 8; ++y)
                if (q[y].size() - head[y] > bestlen) { bestlen = q[y].size() - head[y]; bestx = y; }
            x = bestx;
            out.push_back(items[q[x].back()]);
            q[x].pop_back();
            continue;
        }
        out.push_back(items[q[x][head[x]++]]);
    }
    return out;
}

}  // namespace

namespace {

constexpr int kEvRing = 32;

// hipMalloc / hipFree cost 0.1-1 ms each; a finished session parks its arenas here for the next one
struct Parked { void *p; size_t bytes; int device; };
std::vector<Parked> g_parked;
constexpr size_t kParkMax = 4;
std::mutex g_cache_mu;  // guards g_parked / g_streams (sessions may be created from several host threads)

int arena_get(size_t bytes, void **out, size_t *cap) {
    std::lock_guard<std::mutex> lock(g_cache_mu);
    const int dev = cur_device();
    size_t best = g_parked.size();
    for (size_t i = 0; i < g_parked.size(); ++i)
        if (g_parked[i].device == dev && g_parked[i].bytes >= bytes && g_parked[i].bytes <= 4 * bytes + (1u << 20) &&
            (best == g_parked.size() || g_parked[i].bytes < g_parked[best].bytes)) best = i;
    if (best < g_parked.size()) {
        *out = g_parked[best].p; *cap = g_parked[best].bytes;
        g_parked.erase(g_parked.begin() + (long)best);
        return KAO_OK;
    }
    const size_t want = ((bytes + (1u << 16)) + 4095) & ~(size_t)4095;
    HIP_TRY(hipMalloc(out, want));
    *cap = want;
    return KAO_OK;
}
void arena_put(void *p, size_t bytes, int device) {
    if (!p) return;
    std::lock_guard<std::mutex> lock(g_cache_mu);
    (void)hipSetDevice(device);
    if (g_parked.size() >= kParkMax) {
        size_t small = 0;
        for (size_t i = 1; i < g_parked.size(); ++i) if (g_parked[i].bytes < g_parked[small].bytes) small = i;
        if (g_parked[small].bytes >= bytes) { (void)hipFree(p); return; }
        (void)hipSetDevice(g_parked[small].device);
        (void)hipFree(g_parked[small].p);
        (void)hipSetDevice(device);
        g_parked.erase(g_parked.begin() + (long)small);
    }
    g_parked.push_back({p, bytes, device});
}
std::vector<std::pair<hipStream_t, int>> g_streams;  // parked streams with their device (create/destroy cost ~1 ms)
int stream_get(hipStream_t *out) {
    std::lock_guard<std::mutex> lock(g_cache_mu);
    const int dev = cur_device();
    for (size_t i = 0; i < g_streams.size(); ++i)
        if (g_streams[i].second == dev) { *out = g_streams[i].first; g_streams.erase(g_streams.begin() + (long)i); return KAO_OK; }
    HIP_TRY(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    return KAO_OK;
}
void stream_put(hipStream_t st, int device) {
    if (!st) return;
    std::lock_guard<std::mutex> lock(g_cache_mu);
    if (g_streams.size() < 16) g_streams.push_back({st, device}); else { (void)hipSetDevice(device); (void)hipStreamDestroy(st); }
}
void arena_drop_all() {
    std::lock_guard<std::mutex> lock(g_cache_mu);
    for (auto &a : g_parked) { (void)hipSetDevice(a.device); (void)hipFree(a.p); }
    g_parked.clear();
    for (auto &st : g_streams) { (void)hipSetDevice(st.second); (void)hipStreamDestroy(st.first); }
    g_streams.clear();
    if (g_device >= 0) (void)hipSetDevice(g_device);
}
inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

int session_drain_events(kao_session *s) {
    for (int i = 0; i < s->ev_pending; ++i) {
        float a = 0, b = 0;
        HIP_TRY(hipEventSynchronize(s->ev[i * 3 + 2]));
        HIP_TRY(hipEventElapsedTime(&a, s->ev[i * 3 + 0], s->ev[i * 3 + 1]));
        HIP_TRY(hipEventElapsedTime(&b, s->ev[i * 3 + 1], s->ev[i * 3 + 2]));
        s->ms_search += a;
        s->ms_eval += b;
    }
    s->ev_pending = 0;
    return KAO_OK;
}

}  // namespace

extern "C" {

int kao_version(void) { return KAO_VERSION; }

const char *kao_strerror(int code) {
    switch (code) {
        case KAO_OK: return "ok";
        case KAO_ERR_INVALID: return "invalid argument";
        case KAO_ERR_UNSUPPORTED: return "instance not supported by the gfx950 kernels";
        case KAO_ERR_NO_DEVICE: return "no usable HIP device (libkao has no CPU fallback)";
        case KAO_ERR_HIP: return "HIP runtime error";
        case KAO_ERR_NOMEM: return "out of memory";
        case KAO_ERR_NOT_INIT: return "kao_init not called";
        default: return "unknown error";
    }
}

const char *kao_last_error(void) { return g_err.c_str(); }

int kao_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(KAO_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(KAO_ERR_INVALID, "device ordinal out of range");
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(KAO_ERR_NO_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
    (void)num_cu(device);
    g_device = device;
    g_init = true;
    return KAO_OK;
}

void kao_multi_shutdown_comms(void);

void kao_shutdown(void) {
    kao_multi_shutdown_comms();
    if (g_init) arena_drop_all();
    g_init = false;
}

int kao_device_name(char *buf, int len) {
    int rc = require_init();
    if (rc) return rc;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cur_device()));
    std::snprintf(buf, (size_t)len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return KAO_OK;
}

// ------------------------------------------------------------------------------------------------
// K-eval plans
// ------------------------------------------------------------------------------------------------
int kao_eval_plan_create(const kao_topic *t, kao_eval_plan **out) {
    if (!out) return fail(KAO_ERR_INVALID, "null out");
    *out = nullptr;
    int rc = require_init();
    if (rc) return rc;
    kao_eval_plan *p = new kao_eval_plan();
    p->device = cur_device();
    rc = prepare(t, 0, p->pt);
    if (rc) { delete p; return rc; }
    p->cur_in_lds = eval_lds_bytes(p->pt.d.P, p->pt.d.B, true, p->pt.d.nw) <= 160 * 1024;
    if (eval_lds_bytes(p->pt.d.P, p->pt.d.B, p->cur_in_lds, p->pt.d.nw) > 160 * 1024) { delete p; return fail(KAO_ERR_UNSUPPORTED, "broker tables exceed 160 KiB of LDS"); }
    p->pt.d.best_off = 0; p->pt.d.rackof_off = 0; p->pt.d.curd_off = 0; p->pt.d.bwd_off = 0;
    std::vector<TopicDev> td(1, p->pt.d);
    if ((rc = dev_alloc_copy(&p->d_topic, td)) || (rc = dev_alloc_copy(&p->d_rackof, p->pt.rack_of)) ||
        (rc = dev_alloc_copy(&p->d_curd, p->pt.cur_dense)) || (p->pt.d.has_bw && (rc = dev_alloc_copy(&p->d_bwd, p->pt.bw_dense)))) { kao_eval_plan_destroy(p); return rc; }
    hipError_t e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    if (e == hipSuccess && (int64_t)p->pt.d.P * p->pt.d.RF > 65535) {
        e = hipMalloc(reinterpret_cast<void **>(&p->d_overflow), 4);
        if (e == hipSuccess) e = hipMemset(p->d_overflow, 0, 4);
    }
    if (e == hipSuccess) e = hipEventCreate(&p->ev0);
    if (e == hipSuccess) e = hipEventCreate(&p->ev1);
    if (e != hipSuccess) { kao_eval_plan_destroy(p); return fail(KAO_ERR_HIP, std::string("kao_eval_plan_create: ") + hipGetErrorString(e)); }
    *out = p;
    return KAO_OK;
}

int kao_eval_plan_run(kao_eval_plan *p, const void *d_candidates, int64_t n, void *d_objective, void *d_violations,
                      void *d_best_key) {
    if (!p || !d_candidates || n < 1) return fail(KAO_ERR_INVALID, "bad plan/candidates");
    if (n > (1 << 20)) return fail(KAO_ERR_INVALID, "at most 2^20 candidates per run (packed key id width)");
    HIP_TRY(hipSetDevice(p->device));
    if (n != p->map_n) {
        // small batches of large candidates (KAO-CX: <= 513 assignments of up to 10^5 slots) spread over the compute units: one
        // candidate per wavefront instead of eight, as soon as 32 per workgroup would leave most of the chip idle
        int cpb = p->cands_per_block;
        const int64_t fill = 4 * (int64_t)std::max(num_cu(p->device), 1);
        if (n < fill * 8) cpb = (int)std::min<int64_t>(cpb, std::max<int64_t>(kWaves, ((n + fill - 1) / fill) * kWaves));
        p->coop = n <= fill && p->pt.d.P >= 1024;   // few large candidates: one workgroup each, its wavefronts cooperating
        if (p->coop) cpb = 1;
        const int nb = (int)((n + cpb - 1) / cpb);
        std::vector<int4> map((size_t)nb);
        for (int b = 0; b < nb; ++b) {
            const int first = b * cpb;
            map[b] = make_int4(0, first, (int)std::min<int64_t>(cpb, n - first), first);
        }
        p->map_n = -1;  // no valid map until the new one is uploaded
        if (p->d_map) { int4 *old_map = p->d_map; p->d_map = nullptr; HIP_TRY(hipFree(old_map)); }
        int rc = dev_alloc_copy(&p->d_map, map);
        if (rc) return rc;
        p->map_n = n;
        p->map_blocks = nb;
    }
    EvalPools pl{};
    pl.topics = p->d_topic; pl.block_map = p->d_map; pl.rackof_pool = p->d_rackof; pl.curd_pool = p->d_curd;
    pl.cand = static_cast<const uint16_t *>(d_candidates);
    pl.objective = static_cast<int32_t *>(d_objective);
    pl.violations = static_cast<int32_t *>(d_violations);
    pl.best_key = static_cast<unsigned long long *>(d_best_key);
    pl.maxP = p->pt.d.P; pl.maxB = p->pt.d.B; pl.cur_in_lds = p->cur_in_lds ? 1 : 0; pl.coop = p->coop ? 1 : 0; pl.rf_uniform = p->pt.d.RF;
    pl.overflow = p->d_overflow; pl.bwd_pool = p->d_bwd;
    HIP_TRY(hipEventRecord(p->ev0, p->stream));
    launch_eval(pl, p->map_blocks, p->pt.d.nw, p->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(p->ev1, p->stream));
    p->timed = true;
    return KAO_OK;
}

int kao_eval_plan_sync(kao_eval_plan *p, double *ms_last) {
    if (!p) return fail(KAO_ERR_INVALID, "null plan");
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (p->d_overflow) {
        int32_t flag = 0;
        HIP_TRY(hipMemcpy(&flag, p->d_overflow, 4, hipMemcpyDeviceToHost));
        if (flag) {
            HIP_TRY(hipMemset(p->d_overflow, 0, 4));
            return fail(KAO_ERR_UNSUPPORTED, "a candidate puts more than 65,535 replicas on one broker (16-bit per-broker counters)");
        }
    }
    if (ms_last) {
        float ms = 0;
        if (p->timed) HIP_TRY(hipEventElapsedTime(&ms, p->ev0, p->ev1));
        *ms_last = ms;
    }
    return KAO_OK;
}

void kao_eval_plan_destroy(kao_eval_plan *p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    (void)hipFree(p->d_topic); (void)hipFree(p->d_rackof); (void)hipFree(p->d_curd); (void)hipFree(p->d_map); (void)hipFree(p->d_overflow); (void)hipFree(p->d_bwd);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

namespace {
// a plan plus growable device buffers, reused across batches (kao_canonicalize issues many small ones)
struct EvalCtx {
    kao_eval_plan *plan = nullptr;
    size_t per = 0, cap = 0;
    uint16_t *d_c = nullptr; int32_t *d_o = nullptr, *d_v = nullptr;
    ~EvalCtx() { (void)hipFree(d_c); (void)hipFree(d_o); (void)hipFree(d_v); kao_eval_plan_destroy(plan); }
    int open(const kao_topic *t) {
        per = (size_t)t->n_partitions * t->rf;
        return kao_eval_plan_create(t, &plan);
    }
    int run(const uint16_t *candidates, int64_t n, int32_t *objective, int32_t *violations) {
        const int64_t chunk_max = 1 << 20;
        for (int64_t done = 0; done < n; done += chunk_max) {
            const int64_t c = std::min(chunk_max, n - done);
            if ((size_t)c > cap) {
                (void)hipFree(d_c); (void)hipFree(d_o); (void)hipFree(d_v);
                d_c = nullptr; d_o = d_v = nullptr;
                cap = std::max<size_t>((size_t)c, std::min<size_t>(2 * cap + 64, (size_t)chunk_max));
                if (hipMalloc(reinterpret_cast<void **>(&d_c), cap * per * 2) != hipSuccess ||
                    hipMalloc(reinterpret_cast<void **>(&d_o), cap * 4) != hipSuccess ||
                    hipMalloc(reinterpret_cast<void **>(&d_v), cap * 32) != hipSuccess) { cap = 0; return fail(KAO_ERR_NOMEM, "hipMalloc"); }
            }
            HIP_TRY(hipMemcpy(d_c, candidates + (size_t)done * per, (size_t)c * per * 2, hipMemcpyHostToDevice));
            int rc = kao_eval_plan_run(plan, d_c, c, d_o, d_v, nullptr);
            if (!rc) rc = kao_eval_plan_sync(plan, nullptr);
            if (rc) return rc;
            HIP_TRY(hipMemcpy(objective + done, d_o, (size_t)c * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(violations + done * 8, d_v, (size_t)c * 32, hipMemcpyDeviceToHost));
        }
        return KAO_OK;
    }
};
}  // namespace

int kao_evaluate_batch(const kao_topic *t, const uint16_t *candidates, int64_t n, int32_t *objective, int32_t *violations) {
    if (!candidates || !objective || !violations || n < 1) return fail(KAO_ERR_INVALID, "null buffers");
    EvalCtx ctx;
    int rc = ctx.open(t);
    if (rc) return rc;
    return ctx.run(candidates, n, objective, violations);
}

int kao_evaluate(const kao_topic *t, const uint16_t *assignment, int64_t *objective, int32_t violations[8]) {
    int32_t obj = 0;
    int rc = kao_evaluate_batch(t, assignment, 1, &obj, violations);
    if (!rc && objective) *objective = obj;
    return rc;
}

int kao_canonicalize(const kao_topic *t, uint16_t *a) {
    if (!a) return fail(KAO_ERR_INVALID, "null assignment");
    if (t && (t->broker_w || t->broker_wl)) return KAO_OK;   // moving a replica to another broker changes the objective: nothing to canonicalise
    int rc = require_init();
    if (rc) return rc;
    PreparedTopic pt;
    if ((rc = prepare(t, 0, pt))) return rc;
    const TopicDev &d = pt.d;
    const int P = d.P, RF = d.RF, B = d.B;
    if (canon_lds_bytes(d.Bx) > 160 * 1024) return fail(KAO_ERR_UNSUPPORTED, "broker tables exceed 160 KiB of LDS");
    auto word = [&](uint16_t x) { return x == KAO_NONE ? kNoneW : ((uint32_t)x | ((uint32_t)(x / d.m) << 16)); };
    const int nw = d.nw;
    std::vector<uint32_t> cur_words((size_t)P * nw), a_words((size_t)P * nw, kNoneW);
    for (int p = 0; p < P; ++p) {
        const uint16_t *c = &pt.cur_int[(size_t)p * nw];
        for (int k = 0; k < nw; ++k) cur_words[(size_t)p * nw + k] = word(c[k]);
        for (int k = 0; k < RF; ++k) {
            const unsigned b = a[(size_t)p * RF + k];
            if (b >= (unsigned)B) return KAO_OK;  // an empty slot: infeasible, nothing to polish
            a_words[(size_t)p * nw + k] = word(pt.int_of[b]);
        }
    }
    // one device buffer: [TopicDev][status 16 B][cur words][A words][ext][rsz]
    const size_t wbytes = (size_t)P * nw * 4;
    const size_t o_status = align_up(sizeof(TopicDev)), o_cur = o_status + 256, o_a = o_cur + align_up(wbytes);
    const size_t o_ext = o_a + align_up(wbytes), o_rsz = o_ext + align_up(pt.ext_of.size() * 2);
    const size_t total = o_rsz + align_up(pt.rack_size.size() * 4);
    std::vector<unsigned char> stage(total, 0);
    std::memcpy(stage.data(), &d, sizeof(TopicDev));
    std::memcpy(stage.data() + o_cur, cur_words.data(), wbytes);
    std::memcpy(stage.data() + o_a, a_words.data(), wbytes);
    std::memcpy(stage.data() + o_ext, pt.ext_of.data(), pt.ext_of.size() * 2);
    std::memcpy(stage.data() + o_rsz, pt.rack_size.data(), pt.rack_size.size() * 4);
    void *dev = nullptr; size_t cap = 0;
    if ((rc = arena_get(total, &dev, &cap))) return rc;
    unsigned char *db = static_cast<unsigned char *>(dev);
    hipStream_t st = nullptr;
    if ((rc = stream_get(&st))) { arena_put(dev, cap, cur_device()); return rc; }
    int32_t status[2] = {0, 0};
    hipError_t e = hipMemcpyAsync(db, stage.data(), total, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        launch_canon(reinterpret_cast<const TopicDev *>(db), reinterpret_cast<const uint32_t *>(db + o_cur),
                     reinterpret_cast<const uint16_t *>(db + o_ext), reinterpret_cast<const int32_t *>(db + o_rsz),
                     reinterpret_cast<uint32_t *>(db + o_a), d.Bx, nw, reinterpret_cast<int32_t *>(db + o_status), st);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(a_words.data(), db + o_a, wbytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(status, db + o_status, sizeof status, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    stream_put(st, cur_device());
    arena_put(dev, cap, cur_device());
    if (e != hipSuccess) return fail(KAO_ERR_HIP, std::string("kao_canonicalize: ") + hipGetErrorString(e));
    if (!status[0]) return KAO_OK;  // only feasible assignments are polished
    for (int p = 0; p < P; ++p) {
        for (int k = 0; k < RF; ++k) a[(size_t)p * RF + k] = pt.ext_of[a_words[(size_t)p * nw + k] & 0xFFFFu];
    }
    for (int p = 0; p < P; ++p) {  // followers: retained ones in their current order, then new ones ascending
        std::vector<uint16_t> fol(a + (size_t)p * RF + 1, a + (size_t)p * RF + RF), kept, fresh;
        for (int k = 0; k < t->rf_cur; ++k) {
            const uint16_t c = t->current[(size_t)p * t->rf_cur + k];
            if (std::find(fol.begin(), fol.end(), c) != fol.end() && std::find(kept.begin(), kept.end(), c) == kept.end()) kept.push_back(c);
        }
        for (uint16_t f : fol) if (std::find(kept.begin(), kept.end(), f) == kept.end()) fresh.push_back(f);
        std::sort(fresh.begin(), fresh.end());
        kept.insert(kept.end(), fresh.begin(), fresh.end());
        std::copy(kept.begin(), kept.end(), a + (size_t)p * RF + 1);
    }
    return KAO_OK;
}

// ------------------------------------------------------------------------------------------------
// sessions
// ------------------------------------------------------------------------------------------------
void kao_session_destroy(kao_session *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->stream_bound) (void)hipStreamSynchronize(s->stream_bound);
    if (s->ev_bound0) (void)hipEventDestroy(s->ev_bound0);
    if (s->ev_bound1) (void)hipEventDestroy(s->ev_bound1);
    if (s->ev_search) (void)hipEventDestroy(s->ev_search);
    if (s->stream_bound) (void)hipStreamDestroy(s->stream_bound);
    arena_put(s->arena_ro, s->arena_ro_bytes, s->device);
    arena_put(s->arena_rw, s->arena_rw_bytes, s->device);
    for (hipEvent_t e : s->ev) (void)hipEventDestroy(e);
    stream_put(s->stream, s->device);
    delete s;
}

int kao_session_create(const kao_topic *topics, int32_t n_topics, const kao_opts *opts_in, kao_session **out) {
    if (!out) return fail(KAO_ERR_INVALID, "null out");
    *out = nullptr;
    if (!topics || n_topics < 1) return fail(KAO_ERR_INVALID, "no topics");
    int rc = require_init();
    if (rc) return rc;
    kao_session *s = new kao_session();
    s->device = cur_device();
    s->n_topics = n_topics;
    const int g_num_cu = num_cu(s->device);
    kao_opts o{};
    if (opts_in) o = *opts_in;
    if (o.team < 0 || o.team > kTeamMax || o.schedule < 0 || o.schedule > 1) { kao_session_destroy(s); return fail(KAO_ERR_INVALID, "kao_opts: team must be 0..8, schedule 0 or 1"); }
    if (o.iters_per_launch <= 0) o.iters_per_launch = 512;
    if (o.obj_scale <= 0) o.obj_scale = 4;
    if (o.lam_min <= 0) o.lam_min = 1;
    if (o.lam_max <= 0) o.lam_max = 40;
    if (o.lam_max < o.lam_min) o.lam_max = o.lam_min;
    if (o.period_log2 < 0) o.period_log2 = 0;    // 0 = per topic, by size (auto_period_log2)
    if (o.period_log2 > 20) o.period_log2 = 20;
    if (o.time_limit_s <= 0) o.time_limit_s = 10.0;
    if (o.elite_period < 0) o.elite_period = 0;  // sessions: 0 = never (kao_solve picks its own default before creating the session)
    const bool auto_restarts = o.restarts <= 0;
    if (auto_restarts) {  // one full round of resident wavefronts (8 per SIMD = 32 per CU) across all topics
        const int want = g_num_cu * 32;
        int r = want / n_topics;
        r = (r / kWaves) * kWaves;
        o.restarts = std::min(std::max(r, 8), 8192);
        // large topics need depth (iterations per second) more than breadth: at most 2^22 replica slots over all the
        // restarts of the largest topic, but never fewer than one restart per compute unit
        int64_t slots = 1;
        for (int t = 0; t < n_topics; ++t) slots = std::max<int64_t>(slots, (int64_t)topics[t].n_partitions * std::max(topics[t].rf, 1));
        // (round 4: four per compute unit where the topic lives in HBM -- a restart is then one wavefront waiting on dependent global
        //  loads, and 1024 of them take the time of 256: drifted 1000 x 30000 3.98 -> 4.43 ms per launch, profiles/r04_a_fill_probe.txt)
        const int floor_r = slots >= 32768 ? 4 * g_num_cu : g_num_cu;
        const int cap = (int)std::max<int64_t>(floor_r, (((int64_t)1 << 22) / slots) / kWaves * kWaves);
        o.restarts = std::min(o.restarts, cap);
        // a single topic whose WORKING assignment alone fits LDS (k_search_curg, ~4,900 .. 9,800 partitions): one wavefront per workgroup
        // and compute unit, so a count a little above the compute units (2^22 / slots = 276 at 5,000 partitions) would pay a whole second
        // round of workgroups for 20 restarts
        if (n_topics == 1 && o.restarts > g_num_cu && o.restarts <= g_num_cu + g_num_cu / 4 && o.team <= 1) {
            const kao_topic &t0 = topics[0];
            std::vector<int> rs((size_t)std::max(t0.n_racks, 1), 0);
            for (int b = 0; b < t0.n_brokers; ++b) if (t0.rack_of[b] < rs.size()) rs[t0.rack_of[b]]++;
            const int bx = *std::max_element(rs.begin(), rs.end()) * t0.n_racks, nw0 = (t0.rf > kRFP || t0.rf_cur > kRFP) ? 2 * kRFP : kRFP;
            const bool hbw = t0.broker_w || t0.broker_wl;
            if (search_lds_bytes(t0.n_partitions, bx, 1, false, true, nw0, hbw, t0.n_racks) > 160 * 1024 &&
                search_lds_bytes(t0.n_partitions, bx, 1, false, true, nw0, hbw, t0.n_racks, 0, true) <= 160 * 1024)
                o.restarts = g_num_cu / kWaves * kWaves;
        }
    }
    if (o.restarts > (1 << 20) - 2) o.restarts = (1 << 20) - 2;  // id 0xFFFFF is reserved (kExternalRestart)
    {   // huge topics: bound the per-restart state in HBM (16 B of working words + the snapshot per partition):
        // at most 8 GB and at most a quarter of what the device has free right now (the constant assumed the 288 GB
        // device with nothing beside the session; multi-GPU replicas and capped rounds hold several sessions)
        uint64_t per_restart = 0;
        for (int t = 0; t < n_topics; ++t) per_restart += (uint64_t)topics[t].n_partitions * (32 + 2 * (uint64_t)std::max(topics[t].rf, 1));
        size_t free_b = 0, total_b = 0;
        uint64_t budget = 8ull << 30;
        // keyed to the device's TOTAL memory, not to what happens to be free (same input, same seed, same answer whatever else
        // occupies the device; an allocation that does not fit fails loudly with KAO_ERR_NOMEM instead of changing the search)
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) budget = std::min<uint64_t>(budget, std::max<uint64_t>(total_b / 8, 256ull << 20));
        const uint64_t cap = budget / std::max<uint64_t>(per_restart, 1);
        if ((uint64_t)o.restarts > cap) o.restarts = (int)std::max<uint64_t>(cap / kWaves * kWaves, kWaves);
    }
    s->opts = o;
    s->pts.resize((size_t)n_topics);
    s->topics.assign(topics, topics + n_topics);
    s->ub.resize((size_t)n_topics);

    std::vector<uint32_t> cur_pool, bw_pool, bwd_pool; std::vector<uint16_t> ext_pool, curd_pool, int_pool; std::vector<int32_t> rsz_pool;
    uint64_t price_i32 = 0;
    std::vector<uint8_t> rackof_pool;
    uint64_t state_bytes = 0, best_u16 = 0, win_u16 = 0, dual_i32 = 0, wide_slices = 0;
    s->topic_global.assign((size_t)n_topics, 0);
    s->topic_curg.assign((size_t)n_topics, 0);
    for (int t = 0; t < n_topics; ++t) s->any_bw |= topics[t].broker_w || topics[t].broker_wl;
    int restart_base = 0;
    for (int t = 0; t < n_topics; ++t) {
        PreparedTopic &pt = s->pts[(size_t)t];
        const uint64_t seed = o.seed ^ ((uint64_t)(t + 1) * 0x9E3779B97F4A7C15ull);
        rc = prepare(&topics[t], seed, pt);
        if (rc) { kao_session_destroy(s); return rc; }
        s->ub[(size_t)t] = upper_bound_w(&topics[t]);
        s->topic_infeasible.push_back(infeasible_reason(&topics[t]).empty() ? 0 : 1);
        TopicDev &d = pt.d;
        d.n_restarts = o.restarts;
        d.period_log2 = o.period_log2 > 0 ? o.period_log2 : auto_period_log2(d.P, d.RF);
        d.restart_base = restart_base;
        restart_base += o.restarts;
        auto word = [&](uint16_t x) { return x == KAO_NONE ? kNoneW : ((uint32_t)x | ((uint32_t)(x / d.m) << 16)); };
        while (cur_pool.size() % 4) cur_pool.push_back(kNoneW);   // every topic's words start 16-byte aligned
        d.cur_off = (uint32_t)cur_pool.size();
        for (size_t i = 0; i < (size_t)d.P * d.nw; ++i) cur_pool.push_back(word(pt.cur_int[i]));  // LDS / register form of a replica: internal index | rack << 16
        {   // the holes best insertion has to fill (slots below RF without a surviving replica): leader holes, then follower holes
            d.hole_off = (uint32_t)cur_pool.size();
            const size_t w0 = (size_t)d.cur_off;
            std::vector<uint32_t> lead, foll;
            for (int p = 0; p < d.P; ++p) {
                bool fh = false;
                for (int k = 1; k < d.RF; ++k) fh |= cur_pool[w0 + (size_t)p * d.nw + k] == kNoneW;
                if (cur_pool[w0 + (size_t)p * d.nw] == kNoneW) lead.push_back((uint32_t)p);
                if (fh) foll.push_back((uint32_t)p);
            }
            cur_pool.push_back((uint32_t)lead.size()); cur_pool.push_back((uint32_t)foll.size());
            cur_pool.insert(cur_pool.end(), lead.begin(), lead.end());
            cur_pool.insert(cur_pool.end(), foll.begin(), foll.end());
        }
        // both assignments in LDS when they fit (one wavefront per workgroup at least); else the working one alone, the current one read
        // from global memory / L2 (round 5: ~4,900 .. 9,800 partitions; KAO_CUR_GLOBAL=0: the HBM path as before); else everything in HBM
        const bool both_fit = search_lds_bytes(d.P, d.Bx, 1, false, true, d.nw, s->any_bw, d.R) <= 160 * 1024;
        const bool curg_on = [] { const char *e = std::getenv("KAO_CUR_GLOBAL"); return !(e && e[0] == '0'); }();   // (read per session: a test hook)
        // (one workgroup of ONE wavefront per restart and compute unit: with more restarts than compute units the workgroups run in rounds
        // and the HBM path, four wavefronts per workgroup, is the faster one -- 500 x 5000: 3.1 against 3.9 ms a launch at 256 restarts,
        // 12.4 against 4.2 at 1,024; teams (kao_opts.team) are a global-memory mode)
        static const bool team_env = [] { const char *e = std::getenv("KAO_TEAM"); return e && std::atoi(e) > 1; }();
        const bool curg = !both_fit && curg_on && o.team <= 1 && !team_env && o.restarts <= std::max(num_cu(s->device), 1) &&
                          search_lds_bytes(d.P, d.Bx, 1, false, true, d.nw, s->any_bw, d.R, 0, true) <= 160 * 1024;
        const bool global_a = !both_fit && !curg;
        s->topic_global[(size_t)t] = global_a;
        s->topic_curg[(size_t)t] = curg;
        d.ext_off = (uint32_t)ext_pool.size();
        ext_pool.insert(ext_pool.end(), pt.ext_of.begin(), pt.ext_of.end());
        d.rsz_off = (uint32_t)rsz_pool.size();
        rsz_pool.insert(rsz_pool.end(), pt.rack_size.begin(), pt.rack_size.end());
        d.state_off = state_bytes;  // bytes: 8 per partition (packed, LDS path) or 16 (working words, global path)
        state_bytes += align_up((uint64_t)o.restarts * d.P * d.nw * (global_a ? 4 : 2));
        d.best_off = best_u16;
        best_u16 += (uint64_t)o.restarts * d.P * d.RF;
        d.win_off = (uint32_t)win_u16;
        win_u16 += (uint64_t)d.P * d.RF;
        d.rackof_off = (uint32_t)rackof_pool.size();
        rackof_pool.insert(rackof_pool.end(), pt.rack_of.begin(), pt.rack_of.end());
        d.curd_off = (uint32_t)curd_pool.size();
        curd_pool.insert(curd_pool.end(), pt.cur_dense.begin(), pt.cur_dense.end());
        if (d.has_bw) {
            d.bw_off = (uint32_t)bw_pool.size(); bw_pool.insert(bw_pool.end(), pt.bw_int.begin(), pt.bw_int.end());
            d.bwd_off = (uint32_t)bwd_pool.size(); bwd_pool.insert(bwd_pool.end(), pt.bw_dense.begin(), pt.bw_dense.end());
            s->priced = true;   // broker weights live in the tables of the priced K-search instantiation
            s->any_bw = true;
        }
        d.int_off = (uint32_t)int_pool.size();
        int_pool.insert(int_pool.end(), pt.int_of.begin(), pt.int_of.end());
        d.price_off = (uint32_t)price_i32;
        price_i32 += 2 * (uint64_t)d.B + kRackTab;
        dual_i32 = (dual_i32 + 1) & ~(uint64_t)1;  // the level-control words are 64-bit
        d.dual_off = (uint32_t)dual_i32;
        dual_i32 += 6 * (uint64_t)d.B + 3 * kRackTab + 8;
        d.cnt_off = (uint32_t)dual_i32;             // counters of the sliced K-bound live in the same pool (zeroed with it)
        dual_i32 += 3 * (2 * (uint64_t)d.B + kRackTab) + 4 * (uint64_t)d.B + 2 * kRackTab;   // x 3 + shadow area: k_bound_multi
        wide_slices += (uint64_t)(d.P + 63) / 64;
        s->dual_ok.push_back(dual_supported(&topics[t], s->any_bw) ? 1 : 0);
        // algorithmic bytes: full evaluation = 2*RF*P + 2*rf_cur*P + B per candidate
        s->eval_bytes_per_launch += (uint64_t)o.restarts * (uint64_t)(2 * d.RF * d.P + 2 * d.rf_cur * d.P + d.B);
    }
    s->total_restarts = restart_base;
    // ---- launch groups: topics sorted by single-wave LDS need, a new group whenever the need doubles (<= 8 groups) ----
    std::vector<int> order((size_t)n_topics);
    for (int t = 0; t < n_topics; ++t) order[(size_t)t] = t;
    auto need1 = [&](int t) {  // topics kept in global memory sort last (their LDS need is tiny but they form their own groups)
        const TopicDev &d = s->pts[(size_t)t].d;
        const bool ga = s->topic_global[(size_t)t] != 0;
        const bool cg = s->topic_curg[(size_t)t] != 0;
        return (ga ? ((size_t)1 << 40) : 0) + (cg ? ((size_t)1 << 39) : 0) + (d.nw > kRFP ? ((size_t)1 << 41) : 0) + search_lds_bytes(d.P, d.Bx, 1, ga, true, d.nw, s->any_bw, d.R, 0, cg);
    };
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return need1(x) < need1(y); });
    std::vector<std::vector<int>> members;
    size_t group_base = 0;
    // K-search / K-eval are instantiated per (assignment words in HBM or LDS, words per partition): a launch group must be
    // homogeneous in both, whatever the LDS sizes say -- so a change of either ALWAYS starts a group; the
    // "at most 8 groups" cap only limits the splits by LDS footprint
    auto kind = [&](int t) { return (s->topic_global[(size_t)t] ? 1 : 0) | (s->pts[(size_t)t].d.nw > kRFP ? 2 : 0) | (s->topic_curg[(size_t)t] ? 4 : 0); };
    int lds_groups = 0, group_kind = -1;
    for (int t : order) {
        const bool new_kind = members.empty() || kind(t) != group_kind;
        if (new_kind || (need1(t) > 2 * group_base && lds_groups < 8)) {
            members.emplace_back(); group_base = need1(t); group_kind = kind(t);
            if (new_kind) lds_groups = 1; else ++lds_groups;
        }
        members.back().push_back(t);
    }
    std::vector<int2> smap; std::vector<int4> emap;
    for (const std::vector<int> &mem : members) {
        kao_session::LaunchGroup g;
        // candidates (one best snapshot per restart) per K-eval workgroup: 32 when the group fills the device; fewer when it does
        // not (a 30,000-partition topic has 256 restarts: at 32 per workgroup that was EIGHT workgroups, 2.9 ms per step --
        // a third of the step); few large candidates get one workgroup each, its wavefronts cooperating (k_eval<NE, true>)
        int64_t n_cand = 0;
        for (int t : mem) n_cand += s->pts[(size_t)t].d.n_restarts;
        const int64_t fill = 4 * (int64_t)std::max(g_num_cu, 1);
        int cpb = 32;
        if (n_cand < fill * 32) cpb = (int)std::min<int64_t>(32, std::max<int64_t>(kWaves, ((n_cand + fill - 1) / fill + kWaves - 1) / kWaves * kWaves));
        for (int t : mem) {
            const TopicDev &d = s->pts[(size_t)t].d;
            g.rf_uniform = g.rf_uniform < 0 ? d.RF : (g.rf_uniform == d.RF ? d.RF : 0);
            g.maxP = std::max(g.maxP, d.P); g.maxBx = std::max(g.maxBx, d.Bx); g.maxB = std::max(g.maxB, d.B); g.maxR = std::max(g.maxR, d.R); g.wide = g.wide || (int64_t)d.P * d.RF >= 512;
        }
        g.global_a = s->topic_global[(size_t)mem[0]] != 0;
        g.cur_global = s->topic_curg[(size_t)mem[0]] != 0;
        g.nw = s->pts[(size_t)mem[0]].d.nw;
        g.eval_coop = n_cand <= fill && g.maxP >= 1024;
        if (g.eval_coop) cpb = 1;
        if (g.global_a) if (const char *e = std::getenv("KAO_GLOBAL_WAVES")) g.waves = std::min(kWaves, std::max(1, std::atoi(e)));  // measurement hook
        // Topics that live in global memory: a restart is latency-bound (dependent loads of 16-byte assignment words), so ONE
        // wavefront per restart leaves the restart shallow.  Round 4 tried a TEAM of wavefronts per restart (k_team): W proposals
        // per iteration against the same state, the disjoint ones applied.  Measured (gpurun_out/r04_c2_*): same iterations per
        // second per restart at team 2 / 4, fewer at 8, and the 3-s incumbents of 1000 x 30000 / 1000 x 100000 within noise of
        // one wavefront per restart -- so teams are opt-in (kao_opts.team = n, or KAO_TEAM=n), the default is k_search.
        if (g.global_a) {
            const int tmax = g.nw > kRFP ? kTeamMax / 2 : kTeamMax;   // (8 replica words per partition: 256 threads keep the kernel out of scratch)
            int want = o.team == 0 ? 1 : o.team;
            if (const char *e = std::getenv("KAO_TEAM")) want = std::max(0, std::atoi(e));
            g.team = want <= 1 ? 0 : std::min(want, tmax);
            if (g.team > 0) g.waves = 1;   // the block map holds one workgroup per restart
        }
        while (g.waves > 1 && search_lds_bytes(g.maxP, g.maxBx, g.waves, g.global_a, true, g.nw, s->any_bw, g.maxR, g.team, g.cur_global) > 160 * 1024) g.waves /= 2;
        g.cur_in_lds = eval_lds_bytes(g.maxP, g.maxB, true, g.nw) <= 160 * 1024;
        if (search_lds_bytes(g.maxP, g.maxBx, g.waves, g.global_a, true, g.nw, s->any_bw, g.maxR, g.team, g.cur_global) > 160 * 1024 || eval_lds_bytes(g.maxP, g.maxB, g.cur_in_lds, g.nw) > 160 * 1024) {
            kao_session_destroy(s);
            return fail(KAO_ERR_UNSUPPORTED, "broker tables exceed 160 KiB of LDS (about 30,000 padded brokers)");
        }
        std::vector<int2> gs; std::vector<int> gs_topic;
        std::vector<int4> ge; std::vector<int> ge_topic;
        for (int t : mem) {
            const TopicDev &d = s->pts[(size_t)t].d;
            for (int r = 0; r < d.n_restarts; r += g.waves) { gs.push_back(make_int2(t, r)); gs_topic.push_back(t); }
            for (int r = 0; r < d.n_restarts; r += cpb) {
                ge.push_back(make_int4(t, r, std::min(cpb, d.n_restarts - r), d.restart_base + r));
                ge_topic.push_back(t);
            }
        }
        gs = xcd_order(gs, gs_topic);
        ge = xcd_order(ge, ge_topic);
        g.smap_off = (int)smap.size(); g.smap_n = (int)gs.size();
        g.emap_off = (int)emap.size(); g.emap_n = (int)ge.size();
        smap.insert(smap.end(), gs.begin(), gs.end());
        emap.insert(emap.end(), ge.begin(), ge.end());
        s->groups.push_back(g);
    }
    s->blocks_search = (int)smap.size();
    s->blocks_eval = (int)emap.size();
    std::vector<TopicDev> tds;
    for (auto &pt : s->pts) tds.push_back(pt.d);

    // ---- read-only arena: stage everything on the host, ONE hipMalloc (or a parked arena), ONE H2D copy ----
    struct Sec { const void *src; size_t bytes; size_t off; };
    Sec secs[11] = {{tds.data(), tds.size() * sizeof(TopicDev), 0}, {smap.data(), smap.size() * sizeof(int2), 0},
                   {emap.data(), emap.size() * sizeof(int4), 0}, {cur_pool.data(), cur_pool.size() * sizeof(uint32_t), 0},
                   {ext_pool.data(), ext_pool.size() * 2, 0}, {rsz_pool.data(), rsz_pool.size() * 4, 0},
                   {rackof_pool.data(), rackof_pool.size(), 0}, {curd_pool.data(), curd_pool.size() * 2, 0},
                   {int_pool.data(), int_pool.size() * 2, 0}, {bw_pool.data(), bw_pool.size() * 4, 0},
                   {bwd_pool.data(), bwd_pool.size() * 4, 0}};
    size_t ro_bytes = 0;
    for (Sec &sec : secs) { sec.off = ro_bytes; ro_bytes += align_up(sec.bytes); }
    std::vector<unsigned char> stage(ro_bytes);
    for (const Sec &sec : secs) if (sec.bytes) std::memcpy(stage.data() + sec.off, sec.src, sec.bytes);
    if ((rc = arena_get(ro_bytes, &s->arena_ro, &s->arena_ro_bytes))) { kao_session_destroy(s); return rc; }
    unsigned char *ro = static_cast<unsigned char *>(s->arena_ro);
    s->d_topics = reinterpret_cast<TopicDev *>(ro + secs[0].off);
    s->d_smap = reinterpret_cast<int2 *>(ro + secs[1].off);
    s->d_emap = reinterpret_cast<int4 *>(ro + secs[2].off);
    s->d_cur = reinterpret_cast<uint32_t *>(ro + secs[3].off);
    s->d_ext = reinterpret_cast<uint16_t *>(ro + secs[4].off);
    s->d_rsz = reinterpret_cast<int32_t *>(ro + secs[5].off);
    s->d_rackof = reinterpret_cast<uint8_t *>(ro + secs[6].off);
    s->d_curd = reinterpret_cast<uint16_t *>(ro + secs[7].off);
    s->d_int = reinterpret_cast<uint16_t *>(ro + secs[8].off);
    s->d_bw = reinterpret_cast<uint32_t *>(ro + secs[9].off);
    s->d_bwd = reinterpret_cast<uint32_t *>(ro + secs[10].off);

    // ---- mutable arena ----
    const size_t state_b = align_up(state_bytes), best_b = align_up(best_u16 * 2);
    const size_t info_b = align_up((size_t)s->total_restarts * 16), obj_b = align_up((size_t)s->total_restarts * 4);
    const size_t viol_b = align_up((size_t)s->total_restarts * 32);
    s->rb_viol_off = align_up((size_t)n_topics * 8 + 16, 16);
    s->rb_assign_off = s->rb_viol_off + (size_t)n_topics * 32;
    s->readback_bytes = s->rb_assign_off + win_u16 * 2;
    // behind the per-topic blocks: control blocks (8 x int64 per topic) and the workgroup map of the sliced K-bound
    dual_i32 = (dual_i32 + 1) & ~(uint64_t)1;
    s->wide_ctl_i32 = dual_i32; dual_i32 += 32 * (uint64_t)n_topics;
    s->wide_map_i32 = dual_i32; dual_i32 += 2 * wide_slices;
    const size_t dual_b = align_up(dual_i32 * 4), dtarget_b = align_up((size_t)n_topics * 8), dids_b = align_up((size_t)n_topics * 4);
    s->dual_rb_bytes = (size_t)n_topics * 24;
    s->price_half_i32 = align_up(price_i32 * 4) / 4;
    const size_t price_b = 2 * s->price_half_i32 * 4;
    const size_t rw_bytes = state_b + best_b + info_b + obj_b + viol_b + align_up(s->readback_bytes) + dual_b + dtarget_b + dids_b +
                            align_up(s->dual_rb_bytes) + price_b + align_up((size_t)n_topics * 8);
    if ((rc = arena_get(rw_bytes, &s->arena_rw, &s->arena_rw_bytes))) { kao_session_destroy(s); return rc; }
    unsigned char *rw = static_cast<unsigned char *>(s->arena_rw);
    s->d_state = rw;
    s->d_best = reinterpret_cast<uint16_t *>(rw + state_b);
    s->d_info = reinterpret_cast<int32_t *>(rw + state_b + best_b);
    s->d_obj = reinterpret_cast<int32_t *>(rw + state_b + best_b + info_b);
    s->d_viol = reinterpret_cast<int32_t *>(rw + state_b + best_b + info_b + obj_b);
    s->d_readback = rw + state_b + best_b + info_b + obj_b + viol_b;
    s->d_keys = reinterpret_cast<unsigned long long *>(s->d_readback);
    s->d_drift = reinterpret_cast<int32_t *>(s->d_readback + (size_t)n_topics * 8);
    s->d_win_viol = reinterpret_cast<int32_t *>(s->d_readback + s->rb_viol_off);
    s->d_win_assign = reinterpret_cast<uint16_t *>(s->d_readback + s->rb_assign_off);
    s->h_readback.assign(s->readback_bytes, 0);
    {
        unsigned char *q = s->d_readback + align_up(s->readback_bytes);
        s->d_dual = reinterpret_cast<int32_t *>(q); q += dual_b;
        s->d_dual_target = reinterpret_cast<long long *>(q); q += dtarget_b;
        s->d_dual_ids = reinterpret_cast<int32_t *>(q); q += dids_b;
        s->d_dual_rb = q; q += align_up(s->dual_rb_bytes);
        s->d_price = reinterpret_cast<int32_t *>(q); q += price_b;
        s->d_keys_glob = reinterpret_cast<unsigned long long *>(q);
        s->dual_bytes = dual_b;
        s->dual_flags.assign((size_t)n_topics, 0);
        s->dual_iters.assign((size_t)n_topics, 0);
        for (int t = 0; t < n_topics; ++t) if (!s->dual_ok[(size_t)t]) s->dual_flags[(size_t)t] = 8;
    }

    if ((rc = stream_get(&s->stream))) { kao_session_destroy(s); return rc; }
    // restart states / info / obj / viol are fully written by launch 0 (init) and the first K-eval; only the
    // snapshots ("no snapshot" = all KAO_NONE), the keys (all ones) and the drift counter need initial values
    hipError_t e1 = hipMemcpyAsync(ro, stage.data(), ro_bytes, hipMemcpyHostToDevice, s->stream);
    s->best_bytes = best_u16 * 2 ? best_u16 * 2 : 2;
    hipError_t e2 = hipMemsetAsync(s->d_best, 0xFF, s->best_bytes, s->stream);
    hipError_t e3 = hipMemsetAsync(s->d_readback, 0xFF, (size_t)n_topics * 8, s->stream);
    hipError_t e4 = hipMemsetAsync(s->d_drift, 0, 16, s->stream);
    if (e4 == hipSuccess) e4 = hipMemsetAsync(s->d_price, 0, price_b ? price_b : 4, s->stream);  // no prices yet
    hipError_t e5 = hipStreamSynchronize(s->stream);  // `stage` is pageable host memory and goes out of scope
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess || e5 != hipSuccess) {
        kao_session_destroy(s);
        return fail(KAO_ERR_HIP, "session upload failed");
    }
    if (o.profile) {
        s->ev.resize(kEvRing * 3);
        for (auto &e : s->ev) if (hipEventCreate(&e) != hipSuccess) { kao_session_destroy(s); return fail(KAO_ERR_HIP, "hipEventCreate"); }
    }
    *out = s;
    return KAO_OK;
}

int kao_session_step(kao_session *s) {
    if (!s) return fail(KAO_ERR_INVALID, "null session");
    HIP_TRY(hipSetDevice(s->device));
    const bool prof = s->opts.profile != 0;
    if (prof && s->ev_pending == kEvRing) {
        int rc = session_drain_events(s);
        if (rc) return rc;
    }
    SearchPools sp{};
    sp.topics = s->d_topics; sp.block_map = s->d_smap; sp.cur_pool = s->d_cur; sp.ext_pool = s->d_ext; sp.rsz_pool = s->d_rsz;
    sp.state_pool = s->d_state; sp.best_pool = s->d_best; sp.restart_info = s->d_info; sp.drift = s->d_drift;
    SearchParams prm{};
    prm.obj_scale = s->opts.obj_scale; prm.lam_min = s->opts.lam_min; prm.lam_max = s->opts.lam_max;
    prm.launch = s->launch; prm.iters = (uint32_t)s->opts.iters_per_launch; prm.init = (s->launch == 0 || s->reinit) ? 1 : 0;
    prm.gen = s->gen;
    s->reinit = false;
    const int eper = s->opts.elite_period;
    prm.bw = s->any_bw ? 1 : 0;
    prm.scan2_max = kScanTwoSlots;
    if (const char *e = std::getenv("KAO_X_SCAN2_MAX")) prm.scan2_max = std::atoi(e);   // experiment knob (the replay tests assume the default)
    prm.elite = (eper > 0 && s->launch > 0 && s->launch % (uint32_t)eper == 0) ? 1 : 0;
    sp.price_pool = s->d_price + (size_t)s->price_read * s->price_half_i32;
    sp.int_pool = s->d_int; sp.elite_assign = s->d_win_assign; sp.elite_key = s->d_keys; sp.bw_pool = s->d_bw;
    EvalPools ep{};
    ep.topics = s->d_topics; ep.rackof_pool = s->d_rackof; ep.curd_pool = s->d_curd;
    ep.cand = s->d_best; ep.objective = s->d_obj; ep.violations = s->d_viol; ep.best_key = s->d_keys; ep.bwd_pool = s->d_bwd;
    hipEvent_t *e = prof ? &s->ev[(size_t)s->ev_pending * 3] : nullptr;
    if (prof) HIP_TRY(hipEventRecord(e[0], s->stream));
    for (const kao_session::LaunchGroup &g : s->groups) {
        sp.block_map = s->d_smap + g.smap_off;
        prm.maxP = g.maxP; prm.maxBx = g.maxBx; prm.maxR = g.maxR; prm.wide = g.wide ? 1 : 0; prm.cur_global = g.cur_global ? 1 : 0;
        SearchParams gp = prm;
        if (gp.init && g.global_a) {   // topics in global memory: the holes are filled by a workgroup per restart (K-init), not by one wavefront
            const char *e = std::getenv("KAO_INIT_WAVES");
            if (!(e && e[0] == '0') && launch_init(sp, gp, g.smap_n, g.team > 0 ? 1 : g.waves, s->priced, g.nw, s->stream)) gp.init = 2;
            HIP_TRY(hipGetLastError());
        }
        launch_search(sp, gp, g.smap_n, g.waves, g.global_a, s->priced, g.nw, s->stream, g.team);
        HIP_TRY(hipGetLastError());
    }
    if (prof) HIP_TRY(hipEventRecord(e[1], s->stream));
    for (const kao_session::LaunchGroup &g : s->groups) {
        ep.block_map = s->d_emap + g.emap_off;
        ep.maxP = g.maxP; ep.maxB = g.maxB; ep.cur_in_lds = g.cur_in_lds ? 1 : 0; ep.coop = g.eval_coop ? 1 : 0; ep.rf_uniform = std::max(g.rf_uniform, 0);
        launch_eval(ep, g.emap_n, g.nw, s->stream);
        HIP_TRY(hipGetLastError());
    }
    if (prof) { HIP_TRY(hipEventRecord(e[2], s->stream)); s->ev_pending++; }
    if (eper > 0 && (s->launch + 1) % (uint32_t)eper == 0) {  // the next launch is an elite launch: stage every topic's best assignment
        launch_gather(s->d_topics, s->n_topics, s->d_keys, s->d_best, s->d_viol, s->d_win_assign, s->d_win_viol, s->stream);
        HIP_TRY(hipGetLastError());
    }
    for (const PreparedTopic &pt : s->pts) {
        const uint64_t n = neighbours_in_range(prm.launch * prm.iters, prm.iters, pt.d.RF, pt.d.B, pt.d.P, prm.scan2_max) * (uint64_t)pt.d.n_restarts;
        s->delta_total += n;
        s->search_bytes_total += n * (uint64_t)(8 * pt.d.RF + 10);
    }
    s->launch++;
    return KAO_OK;
}

// A new generation of the population: the next step re-initialises every restart from the current assignment (best
// insertion, tie-break hash salted with the generation number), the snapshots and packed best keys of the old generation are
// dropped -- the caller has read what it wants to keep (kao_solve keeps the incumbent on the host).  K-bound state, prices
// and the launch / iteration counters carry on.
int kao_session_new_generation(kao_session *s) {
    if (!s) return fail(KAO_ERR_INVALID, "null session");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemsetAsync(s->d_best, 0xFF, s->best_bytes, s->stream));
    HIP_TRY(hipMemsetAsync(s->d_readback, 0xFF, (size_t)s->n_topics * 8, s->stream));   // packed best keys: none
    s->gen++;
    s->reinit = true;
    return KAO_OK;
}

int kao_session_sync(kao_session *s) {
    if (!s) return fail(KAO_ERR_INVALID, "null session");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->opts.profile) return session_drain_events(s);
    return KAO_OK;
}

int kao_session_best_keys(kao_session *s, uint64_t *keys) {
    if (!s || !keys) return fail(KAO_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemcpyAsync(keys, s->d_keys, (size_t)s->n_topics * 8, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return KAO_OK;
}

int kao_session_device_keys(kao_session *s, void **d_keys) {
    if (!s || !d_keys) return fail(KAO_ERR_INVALID, "null argument");
    *d_keys = s->d_keys;
    return KAO_OK;
}

int kao_session_best(kao_session *s, kao_result *results) {
    if (!s || !results) return fail(KAO_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(s->device));
    launch_gather(s->d_topics, s->n_topics, s->d_keys, s->d_best, s->d_viol, s->d_win_assign, s->d_win_viol, s->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(s->h_readback.data(), s->d_readback, s->readback_bytes, hipMemcpyDeviceToHost, s->stream));
    int rc = kao_session_sync(s);
    if (rc) return rc;
    const uint64_t *keys = reinterpret_cast<const uint64_t *>(s->h_readback.data());
    const int32_t *wv = reinterpret_cast<const int32_t *>(s->h_readback.data() + s->rb_viol_off);
    const uint16_t *wa = reinterpret_cast<const uint16_t *>(s->h_readback.data() + s->rb_assign_off);
    for (int t = 0; t < s->n_topics; ++t) {
        const TopicDev &d = s->pts[(size_t)t].d;
        kao_result &r = results[t];
        r.upper_bound = s->ub[(size_t)t];
        const uint64_t key = keys[t];
        if (key == ~0ull) {  // no step has run yet
            r.status = s->topic_infeasible[(size_t)t] ? KAO_STATUS_INFEASIBLE_PROVEN : KAO_STATUS_NO_FEASIBLE; r.best_restart = -1; r.objective = -1;
            std::memset(r.violations, 0, sizeof r.violations);
            continue;
        }
        r.best_restart = (key & 0xFFFFF) == kExternalRestart ? -1 : (int)(key & 0xFFFFF);  // -1: adopted from another GPU
        r.objective = (int64_t)kObjCap - (int64_t)((key >> 20) & 0xFFFFFF);
        std::memcpy(r.violations, wv + (size_t)t * 8, 32);
        if (r.assignment) std::memcpy(r.assignment, wa + d.win_off, (size_t)d.P * d.RF * 2);
        if (r.violations[0] != 0) { r.status = s->topic_infeasible[(size_t)t] ? KAO_STATUS_INFEASIBLE_PROVEN : KAO_STATUS_NO_FEASIBLE; r.objective = -1; }
        else r.status = r.objective >= r.upper_bound ? KAO_STATUS_OPTIMAL_PROVEN : KAO_STATUS_FEASIBLE_BOUND_GAP;
    }
    return KAO_OK;
}

int kao_session_stats(kao_session *s, kao_stats *out) {
    if (!s || !out) return fail(KAO_ERR_INVALID, "null argument");
    int rc = kao_session_sync(s);
    if (rc) return rc;
    std::memset(out, 0, sizeof *out);
    out->launches = s->launch;
    out->delta_candidates = s->delta_total;
    out->full_candidates = (uint64_t)s->launch * (uint64_t)s->total_restarts;
    out->ms_search = s->ms_search; out->ms_eval = s->ms_eval;
    out->search_bytes_algo = s->search_bytes_total;
    out->eval_bytes_algo = s->eval_bytes_per_launch * s->launch;
    out->n_restarts_total = s->total_restarts;
    for (const kao_session::LaunchGroup &g : s->groups)
        out->lds_bytes_search = std::max(out->lds_bytes_search, (int32_t)search_lds_bytes(g.maxP, g.maxBx, g.waves, g.global_a, s->priced, g.nw, s->any_bw, g.maxR, g.team, g.cur_global));
    out->launch_groups = (int32_t)s->groups.size();
    out->blocks_search = s->blocks_search;
    HIP_TRY(hipMemcpy(&out->drift, s->d_drift, 4, hipMemcpyDeviceToHost));
    return KAO_OK;
}

static int bound_step_impl(kao_session *s, const int64_t *target, int32_t iters, bool force_step);
int kao_session_bound_step(kao_session *s, const int64_t *target, int32_t iters) { return bound_step_impl(s, target, iters, false); }

// force_step: the launch repeats one that k_bound_multi gave up on (flag 16): the one-iteration-per-launch kernels, from the same state
static int bound_step_impl(kao_session *s, const int64_t *target, int32_t iters, bool force_step) {
    if (!s || !target) return fail(KAO_ERR_INVALID, "null argument");
    if (iters < 1) return fail(KAO_ERR_INVALID, "iters < 1");
    HIP_TRY(hipSetDevice(s->device));
    // the previous launch's H2D copies read the staging vectors below: wait for them before rewriting
    if (s->stream_bound) HIP_TRY(hipStreamSynchronize(s->stream_bound));
    s->h_dual_ids.clear();
    s->h_dual_target.assign((size_t)s->n_topics, -1);
    // two classes of topics, one launch each: RF and current RF <= 4 (k_bound<4>), RF 5..8 (k_bound<8>); ids of the first class first
    auto wide_slots = [&](int t) { return s->pts[(size_t)t].d.RF > kRFP || s->pts[(size_t)t].d.rf_cur > kRFP; };
    int n_class[2] = {0, 0};
    for (int cls = 0; cls < 2; ++cls)
        for (int t = 0; t < s->n_topics; ++t) {
            if (target[t] < 0 || !s->dual_ok[(size_t)t] || s->topic_infeasible[(size_t)t] || (int)wide_slots(t) != cls) continue;
            if (target[t] > (int64_t)1 << 40) return fail(KAO_ERR_INVALID, "target out of range");
            s->h_dual_ids.push_back(t);
            s->h_dual_target[(size_t)t] = target[t];
            n_class[cls]++;
        }
    if (s->h_dual_ids.empty()) return KAO_OK;
    if (!s->stream_bound) {
        // highest priority: a K-bound launch is a handful of workgroups that should not queue behind a full K-search grid
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_TRY(hipStreamCreateWithPriority(&s->stream_bound, hipStreamNonBlocking, hi));
        HIP_TRY(hipEventCreate(&s->ev_bound0));
        HIP_TRY(hipEventCreate(&s->ev_bound1));
    }
    // one K-bound launch in flight at a time (it continues from the state the previous one left in HBM); the session
    // upload was synchronised at creation, K-search and K-bound share read-only tables only
    HIP_TRY(hipStreamSynchronize(s->stream_bound));
    if (!s->dual_state_init) {
        // K-bound state is initialised by the first launch only (most sessions never need K-bound): multipliers and
        // directions 0, best dual value "infinite" (0x7F7F...), info 0
        HIP_TRY(hipMemsetAsync(s->d_dual, 0, s->dual_bytes, s->stream_bound));
        HIP_TRY(hipMemsetAsync(s->d_dual_rb, 0x7F, (size_t)s->n_topics * 8, s->stream_bound));
        HIP_TRY(hipMemsetAsync(s->d_dual_rb + (size_t)s->n_topics * 8, 0, (size_t)s->n_topics * 16, s->stream_bound));
        s->dual_state_init = true;
    }
    // pageable staging: hipMemcpyAsync returns once the host buffers have been consumed
    HIP_TRY(hipMemcpyAsync(s->d_dual_target, s->h_dual_target.data(), (size_t)s->n_topics * 8, hipMemcpyHostToDevice, s->stream_bound));
    HIP_TRY(hipMemcpyAsync(s->d_dual_ids, s->h_dual_ids.data(), s->h_dual_ids.size() * 4, hipMemcpyHostToDevice, s->stream_bound));
    BoundPools bp{};
    bp.topics = s->d_topics; bp.ids = s->d_dual_ids; bp.rackof_pool = s->d_rackof; bp.curd_pool = s->d_curd;
    bp.dual_pool = s->d_dual; bp.target = s->d_dual_target;
    bp.best_L = reinterpret_cast<long long *>(s->d_dual_rb);
    bp.info = reinterpret_cast<int32_t *>(s->d_dual_rb + (size_t)s->n_topics * 8);
    bp.ext_pool = s->d_ext; bp.rsz_pool = s->d_rsz;
    bp.iters = iters;
    bp.bwd_pool = s->any_bw ? s->d_bwd : nullptr;
    // K-search launches already enqueued may still read the half this launch is about to overwrite -- except under kao_solve's
    // deterministic schedule, which starts K-bound only when the enqueued K-search launches read the OTHER half (the launch
    // that read this one has been waited for)
    if (s->priced && !s->bound_no_wait) {
        if (!s->ev_search) HIP_TRY(hipEventCreateWithFlags(&s->ev_search, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(s->ev_search, s->stream));
        HIP_TRY(hipStreamWaitEvent(s->stream_bound, s->ev_search, 0));
    }
    // prices go into the half K-search is NOT reading; kao_session_adopt_prices flips the halves once this launch is done
    const int wh = s->price_read ^ 1;
    if (s->price_write_last >= 0 && s->price_write_last != wh)  // keep the prices of topics this launch does not cover
        HIP_TRY(hipMemcpyAsync(s->d_price + (size_t)wh * s->price_half_i32, s->d_price + (size_t)(wh ^ 1) * s->price_half_i32,
                               s->price_half_i32 * 4, hipMemcpyDeviceToDevice, s->stream_bound));
    bp.price_pool = s->d_price + (size_t)wh * s->price_half_i32;
    bp.export_prices = 1;
    if (const char *e = std::getenv("KAO_X_PRICE_SRC")) bp.export_prices = std::atoi(e);  // experiment knob
    s->price_write_last = wh;
    HIP_TRY(hipEventRecord(s->ev_bound0, s->stream_bound));
    s->h_wide_map.clear();
    for (int cls = 0, first = 0; cls < 2; first += n_class[cls], ++cls) {
        if (!n_class[cls]) continue;
        int maxB = 0, maxP = 0, maxR = 0;
        for (int i = first; i < first + n_class[cls]; ++i) {
            const TopicDev &d = s->pts[(size_t)s->h_dual_ids[(size_t)i]].d;
            maxB = std::max(maxB, d.B); maxP = std::max(maxP, d.P); maxR = std::max(maxR, d.R);
        }
        bp.ids = s->d_dual_ids + first;
        bp.ne = cls ? 8 : 4;
        bp.maxB = maxB; bp.maxP = maxP; bp.maxR = maxR;
        bp.cur_in_lds = bound_lds_bytes(maxB, maxP, maxR, true, bp.ne, bp.bwd_pool != nullptr) <= 160 * 1024 ? 1 : 0;
        // lanes own partitions, wavefronts own racks when the pools are rebuilt: enough wavefronts for either, at most 16
        const int waves = std::min(16, std::max({1, (maxP + 63) / 64, std::min(maxR, 8)}));
        // topics beyond a few thousand partitions: one iteration per launch, the partitions sliced over several workgroups
        // (k_bound_step); a launch that holds such a topic runs all its topics that way.  KAO_BOUND_CHUNK = partitions per
        // slice (test hook: small values slice small topics)
        int chunk = maxP > 2048 ? 512 : 0;
        // Round 3: the sliced topics run on the PERSISTENT multi-workgroup driver (k_bound_multi) -- and so do topics from 1,024
        // partitions up, in slices of 512 (a 2,000-partition topic: 52 us per iteration in k_bound's one workgroup).  Its workgroups
        // wait for each other, so a launch is kept to 96 of them (larger slices otherwise).  KAO_BOUND_MULTI=0: the round-2 drivers.
        const char *multi_env = std::getenv("KAO_BOUND_MULTI");
        const bool multi = !(multi_env && multi_env[0] == '0') && !s->multi_off;
        if (multi && chunk == 0 && maxP >= 1024) chunk = 512;
        if (const char *e = std::getenv("KAO_BOUND_CHUNK")) { chunk = std::max(0, std::atoi(e)) / 64 * 64; }
        if (multi && chunk > 0) {   // workgroups that wait for others (topics of more than one slice): at most 96 per launch
            auto waiting_at = [&](int c) {
                int64_t nb = 0;
                for (int i = first; i < first + n_class[cls]; ++i) { const int n = (s->pts[(size_t)s->h_dual_ids[(size_t)i]].d.P + c - 1) / c; nb += n > 1 ? n : 0; }
                return nb;
            };
            while (chunk < (1 << 20) && waiting_at(chunk) > 96) chunk += chunk;
        }
        launch_bound_center(bp, n_class[cls], s->stream_bound);   // exact line search along the common shift of every family
        if (chunk > 0) {
            const size_t map0 = s->h_wide_map.size();
            for (int i = first; i < first + n_class[cls]; ++i) {
                const int t = s->h_dual_ids[(size_t)i];
                for (int sl = 0, n = (s->pts[(size_t)t].d.P + chunk - 1) / chunk; sl < n; ++sl) s->h_wide_map.push_back(make_int2(t, sl));
            }
            BoundWide wd{};
            wd.map = reinterpret_cast<const int2 *>(s->d_dual + s->wide_map_i32) + map0;
            wd.cnt_pool = s->d_dual;
            wd.ctl = reinterpret_cast<long long *>(s->d_dual + s->wide_ctl_i32);
            wd.chunk = chunk;
            HIP_TRY(hipMemcpyAsync(s->d_dual + s->wide_map_i32 + 2 * map0, s->h_wide_map.data() + map0, (s->h_wide_map.size() - map0) * sizeof(int2),
                                   hipMemcpyHostToDevice, s->stream_bound));
            // 16 wavefronts whatever the slice: the O(B) phases every workgroup repeats (pools, totals, band terms, step) are what
            // an iteration waits for (measured: slices of 256 with 4 wavefronts 32 us, slices of 512 with 8 wavefronts 19 us at 500 x 5,000)
            int multi_waves = 16;
            if (const char *e = std::getenv("KAO_BOUND_WAVES")) multi_waves = std::min(16, std::max(1, std::atoi(e)));
            if (!(multi && !force_step && launch_bound_multi(bp, wd, n_class[cls], (int)(s->h_wide_map.size() - map0), multi_waves, s->stream_bound)))
                launch_bound_wide(bp, wd, n_class[cls], (int)(s->h_wide_map.size() - map0), 16, s->stream_bound);
        } else
            launch_bound(bp, n_class[cls], waves, s->stream_bound);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->ev_bound1, s->stream_bound));
    s->bound_inflight = true;
    s->bound_iters_last = iters;
    s->bound_launches++;
    return KAO_OK;
}

int kao_session_set_prices(kao_session *s, int32_t topic, const int32_t *a, const int32_t *l, const int32_t *g) {
    if (!s || topic < 0 || topic >= s->n_topics || !a || !l || !g) return fail(KAO_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(s->device));
    const TopicDev &d = s->pts[(size_t)topic].d;
    std::vector<int32_t> buf(2 * (size_t)d.B + kRackTab, 0);
    std::memcpy(buf.data(), a, (size_t)d.B * 4);
    std::memcpy(buf.data() + d.B, l, (size_t)d.B * 4);
    std::memcpy(buf.data() + 2 * (size_t)d.B, g, (size_t)d.R * 4);
    // both halves, so that a later adopt (which flips them) keeps host-set prices of topics K-bound does not cover
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->stream_bound) HIP_TRY(hipStreamSynchronize(s->stream_bound));
    for (int h = 0; h < 2; ++h)
        HIP_TRY(hipMemcpy(s->d_price + (size_t)h * s->price_half_i32 + d.price_off, buf.data(), buf.size() * 4, hipMemcpyHostToDevice));
    s->priced = true;
    return KAO_OK;
}

int kao_session_prices(kao_session *s, int32_t topic, int32_t *a, int32_t *l, int32_t *g) {
    if (!s || topic < 0 || topic >= s->n_topics) return fail(KAO_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(s->device));
    const TopicDev &d = s->pts[(size_t)topic].d;
    const int32_t *base = s->d_price + (size_t)s->price_read * s->price_half_i32 + d.price_off;
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (a) HIP_TRY(hipMemcpy(a, base, (size_t)d.B * 4, hipMemcpyDeviceToHost));
    if (l) HIP_TRY(hipMemcpy(l, base + d.B, (size_t)d.B * 4, hipMemcpyDeviceToHost));
    if (g) HIP_TRY(hipMemcpy(g, base + 2 * (size_t)d.B, (size_t)d.R * 4, hipMemcpyDeviceToHost));
    return KAO_OK;
}

int kao_session_adopt_prices(kao_session *s) {
    if (!s) return fail(KAO_ERR_INVALID, "null session");
    if (s->price_write_last < 0) return KAO_OK;  // K-bound has not run: nothing to adopt
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream_bound));
    s->price_read = s->price_write_last;
    s->priced = true;
    return KAO_OK;
}

int kao_session_bound_busy(kao_session *s) {
    if (!s) return fail(KAO_ERR_INVALID, "null session");
    HIP_TRY(hipSetDevice(s->device));
    if (!s->bound_inflight) return 0;
    const hipError_t e = hipEventQuery(s->ev_bound1);
    if (e == hipErrorNotReady) return 1;
    if (e != hipSuccess) return fail(KAO_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(e));
    return 0;
}

int kao_session_bounds(kao_session *s, int64_t *upper_bound, int32_t *flags, int32_t *iters) {
    if (!s) return fail(KAO_ERR_INVALID, "null session");
    HIP_TRY(hipSetDevice(s->device));
    if (s->bound_launches) {
        std::vector<unsigned char> rb(s->dual_rb_bytes);
        HIP_TRY(hipMemcpyAsync(rb.data(), s->d_dual_rb, s->dual_rb_bytes, hipMemcpyDeviceToHost, s->stream_bound));
        HIP_TRY(hipStreamSynchronize(s->stream_bound));
        if (s->bound_inflight) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, s->ev_bound0, s->ev_bound1) == hipSuccess) s->bound_ms_last = ms;
            s->bound_inflight = false;
        }
        const int64_t *best = reinterpret_cast<const int64_t *>(rb.data());
        const int32_t *info = reinterpret_cast<const int32_t *>(rb.data() + (size_t)s->n_topics * 8);
        bool gave_up = false;
        for (int t = 0; t < s->n_topics; ++t) gave_up |= s->dual_ok[(size_t)t] && (info[t * 4 + 1] & 16);
        if (gave_up && !s->multi_off) {
            // k_bound_multi could not get a topic's workgroups resident together and committed nothing: the same launch again on
            // the kernels that do not wait for each other, and no further use of the persistent driver in this session
            // Only the topics that carry flag 16 run again: the others of the launch have committed their iterations (the abort
            // mark is all-or-nothing per topic, kao_bound.hip).  All drivers share one arithmetic, so the repeated topics end in
            // the state the persistent driver would have reached: the answer does not depend on whether a launch gave up.
            s->multi_off = true;
            std::vector<int64_t> again = s->h_dual_target;
            for (int t = 0; t < s->n_topics; ++t)
                if (!(s->dual_ok[(size_t)t] && (info[t * 4 + 1] & 16))) again[(size_t)t] = -1;
            int rc = bound_step_impl(s, again.data(), s->bound_iters_last, true);
            if (rc) return rc;
            return kao_session_bounds(s, upper_bound, flags, iters);
        }
        for (int t = 0; t < s->n_topics; ++t) {
            if (!s->dual_ok[(size_t)t]) continue;
            s->dual_iters[(size_t)t] = info[t * 4 + 0];
            s->dual_flags[(size_t)t] = info[t * 4 + 1];
            if ((info[t * 4 + 1] & 4) || info[t * 4 + 0] == 0 || best[t] >= (int64_t)0x7F7F7F7F7F7F7F7Fll) continue;
            const int64_t b = best[t] >= 0 ? best[t] / kDualScale : -((-best[t] + kDualScale - 1) / kDualScale);  // floor
            s->ub[(size_t)t] = std::min(s->ub[(size_t)t], b);
        }
    }
    for (int t = 0; t < s->n_topics; ++t) {
        if (upper_bound) upper_bound[t] = s->ub[(size_t)t];
        if (flags) flags[t] = s->dual_flags[(size_t)t];
        if (iters) iters[t] = s->dual_iters[(size_t)t];
    }
    return KAO_OK;
}

int kao_session_dual_state(kao_session *s, int32_t topic, int32_t *a, int32_t *l, int32_t *g, int64_t *best_dual) {
    if (!s || topic < 0 || topic >= s->n_topics) return fail(KAO_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(s->device));
    const TopicDev &d = s->pts[(size_t)topic].d;
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->stream_bound) HIP_TRY(hipStreamSynchronize(s->stream_bound));
    const int32_t *base = s->d_dual + d.dual_off;
    if (!s->dual_state_init) {  // no K-bound launch yet: the initial state
        if (a) std::memset(a, 0, (size_t)d.B * 4);
        if (l) std::memset(l, 0, (size_t)d.B * 4);
        if (g) std::memset(g, 0, (size_t)d.R * 4);
        if (best_dual) *best_dual = (int64_t)0x7F7F7F7F7F7F7F7Fll;
        return KAO_OK;
    }
    if (a) HIP_TRY(hipMemcpy(a, base, (size_t)d.B * 4, hipMemcpyDeviceToHost));
    if (l) HIP_TRY(hipMemcpy(l, base + d.B, (size_t)d.B * 4, hipMemcpyDeviceToHost));
    if (g) HIP_TRY(hipMemcpy(g, base + 4 * (size_t)d.B, (size_t)d.R * 4, hipMemcpyDeviceToHost));
    if (best_dual) HIP_TRY(hipMemcpy(best_dual, s->d_dual_rb + (size_t)topic * 8, 8, hipMemcpyDeviceToHost));
    return KAO_OK;
}

int kao_session_set_dual_state(kao_session *s, int32_t topic, const int32_t *a, const int32_t *l, const int32_t *g) {
    if (!s || topic < 0 || topic >= s->n_topics || !a || !l || !g) return fail(KAO_ERR_INVALID, "bad argument");
    if (!s->dual_ok[(size_t)topic]) return fail(KAO_ERR_UNSUPPORTED, "topic outside K-bound's limits");
    HIP_TRY(hipSetDevice(s->device));
    if (!s->stream_bound) {
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_TRY(hipStreamCreateWithPriority(&s->stream_bound, hipStreamNonBlocking, hi));
        HIP_TRY(hipEventCreate(&s->ev_bound0));
        HIP_TRY(hipEventCreate(&s->ev_bound1));
    }
    HIP_TRY(hipStreamSynchronize(s->stream_bound));
    if (!s->dual_state_init) {
        HIP_TRY(hipMemsetAsync(s->d_dual, 0, s->dual_bytes, s->stream_bound));
        HIP_TRY(hipMemsetAsync(s->d_dual_rb, 0x7F, (size_t)s->n_topics * 8, s->stream_bound));
        HIP_TRY(hipMemsetAsync(s->d_dual_rb + (size_t)s->n_topics * 8, 0, (size_t)s->n_topics * 16, s->stream_bound));
        s->dual_state_init = true;
    }
    const TopicDev &d = s->pts[(size_t)topic].d;
    // dual_pool: a[B] l[B] da[B] dl[B] g[kRackTab] dg[kRackTab] lv[8] ...: iterate replaced, direction memory and level control cleared
    std::vector<int32_t> buf(4 * (size_t)d.B + 2 * kRackTab + 8, 0);
    auto clampm = [](int32_t v) { return std::max(-kDualClamp, std::min(kDualClamp, v)); };
    for (int b = 0; b < d.B; ++b) { buf[(size_t)b] = clampm(a[b]); buf[(size_t)d.B + b] = clampm(l[b]); }
    for (int r = 0; r < d.R; ++r) buf[4 * (size_t)d.B + r] = clampm(g[r]);
    HIP_TRY(hipMemcpyAsync(s->d_dual + d.dual_off, buf.data(), buf.size() * 4, hipMemcpyHostToDevice, s->stream_bound));
    HIP_TRY(hipStreamSynchronize(s->stream_bound));
    return KAO_OK;
}

int kao_lp_bound(const kao_topic *t, double tol, int32_t max_iters, int64_t *bound, int64_t *best_dual, int32_t *multipliers, double stats[8]) {
    if (!t) return fail(KAO_ERR_INVALID, "null topic");
    int rc = require_init();
    if (rc) return rc;
    LpCtx *lp = nullptr;
    if ((rc = lp_open(t, &lp))) return rc;
    std::vector<int32_t> mult(2 * (size_t)t->n_brokers + (size_t)t->n_racks);
    rc = lp_solve(lp, tol > 0 ? tol : 1e-7, max_iters > 0 ? max_iters : 80, mult.data(), stats, nullptr);
    lp_close(lp);
    if (rc) return rc;
    if (multipliers) std::memcpy(multipliers, mult.data(), mult.size() * 4);
    // the dual value at those multipliers, in integers: one K-bound iteration from them
    kao_opts o{};
    o.restarts = kWaves;
    kao_session *s = nullptr;
    if ((rc = kao_session_create(t, 1, &o, &s))) return rc;
    if (!s->dual_ok[0]) { kao_session_destroy(s); return fail(KAO_ERR_UNSUPPORTED, "topic outside K-bound's limits"); }
    rc = kao_session_set_dual_state(s, 0, mult.data(), mult.data() + t->n_brokers, mult.data() + 2 * (size_t)t->n_brokers);
    const int64_t target = 0;
    int32_t fl = 0, itn = 0;
    if (!rc) rc = kao_session_bound_step(s, &target, 1);
    if (!rc) rc = kao_session_bounds(s, nullptr, &fl, &itn);
    int64_t bd = 0;
    if (!rc) rc = kao_session_dual_state(s, 0, nullptr, nullptr, nullptr, &bd);
    if (!rc) {
        if (best_dual) *best_dual = bd;
        if (bound) *bound = (fl & 4) || itn == 0 ? INT64_MAX : (bd >= 0 ? bd / kDualScale : -((-bd + kDualScale - 1) / kDualScale));
    }
    kao_session_destroy(s);
    return rc;
}

int kao_lp_round(const kao_topic *t, double pert, uint32_t salt, double tol, int32_t max_iters, int32_t use_fallback, uint16_t *assignment,
                 int64_t *objective, int32_t violations[8], double stats[8]) {
    if (!t || !assignment) return fail(KAO_ERR_INVALID, "null topic / assignment");
    int rc = require_init();
    if (rc) return rc;
    LpCtx *lp = nullptr;
    if ((rc = lp_open(t, &lp))) return rc;
    const size_t slots = (size_t)t->n_partitions * t->rf;
    const double eps = pert > 0 ? pert : lp_default_pert(t);
    double st[8] = {0};
    rc = lp_solve(lp, tol > 0 ? tol : 1e-8, max_iters > 0 ? max_iters : 150, nullptr, st, nullptr, eps, salt);
    std::vector<uint8_t> q((size_t)(2 * t->rf_cur + 2 * t->n_racks) * t->n_partitions);
    std::vector<int32_t> zq(2 * (size_t)t->n_brokers);
    if (!rc) rc = lp_primal(lp, q.data(), zq.data());
    lp_close(lp);
    if (rc) return rc;
    const double t0 = now_s();
    std::vector<uint16_t> fb;
    if (use_fallback) fb.assign(assignment, assignment + slots);
    int32_t rep[4] = {0, 0, 0, 0};
    if ((rc = lp_round_assignment(t, q.data(), zq.data(), use_fallback ? fb.data() : nullptr, assignment, rep))) return rc;
    const double t1 = now_s();
    int64_t obj = 0;
    int32_t viol[8] = {0};
    if ((rc = kao_evaluate(t, assignment, &obj, viol))) return rc;
    if (objective) *objective = obj;
    if (violations) std::memcpy(violations, viol, sizeof viol);
    if (stats) { stats[0] = st[0]; stats[1] = st[3]; stats[2] = rep[0]; stats[3] = rep[1] + rep[2]; stats[4] = rep[3]; stats[5] = st[7]; stats[6] = (t1 - t0) * 1e3; stats[7] = eps; }
    return KAO_OK;
}

int kao_lp_round_host(const kao_topic *t, const uint8_t *q, const int32_t *zq, int32_t use_fallback, uint16_t *assignment, int32_t rep[4]) {
    if (!t || !assignment || (use_fallback != 2 && (!q || !zq))) return fail(KAO_ERR_INVALID, "null argument");
    int rc = validate(t);
    if (rc) return rc;
    if (use_fallback == 2) {   // the band repair alone on the assignment passed in
        if (rep) rep[0] = rep[1] = rep[2] = rep[3] = 0;
        for (size_t i = 0, n = (size_t)t->n_partitions * t->rf; i < n; ++i)
            if (assignment[i] >= t->n_brokers) return fail(KAO_ERR_INVALID, "repair: a complete assignment expected");
        return lp_round_assignment(t, nullptr, zq, nullptr, assignment, rep);
    }
    std::vector<uint16_t> fb;
    if (use_fallback) fb.assign(assignment, assignment + (size_t)t->n_partitions * t->rf);
    return lp_round_assignment(t, q, zq, use_fallback ? fb.data() : nullptr, assignment, rep);
}

int kao_lp_trace(const kao_topic *t, double tol, int32_t max_iters, double *trace, double stats[8], int32_t *multipliers) {
    if (!t) return fail(KAO_ERR_INVALID, "null topic");
    int rc = require_init();
    if (rc) return rc;
    LpCtx *lp = nullptr;
    if ((rc = lp_open(t, &lp))) return rc;
    double pert = 0.0;       // experiment hook KAO_LP_TRACE_PERT=<eps> (-1: the solve's own default): the trace of the PERTURBED solve
    if (const char *e = std::getenv("KAO_LP_TRACE_PERT")) { pert = std::atof(e); if (pert < 0) pert = std::min(1e-4, 1.5 / ((double)t->n_partitions * t->rf)); }
    rc = lp_solve(lp, tol > 0 ? tol : 1e-7, max_iters > 0 ? max_iters : 80, multipliers, stats, trace, pert, 0);
    lp_close(lp);
    return rc;
}

int kao_dual_bound(const kao_topic *t, int64_t target, int32_t iters, int32_t launches, int64_t *bound, int64_t *best_dual,
                   int32_t *iters_done, int32_t *flags, int32_t *multipliers) {
    if (!t) return fail(KAO_ERR_INVALID, "null topic");
    if (target < 0 || iters < 1 || launches < 1) return fail(KAO_ERR_INVALID, "bad target / iters / launches");
    kao_opts o{};
    o.restarts = kWaves;  // no search is run: the smallest session there is
    kao_session *s = nullptr;
    int rc = kao_session_create(t, 1, &o, &s);
    if (rc) return rc;
    if (!s->dual_ok[0]) { kao_session_destroy(s); return fail(KAO_ERR_UNSUPPORTED, "topic outside K-bound's limits"); }
    int32_t fl = 0, itn = 0;
    for (int i = 0; i < launches && !rc; ++i) {
        rc = kao_session_bound_step(s, &target, iters);
        if (!rc) rc = kao_session_bounds(s, nullptr, &fl, &itn);
        if (fl & 7) break;
    }
    int64_t bd = 0;
    if (!rc) rc = kao_session_dual_state(s, 0, multipliers, multipliers ? multipliers + t->n_brokers : nullptr,
                                         multipliers ? multipliers + 2 * (size_t)t->n_brokers : nullptr, &bd);
    if (!rc) {
        if (best_dual) *best_dual = bd;
        if (bound) *bound = (fl & 4) || itn == 0 ? INT64_MAX : (bd >= 0 ? bd / kDualScale : -((-bd + kDualScale - 1) / kDualScale));
        if (iters_done) *iters_done = itn;
        if (flags) *flags = fl;
    }
    kao_session_destroy(s);
    return rc;
}

int kao_session_restart_state(kao_session *s, int32_t topic, int32_t restart, uint16_t *final_state, uint16_t *best_state,
                              int32_t info[4]) {
    if (!s || topic < 0 || topic >= s->n_topics) return fail(KAO_ERR_INVALID, "bad topic");
    const PreparedTopic &pt = s->pts[(size_t)topic];
    const TopicDev &d = pt.d;
    if (restart < 0 || restart >= d.n_restarts) return fail(KAO_ERR_INVALID, "bad restart");
    HIP_TRY(hipSetDevice(s->device));
    int rc = kao_session_sync(s);
    if (rc) return rc;
    if (final_state) {
        const bool ga = s->topic_global[(size_t)topic] != 0;
        const int nw = d.nw;
        std::vector<uint16_t> raw((size_t)d.P * nw * (ga ? 2 : 1));   // global path: nw words per partition; LDS path: nw x u16
        HIP_TRY(hipMemcpy(raw.data(), s->d_state + d.state_off + (uint64_t)restart * d.P * nw * (ga ? 4 : 2), raw.size() * 2, hipMemcpyDeviceToHost));
        for (int p = 0; p < d.P; ++p)
            for (int k = 0; k < d.RF; ++k) {
                // global path: word = x | rack << 16 (little endian: x is the low half; none = all ones); LDS path: the index itself
                const uint16_t x = ga ? raw[((size_t)p * nw + k) * 2] : raw[(size_t)p * nw + k];
                final_state[(size_t)p * d.RF + k] = x < pt.ext_of.size() ? pt.ext_of[x] : (uint16_t)KAO_NONE;
            }
    }
    if (best_state)
        HIP_TRY(hipMemcpy(best_state, s->d_best + d.best_off + (uint64_t)restart * d.P * d.RF, (size_t)d.P * d.RF * 2, hipMemcpyDeviceToHost));
    if (info) HIP_TRY(hipMemcpy(info, s->d_info + (size_t)(d.restart_base + restart) * 4, 16, hipMemcpyDeviceToHost));
    return KAO_OK;
}

}  // extern "C"

namespace kao {

// the topic's winning assignment (dense [P*RF]) as of the last finished launch
int session_topic_best(kao_session *s, int i, uint16_t *out) {
    HIP_TRY(hipSetDevice(s->device));
    launch_gather(s->d_topics, s->n_topics, s->d_keys, s->d_best, s->d_viol, s->d_win_assign, s->d_win_viol, s->stream);
    HIP_TRY(hipGetLastError());
    const TopicDev &d = s->pts[(size_t)i].d;
    HIP_TRY(hipMemcpyAsync(out, s->d_win_assign + d.win_off, (size_t)d.P * d.RF * 2, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return KAO_OK;
}
// every restart's best feasible objective (-1: never feasible) and one restart's best snapshot (dense [P*RF]) as of the last
// finished launch: the starting points of kao_solve's further KAO-CX runs
int session_restart_objs(kao_session *s, int i, std::vector<int32_t> &objs) {
    HIP_TRY(hipSetDevice(s->device));
    const TopicDev &d = s->pts[(size_t)i].d;
    std::vector<int32_t> info((size_t)d.n_restarts * 4);
    HIP_TRY(hipMemcpyAsync(info.data(), s->d_info + (size_t)d.restart_base * 4, info.size() * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    objs.resize((size_t)d.n_restarts);
    for (int r = 0; r < d.n_restarts; ++r) objs[(size_t)r] = info[(size_t)r * 4];
    return KAO_OK;
}
int session_restart_best(kao_session *s, int i, int restart, uint16_t *out) {
    HIP_TRY(hipSetDevice(s->device));
    const TopicDev &d = s->pts[(size_t)i].d;
    if (restart < 0 || restart >= d.n_restarts) return fail(KAO_ERR_INVALID, "bad restart");
    HIP_TRY(hipMemcpyAsync(out, s->d_best + d.best_off + (uint64_t)restart * d.P * d.RF, (size_t)d.P * d.RF * 2, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return KAO_OK;
}
// a feasible assignment found outside K-search (KAO-CX) becomes the topic's incumbent: winner buffer + packed key with the
// reserved restart id, exactly as an assignment adopted from another GPU (k_gather leaves it alone, elite launches re-seed from it)
int session_adopt_external(kao_session *s, int i, const uint16_t *assign, int64_t objective, uint64_t *key_out) {
    HIP_TRY(hipSetDevice(s->device));
    const TopicDev &d = s->pts[(size_t)i].d;
    const uint64_t key = ((uint64_t)((int64_t)kObjCap - objective) << 20) | (uint64_t)kExternalRestart;
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipMemcpyAsync(s->d_win_assign + d.win_off, assign, (size_t)d.P * d.RF * 2, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipMemcpyAsync(s->d_keys + i, &key, 8, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    *key_out = key;
    return KAO_OK;
}

}  // namespace kao
