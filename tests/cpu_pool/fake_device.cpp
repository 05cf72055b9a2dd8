// TEST-ONLY: a fake HIP runtime and a fake libbpgpu back end, so that bulletproofs_amd/csrc/pool.hip -- the pool's scheduler:
// combining queue, tickets, flush planning -- compiles with g++ (-fsanitize=thread) and runs on a machine without a GPU.
//   * streams are threads executing queued operations in order; events complete when the stream reaches them; copies are memcpys
//     done BY THE STREAM THREAD (so a host block that a caller still writes while its chain runs is a data race ThreadSanitizer sees);
//   * a "launch chain" sleeps for a modelled time (base + per item + per chain running beside it) and then writes, for every item,
//     results that are a function of that item's inputs only (fake_model.h) -- the driver recomputes them and thereby checks that
//     every request got ITS results, whatever chain carried it.
// Nothing here is linked into libbpgpu.so; nothing of the product falls back to it.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bpgpu.h"
#include "../../bulletproofs_amd/csrc/rangeproof.h"
#include "fake_model.h"

// ---- the fake runtime ---------------------------------------------------------------------------------------------------
struct ihipStream_t {
    std::mutex mu;
    std::condition_variable cv, idle_cv;
    std::deque<std::function<void()>> q;
    bool stop = false, busy = false;
    std::thread th;
    ihipStream_t() : th([this] { run(); }) {}
    ~ihipStream_t() {
        {
            std::lock_guard<std::mutex> g(mu);
            stop = true;
        }
        cv.notify_all();
        th.join();
    }
    void run() {
        for (;;) {
            std::function<void()> op;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                op = std::move(q.front());
                q.pop_front();
                busy = true;
            }
            op();
            {
                std::lock_guard<std::mutex> g(mu);
                busy = false;
                if (q.empty()) idle_cv.notify_all();
            }
        }
    }
    void push(std::function<void()> f) {
        {
            std::lock_guard<std::mutex> g(mu);
            q.push_back(std::move(f));
        }
        cv.notify_one();
    }
    void sync() {
        std::unique_lock<std::mutex> lk(mu);
        idle_cv.wait(lk, [&] { return q.empty() && !busy; });
    }
    bool idle() {
        std::lock_guard<std::mutex> g(mu);
        return q.empty() && !busy;
    }
};
struct ihipEvent_t {
    std::atomic<uint64_t> recorded{0}, done{0};   // tickets: record k completes when done >= k
};
static ihipStream_t *default_stream() {
    static ihipStream_t *s = new ihipStream_t();
    return s;
}
static inline ihipStream_t *S(hipStream_t s) { return s ? (ihipStream_t *)s : default_stream(); }
static thread_local int t_device = 0;
static std::atomic<int> g_ndev{4};
static std::atomic<int> g_fail_hostmalloc{0};

extern "C" {
hipError_t hipSetDevice(int d) {
    if (d < 0 || d >= g_ndev) return hipErrorInvalidDevice;
    t_device = d;
    return hipSuccess;
}
hipError_t hipGetDevice(int *d) {
    *d = t_device;
    return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "fake HIP error"; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) {
    *s = (hipStream_t) new ihipStream_t();
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
    if (s) {
        S(s)->sync();
        delete (ihipStream_t *)s;
    }
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
    S(s)->sync();
    return hipSuccess;
}
hipError_t hipStreamQuery(hipStream_t s) { return S(s)->idle() ? hipSuccess : hipErrorNotReady; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) {
    *e = (hipEvent_t) new ihipEvent_t();
    return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
    // (the real runtime releases an event once its recorded work completed; the fake waits for that)
    ihipEvent_t *ev = (ihipEvent_t *)e;
    while (ev->done.load(std::memory_order_acquire) < ev->recorded.load(std::memory_order_acquire)) std::this_thread::yield();
    delete ev;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    ihipEvent_t *ev = (ihipEvent_t *)e;
    const uint64_t k = ev->recorded.fetch_add(1, std::memory_order_acq_rel) + 1;
    S(s)->push([ev, k] {
        uint64_t cur = ev->done.load(std::memory_order_relaxed);
        while (cur < k && !ev->done.compare_exchange_weak(cur, k, std::memory_order_release)) {
        }
    });
    return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t e) {
    ihipEvent_t *ev = (ihipEvent_t *)e;
    return ev->done.load(std::memory_order_acquire) >= ev->recorded.load(std::memory_order_acquire) ? hipSuccess : hipErrorNotReady;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
    ihipEvent_t *ev = (ihipEvent_t *)e;
    const uint64_t k = ev->recorded.load(std::memory_order_acquire);
    while (ev->done.load(std::memory_order_acquire) < k) std::this_thread::sleep_for(std::chrono::microseconds(20));
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
    ihipEvent_t *ev = (ihipEvent_t *)e;
    const uint64_t k = ev->recorded.load(std::memory_order_acquire);
    S(s)->push([ev, k] {
        while (ev->done.load(std::memory_order_acquire) < k) std::this_thread::sleep_for(std::chrono::microseconds(10));
    });
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned) {
    if (g_fail_hostmalloc.load()) return hipErrorOutOfMemory;
    *p = malloc(n ? n : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void *p) {
    free(p);
    return hipSuccess;
}
hipError_t hipHostGetDevicePointer(void **dp, void *hp, unsigned) {
    *dp = hp;
    return hipSuccess;
}
hipError_t hipMalloc(void **p, size_t n) {
    *p = malloc(n ? n : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void *p) {
    free(p);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t s) {
    S(s)->push([=] { memcpy(dst, src, n); });
    return hipSuccess;
}
hipError_t hipMemcpyPeerAsync(void *dst, int, const void *src, int, size_t n, hipStream_t s) {
    S(s)->push([=] { memcpy(dst, src, n); });
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t s) {
    S(s)->push([=] { memset(dst, v, n); });
    return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) {
    *v = 100000;
    return hipSuccess;
}
}   // extern "C"

// ---- the fake back end: what pool.hip calls of bpgpu.hip ---------------------------------------------------------------------
static std::atomic<int> g_chains_running{0};
static std::atomic<uint64_t> g_model_base_us{550}, g_model_item_ns{200}, g_model_beside_us{100};
static std::atomic<int> g_fail_chains{0};   // > 0: the next chains fail to issue (and count down)
extern "C" void fake_set_model(uint64_t base_us, uint64_t item_ns, uint64_t beside_us) {
    g_model_base_us = base_us, g_model_item_ns = item_ns, g_model_beside_us = beside_us;
}
extern "C" void fake_fail_next_chains(int n) { g_fail_chains = n; }
extern "C" void fake_set_devices(int n) { g_ndev = n; }

struct bpgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::string err;
    std::map<std::string, int64_t> opts;
    size_t gens_n = 0, gens_m = 0, sec_n = 0, sec_m = 0;
    bool tables = false;
    struct pend_t {
        bool active = false;
    } pend;
};
static int cfail(bpgpu_ctx *c, int code, const char *msg) {
    c->err = msg;
    return code;
}
// the modelled duration of one chain of K items, measured from when the stream reaches it
static void model_sleep(size_t K, uint64_t item_ns_scale = 1) {
    const int beside = g_chains_running.fetch_add(1);
    const uint64_t us = g_model_base_us + (K * g_model_item_ns * item_ns_scale) / 1000 + (uint64_t)beside * g_model_beside_us;
    std::this_thread::sleep_for(std::chrono::microseconds(us));
    g_chains_running.fetch_sub(1);
}

extern "C" {
int bpgpu_ctx_create(int device, bpgpu_ctx **out) {
    if (device < 0 || device >= g_ndev) return BPGPU_ERR_NO_DEVICE;
    bpgpu_ctx *c = new bpgpu_ctx();
    c->device = device;
    hipStreamCreateWithFlags(&c->stream, 0);
    *out = c;
    return BPGPU_OK;
}
void bpgpu_ctx_destroy(bpgpu_ctx *c) {
    if (!c) return;
    hipStreamDestroy(c->stream);
    delete c;
}
const char *bpgpu_last_error(bpgpu_ctx *c) { return c ? c->err.c_str() : "null context"; }
int bpgpu_ctx_set_option(bpgpu_ctx *c, const char *key, int64_t v) {
    std::lock_guard<std::mutex> g(c->mu);
    c->opts[key] = v;
    return BPGPU_OK;
}
int bpgpu_ctx_get_option(bpgpu_ctx *c, const char *key, int64_t *v) {
    std::lock_guard<std::mutex> g(c->mu);
    auto it = c->opts.find(key);
    *v = it == c->opts.end() ? 0 : it->second;
    return BPGPU_OK;
}
int bpgpu_synchronize(bpgpu_ctx *c) {
    hipStreamSynchronize(c->stream);
    return BPGPU_OK;
}
int bpgpu_gens_create(bpgpu_ctx *c, size_t n, size_t m) {
    std::lock_guard<std::mutex> g(c->mu);
    c->gens_n = n, c->gens_m = m, c->tables = true;
    return BPGPU_OK;
}
int bpgpu_gens_load(bpgpu_ctx *c, size_t n, size_t m, const uint8_t *, const uint8_t *, const uint8_t *, const uint8_t *) { return bpgpu_gens_create(c, n, m); }
int bpgpu_gens_export(bpgpu_ctx *c, uint8_t *G, uint8_t *H, uint8_t B[32], uint8_t Bb[32]) {
    std::lock_guard<std::mutex> g(c->mu);
    memset(G, 1, c->gens_n * c->gens_m * 32);
    memset(H, 2, c->gens_n * c->gens_m * 32);
    memset(B, 3, 32);
    memset(Bb, 4, 32);
    return BPGPU_OK;
}
int bpgpu_gens_add_shape(bpgpu_ctx *c, size_t n2, size_t m2) {
    std::lock_guard<std::mutex> g(c->mu);
    c->sec_n = n2, c->sec_m = m2, c->tables = true;
    return BPGPU_OK;
}
int bpgpu_internal_release_tables(bpgpu_ctx *c) {
    std::lock_guard<std::mutex> g(c->mu);
    c->tables = false;
    return BPGPU_OK;
}
int bpgpu_transcript_new(const uint8_t *label, size_t label_len, uint8_t st[BPGPU_TRANSCRIPT_BYTES]) {
    fake_transcript_new(label, label_len, st);
    return BPGPU_OK;
}
// the synchronous entry points the pool uses for requests no chain can take: every proof FormatError-like (verdict 2), states handed back untouched
int bpgpu_rangeproof_verify_batch_ts(bpgpu_ctx *c, size_t, size_t, size_t nbatch, const uint8_t *, size_t, const uint8_t *, const uint8_t *ts, size_t stride,
                                     const uint8_t *, uint8_t *verdict, uint8_t *msm_out, uint8_t *ts_out) {
    std::lock_guard<std::mutex> g(c->mu);
    memset(verdict, FAKE_VERDICT_DIRECT, nbatch);
    if (msm_out) memset(msm_out, 0, nbatch * 32);
    if (ts_out)
        for (size_t i = 0; i < nbatch; i++) memcpy(ts_out + i * BPGPU_TRANSCRIPT_BYTES, ts + i * stride, BPGPU_TRANSCRIPT_BYTES);
    return BPGPU_OK;
}
int bpgpu_msm_batch_shared(bpgpu_ctx *c, size_t, size_t, size_t, size_t, const uint8_t *, const uint8_t *, const uint8_t *, uint8_t *, uint8_t *) {
    return cfail(c, BPGPU_ERR_NO_GENS, "fake: shape not served");
}
int bpgpu_ipp_verify_batch(bpgpu_ctx *c, size_t, size_t nbatch, const uint8_t *, size_t, const uint8_t *, size_t, const uint8_t *, const uint8_t *, const uint8_t *,
                           const uint8_t *, const uint8_t *, const uint8_t *, uint8_t *verdict, uint8_t *msm_out) {
    std::lock_guard<std::mutex> g(c->mu);
    memset(verdict, FAKE_VERDICT_DIRECT, nbatch);
    if (msm_out) memset(msm_out, 0, nbatch * 32);
    return BPGPU_OK;
}
// round 3's host path (option host_path_combining = 0)
int bpgpu_rangeproof_verify_batch_submit(bpgpu_ctx *c, size_t n, size_t m, size_t nbatch, const uint8_t *proofs, size_t proof_len, const uint8_t *coms,
                                         const uint8_t *label, size_t label_len, const uint8_t *, uint8_t *verdict, uint8_t *msm_out) {
    std::lock_guard<std::mutex> g(c->mu);
    uint8_t st0[BPGPU_TRANSCRIPT_BYTES];
    fake_transcript_new(label, label_len, st0);
    for (size_t i = 0; i < nbatch; i++) fake_rp_result(n, m, proofs + i * proof_len, proof_len, coms + i * m * 32, st0, verdict + i, nullptr, msm_out ? msm_out + i * 32 : nullptr);
    return BPGPU_OK;
}
int bpgpu_ctx_collect(bpgpu_ctx *) { return BPGPU_OK; }
int bpgpu_rangeproof_verify_batch_dev(bpgpu_ctx *c, size_t, size_t, size_t nbatch, const void *, size_t, const void *, const uint8_t *, size_t, const void *,
                                      void *d_verdict, void *d_msm_out, void *) {
    hipMemsetAsync(d_verdict, FAKE_VERDICT_DIRECT, nbatch, c->stream);
    if (d_msm_out) hipMemsetAsync(d_msm_out, 0, nbatch * 32, c->stream);
    return BPGPU_OK;
}
int bpgpu_rangeproof_verify_rlc_dev(bpgpu_ctx *c, size_t, size_t, size_t nbatch, const void *, size_t, const void *, const uint8_t *, size_t, const void *, const void *,
                                    void *d_verdict, void *d_batch_out, void *) {
    hipMemsetAsync(d_verdict, FAKE_VERDICT_DIRECT, nbatch, c->stream);
    if (d_batch_out) hipMemsetAsync(d_batch_out, 0, 33, c->stream);
    return BPGPU_OK;
}
int bpgpu_msm_batch_shared_dev(bpgpu_ctx *c, size_t n, size_t m, size_t nbatch, size_t nu, const void *d_gs, const void *d_us, const void *d_up, void *d_out,
                               void *d_status, void *) {
    if (g_fail_chains.load() > 0 && g_fail_chains.fetch_sub(1) > 0) return cfail(c, BPGPU_ERR_HIP, "fake: injected chain failure");
    {
        std::lock_guard<std::mutex> g(c->mu);
        if (!c->tables || n > c->gens_n || m > c->gens_m) return cfail(c, BPGPU_ERR_NO_GENS, "fake: generators too small");
    }
    const size_t ng = 2 * n * m + 2;
    S(c->stream)->push([=] {
        model_sleep(nbatch, 40);
        for (size_t i = 0; i < nbatch; i++)
            fake_msm_shared_result((const uint8_t *)d_gs + i * ng * 32, ng, nu ? (const uint8_t *)d_us + i * nu * 32 : nullptr, nu ? (const uint8_t *)d_up + i * nu * 32 : nullptr, nu,
                                   (uint8_t *)d_out + i * 32, (uint8_t *)d_status + i);
    });
    return BPGPU_OK;
}
int bpgpu_msm_batch_dev(bpgpu_ctx *c, size_t nbatch, const uint32_t *n_terms, const void *d_s, const void *d_p, void *d_out, void *d_status, void *) {
    if (g_fail_chains.load() > 0 && g_fail_chains.fetch_sub(1) > 0) return cfail(c, BPGPU_ERR_HIP, "fake: injected chain failure");
    std::vector<uint32_t> nt(n_terms, n_terms + nbatch);
    S(c->stream)->push([=] {
        model_sleep(nbatch, 10);
        size_t t0 = 0;
        for (size_t i = 0; i < nbatch; i++) {
            fake_msm_result((const uint8_t *)d_s + t0 * 32, (const uint8_t *)d_p + t0 * 32, nt[i], (uint8_t *)d_out + i * 32, (uint8_t *)d_status + i);
            t0 += nt[i];
        }
    });
    return BPGPU_OK;
}
int bpgpu_ipp_verify_batch_dev(bpgpu_ctx *c, size_t n, size_t nbatch, const void *d_proofs, size_t proof_len, const uint8_t *label, size_t label_len,
                               const uint8_t *shared_ts, const void *d_Gf, const void *d_Hf, const void *d_P, const void *d_Q, const void *d_G, const void *d_H, int,
                               void *d_verdict, void *d_msm_out, void *) {
    if (g_fail_chains.load() > 0 && g_fail_chains.fetch_sub(1) > 0) return cfail(c, BPGPU_ERR_HIP, "fake: injected chain failure");
    std::vector<uint8_t> st0(BPGPU_TRANSCRIPT_BYTES, 0);
    if (shared_ts) memcpy(st0.data(), shared_ts, 203);
    else fake_transcript_new(label, label_len, st0.data());
    S(c->stream)->push([=] {
        model_sleep(nbatch, 4);
        for (size_t i = 0; i < nbatch; i++)
            fake_ipp_result(n, (const uint8_t *)d_proofs + i * proof_len, proof_len, st0.data(), (const uint8_t *)d_Gf + i * n * 32, (const uint8_t *)d_Hf + i * n * 32,
                            (const uint8_t *)d_P + i * 32, (const uint8_t *)d_Q + i * 32, (const uint8_t *)d_G + i * n * 32, (const uint8_t *)d_H + i * n * 32,
                            (uint8_t *)d_verdict + i, d_msm_out ? (uint8_t *)d_msm_out + i * 32 : nullptr);
    });
    return BPGPU_OK;
}
}   // extern "C"

// ---- hooks (C++ linkage, as bpgpu.hip defines them) ----------------------------------------------------------------------------
bool bpgpu_internal_rp_coalescible(bpgpu_ctx *c, size_t n, size_t m, size_t proof_len) {
    if (proof_len % 32 != 0 || proof_len < 9 * 32 || ((proof_len / 32 - 9) & 1)) return false;
    const size_t k = (proof_len / 32 - 9) / 2;
    if (k > 16 || !(n == 8 || n == 16 || n == 32 || n == 64) || m == 0) return false;
    std::lock_guard<std::mutex> lk(c->mu);
    return c->tables && c->gens_n >= n && c->gens_m >= m && n * m == ((size_t)1 << k);
}
bool bpgpu_internal_idle(bpgpu_ctx *c) { return hipStreamQuery(c->stream) == hipSuccess; }
void bpgpu_internal_set_busy_hint(bpgpu_ctx *, int) {}
void *bpgpu_internal_stream(bpgpu_ctx *c) { return c ? (void *)c->stream : nullptr; }
int bpgpu_internal_rp_reserve(bpgpu_ctx *, size_t, size_t, size_t, size_t) { return BPGPU_OK; }
int bpgpu_internal_rp_verify_chain(bpgpu_ctx *c, size_t n, size_t m, size_t nbatch, const void *d_proofs, size_t proof_len, const void *d_coms, const uint8_t *shared_ts,
                                   const void *d_ts_in, void *d_ts_out, int, uint32_t, uint32_t, uint32_t, const void *d_rng64, void *d_verdict, void *d_msm_out, uint32_t,
                                   int) {
    if (!d_rng64 || (shared_ts == nullptr) == (d_ts_in == nullptr)) return cfail(c, BPGPU_ERR_INVALID_ARG, "fake: bad chain arguments");
    if (g_fail_chains.load() > 0 && g_fail_chains.fetch_sub(1) > 0) return cfail(c, BPGPU_ERR_HIP, "fake: injected chain failure");
    std::vector<uint8_t> sh;
    if (shared_ts) sh.assign(shared_ts, shared_ts + BPGPU_TRANSCRIPT_BYTES);
    S(c->stream)->push([=] {
        model_sleep(nbatch);
        for (size_t i = 0; i < nbatch; i++) {
            const uint8_t *ts = d_ts_in ? (const uint8_t *)d_ts_in + i * BPGPU_TRANSCRIPT_BYTES : sh.data();
            fake_rp_result(n, m, (const uint8_t *)d_proofs + i * proof_len, proof_len, (const uint8_t *)d_coms + i * m * 32, ts, (uint8_t *)d_verdict + i,
                           d_ts_out ? (uint8_t *)d_ts_out + i * BPGPU_TRANSCRIPT_BYTES : nullptr, d_msm_out ? (uint8_t *)d_msm_out + i * 32 : nullptr);
        }
    });
    return BPGPU_OK;
}
int bpgpu_internal_rp_verify_segs(bpgpu_ctx *c, size_t n, size_t m, size_t proof_len, const uint8_t *const *labels, size_t label_len, const bp::rp_seg *segs, uint32_t nseg,
                                  bool any_msm, uint32_t, int, bool rlc) {
    if (g_fail_chains.load() > 0 && g_fail_chains.fetch_sub(1) > 0) return cfail(c, BPGPU_ERR_HIP, "fake: injected chain failure");
    std::vector<bp::rp_seg> sg(segs, segs + nseg);
    std::vector<std::vector<uint8_t>> st0(nseg, std::vector<uint8_t>(BPGPU_TRANSCRIPT_BYTES));
    for (uint32_t i = 0; i < nseg; i++) fake_transcript_new(labels[i], label_len, st0[i].data());   // (one label per segment, all of one length)
    size_t total = 0;
    for (const bp::rp_seg &x : sg) total += x.count;
    S(c->stream)->push([=] {
        model_sleep(total);
        for (uint32_t si = 0; si < (uint32_t)sg.size(); si++) {
            const bp::rp_seg &x = sg[si];
            for (uint32_t i = 0; i < x.count; i++) {
                uint8_t v, enc[32];
                fake_rp_result(n, m, x.proofs + (size_t)i * proof_len, proof_len, x.commitments + (size_t)i * m * 32, st0[si].data(), &v, nullptr, enc);
                x.verdict[i] = v;
                if (!rlc && any_msm && x.msm_out) memcpy((uint8_t *)x.msm_out + (size_t)i * 32, enc, 32);
            }
        }
        if (rlc && any_msm)
            for (const bp::rp_seg &x : sg)
                if (x.msm_out) memset(x.msm_out, 0, 33);
    });
    return BPGPU_OK;
}
