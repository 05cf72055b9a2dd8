// pool.hip: the scheduler of libbpgpu.so -- bpgpu_pool_* of include/bpgpu.h.
//
// What the reference offers is ONE call for any number of proofs (a loop over RangeProof::verify_multiple,
// src/range_proof/mod.rs:457-470, from as many threads as the caller likes).  What the device wants is several launch
// chains in flight, each a few thousand proofs wide (DESIGN.md "scheduling").  The pool sits between the two:
//   * a pool owns `lanes` contexts (one HIP stream, one arena each; the generator tables are shared) on every device it
//     was given, and a few host worker threads per device;
//   * bpgpu_pool_rangeproof_verify (host pointers, synchronous, any nbatch): contiguous shard per device, sliced, the
//     slices staged / enqueued / collected by the workers, verdicts gathered in order into the caller's buffer;
//   * bpgpu_pool_rangeproof_submit_dev (device pointers, asynchronous): items queue up; a flush packs consecutive items of
//     one shape into coalesced launch chains (rp_seg, rangeproof.h) of about `coalesce_proofs` proofs and issues them on
//     the lanes round-robin -- a burst of small batches is served as a few wide chains instead of many narrow ones.
// No CPU fallback: creation fails without a device.
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bpgpu.h"
#include "rangeproof.h"

using namespace bp;

bool bpgpu_internal_rp_coalescible(bpgpu_ctx *c, size_t n, size_t m, size_t proof_len);
bool bpgpu_internal_idle(bpgpu_ctx *c);
void bpgpu_internal_set_busy_hint(bpgpu_ctx *c, int busy);
int bpgpu_internal_rp_verify_segs(bpgpu_ctx *c, size_t n, size_t m, size_t proof_len, const uint8_t *label, size_t label_len, const rp_seg *segs,
                                  uint32_t nseg, bool any_msm, uint32_t splits_hint);

// The ROCm runtime maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and reads the variable when it
// initialises, i.e. at the process's first HIP call.  Kernels of different lanes only overlap when the lanes sit on different
// queues (measured: 4 queues 3.0 M/s, 16 queues 4.4 M/s, more than 16 collapses), so the library asks for 16 when it is
// loaded, unless the caller chose a value.
namespace {
struct hwq_init {
    hwq_init() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }
} g_hwq_init;
}  // namespace

namespace {

struct dev_item {   // a submitted device-pointer batch waiting for the next flush
    size_t n, m, nbatch, proof_len;
    const uint8_t *proofs, *coms, *rng;
    uint8_t *verdict, *msm;
    std::string label;
    bool same_shape(const dev_item &o) const { return n == o.n && m == o.m && proof_len == o.proof_len && label == o.label; }
};

struct pool_dev {
    int device = 0;
    std::vector<bpgpu_ctx *> lanes;
    size_t next_lane = 0, used_lanes = 0;   // lanes [0, used_lanes) have been given chains since the pool was last found idle
    std::vector<dev_item> pending;
    size_t pending_proofs = 0;
    // host workers: worker w serves the slices of host-pointer calls on lane w (synchronously; several workers = several
    // chains in flight, their staging copies running on as many cores)
    std::vector<std::thread> workers;
    std::vector<std::deque<std::function<void(bpgpu_ctx *)>>> tasks;   // one queue per worker: slice i of a call always goes to worker i mod W, so a
                                                                       // lane sees the same slice widths call after call (its arena is sized once)
    std::mutex tmu;
    std::condition_variable tcv;
    bool stop = false;
};

}  // namespace

struct bpgpu_pool {
    std::vector<pool_dev *> devs;
    std::mutex mu;        // serialises the pool's own state (pending lists, options); lane contexts have their own locks
    std::string err;
    size_t coalesce_proofs = 5120;   // target width of a coalesced launch chain
    size_t pair_limit_proofs = 24576;   // a flush of up to this many proofs is issued as at most two chains (flush_dev)
    size_t max_chain_proofs = 16384; // never wider than this (arena of a lane: ~55 KB per proof)
    size_t slice_proofs = 0;         // host-pointer calls: proofs per slice (0 = automatic)
    size_t latency_proofs = 6144;    // a host call / a flush on an idle device of up to this many proofs is "alone": its chains take the latency forms
    size_t auto_flush_items = 0;     // flush by itself once this many items wait on a device (0 = lanes)
    size_t auto_flush_proofs = 0;    // ... or once this many proofs wait: they go out as ONE chain while the caller keeps submitting (0 = off)
    size_t host_workers = 0;
    // statistics of the coalesced path (get_option "stat_chains" / "stat_chain_proofs" / "stat_last_splits"; set "stat_reset")
    uint64_t stat_chains = 0, stat_chain_proofs = 0, stat_last_splits = 0;
};

static int pfail(bpgpu_pool *p, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (p) p->err = buf;
    return code;
}

static void worker_main(pool_dev *d, size_t w) {
    hipSetDevice(d->device);
    for (;;) {
        std::function<void(bpgpu_ctx *)> job;
        {
            std::unique_lock<std::mutex> lk(d->tmu);
            d->tcv.wait(lk, [&] { return d->stop || !d->tasks[w].empty(); });
            if (d->tasks[w].empty()) return;   // stop requested and nothing left
            job = std::move(d->tasks[w].front());
            d->tasks[w].pop_front();
        }
        job(d->lanes[w]);
    }
}

static void stop_workers(pool_dev *d) {
    {
        std::lock_guard<std::mutex> lk(d->tmu);
        d->stop = true;
    }
    d->tcv.notify_all();
    for (auto &t : d->workers) t.join();
    d->workers.clear();
}

extern "C" {

int bpgpu_pool_create(const int *devices, int ndev, int lanes_per_device, bpgpu_pool **out) {
    if (!out) return BPGPU_ERR_INVALID_ARG;
    *out = nullptr;
    if (!devices || ndev <= 0 || ndev > 64 || lanes_per_device < 0 || lanes_per_device > 1024) return BPGPU_ERR_INVALID_ARG;
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    const int hwq = q ? atoi(q) : 0;
    if (lanes_per_device == 0) lanes_per_device = 32;
    if (lanes_per_device > 4 && (hwq < 8 || hwq > 16)) {
        // fail loudly: with the default 4 hardware queues the lanes would serialise and the pool would deliver a fraction of its rate
        fprintf(stderr, "libbpgpu: GPU_MAX_HW_QUEUES=%s -- the pool's lanes need 8..16 hardware queues (export GPU_MAX_HW_QUEUES=16 before the "
                        "process's first HIP call, or leave it unset and load libbpgpu before HIP initialises)\n", q ? q : "(unset)");
        return BPGPU_ERR_HW_QUEUES;
    }
    bpgpu_pool *p = new bpgpu_pool();
    for (int i = 0; i < ndev; i++) {
        pool_dev *d = new pool_dev();
        d->device = devices[i];
        p->devs.push_back(d);
        for (int l = 0; l < lanes_per_device; l++) {
            bpgpu_ctx *c = nullptr;
            const int rc = bpgpu_ctx_create(devices[i], &c);
            if (rc) {
                bpgpu_pool_destroy(p);
                return rc;
            }
            d->lanes.push_back(c);
        }
    }
    *out = p;
    return BPGPU_OK;
}

void bpgpu_pool_destroy(bpgpu_pool *p) {
    if (!p) return;
    for (pool_dev *d : p->devs) {
        stop_workers(d);
        for (bpgpu_ctx *c : d->lanes) bpgpu_ctx_destroy(c);
        delete d;
    }
    delete p;
}

const char *bpgpu_pool_last_error(bpgpu_pool *p) { return p ? p->err.c_str() : "null pool"; }

int bpgpu_pool_devices(bpgpu_pool *p) { return p ? (int)p->devs.size() : 0; }
int bpgpu_pool_lanes(bpgpu_pool *p) { return (p && !p->devs.empty()) ? (int)p->devs[0]->lanes.size() : 0; }
bpgpu_ctx *bpgpu_pool_lane(bpgpu_pool *p, int dev_index, int lane) {
    if (!p || dev_index < 0 || dev_index >= (int)p->devs.size()) return nullptr;
    pool_dev *d = p->devs[dev_index];
    return (lane >= 0 && lane < (int)d->lanes.size()) ? d->lanes[lane] : nullptr;
}

int bpgpu_pool_set_option(bpgpu_pool *p, const char *key, int64_t value) {
    if (!p || !key) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(p->mu);
    if (!strcmp(key, "coalesce_proofs")) {
        if (value < 1 || value > (1 << 20)) return pfail(p, BPGPU_ERR_INVALID_ARG, "coalesce_proofs out of range");
        p->coalesce_proofs = (size_t)value;
        if (p->max_chain_proofs < p->coalesce_proofs) p->max_chain_proofs = p->coalesce_proofs;
        return BPGPU_OK;
    }
    if (!strcmp(key, "max_chain_proofs")) {
        if (value < 64 || value > (1 << 22)) return pfail(p, BPGPU_ERR_INVALID_ARG, "max_chain_proofs out of range");
        p->max_chain_proofs = (size_t)value;
        if (p->coalesce_proofs > p->max_chain_proofs) p->coalesce_proofs = p->max_chain_proofs;
        return BPGPU_OK;
    }
    if (!strcmp(key, "slice_proofs")) {
        if (value < 0 || value > (1 << 22)) return pfail(p, BPGPU_ERR_INVALID_ARG, "slice_proofs out of range");
        p->slice_proofs = (size_t)value;
        return BPGPU_OK;
    }
    if (!strcmp(key, "pair_limit_proofs")) {
        if (value < 0 || value > (1 << 24)) return pfail(p, BPGPU_ERR_INVALID_ARG, "pair_limit_proofs out of range");
        p->pair_limit_proofs = (size_t)value;
        return BPGPU_OK;
    }
    if (!strcmp(key, "latency_proofs")) {
        if (value < 0 || value > (1 << 24)) return pfail(p, BPGPU_ERR_INVALID_ARG, "latency_proofs out of range");
        p->latency_proofs = (size_t)value;
        return BPGPU_OK;
    }
    if (!strcmp(key, "auto_flush_proofs")) {
        if (value < 0 || value > (1 << 22)) return pfail(p, BPGPU_ERR_INVALID_ARG, "auto_flush_proofs out of range");
        p->auto_flush_proofs = (size_t)value;
        return BPGPU_OK;
    }
    if (!strcmp(key, "auto_flush_items")) {
        if (value < 0 || value > (1 << 20)) return pfail(p, BPGPU_ERR_INVALID_ARG, "auto_flush_items out of range");
        p->auto_flush_items = (size_t)value;
        return BPGPU_OK;
    }
    if (!strcmp(key, "stat_reset")) {
        p->stat_chains = p->stat_chain_proofs = 0;
        return BPGPU_OK;
    }
    if (!strcmp(key, "host_workers")) {
        if (value < 0 || value > 256) return pfail(p, BPGPU_ERR_INVALID_ARG, "host_workers out of range");
        for (pool_dev *d : p->devs)
            if (!d->workers.empty()) return pfail(p, BPGPU_ERR_INVALID_ARG, "set host_workers before the first host-pointer call");
        p->host_workers = (size_t)value;
        return BPGPU_OK;
    }
    // everything else is an option of the lane contexts (fixed_window_bits, horner_lanes, ...)
    for (pool_dev *d : p->devs)
        for (bpgpu_ctx *c : d->lanes) {
            const int rc = bpgpu_ctx_set_option(c, key, value);
            if (rc) return pfail(p, rc, "%s", bpgpu_last_error(c));
        }
    return BPGPU_OK;
}

int bpgpu_pool_get_option(bpgpu_pool *p, const char *key, int64_t *value) {
    if (!p || !key || !value) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(p->mu);
    if (!strcmp(key, "coalesce_proofs")) *value = (int64_t)p->coalesce_proofs;
    else if (!strcmp(key, "max_chain_proofs")) *value = (int64_t)p->max_chain_proofs;
    else if (!strcmp(key, "slice_proofs")) *value = (int64_t)p->slice_proofs;
    else if (!strcmp(key, "auto_flush_items")) *value = (int64_t)p->auto_flush_items;
    else if (!strcmp(key, "auto_flush_proofs")) *value = (int64_t)p->auto_flush_proofs;
    else if (!strcmp(key, "latency_proofs")) *value = (int64_t)p->latency_proofs;
    else if (!strcmp(key, "pair_limit_proofs")) *value = (int64_t)p->pair_limit_proofs;
    else if (!strcmp(key, "host_workers")) *value = (int64_t)p->host_workers;
    else if (!strcmp(key, "stat_chains")) *value = (int64_t)p->stat_chains;
    else if (!strcmp(key, "stat_chain_proofs")) *value = (int64_t)p->stat_chain_proofs;
    else if (!strcmp(key, "stat_last_splits")) *value = (int64_t)p->stat_last_splits;
    else if (p->devs.empty() || p->devs[0]->lanes.empty()) return BPGPU_ERR_INVALID_ARG;
    else return bpgpu_ctx_get_option(p->devs[0]->lanes[0], key, value);
    return BPGPU_OK;
}

// generators: derived (or loaded) on lane 0 of every device, which builds that device's window tables; the other lanes load
// the same encodings and find the tables already there
static int pool_spread_gens(bpgpu_pool *p, size_t gens_capacity, size_t party_capacity) {
    const size_t tot = gens_capacity * party_capacity;
    std::vector<uint8_t> G(tot * 32), H(tot * 32);
    uint8_t B[32], Bb[32];
    for (pool_dev *d : p->devs) {
        int rc = bpgpu_gens_export(d->lanes[0], G.data(), H.data(), B, Bb);
        if (rc) return pfail(p, rc, "%s", bpgpu_last_error(d->lanes[0]));
        for (size_t l = 1; l < d->lanes.size(); l++) {
            rc = bpgpu_gens_load(d->lanes[l], gens_capacity, party_capacity, G.data(), H.data(), B, Bb);
            if (rc) return pfail(p, rc, "%s", bpgpu_last_error(d->lanes[l]));
        }
    }
    return BPGPU_OK;
}
int bpgpu_pool_gens_create(bpgpu_pool *p, size_t gens_capacity, size_t party_capacity) {
    if (!p || gens_capacity == 0 || party_capacity == 0) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(p->mu);
    for (pool_dev *d : p->devs) {
        if (d->lanes.empty()) return BPGPU_ERR_INVALID_ARG;
        const int rc = bpgpu_gens_create(d->lanes[0], gens_capacity, party_capacity);
        if (rc) return pfail(p, rc, "%s", bpgpu_last_error(d->lanes[0]));
    }
    return pool_spread_gens(p, gens_capacity, party_capacity);
}
int bpgpu_pool_gens_load(bpgpu_pool *p, size_t gens_capacity, size_t party_capacity, const uint8_t *G, const uint8_t *H, const uint8_t B[32],
                         const uint8_t Bb[32]) {
    if (!p || !G || !H || !B || !Bb || gens_capacity == 0 || party_capacity == 0) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(p->mu);
    for (pool_dev *d : p->devs) {
        if (d->lanes.empty()) return BPGPU_ERR_INVALID_ARG;
        const int rc = bpgpu_gens_load(d->lanes[0], gens_capacity, party_capacity, G, H, B, Bb);
        if (rc) return pfail(p, rc, "%s", bpgpu_last_error(d->lanes[0]));
    }
    return pool_spread_gens(p, gens_capacity, party_capacity);
}

// ---- host pointers, synchronous ---------------------------------------------------------------------------------
// Workers: few threads per device (default 2), each driving its share of the lanes ASYNCHRONOUSLY: stage a slice into a lane's
// pinned block and enqueue it (bpgpu_rangeproof_verify_batch_submit), move on to the next slice on the next lane, collect a lane
// (bpgpu_ctx_collect) only when it is needed again or at the end.  One thread thereby keeps many chains in flight -- the waiting
// is spin-waiting, so the thread count must stay well below the cores the process may use (8 devices x 2 workers = 16 threads;
// one spinning thread per lane, the first design, throttled itself under a 16-core quota: 23 ms instead of 13 for 65536 proofs).
static void ensure_workers(bpgpu_pool *p, pool_dev *d) {
    if (!d->workers.empty()) return;
    size_t w = p->host_workers ? p->host_workers : 2;
    if (w > d->lanes.size()) w = d->lanes.size();
    d->tasks.resize(w);
    for (size_t i = 0; i < d->lanes.size(); i++) bpgpu_ctx_set_option(d->lanes[i], "host_sync_blocking", 0);
    for (size_t i = 0; i < w; i++) d->workers.emplace_back(worker_main, d, i);
}

int bpgpu_pool_rangeproof_verify(bpgpu_pool *p, size_t n, size_t m, size_t nbatch, const uint8_t *proofs, size_t proof_len, const uint8_t *commitments,
                                 const uint8_t *label, size_t label_len, const uint8_t *rng64, uint8_t *verdict, uint8_t *msm_out) {
    if (!p || (nbatch && (!proofs || !verdict || (m && !commitments))) || (label_len && !label)) return BPGPU_ERR_INVALID_ARG;
    if (nbatch == 0) return BPGPU_OK;
    std::lock_guard<std::mutex> lk(p->mu);   // one pool call at a time (its slices use every worker anyway)
    const size_t ndev = p->devs.size();
    struct shared_state {
        std::mutex mu;
        std::condition_variable cv;
        size_t left = 0;
        int rc = 0;
        std::string err;
    } st;
    size_t n_jobs = 0;
    for (size_t di = 0; di < ndev; di++) {
        pool_dev *d = p->devs[di];
        ensure_workers(p, d);
        const size_t lo = nbatch * di / ndev, hi = nbatch * (di + 1) / ndev;   // contiguous shard of this device
        if (hi == lo) continue;
        size_t S = p->slice_proofs;
        if (!S) {
            // wide slices: a launch chain costs ~1 ms of latency however narrow it is, so few chains of up to 4096 proofs beat many
            // narrow ones (measured, 16384 proofs: slices of 512 / 1024 / 2048 / 4096 -> 2.2 / 3.6 / 4.5 / 4.7 M/s); a small call is
            // still cut in two so that the second slice's staging copy overlaps the first one's chain
            const size_t T = hi - lo;
            S = (T + 1) / 2;
            S = (S + 63) & ~(size_t)63;
            if (S < 2048) S = 2048;
            if (S > 4096) S = 4096;
        }
        const size_t W = d->workers.size(), n_slices = (hi - lo + S - 1) / S;
        // a call of a few thousand proofs is one or two chains alone on the device: latency forms; a large one keeps the device full
        const int busy = (hi - lo) > p->latency_proofs ? 1 : 0;
        for (bpgpu_ctx *lc : d->lanes) bpgpu_internal_set_busy_hint(lc, busy);
        for (size_t w = 0; w < W && w < n_slices; w++) {
            // worker w: slices w, w + W, ...  on lanes w, w + W, ... (a lane always sees the same slice widths: its buffers are sized once)
            auto job = [=, &st](bpgpu_ctx *) {
                std::vector<bpgpu_ctx *> mine;
                for (size_t l = w; l < d->lanes.size(); l += W) mine.push_back(d->lanes[l]);
                std::vector<char> busy(mine.size(), 0);
                int rc = 0;
                std::string err;
                size_t k = 0;
                for (size_t sl = w; sl < n_slices && !rc; sl += W, k++) {
                    const size_t a = lo + sl * S, cnt = hi - a < S ? hi - a : S, li = k % mine.size();
                    bpgpu_ctx *c = mine[li];
                    if (busy[li]) {   // the lane still carries an earlier slice: deliver that one first
                        rc = bpgpu_ctx_collect(c);
                        busy[li] = 0;
                        if (rc) {
                            err = bpgpu_last_error(c);
                            break;
                        }
                    }
                    rc = bpgpu_rangeproof_verify_batch_submit(c, n, m, cnt, proofs + a * proof_len, proof_len, commitments ? commitments + a * m * 32 : nullptr, label,
                                                              label_len, rng64 ? rng64 + a * 64 : nullptr, verdict + a, msm_out ? msm_out + a * 32 : nullptr);
                    if (rc) err = bpgpu_last_error(c);
                    else busy[li] = 1;
                }
                for (size_t li = 0; li < mine.size(); li++)
                    if (busy[li]) {   // (also after an error: nothing stays in flight behind the caller's back)
                        const int rc2 = bpgpu_ctx_collect(mine[li]);
                        if (rc2 && !rc) {
                            rc = rc2;
                            err = bpgpu_last_error(mine[li]);
                        }
                    }
                std::lock_guard<std::mutex> g(st.mu);
                if (rc && !st.rc) {
                    st.rc = rc;
                    st.err = err;
                }
                if (--st.left == 0) st.cv.notify_all();
            };
            {
                std::lock_guard<std::mutex> g(st.mu);
                st.left++;
            }
            n_jobs++;
            {
                std::lock_guard<std::mutex> g(d->tmu);
                d->tasks[w].push_back(std::move(job));
            }
            d->tcv.notify_all();
        }
    }
    if (n_jobs) {
        std::unique_lock<std::mutex> g(st.mu);
        st.cv.wait(g, [&] { return st.left == 0; });
    }
    if (st.rc) return pfail(p, st.rc, "%s", st.err.c_str());
    return BPGPU_OK;
}

// ---- device pointers, asynchronous ------------------------------------------------------------------------------
static int flush_dev(bpgpu_pool *p, pool_dev *d, bool one_chain = false) {
    if (d->pending.empty()) return BPGPU_OK;
    std::vector<dev_item> items;
    items.swap(d->pending);
    const size_t T = d->pending_proofs;
    d->pending_proofs = 0;
    bool was_idle = false;
    // an idle pool starts again at lane 0: a caller that sends bursts keeps hitting the same few lanes, whose arenas and cached
    // work decompositions already have the right size
    {
        bool idle = true;
        for (size_t l = 0; l < d->used_lanes && idle; l++) idle = bpgpu_internal_idle(d->lanes[l]);
        if (idle) d->next_lane = d->used_lanes = 0;
        was_idle = idle;
    }
    // number of chains: about T / coalesce_proofs, at most one per lane; a chain takes consecutive items of one shape
    size_t G = (T + p->coalesce_proofs / 2) / p->coalesce_proofs;
    if (G < 1) G = 1;
    // a burst that fits two chains takes two: with the one-lane Horner chains aside, 20 x 1024 from idle measured 5.6 / 5.6 / 5.8 M/s as four
    // chains of 5120 and 5.9 ... 6.2 / 5.7 ... 5.8 / 6.0 as two of 10240 on three boxes; 40 x 1024 prefers eight of 5120 (6.05 against 5.85 as
    // four of 10240), 8 x 1024 two of 4096 (4.7 against 4.3 ... 4.6 as one) -- profiles/r03/coalesce_sweep_after_horner_aside.txt
    if (G > 2 && T <= p->pair_limit_proofs) G = 2;
    if (one_chain) G = 1;   // a chain's worth has accumulated while the caller is still submitting: it goes out now, as it is
    if (G > d->lanes.size()) G = d->lanes.size();
    size_t per = (T + G - 1) / G;
    if (per > p->max_chain_proofs) per = p->max_chain_proofs;
    // table-walk workgroups per proof block (bpgpu.hip pick_splits): a few chains alone on the device want many small
    // workgroups (short tail: measured on 20 x 1024 from idle, 5 chains: 16 splits 5.06 M/s, 32: 5.24, 64: 5.21), a full
    // pipeline wants few (less reduction work) -- aim at ~16 k wavefronts of table walk per flush
    const size_t n_chains = (T + per - 1) / per;
    uint32_t hint = (uint32_t)(16384 / (n_chains * ((per + 63) / 64)));
    hint = (hint + 7) & ~7u;
    if (hint < 16) hint = 16;
    if (hint > 64) hint = 64;
    int rc_all = BPGPU_OK;
    std::vector<rp_seg> segs;
    size_t i = 0, off = 0;   // item i, `off` proofs of it already placed
    while (i < items.size()) {
        const dev_item &head = items[i];
        bpgpu_ctx *c = d->lanes[d->next_lane++ % d->lanes.size()];
        if (d->used_lanes < d->lanes.size() && d->next_lane > d->used_lanes) d->used_lanes = d->next_lane < d->lanes.size() ? d->next_lane : d->lanes.size();
        if (!bpgpu_internal_rp_coalescible(c, head.n, head.m, head.proof_len)) {
            // malformed length / parameter error / missing generators: the ordinary entry point reports it proof by proof
            const int rc = bpgpu_rangeproof_verify_batch_dev(c, head.n, head.m, head.nbatch, head.proofs, head.proof_len, head.coms,
                                                             (const uint8_t *)head.label.data(), head.label.size(), head.rng, head.verdict, head.msm, nullptr);
            if (rc && !rc_all) rc_all = pfail(p, rc, "%s", bpgpu_last_error(c));
            i++;
            off = 0;
            continue;
        }
        segs.clear();
        uint32_t filled = 0;
        bool any_msm = false;
        while (i < items.size() && filled < per && items[i].same_shape(head)) {
            const dev_item &it = items[i];
            size_t take = it.nbatch - off;
            if (take > per - filled) take = per - filled;
            rp_seg sg;
            sg.proofs = it.proofs + off * it.proof_len;
            sg.commitments = it.coms + off * it.m * 32;
            sg.rng64 = it.rng ? it.rng + off * 64 : nullptr;
            sg.verdict = it.verdict + off;
            sg.msm_out = it.msm ? (uint32_t *)(it.msm + off * 32) : nullptr;
            sg.first = filled;
            sg.count = (uint32_t)take;
            any_msm = any_msm || it.msm;
            segs.push_back(sg);
            filled += (uint32_t)take;
            off += take;
            if (off == it.nbatch) {
                i++;
                off = 0;
            }
        }
        bpgpu_internal_set_busy_hint(c, (was_idle && T <= p->latency_proofs) ? 0 : 1);
        const int rc = bpgpu_internal_rp_verify_segs(c, head.n, head.m, head.proof_len, (const uint8_t *)head.label.data(), head.label.size(), segs.data(),
                                                     (uint32_t)segs.size(), any_msm, hint);
        if (rc && !rc_all) rc_all = pfail(p, rc, "%s", bpgpu_last_error(c));
        p->stat_chains++;
        p->stat_chain_proofs += filled;
        p->stat_last_splits = hint;
    }
    return rc_all;
}

int bpgpu_pool_rangeproof_submit_dev(bpgpu_pool *p, int dev_index, size_t n, size_t m, size_t nbatch, const void *d_proofs, size_t proof_len,
                                     const void *d_commitments, const uint8_t *label, size_t label_len, const void *d_rng64, void *d_verdict,
                                     void *d_msm_out) {
    if (!p || dev_index < 0 || dev_index >= (int)p->devs.size() || (label_len && !label)) return BPGPU_ERR_INVALID_ARG;
    if (nbatch == 0) return BPGPU_OK;
    if (!d_proofs || !d_verdict || (m && !d_commitments)) return BPGPU_ERR_INVALID_ARG;
    if (((uintptr_t)d_proofs | (uintptr_t)d_commitments | (uintptr_t)d_rng64 | (uintptr_t)d_msm_out) & 3)
        return pfail(p, BPGPU_ERR_INVALID_ARG, "device buffers must be 4-byte aligned");
    if (nbatch > 0x7fffffffu / 64) return pfail(p, BPGPU_ERR_INVALID_ARG, "batch too large");
    std::lock_guard<std::mutex> lk(p->mu);
    pool_dev *d = p->devs[dev_index];
    dev_item it;
    it.n = n;
    it.m = m;
    it.nbatch = nbatch;
    it.proof_len = proof_len;
    it.proofs = (const uint8_t *)d_proofs;
    it.coms = (const uint8_t *)d_commitments;
    it.rng = (const uint8_t *)d_rng64;
    it.verdict = (uint8_t *)d_verdict;
    it.msm = (uint8_t *)d_msm_out;
    it.label.assign((const char *)label, label_len);
    d->pending.push_back(std::move(it));
    d->pending_proofs += nbatch;
    const size_t limit = p->auto_flush_items ? p->auto_flush_items : d->lanes.size();
    if (d->pending.size() >= limit) return flush_dev(p, d);
    if (p->auto_flush_proofs && d->pending_proofs >= p->auto_flush_proofs) return flush_dev(p, d, true);
    return BPGPU_OK;
}

int bpgpu_pool_flush(bpgpu_pool *p) {
    if (!p) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(p->mu);
    int rc_all = BPGPU_OK;
    for (pool_dev *d : p->devs) {
        const int rc = flush_dev(p, d);
        if (rc && !rc_all) rc_all = rc;
    }
    return rc_all;
}

int bpgpu_pool_wait(bpgpu_pool *p) {
    if (!p) return BPGPU_ERR_INVALID_ARG;
    int rc_all = bpgpu_pool_flush(p);
    std::lock_guard<std::mutex> lk(p->mu);
    for (pool_dev *d : p->devs)
        for (bpgpu_ctx *c : d->lanes) {
            const int rc = bpgpu_synchronize(c);
            if (rc && !rc_all) rc_all = pfail(p, rc, "%s", bpgpu_last_error(c));
        }
    return rc_all;
}

}  // extern "C"
