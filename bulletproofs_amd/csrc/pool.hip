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
//     the lanes round-robin -- a burst of small batches is served as a few wide chains instead of many narrow ones;
//   * bpgpu_pool_rangeproof_verify_ts / _submit_ts (host pointers, the reference's literal call shape: a few proofs per call,
//     each with its own `transcript: &mut Transcript`, from any number of threads at once): the COMBINING QUEUE below -- callers
//     copy their proofs into the open staging buffer of their (shape, transcript position) class; a buffer leaves as one launch
//     chain when it is full or when its deadline expires; every caller is woken when ITS proofs are done.
// No CPU fallback: creation fails without a device.
#include <hip/hip_runtime.h>
#include <linux/futex.h>
#include <pthread.h>
#include <sched.h>
#include <sys/prctl.h>
#include <sys/resource.h>
#include <sys/random.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bpgpu.h"
#include "hostrng.h"
#include "rangeproof.h"

using namespace bp;

static inline uint64_t now_ns() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

bool bpgpu_internal_rp_coalescible(bpgpu_ctx *c, size_t n, size_t m, size_t proof_len);
bool bpgpu_internal_idle(bpgpu_ctx *c);
void bpgpu_internal_set_busy_hint(bpgpu_ctx *c, int busy);
int bpgpu_internal_rp_verify_segs(bpgpu_ctx *c, size_t n, size_t m, size_t proof_len, const uint8_t *label, size_t label_len, const rp_seg *segs,
                                  uint32_t nseg, bool any_msm, uint32_t splits_hint, int busy, bool rlc);
void *bpgpu_internal_stream(bpgpu_ctx *c);
int bpgpu_internal_rp_reserve(bpgpu_ctx *c, size_t n, size_t m, size_t proof_len, size_t nbatch_max);
extern "C" int bpgpu_internal_release_tables(bpgpu_ctx *c);
int bpgpu_internal_rp_verify_chain(bpgpu_ctx *c, size_t n, size_t m, size_t nbatch, const void *d_proofs, size_t proof_len, const void *d_commitments,
                                   const uint8_t *shared_ts, const void *d_ts_in, void *d_ts_out, int ts_uniform, uint32_t pos, uint32_t pos_begin,
                                   uint32_t flags, const void *d_rng64, void *d_verdict, void *d_msm_out, uint32_t splits_hint, int busy);

// The ROCm runtime maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and reads the variable when it
// initialises, i.e. at the process's first HIP call.  Kernels of different lanes only overlap when the lanes sit on different
// queues (measured: 4 queues 3.0 M/s, 16 queues 4.4 M/s, more than 16 collapses), so the library asks for 16 when it is
// loaded, unless the caller chose a value.
namespace {
struct hwq_init {
    bool set_by_library = false;
    hwq_init() {
        if (!getenv("GPU_MAX_HW_QUEUES")) {
            setenv("GPU_MAX_HW_QUEUES", "16", 0);
            set_by_library = true;
        }
    }
} g_hwq_init;
}  // namespace

// How many streams' kernels overlap on this device right now?  Sixteen streams get one single-wavefront kernel each that spins
// for ~1 ms (long against the launch overhead of sixteen launches); with q hardware queues the batch takes ceil(16 / q) x 1 ms.  (Used by bpgpu_pool_create when the library itself
// set GPU_MAX_HW_QUEUES at load time: the variable then says nothing about what the runtime read.)
__global__ void k_pool_spin(uint64_t ticks, uint32_t *sink) {
    const uint64_t t0 = wall_clock64();
    uint32_t x = 0;
    while (wall_clock64() - t0 < ticks) x++;
    if (ticks == ~0ull) *sink = x;
}
static int probe_hw_queues(int device) {
    if (hipSetDevice(device) != hipSuccess) return -1;
    const int NS = 16;
    hipStream_t st[NS];
    for (int i = 0; i < NS; i++)
        if (hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) != hipSuccess) return -1;
    const uint64_t ticks = 100000;   // wall_clock64: 100 MHz -> 1 ms
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {   // the first round also pays for loading the code object
        for (int i = 0; i < NS; i++) hipStreamSynchronize(st[i]);
        const uint64_t t0 = now_ns();
        for (int i = 0; i < NS; i++) hipLaunchKernelGGL(k_pool_spin, dim3(1), dim3(64), 0, st[i], ticks, (uint32_t *)nullptr);
        for (int i = 0; i < NS; i++) hipStreamSynchronize(st[i]);
        const double us = (double)(now_ns() - t0) / 1000.0;
        if (us < best) best = us;
    }
    for (int i = 0; i < NS; i++) hipStreamDestroy(st[i]);
    if (hipGetLastError() != hipSuccess) return -1;
    const double rounds = best / 1000.0;   // ~1 with >= 16 queues, ~2 with 8, ~4 with 4
    int q = (int)(16.0 / (rounds < 1.0 ? 1.0 : rounds) + 0.5);
    return q < 1 ? 1 : q;
}

namespace {

// std::atomic<uint32_t> as a futex word (C++17: no atomic::wait yet)
inline void futex_wait(std::atomic<uint32_t> *a, uint32_t expected) { syscall(SYS_futex, (uint32_t *)a, FUTEX_WAIT_PRIVATE, expected, nullptr, nullptr, 0); }
inline void futex_wake_all(std::atomic<uint32_t> *a) { syscall(SYS_futex, (uint32_t *)a, FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0); }

// (the batching challenge's randomness when the caller brings none: hostrng.h)
using bp::fast_random;

// A ticket is either a request of the combining queue (host pointers) or a submitted device-pointer batch
enum { TK_COMBINE = 0x7c0b, TK_DEVICE = 0x7de0 };
struct ev_holder {   // one recorded event, shared by the tickets of the batches a chain carried
    hipEvent_t ev = nullptr;
    ~ev_holder() {
        if (ev) hipEventDestroy(ev);
    }
};
struct pool_dev;
struct dev_ticket {
    uint32_t kind = TK_DEVICE;
    pool_dev *d = nullptr;
    size_t unissued = 0;                              // proofs of the batch no chain has taken yet (under bpgpu_pool::mu)
    std::vector<std::shared_ptr<ev_holder>> done;     // recorded behind every chain that carries a piece of the batch
    int rc = 0;
    std::string err;
};

struct dev_item {   // a submitted device-pointer batch waiting for the next flush
    size_t n, m, nbatch, proof_len;
    const uint8_t *proofs, *coms, *rng;
    uint8_t *verdict, *msm;
    std::string label;
    std::shared_ptr<ev_holder> ready;   // recorded on the producer's stream at submission: the chain waits for it (may be empty)
    dev_ticket *ticket = nullptr;       // may be null
    bool rlc = false;                   // batch-combined check (bpgpu_pool_rangeproof_submit_rlc_dev): `msm` is then the item's 33-byte batch_out
    bool same_shape(const dev_item &o) const { return n == o.n && m == o.m && proof_len == o.proof_len && rlc == o.rlc && label == o.label; }
};

// ---- combining queue ---------------------------------------------------------------------------------------------------
// What a launch chain can share: the shape, and how its transcripts start --
//   CK_SHARED  every proof from ONE 208-byte state (Transcript::new(label) of a common label, or a transcript the callers
//              share), no states handed back: nothing per proof to upload;
//   CK_UNIFORM one state per proof in / out, all at the same STROBE position (pos, pos_begin, cur_flags): the per-shape script
//              (rp_script.h) still applies, only the sponge words differ;
//   CK_MIXED   one state per proof at whatever position: the byte-wise replay.  Catch-all when too many position classes are open.
enum { CK_SHARED = 0, CK_UNIFORM = 1, CK_MIXED = 2 };
struct comb_key {
    uint32_t n = 0, m = 0, proof_len = 0, mode = 0;
    uint32_t pos = 0, pos_begin = 0, flags = 0;      // CK_UNIFORM
    uint8_t shared[BPGPU_TRANSCRIPT_BYTES] = {0};    // CK_SHARED
    bool operator==(const comb_key &o) const {
        if (n != o.n || m != o.m || proof_len != o.proof_len || mode != o.mode) return false;
        if (mode == CK_UNIFORM) return pos == o.pos && pos_begin == o.pos_begin && flags == o.flags;
        if (mode == CK_SHARED) return memcmp(shared, o.shared, sizeof shared) == 0;
        return true;
    }
};

struct comb_req;
enum { CB_FREE = 0, CB_OPEN, CB_SEALED, CB_ISSUING, CB_ISSUED, CB_DONE };
// One staging buffer + the lane that runs its chain.  Inputs [proofs | commitments | rng | transcripts in] and outputs
// [verdicts | transcripts out | encodings] sit at the same offsets of a pinned host block and of a device block.
struct comb_buf {
    pool_dev *dev = nullptr;
    bpgpu_ctx *ctx = nullptr;
    std::atomic<int> poison{0};
    uint32_t reserved_n = 0, reserved_m = 0, reserved_len = 0, reserved_cap = 0;   // shape / width the lane's buffers were last sized for
    hipEvent_t done_ev = nullptr;
    char *h = nullptr, *d = nullptr;
    size_t mem_cap = 0;
    comb_key key;
    uint32_t cap = 0, cap_max = 0;
    size_t off_p = 0, off_c = 0, off_r = 0, off_t = 0, in_bytes = 0, off_v = 0, off_to = 0, off_m = 0, total = 0;
    int st = CB_FREE;                       // under pool_dev::cmu
    uint32_t reserved = 0;                  // proofs handed out (under cmu)
    std::atomic<uint32_t> written{0};       // proofs whose inputs are in place
    std::atomic<uint32_t> refs{0};          // pieces whose results have not been taken yet
    std::atomic<uint32_t> phase{0};         // futex word: 0 until the chain's results are in `h`
    bool any_msm = false;
    uint64_t t_first = 0, t_last = 0;       // arrival of the first / the latest piece (ns)
    int rc = 0;
    std::string err;
    struct apiece {
        comb_req *req;
        uint32_t first, count;
        size_t off;
    };
    std::vector<apiece> async_pieces;       // pieces of tickets: delivered by the service thread
};

// one call (bpgpu_pool_rangeproof_verify_ts) or one ticket (bpgpu_pool_rangeproof_submit_ts)
struct comb_req {
    uint32_t kind = TK_COMBINE;   // (first member of both ticket types)
    size_t n = 0, m = 0, nbatch = 0, proof_len = 0;
    const uint8_t *proofs = nullptr, *coms = nullptr, *rng = nullptr, *ts_in = nullptr;   // ts_in: per proof, or null (key.shared)
    uint8_t *verdict = nullptr, *msm = nullptr, *ts_out = nullptr;
    bool async = false;
    struct piece {
        comb_buf *b;
        uint32_t first, count;
        size_t off;
    };
    std::vector<piece> pieces;              // synchronous calls: collected by the caller itself
    size_t next_piece = 0;
    std::atomic<uint32_t> left{1};          // tickets: pieces not delivered yet (+1 while the request is being placed); futex word
    std::atomic<uint32_t> waiting{0};
    std::mutex emu;                         // guards rc / err of a ticket (service thread vs. caller)
    int rc = 0;
    std::string err;
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
    // combining queue: its own lanes (a lane whose staging buffer callers are writing into cannot take a flush's chain meanwhile),
    // one service thread (seals buffers by deadline, issues their chains, notices completions), one context for the requests no
    // chain can take (malformed lengths, parameter errors: reported per proof by the ordinary entry point)
    std::vector<comb_buf *> cbufs;
    bpgpu_ctx *misc = nullptr;
    std::mutex misc_mu;
    std::mutex cmu;
    std::condition_variable ccv, free_cv;
    std::thread svc;
    bool svc_running = false, cstop = false;
    uint64_t stat_chains = 0, stat_proofs = 0, stat_requests = 0;
    uint64_t stat_issue_ns = 0, stat_complete_ns = 0, stat_polls = 0;   // service thread: time spent issuing / completing, loop count
    uint32_t recent_K = 0;                  // width of the chain issued last (sizes the next buffer, comb_place)
};

}  // namespace

struct bpgpu_pool {
    std::vector<pool_dev *> devs;
    std::mutex mu;        // serialises the pool's own state (pending lists, options); lane contexts have their own locks
    size_t coalesce_proofs = 5120;   // target width of a coalesced launch chain
    size_t pair_limit_proofs = 24576;   // a flush of up to this many proofs is issued as at most two chains (flush_dev)
    size_t max_chain_proofs = 16384; // never wider than this (arena of a lane: ~55 KB per proof)
    size_t slice_proofs = 0;         // host-pointer calls: proofs per slice (0 = automatic)
    size_t latency_proofs = 6144;    // a host call / a flush on an idle device of up to this many proofs is "alone": its chains take the latency forms
    size_t auto_flush_items = 0;     // flush by itself once this many items wait on a device (0 = lanes)
    size_t auto_flush_proofs = 0;    // ... or once this many proofs wait: they go out as ONE chain while the caller keeps submitting (0 = off)
    size_t host_workers = 0;
    // combining queue
    std::atomic<uint64_t> combine_wait_ns{100000};    // a buffer leaves at the latest this long after its first proof arrived ...
    std::atomic<uint64_t> combine_quiet_ns{20000};    // ... or when nothing has joined it for this long
    std::atomic<uint64_t> combine_poll_ns{15000};     // the service thread's polling period while anything is open or in flight
    std::atomic<uint32_t> combine_max_open{4};        // position classes with a buffer of their own; further classes share a CK_MIXED buffer
    std::atomic<uint32_t> combine_busy_chains{2};     // a chain issued beside this many others (in flight or waiting) takes the throughput forms
    std::atomic<uint32_t> combine_inflight{6};        // deadlines seal buffers only while fewer chains than this are in flight: beyond, load widens the chains
    std::atomic<uint64_t> combine_max_age_ns{1500000}; // ... but no proof waits longer than this for its chain to be issued
    std::atomic<int> host_path{1};                    // bpgpu_pool_rangeproof_verify: 1 = through the combining queue, 0 = the slicing workers of round 3
    std::atomic<uint32_t> rr_dev{0};
    std::atomic<uint64_t> gens_epoch{1};
    // statistics of the coalesced path (get_option "stat_chains" / "stat_chain_proofs" / "stat_last_splits"; set "stat_reset")
    uint64_t stat_chains = 0, stat_chain_proofs = 0, stat_last_splits = 0;
};

// Errors are per calling thread: bpgpu_pool_last_error returns what the LAST pool call of THIS thread reported (any number of
// threads may be inside the pool at once).
static thread_local std::string t_pool_err;
static int pfail(bpgpu_pool *, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    t_pool_err = buf;
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

// every context of a device: submit lanes, the combining queue's lanes, the odd-jobs context
static std::vector<bpgpu_ctx *> all_ctxs(pool_dev *d) {
    std::vector<bpgpu_ctx *> v(d->lanes);
    for (comb_buf *b : d->cbufs) v.push_back(b->ctx);
    if (d->misc) v.push_back(d->misc);
    return v;
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
    if (lanes_per_device > 4 && g_hwq_init.set_by_library) {
        // The variable reads 16 because THIS library set it when it was loaded -- which only counts if the runtime had not read it
        // before (a process that touched HIP first runs on the default 4 queues whatever the variable says now).  Ask the device.
        const int qs = probe_hw_queues(devices[0]);
        if (qs > 0 && qs < 6) {   // (4 queues -- the runtime's default -- measure 4 rounds; 8 queues 2; timing noise stays well inside)
            fprintf(stderr, "libbpgpu: GPU_MAX_HW_QUEUES was set by libbpgpu at load time, but HIP had been initialised before: kernels of only "
                            "~%d streams overlap.  Export GPU_MAX_HW_QUEUES=16 before the process's first HIP call.\n", qs);
            return BPGPU_ERR_HW_QUEUES;
        }
    }
    // lanes of the combining queue (bpgpu_pool_rangeproof_verify_ts and the host-pointer call): BPGPU_COMBINE_LANES, default 12
    int n_comb = 12;
    if (const char *e = getenv("BPGPU_COMBINE_LANES")) n_comb = atoi(e);
    if (n_comb < 2) n_comb = 2;
    if (n_comb > 64) n_comb = 64;
    bpgpu_pool *p = new bpgpu_pool();
    for (int i = 0; i < ndev; i++) {
        pool_dev *d = new pool_dev();
        d->device = devices[i];
        p->devs.push_back(d);
        for (int l = 0; l < lanes_per_device + n_comb + 1; l++) {
            bpgpu_ctx *c = nullptr;
            const int rc = bpgpu_ctx_create(devices[i], &c);
            if (rc) {
                bpgpu_pool_destroy(p);
                return rc;
            }
            if (l < lanes_per_device) d->lanes.push_back(c);
            else if (l < lanes_per_device + n_comb) {
                comb_buf *b = new comb_buf();
                b->ctx = c;
                b->dev = d;
                d->cbufs.push_back(b);
                if (hipSetDevice(devices[i]) != hipSuccess || hipEventCreateWithFlags(&b->done_ev, hipEventDisableTiming) != hipSuccess) {
                    bpgpu_pool_destroy(p);
                    return BPGPU_ERR_HIP;
                }
            } else d->misc = c;
        }
    }
    *out = p;
    return BPGPU_OK;
}

static void stop_service(pool_dev *d);
void bpgpu_pool_destroy(bpgpu_pool *p) {
    if (!p) return;
    for (pool_dev *d : p->devs) {
        stop_workers(d);
        stop_service(d);
        for (bpgpu_ctx *c : d->lanes) bpgpu_ctx_destroy(c);
        for (comb_buf *b : d->cbufs) {
            bpgpu_ctx_destroy(b->ctx);   // (synchronises the device: nothing of the buffer is in flight afterwards)
            if (b->done_ev) hipEventDestroy(b->done_ev);
            if (b->h) hipHostFree(b->h);
            if (b->d) hipFree(b->d);
            delete b;
        }
        if (d->misc) bpgpu_ctx_destroy(d->misc);
        delete d;
    }
    delete p;
}

const char *bpgpu_pool_last_error(bpgpu_pool *p) { return p ? t_pool_err.c_str() : "null pool"; }

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
        for (pool_dev *d : p->devs) {
            std::lock_guard<std::mutex> g(d->cmu);
            d->stat_chains = d->stat_proofs = d->stat_requests = 0;
            d->stat_issue_ns = d->stat_complete_ns = d->stat_polls = 0;
        }
        return BPGPU_OK;
    }
    if (!strcmp(key, "combine_wait_us") || !strcmp(key, "combine_quiet_us") || !strcmp(key, "combine_poll_us")) {
        if (value < 1 || value > 1000000) return pfail(p, BPGPU_ERR_INVALID_ARG, "%s out of range", key);
        (key[8] == 'w' ? p->combine_wait_ns : key[8] == 'q' ? p->combine_quiet_ns : p->combine_poll_ns) = (uint64_t)value * 1000;
        return BPGPU_OK;
    }
    if (!strcmp(key, "combine_busy_chains") || !strcmp(key, "combine_inflight")) {
        if (value < 0 || value > 64) return pfail(p, BPGPU_ERR_INVALID_ARG, "%s out of range", key);
        (key[8] == 'b' ? p->combine_busy_chains : p->combine_inflight) = (uint32_t)value;
        return BPGPU_OK;
    }
    if (!strcmp(key, "combine_max_age_us")) {
        if (value < 1 || value > 10000000) return pfail(p, BPGPU_ERR_INVALID_ARG, "%s out of range", key);
        p->combine_max_age_ns = (uint64_t)value * 1000;
        return BPGPU_OK;
    }
    if (!strcmp(key, "combine_max_open")) {
        if (value < 1 || value > 64) return pfail(p, BPGPU_ERR_INVALID_ARG, "combine_max_open out of range");
        p->combine_max_open = (uint32_t)value;
        return BPGPU_OK;
    }
    if (!strcmp(key, "host_path_combining")) {
        p->host_path = value != 0;
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
        for (bpgpu_ctx *c : all_ctxs(d)) {
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
    else if (!strcmp(key, "combine_wait_us")) *value = (int64_t)(p->combine_wait_ns / 1000);
    else if (!strcmp(key, "combine_quiet_us")) *value = (int64_t)(p->combine_quiet_ns / 1000);
    else if (!strcmp(key, "combine_poll_us")) *value = (int64_t)(p->combine_poll_ns / 1000);
    else if (!strcmp(key, "combine_max_open")) *value = (int64_t)p->combine_max_open;
    else if (!strcmp(key, "combine_busy_chains")) *value = (int64_t)p->combine_busy_chains;
    else if (!strcmp(key, "combine_inflight")) *value = (int64_t)p->combine_inflight;
    else if (!strcmp(key, "combine_max_age_us")) *value = (int64_t)(p->combine_max_age_ns / 1000);
    else if (!strcmp(key, "stat_svc_issue_us") || !strcmp(key, "stat_svc_complete_us") || !strcmp(key, "stat_svc_polls")) {
        uint64_t v = 0;
        for (pool_dev *d : p->devs) {
            std::lock_guard<std::mutex> g(d->cmu);
            v += key[9] == 'i' ? d->stat_issue_ns / 1000 : key[9] == 'c' ? d->stat_complete_ns / 1000 : d->stat_polls;
        }
        *value = (int64_t)v;
    }
    else if (!strcmp(key, "combine_lanes")) *value = p->devs.empty() ? 0 : (int64_t)p->devs[0]->cbufs.size();
    else if (!strcmp(key, "host_path_combining")) *value = p->host_path;
    else if (!strcmp(key, "stat_combined_chains") || !strcmp(key, "stat_combined_proofs") || !strcmp(key, "stat_combined_requests")) {
        uint64_t v = 0;
        for (pool_dev *d : p->devs) {
            std::lock_guard<std::mutex> g(d->cmu);
            v += key[14] == 'c' ? d->stat_chains : key[14] == 'p' ? d->stat_proofs : d->stat_requests;
        }
        *value = (int64_t)v;
    }
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
        const std::vector<bpgpu_ctx *> ctxs = all_ctxs(d);
        for (size_t l = 1; l < ctxs.size(); l++) {
            rc = bpgpu_gens_load(ctxs[l], gens_capacity, party_capacity, G.data(), H.data(), B, Bb);
            if (rc) return pfail(p, rc, "%s", bpgpu_last_error(ctxs[l]));
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
    p->gens_epoch.fetch_add(1);
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
    p->gens_epoch.fetch_add(1);
    return pool_spread_gens(p, gens_capacity, party_capacity);
}

// A second shape with a window table of its own (bpgpu_gens_add_shape) on every lane of every device: all lanes let go of their tables
// first -- the old and the new pair of tables never coexist in HBM --, then the first lane of a device builds both and the others share.
// Call it while the pool is idle (it waits for every lane's streams).
int bpgpu_pool_gens_add_shape(bpgpu_pool *p, size_t n2, size_t m2) {
    if (!p || n2 == 0 || m2 == 0) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(p->mu);
    for (pool_dev *d : p->devs) {
        const std::vector<bpgpu_ctx *> ctxs = all_ctxs(d);
        for (bpgpu_ctx *c : ctxs) {
            const int rc = bpgpu_internal_release_tables(c);
            if (rc) return pfail(p, rc, "%s", bpgpu_last_error(c));
        }
        for (bpgpu_ctx *c : ctxs) {
            const int rc = bpgpu_gens_add_shape(c, n2, m2);
            if (rc) return pfail(p, rc, "%s", bpgpu_last_error(c));
        }
    }
    p->gens_epoch.fetch_add(1);
    return BPGPU_OK;
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
    if (p->host_path) {
        // through the combining queue (below): any number of threads may be in here at once, their proofs share launch chains
        if (label_len > 0xffffffffu) return BPGPU_ERR_INVALID_ARG;
        uint8_t st0[BPGPU_TRANSCRIPT_BYTES];
        bpgpu_transcript_new(label, label_len, st0);
        return bpgpu_pool_rangeproof_verify_ts(p, n, m, nbatch, proofs, proof_len, commitments, st0, 0, rng64, verdict, msm_out, nullptr);
    }
    std::lock_guard<std::mutex> lk(p->mu);   // option "host_path_combining" = 0: round 3's slicing workers, one pool call at a time
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

}  // extern "C"

// ---- combining queue -----------------------------------------------------------------------------------------------
// The reference's API is one proof per call, synchronous, from as many threads as the caller likes
// (RangeProof::verify_multiple / verify_multiple_with_rng, src/range_proof/mod.rs:345-353, 455-470).  One such call cannot fill a
// device and a launch chain costs ~0.6 ms of latency however narrow it is, so calls that arrive close together must share a
// chain.  Per device:
//   * callers (any thread, short lock): find the OPEN staging buffer of their class (comb_key) or open a free one, reserve slots,
//     copy their inputs into the pinned block OUTSIDE the lock, count themselves in (`written`);
//   * the service thread: seals an OPEN buffer when it is full, when its first proof has waited `combine_wait_us`, or when nothing
//     joined for `combine_quiet_us`; issues a sealed buffer whose writers are done as ONE chain on the buffer's lane (copies in,
//     bpgpu_internal_rp_verify_chain, copies out, event); polls the events of the chains in flight;
//   * completion: the buffer's futex word flips, every synchronous caller with a piece in it wakes, takes its own verdicts (and
//     advanced transcripts) out of the pinned block and drops its reference; the last one returns the buffer.  Pieces of tickets
//     (bpgpu_pool_rangeproof_submit_ts) are delivered by the service thread.
// With all lanes busy a buffer simply keeps filling: load widens the chains by itself.
static size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

// Layout of a buffer for `cap` proofs of class `key`.  Inputs: [commitments | rng | transcripts in | proofs] -- the small
// per-proof records first, sized for `cap`; the proofs last, so that ONE copy [0, off_p + K proof_len) carries a chain of K <= cap
// proofs (the unused tail of the small regions rides along: a buffer opened under light traffic has a small `cap`, comb_place).
// Outputs: [verdicts | transcripts out | encodings]: ONE copy [off_v, off_to + 208 K) back.
static size_t cbuf_layout(comb_buf *b, const comb_key &key, uint32_t cap) {
    const bool per = key.mode != CK_SHARED;
    size_t o = 0;
    b->off_c = o, o += up256((size_t)cap * key.m * 32);
    b->off_r = o, o += up256((size_t)cap * 64);
    b->off_t = o, o += per ? up256((size_t)cap * BPGPU_TRANSCRIPT_BYTES) : 0;
    b->off_p = o, o += up256((size_t)cap * key.proof_len);
    b->in_bytes = o;
    b->off_v = o, o += up256(cap);
    b->off_to = o, o += per ? up256((size_t)cap * BPGPU_TRANSCRIPT_BYTES) : 0;
    b->off_m = o, o += up256((size_t)cap * 32);
    b->total = o;
    return o;
}
static int cbuf_configure(bpgpu_pool *p, pool_dev *d, comb_buf *b, const comb_key &key, uint32_t cap, uint32_t cap_max) {
    const size_t need = cbuf_layout(b, key, cap_max);   // the blocks are sized for the widest chain of this shape once and for all
    b->key = key;
    b->cap = cap;
    b->cap_max = cap_max;
    const size_t o = cap == cap_max ? need : (cbuf_layout(b, key, cap), need);
    if (o > b->mem_cap) {   // the first chains of a shape (the calling thread's current device is put back afterwards)
        int prev = -1;
        (void)hipGetDevice(&prev);
        hipError_t e = hipSetDevice(d->device);
        if (e == hipSuccess && b->h) e = hipHostFree(b->h);
        b->h = nullptr;
        if (e == hipSuccess && b->d) e = hipFree(b->d);
        b->d = nullptr;
        b->mem_cap = 0;
        const size_t want = o + o / 4;
        if (e == hipSuccess) e = hipHostMalloc((void **)&b->h, want, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&b->d, want);
        if (prev >= 0) (void)hipSetDevice(prev);
        if (e != hipSuccess) {
            if (b->h) hipHostFree(b->h);
            b->h = nullptr;
            return pfail(p, BPGPU_ERR_HIP, "staging buffers of the combining queue: %s", hipGetErrorString(e));
        }
        b->mem_cap = want;
    }
    return BPGPU_OK;
}

static void cbuf_release(comb_buf *b) {
    pool_dev *d = b->dev;
    {
        std::lock_guard<std::mutex> lk(d->cmu);
        b->st = CB_FREE;
        b->reserved = 0;
        b->written.store(0, std::memory_order_relaxed);
        b->phase.store(0, std::memory_order_relaxed);
        b->any_msm = false;
        b->poison.store(0, std::memory_order_relaxed);
        b->rc = 0;
        b->err.clear();
    }
    d->free_cv.notify_all();
}

// results of proofs [first, first + count) of a finished buffer -> proofs [off, ..) of the request
static void comb_deliver(comb_buf *b, uint32_t first, uint32_t count, comb_req *r, size_t off) {
    if (b->rc) {   // the chain did not run: nothing may read as "verified"
        memset(r->verdict + off, BPGPU_VERDICT_UNDECIDED, count);
        std::lock_guard<std::mutex> g(r->emu);
        if (!r->rc) {
            r->rc = b->rc;
            r->err = b->err;
        }
        return;
    }
    memcpy(r->verdict + off, b->h + b->off_v + first, count);
    if (r->msm) memcpy(r->msm + off * 32, b->h + b->off_m + (size_t)first * 32, (size_t)count * 32);
    if (r->ts_out) memcpy(r->ts_out + off * BPGPU_TRANSCRIPT_BYTES, b->h + b->off_to + (size_t)first * BPGPU_TRANSCRIPT_BYTES, (size_t)count * BPGPU_TRANSCRIPT_BYTES);
}

// a synchronous caller takes the next of its pieces: sleeps until that buffer's chain is done
static void comb_collect_one(comb_req *r) {
    const comb_req::piece pc = r->pieces[r->next_piece++];
    comb_buf *b = pc.b;
    while (b->phase.load(std::memory_order_acquire) == 0) futex_wait(&b->phase, 0);
    comb_deliver(b, pc.first, pc.count, r, pc.off);
    if (b->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) cbuf_release(b);
}

static void comb_issue(bpgpu_pool *p, pool_dev *d, comb_buf *b, uint32_t inflight, bool more_waiting) {
    hipStream_t s = (hipStream_t)bpgpu_internal_stream(b->ctx);
    const comb_key &k = b->key;
    const uint32_t K = b->reserved;
    const bool per = k.mode != CK_SHARED;
    const size_t TS = BPGPU_TRANSCRIPT_BYTES;
    hipError_t e = hipSuccess;
    if (b->poison.load(std::memory_order_acquire)) {   // a writer could not draw its batching challenge: the chain must not run on predictable bytes
        b->rc = BPGPU_ERR_HIP;
        b->err = "getrandom failed";
        return;
    }
    if (b->reserved_n != k.n || b->reserved_m != k.m || b->reserved_len != k.proof_len || b->reserved_cap < b->cap_max) {
        // first chain of this shape on this lane: size the lane's arena for the widest chain now, not in steps on the way up
        (void)bpgpu_internal_rp_reserve(b->ctx, k.n, k.m, k.proof_len, b->cap_max);
        b->reserved_n = k.n, b->reserved_m = k.m, b->reserved_len = k.proof_len, b->reserved_cap = b->cap_max;
    }
    auto cp_in = [&](size_t off, size_t bytes) {
        if (e == hipSuccess && bytes) e = hipMemcpyAsync(b->d + off, b->h + off, bytes, hipMemcpyHostToDevice, s);
    };
    auto cp_out = [&](size_t off, size_t bytes) {
        if (e == hipSuccess && bytes) e = hipMemcpyAsync(b->h + off, b->d + off, bytes, hipMemcpyDeviceToHost, s);
    };
    cp_in(0, b->off_p + (size_t)K * k.proof_len);
    int rc = BPGPU_OK;
    if (e == hipSuccess) {
        // what the chain should expect beside it (rp_chain_forms, pick_splits): a chain that leaves a full buffer, or while others
        // wait or run, takes the throughput forms; a lone small one the latency forms
        const int busy = (inflight + (more_waiting ? 1u : 0u) >= p->combine_busy_chains || K > p->latency_proofs) ? 1 : 0;
        uint32_t hint = (uint32_t)(16384 / ((size_t)(inflight + 1) * ((K + 63) / 64)));
        hint = (hint + 7) & ~7u;
        if (hint < 16) hint = 16;
        if (hint > 64) hint = 64;
        rc = bpgpu_internal_rp_verify_chain(b->ctx, k.n, k.m, K, b->d + b->off_p, k.proof_len, b->d + b->off_c, per ? nullptr : k.shared,
                                            per ? b->d + b->off_t : nullptr, per ? b->d + b->off_to : nullptr, k.mode == CK_UNIFORM, k.pos, k.pos_begin,
                                            k.flags, b->d + b->off_r, b->d + b->off_v, b->any_msm ? b->d + b->off_m : nullptr, hint, busy);
        if (rc) b->err = bpgpu_last_error(b->ctx);
    }
    if (e == hipSuccess && !rc) {
        cp_out(b->off_v, per ? (b->off_to - b->off_v) + (size_t)K * TS : (size_t)K);
        if (b->any_msm) cp_out(b->off_m, (size_t)K * 32);
        if (e == hipSuccess) e = hipEventRecord(b->done_ev, s);
    }
    if (e != hipSuccess) {
        rc = BPGPU_ERR_HIP;
        b->err = std::string("combining queue: ") + hipGetErrorString(e);
        (void)hipGetLastError();
    }
    b->rc = rc;
}

// the chain of buffer `b` is over (or never went out): publish, deliver the tickets' pieces
static void comb_complete(pool_dev *d, comb_buf *b) {
    std::vector<comb_buf::apiece> ap;
    {
        std::lock_guard<std::mutex> lk(d->cmu);
        b->st = CB_DONE;
        ap.swap(b->async_pieces);
        b->refs.fetch_add(1, std::memory_order_relaxed);   // the service thread's own hold while it delivers
    }
    b->phase.store(1, std::memory_order_release);
    futex_wake_all(&b->phase);
    uint32_t drop = 1;
    for (const comb_buf::apiece &a : ap) {
        comb_req *r = a.req;
        comb_deliver(b, a.first, a.count, r, a.off);
        drop++;
        // the ticket's owner frees it once `left` reads 0 -- under emu, so not before this thread is done with it
        std::lock_guard<std::mutex> g(r->emu);
        if (r->left.fetch_sub(1, std::memory_order_seq_cst) == 1 && r->waiting.load(std::memory_order_seq_cst)) futex_wake_all(&r->left);
    }
    if (b->refs.fetch_sub(drop, std::memory_order_acq_rel) == drop) cbuf_release(b);
}

static void svc_main(bpgpu_pool *p, pool_dev *d) {
    (void)hipSetDevice(d->device);
    prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);   // this thread's timed waits are tens of microseconds: the default slack is 50
    // Hundreds of caller threads become runnable whenever a chain ends; the one thread that issues the next chain must not queue
    // behind them for a time slice.  Best effort (needs CAP_SYS_NICE): a real-time class, else a negative nice value.  The thread
    // sleeps whenever it has nothing to do, so it cannot monopolise a core.
    {
        sched_param sp{};
        sp.sched_priority = 1;
        if (pthread_setschedparam(pthread_self(), SCHED_RR, &sp) != 0) (void)setpriority(PRIO_PROCESS, (id_t)syscall(SYS_gettid), -15);
    }
    std::unique_lock<std::mutex> lk(d->cmu);
    std::vector<comb_buf *> to_issue, to_complete;
    for (;;) {
        if (d->cstop) return;
        bool active = false;
        const uint64_t now = now_ns(), wait_ns = p->combine_wait_ns, quiet_ns = p->combine_quiet_ns, max_age_ns = p->combine_max_age_ns;
        const uint32_t target = p->combine_inflight;
        d->stat_polls++;
        uint32_t inflight = 0, waiting = 0;
        for (comb_buf *b : d->cbufs) {
            if (b->st == CB_ISSUED || b->st == CB_ISSUING) inflight++;
            if (b->st == CB_OPEN || b->st == CB_SEALED) waiting++;
        }
        to_issue.clear();
        to_complete.clear();
        for (comb_buf *b : d->cbufs) {
            if (b->st == CB_OPEN) {
                active = true;
                // the deadlines are for an idle-ish device (latency); with `target` chains already running the buffer goes on
                // filling until one of them ends -- load widens the chains by itself
                const bool due = now - b->t_first >= wait_ns || now - b->t_last >= quiet_ns;
                if ((due && inflight < target) || now - b->t_first >= max_age_ns) {
                    b->st = CB_SEALED;
                    inflight++;   // (counts against the target at once: two buffers due in the same pass)
                }
            }
            if (b->st == CB_SEALED) {
                active = true;
                if (b->written.load(std::memory_order_acquire) == b->reserved) {
                    b->st = CB_ISSUING;
                    to_issue.push_back(b);
                }
            } else if (b->st == CB_ISSUED) {
                active = true;
                const hipError_t e = hipEventQuery(b->done_ev);
                if (e != hipErrorNotReady) {
                    if (e != hipSuccess) {
                        b->rc = BPGPU_ERR_HIP;
                        b->err = std::string("combining queue: ") + hipGetErrorString(e);
                    }
                    to_complete.push_back(b);
                } else (void)hipGetLastError();
            }
        }
        if (!to_issue.empty() || !to_complete.empty()) {
            lk.unlock();
            const uint64_t ta = now_ns();
            for (comb_buf *b : to_complete) comb_complete(d, b);
            const uint64_t tb = now_ns();
            uint32_t running = 0;
            for (comb_buf *b : d->cbufs) running += (b->st == CB_ISSUED);   // (st of other buffers: only this thread moves them in / out of ISSUED)
            for (comb_buf *b : to_issue) {
                waiting--;
                comb_issue(p, d, b, running, waiting > 0);
                running++;
                if (b->rc) comb_complete(d, b);   // never went out
                else {
                    std::lock_guard<std::mutex> g(d->cmu);
                    b->st = CB_ISSUED;
                    d->stat_chains++;
                    d->stat_proofs += b->reserved;
                    d->recent_K = b->reserved;
                }
            }
            lk.lock();
            d->stat_complete_ns += tb - ta;
            d->stat_issue_ns += now_ns() - tb;
            continue;   // look again at once: issuing took tens of microseconds
        }
        if (active) d->ccv.wait_for(lk, std::chrono::nanoseconds((uint64_t)p->combine_poll_ns));
        else d->ccv.wait(lk);
    }
}

static void stop_service(pool_dev *d) {
    {
        std::lock_guard<std::mutex> lk(d->cmu);
        d->cstop = true;
    }
    d->ccv.notify_all();
    if (d->svc.joinable()) d->svc.join();
}

// STROBE position class of a 208-byte state
static inline void ts_class(const uint8_t *st, comb_key &k) {
    k.pos = st[200];
    k.pos_begin = st[201];
    k.flags = st[202];
}
static inline bool ts_ok(const uint8_t *st) { return st[200] < 166 && st[201] <= 166; }

// proofs [lo, hi) of the request go to device d's queue
static int comb_place(bpgpu_pool *p, pool_dev *d, comb_req *r, const comb_key &base, const uint8_t *shared, size_t lo, size_t hi) {
    const size_t TS = BPGPU_TRANSCRIPT_BYTES;
    size_t off = lo;
    while (off < hi) {
        comb_key key = base;
        size_t run = hi - off;
        if (key.mode == CK_UNIFORM) {
            const uint8_t *st0 = r->ts_in ? r->ts_in + off * TS : shared;
            ts_class(st0, key);
            if (r->ts_in) {   // the stretch of this request that sits at one STROBE position
                size_t e = off + 1;
                while (e < hi && r->ts_in[e * TS + 200] == st0[200] && r->ts_in[e * TS + 201] == st0[201] && r->ts_in[e * TS + 202] == st0[202]) e++;
                run = e - off;
            }
        }
        comb_buf *b = nullptr;
        uint32_t first = 0, take = 0;
        bool wake = false;
        {
            std::unique_lock<std::mutex> lk(d->cmu);
            if (!d->svc_running) {
                d->svc = std::thread(svc_main, p, d);
                d->svc_running = true;
            }
            uint32_t n_open_classes = 0;
            comb_buf *mixed = nullptr, *freeb = nullptr;
            for (comb_buf *cb : d->cbufs) {
                if (cb->st == CB_FREE) {
                    if (!freeb || cb->mem_cap > freeb->mem_cap) freeb = cb;   // (prefer one whose blocks are already large enough)
                    continue;
                }
                if (cb->st != CB_OPEN) continue;
                if (cb->key == key) b = cb;
                if (cb->key.mode == CK_UNIFORM) n_open_classes++;
                if (cb->key.mode == CK_MIXED && cb->key.n == key.n && cb->key.m == key.m && cb->key.proof_len == key.proof_len) mixed = cb;
            }
            if (!b && key.mode == CK_UNIFORM && !(freeb && n_open_classes < p->combine_max_open) && (mixed || freeb)) {
                // too many position classes open at once (or no buffer left for a new one): the catch-all, replayed byte-wise
                key.mode = CK_MIXED;
                key.pos = key.pos_begin = key.flags = 0;
                run = hi - off;
                b = mixed;
            }
            if (!b) {
                if (!freeb) {   // every lane is filling or running: take a finished piece of our own meanwhile, or wait for a buffer
                    if (!r->async && r->next_piece < r->pieces.size()) {
                        lk.unlock();
                        comb_collect_one(r);
                    } else {
                        d->free_cv.wait_for(lk, std::chrono::microseconds(200));
                    }
                    continue;
                }
                size_t cap_max = p->coalesce_proofs;
                if (cap_max > p->max_chain_proofs) cap_max = p->max_chain_proofs;
                // light traffic opens a small buffer (its staging copy carries the small regions whole), heavy traffic a full-width one
                const size_t want = std::max<size_t>((size_t)4 * d->recent_K, run);
                const size_t cap = want <= 256 ? std::min<size_t>(256, cap_max) : want <= 1024 ? std::min<size_t>(1024, cap_max) : cap_max;
                const int rc = cbuf_configure(p, d, freeb, key, (uint32_t)cap, (uint32_t)cap_max);
                if (rc) return rc;
                b = freeb;
                b->st = CB_OPEN;
                b->t_first = now_ns();
                wake = true;   // the service thread may be asleep with nothing to watch
            }
            first = b->reserved;
            take = (uint32_t)(run < (size_t)(b->cap - first) ? run : (size_t)(b->cap - first));
            b->reserved += take;
            b->t_last = first ? now_ns() : b->t_first;
            b->refs.fetch_add(1, std::memory_order_relaxed);
            if (r->msm) b->any_msm = true;
            if (b->reserved == b->cap) {
                b->st = CB_SEALED;
                wake = true;
            }
            if (r->async) {
                b->async_pieces.push_back({r, first, take, off});
                r->left.fetch_add(1, std::memory_order_relaxed);
            } else {
                r->pieces.push_back({b, first, take, off});
            }
        }
        if (wake) d->ccv.notify_one();
        // ---- inputs into the pinned block (no lock held) ----
        memcpy(b->h + b->off_p + (size_t)first * r->proof_len, r->proofs + off * r->proof_len, (size_t)take * r->proof_len);
        memcpy(b->h + b->off_c + (size_t)first * r->m * 32, r->coms + off * r->m * 32, (size_t)take * r->m * 32);
        if (r->rng) memcpy(b->h + b->off_r + (size_t)first * 64, r->rng + off * 64, (size_t)take * 64);
        else if (!fast_random((uint8_t *)b->h + b->off_r + (size_t)first * 64, (size_t)take * 64)) b->poison.store(1, std::memory_order_release);
        if (key.mode != CK_SHARED) {
            char *dst = b->h + b->off_t + (size_t)first * TS;
            if (r->ts_in) memcpy(dst, r->ts_in + off * TS, (size_t)take * TS);
            else
                for (uint32_t i = 0; i < take; i++) memcpy(dst + (size_t)i * TS, shared, TS);
        }
        b->written.fetch_add(take, std::memory_order_release);
        off += take;
    }
    return BPGPU_OK;
}

// requests no chain can take: malformed lengths, parameter errors, missing generators -- the ordinary entry point on the odd-jobs
// context reports them proof by proof (ProofError::FormatError / InvalidBitsize / InvalidGeneratorsLength, mod.rs:358-366, 505-510)
static int comb_direct(bpgpu_pool *p, pool_dev *d, comb_req *r, const uint8_t *shared) {
    std::lock_guard<std::mutex> lk(d->misc_mu);
    const int rc = bpgpu_rangeproof_verify_batch_ts(d->misc, r->n, r->m, r->nbatch, r->proofs, r->proof_len, r->coms, r->ts_in ? r->ts_in : shared,
                                                    r->ts_in ? BPGPU_TRANSCRIPT_BYTES : 0, r->rng, r->verdict, r->msm, r->ts_out);
    if (rc) return pfail(p, rc, "%s", bpgpu_last_error(d->misc));
    return BPGPU_OK;
}

// validate, classify, place.  Synchronous requests also collect; tickets return once everything is placed.
static int comb_run_inner(bpgpu_pool *p, comb_req *r, const uint8_t *transcripts, size_t stride) {
    if (!p || p->devs.empty()) return BPGPU_ERR_INVALID_ARG;
    if (r->nbatch == 0) return BPGPU_OK;
    if (!r->proofs || !r->verdict || (r->m && !r->coms) || !transcripts) return pfail(p, BPGPU_ERR_INVALID_ARG, "null argument");
    if (stride != 0 && stride != BPGPU_TRANSCRIPT_BYTES) return pfail(p, BPGPU_ERR_INVALID_ARG, "transcript_stride neither 0 nor BPGPU_TRANSCRIPT_BYTES");
    if (r->nbatch > 0x7fffffffu / 64) return pfail(p, BPGPU_ERR_INVALID_ARG, "batch too large");
    for (size_t i = 0; i < (stride ? r->nbatch : 1); i++)
        if (!ts_ok(transcripts + i * BPGPU_TRANSCRIPT_BYTES)) return pfail(p, BPGPU_ERR_INVALID_ARG, "malformed transcript state %zu", i);
    const uint8_t *shared = stride ? nullptr : transcripts;
    r->ts_in = stride ? transcripts : nullptr;
    comb_key base;
    base.n = (uint32_t)r->n;
    base.m = (uint32_t)r->m;
    base.proof_len = (uint32_t)r->proof_len;
    if (shared && !r->ts_out) {
        base.mode = CK_SHARED;
        memcpy(base.shared, shared, BPGPU_TRANSCRIPT_BYTES);
        memset(base.shared + 203, 0, BPGPU_TRANSCRIPT_BYTES - 203);   // (bytes behind the STROBE bookkeeping carry nothing)
    } else base.mode = CK_UNIFORM;
    const size_t ndev = p->devs.size();
    pool_dev *d0 = p->devs[p->rr_dev.fetch_add(1, std::memory_order_relaxed) % ndev];
    // can chains take this shape?  (asked of the odd-jobs context; the answer is remembered per thread until generators change)
    static thread_local struct {
        const bpgpu_pool *p;
        size_t n, m, len;
        uint64_t epoch;
    } ok_shape = {nullptr, 0, 0, 0, 0};
    const uint64_t epoch = p->gens_epoch.load(std::memory_order_acquire);
    bool ok = ok_shape.p == p && ok_shape.n == r->n && ok_shape.m == r->m && ok_shape.len == r->proof_len && ok_shape.epoch == epoch;
    if (!ok && r->n <= 0xffff && r->m <= 0xffffff && r->proof_len <= 0xffffff && bpgpu_internal_rp_coalescible(d0->misc, r->n, r->m, r->proof_len)) {
        ok_shape = {p, r->n, r->m, r->proof_len, epoch};
        ok = true;
    }
    if (!ok) {
        return comb_direct(p, d0, r, shared);
    }
    {
        std::lock_guard<std::mutex> g(d0->cmu);
        d0->stat_requests++;
    }
    // Proofs are independent units: a large request takes a contiguous shard per device, a small one a single device
    // (round-robin over the requests).  The "gather" is the placement of every piece's verdicts at its offset of the caller's buffer.
    int rc = BPGPU_OK;
    if (ndev == 1 || r->nbatch < 512 * ndev) {
        rc = comb_place(p, d0, r, base, shared, 0, r->nbatch);
    } else {
        for (size_t di = 0; di < ndev && !rc; di++) {
            const size_t lo = r->nbatch * di / ndev, hi = r->nbatch * (di + 1) / ndev;
            if (hi > lo) rc = comb_place(p, p->devs[di], r, base, shared, lo, hi);
        }
    }
    if (r->async) return rc;   // (comb_run drops the placement guard)
    while (r->next_piece < r->pieces.size()) comb_collect_one(r);   // (also after a placement error: nothing stays referenced behind the caller's back)
    if (!rc && r->rc) rc = pfail(p, r->rc, "%s", r->err.c_str());
    return rc;
}
static int comb_run(bpgpu_pool *p, comb_req *r, const uint8_t *transcripts, size_t stride) {
    const int rc = comb_run_inner(p, r, transcripts, stride);
    if (r->async) {
        // the placement guard goes; the ticket is complete when its last piece has been delivered (bpgpu_pool_ticket_wait)
        std::lock_guard<std::mutex> g(r->emu);
        r->left.fetch_sub(1, std::memory_order_seq_cst);
        if (rc && !r->rc) {
            r->rc = rc;
            r->err = t_pool_err;
        }
    }
    return rc;
}

// a ticket's owner sleeps until its last piece has been delivered
static void ticket_block(comb_req *r) {
    for (;;) {
        uint32_t v = r->left.load(std::memory_order_acquire);
        if (v == 0) break;
        r->waiting.store(1, std::memory_order_seq_cst);
        v = r->left.load(std::memory_order_seq_cst);
        if (v == 0) break;
        futex_wait(&r->left, v);
    }
    std::lock_guard<std::mutex> g(r->emu);   // the delivering thread's last touch of the request happens under emu
}

extern "C" {

int bpgpu_pool_rangeproof_verify_ts(bpgpu_pool *p, size_t n, size_t m, size_t nbatch, const uint8_t *proofs, size_t proof_len, const uint8_t *commitments,
                                    const uint8_t *transcripts, size_t transcript_stride, const uint8_t *rng64, uint8_t *verdict, uint8_t *msm_out,
                                    uint8_t *transcripts_out) {
    if (!p) return BPGPU_ERR_INVALID_ARG;
    comb_req r;
    r.n = n, r.m = m, r.nbatch = nbatch, r.proof_len = proof_len;
    r.proofs = proofs, r.coms = commitments, r.rng = rng64;
    r.verdict = verdict, r.msm = msm_out, r.ts_out = transcripts_out;
    return comb_run(p, &r, transcripts, transcript_stride);
}

int bpgpu_pool_rangeproof_submit_ts(bpgpu_pool *p, size_t n, size_t m, size_t nbatch, const uint8_t *proofs, size_t proof_len, const uint8_t *commitments,
                                    const uint8_t *transcripts, size_t transcript_stride, const uint8_t *rng64, uint8_t *verdict, uint8_t *msm_out,
                                    uint8_t *transcripts_out, bpgpu_ticket **ticket) {
    if (!p || !ticket) return BPGPU_ERR_INVALID_ARG;
    *ticket = nullptr;
    comb_req *r = new comb_req();
    r->async = true;
    r->n = n, r->m = m, r->nbatch = nbatch, r->proof_len = proof_len;
    r->proofs = proofs, r->coms = commitments, r->rng = rng64;
    r->verdict = verdict, r->msm = msm_out, r->ts_out = transcripts_out;
    const int rc = comb_run(p, r, transcripts, transcript_stride);
    if (rc) {   // argument errors, or placement broke off: whatever is in flight is waited for, then the error is this call's
        const std::string keep = t_pool_err;
        ticket_block(r);
        delete r;
        t_pool_err = keep;
        return rc;
    }
    *ticket = (bpgpu_ticket *)r;
    return BPGPU_OK;
}

static int flush_dev(bpgpu_pool *p, pool_dev *d, bool one_chain);
// the chains that carry a device-pointer batch: issue them if they have not left yet, hand back their events
static int dev_ticket_events(bpgpu_pool *p, dev_ticket *t, std::vector<std::shared_ptr<ev_holder>> &evs) {
    std::lock_guard<std::mutex> lk(p->mu);
    int rc = BPGPU_OK;
    if (t->unissued) rc = flush_dev(p, t->d, false);
    evs = t->done;
    if (!rc && t->rc) rc = pfail(p, t->rc, "%s", t->err.c_str());
    return rc;
}

int bpgpu_pool_ticket_done(bpgpu_pool *p, bpgpu_ticket *ticket) {
    if (!p || !ticket) return BPGPU_ERR_INVALID_ARG;
    if (*(uint32_t *)ticket == TK_DEVICE) {
        dev_ticket *t = (dev_ticket *)ticket;
        std::lock_guard<std::mutex> lk(p->mu);
        if (t->unissued) return 0;
        if (hipSetDevice(t->d->device) != hipSuccess) return BPGPU_ERR_HIP;
        for (auto &e : t->done)
            if (hipEventQuery(e->ev) != hipSuccess) {
                (void)hipGetLastError();
                return 0;
            }
        return 1;
    }
    return ((comb_req *)ticket)->left.load(std::memory_order_acquire) == 0 ? 1 : 0;
}

int bpgpu_pool_ticket_wait(bpgpu_pool *p, bpgpu_ticket *ticket) {
    if (!p || !ticket) return BPGPU_ERR_INVALID_ARG;
    if (*(uint32_t *)ticket == TK_DEVICE) {
        dev_ticket *t = (dev_ticket *)ticket;
        std::vector<std::shared_ptr<ev_holder>> evs;
        int rc = dev_ticket_events(p, t, evs);
        if (hipSetDevice(t->d->device) != hipSuccess) rc = rc ? rc : BPGPU_ERR_HIP;
        for (auto &e : evs)
            if (hipEventSynchronize(e->ev) != hipSuccess && !rc) rc = pfail(p, BPGPU_ERR_HIP, "waiting for a chain of the batch failed");
        delete t;
        return rc;
    }
    comb_req *r = (comb_req *)ticket;
    ticket_block(r);
    const int rc = r->rc;
    if (rc) t_pool_err = r->err;
    delete r;
    return rc;
}

int bpgpu_pool_ticket_stream_wait(bpgpu_pool *p, bpgpu_ticket *ticket, void *stream) {
    if (!p || !ticket) return BPGPU_ERR_INVALID_ARG;
    if (*(uint32_t *)ticket != TK_DEVICE) return pfail(p, BPGPU_ERR_INVALID_ARG, "only tickets of device-pointer batches have a device-side completion");
    dev_ticket *t = (dev_ticket *)ticket;
    std::vector<std::shared_ptr<ev_holder>> evs;
    int rc = dev_ticket_events(p, t, evs);
    if (hipSetDevice(t->d->device) != hipSuccess) return BPGPU_ERR_HIP;
    for (auto &e : evs)
        if (hipStreamWaitEvent((hipStream_t)stream, e->ev, 0) != hipSuccess && !rc) rc = pfail(p, BPGPU_ERR_HIP, "hipStreamWaitEvent failed");
    return rc;
}

}  // extern "C"

extern "C" {

// ---- device pointers, asynchronous ------------------------------------------------------------------------------
// a chain (or a stand-alone call) on lane `c` carries `count` proofs of item `it`: before it is issued its stream waits for the item's
// producer; afterwards the item's ticket gets the event recorded behind the chain
static void chain_waits_for(bpgpu_ctx *c, const dev_item &it) {
    if (it.ready) (void)hipStreamWaitEvent((hipStream_t)bpgpu_internal_stream(c), it.ready->ev, 0);
}
static void chain_carried(const dev_item &it, size_t count, const std::shared_ptr<ev_holder> &done, int rc, const char *err) {
    if (!it.ticket) return;
    it.ticket->unissued -= count;
    if (done && (it.ticket->done.empty() || it.ticket->done.back() != done)) it.ticket->done.push_back(done);
    if (rc && !it.ticket->rc) {
        it.ticket->rc = rc;
        it.ticket->err = err;
    }
}
static std::shared_ptr<ev_holder> record_done(bpgpu_ctx *c, bool wanted) {
    if (!wanted) return nullptr;
    auto h = std::make_shared<ev_holder>();
    if (hipEventCreateWithFlags(&h->ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(h->ev, (hipStream_t)bpgpu_internal_stream(c)) != hipSuccess) return nullptr;
    return h;
}

// ---- how a flush of T pending proofs is cut into launch chains (plain host logic: tests/test_abi_and_host.py) -------------------------
struct flush_plan {
    size_t chains, per;      // at most `chains` chains of up to `per` proofs (a chain takes consecutive items of one shape)
    uint32_t splits_hint;    // table-walk workgroups per proof block handed to the chains (bpgpu.hip pick_splits)
};
static flush_plan plan_flush(size_t T, size_t coalesce_proofs, size_t pair_limit_proofs, size_t max_chain_proofs, size_t lanes, bool all_rlc, bool one_chain) {
    // number of chains: about T / coalesce_proofs, at most one per lane
    size_t G = (T + coalesce_proofs / 2) / coalesce_proofs;
    if (G < 1) G = 1;
    // a burst that fits two chains takes two: with the one-lane Horner chains aside, 20 x 1024 from idle measured 5.6 / 5.6 / 5.8 M/s as four
    // chains of 5120 and 5.9 ... 6.2 / 5.7 ... 5.8 / 6.0 as two of 10240 on three boxes; 40 x 1024 prefers eight of 5120 (6.05 against 5.85 as
    // four of 10240), 8 x 1024 two of 4096 (4.7 against 4.3 ... 4.6 as one) -- profiles/r03/coalesce_sweep_after_horner_aside.txt
    // Batch-combined bursts keep the chains of coalesce_proofs: 20 x 1024 as four combinations 8.1 ... 8.3 M/s, as two 7.7 ... 7.9
    // (profiles/r04/ab_rlc_burst_chains.txt; measured the other way round -- 3.9 against 5.5 -- while short weights still crowded one
    // bucket of every combination)
    if (G > 2 && T <= pair_limit_proofs && !all_rlc) G = 2;
    if (one_chain) G = 1;   // a chain's worth has accumulated while the caller is still submitting: it goes out now, as it is
    if (G > lanes) G = lanes;
    size_t per = (T + G - 1) / G;
    if (per > max_chain_proofs) per = max_chain_proofs;
    if (per < 1) per = 1;
    // table-walk workgroups per proof block: a few chains alone on the device want many small workgroups (short tail: measured on
    // 20 x 1024 from idle, 5 chains: 16 splits 5.06 M/s, 32: 5.24, 64: 5.21), a full pipeline wants few (less reduction work) -- aim
    // at ~16 k wavefronts of table walk per flush
    const size_t n_chains = (T + per - 1) / per;
    uint32_t hint = (uint32_t)(16384 / ((n_chains ? n_chains : 1) * ((per + 63) / 64)));
    hint = (hint + 7) & ~7u;
    if (hint < 16) hint = 16;
    if (hint > 64) hint = 64;
    flush_plan fp;
    fp.chains = n_chains;
    fp.per = per;
    fp.splits_hint = hint;
    return fp;
}
extern "C" void bpgpu_internal_plan_flush(uint64_t T, uint64_t coalesce_proofs, uint64_t pair_limit_proofs, uint64_t max_chain_proofs, uint64_t lanes, int all_rlc,
                                          int one_chain, uint64_t *chains, uint64_t *per, uint32_t *splits_hint) {
    const flush_plan fp = plan_flush((size_t)T, (size_t)coalesce_proofs, (size_t)pair_limit_proofs, (size_t)max_chain_proofs, (size_t)lanes, all_rlc != 0, one_chain != 0);
    *chains = fp.chains, *per = fp.per, *splits_hint = fp.splits_hint;
}

static int flush_dev(bpgpu_pool *p, pool_dev *d, bool one_chain) {
    if (d->pending.empty()) return BPGPU_OK;
    (void)hipSetDevice(d->device);
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
    bool all_rlc = !items.empty();
    for (const dev_item &it : items) all_rlc = all_rlc && it.rlc;
    const flush_plan fp = plan_flush(T, p->coalesce_proofs, p->pair_limit_proofs, p->max_chain_proofs, d->lanes.size(), all_rlc, one_chain);
    const size_t per = fp.per;
    const uint32_t hint = fp.splits_hint;
    int rc_all = BPGPU_OK;
    size_t n_undecided = 0;
    std::string first_err;
    std::vector<rp_seg> segs;
    std::vector<std::pair<size_t, size_t>> carried;   // (item, proofs of it) in the chain being packed
    size_t i = 0, off = 0;   // item i, `off` proofs of it already placed
    while (i < items.size()) {
        const dev_item &head = items[i];
        bpgpu_ctx *c = d->lanes[d->next_lane++ % d->lanes.size()];
        if (d->used_lanes < d->lanes.size() && d->next_lane > d->used_lanes) d->used_lanes = d->next_lane < d->lanes.size() ? d->next_lane : d->lanes.size();
        if (!bpgpu_internal_rp_coalescible(c, head.n, head.m, head.proof_len)) {
            // malformed length / parameter error / missing generators: the ordinary entry point reports it proof by proof
            chain_waits_for(c, head);
            const int rc = head.rlc ? bpgpu_rangeproof_verify_rlc_dev(c, head.n, head.m, head.nbatch, head.proofs, head.proof_len, head.coms,
                                                                      (const uint8_t *)head.label.data(), head.label.size(), head.rng, nullptr, head.verdict, head.msm, nullptr)
                                    : bpgpu_rangeproof_verify_batch_dev(c, head.n, head.m, head.nbatch, head.proofs, head.proof_len, head.coms,
                                                                        (const uint8_t *)head.label.data(), head.label.size(), head.rng, head.verdict, head.msm, nullptr);
            chain_carried(head, head.nbatch, record_done(c, head.ticket != nullptr), rc, rc ? bpgpu_last_error(c) : "");
            if (rc) {
                (void)hipMemsetAsync(head.verdict, BPGPU_VERDICT_UNDECIDED, head.nbatch, (hipStream_t)bpgpu_internal_stream(c));
                n_undecided += head.nbatch;
                if (!rc_all) {
                    rc_all = rc;
                    first_err = bpgpu_last_error(c);
                }
            }
            i++;
            off = 0;
            continue;
        }
        segs.clear();
        carried.clear();
        uint32_t filled = 0;
        bool any_msm = false, any_ticket = false;
        while (i < items.size() && filled < per && items[i].same_shape(head)) {
            const dev_item &it = items[i];
            size_t take = it.nbatch - off;
            // a batch-combined item stays whole (its 33-byte result is the result of ONE chain): it opens the next chain rather than being cut,
            // and may stretch a chain up to max_chain_proofs; only an item wider than that is cut (its batch_out is then the last chain's)
            if (head.rlc && off == 0 && filled > 0 && take > per - filled) break;
            const size_t room = (head.rlc && off == 0 && take <= p->max_chain_proofs) ? take : per - filled;
            if (take > room) take = room;
            chain_waits_for(c, it);
            carried.push_back({i, take});
            any_ticket = any_ticket || it.ticket;
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
        const int rc = bpgpu_internal_rp_verify_segs(c, head.n, head.m, head.proof_len, (const uint8_t *)head.label.data(), head.label.size(), segs.data(),
                                                     (uint32_t)segs.size(), any_msm, head.rlc ? 0u : hint, (was_idle && T <= p->latency_proofs) ? 0 : 1, head.rlc);   // (the hint is for per-proof table walks: a combined chain walks ONE 1690-pair MSM, which wants all the workgroups it can get)
        {
            const std::shared_ptr<ev_holder> done = record_done(c, any_ticket);   // (also behind the memsets of a failed chain below: same stream)
            for (const auto &cr : carried) chain_carried(items[cr.first], cr.second, done, rc, rc ? bpgpu_last_error(c) : "");
        }
        if (rc) {
            // The chain did not go out: its items' verdict bytes must not read 0 = "verified" (a caller with zero-initialised
            // buffers would take that for acceptance).  Every affected range gets BPGPU_VERDICT_UNDECIDED; the flush goes on with
            // the next chain and reports the first error together with the number of proofs left undecided.
            for (const rp_seg &sg : segs) {
                (void)hipMemsetAsync(sg.verdict, BPGPU_VERDICT_UNDECIDED, sg.count, (hipStream_t)bpgpu_internal_stream(c));
                n_undecided += sg.count;
            }
            if (!rc_all) {
                rc_all = rc;
                first_err = bpgpu_last_error(c);
            }
        }
        p->stat_chains++;
        p->stat_chain_proofs += filled;
        p->stat_last_splits = hint;
    }
    if (rc_all) return pfail(p, rc_all, "%s (%zu proofs of this flush are marked BPGPU_VERDICT_UNDECIDED; every other verdict of it is valid)", first_err.c_str(), n_undecided);
    return rc_all;
}

}  // extern "C"

static int submit_dev_common(bpgpu_pool *p, int dev_index, size_t n, size_t m, size_t nbatch, const void *d_proofs, size_t proof_len, const void *d_commitments,
                             const uint8_t *label, size_t label_len, const void *d_rng64, void *d_verdict, void *d_msm_out, void *producer_stream,
                             int have_producer, bpgpu_ticket **ticket, bool rlc) {
    if (ticket) *ticket = nullptr;
    if (!p || dev_index < 0 || dev_index >= (int)p->devs.size() || (label_len && !label)) return BPGPU_ERR_INVALID_ARG;
    if (nbatch == 0 && !ticket) return BPGPU_OK;
    if (nbatch && (!d_proofs || !d_verdict || (m && !d_commitments))) return BPGPU_ERR_INVALID_ARG;
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
    it.rlc = rlc;
    if (have_producer && nbatch) {   // the inputs are complete when the work queued on the producer's stream so far is: the chain will wait for exactly that
        (void)hipSetDevice(d->device);
        it.ready = std::make_shared<ev_holder>();
        if (hipEventCreateWithFlags(&it.ready->ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(it.ready->ev, (hipStream_t)producer_stream) != hipSuccess)
            return pfail(p, BPGPU_ERR_HIP, "recording the producer's event failed");
    }
    if (ticket) {
        dev_ticket *t = new dev_ticket();
        t->d = d;
        t->unissued = nbatch;
        it.ticket = t;
        *ticket = (bpgpu_ticket *)t;
        if (nbatch == 0) return BPGPU_OK;
    }
    d->pending.push_back(std::move(it));
    d->pending_proofs += nbatch;
    const size_t limit = p->auto_flush_items ? p->auto_flush_items : d->lanes.size();
    if (d->pending.size() >= limit) return flush_dev(p, d, false);
    if (p->auto_flush_proofs && d->pending_proofs >= p->auto_flush_proofs) return flush_dev(p, d, true);
    return BPGPU_OK;
}

extern "C" {

int bpgpu_pool_rangeproof_submit_dev_ex(bpgpu_pool *p, int dev_index, size_t n, size_t m, size_t nbatch, const void *d_proofs, size_t proof_len,
                                        const void *d_commitments, const uint8_t *label, size_t label_len, const void *d_rng64, void *d_verdict,
                                        void *d_msm_out, void *producer_stream, int have_producer, bpgpu_ticket **ticket) {
    return submit_dev_common(p, dev_index, n, m, nbatch, d_proofs, proof_len, d_commitments, label, label_len, d_rng64, d_verdict, d_msm_out, producer_stream,
                             have_producer, ticket, false);
}

int bpgpu_pool_rangeproof_submit_rlc_dev(bpgpu_pool *p, int dev_index, size_t n, size_t m, size_t nbatch, const void *d_proofs, size_t proof_len,
                                         const void *d_commitments, const uint8_t *label, size_t label_len, const void *d_rng64, void *d_verdict,
                                         void *d_batch_out, void *producer_stream, int have_producer, bpgpu_ticket **ticket) {
    return submit_dev_common(p, dev_index, n, m, nbatch, d_proofs, proof_len, d_commitments, label, label_len, d_rng64, d_verdict, d_batch_out, producer_stream,
                             have_producer, ticket, true);
}

int bpgpu_pool_rangeproof_submit_dev(bpgpu_pool *p, int dev_index, size_t n, size_t m, size_t nbatch, const void *d_proofs, size_t proof_len,
                                     const void *d_commitments, const uint8_t *label, size_t label_len, const void *d_rng64, void *d_verdict,
                                     void *d_msm_out) {
    return bpgpu_pool_rangeproof_submit_dev_ex(p, dev_index, n, m, nbatch, d_proofs, proof_len, d_commitments, label, label_len, d_rng64, d_verdict, d_msm_out,
                                               nullptr, 0, nullptr);
}

// The "final identity-check gather" for callers that keep verdicts on the devices: every shard's verdict bytes to ONE device buffer,
// over the peer links (xGMI on an MI355X node), ordered behind the shard's own work.  Proofs are independent units, nothing else ever
// crosses devices.  `part[d]` (device memory on pool device d, `bytes[d]` bytes, may be 0) lands at d_dst + sum of the
// sizes before it; d_dst lives on pool device `root`.  Asynchronous on `stream` (a hipStream_t of the root device; NULL = its default
// stream); the copies wait for everything issued so far on their source device's lanes.
int bpgpu_pool_gather_dev(bpgpu_pool *p, int root, const void *const *part, const size_t *bytes, void *d_dst, void *stream) {
    if (!p || root < 0 || root >= (int)p->devs.size() || !part || !bytes || !d_dst) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(p->mu);
    pool_dev *rd = p->devs[root];
    if (hipSetDevice(rd->device) != hipSuccess) return pfail(p, BPGPU_ERR_HIP, "hipSetDevice failed");
    size_t off = 0;
    for (size_t di = 0; di < p->devs.size(); di++) {
        pool_dev *d = p->devs[di];
        if (bytes[di] == 0) continue;
        if (!part[di]) return pfail(p, BPGPU_ERR_INVALID_ARG, "part %zu is null", di);
        // the gather stream waits for the lanes of the source device (an event per lane that has work in flight)
        if (hipSetDevice(d->device) != hipSuccess) return pfail(p, BPGPU_ERR_HIP, "hipSetDevice failed");
        for (bpgpu_ctx *c : d->lanes) {
            hipEvent_t ev;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return pfail(p, BPGPU_ERR_HIP, "hipEventCreate failed");
            hipError_t e = hipEventRecord(ev, (hipStream_t)bpgpu_internal_stream(c));
            if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)stream, ev, 0);
            hipEventDestroy(ev);   // (released by the runtime once the recorded work has completed)
            if (e != hipSuccess) return pfail(p, BPGPU_ERR_HIP, "ordering the gather behind device %d failed: %s", d->device, hipGetErrorString(e));
        }
        if (hipSetDevice(rd->device) != hipSuccess) return pfail(p, BPGPU_ERR_HIP, "hipSetDevice failed");
        const hipError_t e = d->device == rd->device ? hipMemcpyAsync((char *)d_dst + off, part[di], bytes[di], hipMemcpyDeviceToDevice, (hipStream_t)stream)
                                                     : hipMemcpyPeerAsync((char *)d_dst + off, rd->device, part[di], d->device, bytes[di], (hipStream_t)stream);
        if (e != hipSuccess) return pfail(p, BPGPU_ERR_HIP, "gather copy from device %d failed: %s", d->device, hipGetErrorString(e));
        off += bytes[di];
    }
    return BPGPU_OK;
}

int bpgpu_pool_flush(bpgpu_pool *p) {
    if (!p) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(p->mu);
    int rc_all = BPGPU_OK;
    for (pool_dev *d : p->devs) {
        const int rc = flush_dev(p, d, false);
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
