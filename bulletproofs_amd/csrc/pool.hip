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
int bpgpu_internal_rp_verify_segs(bpgpu_ctx *c, size_t n, size_t m, size_t proof_len, const uint8_t *const *labels, size_t label_len, const rp_seg *segs,
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
#ifndef BPGPU_POOL_HOST_TEST
__global__ void k_pool_spin(uint64_t ticks, uint32_t *sink) {
    const uint64_t t0 = wall_clock64();
    uint32_t x = 0;
    while (wall_clock64() - t0 < ticks) x++;
    if (ticks == ~0ull) *sink = x;
}
// Advisory: a device shared with another tenant queues the spin kernels behind foreign work and can read low although
// sixteen queues exist -- the caller (bpgpu_pool_create) only refuses when a second measurement agrees.  The calling thread's current
// device is put back; the tick count comes from the device's wall-clock rate.
static int probe_hw_queues(int device) {
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) return -1;
    struct restore {
        int prev;
        ~restore() {
            if (prev >= 0) (void)hipSetDevice(prev);
        }
    } put_back{prev};
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || khz <= 0) khz = 100000;   // (gfx950: 100 MHz)
    const int NS = 16;
    hipStream_t st[NS];
    for (int i = 0; i < NS; i++)
        if (hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) != hipSuccess) return -1;
    const uint64_t ticks = (uint64_t)khz;   // 1 ms
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {   // the first round also pays for loading the code object
        for (int i = 0; i < NS; i++) (void)hipStreamSynchronize(st[i]);
        const uint64_t t0 = now_ns();
        for (int i = 0; i < NS; i++) hipLaunchKernelGGL(k_pool_spin, dim3(1), dim3(64), 0, st[i], ticks, (uint32_t *)nullptr);
        for (int i = 0; i < NS; i++) (void)hipStreamSynchronize(st[i]);
        const double us = (double)(now_ns() - t0) / 1000.0;
        if (us < best) best = us;
    }
    for (int i = 0; i < NS; i++) (void)hipStreamDestroy(st[i]);
    if (hipGetLastError() != hipSuccess) return -1;
    const double rounds = best / 1000.0;   // ~1 with >= 16 queues, ~2 with 8, ~4 with 4
    int q = (int)(16.0 / (rounds < 1.0 ? 1.0 : rounds) + 0.5);
    return q < 1 ? 1 : q;
}
#else
static int probe_hw_queues(int) { return 16; }   // (host test build: tests/cpu_pool, no device code)
#endif

namespace {

// Host-test builds only (tests/cpu_pool): BP_TEST_DELAY="point:microseconds" holds a thread at one of the queue's publication points, which
// widens a window of a few instructions to something a scheduler test can hit -- how the open-order bug of round 5 reproduces on demand.
#ifdef BPGPU_POOL_HOST_TEST
static void test_delay(int point) {
    static const int want = [] { const char *e = getenv("BP_TEST_DELAY"); return e ? atoi(e) : -1; }();
    static const int us = [] { const char *e = getenv("BP_TEST_DELAY"); const char *c = e ? strchr(e, ':') : nullptr; return c ? atoi(c + 1) : 100; }();
    if (point == want) usleep((useconds_t)us);
}
#define TEST_DELAY(point) test_delay(point)
#else
#define TEST_DELAY(point) ((void)0)
#endif

// std::atomic<uint32_t> as a futex word (C++17: no atomic::wait yet)
inline void futex_wait(std::atomic<uint32_t> *a, uint32_t expected) { syscall(SYS_futex, (uint32_t *)a, FUTEX_WAIT_PRIVATE, expected, nullptr, nullptr, 0); }
inline void futex_wake_all(std::atomic<uint32_t> *a) { syscall(SYS_futex, (uint32_t *)a, FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0); }
inline void futex_wait_ns(std::atomic<uint32_t> *a, uint32_t expected, uint64_t ns) {   // relative timeout
    timespec ts;
    ts.tv_sec = (time_t)(ns / 1000000000ull);
    ts.tv_nsec = (long)(ns % 1000000000ull);
    syscall(SYS_futex, (uint32_t *)a, FUTEX_WAIT_PRIVATE, expected, &ts, nullptr, 0);
}

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
    // what lets two items share a launch chain: shape and mode -- and the LENGTH of the label, not its bytes (every transcript position
    // depends on lengths only; items under different labels start from their own states, rp_seg::init_w)
    bool same_shape(const dev_item &o) const { return n == o.n && m == o.m && proof_len == o.proof_len && rlc == o.rlc && label.size() == o.label.size(); }
};

// ---- combining queue ---------------------------------------------------------------------------------------------------
// Kinds of work the queue combines (one staging-buffer class never mixes kinds):
//   CQ_RP          range proofs, proof bytes -> verdict (bpgpu_pool_rangeproof_verify[_ts], _submit_ts)
//   CQ_MSM_SHARED  optional_multiscalar_mul in its mega-check shape: 2nm+2 generator scalars + n_unique (scalar, point) pairs per MSM
//                  (bpgpu_pool_msm_batch_shared; src/range_proof/mod.rs:421-445, src/r1cs/verifier.rs:459-491)
//   CQ_MSM         the same call with arbitrary points, all MSMs of a buffer `n_terms` terms long (bpgpu_pool_msm_batch; ipp.rs:308-319)
//   CQ_IPP         InnerProductProof::verify (bpgpu_pool_ipp_verify; ipp.rs:260-326)
enum { CQ_RP = 0, CQ_MSM_SHARED = 1, CQ_MSM = 2, CQ_IPP = 3 };
// What a range-proof chain can share beside the shape: how its transcripts start --
//   CK_SHARED  every proof from ONE 208-byte state (Transcript::new(label) of a common label, or a transcript the callers
//              share), no states handed back: nothing per proof to upload;
//   CK_UNIFORM one state per proof in / out, all at the same STROBE position (pos, pos_begin, cur_flags): the per-shape script
//              (rp_script.h) still applies, only the sponge words differ;
//   CK_MIXED   one state per proof at whatever position: the byte-wise replay.  Catch-all when too many position classes are open.
enum { CK_SHARED = 0, CK_UNIFORM = 1, CK_MIXED = 2 };
#define CQ_MAX_IN 7
#define CQ_MAX_OUT 3
struct comb_key {   // compared bytewise: no padding, unused fields zero
    uint32_t kind = 0, mode = 0;
    uint32_t a = 0, b = 0, c = 0, d = 0;   // CQ_RP: n, m, proof_len | CQ_MSM_SHARED: n, m, n_unique | CQ_MSM: n_terms | CQ_IPP: n, proof_len
    uint32_t pos = 0, pos_begin = 0, flags = 0, pad = 0;   // CK_UNIFORM
    uint8_t shared[BPGPU_TRANSCRIPT_BYTES] = {0};          // CK_SHARED / CQ_IPP: the common start state
    bool operator==(const comb_key &o) const { return memcmp(this, &o, sizeof *this) == 0; }
};
static_assert(sizeof(comb_key) == 40 + BPGPU_TRANSCRIPT_BYTES, "comb_key must have no padding");

// bytes per item of every input / output region of a class.  The LAST input region is the one a partly filled buffer copies by
// fill; regions of size 0 do not exist for the class.
struct comb_regions {
    uint32_t n_in = 0, n_out = 0, fail_out = 0;   // fail_out: the output region that receives BPGPU_VERDICT_UNDECIDED when the chain failed
    size_t in_sz[CQ_MAX_IN] = {0}, out_sz[CQ_MAX_OUT] = {0};
};
enum { RP_IN_COMS = 0, RP_IN_RNG = 1, RP_IN_TS = 2, RP_IN_PROOFS = 3, RP_OUT_VERDICT = 0, RP_OUT_TS = 1, RP_OUT_MSM = 2 };
static comb_regions regions_of(const comb_key &k) {
    comb_regions g;
    const size_t TS = BPGPU_TRANSCRIPT_BYTES;
    if (k.kind == CQ_RP) {
        const bool per = k.mode != CK_SHARED;
        g.n_in = 4, g.n_out = 3, g.fail_out = RP_OUT_VERDICT;
        g.in_sz[RP_IN_COMS] = (size_t)k.b * 32, g.in_sz[RP_IN_RNG] = 64, g.in_sz[RP_IN_TS] = per ? TS : 0, g.in_sz[RP_IN_PROOFS] = k.c;
        g.out_sz[RP_OUT_VERDICT] = 1, g.out_sz[RP_OUT_TS] = per ? TS : 0, g.out_sz[RP_OUT_MSM] = 32;
    } else if (k.kind == CQ_MSM_SHARED) {
        g.n_in = 3, g.n_out = 2, g.fail_out = 1;
        g.in_sz[0] = (size_t)k.c * 32, g.in_sz[1] = (size_t)k.c * 32, g.in_sz[2] = ((size_t)2 * k.a * k.b + 2) * 32;   // unique scalars, unique points, generator scalars (the largest: last)
        g.out_sz[0] = 32, g.out_sz[1] = 1;
    } else if (k.kind == CQ_MSM) {
        g.n_in = 2, g.n_out = 2, g.fail_out = 1;
        g.in_sz[0] = (size_t)k.a * 32, g.in_sz[1] = (size_t)k.a * 32;   // scalars, points
        g.out_sz[0] = 32, g.out_sz[1] = 1;
    } else {   // CQ_IPP
        g.n_in = 7, g.n_out = 2, g.fail_out = 0;
        g.in_sz[0] = 32, g.in_sz[1] = 32, g.in_sz[2] = k.b;                                // P, Q, proof
        g.in_sz[3] = g.in_sz[4] = g.in_sz[5] = g.in_sz[6] = (size_t)k.a * 32;             // G_factors, H_factors, G, H
        g.out_sz[0] = 1, g.out_sz[1] = 32;
    }
    return g;
}

struct comb_req;
enum { CB_FREE = 0, CB_OPEN, CB_SEALED, CB_ISSUING, CB_ISSUED, CB_DONE };
// The reservation word of a staging buffer: [epoch : 32 | sealed : 1 | reserved : 31].  A caller takes slots with ONE compare-and-swap
// (no lock): it succeeds only on the incarnation (`epoch`) the caller looked at, only while that is unsealed, and never past `cap`.
// The service thread seals with a fetch_or; the count it reads back is the chain's final width.  A FREE buffer keeps the sealed bit.
static constexpr uint64_t CBS_SEALED = 1ull << 31;
static inline uint32_t cbs_epoch(uint64_t s) { return (uint32_t)(s >> 32); }
static inline uint32_t cbs_reserved(uint64_t s) { return (uint32_t)(s & 0x7fffffffu); }
static inline bool cbs_sealed(uint64_t s) { return (s & CBS_SEALED) != 0; }
static inline uint64_t cbs_pack(uint32_t epoch, bool sealed, uint32_t reserved) { return ((uint64_t)epoch << 32) | (sealed ? CBS_SEALED : 0) | reserved; }

// one timeline record per launch chain of the combining queue (option "combine_trace"; bpgpu_pool_trace_dump)
struct chain_ev {
    uint32_t buf = 0, epoch = 0, K = 0, kind = 0, n_sync = 0, n_async = 0, inflight = 0, cap = 0;
    uint64_t t_open = 0, t_seal = 0, t_issue0 = 0, t_issue1 = 0, t_done = 0, t_deliv0 = 0, t_deliv1 = 0, t_free = 0;
};
struct req_ev {   // one sampled request
    uint32_t buf = 0, epoch = 0, async = 0, nbatch = 0;
    uint64_t t_submit = 0, t_reserved = 0, t_written = 0, t_delivered = 0, t_woken = 0;
};

// One staging buffer + the lane that runs its chain.  Input regions and output regions sit at the same offsets of a pinned host
// block and of a device block.
struct comb_buf {
    pool_dev *dev = nullptr;
    bpgpu_ctx *ctx = nullptr;
    uint32_t index = 0;
    uint32_t reserved_n = 0, reserved_m = 0, reserved_len = 0, reserved_cap = 0;   // (CQ_RP) shape / width the lane's buffers were last sized for
    hipEvent_t done_ev = nullptr;
    char *h = nullptr, *d = nullptr, *hd = nullptr;   // pinned host block, device block, the host block as the device sees it (may be null)
    size_t mem_cap = 0;
    // ---- written by the opener (under pool_dev::cmu, buffer FREE), published by the release store of `state`; stable until FREE again
    comb_key key;
    comb_regions reg;
    std::atomic<uint64_t> class_id{0};
    std::atomic<uint32_t> cap{0};
    uint32_t cap_max = 0;
    size_t in_off[CQ_MAX_IN] = {0}, out_off[CQ_MAX_OUT] = {0}, in_end = 0, total = 0;
    uint64_t t_open = 0;
    struct piece_desc {   // written by the caller that reserved slots [first, first + count) at index `first`, before it counts itself into `written`
        comb_req *req;
        uint32_t count, async;
        size_t off;
    };
    std::vector<piece_desc> desc;
    // ---- shared between callers, the service thread and the deliverers
    std::atomic<uint64_t> state{CBS_SEALED};
    std::atomic<int> st{CB_FREE};
    std::atomic<uint32_t> written{0};       // items whose inputs are in place
    std::atomic<uint32_t> delivered{0};     // items whose results have been taken; the deliverer that completes K returns the buffer
    std::atomic<uint32_t> phase{0};         // futex word: epoch of the latest incarnation whose results are in `h`
    std::atomic<uint32_t> n_sync{0}, n_async{0};   // pieces of blocking callers / of tickets
    std::atomic<uint32_t> want_opt{0};      // some request wants the optional output region (encodings)
    std::atomic<int> poison{0};
    // ---- service thread only (and, after `phase`, whoever delivers)
    uint32_t K = 0;                         // final width
    uint32_t seen_reserved = 0, seen_epoch = 0;
    uint32_t sealed_epoch = 0;              // the incarnation the service thread sealed last: a word that still carries it is not the open buffer's
    uint64_t t_change = 0;                  // when `reserved` was last seen to move (quiet detection)
    int rc = 0;
    std::string err;
    chain_ev ev;
};

// one call (bpgpu_pool_rangeproof_verify_ts, bpgpu_pool_msm_batch_shared, ...) or one ticket (bpgpu_pool_rangeproof_submit_ts)
struct comb_req {
    uint32_t kind = TK_COMBINE;   // (first member of both ticket types)
    size_t nbatch = 0;
    uint8_t *out[CQ_MAX_OUT] = {nullptr, nullptr, nullptr};   // per item, strides = the class's output region sizes
    bool async = false;
    struct piece {
        comb_buf *b;
        uint32_t epoch, first, count;
        size_t off;
    };
    std::vector<piece> pieces;              // blocking calls: collected by the caller itself
    size_t next_piece = 0;
    // tickets: [waiting : 1 | pieces not delivered yet, +1 while the request is being placed : 31]; futex word.  The deliverer's last
    // access to the request is its fetch_sub.
    std::atomic<uint32_t> left{1};
    std::mutex emu;                         // guards rc / err (several deliverers may report errors)
    int rc = 0;
    std::string err;
    req_ev ev;
    bool traced = false;
};
static constexpr uint32_t TKT_WAITING = 1u << 31;

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
    // one service thread (seals buffers by policy, issues their chains, notices completions), one delivery thread (hands tickets
    // their results), one context for the requests no chain can take (malformed lengths, parameter errors: reported per proof by
    // the ordinary entry point)
    std::vector<comb_buf *> cbufs;
    bpgpu_ctx *misc = nullptr;
    std::mutex misc_mu;
    std::mutex cmu;                         // opening / freeing buffers, the class table; NOT taken to reserve slots in an open buffer
    std::condition_variable free_cv;
    // the service thread's doorbell: callers that open a buffer or take its last slot bump `kick` (and wake the thread if it sleeps);
    // the thread reads `kick` before it looks at the buffers and sleeps only while the word still holds that value
    std::atomic<uint32_t> kick{0}, svc_sleeping{0};
    std::atomic<uint32_t> free_waiters{0};  // callers asleep on free_cv (counted in under cmu)
    std::thread svc, dlv;
    bool svc_running = false;
    std::atomic<bool> cstop{false};   // (atomic: the service thread reads it on every poll and must never wait for `cmu`, which a caller may hold across a buffer's first allocation)
    struct cls_ent {
        comb_key key;
        uint64_t id, last_use;
    };
    std::vector<cls_ent> classes;           // interned class keys (under cmu); ids are never reused
    uint64_t next_class_id = 1, class_tick = 0;
    std::mutex dq_mu;                       // delivery queue: finished buffers that carry pieces of tickets
    std::condition_variable dq_cv;
    std::deque<comb_buf *> dq;
    bool dstop = false;
    std::atomic<uint64_t> stat_chains{0}, stat_proofs{0}, stat_requests{0};
    std::atomic<uint64_t> stat_issue_ns{0}, stat_complete_ns{0}, stat_polls{0}, stat_deliver_ns{0};
    std::atomic<uint32_t> recent_K{0};      // width of the chain issued last (sizes the next buffer)
    uint64_t ema_small_ns = 0;                    // (service thread) running mean of issue end -> completion seen over range-proof chains of <= 4 proofs: when such a chain
                                                  // is due, the thread polls instead of sleeping out its period (svc_main)
    uint32_t last_done_K[4] = {0, 0, 0, 0};      // (service thread) width of the chain of each kind that completed last: the group about to come back
    uint64_t t_last_done[4] = {0, 0, 0, 0};
    std::mutex trace_mu;
    std::vector<chain_ev> trace_chains;     // rings (option "combine_trace")
    std::vector<req_ev> trace_reqs;
    size_t trace_chain_n = 0, trace_req_n = 0;
};

}  // namespace

static std::atomic<uint64_t> g_pool_uid{1};   // (a pool created at the address of a destroyed one is still another pool: per-thread caches key on this)

struct bpgpu_pool {
    std::vector<pool_dev *> devs;
    const uint64_t uid = g_pool_uid.fetch_add(1);
    std::mutex mu;        // serialises the pool's own state (pending lists, options); lane contexts have their own locks
    size_t coalesce_proofs = 5120;   // target width of a coalesced launch chain
    size_t pair_limit_proofs = 24576;   // a flush of up to this many proofs is issued as at most two chains (flush_dev)
    size_t max_chain_proofs = 16384; // never wider than this (arena of a lane: ~55 KB per proof)
    size_t slice_proofs = 0;         // host-pointer calls: proofs per slice (0 = automatic)
    size_t latency_proofs = 6144;    // a host call / a flush on an idle device of up to this many proofs is "alone": its chains take the latency forms
    size_t auto_flush_items = 0;     // flush by itself once this many items wait on a device (0 = lanes)
    size_t auto_flush_proofs = 0;    // ... or once this many proofs wait: they go out as ONE chain while the caller keeps submitting (0 = off)
    // A flush is cut into chains by proof count, except that a LONE chain carrying two chains' worth of table-walk work is cut in two
    // (split_lone_heavy).  Two alternatives were built, measured and removed in round 6: chains in proportion to work (BASELINE configs 3 / 4:
    // -4 % / -23 %, profiles/r05/plan_by_work_ab.txt -- a burst wants TWO overlapping chains whatever the proofs weigh) and staggered starts
    // (chain i + 1 behind chain i's early phase: slower in every row of profiles/r05/stagger_chains_ab.txt).
    // Batch-combined items of unrelated submitters normally share ONE identity check per chain: a single bad proof then leaves every
    // proof of every batch of that chain undecided, and each submitter has to find out alone (one hostile submitter sends
    // everybody to the slow path).  1 = a chain never carries more than one batch-combined item: a failure stays with its own batch.
    int rlc_isolate = 0;
    size_t host_workers = 0;
    // combining queue
    std::atomic<uint32_t> comb_cap_max{5120};         // = coalesce_proofs (readable without `mu`)
    std::atomic<uint64_t> combine_wait_ns{100000};    // a buffer leaves at the latest this long after its first proof arrived ...
    std::atomic<uint64_t> combine_quiet_ns{20000};    // ... or when nothing has joined it for this long
    std::atomic<uint64_t> combine_poll_ns{15000};     // the service thread's polling period while anything is open or in flight
    std::atomic<uint32_t> combine_max_open{4};        // position classes with a buffer of their own; further classes share a CK_MIXED buffer
    std::atomic<uint32_t> combine_busy_chains{2};     // a chain issued beside this many others (in flight or waiting) takes the throughput forms
    std::atomic<uint32_t> combine_inflight{4};        // deadlines seal buffers only while fewer chains than this are in flight: beyond, load widens the chains.  (6 until the narrow
                                                      // chain was rebuilt in round 6; re-swept on the shorter chain, profiles/r06/seal_sweep_1.txt / _2.txt: 256 threads +4 %, tickets +2.5 ... 7 %)
    std::atomic<uint64_t> combine_max_age_ns{1500000}; // ... but no proof waits longer than this for its chain to be issued
    // throughput regime (the chains in flight average >= combine_wide_proofs): fewer, wider chains -- at most combine_inflight_wide run,
    // the time deadline is combine_hold_us instead of combine_wait_us, and a burst smaller than half of what runs waits for company
    std::atomic<uint32_t> combine_wide_proofs{384};
    std::atomic<uint32_t> combine_inflight_wide{3};
    std::atomic<uint64_t> combine_hold_ns{400000};
    // cohort policy (the multiscalar-multiplication and inner-product kinds, see svc_main; range proofs keep the two regimes above): callers come back in the groups their chains released them in
    // -- a chain's worth of requests completes at once, its owners resubmit within tens of microseconds --, so a buffer leaves as soon as the
    // group that just finished is back (no quiet period to sit out), at most combine_cohort_inflight chains run (a chain's latency is nearly
    // flat in its width and grows with every chain beside it: few wide chains beat many narrow ones on throughput AND latency), and a
    // fragment that arrives while the device is busy waits for company
    std::atomic<uint32_t> combine_cohort_inflight{2};
    std::atomic<uint64_t> combine_regroup_ns{60000};     // after a completion with nothing else in flight: this long for its callers to come back before `quiet` may seal
    std::atomic<uint64_t> combine_msm_bytes{32u << 20};   // staging block of a multiscalar-multiplication class (bpgpu_pool_msm_*): items per chain = this / bytes per MSM
    // (Reading an MSM chain's 7 MB of inputs in place from the pinned block instead of staging them was measured and removed: 64 blocking
    // threads 34 -> 26 k MSMs/s, profiles/r05/msm_queue_ab.txt -- 133 000 lanes fetching 64 bytes each across PCIe pay its latency inside
    // the decode kernel, the copy engine streams the same bytes at line rate beside other chains' kernels.)
    std::atomic<uint32_t> combine_mapped_out{1024};   // chains of up to this many proofs write their results straight into the pinned host block (no copy command behind the chain)
    std::atomic<uint32_t> combine_trace{0};           // ring sizes of the timeline records (0 = off)
    std::atomic<int> host_path{1};                    // bpgpu_pool_rangeproof_verify: 1 = through the combining queue, 0 = the slicing workers of round 3
    std::atomic<uint32_t> rr_dev{0};
    std::atomic<uint64_t> gens_epoch{g_pool_uid.fetch_add(1) << 20};   // (process-wide unique: a per-thread "this shape is coalescible" answer never survives into another pool)
    std::atomic<int> closing{0};                      // bpgpu_pool_destroy has begun: new requests are refused, waiting ones are failed
    std::atomic<int64_t> active_calls{0};             // threads inside a combining-queue entry point
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
        int qs = probe_hw_queues(devices[0]);
        if (qs > 0 && qs < 6) {   // a busy or shared device can read low once: only two measurements in a row count
            usleep(20000);
            const int again = probe_hw_queues(devices[0]);
            if (again > qs) qs = again;
        }
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
                b->index = (uint32_t)d->cbufs.size();
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

static void stop_service(bpgpu_pool *p, pool_dev *d);
static void signal_service_stop(pool_dev *d);
// May be called while other threads are inside the pool's blocking entry points (or hold unfinished tickets): new requests are
// refused from now on, requests that wait in a staging buffer are failed (BPGPU_VERDICT_UNDECIDED + an error: nothing reads "verified"),
// chains already on the device finish and are delivered, every sleeper is woken; the call returns when the last such thread has left.
// Tickets of the combining queue stay valid (bpgpu_pool_ticket_wait reads only the ticket); tickets of device batches must have
// been waited for.  Calling INTO the pool after bpgpu_pool_destroy has returned is the caller's bug.
void bpgpu_pool_destroy(bpgpu_pool *p) {
    if (!p) return;
    p->closing.store(1, std::memory_order_seq_cst);
    for (pool_dev *d : p->devs) signal_service_stop(d);
    for (pool_dev *d : p->devs) {
        stop_workers(d);
        stop_service(p, d);
    }
    while (p->active_calls.load(std::memory_order_acquire) != 0) usleep(50);   // (woken callers are on their way out)
    for (pool_dev *d : p->devs) {
        for (bpgpu_ctx *c : d->lanes) bpgpu_ctx_destroy(c);
        for (comb_buf *b : d->cbufs) {
            bpgpu_ctx_destroy(b->ctx);   // (synchronises the device: nothing of the buffer is in flight afterwards)
            if (b->done_ev) (void)hipEventDestroy(b->done_ev);
            if (b->h) (void)hipHostFree(b->h);
            if (b->d) (void)hipFree(b->d);
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
        p->comb_cap_max = (uint32_t)p->coalesce_proofs;
        return BPGPU_OK;
    }
    if (!strcmp(key, "max_chain_proofs")) {
        if (value < 64 || value > (1 << 22)) return pfail(p, BPGPU_ERR_INVALID_ARG, "max_chain_proofs out of range");
        p->max_chain_proofs = (size_t)value;
        if (p->coalesce_proofs > p->max_chain_proofs) p->coalesce_proofs = p->max_chain_proofs;
        p->comb_cap_max = (uint32_t)p->coalesce_proofs;
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
    if (!strcmp(key, "rlc_isolate")) {
        p->rlc_isolate = value != 0;
        return BPGPU_OK;
    }
    if (!strcmp(key, "stat_reset")) {
        p->stat_chains = p->stat_chain_proofs = 0;
        for (pool_dev *d : p->devs) {
            d->stat_chains = d->stat_proofs = d->stat_requests = 0;
            d->stat_issue_ns = d->stat_complete_ns = d->stat_polls = d->stat_deliver_ns = 0;
        }
        return BPGPU_OK;
    }
    if (!strcmp(key, "combine_wait_us") || !strcmp(key, "combine_quiet_us")) {
        if (value < 1 || value > 1000000) return pfail(p, BPGPU_ERR_INVALID_ARG, "%s out of range", key);
        (key[8] == 'w' ? p->combine_wait_ns : p->combine_quiet_ns) = (uint64_t)value * 1000;
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
    if (!strcmp(key, "combine_mapped_out")) {
        if (value < 0 || value > (1 << 22)) return pfail(p, BPGPU_ERR_INVALID_ARG, "%s out of range", key);
        p->combine_mapped_out = (uint32_t)value;
        return BPGPU_OK;
    }
    if (!strcmp(key, "combine_msm_bytes")) {
        if (value < (1 << 16) || value > ((int64_t)1 << 32)) return pfail(p, BPGPU_ERR_INVALID_ARG, "%s out of range", key);
        p->combine_msm_bytes = (uint64_t)value;
        return BPGPU_OK;
    }
    if (!strcmp(key, "combine_trace")) {   // ring sizes of the timeline records (chains; four times as many sampled requests); 0 = off
        if (value < 0 || value > (1 << 20)) return pfail(p, BPGPU_ERR_INVALID_ARG, "%s out of range", key);
        for (pool_dev *d : p->devs) {
            std::lock_guard<std::mutex> g(d->trace_mu);
            d->trace_chains.assign((size_t)value, chain_ev());
            d->trace_reqs.assign((size_t)value * 4, req_ev());
            d->trace_chain_n = d->trace_req_n = 0;
        }
        p->combine_trace = (uint32_t)value;
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

// The constants of the two sealing policies and the service thread's polling period: chosen by the sweeps of round 5 (profiles/r05/
// combine_sweep_first_call.txt, combine_policy_ab.txt), not part of the option surface; the scheduler's test suite (tests/cpu_pool) moves them
// to reach every branch.  hold_us, wide_proofs, inflight_wide, cohort_inflight, regroup_us, poll_us.
int bpgpu_internal_pool_tune(bpgpu_pool *p, const char *key, int64_t value) {
    if (!p || !key || value < 0) return BPGPU_ERR_INVALID_ARG;
    if (!strcmp(key, "hold_us")) p->combine_hold_ns = (uint64_t)value * 1000;
    else if (!strcmp(key, "wide_proofs")) p->combine_wide_proofs = (uint32_t)value;
    else if (!strcmp(key, "inflight_wide")) p->combine_inflight_wide = value < 1 ? 1u : (uint32_t)value;
    else if (!strcmp(key, "cohort_inflight")) p->combine_cohort_inflight = value < 1 ? 1u : (uint32_t)value;
    else if (!strcmp(key, "regroup_us")) p->combine_regroup_ns = (uint64_t)value * 1000;
    else if (!strcmp(key, "poll_us")) p->combine_poll_ns = (value < 1 ? 1 : (uint64_t)value) * 1000;
    else return BPGPU_ERR_INVALID_ARG;
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
    else if (!strcmp(key, "rlc_isolate")) *value = (int64_t)p->rlc_isolate;
    else if (!strcmp(key, "latency_proofs")) *value = (int64_t)p->latency_proofs;
    else if (!strcmp(key, "pair_limit_proofs")) *value = (int64_t)p->pair_limit_proofs;
    else if (!strcmp(key, "host_workers")) *value = (int64_t)p->host_workers;
    else if (!strcmp(key, "stat_chains")) *value = (int64_t)p->stat_chains;
    else if (!strcmp(key, "stat_chain_proofs")) *value = (int64_t)p->stat_chain_proofs;
    else if (!strcmp(key, "stat_last_splits")) *value = (int64_t)p->stat_last_splits;
    else if (!strcmp(key, "combine_wait_us")) *value = (int64_t)(p->combine_wait_ns / 1000);
    else if (!strcmp(key, "combine_quiet_us")) *value = (int64_t)(p->combine_quiet_ns / 1000);
    else if (!strcmp(key, "combine_max_open")) *value = (int64_t)p->combine_max_open;
    else if (!strcmp(key, "combine_busy_chains")) *value = (int64_t)p->combine_busy_chains;
    else if (!strcmp(key, "combine_inflight")) *value = (int64_t)p->combine_inflight;
    else if (!strcmp(key, "combine_max_age_us")) *value = (int64_t)(p->combine_max_age_ns / 1000);
    else if (!strcmp(key, "combine_msm_bytes")) *value = (int64_t)p->combine_msm_bytes;
    else if (!strcmp(key, "combine_mapped_out")) *value = (int64_t)p->combine_mapped_out;
    else if (!strcmp(key, "combine_trace")) *value = (int64_t)p->combine_trace;
    else if (!strcmp(key, "stat_svc_issue_us") || !strcmp(key, "stat_svc_complete_us") || !strcmp(key, "stat_svc_polls") || !strcmp(key, "stat_svc_deliver_us")) {
        uint64_t v = 0;
        for (pool_dev *d : p->devs)
            v += key[9] == 'i' ? d->stat_issue_ns / 1000 : key[9] == 'c' ? d->stat_complete_ns / 1000 : key[9] == 'd' ? d->stat_deliver_ns / 1000 : d->stat_polls.load();
        *value = (int64_t)v;
    }
    else if (!strcmp(key, "stat_active_calls")) *value = (int64_t)p->active_calls.load();
    else if (!strcmp(key, "combine_lanes")) *value = p->devs.empty() ? 0 : (int64_t)p->devs[0]->cbufs.size();
    else if (!strcmp(key, "host_path_combining")) *value = p->host_path;
    else if (!strcmp(key, "stat_combined_chains") || !strcmp(key, "stat_combined_proofs") || !strcmp(key, "stat_combined_requests")) {
        uint64_t v = 0;
        for (pool_dev *d : p->devs) v += key[14] == 'c' ? d->stat_chains.load() : key[14] == 'p' ? d->stat_proofs.load() : d->stat_requests.load();
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
// What the FIRST request of a service used to pay (tools/combine_rate.cpp's outlier report, profiles/r06/call_shape_outliers_before.txt: every
// latency maximum of the call-shape rows -- 16 .. 31 ms against a p99 of 0.9 .. 4 ms -- was call number 0 of its thread, 23 ms): the
// staging blocks of its buffer class (pinned + device allocation), the lane's arena for chains of that shape, and the code objects of the
// chain's kernels (the runtime loads them at their first launch).  bpgpu_pool_gens_* now does all three once: every free staging buffer
// gets blocks for the per-proof class of the set's own shape, every combining lane sizes its arena for it, and one chain of one (invalid)
// proof runs on every device.
static void comb_caps(bpgpu_pool *p, pool_dev *d, const comb_key &key, size_t run, uint32_t *cap, uint32_t *cap_max);
static size_t cbuf_layout(comb_buf *b, const comb_regions &g, uint32_t cap);
static int cbuf_alloc(bpgpu_pool *p, pool_dev *d, comb_buf *b, size_t need);
static void pool_prewarm_buffers(bpgpu_pool *p, size_t n, size_t m) {
    size_t nm = n * m, lg = 0;
    while (((size_t)1 << lg) < nm) lg++;
    if (((size_t)1 << lg) != nm) return;   // (no proof of this shape exists: nothing to foresee)
    comb_key key;
    key.kind = CQ_RP;
    key.a = (uint32_t)n, key.b = (uint32_t)m, key.c = (uint32_t)(32 * (9 + 2 * lg));
    key.mode = CK_UNIFORM;
    for (pool_dev *d : p->devs) {
        std::lock_guard<std::mutex> lk(d->cmu);
        uint32_t cap = 0, cap_max = 0;
        comb_caps(p, d, key, 1, &cap, &cap_max);
        uint32_t n_msm_sized = 0;
        for (comb_buf *b : d->cbufs) {
            if (b->st.load(std::memory_order_acquire) != CB_FREE) continue;   // (a pool that is already serving: whatever is in use stays as it is)
            size_t need = cbuf_layout(b, regions_of(key), cap_max);
            // three of them also get room for a multiscalar-multiplication class (bpgpu_pool_msm_*: combine_msm_bytes of inputs + results) -- a
            // request picks the free buffer with the largest blocks, and two chains in flight + one filling is what the cohort policy runs
            if (n_msm_sized < 3) {
                need = std::max<size_t>(need, (size_t)p->combine_msm_bytes.load(std::memory_order_relaxed) + ((size_t)1 << 20));
                n_msm_sized++;
            }
            if (need > b->mem_cap && cbuf_alloc(p, d, b, need)) continue;
            // (an aggregated shape's arena is ~0.3 MB per proof: sized here for the narrow chains a service's first callers form, a wide
            // chain grows it when it comes -- twelve lanes x 5120 proofs would take 17 GB at nm = 2048 before anybody asked)
            const uint32_t rcap = nm > 256 ? std::min<uint32_t>(cap_max, 512) : cap_max;
            if (b->reserved_n != key.a || b->reserved_m != key.b || b->reserved_len != key.c || b->reserved_cap < rcap) {
                if (bpgpu_internal_rp_reserve(b->ctx, key.a, key.b, key.c, rcap) == BPGPU_OK)
                    b->reserved_n = key.a, b->reserved_m = key.b, b->reserved_len = key.c, b->reserved_cap = rcap;
            }
        }
    }
}
// one chain of one all-zero proof per device through the queue (verdict: not verified; nobody looks at it)
static void pool_prewarm_chain(bpgpu_pool *p, size_t n, size_t m) {
    size_t nm = n * m, lg = 0;
    while (((size_t)1 << lg) < nm) lg++;
    if (((size_t)1 << lg) != nm || getenv("BPGPU_NO_PREWARM")) return;
    const size_t plen = 32 * (9 + 2 * lg);
    std::vector<uint8_t> proof(plen, 0), coms(32 * m, 0), rng(64, 1);
    uint8_t st[BPGPU_TRANSCRIPT_BYTES], v = 0;
    bpgpu_transcript_new((const uint8_t *)"", 0, st);
    const std::string keep = t_pool_err;
    // (the warm chains are nobody's requests: the stat_combined_* counters read as before them -- unless the pool is already serving)
    const bool quiet = p->active_calls.load() == 0;
    std::vector<uint64_t> before;
    for (pool_dev *d : p->devs) before.insert(before.end(), {d->stat_chains.load(), d->stat_proofs.load(), d->stat_requests.load()});
    for (size_t i = 0; i < p->devs.size(); i++)   // (the queue hands consecutive calls to consecutive devices)
        (void)bpgpu_pool_rangeproof_verify_ts(p, n, m, 1, proof.data(), plen, coms.data(), st, 0, rng.data(), &v, nullptr, nullptr);
    // ... and one multiscalar multiplication over the set's generators + 1 536 identity points (all scalars zero): the fused bucket chain's
    // kernels (their own code object) are resident before the first bpgpu_pool_msm_* call (which used to take 13 - 31 ms)
    if (nm <= 4096) {
        const size_t ng = 2 * nm + 2, nu = 1536;
        std::vector<uint8_t> gs(ng * 32, 0), us(nu * 32, 0), up(nu * 32, 0), out(32, 0);
        uint8_t st1 = 0;
        for (size_t i = 0; i < p->devs.size(); i++) (void)bpgpu_pool_msm_batch_shared(p, n, m, 1, nu, gs.data(), us.data(), up.data(), out.data(), &st1);
    }
    if (quiet && p->active_calls.load() == 0)
        for (size_t i = 0; i < p->devs.size(); i++) {
            p->devs[i]->stat_chains.store(before[3 * i]);
            p->devs[i]->stat_proofs.store(before[3 * i + 1]);
            p->devs[i]->stat_requests.store(before[3 * i + 2]);
        }
    t_pool_err = keep;
}

// The generator tables of a pool over N devices are built by N threads at once (a table is seconds of one device's time: 2.3 s for the
// 113 GB of the (64, 1) set; eight devices in a row were eight times that before the first proof could be verified).  Devices that repeat
// in the pool's list (two shards on one GPU: the one-GPU tests) share a table through the context layer's cache and are served in turn.
static int on_every_device(bpgpu_pool *p, const std::function<int(pool_dev *)> &f) {
    for (pool_dev *d : p->devs)
        if (d->lanes.empty()) return BPGPU_ERR_INVALID_ARG;
    std::vector<int> rcs(p->devs.size(), BPGPU_OK);
    std::vector<std::thread> th;
    std::vector<std::vector<size_t>> by_ordinal;   // shards of one physical device run one after the other (they share the table cache)
    for (size_t i = 0; i < p->devs.size(); i++) {
        size_t g = 0;
        while (g < by_ordinal.size() && p->devs[by_ordinal[g][0]]->device != p->devs[i]->device) g++;
        if (g == by_ordinal.size()) by_ordinal.emplace_back();
        by_ordinal[g].push_back(i);
    }
    for (const std::vector<size_t> &grp : by_ordinal)
        th.emplace_back([&, grp] {
            for (size_t i : grp) {
                (void)hipSetDevice(p->devs[i]->device);
                rcs[i] = f(p->devs[i]);
            }
        });
    for (std::thread &t : th) t.join();
    for (size_t i = 0; i < p->devs.size(); i++)
        if (rcs[i]) return pfail(p, rcs[i], "%s", bpgpu_last_error(p->devs[i]->lanes[0]));
    return BPGPU_OK;
}

int bpgpu_pool_gens_create(bpgpu_pool *p, size_t gens_capacity, size_t party_capacity) {
    if (!p || gens_capacity == 0 || party_capacity == 0) return BPGPU_ERR_INVALID_ARG;
    std::unique_lock<std::mutex> lk(p->mu);
    {
        const int rc = on_every_device(p, [&](pool_dev *d) { return bpgpu_gens_create(d->lanes[0], gens_capacity, party_capacity); });
        if (rc) return rc;
    }
    p->gens_epoch.fetch_add(1);
    const int rc = pool_spread_gens(p, gens_capacity, party_capacity);
    if (rc) return rc;
    pool_prewarm_buffers(p, gens_capacity, party_capacity);
    lk.unlock();
    pool_prewarm_chain(p, gens_capacity, party_capacity);
    return BPGPU_OK;
}
int bpgpu_pool_gens_load(bpgpu_pool *p, size_t gens_capacity, size_t party_capacity, const uint8_t *G, const uint8_t *H, const uint8_t B[32],
                         const uint8_t Bb[32]) {
    if (!p || !G || !H || !B || !Bb || gens_capacity == 0 || party_capacity == 0) return BPGPU_ERR_INVALID_ARG;
    std::unique_lock<std::mutex> lk(p->mu);
    {
        const int rc = on_every_device(p, [&](pool_dev *d) { return bpgpu_gens_load(d->lanes[0], gens_capacity, party_capacity, G, H, B, Bb); });
        if (rc) return rc;
    }
    p->gens_epoch.fetch_add(1);
    const int rc = pool_spread_gens(p, gens_capacity, party_capacity);
    if (rc) return rc;
    pool_prewarm_buffers(p, gens_capacity, party_capacity);
    lk.unlock();
    pool_prewarm_chain(p, gens_capacity, party_capacity);
    return BPGPU_OK;
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
// The reference's API is one proof (one multiscalar multiplication) per call, synchronous, from as many threads as the caller likes
// (RangeProof::verify_multiple / verify_multiple_with_rng, src/range_proof/mod.rs:345-353, 455-470; optional_multiscalar_mul at
// mod.rs:421-445, src/r1cs/verifier.rs:459-491, src/inner_product_proof.rs:308-319).  One such call cannot fill a device and a
// launch chain costs ~0.5 ms of latency however narrow it is, so calls that arrive close together must share a chain.  Per device:
//   * callers (any thread): find the open staging buffer of their class and take slots with one compare-and-swap on its
//     reservation word -- no lock; only opening a buffer for a class that has none takes the device's queue lock --, copy their
//     inputs into the pinned block, count themselves in (`written`);
//   * the service thread seals an open buffer by policy (policy_seal: full, quiet, deadline, age; fewer and wider chains once the
//     ones in flight are wide), issues a sealed buffer whose writers are done as ONE chain on the buffer's lane (copy in, the
//     chain, results out, event), polls the events of the chains in flight, publishes completions -- and does nothing else: it
//     never touches a request;
//   * completion: the buffer's futex word takes the incarnation's number; every blocking caller with a piece in it wakes (one
//     FUTEX_WAKE per chain), takes its own results out of the pinned block; pieces of tickets are handed over by the DELIVERY
//     thread (a second thread: issuing the next chain never waits behind a thousand memcpys); whoever takes the last item returns
//     the buffer.
// With all chain slots busy a buffer simply keeps filling: load widens the chains by itself.
static size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

// Layout of a buffer for `cap` items of class `key`: every input region sized for `cap`, then every output region.  The regions
// are ordered small-to-large (regions_of), the largest last, so that a partly filled buffer of a class with small side records
// travels as ONE copy [0, in_off[last] + K in_sz[last]) (comb_issue decides between that and one copy per region).
static size_t cbuf_layout(comb_buf *b, const comb_regions &g, uint32_t cap) {
    size_t o = 0;
    for (uint32_t i = 0; i < g.n_in; i++) b->in_off[i] = o, o += up256((size_t)cap * g.in_sz[i]);
    b->in_end = o;
    for (uint32_t i = 0; i < g.n_out; i++) b->out_off[i] = o, o += up256((size_t)cap * g.out_sz[i]);
    b->total = o;
    return o;
}
// the two staging blocks of a buffer (pinned host + device), at least `need` bytes each (the calling thread's current device is put back)
static int cbuf_alloc(bpgpu_pool *p, pool_dev *d, comb_buf *b, size_t need) {
    {
        int prev = -1;
        (void)hipGetDevice(&prev);
        hipError_t e = hipSetDevice(d->device);
        if (e == hipSuccess && b->h) e = hipHostFree(b->h);
        b->h = b->hd = nullptr;
        if (e == hipSuccess && b->d) e = hipFree(b->d);
        b->d = nullptr;
        b->mem_cap = 0;
        const size_t want = need + need / 4;
        // Kernels of narrow chains write verdicts straight into this block and the host reads them after hipEventQuery on an event without
        // a system-scope release: the block must be fine-grained (coherent) whatever HIP_HOST_COHERENT says.  When the coherent mapped
        // allocation is refused the block is plain pinned memory and every chain takes the copy path (hd stays null).
        bool coherent = true;
        if (e == hipSuccess) {
            e = hipHostMalloc((void **)&b->h, want, hipHostMallocCoherent | hipHostMallocMapped);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                coherent = false;
                b->h = nullptr;
                e = hipHostMalloc((void **)&b->h, want, hipHostMallocDefault);
            }
        }
        if (e == hipSuccess) e = hipMalloc((void **)&b->d, want);
        if (e == hipSuccess && coherent) {
            void *dp = nullptr;
            if (hipHostGetDevicePointer(&dp, b->h, 0) == hipSuccess) b->hd = (char *)dp;
            else (void)hipGetLastError();
        }
        if (prev >= 0) (void)hipSetDevice(prev);
        if (e != hipSuccess) {
            if (b->h) (void)hipHostFree(b->h);
            b->h = nullptr;
            return pfail(p, BPGPU_ERR_HIP, "staging buffers of the combining queue: %s", hipGetErrorString(e));
        }
        b->mem_cap = want;
    }
    return BPGPU_OK;
}
static int cbuf_configure(bpgpu_pool *p, pool_dev *d, comb_buf *b, const comb_key &key, uint32_t cap, uint32_t cap_max) {
    const comb_regions g = regions_of(key);
    const size_t need = cbuf_layout(b, g, cap_max);   // the blocks are sized for the widest chain of this class once and for all
    b->key = key;
    b->reg = g;
    b->cap_max = cap_max;
    if (cap != cap_max) cbuf_layout(b, g, cap);
    if (b->desc.size() < cap_max) b->desc.resize(cap_max);
    if (need > b->mem_cap) return cbuf_alloc(p, d, b, need);   // the first chains of a class that bpgpu_pool_gens_* did not foresee
    return BPGPU_OK;
}

static void trace_chain(bpgpu_pool *p, pool_dev *d, const chain_ev &ev) {
    if (!p->combine_trace.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> g(d->trace_mu);
    if (d->trace_chains.empty()) return;
    d->trace_chains[d->trace_chain_n++ % d->trace_chains.size()] = ev;
}
static void trace_req(bpgpu_pool *p, pool_dev *d, const req_ev &ev) {
    std::lock_guard<std::mutex> g(d->trace_mu);
    if (d->trace_reqs.empty()) return;
    d->trace_reqs[d->trace_req_n++ % d->trace_reqs.size()] = ev;
}

// every item of the buffer has been taken: it can serve another class.  No lock unless somebody waits for a buffer; ONE waiter is
// woken per buffer (it passes the baton on if it ends up not needing it: comb_reserve_slow).
static void cbuf_release(bpgpu_pool *p, comb_buf *b) {
    pool_dev *d = b->dev;
    b->ev.t_free = now_ns();
    trace_chain(p, d, b->ev);
    b->st.store(CB_FREE, std::memory_order_seq_cst);
    TEST_DELAY(2);
    if (d->free_waiters.load(std::memory_order_seq_cst) != 0) {
        { std::lock_guard<std::mutex> lk(d->cmu); }   // (a waiter that has counted itself in is inside free_cv.wait by the time this lock is granted)
        d->free_cv.notify_one();
    }
}
static inline void cbuf_taken(bpgpu_pool *p, comb_buf *b, uint32_t count) {
    const uint32_t K = b->K;   // (read BEFORE the count goes in: afterwards another deliverer may complete the buffer and hand it to a new class)
    if (b->delivered.fetch_add(count, std::memory_order_acq_rel) + count == K) cbuf_release(p, b);
}

// results of items [first, first + count) of a finished buffer -> items [off, ..) of the request
static void comb_deliver(comb_buf *b, uint32_t first, uint32_t count, comb_req *r, size_t off) {
    const comb_regions &g = b->reg;
    if (b->rc) {   // the chain did not run: nothing may read as "verified" / "computed"
        if (r->out[g.fail_out]) memset(r->out[g.fail_out] + off * g.out_sz[g.fail_out], BPGPU_VERDICT_UNDECIDED, (size_t)count * g.out_sz[g.fail_out]);
        std::lock_guard<std::mutex> lk(r->emu);
        if (!r->rc) {
            r->rc = b->rc;
            r->err = b->err;
        }
        return;
    }
    for (uint32_t i = 0; i < g.n_out; i++)
        if (r->out[i] && g.out_sz[i]) memcpy(r->out[i] + off * g.out_sz[i], b->h + b->out_off[i] + (size_t)first * g.out_sz[i], (size_t)count * g.out_sz[i]);
}

// a blocking caller takes the next of its pieces: sleeps until that incarnation of the buffer has its results
static void comb_collect_one(bpgpu_pool *p, comb_req *r) {
    const comb_req::piece pc = r->pieces[r->next_piece++];
    comb_buf *b = pc.b;
    for (;;) {
        const uint32_t ph = b->phase.load(std::memory_order_acquire);
        if ((int32_t)(ph - pc.epoch) >= 0) break;
        futex_wait(&b->phase, ph);
    }
    if (r->traced && !r->ev.t_woken) r->ev.t_woken = now_ns();
    comb_deliver(b, pc.first, pc.count, r, pc.off);
    cbuf_taken(p, b, pc.count);
}

// one launch chain over the K items of a sealed buffer, on the buffer's lane
static void comb_issue(bpgpu_pool *p, pool_dev *d, comb_buf *b, uint32_t inflight, bool more_waiting) {
    hipStream_t s = (hipStream_t)bpgpu_internal_stream(b->ctx);
    const comb_key &k = b->key;
    const comb_regions &g = b->reg;
    const uint32_t K = b->K;
    hipError_t e = hipSuccess;
    if (b->poison.load(std::memory_order_acquire)) {   // a writer could not draw its batching challenge: the chain must not run on predictable bytes
        b->rc = BPGPU_ERR_HIP;
        b->err = "getrandom failed";
        return;
    }
    if (k.kind == CQ_RP && (b->reserved_n != k.a || b->reserved_m != k.b || b->reserved_len != k.c || b->reserved_cap < b->cap_max)) {
        // first chain of this shape on this lane: size the lane's arena for the widest chain now, not in steps on the way up
        (void)bpgpu_internal_rp_reserve(b->ctx, k.a, k.b, k.c, b->cap_max);
        b->reserved_n = k.a, b->reserved_m = k.b, b->reserved_len = k.c, b->reserved_cap = b->cap_max;
    }
    auto cp = [&](size_t off, size_t bytes, bool in) {
        if (e != hipSuccess || !bytes) return;
        e = in ? hipMemcpyAsync(b->d + off, b->h + off, bytes, hipMemcpyHostToDevice, s) : hipMemcpyAsync(b->h + off, b->d + off, bytes, hipMemcpyDeviceToHost, s);
    };
    // inputs: one copy up to the fill of the last region when the unused tails that ride along are small, else one copy per region
    char *ib = b->d;
    {
        const uint32_t last = g.n_in - 1;
        const size_t span = b->in_off[last] + (size_t)K * g.in_sz[last];
        size_t useful = 0;
        for (uint32_t i = 0; i < g.n_in; i++) useful += (size_t)K * g.in_sz[i];
        if (span <= (256u << 10) || span - useful <= useful / 4) cp(0, span, true);
        else
            for (uint32_t i = 0; i < g.n_in; i++) cp(b->in_off[i], (size_t)K * g.in_sz[i], true);
    }
    // outputs: narrow chains write them straight into the pinned host block (device-visible: hipHostMalloc), wide ones into the device
    // block and one copy behind the chain brings them back
    const bool mapped = b->hd && K <= p->combine_mapped_out.load(std::memory_order_relaxed);
    char *ob = mapped ? b->hd : b->d;
    const bool want_opt = b->want_opt.load(std::memory_order_acquire) != 0;
    int rc = BPGPU_OK;
    if (e == hipSuccess) {
        if (k.kind == CQ_RP) {
            const bool per = k.mode != CK_SHARED;
            // what the chain should expect beside it (rp_chain_forms, pick_splits): a chain that leaves a full buffer, or while others
            // wait or run, takes the throughput forms; a lone small one the latency forms
            const int busy = (inflight + (more_waiting ? 1u : 0u) >= p->combine_busy_chains || K > p->latency_proofs) ? 1 : 0;
            uint32_t hint = (uint32_t)(16384 / ((size_t)(inflight + 1) * ((K + 63) / 64)));
            hint = (hint + 7) & ~7u;
            if (hint < 16) hint = 16;
            if (hint > 64) hint = 64;
            rc = bpgpu_internal_rp_verify_chain(b->ctx, k.a, k.b, K, b->d + b->in_off[RP_IN_PROOFS], k.c, b->d + b->in_off[RP_IN_COMS], per ? nullptr : k.shared,
                                                per ? b->d + b->in_off[RP_IN_TS] : nullptr, per ? ob + b->out_off[RP_OUT_TS] : nullptr, k.mode == CK_UNIFORM, k.pos,
                                                k.pos_begin, k.flags, b->d + b->in_off[RP_IN_RNG], ob + b->out_off[RP_OUT_VERDICT],
                                                want_opt ? ob + b->out_off[RP_OUT_MSM] : nullptr, hint, busy);
        } else if (k.kind == CQ_MSM_SHARED) {
            rc = bpgpu_msm_batch_shared_dev(b->ctx, k.a, k.b, K, k.c, ib + b->in_off[2], k.c ? ib + b->in_off[0] : nullptr, k.c ? ib + b->in_off[1] : nullptr,
                                            ob + b->out_off[0], ob + b->out_off[1], nullptr);
        } else if (k.kind == CQ_MSM) {
            const std::vector<uint32_t> nt(K, k.a);
            rc = bpgpu_msm_batch_dev(b->ctx, K, nt.data(), ib + b->in_off[0], ib + b->in_off[1], ob + b->out_off[0], ob + b->out_off[1], nullptr);
        } else {
            rc = bpgpu_ipp_verify_batch_dev(b->ctx, k.a, K, b->d + b->in_off[2], k.b, nullptr, 0, k.shared, b->d + b->in_off[3], b->d + b->in_off[4], b->d + b->in_off[0],
                                            b->d + b->in_off[1], b->d + b->in_off[5], b->d + b->in_off[6], 0, ob + b->out_off[0], want_opt ? ob + b->out_off[1] : nullptr,
                                            nullptr);
        }
        if (rc) b->err = bpgpu_last_error(b->ctx);
    }
    if (e == hipSuccess && !rc) {
        if (!mapped) {
            // the required regions in one copy (their unused tails are small: one byte / one state per item), the optional one on demand
            const bool opt_last = (k.kind == CQ_RP || k.kind == CQ_IPP);
            const uint32_t n_req = opt_last ? g.n_out - 1 : g.n_out;
            uint32_t lastr = n_req - 1;
            while (lastr > 0 && g.out_sz[lastr] == 0) lastr--;
            cp(b->out_off[0], (b->out_off[lastr] - b->out_off[0]) + (size_t)K * g.out_sz[lastr], false);
            if (opt_last && want_opt) cp(b->out_off[g.n_out - 1], (size_t)K * g.out_sz[g.n_out - 1], false);
        }
        if (e == hipSuccess) e = hipEventRecord(b->done_ev, s);
    }
    if (e != hipSuccess) {
        rc = BPGPU_ERR_HIP;
        b->err = std::string("combining queue: ") + hipGetErrorString(e);
        (void)hipGetLastError();
    }
    // the chain did not go out, but its staging copy may be on the stream already: nobody may refill (or free) the blocks under it
    if (rc) (void)hipStreamSynchronize(s);
    b->rc = rc;
}

// the chain of buffer `b` is over (or never went out): publish.  Blocking callers wake and help themselves; tickets are the
// delivery thread's.
static void comb_complete(pool_dev *d, comb_buf *b) {
    b->ev.t_done = now_ns();
    if (b->key.kind == CQ_RP && b->K <= 4 && b->rc == 0 && b->ev.t_issue1 && b->ev.t_done > b->ev.t_issue1) {
        const uint64_t dur = b->ev.t_done - b->ev.t_issue1;
        if (dur < 5000000) d->ema_small_ns = d->ema_small_ns ? (7 * d->ema_small_ns + dur) / 8 : dur;
    }
    d->last_done_K[b->key.kind & 3] = b->K;   // (service thread only: the group that is about to come back)
    d->t_last_done[b->key.kind & 3] = b->ev.t_done;
    const uint32_t epoch = cbs_epoch(b->state.load(std::memory_order_relaxed));
    const bool any_async = b->n_async.load(std::memory_order_relaxed) != 0, any_sync = b->n_sync.load(std::memory_order_relaxed) != 0;
    b->st.store(CB_DONE, std::memory_order_release);
    TEST_DELAY(3);
    b->phase.store(epoch, std::memory_order_release);
    TEST_DELAY(4);
    if (any_sync) futex_wake_all(&b->phase);
    if (any_async) {   // (after this push the buffer may be released, reopened, ... at any moment: nothing of it is touched below)
        {
            std::lock_guard<std::mutex> g(d->dq_mu);
            d->dq.push_back(b);
        }
        d->dq_cv.notify_one();
    }
}

// the delivery thread: results of finished chains -> the buffers of the tickets' owners
static void dlv_main(bpgpu_pool *p, pool_dev *d) {
    for (;;) {
        comb_buf *b;
        {
            std::unique_lock<std::mutex> lk(d->dq_mu);
            d->dq_cv.wait(lk, [&] { return d->dstop || !d->dq.empty(); });
            if (d->dq.empty()) return;
            b = d->dq.front();
            d->dq.pop_front();
        }
        const uint64_t t0 = now_ns();
        b->ev.t_deliv0 = t0;
        const bool tracing = p->combine_trace.load(std::memory_order_relaxed) != 0;
        uint32_t taken = 0;
        for (uint32_t i = 0; i < b->K;) {
            const comb_buf::piece_desc pc = b->desc[i];
            if (pc.async) {
                comb_req *r = pc.req;
                comb_deliver(b, i, pc.count, r, pc.off);
                if (tracing && r->traced) {   // (the owner cannot free the request before the fetch_sub below)
                    r->ev.t_delivered = now_ns();
                    trace_req(p, d, r->ev);
                }
                taken += pc.count;
                // the request may be freed by its owner as soon as `left` reads 0: the fetch_sub is this thread's last access to it
                TEST_DELAY(9);
                const uint32_t old = r->left.fetch_sub(1, std::memory_order_acq_rel);
                if (old == (TKT_WAITING | 1u)) futex_wake_all(&r->left);   // (a wake on an address whose owner has moved on is harmless)
            }
            i += pc.count;
        }
        const uint64_t t1 = now_ns();
        b->ev.t_deliv1 = t1;
        d->stat_deliver_ns.fetch_add(t1 - t0, std::memory_order_relaxed);
        cbuf_taken(p, b, taken);
    }
}

// Should the open buffer leave now?  r items reserved, first one arrived at t_open, the count last moved at t_change.
// `inflight` chains carrying `inflight_items` items of range-proof classes run on the device.
//   latency regime (what runs is narrow, or nothing runs): as soon as nothing has joined for combine_quiet_us, or the first item has
//     waited combine_wait_us -- while fewer than combine_inflight chains run; beyond, the buffer keeps filling until one ends;
//   throughput regime (the chains in flight average >= combine_wide_proofs): at most combine_inflight_wide chains run, the time
//     deadline is combine_hold_us, and a burst that is quiet but smaller than half of what runs waits for company (the next
//     completion's resubmissions) -- chains then stay about as wide as the population of requests allows instead of being cut into
//     combine_wait_us slices of the arrival stream;
//   always: nothing waits longer than combine_max_age_us.
// Cohort policy (the multiscalar-multiplication and inner-product kinds).  `expected` / t_done: width and completion time of the chain of this kind that finished last --
// its callers are on their way back.
//   * the group is back (r >= expected): leave at once -- a lone caller's next call does not sit out a quiet period, sixty-four
//     callers released together travel together again;
//   * nothing runs: leave when nothing has joined for combine_quiet_us -- but within combine_regroup_us of a completion whose group
//     is not back yet, wait for it (a straggler would otherwise leave alone the moment the big chain ends and stay out of step for
//     good); combine_wait_us after the first arrival at the latest;
//   * something runs (fewer than combine_cohort_inflight chains): a chain issued now slows the running one and is slowed by it -- worth
//     it when it carries at least half the width of what runs (and is quiet), or after combine_hold_us;
//   * combine_cohort_inflight chains run: the buffer fills until one ends;
//   * always: nothing waits longer than combine_max_age_us.
static bool policy_seal_cohort(bpgpu_pool *p, uint32_t r, uint64_t now, uint64_t t_open, uint64_t t_change, uint32_t inflight, uint64_t inflight_items,
                               uint32_t n_free, uint32_t expected, uint64_t t_done) {
    const uint64_t age = now - t_open;
    // (the safety net is a chain's duration away, not a policy lever: with combine_cohort_inflight chains running nothing can leave anyway
    // until one ends, and a chain a thousand proofs wide beside another takes ~1 ms)
    if (age >= 3 * p->combine_max_age_ns.load(std::memory_order_relaxed)) return true;
    if (n_free <= 1 && inflight != 0) return false;   // (the last place callers can gather in: see policy_seal)
    if (inflight >= p->combine_cohort_inflight.load(std::memory_order_relaxed)) return false;
    const uint64_t regroup = p->combine_regroup_ns.load(std::memory_order_relaxed), quiet_ns = p->combine_quiet_ns.load(std::memory_order_relaxed);
    const bool fresh = expected != 0 && now - t_done < 4 * regroup;   // (a completion long ago says nothing about who is coming)
    // the group is back: a handful of callers leave at once; a large group while requests are still pouring in takes them along (one poll without a newcomer)
    if (fresh && r >= expected && (expected <= 4 || now - t_change >= quiet_ns / 2)) return true;
    const bool quiet = now - t_change >= quiet_ns;
    if (inflight == 0) {
        const bool regrouping = fresh && now - t_done < regroup;
        if (quiet && !regrouping) return true;
        return age >= p->combine_wait_ns.load(std::memory_order_relaxed);
    }
    const uint64_t avg = inflight_items / inflight;
    if (quiet && (uint64_t)r * 2 >= avg) return true;
    return age >= p->combine_hold_ns.load(std::memory_order_relaxed);
}
extern "C" int bpgpu_internal_policy_seal_cohort(bpgpu_pool *p, uint32_t r, uint64_t age_ns, uint64_t quiet_ns, uint32_t inflight, uint64_t inflight_items, uint32_t n_free,
                                                 uint32_t expected, uint64_t since_done_ns) {
    const uint64_t now = 1ull << 40;
    return policy_seal_cohort(p, r, now, now - age_ns, now - quiet_ns, inflight, inflight_items, n_free, expected, now - since_done_ns) ? 1 : 0;
}

static bool policy_seal(bpgpu_pool *p, uint32_t r, uint64_t now, uint64_t t_open, uint64_t t_change, uint32_t inflight, uint64_t inflight_items, bool rp_class,
                        uint32_t n_free) {
    const uint64_t age = now - t_open;
    if (age >= p->combine_max_age_ns.load(std::memory_order_relaxed)) return true;
    // staging buffers are a resource too: with at most one left while chains run, a buffer that leaves narrow takes the last place
    // callers could gather in -- and they would queue for a buffer instead of filling one (a convoy that keeps itself narrow)
    if (n_free <= 1 && inflight != 0) return false;
    const uint64_t avg = inflight ? inflight_items / inflight : 0;
    const bool wide = rp_class && inflight && avg >= p->combine_wide_proofs.load(std::memory_order_relaxed);
    const uint32_t c_max = wide ? p->combine_inflight_wide.load(std::memory_order_relaxed) : p->combine_inflight.load(std::memory_order_relaxed);
    if (inflight >= c_max) return false;
    const bool quiet = now - t_change >= p->combine_quiet_ns.load(std::memory_order_relaxed);
    if (!wide) return quiet || age >= p->combine_wait_ns.load(std::memory_order_relaxed);
    if (age >= p->combine_hold_ns.load(std::memory_order_relaxed)) return true;
    return quiet && (uint64_t)r * 2 >= avg;
}
extern "C" int bpgpu_internal_policy_seal(bpgpu_pool *p, uint32_t r, uint64_t age_ns, uint64_t quiet_ns, uint32_t inflight, uint64_t inflight_items, uint32_t n_free) {
    return policy_seal(p, r, 1ull << 40, (1ull << 40) - age_ns, (1ull << 40) - quiet_ns, inflight, inflight_items, true, n_free) ? 1 : 0;
}

static void svc_main(bpgpu_pool *p, pool_dev *d) {
    (void)hipSetDevice(d->device);
    prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);   // this thread's timed waits are tens of microseconds: the default slack is 50
    // Hundreds of caller threads become runnable whenever a chain ends; the one thread that issues the next chain must not queue
    // behind them for a time slice.  By default the thread only asks for a negative nice value (best effort, no effect without the
    // privilege); a real-time class inside a host application is the application's decision: environment BPGPU_SERVICE_SCHED=rr
    // (SCHED_RR priority 1, needs CAP_SYS_NICE), =none (leave the thread alone).  The thread sleeps whenever it has nothing to do.
    {
        const char *how = getenv("BPGPU_SERVICE_SCHED");
        if (how && !strcmp(how, "rr")) {
            sched_param sp{};
            sp.sched_priority = 1;
            if (pthread_setschedparam(pthread_self(), SCHED_RR, &sp) != 0) (void)setpriority(PRIO_PROCESS, (id_t)syscall(SYS_gettid), -15);
        } else if (!how || strcmp(how, "none")) {
            (void)setpriority(PRIO_PROCESS, (id_t)syscall(SYS_gettid), -15);
        }
    }
    std::vector<comb_buf *> to_issue, to_complete;
    for (;;) {
        const uint32_t kick0 = d->kick.load(std::memory_order_seq_cst);   // (before the scan: whatever happens after this read rings the bell)
        const bool stopping = d->cstop.load(std::memory_order_acquire);
        bool active = false, all_free = true;
        uint64_t soonest_due = UINT64_MAX;   // when the next narrow chain in flight should be complete (ema_small_ns)
        const uint64_t now = now_ns();
        d->stat_polls.fetch_add(1, std::memory_order_relaxed);
        uint32_t inflight = 0, waiting = 0, n_free = 0;
        uint64_t inflight_items = 0, inflight_all_items = 0;   // (range-proof classes only / every kind)
        for (comb_buf *b : d->cbufs) {
            const int st = b->st.load(std::memory_order_acquire);
            if (st == CB_FREE) n_free++;
            if (st == CB_ISSUED || st == CB_ISSUING || st == CB_SEALED) {
                inflight++;
                inflight_all_items += b->K;
                if (b->key.kind == CQ_RP) inflight_items += b->K;
            }
            if (st == CB_OPEN || st == CB_SEALED) waiting++;
        }
        to_issue.clear();
        to_complete.clear();
        for (comb_buf *b : d->cbufs) {
            int st = b->st.load(std::memory_order_acquire);
            if (st != CB_FREE) all_free = false;
            if (st == CB_OPEN) {
                active = true;
                uint64_t s = b->state.load(std::memory_order_acquire);
                const uint32_t e = cbs_epoch(s), r = cbs_reserved(s);
                if (e == b->sealed_epoch) continue;   // (belt and braces beside the store order in comb_reserve_slow: never act on the previous incarnation's word)
                if (b->seen_epoch != e) b->seen_epoch = e, b->seen_reserved = r, b->t_change = b->t_open;
                else if (r != b->seen_reserved) b->seen_reserved = r, b->t_change = now;
                bool seal = cbs_sealed(s);   // a caller took the last slot
                const uint32_t kd = b->key.kind & 3;
                // per kind of work, as the device measured best (profiles/r05/combine_policy_ab.txt; "cohorts for every kind" and "regimes for
                // every kind" were options until round 6): range proofs keep the regimes -- four to six narrow chains in flight overlap well, 64
                // blocking threads 81 k/s against 66 k/s in cohorts, 16 x 128 tickets 795 against 700 k/s -- plus the cohort rule for a handful
                // of callers (a lone caller's next call leaves at once: 0.557 -> 0.529 ms per call); multiscalar multiplications and
                // inner-product proofs, whose chains are a dozen launches of bucket work each, travel in cohorts (64 threads 26.7 -> 32.5 k MSMs/s)
                const bool cohort = b->key.kind != CQ_RP;
                bool due = cohort ? policy_seal_cohort(p, r, now, b->t_open, b->t_change, inflight, inflight_all_items, n_free, d->last_done_K[kd], d->t_last_done[kd])
                                  : policy_seal(p, r, now, b->t_open, b->t_change, inflight, inflight_items, b->key.kind == CQ_RP, n_free);
                if (!due && !cohort && d->last_done_K[kd] != 0 && d->last_done_K[kd] <= 4 && r >= d->last_done_K[kd] &&
                    now - d->t_last_done[kd] < 4 * p->combine_regroup_ns.load(std::memory_order_relaxed) && inflight == 0)
                    due = true;   // the few callers the last chain released are all back and nothing else runs: nothing to wait for
                if (!seal && (stopping || due)) {
                    s = b->state.fetch_or(CBS_SEALED, std::memory_order_acq_rel);   // (slots taken since the load above are in the value this returns)
                    TEST_DELAY(5);
                    seal = true;
                }
                if (seal) {
                    b->sealed_epoch = e;
                    b->K = cbs_reserved(s);
                    b->ev = chain_ev();
                    b->ev.buf = b->index, b->ev.epoch = e, b->ev.K = b->K, b->ev.kind = b->key.kind, b->ev.cap = b->cap.load(std::memory_order_relaxed);
                    b->ev.t_open = b->t_open, b->ev.t_seal = now > b->t_open ? now : b->t_open, b->ev.inflight = inflight;   // (`now` was read before the scan)
                    b->st.store(CB_SEALED, std::memory_order_release);
                    st = CB_SEALED;
                    inflight++;   // (counts against the limits at once: two buffers due in the same pass)
                    inflight_all_items += b->K;
                    if (b->key.kind == CQ_RP) inflight_items += b->K;
                }
            }
            if (st == CB_SEALED) {
                active = true;
                if (b->written.load(std::memory_order_acquire) == b->K) {
                    b->st.store(CB_ISSUING, std::memory_order_release);
                    to_issue.push_back(b);
                }
            } else if (st == CB_ISSUED) {
                active = true;
                const hipError_t e = hipEventQuery(b->done_ev);
                if (e != hipErrorNotReady) {
                    if (e != hipSuccess) {
                        b->rc = BPGPU_ERR_HIP;
                        b->err = std::string("combining queue: ") + hipGetErrorString(e);
                    }
                    to_complete.push_back(b);
                } else {
                    (void)hipGetLastError();
                    if (b->key.kind == CQ_RP && b->K <= 4 && d->ema_small_ns) {   // a narrow chain's caller is blocked on exactly this: know when it is due (one call -3 us: profiles/r06/due_poll_ab.txt)
                        const uint64_t due = b->ev.t_issue1 + d->ema_small_ns;
                        if (due < soonest_due) soonest_due = due;
                    }
                }
            } else if (st == CB_DONE) active = true;   // (being delivered)
        }
        if (!to_issue.empty() || !to_complete.empty()) {
            const uint64_t ta = now_ns();
            for (comb_buf *b : to_complete) comb_complete(d, b);
            const uint64_t tb = now_ns();
            uint32_t running = 0;
            for (comb_buf *b : d->cbufs) running += (b->st.load(std::memory_order_relaxed) == CB_ISSUED);
            for (comb_buf *b : to_issue) {
                waiting--;
                b->n_sync.load(std::memory_order_acquire);
                b->ev.n_sync = b->n_sync.load(std::memory_order_relaxed), b->ev.n_async = b->n_async.load(std::memory_order_relaxed);
                b->ev.t_issue0 = now_ns();
                if (stopping) {   // the pool is being destroyed: what has not left does not leave
                    b->rc = BPGPU_ERR_INVALID_ARG;
                    b->err = "the pool was destroyed while this request waited in the combining queue";
                } else comb_issue(p, d, b, running, waiting > 0);
                b->ev.t_issue1 = now_ns();
                if (b->rc) comb_complete(d, b);   // never went out
                else {
                    running++;
                    d->stat_chains.fetch_add(1, std::memory_order_relaxed);
                    d->stat_proofs.fetch_add(b->K, std::memory_order_relaxed);
                    if (b->key.kind == CQ_RP) d->recent_K.store(b->K, std::memory_order_relaxed);
                    b->st.store(CB_ISSUED, std::memory_order_release);
                }
            }
            d->stat_complete_ns.fetch_add(tb - ta, std::memory_order_relaxed);
            d->stat_issue_ns.fetch_add(now_ns() - tb, std::memory_order_relaxed);
            continue;   // look again at once: issuing took tens of microseconds
        }
        if (stopping && all_free && p->active_calls.load(std::memory_order_acquire) == 0) return;
        // sleep: a poll period while anything is open or in flight, until the doorbell rings otherwise
        TEST_DELAY(11);
        d->svc_sleeping.store(1, std::memory_order_seq_cst);
        TEST_DELAY(12);
        if (active || stopping) {
            // a chain of a proof or two that is due within the polling period: sleep up to 3 us before it is due, then poll without sleeping
            // (at most 40 us past the estimate) -- its caller sits on a futex for exactly this completion (one blocking call: -7 us)
            uint64_t nap = (uint64_t)p->combine_poll_ns;
            if (soonest_due != UINT64_MAX && !stopping) {
                const uint64_t t = now_ns();
                if (t + 3000 >= soonest_due) nap = t < soonest_due + 40000 ? 0 : nap;
                else if (soonest_due - t - 3000 < nap) nap = soonest_due - t - 3000;
            }
            if (nap) futex_wait_ns(&d->kick, kick0, nap);
        } else futex_wait(&d->kick, kick0);
        d->svc_sleeping.store(0, std::memory_order_seq_cst);
    }
}
static inline void svc_kick(pool_dev *d) {
    d->kick.fetch_add(1, std::memory_order_seq_cst);
    if (d->svc_sleeping.load(std::memory_order_seq_cst)) syscall(SYS_futex, (uint32_t *)&d->kick, FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0);
}

// bpgpu_pool_destroy: first every device is told to stop (callers blocked on ANY device keep `active_calls` up, and a service thread
// leaves only when that reads zero), then the threads are joined
static void signal_service_stop(pool_dev *d) {
    d->cstop.store(true, std::memory_order_release);
    svc_kick(d);
    d->free_cv.notify_all();   // callers waiting for a free buffer see `closing`
}
static void stop_service(bpgpu_pool *, pool_dev *d) {
    signal_service_stop(d);
    if (d->svc.joinable()) d->svc.join();
    {
        std::lock_guard<std::mutex> lk(d->dq_mu);
        d->dstop = true;
    }
    d->dq_cv.notify_all();
    if (d->dlv.joinable()) d->dlv.join();
}

// STROBE position class of a 208-byte state
static inline void ts_class(const uint8_t *st, comb_key &k) {
    k.pos = st[200];
    k.pos_begin = st[201];
    k.flags = st[202];
}
static inline bool ts_ok(const uint8_t *st) { return st[200] < 166 && st[201] <= 166; }

// the number of a class on a device: interned under the queue lock the first time a thread meets the class, then found in the
// thread's own small cache (no shared memory touched).  Numbers are never reused.
static uint64_t class_id_locked(pool_dev *d, const comb_key &key) {
    for (pool_dev::cls_ent &c : d->classes)
        if (c.key == key) {
            c.last_use = ++d->class_tick;
            return c.id;
        }
    if (d->classes.size() >= 64) {   // forget the class that has not been asked for for the longest time (a buffer still open under its number simply finishes)
        size_t lru = 0;
        for (size_t i = 1; i < d->classes.size(); i++)
            if (d->classes[i].last_use < d->classes[lru].last_use) lru = i;
        d->classes.erase(d->classes.begin() + (long)lru);
    }
    d->classes.push_back({key, d->next_class_id++, ++d->class_tick});
    return d->classes.back().id;
}
struct tls_class {
    uint64_t uid = 0;
    pool_dev *d = nullptr;
    comb_key key;
    uint64_t id = 0;
};
static uint64_t class_id_of(bpgpu_pool *p, pool_dev *d, const comb_key &key) {
    static thread_local tls_class cache[4];
    static thread_local unsigned next = 0;
    for (const tls_class &c : cache)
        if (c.uid == p->uid && c.d == d && c.key == key) return c.id;
    uint64_t id;
    {
        std::lock_guard<std::mutex> lk(d->cmu);
        id = class_id_locked(d, key);
    }
    tls_class &c = cache[next++ & 3];
    c.uid = p->uid, c.d = d, c.key = key, c.id = id;
    return id;
}

struct comb_slot {
    comb_buf *b = nullptr;
    uint32_t epoch = 0, first = 0, take = 0;
    bool filled = false;   // this reservation took the last slot: the caller seals
};
// slots in an open buffer of class `id`: one compare-and-swap, no lock
static bool comb_reserve_fast(pool_dev *d, uint64_t id, size_t run, comb_slot &out) {
    for (comb_buf *b : d->cbufs) {
        uint64_t s = b->state.load(std::memory_order_acquire);
        for (int tries = 0; tries < 256; tries++) {
            if (cbs_sealed(s) || b->class_id.load(std::memory_order_relaxed) != id) break;
            const uint32_t cap = b->cap.load(std::memory_order_relaxed), r = cbs_reserved(s);
            if (r >= cap) break;   // (full: whoever took the last slot is about to seal it)
            const uint32_t t = (uint32_t)(run < (size_t)(cap - r) ? run : (size_t)(cap - r));
            // succeeds only on the incarnation `s` belongs to (its number is in the word), so `id` and `cap`, written before that
            // incarnation was published, are the ones read above
            if (b->state.compare_exchange_weak(s, s + t, std::memory_order_acq_rel, std::memory_order_acquire)) {
                out.b = b, out.epoch = cbs_epoch(s), out.first = r, out.take = t, out.filled = (r + t == cap);
                return true;
            }
        }
    }
    return false;
}

// capacity of a new buffer of class `key`
static void comb_caps(bpgpu_pool *p, pool_dev *d, const comb_key &key, size_t run, uint32_t *cap, uint32_t *cap_max) {
    if (key.kind == CQ_RP) {
        size_t cm = p->comb_cap_max.load(std::memory_order_relaxed);
        if (cm < 1) cm = 1;
        // light traffic opens a small buffer (its staging copy carries the small regions whole), heavy traffic a full-width one
        const size_t want = std::max<size_t>((size_t)4 * d->recent_K.load(std::memory_order_relaxed), run);
        *cap = (uint32_t)(want <= 256 ? std::min<size_t>(256, cm) : want <= 1024 ? std::min<size_t>(1024, cm) : cm);
        *cap_max = (uint32_t)cm;
        return;
    }
    const comb_regions g = regions_of(key);
    size_t item = 0;
    for (uint32_t i = 0; i < g.n_in; i++) item += g.in_sz[i];
    size_t c = (size_t)p->combine_msm_bytes.load(std::memory_order_relaxed) / (item ? item : 1);
    if (c < 1) c = 1;
    if (c > 1024) c = 1024;
    *cap = *cap_max = (uint32_t)c;
}

// No open buffer of the class had room: open one (queue lock), or wait for one to come back.  `key` / `id` may change to the
// catch-all class (CK_MIXED) when too many transcript-position classes are open.
static int comb_reserve_slow(bpgpu_pool *p, pool_dev *d, comb_req *r, comb_key &key, uint64_t &id, size_t &run, size_t run_mixed, comb_slot &out) {
    std::unique_lock<std::mutex> lk(d->cmu);
    for (;;) {
        if (p->closing.load(std::memory_order_acquire)) return pfail(p, BPGPU_ERR_INVALID_ARG, "the pool is being destroyed");
        if (!d->svc_running) {
            d->svc = std::thread(svc_main, p, d);
            d->dlv = std::thread(dlv_main, p, d);
            d->svc_running = true;
        }
        if (comb_reserve_fast(d, id, run, out)) {   // (opened by another thread meanwhile)
            if (d->free_waiters.load(std::memory_order_relaxed)) {   // this thread may have been woken for a free buffer it does not need: pass it on
                lk.unlock();
                d->free_cv.notify_one();
            }
            return BPGPU_OK;
        }
        uint32_t n_open_classes = 0;
        comb_buf *freeb = nullptr;
        bool mixed_open = false;
        for (comb_buf *cb : d->cbufs) {
            const int st = cb->st.load(std::memory_order_acquire);   // (FREE: everything its last deliverers did is visible before the buffer is rewritten)
            if (st == CB_FREE) {
                if (!freeb || cb->mem_cap > freeb->mem_cap) freeb = cb;   // (prefer one whose blocks are already large enough)
                continue;
            }
            if (st != CB_OPEN || cbs_sealed(cb->state.load(std::memory_order_relaxed))) continue;
            if (cb->key.kind == CQ_RP && cb->key.mode == CK_UNIFORM) n_open_classes++;
            if (cb->key.kind == CQ_RP && cb->key.mode == CK_MIXED && cb->key.a == key.a && cb->key.b == key.b && cb->key.c == key.c) mixed_open = true;
        }
        if (key.kind == CQ_RP && key.mode == CK_UNIFORM && !(freeb && n_open_classes < p->combine_max_open) && (mixed_open || freeb)) {
            // too many position classes open at once (or no buffer left for a new one): the catch-all, replayed byte-wise
            key.mode = CK_MIXED;
            key.pos = key.pos_begin = key.flags = 0;
            run = run_mixed;
            id = class_id_locked(d, key);
            continue;
        }
        if (freeb) {
            uint32_t cap, cap_max;
            comb_caps(p, d, key, run, &cap, &cap_max);
            const int rc = cbuf_configure(p, d, freeb, key, cap, cap_max);
            if (rc) return rc;
            comb_buf *b = freeb;
            const uint32_t t = (uint32_t)(run < (size_t)cap ? run : (size_t)cap);
            b->class_id.store(id, std::memory_order_relaxed);
            b->cap.store(cap, std::memory_order_relaxed);
            b->t_open = now_ns();
            b->K = 0;
            b->rc = 0;
            b->err.clear();
            b->written.store(0, std::memory_order_relaxed);
            b->delivered.store(0, std::memory_order_relaxed);
            b->n_sync.store(0, std::memory_order_relaxed);
            b->n_async.store(0, std::memory_order_relaxed);
            b->want_opt.store(0, std::memory_order_relaxed);
            b->poison.store(0, std::memory_order_relaxed);
            const uint32_t e = cbs_epoch(b->state.load(std::memory_order_relaxed)) + 1;
            // The reservation word FIRST, then the buffer's state.  The other order (round 5's first build) let the service thread find
            // `st == CB_OPEN` beside the PREVIOUS incarnation's word -- sealed, with that chain's width in it: it "sealed" the new incarnation
            // at the old width, callers went on reserving in a word that was never sealed, `written` passed K and the buffer waited for
            // `written == K` forever (caught by tools/combine_rate.cpp's watchdog: "SEALED epoch 496 sealed 0 reserved 256 K 81 written 256";
            // three of twenty-three runs of the first sweep).  Callers look at the word only; the service thread looks at `st` first and its
            // acquire load of CB_OPEN now brings the new word with it.
            b->state.store(cbs_pack(e, false, t), std::memory_order_release);   // publishes the incarnation (its first reservation is ours)
            TEST_DELAY(1);   // (between the two stores)
            b->st.store(CB_OPEN, std::memory_order_release);
            out.b = b, out.epoch = e, out.first = 0, out.take = t, out.filled = (t == cap);
            const bool others_wait = d->free_waiters.load(std::memory_order_relaxed) != 0;
            lk.unlock();
            svc_kick(d);   // the service thread may be asleep with nothing to watch
            if (others_wait) d->free_cv.notify_all();   // whoever waits for a buffer of this class can join this one now
            return BPGPU_OK;
        }
        // every lane is filling or running: take a finished piece of our own meanwhile, or wait for a buffer to come back
        if (!r->async && r->next_piece < r->pieces.size()) {
            lk.unlock();
            comb_collect_one(p, r);
            lk.lock();
        } else {
            d->free_waiters.fetch_add(1, std::memory_order_seq_cst);
            TEST_DELAY(10);
            bool any_free = false;   // (a buffer released between the scan above and the count: its releaser saw no waiter)
            for (comb_buf *cb : d->cbufs) any_free = any_free || cb->st.load(std::memory_order_seq_cst) == CB_FREE;
            if (!any_free) d->free_cv.wait(lk);
            d->free_waiters.fetch_sub(1, std::memory_order_seq_cst);
        }
    }
}

// items [lo, hi) of the request go to device d's queue.  src[i]: the request's region-i inputs of item `lo` (null: the class fills
// the region itself -- rng bytes drawn here, the shared start state replicated).
static int comb_place(bpgpu_pool *p, pool_dev *d, comb_req *r, const comb_key &base, const uint8_t *const src[CQ_MAX_IN], const uint8_t *shared, size_t lo, size_t hi) {
    const size_t TS = BPGPU_TRANSCRIPT_BYTES;
    const uint8_t *ts_in = base.kind == CQ_RP ? src[RP_IN_TS] : nullptr;
    size_t off = lo;
    while (off < hi) {
        comb_key key = base;
        size_t run = hi - off;
        if (key.kind == CQ_RP && key.mode == CK_UNIFORM) {
            const uint8_t *st0 = ts_in ? ts_in + (off - lo) * TS : shared;
            ts_class(st0, key);
            if (ts_in) {   // the stretch of this request that sits at one STROBE position
                size_t e = off + 1;
                while (e < hi && ts_in[(e - lo) * TS + 200] == st0[200] && ts_in[(e - lo) * TS + 201] == st0[201] && ts_in[(e - lo) * TS + 202] == st0[202]) e++;
                run = e - off;
            }
        }
        uint64_t id = class_id_of(p, d, key);
        comb_slot sl;
        if (!comb_reserve_fast(d, id, run, sl)) {
            const int rc = comb_reserve_slow(p, d, r, key, id, run, hi - off, sl);
            if (rc) return rc;
        }
        comb_buf *b = sl.b;
        const uint32_t first = sl.first, take = sl.take;
        if (r->traced && !r->ev.t_reserved) r->ev.t_reserved = now_ns(), r->ev.buf = b->index, r->ev.epoch = sl.epoch;
        // the piece's record, then the inputs, then `written`: the service thread issues the chain only after every reserved slot
        // has been counted in, the deliverers read the records only after the chain
        TEST_DELAY(6);
        b->desc[first] = {r, take, r->async ? 1u : 0u, off};
        if (r->async) {
            r->left.fetch_add(1, std::memory_order_relaxed);
            b->n_async.fetch_add(1, std::memory_order_relaxed);
        } else {
            r->pieces.push_back({b, sl.epoch, first, take, off});
            b->n_sync.fetch_add(1, std::memory_order_relaxed);
        }
        const comb_regions &g = b->reg;
        const bool opt_last = (key.kind == CQ_RP || key.kind == CQ_IPP);
        if (opt_last && r->out[g.n_out - 1]) b->want_opt.store(1, std::memory_order_relaxed);
        if (sl.filled) {
            TEST_DELAY(8);
            b->state.fetch_or(CBS_SEALED, std::memory_order_acq_rel);
            svc_kick(d);
        }
        // ---- inputs into the pinned block (no lock held) ----
        for (uint32_t i = 0; i < g.n_in; i++) {
            const size_t sz = g.in_sz[i];
            if (!sz) continue;
            char *dst = b->h + b->in_off[i] + (size_t)first * sz;
            if (src[i]) memcpy(dst, src[i] + (off - lo) * sz, (size_t)take * sz);
            else if (key.kind == CQ_RP && i == RP_IN_RNG) {
                if (!fast_random((uint8_t *)dst, (size_t)take * 64)) b->poison.store(1, std::memory_order_release);
            } else if (key.kind == CQ_RP && i == RP_IN_TS) {
                for (uint32_t j = 0; j < take; j++) memcpy(dst + (size_t)j * TS, shared, TS);
            }
        }
        if (r->traced) r->ev.t_written = now_ns();   // (before the count: a ticket's record is read by the delivery thread)
        TEST_DELAY(7);
        b->written.fetch_add(take, std::memory_order_release);
        off += take;
    }
    return BPGPU_OK;
}

// requests no chain can take: malformed lengths, parameter errors, missing generators -- the ordinary entry point on the odd-jobs
// context reports them proof by proof (ProofError::FormatError / InvalidBitsize / InvalidGeneratorsLength, mod.rs:358-366, 505-510)
static int comb_direct_rp(bpgpu_pool *p, pool_dev *d, comb_req *r, size_t n, size_t m, size_t proof_len, const uint8_t *const src[CQ_MAX_IN], const uint8_t *shared) {
    std::lock_guard<std::mutex> lk(d->misc_mu);
    const uint8_t *ts_in = src[RP_IN_TS];
    const int rc = bpgpu_rangeproof_verify_batch_ts(d->misc, n, m, r->nbatch, src[RP_IN_PROOFS], proof_len, src[RP_IN_COMS], ts_in ? ts_in : shared,
                                                    ts_in ? BPGPU_TRANSCRIPT_BYTES : 0, src[RP_IN_RNG], r->out[RP_OUT_VERDICT], r->out[RP_OUT_MSM], r->out[RP_OUT_TS]);
    if (rc) return pfail(p, rc, "%s", bpgpu_last_error(d->misc));
    return BPGPU_OK;
}

// Nothing of a request may read "verified" / "computed" unless a chain delivered it: the status bytes start as BPGPU_VERDICT_UNDECIDED, so
// every early return (refused while the pool is destroyed, argument errors, a placement that broke off) leaves them that way.
static inline void mark_undecided(uint8_t *status, size_t nbatch) {
    if (status && nbatch) memset(status, BPGPU_VERDICT_UNDECIDED, nbatch);
}

// every combining-queue entry point runs inside one of these: bpgpu_pool_destroy waits for the last to leave
struct call_guard {
    bpgpu_pool *p;
    bool ok;
    explicit call_guard(bpgpu_pool *pool) : p(pool) {
        p->active_calls.fetch_add(1, std::memory_order_acq_rel);
        ok = !p->closing.load(std::memory_order_seq_cst);
    }
    ~call_guard() { p->active_calls.fetch_sub(1, std::memory_order_acq_rel); }
};

// place (one device for a small request, a contiguous shard per device for a large one: proofs / MSMs are independent units,
// SURVEY 8e; the "gather" is the placement of every piece's results at its offset of the caller's buffers), then -- blocking
// requests -- collect
static int comb_run_placed(bpgpu_pool *p, comb_req *r, const comb_key &base, const uint8_t *const src[CQ_MAX_IN], const size_t in_sz[CQ_MAX_IN], const uint8_t *shared,
                           pool_dev *d0, size_t shard_min) {
    const size_t ndev = p->devs.size();
    if (p->combine_trace.load(std::memory_order_relaxed)) {
        static thread_local uint32_t tick = 0;
        if ((tick++ & 7) == 0) {
            r->traced = true;
            r->ev.t_submit = now_ns();
            r->ev.async = r->async, r->ev.nbatch = (uint32_t)r->nbatch;
        }
    }
    d0->stat_requests.fetch_add(1, std::memory_order_relaxed);
    int rc = BPGPU_OK;
    if (ndev == 1 || r->nbatch < shard_min * ndev) {
        rc = comb_place(p, d0, r, base, src, shared, 0, r->nbatch);
    } else {
        for (size_t di = 0; di < ndev && !rc; di++) {
            const size_t lo = r->nbatch * di / ndev, hi = r->nbatch * (di + 1) / ndev;
            if (hi <= lo) continue;
            const uint8_t *s2[CQ_MAX_IN];
            for (int i = 0; i < CQ_MAX_IN; i++) s2[i] = src[i] ? src[i] + lo * in_sz[i] : nullptr;
            rc = comb_place(p, p->devs[di], r, base, s2, shared, lo, hi);
        }
    }
    if (r->async) return rc;   // (comb_finish_async drops the placement guard)
    while (r->next_piece < r->pieces.size()) comb_collect_one(p, r);   // (also after a placement error: nothing stays referenced behind the caller's back)
    if (r->traced) {
        r->ev.t_delivered = now_ns();
        trace_req(p, d0, r->ev);
    }
    if (!rc && r->rc) rc = pfail(p, r->rc, "%s", r->err.c_str());
    return rc;
}
// a ticket's placement is over: the guard goes; the ticket is complete when its last piece has been delivered
static void comb_finish_async(comb_req *r, int rc) {
    if (rc) {
        std::lock_guard<std::mutex> g(r->emu);
        if (!r->rc) {
            r->rc = rc;
            r->err = t_pool_err;
        }
    }
    const uint32_t old = r->left.fetch_sub(1, std::memory_order_acq_rel);
    if (old == (TKT_WAITING | 1u)) futex_wake_all(&r->left);
}

// a ticket's owner sleeps until its last piece has been delivered
static void ticket_block(comb_req *r) {
    for (;;) {
        uint32_t v = r->left.load(std::memory_order_acquire);
        if ((v & ~TKT_WAITING) == 0) return;
        if (!(v & TKT_WAITING)) {
            if (!r->left.compare_exchange_weak(v, v | TKT_WAITING, std::memory_order_acq_rel, std::memory_order_acquire)) continue;
            v |= TKT_WAITING;
        }
        futex_wait(&r->left, v);
    }
}

// ---- range proofs through the queue -------------------------------------------------------------------------------------
static int rp_comb_run(bpgpu_pool *p, comb_req *r, size_t n, size_t m, size_t proof_len, const uint8_t *proofs, const uint8_t *coms, const uint8_t *rng,
                       const uint8_t *transcripts, size_t stride) {
    if (!p || p->devs.empty()) return BPGPU_ERR_INVALID_ARG;
    if (r->nbatch == 0) return BPGPU_OK;
    if (!proofs || !r->out[RP_OUT_VERDICT] || (m && !coms) || !transcripts) return pfail(p, BPGPU_ERR_INVALID_ARG, "null argument");
    if (stride != 0 && stride != BPGPU_TRANSCRIPT_BYTES) return pfail(p, BPGPU_ERR_INVALID_ARG, "transcript_stride neither 0 nor BPGPU_TRANSCRIPT_BYTES");
    if (r->nbatch > 0x7fffffffu / 64) return pfail(p, BPGPU_ERR_INVALID_ARG, "batch too large");
    for (size_t i = 0; i < (stride ? r->nbatch : 1); i++)
        if (!ts_ok(transcripts + i * BPGPU_TRANSCRIPT_BYTES)) return pfail(p, BPGPU_ERR_INVALID_ARG, "malformed transcript state %zu", i);
    const uint8_t *shared = stride ? nullptr : transcripts;
    const uint8_t *src[CQ_MAX_IN] = {nullptr};
    src[RP_IN_COMS] = coms, src[RP_IN_RNG] = rng, src[RP_IN_TS] = stride ? transcripts : nullptr, src[RP_IN_PROOFS] = proofs;
    comb_key base;
    base.kind = CQ_RP;
    base.a = (uint32_t)n;
    base.b = (uint32_t)m;
    base.c = (uint32_t)proof_len;
    if (shared && !r->out[RP_OUT_TS]) {
        base.mode = CK_SHARED;
        memcpy(base.shared, shared, 203);   // (bytes behind the STROBE bookkeeping carry nothing)
    } else base.mode = CK_UNIFORM;
    const size_t ndev = p->devs.size();
    pool_dev *d0 = p->devs[p->rr_dev.fetch_add(1, std::memory_order_relaxed) % ndev];
    // can chains take this shape?  (asked of the odd-jobs context; the answer is remembered per thread until generators change)
    static thread_local struct {
        uint64_t uid;
        size_t n, m, len;
        uint64_t epoch;
    } ok_shape = {0, 0, 0, 0, 0};
    const uint64_t epoch = p->gens_epoch.load(std::memory_order_acquire);
    bool ok = ok_shape.uid == p->uid && ok_shape.n == n && ok_shape.m == m && ok_shape.len == proof_len && ok_shape.epoch == epoch;
    if (!ok && n <= 0xffff && m <= 0xffffff && proof_len <= 0xffffff && bpgpu_internal_rp_coalescible(d0->misc, n, m, proof_len)) {
        ok_shape = {p->uid, n, m, proof_len, epoch};
        ok = true;
    }
    if (!ok) return comb_direct_rp(p, d0, r, n, m, proof_len, src, shared);
    const comb_regions g = regions_of(base);
    size_t in_sz[CQ_MAX_IN] = {0};
    for (uint32_t i = 0; i < g.n_in; i++) in_sz[i] = g.in_sz[i];
    in_sz[RP_IN_TS] = BPGPU_TRANSCRIPT_BYTES;   // (the caller's array stride, whatever mode the pieces end up in)
    return comb_run_placed(p, r, base, src, in_sz, shared, d0, 512);
}

extern "C" {

int bpgpu_pool_rangeproof_verify_ts(bpgpu_pool *p, size_t n, size_t m, size_t nbatch, const uint8_t *proofs, size_t proof_len, const uint8_t *commitments,
                                    const uint8_t *transcripts, size_t transcript_stride, const uint8_t *rng64, uint8_t *verdict, uint8_t *msm_out,
                                    uint8_t *transcripts_out) {
    if (!p) return BPGPU_ERR_INVALID_ARG;
    mark_undecided(verdict, nbatch);
    call_guard cg(p);
    if (!cg.ok) return pfail(p, BPGPU_ERR_INVALID_ARG, "the pool is being destroyed");
    comb_req r;
    r.nbatch = nbatch;
    r.out[RP_OUT_VERDICT] = verdict, r.out[RP_OUT_TS] = transcripts_out, r.out[RP_OUT_MSM] = msm_out;
    return rp_comb_run(p, &r, n, m, proof_len, proofs, commitments, rng64, transcripts, transcript_stride);
}

int bpgpu_pool_rangeproof_submit_ts(bpgpu_pool *p, size_t n, size_t m, size_t nbatch, const uint8_t *proofs, size_t proof_len, const uint8_t *commitments,
                                    const uint8_t *transcripts, size_t transcript_stride, const uint8_t *rng64, uint8_t *verdict, uint8_t *msm_out,
                                    uint8_t *transcripts_out, bpgpu_ticket **ticket) {
    if (!p || !ticket) return BPGPU_ERR_INVALID_ARG;
    *ticket = nullptr;
    mark_undecided(verdict, nbatch);
    call_guard cg(p);
    if (!cg.ok) return pfail(p, BPGPU_ERR_INVALID_ARG, "the pool is being destroyed");
    comb_req *r = new comb_req();
    r->async = true;
    r->nbatch = nbatch;
    r->out[RP_OUT_VERDICT] = verdict, r->out[RP_OUT_TS] = transcripts_out, r->out[RP_OUT_MSM] = msm_out;
    const int rc = rp_comb_run(p, r, n, m, proof_len, proofs, commitments, rng64, transcripts, transcript_stride);
    comb_finish_async(r, rc);
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

// ---- the boundary function itself through the queue: optional_multiscalar_mul, one MSM (or a few) per call, any thread ------------
// (src/range_proof/mod.rs:421-445, src/r1cs/verifier.rs:459-491: the mega-check shape -- generator scalars + per-MSM points)
static int msm_shared_run(bpgpu_pool *p, comb_req *r, size_t n, size_t m, size_t n_unique, const uint8_t *gen_scalars, const uint8_t *uniq_scalars,
                          const uint8_t *uniq_points) {
    if (!p || p->devs.empty()) return BPGPU_ERR_INVALID_ARG;
    if (r->nbatch == 0) return BPGPU_OK;
    if (!gen_scalars || !r->out[0] || !r->out[1] || (n_unique && (!uniq_scalars || !uniq_points))) return pfail(p, BPGPU_ERR_INVALID_ARG, "null argument");
    pool_dev *d0 = p->devs[p->rr_dev.fetch_add(1, std::memory_order_relaxed) % p->devs.size()];
    if (n == 0 || m == 0 || n > 65536 || m > 65536 || n * m > (1u << 20) || n_unique > (1u << 22) || r->nbatch > (1u << 24)) {
        // nothing a chain can be sized for: the ordinary entry point on the odd-jobs context reports what is wrong
        if (r->async) return pfail(p, BPGPU_ERR_INVALID_ARG, "shape not served by the combining queue");
        std::lock_guard<std::mutex> lk(d0->misc_mu);
        const int rc = bpgpu_msm_batch_shared(d0->misc, n, m, r->nbatch, n_unique, gen_scalars, uniq_scalars, uniq_points, r->out[0], r->out[1]);
        if (rc) mark_undecided(r->out[1], r->nbatch);   // (no status byte of a failed call reads 0, whatever the context wrote)
        return rc ? pfail(p, rc, "%s", bpgpu_last_error(d0->misc)) : BPGPU_OK;
    }
    comb_key base;
    base.kind = CQ_MSM_SHARED;
    base.a = (uint32_t)n, base.b = (uint32_t)m, base.c = (uint32_t)n_unique;
    const comb_regions g = regions_of(base);
    const uint8_t *src[CQ_MAX_IN] = {nullptr};
    src[0] = uniq_scalars, src[1] = uniq_points, src[2] = gen_scalars;
    size_t in_sz[CQ_MAX_IN] = {0};
    for (uint32_t i = 0; i < g.n_in; i++) in_sz[i] = g.in_sz[i];
    return comb_run_placed(p, r, base, src, in_sz, nullptr, d0, 64);
}

int bpgpu_pool_msm_batch_shared(bpgpu_pool *p, size_t n, size_t m, size_t nbatch, size_t n_unique, const uint8_t *gen_scalars, const uint8_t *uniq_scalars,
                                const uint8_t *uniq_points, uint8_t *out, uint8_t *status) {
    if (!p) return BPGPU_ERR_INVALID_ARG;
    mark_undecided(status, nbatch);
    call_guard cg(p);
    if (!cg.ok) return pfail(p, BPGPU_ERR_INVALID_ARG, "the pool is being destroyed");
    comb_req r;
    r.nbatch = nbatch;
    r.out[0] = out, r.out[1] = status;
    return msm_shared_run(p, &r, n, m, n_unique, gen_scalars, uniq_scalars, uniq_points);
}

int bpgpu_pool_msm_batch_shared_submit(bpgpu_pool *p, size_t n, size_t m, size_t nbatch, size_t n_unique, const uint8_t *gen_scalars, const uint8_t *uniq_scalars,
                                       const uint8_t *uniq_points, uint8_t *out, uint8_t *status, bpgpu_ticket **ticket) {
    if (!p || !ticket) return BPGPU_ERR_INVALID_ARG;
    *ticket = nullptr;
    mark_undecided(status, nbatch);
    call_guard cg(p);
    if (!cg.ok) return pfail(p, BPGPU_ERR_INVALID_ARG, "the pool is being destroyed");
    comb_req *r = new comb_req();
    r->async = true;
    r->nbatch = nbatch;
    r->out[0] = out, r->out[1] = status;
    const int rc = msm_shared_run(p, r, n, m, n_unique, gen_scalars, uniq_scalars, uniq_points);
    comb_finish_async(r, rc);
    if (rc) {
        const std::string keep = t_pool_err;
        ticket_block(r);
        delete r;
        t_pool_err = keep;
        return rc;
    }
    *ticket = (bpgpu_ticket *)r;
    return BPGPU_OK;
}

// vartime_multiscalar_mul / optional_multiscalar_mul with arbitrary points (ipp.rs:308-319, linear_proof.rs:217-225, messages.rs:128-149):
// a ragged batch; MSMs of equal length share chains (class = term count), so a request is placed stretch by stretch
int bpgpu_pool_msm_batch(bpgpu_pool *p, size_t nbatch, const uint32_t *n_terms, const uint8_t *scalars, const uint8_t *points, uint8_t *out, uint8_t *status) {
    if (!p || p->devs.empty()) return BPGPU_ERR_INVALID_ARG;
    if (nbatch == 0) return BPGPU_OK;
    mark_undecided(status, nbatch);
    call_guard cg(p);
    if (!cg.ok) return pfail(p, BPGPU_ERR_INVALID_ARG, "the pool is being destroyed");
    if (!n_terms || !out || !status) return pfail(p, BPGPU_ERR_INVALID_ARG, "null argument");
    uint64_t total = 0;
    for (size_t i = 0; i < nbatch; i++) total += n_terms[i];
    if (total && (!scalars || !points)) return pfail(p, BPGPU_ERR_INVALID_ARG, "null argument");
    if (nbatch > 0x7fffffffu / 64 || total > 0x7fffffffu / 64) return pfail(p, BPGPU_ERR_INVALID_ARG, "batch too large");
    pool_dev *d0 = p->devs[p->rr_dev.fetch_add(1, std::memory_order_relaxed) % p->devs.size()];
    comb_req r;
    r.nbatch = nbatch;
    r.out[0] = out, r.out[1] = status;
    d0->stat_requests.fetch_add(1, std::memory_order_relaxed);
    int rc = BPGPU_OK;
    size_t i = 0, term0 = 0;
    while (i < nbatch && !rc) {
        size_t j = i + 1;
        while (j < nbatch && n_terms[j] == n_terms[i]) j++;
        const size_t nt = n_terms[i];
        if (nt == 0) {   // the empty sum (written below, once every other stretch has come back without an error)
        } else if (nt > (1u << 22)) {
            rc = pfail(p, BPGPU_ERR_INVALID_ARG, "MSM %zu has too many terms for the combining queue", i);
        } else {
            comb_key base;
            base.kind = CQ_MSM;
            base.a = (uint32_t)nt;
            const uint8_t *src[CQ_MAX_IN] = {nullptr};
            src[0] = scalars + term0 * 32, src[1] = points + term0 * 32;
            rc = comb_place(p, d0, &r, base, src, nullptr, i, j);
        }
        term0 += nt * (j - i);
        i = j;
    }
    while (r.next_piece < r.pieces.size()) comb_collect_one(p, &r);
    if (!rc && r.rc) rc = pfail(p, r.rc, "%s", r.err.c_str());
    if (rc) {   // the header's promise: on a non-zero return no status byte of the call reads 0
        mark_undecided(status, nbatch);
        return rc;
    }
    for (size_t k = 0; k < nbatch; k++)
        if (n_terms[k] == 0) {   // the empty sum: the identity (encoding 0), status 0
            memset(out + k * 32, 0, 32);
            status[k] = 0;
        }
    return rc;
}

// InnerProductProof::verify, one proof (or a few) per call from any thread (src/inner_product_proof.rs:260-326): proofs of one size and
// one label share chains
int bpgpu_pool_ipp_verify(bpgpu_pool *p, size_t n, size_t nbatch, const uint8_t *proofs, size_t proof_len, const uint8_t *label, size_t label_len,
                          const uint8_t *G_factors, const uint8_t *H_factors, const uint8_t *P, const uint8_t *Q, const uint8_t *G, const uint8_t *H,
                          uint8_t *verdict, uint8_t *msm_out) {
    if (!p || p->devs.empty() || (label_len && !label) || label_len > 0xffffffffu) return BPGPU_ERR_INVALID_ARG;
    if (nbatch == 0) return BPGPU_OK;
    mark_undecided(verdict, nbatch);
    call_guard cg(p);
    if (!cg.ok) return pfail(p, BPGPU_ERR_INVALID_ARG, "the pool is being destroyed");
    if (!proofs || !verdict || !P || !Q || (n && (!G_factors || !H_factors || !G || !H))) return pfail(p, BPGPU_ERR_INVALID_ARG, "null argument");
    pool_dev *d0 = p->devs[p->rr_dev.fetch_add(1, std::memory_order_relaxed) % p->devs.size()];
    if (n == 0 || n > 65536 || proof_len == 0 || proof_len > 4096 || proof_len % 4 != 0 || nbatch > (1u << 24)) {   // (malformed lengths: reported per proof by the ordinary entry point)
        std::lock_guard<std::mutex> lk(d0->misc_mu);
        const int rc = bpgpu_ipp_verify_batch(d0->misc, n, nbatch, proofs, proof_len, label, label_len, G_factors, H_factors, P, Q, G, H, verdict, msm_out);
        if (rc) mark_undecided(verdict, nbatch);   // (whatever the context wrote before it failed: no verdict byte of a failed call reads 0)
        return rc ? pfail(p, rc, "%s", bpgpu_last_error(d0->misc)) : BPGPU_OK;
    }
    comb_req r;
    r.nbatch = nbatch;
    r.out[0] = verdict, r.out[1] = msm_out;
    comb_key base;
    base.kind = CQ_IPP;
    base.a = (uint32_t)n, base.b = (uint32_t)proof_len;
    uint8_t st0[BPGPU_TRANSCRIPT_BYTES];
    bpgpu_transcript_new(label, label_len, st0);
    memcpy(base.shared, st0, 203);
    const comb_regions g = regions_of(base);
    const uint8_t *src[CQ_MAX_IN] = {P, Q, proofs, G_factors, H_factors, G, H};
    size_t in_sz[CQ_MAX_IN] = {0};
    for (uint32_t i = 0; i < g.n_in; i++) in_sz[i] = g.in_sz[i];
    return comb_run_placed(p, &r, base, src, in_sz, nullptr, d0, 256);
}

// The queue's timeline (option "combine_trace" = ring size): one JSON object per line -- {"chain": ...} for every launch chain
// (open -> sealed -> issue begin / end -> completion seen -> delivery begin / end -> buffer free; nanoseconds of CLOCK_MONOTONIC),
// {"req": ...} for every eighth request per thread (submit -> slots reserved -> inputs written -> delivered -> woken).
int bpgpu_pool_trace_dump(bpgpu_pool *p, const char *path) {
    if (!p || !path) return BPGPU_ERR_INVALID_ARG;
    FILE *f = fopen(path, "w");
    if (!f) return pfail(p, BPGPU_ERR_INVALID_ARG, "cannot write %s", path);
    for (size_t di = 0; di < p->devs.size(); di++) {
        pool_dev *d = p->devs[di];
        std::lock_guard<std::mutex> g(d->trace_mu);
        const size_t nc = std::min(d->trace_chain_n, d->trace_chains.size()), nr = std::min(d->trace_req_n, d->trace_reqs.size());
        for (size_t i = 0; i < nc; i++) {
            const chain_ev &e = d->trace_chains[i];
            fprintf(f, "{\"chain\": {\"dev\": %zu, \"buf\": %u, \"epoch\": %u, \"kind\": %u, \"K\": %u, \"cap\": %u, \"n_sync\": %u, \"n_async\": %u, \"inflight_at_seal\": %u, "
                       "\"t_open\": %llu, \"t_seal\": %llu, \"t_issue0\": %llu, \"t_issue1\": %llu, \"t_done\": %llu, \"t_deliv0\": %llu, \"t_deliv1\": %llu, \"t_free\": %llu}}\n",
                    di, e.buf, e.epoch, e.kind, e.K, e.cap, e.n_sync, e.n_async, e.inflight, (unsigned long long)e.t_open, (unsigned long long)e.t_seal,
                    (unsigned long long)e.t_issue0, (unsigned long long)e.t_issue1, (unsigned long long)e.t_done, (unsigned long long)e.t_deliv0,
                    (unsigned long long)e.t_deliv1, (unsigned long long)e.t_free);
        }
        for (size_t i = 0; i < nr; i++) {
            const req_ev &e = d->trace_reqs[i];
            fprintf(f, "{\"req\": {\"dev\": %zu, \"buf\": %u, \"epoch\": %u, \"async\": %u, \"nbatch\": %u, \"t_submit\": %llu, \"t_reserved\": %llu, \"t_written\": %llu, "
                       "\"t_delivered\": %llu, \"t_woken\": %llu}}\n",
                    di, e.buf, e.epoch, e.async, e.nbatch, (unsigned long long)e.t_submit, (unsigned long long)e.t_reserved, (unsigned long long)e.t_written,
                    (unsigned long long)e.t_delivered, (unsigned long long)e.t_woken);
        }
    }
    fclose(f);
    return BPGPU_OK;
}

// Diagnostics (not part of the ABI; tools/combine_rate.cpp's watchdog prints it when a run does not end): where every staging buffer of the
// combining queue stands.  Reads the atomics only -- safe to call from any thread while the queue runs.
extern "C" int bpgpu_internal_pool_state(bpgpu_pool *p, char *out, size_t cap) {
    if (!p || !out || cap < 2) return BPGPU_ERR_INVALID_ARG;
    size_t o = 0;
    auto put = [&](const char *fmt, ...) {
        if (o + 1 >= cap) return;
        va_list ap;
        va_start(ap, fmt);
        const int n = vsnprintf(out + o, cap - o, fmt, ap);
        va_end(ap);
        if (n > 0) o += (size_t)n < cap - o ? (size_t)n : cap - o - 1;
    };
    put("pool: closing %d active_calls %lld\n", p->closing.load(), (long long)p->active_calls.load());
    static const char *names[] = {"FREE", "OPEN", "SEALED", "ISSUING", "ISSUED", "DONE"};
    for (size_t di = 0; di < p->devs.size(); di++) {
        pool_dev *d = p->devs[di];
        put(" dev %zu: kick %u svc_sleeping %u free_waiters %u chains %llu proofs %llu requests %llu polls %llu\n", di, d->kick.load(), d->svc_sleeping.load(),
            d->free_waiters.load(), (unsigned long long)d->stat_chains.load(), (unsigned long long)d->stat_proofs.load(), (unsigned long long)d->stat_requests.load(),
            (unsigned long long)d->stat_polls.load());
        for (comb_buf *b : d->cbufs) {
            const uint64_t s = b->state.load();
            const int st = b->st.load();
            put("  buf %2u %-7s epoch %u sealed %d reserved %u cap %u K %u written %u delivered %u phase %u n_sync %u n_async %u kind %u\n", b->index,
                st >= 0 && st <= CB_DONE ? names[st] : "?", cbs_epoch(s), cbs_sealed(s) ? 1 : 0, cbs_reserved(s), b->cap.load(), b->K, b->written.load(),
                b->delivered.load(), b->phase.load(), b->n_sync.load(), b->n_async.load(), b->key.kind);
        }
    }
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
    return (((comb_req *)ticket)->left.load(std::memory_order_acquire) & ~TKT_WAITING) == 0 ? 1 : 0;
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

// Work, not proofs: an aggregated proof walks (2nm + 2) generator terms through the window tables, a (64, 1) proof 130 -- a burst of
// 5120 proofs of (64, 16) is sixteen times the work of 5120 single proofs and must be cut into overlapping chains the way a burst of
// 81 000 single proofs would be (20 x 256 proofs of m = 16 went out as ONE chain alone on the device).  plan_flush
// therefore counts in (64,1)-proof equivalents; a chain's width in proofs of ITS shape follows from the equivalents it may carry.
static size_t work_equiv(size_t nbatch, size_t n, size_t m) { return (nbatch * (2 * n * m + 2) + 129) / 130; }
extern "C" uint64_t bpgpu_internal_work_equiv(uint64_t nbatch, uint64_t n, uint64_t m) { return work_equiv((size_t)nbatch, (size_t)n, (size_t)m); }
// (Round 6 re-checked the count: config 3's burst of 5120 proofs as 2 / 3 / 4 / 6 chains: 564 / 552 / 549 / 547 k/s, profiles/r06/cfg3_burst_chain_count_ab.txt.)
// The plan by proof count stands, except that ONE chain carrying at least two chains' worth of work (2 x coalesce_proofs
// equivalents) becomes two -- the second chain's launch 1 then overlaps the first one's table walk, as it does in every burst of single proofs
static flush_plan split_lone_heavy(flush_plan fp, size_t T_proofs, size_t T_work, size_t coalesce_proofs, size_t lanes, bool all_rlc, bool one_chain) {
    if (fp.chains != 1 || one_chain || all_rlc || lanes < 2 || T_proofs < 128 || T_work < 2 * coalesce_proofs) return fp;
    fp.per = (((T_proofs + 1) / 2) + 63) & ~(size_t)63;
    fp.chains = (T_proofs + fp.per - 1) / fp.per;
    return fp;
}
extern "C" void bpgpu_internal_split_lone_heavy(uint64_t chains, uint64_t per, uint64_t T_proofs, uint64_t T_work, uint64_t coalesce_proofs, uint64_t lanes, int all_rlc,
                                                int one_chain, uint64_t *chains_out, uint64_t *per_out) {
    flush_plan fp;
    fp.chains = (size_t)chains, fp.per = (size_t)per, fp.splits_hint = 0;
    fp = split_lone_heavy(fp, (size_t)T_proofs, (size_t)T_work, (size_t)coalesce_proofs, (size_t)lanes, all_rlc != 0, one_chain != 0);
    *chains_out = fp.chains, *per_out = fp.per;
}
// The order a flush packs its items in: grouped by what lets them share a chain (dev_item::same_shape), groups in order of first
// appearance, submission order inside a group -- alternating submissions of two shapes become two runs instead of one chain per item
// (VERDICT r04 #3).  key[i] = any number that is equal exactly for items that may share a chain.
static std::vector<size_t> flush_group_order(const std::vector<uint64_t> &key) {
    std::vector<size_t> order;
    std::vector<char> taken(key.size(), 0);
    for (size_t i = 0; i < key.size(); i++) {
        if (taken[i]) continue;
        for (size_t j = i; j < key.size(); j++)
            if (!taken[j] && key[j] == key[i]) {
                taken[j] = 1;
                order.push_back(j);
            }
    }
    return order;
}
extern "C" void bpgpu_internal_flush_group_order(const uint64_t *key, uint64_t count, uint64_t *order) {
    const std::vector<size_t> o = flush_group_order(std::vector<uint64_t>(key, key + count));
    for (size_t i = 0; i < o.size(); i++) order[i] = o[i];
}

static int flush_dev(bpgpu_pool *p, pool_dev *d, bool one_chain) {
    if (d->pending.empty()) return BPGPU_OK;
    (void)hipSetDevice(d->device);
    std::vector<dev_item> items;
    {
        // group the pending items by shape (stable): whatever order they were submitted in, items that can share a chain lie together
        std::vector<dev_item> pend;
        pend.swap(d->pending);
        std::vector<uint64_t> key(pend.size());
        for (size_t i = 0; i < pend.size(); i++) {
            size_t j = 0;
            while (j < i && !pend[j].same_shape(pend[i])) j++;
            key[i] = j;   // (index of the first item of the same shape)
        }
        items.reserve(pend.size());
        for (size_t i : flush_group_order(key)) items.push_back(std::move(pend[i]));
    }
    size_t T = d->pending_proofs, T_work = 0;
    d->pending_proofs = 0;
    for (const dev_item &it : items) T_work += work_equiv(it.nbatch, it.n ? it.n : 1, it.m ? it.m : 1);
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
    flush_plan fp = plan_flush(T, p->coalesce_proofs, p->pair_limit_proofs, p->max_chain_proofs, d->lanes.size(), all_rlc, one_chain);
    fp = split_lone_heavy(fp, T, T_work, p->coalesce_proofs, d->lanes.size(), all_rlc, one_chain);
    const uint32_t hint = fp.splits_hint;
    int rc_all = BPGPU_OK;
    size_t n_undecided = 0;
    std::string first_err;
    std::vector<rp_seg> segs;
    std::vector<const uint8_t *> labs;   // the label of every segment (all of one length within a chain)
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
        labs.clear();
        carried.clear();
        const size_t per = fp.per;   // proofs per chain.  (A small LEADING chain, whose table walk would start while the others are still in their early
                                     // phases, was measured on the 20-step bursts: worse at every size tried, profiles/r06/burst_small_leading_chain_ab.txt.)
        uint32_t filled = 0;
        bool any_msm = false, any_ticket = false;
        while (i < items.size() && filled < per && items[i].same_shape(head)) {
            const dev_item &it = items[i];
            size_t take = it.nbatch - off;
            // a batch-combined item stays whole (its 33-byte result is the result of ONE chain): it opens the next chain rather than being cut,
            // and may stretch a chain up to max_chain_proofs (submit refuses wider ones: ADVICE r04)
            if (head.rlc && off == 0 && filled > 0 && (take > per - filled || p->rlc_isolate)) break;
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
            // (a batch-combined item's `msm` is its ONE 33-byte result, not an array: never offset -- submit refuses items a chain cannot hold whole)
            sg.msm_out = it.msm ? (uint32_t *)(it.rlc ? it.msm : it.msm + off * 32) : nullptr;
            sg.init_w = nullptr;   // (start states per label are staged by the lane: bpgpu_internal_rp_verify_segs)
            sg.first = filled;
            sg.count = (uint32_t)take;
            any_msm = any_msm || it.msm;
            segs.push_back(sg);
            labs.push_back((const uint8_t *)it.label.data());
            filled += (uint32_t)take;
            off += take;
            if (off == it.nbatch) {
                i++;
                off = 0;
            }
        }
        const int rc = bpgpu_internal_rp_verify_segs(c, head.n, head.m, head.proof_len, labs.data(), head.label.size(), segs.data(),
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
    if (rlc && nbatch > p->max_chain_proofs)   // ONE combined identity check = ONE chain: a cut item would need one 33-byte result per piece
        return pfail(p, BPGPU_ERR_INVALID_ARG, "a batch-combined batch of %zu proofs does not fit one launch chain (max_chain_proofs = %zu): split it, or raise the option",
                     nbatch, p->max_chain_proofs);
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
    int rc = BPGPU_OK;
    if (d->pending.size() >= limit) rc = flush_dev(p, d, false);
    else if (p->auto_flush_proofs && d->pending_proofs >= p->auto_flush_proofs) rc = flush_dev(p, d, true);
    if (rc && ticket && *ticket) {
        // the flush this call triggered failed: an error AND a live ticket would leave the caller guessing who owns it.  Every pending item
        // (this one included) has been through the flush -- its chains are issued or its verdicts marked undecided --, so the ticket can go;
        // bpgpu_pool_wait orders the caller behind whatever did go out
        delete (dev_ticket *)*ticket;
        *ticket = nullptr;
    }
    return rc;
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

// The MSM call site for callers whose inputs are already in HBM (a service that keeps proofs on the device; bench.py's config-5 figure): the
// batch goes out at once as ONE launch chain on the next lane of the device -- the pool's lanes are what a caller would otherwise build by
// hand as (context, stream) pairs --, ordered behind the producer's stream if one is given; the ticket completes with that chain
// (bpgpu_pool_ticket_wait / _done / _stream_wait), bpgpu_pool_wait waits for everything.  Nothing is combined: device-resident batches
// are as wide as their owner made them.
int bpgpu_pool_msm_batch_shared_submit_dev(bpgpu_pool *p, int dev_index, size_t n, size_t m, size_t nbatch, size_t n_unique, const void *d_gen_scalars,
                                           const void *d_uniq_scalars, const void *d_uniq_points, void *d_out, void *d_status, void *producer_stream, int have_producer,
                                           bpgpu_ticket **ticket) {
    if (ticket) *ticket = nullptr;
    if (!p || dev_index < 0 || dev_index >= (int)p->devs.size()) return BPGPU_ERR_INVALID_ARG;
    if (nbatch && (!d_gen_scalars || !d_out || !d_status || (n_unique && (!d_uniq_scalars || !d_uniq_points)))) return pfail(p, BPGPU_ERR_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(p->mu);
    pool_dev *d = p->devs[dev_index];
    if (d->lanes.empty()) return pfail(p, BPGPU_ERR_INVALID_ARG, "the pool has no submit lanes");
    if (hipSetDevice(d->device) != hipSuccess) return pfail(p, BPGPU_ERR_HIP, "hipSetDevice failed");
    bpgpu_ctx *c = d->lanes[d->next_lane++ % d->lanes.size()];
    if (d->used_lanes < d->lanes.size() && d->next_lane > d->used_lanes) d->used_lanes = d->next_lane < d->lanes.size() ? d->next_lane : d->lanes.size();
    hipStream_t s = (hipStream_t)bpgpu_internal_stream(c);
    if (have_producer && nbatch) {   // the chain waits for what the producer has queued so far
        ev_holder ready;
        if (hipEventCreateWithFlags(&ready.ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ready.ev, (hipStream_t)producer_stream) != hipSuccess ||
            hipStreamWaitEvent(s, ready.ev, 0) != hipSuccess)
            return pfail(p, BPGPU_ERR_HIP, "ordering the chain behind the producer's stream failed");
    }
    int rc = BPGPU_OK;
    if (nbatch) {
        rc = bpgpu_msm_batch_shared_dev(c, n, m, nbatch, n_unique, d_gen_scalars, d_uniq_scalars, d_uniq_points, d_out, d_status, nullptr);
        if (rc) return pfail(p, rc, "%s", bpgpu_last_error(c));
    }
    if (ticket) {
        dev_ticket *t = new dev_ticket();
        t->d = d;
        t->unissued = 0;
        if (nbatch) {
            std::shared_ptr<ev_holder> done = record_done(c, true);
            if (!done) {
                delete t;
                return pfail(p, BPGPU_ERR_HIP, "recording the chain's completion failed");
            }
            t->done.push_back(done);
        }
        *ticket = (bpgpu_ticket *)t;
    }
    p->stat_chains++;
    p->stat_chain_proofs += nbatch;
    return BPGPU_OK;
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
