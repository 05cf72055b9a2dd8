// libbpgpu.so: host runtime + C ABI (include/bpgpu.h).  The kernels (gfx950) live in k_*.hip, thin __global__
// wrappers around the per-lane bodies in msm_vb.h / msm_fixed.h / rangeproof.h; the bodies are shared with the
// CPU test harness.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <sys/random.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/bpgpu.h"
#include "kernels.h"
#include "hostrng.h"

using namespace bp;

// ============================================================================
// host runtime
// ============================================================================
struct kstat {
    uint64_t launches = 0;
    double ms = 0;
};

// Generator tables are read-only and depend only on (device, generator encodings, W): contexts of
// one process share them (a service runs one context per host thread / stream).
struct shared_table {
    int device;
    uint32_t W;
    std::vector<uint8_t> gens;
    fb_entry *d_table;
    int refs;
};
static std::mutex g_tab_mu;                 // the list below
static std::mutex g_tab_build_mu[64];       // one table under construction per DEVICE (two contexts of a device asking for the same table: the second
                                            // one waits and finds it); tables of different devices are built at the same time (pool over N devices)
static std::vector<shared_table *> g_tables;

struct bpgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::string err;
    // bump arena for per-call scratch
    char *arena = nullptr;
    size_t arena_cap = 0, arena_off = 0;
    // options
    uint32_t W = 0;                          // fixed-base window bits; 0 = largest that fits table_budget
    uint64_t table_budget = 160ull << 30;    // bytes of HBM the generator tables may take: 160 GiB of the MI355X's 288 GB (W = 20 / 16 / 15 for m = 1 / 16 / 32:
                                             // +3 / +5 / +8 % over the 96 GiB of rounds 1-2, profiles/r03/window_sweep.txt); halved automatically when the allocation fails
    uint32_t splits = 0;
    uint32_t splits_hint = 0;                // per-call suggestion of the pool (pick_splits), used when `splits` is 0
    int busy_hint = -1;                      // set by the pool per chain: 1 = other chains run beside this one (throughput forms), 0 = it is alone (latency), -1 = unknown
    uint32_t vb_radix = 0;                   // radix of the proofs' own points in the range-proof path: 16, 32 (wide chains only), 0 = default (16)
    int a_outside = 1;                       // wide chains: A (coefficient 1) added after the Horner chain instead of carried through the window sums
    uint32_t horner_lanes = 0;               // lanes per Horner chain in the range-proof path: 1, 4, 64, 0 = auto (1 on wide chains, 64 up to 256 proofs, else 4)
    struct shared_table *tab_ref = nullptr;  // refcounted, shared by the contexts of one device
    // generators
    size_t gens_capacity = 0, party_capacity = 0;
    uint32_t *d_gens = nullptr;       // [B_blinding, B, G..., H...] compressed, n_gens x 8 words
    fb_entry *d_table = nullptr;
    fb_params prm{};
    std::vector<uint8_t> h_gens;      // host copy of the encodings
    // A second, larger-window table for a SMALLER shape (n2 <= gens_capacity, m2 <= party_capacity) that the context also serves
    // (bpgpu_gens_add_shape): its generators are a subset of the primary set's, re-listed compactly [B~, B, G(n2, m2), H(n2, m2)].
    // Range proofs with n <= n2 and m <= m2 walk this table; the window pair is chosen under ONE budget (pick_window_pair).
    struct sec_table {
        size_t n = 0, m = 0;
        std::vector<uint8_t> h_gens;
        uint32_t *d_gens = nullptr;
        fb_params prm{};
        fb_entry *d_table = nullptr;
        struct shared_table *ref = nullptr;
        std::map<std::pair<size_t, size_t>, uint32_t *> ids_cache;
    } sec;
    std::map<std::pair<size_t, size_t>, uint32_t *> gen_ids_cache;  // (n,m) -> device id list
    struct script_ent {
        uint32_t *mem = nullptr;
        size_t cap = 0;
        uint64_t last_use = 0;
    };
    std::map<std::vector<uint32_t>, script_ent> script_cache;       // (n, m, k, pos, pos_begin, flags, domsep) -> transcript script (rp_script.h), LRU-bounded (script_for)
    std::vector<uint32_t *> script_retired;
    uint64_t script_tick = 0;
    int coop_defer_emit = 1;                                        // narrow chains: the scalar role's U coefficient recodings on U lanes at once (rangeproof.h rp_defer): one blocking call
                                                                    // 0.528 / 0.528 / 0.533 -> 0.521 / 0.526 / 0.518 ms, 16 threads p50 0.64 -> 0.62 ms, 16 x 128 tickets 801 -> 819 k/s
                                                                    // (profiles/r05/coop_defer_emit_ab.txt; the kernel's duration for ONE proof is unchanged, 238 us: the gain is at several
                                                                    // groups per launch); 0: the leader recodes them one after the other
    int narrow_chunk = 0;                                           // narrow chains: per-proof points per (chunk, window) lane of launch 3; 0 = 8 (the Horner wavefront adds the chunks' rows), 32 = one chunk
    int exp_single = 1;                                             // narrow chains: the generator-exponent role with one index per lane (rp_expand_b1_thread); 0: as wide chains
    int narrow_walk = 1;                                            // narrow chains: the table walk with lane = split and the partial sums folded in launch 4 (k_rp34.hip rp_walk_narrow); 0: thread = proof
    int narrow_hi_max = 32;                                         // chains of up to this many proofs give every per-proof point a second table (2^128 P) and run a 32-window Horner chain; 0: never
    int narrow_fused_finish = 1;                                    // narrow chains, verdicts only: the last workgroup of a proof in launch 4 finishes it (no finish launch)
    int msm_narrow = 1;                                             // bpgpu_msm_batch with <= 16 MSMs of <= 768 terms in all: second tables, ~sqrt(N) chunks, one tail launch (k_vb_*_hi / k_vb_tail_narrow)
    int narrow_hi4_max = 4;                                         // chains of up to this many proofs: tables of the 2^64, 2^128 and 2^192 multiples, 16-window chain (0: never).  Same-box A/B
                                                                    // (profiles/r06/narrow_hi4_ab.txt): one call 0.315 -> 0.29 ms; at <= 8 the 16 / 64 / 256-thread rows are unchanged, at 32 they lose 5 - 15 %
    int exp_w3_min_nm = 1024;                                       // wide chains of shapes with at least this many generator pairs run the exponent launch at three wavefronts per SIMD (k_rp_exponents_w3)
    int coop_split = 1;                                             // narrow chains, per-proof check: the k + 1 inversions on k + 1 lanes of the group at once, the basepoint coefficients as a
                                                                    // role of launch 3 (rangeproof.h rp_split_invert_lane / rp_rows_thread); 0: the leader does it all (A/B: profiles/r06/coop_split_ab.txt)
    int transcript_coop = 1;                                        // chains of up to 256 proofs replay their transcripts 32 lanes per proof (keccak.h): one call of 1 / 8 / 64 / 256 proofs 0.62 -> 0.53 / 0.56 / 0.57 / 0.58 ms; 0: lane = proof everywhere
    int msm_fork = 1;                                               // bpgpu_msm_batch_shared: the generator-table half on the second stream beside the per-MSM points (0: one stream -- with
                                                                    // many contexts in flight two streams each outnumber the hardware queues: config 5 on 16 / 24 / 32 contexts +1.6 / +1.7 /
                                                                    // +3.2 % MSMs/s, but one MSM alone 0.77 -> 0.98 ms: stays on; profiles/r05/msm_queue_ab.txt)
    int split_stage3 = -1;                                          // window sums and generator exponents as two launches: 1 yes, 0 no, -1 auto (chains of >= 2048 proofs)
    bool no_script = false;                                         // option "transcript_script" = 0: byte-wise replay everywhere (A/B)
    // device-resident work decomposition of the uniform (nbatch, terms-per-MSM) variable-base plans
    // One decomposition per terms-per-MSM value, built for a CAPACITY of MSMs: entry b of every array depends on b only, so the plan
    // of `cap` MSMs serves any batch of up to `cap` as a prefix (the pool's combining queue issues chains of every width; keyed by
    // (nbatch, per) the cache missed on almost every chain -- a host-side build, a hipMalloc and three blocking copies each).
    struct plan_dev {
        char *mem = nullptr;
        size_t cap = 0, o1 = 0, o2 = 0;      // MSMs it was built for; offsets of chunk_first / term_chunk inside mem
        uint64_t last_use = 0;
    };
    struct plan_view {                       // what a call of `nbatch` MSMs sees of it
        char *mem = nullptr;
        size_t o1 = 0, o2 = 0, n_chunks = 0;
        uint32_t total = 0;
    };
    uint64_t plan_tick = 0;                  // plan_cache is bounded: least-recently-used entries are retired
    std::map<size_t, plan_dev> plan_cache;   // terms per MSM (| chunk size << 40) -> plan
    std::vector<char *> plan_retired;        // outgrown / evicted blocks that launches in flight may still read: freed with the context
    // per-proof status words of the range-proof path: zero between calls (the last kernel of a call resets the
    // entries it used), so no memset launch is needed per call; `dirty` forces one after an aborted enqueue
    uint32_t *rp_status = nullptr;
    size_t rp_status_cap = 0;
    uint32_t *fin_cnt = nullptr;             // [256] arrival counters of the narrow chain's fused finish (k_rp_stage4<64>: the last workgroup of a proof finishes it and
                                             // hands its counter back zeroed)
    bool rp_status_dirty = false;
    bool test_seed_set = false;   // bpgpu_internal_set_chain_seed (tests): the per-chain key of the device-expanded randomness
    uint8_t test_seed[32];
    // A context has ONE arena / status buffer / plan cache, so the work of consecutive calls must not overlap on the
    // device: every call records order_ev at its end, and a call that arrives on a different stream than its
    // predecessor makes its stream wait for that event first (ctx_enter / ctx_leave).
    hipEvent_t order_ev = nullptr;
    hipStream_t last_stream = nullptr;
    bool have_last = false;
    // pinned host staging (ragged work decompositions, library-drawn randomness, the host-pointer entry points' IO):
    // copies out of it are truly asynchronous; pin_ev guards its reuse by the next call
    char *pin = nullptr;
    size_t last_secret_bytes = 0;                      // leading bytes of the pinned block that held the last prover call's secrets
    size_t pin_cap = 0, pin_off = 0, pin_marked = 0;   // pin_marked: bytes already guarded by a recorded pin_ev (pin_mark)
    hipEvent_t pin_ev = nullptr;
    bool pin_pending = false;
    std::vector<char *> pin_retired;         // outgrown mid-call: still referenced by that call, freed at the next one
    // persistent device-side IO buffer of the host-pointer entry points (no hipMalloc / hipFree per call)
    char *io_dev = nullptr;
    size_t io_cap = 0;
    char *ipp_buf = nullptr;                 // term lists of the stand-alone inner-product verifier
    size_t ipp_cap = 0;
    char *rpp_buf = nullptr;                 // working set of the batched range-proof prover
    size_t rpp_cap = 0;
    uint32_t bucket_min = 0;                 // terms per MSM from which the bucket path is taken (0 = BK_MIN_TERMS; huge = never)
    int bucket_chain = 0;                    // option "bucket_chain": 0 = the fused chain (bucket2.h) where it applies, 1 = bucket.h's chain everywhere (A/B)
    int bucket_lanes = 0;                    // option "bucket_lanes": lanes of a (MSM, window) workgroup of the fused chain (0 = by batch width; 64, 128, 256)
    int exp_pairs = 1;                       // option "exponent_pairs": the generator-exponent role handles mirrored pairs of indices (0: four consecutive indices per lane, for A/B)
    int fast_tail = -1;                      // option "bucket_fast_tail" (A/B): -1 = by batch width, 0 / 1 = never / always the short-chain tail
    int walk_waves = 0;                      // option "fb_walk_waves": wavefronts the generator half of a fused chain is cut into (0 = 1024)
    // constant-time generator-table MSMs for the prover's secret-dependent commitments (msm_fixed.h fb_accum_ct_thread): their own
    // small-window table, built on first use
    bool prover_ct = false;
    fb_entry *d_table_ct = nullptr;
    fb_params prm_ct{};
    // second stream for the generator-table half of a shared-generator MSM: it is independent of the per-MSM points'
    // half until the finish, so the two halves run side by side (fork after the status memset, join before the finish)
    hipStream_t stream2 = nullptr;
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    // result of a submitted (not yet collected) host-pointer call: where its outputs sit in the pinned buffer and where the
    // caller wants them (bpgpu_rangeproof_verify_batch_submit / bpgpu_ctx_collect)
    struct pending_result {
        bool active = false;
        const char *h_out = nullptr;
        size_t nbatch = 0, off_msm = 0, off_ts = 0;
        uint8_t *verdict = nullptr, *msm_out = nullptr, *ts_out = nullptr;
    } pend;
    bool sync_blocking = false;              // host entry points: sleep on a blocking event instead of spinning
    hipEvent_t done_ev = nullptr;
    // profiling
    bool prof = false;
    std::map<std::string, kstat> stats;
    std::vector<std::tuple<std::string, hipEvent_t, hipEvent_t>> pending;
    std::vector<hipEvent_t> ev_pool;
};

static void release_ref(shared_table *&ref, fb_entry *&tab) {
    std::lock_guard<std::mutex> lk(g_tab_mu);
    if (ref && --ref->refs == 0) {
        hipFree(ref->d_table);
        for (size_t i = 0; i < g_tables.size(); i++)
            if (g_tables[i] == ref) {
                g_tables.erase(g_tables.begin() + i);
                break;
            }
        delete ref;
    }
    ref = nullptr;
    tab = nullptr;
}
static void release_table(bpgpu_ctx *c) { release_ref(c->tab_ref, c->d_table); }
static void release_secondary(bpgpu_ctx *c) {
    release_ref(c->sec.ref, c->sec.d_table);
    if (c->sec.d_gens) hipFree(c->sec.d_gens);
    c->sec.d_gens = nullptr;
    for (auto &kv : c->sec.ids_cache) hipFree(kv.second);
    c->sec.ids_cache.clear();
    c->sec.h_gens.clear();
    c->sec.n = c->sec.m = 0;
}

static int fail(bpgpu_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}
#define HIPCHK(c, call)                                                                                    \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess) return fail(c, BPGPU_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

static hipEvent_t get_event(bpgpu_ctx *c) {
    if (!c->ev_pool.empty()) {
        hipEvent_t e = c->ev_pool.back();
        c->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}

// Kernel launch; with profiling on, a start/stop event pair is attached to the dispatch itself
// (hipExtLaunchKernelGGL), so the elapsed time is the kernel's own begin..end -- the timestamps rocprofv3's
// kernel trace reports -- and excludes time spent queued behind other streams' kernels.
#define LAUNCH(c, s, name, kern, grid, block, ...)                                                          \
    do {                                                                                                    \
        if ((c)->prof) {                                                                                    \
            hipEvent_t a_ = get_event(c), b_ = get_event(c);                                                \
            hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, s, a_, b_, 0, __VA_ARGS__);             \
            (c)->pending.emplace_back(name, a_, b_);                                                        \
        } else {                                                                                            \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, s, __VA_ARGS__);                           \
        }                                                                                                   \
    } while (0)

// the same with `shm` bytes of dynamic LDS
#define LAUNCH_SHM(c, s, name, kern, grid, block, shm, ...)                                                 \
    do {                                                                                                    \
        if ((c)->prof) {                                                                                    \
            hipEvent_t a_ = get_event(c), b_ = get_event(c);                                                \
            hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), shm, s, a_, b_, 0, __VA_ARGS__);           \
            (c)->pending.emplace_back(name, a_, b_);                                                        \
        } else {                                                                                            \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(block), shm, s, __VA_ARGS__);                         \
        }                                                                                                   \
    } while (0)

static void drain_profile(bpgpu_ctx *c) {
    for (auto &t : c->pending) {
        hipEventSynchronize(std::get<2>(t));
        float ms = 0;
        hipEventElapsedTime(&ms, std::get<1>(t), std::get<2>(t));
        auto &st = c->stats[std::get<0>(t)];
        st.launches++;
        st.ms += ms;
        c->ev_pool.push_back(std::get<1>(t));
        c->ev_pool.push_back(std::get<2>(t));
    }
    c->pending.clear();
}

// arena: reset at the start of each API call; grows (with a device sync) when too small
static int arena_reserve(bpgpu_ctx *c, size_t bytes) {
    if (bytes <= c->arena_cap) return BPGPU_OK;
    HIPCHK(c, hipDeviceSynchronize());
    if (c->arena) HIPCHK(c, hipFree(c->arena));
    c->arena = nullptr;
    c->arena_cap = 0;
    size_t cap = bytes + bytes / 4 + (1 << 20);
    HIPCHK(c, hipMalloc((void **)&c->arena, cap));
    c->arena_cap = cap;
    return BPGPU_OK;
}
static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }
struct arena_plan {
    size_t total = 0;
    size_t add(size_t bytes) {
        size_t off = total;
        total += align_up(bytes);
        return off;
    }
};

static uint64_t table_bytes(uint32_t n_gens, uint32_t W);

static int collect_locked(bpgpu_ctx *c);
static int host_wait(bpgpu_ctx *c, hipStream_t s);
// ---- ordering of calls on one context (see bpgpu_ctx::order_ev) ----
// keep_staging: the call continues an earlier one of the same entry point and still holds pointers into the pinned staging
// buffers (the per-proof fallback of the batch-combined check): nothing retired is freed
static int ctx_enter(bpgpu_ctx *c, hipStream_t s, bool keep_staging = false) {
    if (c->pend.active) {   // a submitted call's results still sit in the staging buffers: deliver them first
        int rcp = collect_locked(c);
        if (rcp) return rcp;
    }
    c->pin_off = 0;
    c->pin_marked = 0;
    if (!c->pin_retired.empty() && !keep_staging) {
        if (c->pin_pending) HIPCHK(c, hipEventSynchronize(c->pin_ev));
        c->pin_pending = false;
        for (char *q : c->pin_retired) hipHostFree(q);
        c->pin_retired.clear();
    }
    if (c->have_last && c->last_stream != s) HIPCHK(c, hipStreamWaitEvent(s, c->order_ev, 0));
    return BPGPU_OK;
}
static int ctx_leave(bpgpu_ctx *c, hipStream_t s) {
    HIPCHK(c, hipEventRecord(c->order_ev, s));
    c->last_stream = s;
    c->have_last = true;
    if (c->pin_off > c->pin_marked) {   // staged bytes are in flight on s: the next call waits for them before overwriting
        HIPCHK(c, hipEventRecord(c->pin_ev, s));
        c->pin_pending = true;
    }
    return BPGPU_OK;
}
// bump allocation from the pinned staging buffer; the first allocation of a call waits until the previous
// call's copies out of the buffer are done
static int pin_alloc(bpgpu_ctx *c, hipStream_t /*s*/, size_t bytes, char **out) {
    if (c->pin_pending) {
        HIPCHK(c, hipEventSynchronize(c->pin_ev));
        c->pin_pending = false;
    }
    const size_t need = c->pin_off + align_up(bytes);
    if (need > c->pin_cap) {
        if (c->pin && c->pin_off) c->pin_retired.push_back(c->pin);   // earlier pieces of this call stay valid
        else if (c->pin) HIPCHK(c, hipHostFree(c->pin));
        c->pin = nullptr;
        c->pin_cap = 0;
        const size_t cap = need + need / 2 + (64 << 10);
        HIPCHK(c, hipHostMalloc((void **)&c->pin, cap, hipHostMallocDefault));
        c->pin_cap = cap;
        c->pin_off = 0;
        c->pin_marked = 0;
    }
    *out = c->pin + c->pin_off;
    c->pin_off += align_up(bytes);
    return BPGPU_OK;
}
// An H2D copy out of the staging buffer has just been enqueued on s and nothing else of this call will touch the bytes
// allocated so far: record the guard event NOW rather than at the end of the call, so that the next call on this context
// waits for the copy, not for the whole launch chain behind it (device-pointer calls that stage only their rng bytes or a
// segment table: with many lanes in flight the host would otherwise stall on every reuse of a lane)
static int pin_mark(bpgpu_ctx *c, hipStream_t s) {
    HIPCHK(c, hipEventRecord(c->pin_ev, s));
    c->pin_pending = true;
    c->pin_marked = c->pin_off;
    return BPGPU_OK;
}
static int io_reserve(bpgpu_ctx *c, size_t bytes) {
    if (bytes <= c->io_cap) return BPGPU_OK;
    HIPCHK(c, hipDeviceSynchronize());
    if (c->io_dev) HIPCHK(c, hipFree(c->io_dev));
    c->io_dev = nullptr;
    c->io_cap = 0;
    const size_t cap = bytes + bytes / 2 + (64 << 10);
    HIPCHK(c, hipMalloc((void **)&c->io_dev, cap));
    c->io_cap = cap;
    return BPGPU_OK;
}
// wait for everything enqueued on s (host-pointer entry points): spin, or sleep on a blocking event
static int host_wait(bpgpu_ctx *c, hipStream_t s) {
    if (c->sync_blocking) {
        HIPCHK(c, hipEventRecord(c->done_ev, s));
        HIPCHK(c, hipEventSynchronize(c->done_ev));
    } else {
        HIPCHK(c, hipStreamSynchronize(s));
    }
    c->pin_pending = false;
    c->pin_off = 0;
    return BPGPU_OK;
}
static int os_random(bpgpu_ctx *c, char *dst, size_t bytes);
// finish a submitted call: wait for its stream work and hand the results to the caller's buffers
static int collect_locked(bpgpu_ctx *c) {
    if (!c->pend.active) return BPGPU_OK;
    bpgpu_ctx::pending_result p = c->pend;
    c->pend.active = false;
    int rc = host_wait(c, c->stream);
    if (rc) return rc;
    memcpy(p.verdict, p.h_out, p.nbatch);
    if (p.msm_out) memcpy(p.msm_out, p.h_out + p.off_msm, p.nbatch * 32);
    if (p.ts_out) memcpy(p.ts_out, p.h_out + p.off_ts, p.nbatch * BPGPU_TRANSCRIPT_BYTES);
    return BPGPU_OK;
}

extern "C" {

int bpgpu_version(void) { return 100; }

int bpgpu_ctx_create(int device, bpgpu_ctx **out) {
    if (!out) return BPGPU_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return BPGPU_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return BPGPU_ERR_NO_DEVICE;
    bpgpu_ctx *c = new bpgpu_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->order_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->pin_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done_ev, hipEventDisableTiming | hipEventBlockingSync) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->join_ev, hipEventDisableTiming) != hipSuccess) {
        delete c;
        return BPGPU_ERR_HIP;
    }
    if (const char *e = getenv("BPGPU_COOP_SPLIT")) c->coop_split = atoi(e) != 0;
    if (const char *e = getenv("BPGPU_EXP_W3_MIN_NM")) c->exp_w3_min_nm = atoi(e);   // (A/B only)
    if (const char *e = getenv("BPGPU_NARROW_WALK")) c->narrow_walk = atoi(e) != 0;
    if (const char *e = getenv("BPGPU_NARROW_FUSED_FINISH")) c->narrow_fused_finish = atoi(e) != 0;
    if (const char *e = getenv("BPGPU_NARROW_HI_MAX")) c->narrow_hi_max = atoi(e);
    if (const char *e = getenv("BPGPU_NARROW_HI4_MAX")) c->narrow_hi4_max = atoi(e);
    if (const char *e = getenv("BPGPU_EXP_SINGLE")) c->exp_single = atoi(e) != 0;
    if (const char *e = getenv("BPGPU_NARROW_CHUNK")) c->narrow_chunk = atoi(e);
    if (const char *e = getenv("BPGPU_COOP_DEFER_EMIT")) c->coop_defer_emit = atoi(e) != 0;   // (A/B of whole test suites: the option's default for every context of the process)
    *out = c;
    return BPGPU_OK;
}

void bpgpu_ctx_destroy(bpgpu_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    c->pend.active = false;   // results of a submitted call nobody collected are dropped
    drain_profile(c);
    for (auto e : c->ev_pool) hipEventDestroy(e);
    for (auto &kv : c->gen_ids_cache) hipFree(kv.second);
    for (auto &kv : c->script_cache) hipFree(kv.second.mem);
    for (uint32_t *q : c->script_retired) hipFree(q);
    for (auto &kv : c->plan_cache) hipFree(kv.second.mem);
    for (char *q : c->plan_retired) hipFree(q);
    if (c->arena) hipFree(c->arena);
    if (c->io_dev) hipFree(c->io_dev);
    if (c->ipp_buf) hipFree(c->ipp_buf);
    if (c->rpp_buf) hipFree(c->rpp_buf);
    if (c->pin) hipHostFree(c->pin);
    for (char *q : c->pin_retired) hipHostFree(q);
    if (c->order_ev) hipEventDestroy(c->order_ev);
    if (c->pin_ev) hipEventDestroy(c->pin_ev);
    if (c->done_ev) hipEventDestroy(c->done_ev);
    if (c->fork_ev) hipEventDestroy(c->fork_ev);
    if (c->join_ev) hipEventDestroy(c->join_ev);
    if (c->stream2) hipStreamDestroy(c->stream2);
    if (c->rp_status) hipFree(c->rp_status);
    if (c->fin_cnt) hipFree(c->fin_cnt);
    if (c->d_table_ct) hipFree(c->d_table_ct);
    if (c->d_gens) hipFree(c->d_gens);
    release_secondary(c);
    release_table(c);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

const char *bpgpu_last_error(bpgpu_ctx *c) { return c ? c->err.c_str() : "null context"; }

int bpgpu_ctx_set_option(bpgpu_ctx *c, const char *key, int64_t value) {
    if (!c || !key) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!strcmp(key, "fixed_window_bits")) {
        if (value != 0 && (value < 2 || value > 20)) return fail(c, BPGPU_ERR_INVALID_ARG, "fixed_window_bits must be 0 (auto) or 2..20");
        if (c->d_table) return fail(c, BPGPU_ERR_INVALID_ARG, "set fixed_window_bits before loading generators");
        c->W = (uint32_t)value;
        return BPGPU_OK;
    }
    if (!strcmp(key, "fixed_table_max_bytes")) {
        if (value < (1 << 20)) return fail(c, BPGPU_ERR_INVALID_ARG, "fixed_table_max_bytes too small");
        if (c->d_table) return fail(c, BPGPU_ERR_INVALID_ARG, "set fixed_table_max_bytes before loading generators");
        c->table_budget = (uint64_t)value;
        return BPGPU_OK;
    }
    if (!strcmp(key, "fixed_splits")) {
        if (value < 0 || value > 4096) return fail(c, BPGPU_ERR_INVALID_ARG, "fixed_splits out of range");
        c->splits = (uint32_t)value;
        return BPGPU_OK;
    }
    if (!strcmp(key, "per_proof_radix")) {
        if (value != 0 && value != 16 && value != 32) return fail(c, BPGPU_ERR_INVALID_ARG, "per_proof_radix must be 0 (auto), 16 or 32");
        c->vb_radix = (uint32_t)value;
        return BPGPU_OK;
    }
    if (!strcmp(key, "a_outside")) {
        c->a_outside = value != 0;
        return BPGPU_OK;
    }
    if (!strcmp(key, "horner_lanes")) {
        if (value != 0 && value != 1 && value != 4 && value != 64) return fail(c, BPGPU_ERR_INVALID_ARG, "horner_lanes must be 0 (auto), 1, 4 or 64");
        c->horner_lanes = (uint32_t)value;
        return BPGPU_OK;
    }
    if (!strcmp(key, "host_sync_blocking")) {
        c->sync_blocking = value != 0;
        return BPGPU_OK;
    }
    if (!strcmp(key, "transcript_script")) {
        c->no_script = value == 0;
        return BPGPU_OK;
    }
    if (!strcmp(key, "exp_single")) {
        c->exp_single = value != 0;
        return BPGPU_OK;
    }
    if (!strcmp(key, "narrow_hi4_max")) {
        if (value < 0 || value > 256) return fail(c, BPGPU_ERR_INVALID_ARG, "narrow_hi4_max must be 0 .. 256");
        c->narrow_hi4_max = (int)value;
        return BPGPU_OK;
    }
    if (!strcmp(key, "narrow_hi_max")) {
        if (value < 0 || value > 256) return fail(c, BPGPU_ERR_INVALID_ARG, "narrow_hi_max must be 0 .. 256");
        c->narrow_hi_max = (int)value;
        return BPGPU_OK;
    }
    if (!strcmp(key, "narrow_fused_finish")) {
        c->narrow_fused_finish = value != 0;
        return BPGPU_OK;
    }
    if (!strcmp(key, "msm_narrow")) {
        c->msm_narrow = value != 0;
        return BPGPU_OK;
    }
    if (!strcmp(key, "narrow_walk")) {
        c->narrow_walk = value != 0;
        return BPGPU_OK;
    }
    if (!strcmp(key, "coop_split")) {
        c->coop_split = value != 0;
        return BPGPU_OK;
    }
    if (!strcmp(key, "narrow_chunk")) {
        if (value < 0 || value > BP_VB_CHUNK) return fail(c, BPGPU_ERR_INVALID_ARG, "narrow_chunk must be 0 (automatic) or 2 .. 32");
        c->narrow_chunk = (int)value;
        return BPGPU_OK;
    }
    if (!strcmp(key, "coop_defer_emit")) {
        c->coop_defer_emit = value != 0;
        return BPGPU_OK;
    }
    if (!strcmp(key, "transcript_coop")) {
        if (value < 0 || value > 1) return fail(c, BPGPU_ERR_INVALID_ARG, "transcript_coop must be 0 or 1");
        c->transcript_coop = (int)value;
        return BPGPU_OK;
    }
    if (!strcmp(key, "msm_fork")) {
        c->msm_fork = value != 0;
        return BPGPU_OK;
    }
    if (!strcmp(key, "split_stage3")) {
        c->split_stage3 = value < 0 ? -1 : (value != 0);
        return BPGPU_OK;
    }
    if (!strcmp(key, "prover_constant_time")) {
        c->prover_ct = value != 0;
        return BPGPU_OK;
    }
    if (!strcmp(key, "bucket_min_terms")) {
        if (value < 0 || value > 0x7fffffff) return fail(c, BPGPU_ERR_INVALID_ARG, "bucket_min_terms out of range");
        c->bucket_min = (uint32_t)value;
        return BPGPU_OK;
    }
    if (!strcmp(key, "bucket_chain")) {
        if (value < 0 || value > 1) return fail(c, BPGPU_ERR_INVALID_ARG, "bucket_chain must be 0 (fused chain where it applies) or 1 (bucket.h's chain)");
        c->bucket_chain = (int)value;
        return BPGPU_OK;
    }
    if (!strcmp(key, "exponent_pairs")) {
        c->exp_pairs = value != 0;
        return BPGPU_OK;
    }
    if (!strcmp(key, "bucket_fast_tail")) {
        c->fast_tail = value < 0 ? -1 : (value != 0);
        return BPGPU_OK;
    }
    if (!strcmp(key, "fb_walk_waves")) {
        if (value < 0 || value > 65536) return fail(c, BPGPU_ERR_INVALID_ARG, "fb_walk_waves out of range");
        c->walk_waves = (int)value;
        return BPGPU_OK;
    }
    if (!strcmp(key, "bucket_lanes")) {
        if (value != 0 && value != 64 && value != 128 && value != 256) return fail(c, BPGPU_ERR_INVALID_ARG, "bucket_lanes must be 0 (auto), 64, 128 or 256");
        c->bucket_lanes = (int)value;
        return BPGPU_OK;
    }
    return fail(c, BPGPU_ERR_INVALID_ARG, "unknown option %s", key);
}

int bpgpu_ctx_get_option(bpgpu_ctx *c, const char *key, int64_t *value) {
    if (!c || !key || !value) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!strcmp(key, "fixed_window_bits")) *value = c->d_table ? c->prm.W : c->W;          // effective once tables exist
    else if (!strcmp(key, "fixed_table_bytes")) *value = c->d_table ? (int64_t)table_bytes(c->prm.n_gens, c->prm.W) : 0;
    else if (!strcmp(key, "fixed_table_max_bytes")) *value = (int64_t)c->table_budget;
    else if (!strcmp(key, "secondary_window_bits")) *value = c->sec.d_table ? c->sec.prm.W : 0;
    else if (!strcmp(key, "secondary_table_bytes")) *value = c->sec.d_table ? (int64_t)table_bytes(c->sec.prm.n_gens, c->sec.prm.W) : 0;
    else if (!strcmp(key, "secondary_shape_n")) *value = (int64_t)c->sec.n;
    else if (!strcmp(key, "secondary_shape_m")) *value = (int64_t)c->sec.m;
    else if (!strcmp(key, "fixed_splits")) *value = c->splits;
    else if (!strcmp(key, "horner_lanes")) *value = c->horner_lanes;
    else if (!strcmp(key, "per_proof_radix")) *value = c->vb_radix;
    else if (!strcmp(key, "a_outside")) *value = c->a_outside;
    else if (!strcmp(key, "host_sync_blocking")) *value = c->sync_blocking ? 1 : 0;
    else if (!strcmp(key, "transcript_script")) *value = c->no_script ? 0 : 1;
    else if (!strcmp(key, "prover_constant_time")) *value = c->prover_ct ? 1 : 0;
    else if (!strcmp(key, "bucket_min_terms")) *value = c->bucket_min ? c->bucket_min : BK_MIN_TERMS;
    else if (!strcmp(key, "bucket_chain")) *value = c->bucket_chain;
    else if (!strcmp(key, "bucket_lanes")) *value = c->bucket_lanes;
    else if (!strcmp(key, "bucket_fast_tail")) *value = c->fast_tail;
    else if (!strcmp(key, "exponent_pairs")) *value = c->exp_pairs;
    else if (!strcmp(key, "fb_walk_waves")) *value = c->walk_waves ? c->walk_waves : 2048;
    else if (!strcmp(key, "staging_residue")) {
        // test hook: non-zero bytes left in the persistent staging buffers (pinned block, device IO buffer, prover working sets,
        // arena) -- 0 after a prover entry point has returned (prover_exit)
        if (hipSetDevice(c->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(c, BPGPU_ERR_HIP, "synchronize failed");
        int64_t nz = 0;
        for (size_t i = 0; i < c->pin_cap && i < c->last_secret_bytes; i++) nz += c->pin[i] != 0;   // (behind the secrets: public outputs)
        std::vector<char> tmp;
        const std::pair<const char *, size_t> bufs[4] = {{c->io_dev, c->io_cap}, {c->rpp_buf, c->rpp_cap}, {c->ipp_buf, c->ipp_cap}, {c->arena, c->arena_cap}};
        for (const auto &b : bufs) {
            if (!b.first || !b.second) continue;
            tmp.resize(b.second);
            if (hipMemcpy(tmp.data(), b.first, b.second, hipMemcpyDeviceToHost) != hipSuccess) return fail(c, BPGPU_ERR_HIP, "read-back failed");
            for (char ch : tmp) nz += ch != 0;
        }
        *value = nz;
    }
    else return fail(c, BPGPU_ERR_INVALID_ARG, "unknown option %s", key);
    return BPGPU_OK;
}

int bpgpu_synchronize(bpgpu_ctx *c) {
    if (!c) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BPGPU_OK;
}

int bpgpu_profile_enable(bpgpu_ctx *c, int on) {
    if (!c) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    c->prof = on != 0;
    return BPGPU_OK;
}
int bpgpu_profile_reset(bpgpu_ctx *c) {
    if (!c) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    drain_profile(c);
    c->stats.clear();
    return BPGPU_OK;
}
int bpgpu_profile_report(bpgpu_ctx *c, char *buf, size_t cap) {
    if (!c || !buf || cap == 0) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    drain_profile(c);
    std::string s;
    for (auto &kv : c->stats) {
        char line[256];
        snprintf(line, sizeof line, "%s %llu %.6f\n", kv.first.c_str(), (unsigned long long)kv.second.launches, kv.second.ms);
        s += line;
    }
    snprintf(buf, cap, "%s", s.c_str());
    return BPGPU_OK;
}

}  // extern "C"

// ============================================================================
// generators
// ============================================================================
static uint64_t table_bytes(uint32_t n_gens, uint32_t W) { return (uint64_t)n_gens * fb_nwin(W) * (1ull << (W - 1)) * sizeof(fb_entry); }

// One window table for the generator list (h_gens on the host, d_gens on the device: n_gens compressed points): found among the tables
// the process already holds on this device, or built.  W_fixed = 0: the fewest windows (= additions per generator term) whose table
// fits `budget`; ties: the smaller table.
static int build_table_set(bpgpu_ctx *c, const std::vector<uint8_t> &h_gens, const uint32_t *d_gens, uint32_t W_fixed, uint64_t budget0, fb_params *prm_out,
                           shared_table **ref_out, fb_entry **tab_out) {
    fb_params prm;
    prm.n_gens = (uint32_t)(h_gens.size() / 32);
    std::lock_guard<std::mutex> lk_build(g_tab_build_mu[(unsigned)c->device & 63u]);
    uint32_t W = W_fixed;
    fb_entry *d_table = nullptr;
    size_t entries = 0;
    for (uint64_t budget = budget0;; budget /= 2) {
        if (W_fixed == 0) {
            W = 4;
            for (uint32_t w = 5; w <= 20; w++)
                if (table_bytes(prm.n_gens, w) <= budget && fb_nwin(w) < fb_nwin(W)) W = w;
        }
        prm.W = W;
        prm.nwin = fb_nwin(W);
        prm.half = 1u << (W - 1);
        *prm_out = prm;
        {
            std::lock_guard<std::mutex> lk(g_tab_mu);
            for (shared_table *t : g_tables)
                if (t->device == c->device && t->W == W && t->gens == h_gens) {
                    t->refs++;
                    *ref_out = t;
                    *tab_out = t->d_table;
                    return BPGPU_OK;
                }
        }
        entries = (size_t)prm.n_gens * prm.nwin * prm.half;
        if (hipMalloc((void **)&d_table, entries * sizeof(fb_entry)) == hipSuccess) break;
        (void)hipGetLastError();
        d_table = nullptr;
        // automatic window: the HBM may be shared with other tenants -- settle for a smaller table
        if (W_fixed != 0 || W <= 8) return fail(c, BPGPU_ERR_HIP, "hipMalloc of %zu table bytes (W = %u) failed", entries * sizeof(fb_entry), W);
    }
    ge_ext *d_base = nullptr;
    uint32_t *d_bad = nullptr;
    if (hipMalloc((void **)&d_base, (size_t)prm.n_gens * prm.nwin * sizeof(ge_ext)) != hipSuccess ||
        hipMalloc((void **)&d_bad, 4) != hipSuccess) {
        hipFree(d_table);
        return fail(c, BPGPU_ERR_HIP, "hipMalloc of table scratch failed");
    }
    hipMemsetAsync(d_bad, 0, 4, c->stream);
    LAUNCH(c, c->stream, "fb_base", k_fb_base, (prm.n_gens + 63) / 64, 64, prm, d_gens, d_base, d_bad);
    LAUNCH(c, c->stream, "fb_fill", k_fb_fill, (prm.n_gens * prm.nwin + 63) / 64, 64, prm, d_base, d_table);
    const uint64_t n_groups = (entries + BP_FB_NORM_GROUP - 1) / BP_FB_NORM_GROUP;
    LAUNCH(c, c->stream, "fb_norm", k_fb_norm, (uint32_t)((n_groups + 63) / 64), 64, n_groups, (uint64_t)entries, d_table);
    uint32_t bad = 0;
    hipError_t e1 = hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c->stream);
    hipError_t e2 = hipStreamSynchronize(c->stream);
    hipError_t e3 = hipGetLastError();
    hipFree(d_base);
    hipFree(d_bad);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
        hipFree(d_table);
        return fail(c, BPGPU_ERR_HIP, "table construction failed: %s", hipGetErrorString(e2 != hipSuccess ? e2 : (e1 != hipSuccess ? e1 : e3)));
    }
    if (bad) {
        hipFree(d_table);
        return fail(c, BPGPU_ERR_BAD_GENERATOR, "a generator encoding does not decode");
    }
    shared_table *t = new shared_table{c->device, W, h_gens, d_table, 1};
    {
        std::lock_guard<std::mutex> lk(g_tab_mu);
        g_tables.push_back(t);
    }
    *ref_out = t;
    *tab_out = d_table;
    return BPGPU_OK;
}

static int build_tables(bpgpu_ctx *c, uint32_t W_override = 0) {
    // c->d_gens / c->h_gens hold n_gens compressed points
    release_secondary(c);   // (belongs to the previous generator set)
    release_table(c);
    if (c->d_table_ct) {   // belongs to the previous generator set
        hipFree(c->d_table_ct);
        c->d_table_ct = nullptr;
    }
    for (auto &kv : c->gen_ids_cache) hipFree(kv.second);
    c->gen_ids_cache.clear();
    return build_table_set(c, c->h_gens, c->d_gens, W_override ? W_override : c->W, c->table_budget, &c->prm, &c->tab_ref, &c->d_table);
}

// Windows for TWO shapes under one budget (bpgpu_gens_add_shape): the pair (W1 for the primary set of n1 generators, W2 for the secondary
// set of n2) that minimises the sum of the two walks' lengths RELATIVE to what each would get alone with the whole budget --
// nwin(W1) / nwin(W1*) + nwin(W2) / nwin(W2*) -- among the pairs whose tables fit together; ties: fewer bytes.  W1_fixed != 0 pins W1.
static void pick_window_pair(uint32_t n1, uint32_t n2, uint64_t budget, uint32_t W1_fixed, uint32_t *W1, uint32_t *W2) {
    auto best_alone = [&](uint32_t ng) {
        uint32_t W = 4;
        for (uint32_t w = 5; w <= 20; w++)
            if (table_bytes(ng, w) <= budget && fb_nwin(w) < fb_nwin(W)) W = w;
        return W;
    };
    const double a1 = fb_nwin(best_alone(n1)), a2 = fb_nwin(best_alone(n2));
    double best = 1e30;
    uint64_t best_bytes = ~0ull;
    *W1 = W1_fixed ? W1_fixed : 4;
    *W2 = 4;
    for (uint32_t w1 = W1_fixed ? W1_fixed : 4; w1 <= (W1_fixed ? W1_fixed : 20); w1++)
        for (uint32_t w2 = 4; w2 <= 20; w2++) {
            const uint64_t bytes = table_bytes(n1, w1) + table_bytes(n2, w2);
            if (bytes > budget && !(w1 == (W1_fixed ? W1_fixed : 4) && w2 == 4)) continue;
            const double f = fb_nwin(w1) / a1 + fb_nwin(w2) / a2;
            if (f < best - 1e-12 || (f < best + 1e-12 && bytes < best_bytes)) {
                best = f;
                best_bytes = bytes;
                *W1 = w1;
                *W2 = w2;
            }
        }
}
// (tests/test_abi_and_host.py: the choice is plain host logic)
extern "C" void bpgpu_internal_window_pair(uint32_t n_gens_primary, uint32_t n_gens_secondary, uint64_t budget, uint32_t W1_fixed, uint32_t *W1, uint32_t *W2) {
    pick_window_pair(n_gens_primary, n_gens_secondary, budget, W1_fixed, W1, W2);
}

// BulletproofGens::new(gens_capacity, party_capacity) serves every (n <= gens_capacity, m <= party_capacity) (generators.rs:157-259); the
// window table that replaces the doublings is sized for the whole set, so a service that verifies m = 16 AND m = 1 proofs walked the
// m = 1 ones through 16-bit windows (-9 %).  This adds a second table over the sub-set (n2, m2) and re-balances both windows under the
// context's one budget (fixed_table_max_bytes).  Called again, it replaces the secondary shape.
extern "C" int bpgpu_gens_add_shape(bpgpu_ctx *c, size_t n2, size_t m2) {
    if (!c || n2 == 0 || m2 == 0) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (c->h_gens.empty()) return fail(c, BPGPU_ERR_NO_GENS, "generators not loaded");
    if (n2 > c->gens_capacity || m2 > c->party_capacity) return fail(c, BPGPU_ERR_NO_GENS, "the secondary shape exceeds the generator set");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream2));
    const uint32_t n1 = (uint32_t)(2 + 2 * c->gens_capacity * c->party_capacity), ng2 = (uint32_t)(2 + 2 * n2 * m2);
    uint32_t W1 = 0, W2 = 0;
    pick_window_pair(n1, ng2, c->table_budget, c->W, &W1, &W2);
    release_secondary(c);
    if (!c->d_table || c->prm.W != W1) {   // the primary table makes room (or was dropped by the pool before the re-balancing)
        const int rc = build_tables(c, W1);
        if (rc) return rc;
    }
    const size_t tot = c->gens_capacity * c->party_capacity;
    std::vector<uint8_t> &h2 = c->sec.h_gens;
    h2.assign((size_t)ng2 * 32, 0);
    memcpy(&h2[0], &c->h_gens[0], 64);   // B~, B
    for (size_t j = 0; j < m2; j++) {
        memcpy(&h2[64 + j * n2 * 32], &c->h_gens[64 + j * c->gens_capacity * 32], n2 * 32);
        memcpy(&h2[64 + (m2 * n2 + j * n2) * 32], &c->h_gens[64 + (tot + j * c->gens_capacity) * 32], n2 * 32);
    }
    HIPCHK(c, hipMalloc((void **)&c->sec.d_gens, h2.size()));
    HIPCHK(c, hipMemcpy(c->sec.d_gens, h2.data(), h2.size(), hipMemcpyHostToDevice));
    const int rc = build_table_set(c, h2, c->sec.d_gens, W2, c->table_budget, &c->sec.prm, &c->sec.ref, &c->sec.d_table);
    if (rc) {
        release_secondary(c);
        return rc;
    }
    c->sec.n = n2;
    c->sec.m = m2;
    return BPGPU_OK;
}
// the pool re-balances a whole device: every lane lets go of its tables first, so that the old and the new pair never coexist in HBM
extern "C" int bpgpu_internal_release_tables(bpgpu_ctx *c) {
    if (!c) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream2));
    release_secondary(c);
    release_table(c);
    return BPGPU_OK;
}

// the small-window table of the constant-time path (W = 4: 64 KiB per generator), built on first use
static int build_ct_table(bpgpu_ctx *c) {
    if (c->d_table_ct) return BPGPU_OK;
    fb_params prm;
    prm.n_gens = c->prm.n_gens;
    prm.W = BP_FB_CT_W;
    prm.nwin = fb_nwin(prm.W);
    prm.half = 1u << (prm.W - 1);
    const size_t entries = (size_t)prm.n_gens * prm.nwin * prm.half;
    fb_entry *d_table = nullptr;
    ge_ext *d_base = nullptr;
    uint32_t *d_bad = nullptr;
    if (hipMalloc((void **)&d_table, entries * sizeof(fb_entry)) != hipSuccess || hipMalloc((void **)&d_base, (size_t)prm.n_gens * prm.nwin * sizeof(ge_ext)) != hipSuccess ||
        hipMalloc((void **)&d_bad, 4) != hipSuccess) {
        if (d_table) hipFree(d_table);
        if (d_base) hipFree(d_base);
        return fail(c, BPGPU_ERR_HIP, "hipMalloc of the constant-time table failed");
    }
    hipMemsetAsync(d_bad, 0, 4, c->stream);
    LAUNCH(c, c->stream, "fb_base", k_fb_base, (prm.n_gens + 63) / 64, 64, prm, c->d_gens, d_base, d_bad);
    LAUNCH(c, c->stream, "fb_fill", k_fb_fill, (prm.n_gens * prm.nwin + 63) / 64, 64, prm, d_base, d_table);
    const uint64_t n_groups = (entries + BP_FB_NORM_GROUP - 1) / BP_FB_NORM_GROUP;
    LAUNCH(c, c->stream, "fb_norm", k_fb_norm, (uint32_t)((n_groups + 63) / 64), 64, n_groups, (uint64_t)entries, d_table);
    const hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(d_base);
    hipFree(d_bad);
    if (e != hipSuccess) {
        hipFree(d_table);
        return fail(c, BPGPU_ERR_HIP, "constant-time table construction failed: %s", hipGetErrorString(e));
    }
    c->d_table_ct = d_table;
    c->prm_ct = prm;
    return BPGPU_OK;
}

static int load_gens_locked(bpgpu_ctx *c, size_t gens_capacity, size_t party_capacity, const uint8_t *G, const uint8_t *H,
                            const uint8_t *B, const uint8_t *Bb) {
    const size_t tot = gens_capacity * party_capacity;
    c->gens_capacity = gens_capacity;
    c->party_capacity = party_capacity;
    c->h_gens.assign((2 + 2 * tot) * 32, 0);
    memcpy(&c->h_gens[0], Bb, 32);
    memcpy(&c->h_gens[32], B, 32);
    memcpy(&c->h_gens[64], G, tot * 32);
    memcpy(&c->h_gens[64 + tot * 32], H, tot * 32);
    if (c->d_gens) HIPCHK(c, hipFree(c->d_gens));
    c->d_gens = nullptr;
    HIPCHK(c, hipMalloc((void **)&c->d_gens, c->h_gens.size()));
    HIPCHK(c, hipMemcpy(c->d_gens, c->h_gens.data(), c->h_gens.size(), hipMemcpyHostToDevice));
    return build_tables(c);
}

extern "C" int bpgpu_gens_load(bpgpu_ctx *c, size_t gens_capacity, size_t party_capacity, const uint8_t *G, const uint8_t *H,
                               const uint8_t B[32], const uint8_t Bb[32]) {
    if (!c || !G || !H || !B || !Bb || gens_capacity == 0 || party_capacity == 0) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    return load_gens_locked(c, gens_capacity, party_capacity, G, H, B, Bb);
}

extern "C" int bpgpu_gens_export(bpgpu_ctx *c, uint8_t *G, uint8_t *H, uint8_t B[32], uint8_t Bb[32]) {
    if (!c) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->h_gens.empty()) return fail(c, BPGPU_ERR_NO_GENS, "generators not loaded");
    const size_t tot = c->gens_capacity * c->party_capacity;
    if (Bb) memcpy(Bb, &c->h_gens[0], 32);
    if (B) memcpy(B, &c->h_gens[32], 32);
    if (G) memcpy(G, &c->h_gens[64], tot * 32);
    if (H) memcpy(H, &c->h_gens[64 + tot * 32], tot * 32);
    return BPGPU_OK;
}

// id list of the generator terms of an (n, m) proof in the reference's order
// (B_blinding, B, G(n,m), H(n,m)); ids index the loaded table.  g_only: (B_blinding, B, G(n,m)) -- the bases of a
// LinearProof over bp_gens.share(j).G(n) (linear_proof.rs:405-411)
static int gen_ids_for(bpgpu_ctx *c, size_t n, size_t m, uint32_t **out, bool g_only = false) {
    auto key = std::make_pair(n, g_only ? m + ((size_t)1 << 40) : m);
    auto it = c->gen_ids_cache.find(key);
    if (it != c->gen_ids_cache.end()) {
        *out = it->second;
        return BPGPU_OK;
    }
    std::vector<uint32_t> ids;
    const size_t tot = c->gens_capacity * c->party_capacity;
    ids.push_back(0);
    ids.push_back(1);
    for (size_t j = 0; j < m; j++)
        for (size_t i = 0; i < n; i++) ids.push_back((uint32_t)(2 + j * c->gens_capacity + i));
    for (size_t j = 0; j < m && !g_only; j++)
        for (size_t i = 0; i < n; i++) ids.push_back((uint32_t)(2 + tot + j * c->gens_capacity + i));
    uint32_t *d = nullptr;
    HIPCHK(c, hipMalloc((void **)&d, ids.size() * 4));
    HIPCHK(c, hipMemcpy(d, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
    c->gen_ids_cache[key] = d;
    *out = d;
    return BPGPU_OK;
}

// the same list relative to the secondary table's compact layout [B~, B, G(n2, m2), H(n2, m2)]
static int gen_ids_for_secondary(bpgpu_ctx *c, size_t n, size_t m, uint32_t **out) {
    auto key = std::make_pair(n, m);
    auto it = c->sec.ids_cache.find(key);
    if (it != c->sec.ids_cache.end()) {
        *out = it->second;
        return BPGPU_OK;
    }
    std::vector<uint32_t> ids = {0, 1};
    for (size_t j = 0; j < m; j++)
        for (size_t i = 0; i < n; i++) ids.push_back((uint32_t)(2 + j * c->sec.n + i));
    for (size_t j = 0; j < m; j++)
        for (size_t i = 0; i < n; i++) ids.push_back((uint32_t)(2 + c->sec.n * c->sec.m + j * c->sec.n + i));
    uint32_t *d = nullptr;
    HIPCHK(c, hipMalloc((void **)&d, ids.size() * 4));
    HIPCHK(c, hipMemcpy(d, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
    c->sec.ids_cache[key] = d;
    *out = d;
    return BPGPU_OK;
}

// ============================================================================
// variable-base MSM
// ============================================================================
struct vb_plan {
    std::vector<vb_chunk> chunks;
    std::vector<uint32_t> chunk_first, term_chunk;
    uint32_t total = 0;
};
static void make_vb_plan(vb_plan &pl, size_t nbatch, const uint32_t *n_terms, uint32_t chunk = BP_VB_CHUNK) {
    pl.chunk_first.resize(nbatch + 1);
    uint32_t t0 = 0;
    for (size_t b = 0; b < nbatch; b++) {
        pl.chunk_first[b] = (uint32_t)pl.chunks.size();
        for (uint32_t k = 0; k < n_terms[b]; k += chunk) {
            vb_chunk ch;
            ch.msm = (uint32_t)b;
            ch.first = t0 + k;
            ch.count = n_terms[b] - k < chunk ? n_terms[b] - k : chunk;
            ch.pad = 0;
            for (uint32_t i = 0; i < ch.count; i++) pl.term_chunk.push_back((uint32_t)pl.chunks.size());
            pl.chunks.push_back(ch);
        }
        t0 += n_terms[b];
    }
    pl.chunk_first[nbatch] = (uint32_t)pl.chunks.size();
    pl.total = t0;
}

// Enqueue the variable-base stages up to the per-MSM column sums.  Scratch comes from the
// arena at [base + offsets].  Returns device pointers through the struct.
struct vb_dev {
    vb_chunk *chunks;
    uint32_t *chunk_first, *term_chunk, *recoded;
    ge_cached *tab;
    ge_ext *part;
    uint32_t *colq16;   // [msm][64][4][8 words]: column sums as 16-bit limbs for the wavefront Horner
    ge_ext *hq;         // [msm] Horner results
};
static void plan_vb(arena_plan &ap, const vb_plan &pl, size_t nbatch, size_t off[7], size_t tab_entries = 8) {
    off[0] = ap.add(pl.chunks.size() * sizeof(vb_chunk) + 16);
    off[1] = ap.add((nbatch + 1) * 4);
    off[2] = ap.add((size_t)pl.total * 4 + 16);
    off[3] = ap.add((size_t)pl.total * 32 + 16);
    off[4] = ap.add((size_t)pl.total * tab_entries * sizeof(ge_cached) + 16);
    off[5] = ap.add(pl.chunks.size() * 64 * sizeof(ge_ext) + 16);
    off[6] = ap.add(nbatch * 64 * sizeof(ge_cached) + nbatch * sizeof(ge_ext) + 64);   // colq16 (128 B) or colc (160 B) per column, then hq
}
static void vb_bind(bpgpu_ctx *c, const size_t off[7], vb_dev &d) {
    char *a = c->arena;
    d.chunks = (vb_chunk *)(a + off[0]);
    d.chunk_first = (uint32_t *)(a + off[1]);
    d.term_chunk = (uint32_t *)(a + off[2]);
    d.recoded = (uint32_t *)(a + off[3]);
    d.tab = (ge_cached *)(a + off[4]);
    d.part = (ge_ext *)(a + off[5]);
    d.colq16 = (uint32_t *)(a + off[6]);
    d.hq = nullptr;   // set by vb_launch (behind colq16)
}
static int vb_launch(bpgpu_ctx *c, hipStream_t s, uint32_t total, uint32_t n_chunks, size_t nbatch, const uint32_t *d_scalars,
                     const uint32_t *d_points, uint32_t *d_status, vb_dev &d) {
    if (total) {
        LAUNCH(c, s, "vb_prepare", k_vb_prepare, (total + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, total, d.chunks, d.term_chunk, d_scalars,
               d_points, d.tab, d.recoded, d_status);
        const uint32_t nt = n_chunks * 64;
        LAUNCH(c, s, "vb_window", k_vb_window, (nt + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, nt, d.chunks, d.tab, d.recoded, d.part);
    }
    const uint32_t nc = (uint32_t)nbatch * 64;
    d.hq = (ge_ext *)(d.colq16 + (size_t)nbatch * 64 * 32);
    LAUNCH(c, s, "vb_colsum", k_vb_colsum, (nc + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, nc, d.chunk_first, d.part, d.colq16, (ge_cached *)nullptr);
    LAUNCH(c, s, "horner_wave", k_horner_wave, (uint32_t)nbatch, 64, d.colq16, d.hq);
    return BPGPU_OK;
}
// ragged plans: upload the decomposition into the arena every call
static int enqueue_vb(bpgpu_ctx *c, hipStream_t s, const vb_plan &pl, size_t nbatch, const size_t off[7],
                      const uint32_t *d_scalars, const uint32_t *d_points, uint32_t *d_status, vb_dev &d, bool upload_only = false) {
    vb_bind(c, off, d);
    // the plan is a local of the caller: stage it in pinned memory, which outlives the asynchronous copies
    const size_t b0 = pl.chunks.size() * sizeof(vb_chunk), b1 = (size_t)pl.total * 4, b2 = (nbatch + 1) * 4;
    char *h = nullptr;
    int rc = pin_alloc(c, s, align_up(b0) + align_up(b1) + align_up(b2), &h);
    if (rc) return rc;
    char *h0 = h, *h1 = h0 + align_up(b0), *h2 = h1 + align_up(b1);
    if (!pl.chunks.empty()) {
        memcpy(h0, pl.chunks.data(), b0);
        memcpy(h1, pl.term_chunk.data(), b1);
        HIPCHK(c, hipMemcpyAsync(d.chunks, h0, b0, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(d.term_chunk, h1, b1, hipMemcpyHostToDevice, s));
    }
    memcpy(h2, pl.chunk_first.data(), b2);
    HIPCHK(c, hipMemcpyAsync(d.chunk_first, h2, b2, hipMemcpyHostToDevice, s));
    if (upload_only) return BPGPU_OK;
    return vb_launch(c, s, pl.total, (uint32_t)pl.chunks.size(), nbatch, d_scalars, d_points, d_status, d);
}
// uniform plans (every MSM of the batch has `per` variable-base terms): decomposition cached on the device
// chunk: terms per (chunk, window) lane -- BP_VB_CHUNK, or fewer for narrow chains (rp_narrow_chunk): part of the cache key
static int uniform_plan(bpgpu_ctx *c, size_t nbatch, size_t per, bpgpu_ctx::plan_view *out, uint32_t chunk = BP_VB_CHUNK) {
    const size_t key = per | ((size_t)chunk << 40);
    auto it = c->plan_cache.find(key);
    if (it == c->plan_cache.end() || it->second.cap < nbatch) {
        if (it != c->plan_cache.end()) {   // outgrown: launches in flight may still read the old block
            c->plan_retired.push_back(it->second.mem);
            c->plan_cache.erase(it);
        } else if (c->plan_cache.size() >= 32) {   // bounded: retire the least recently used decomposition
            auto victim = c->plan_cache.begin();
            for (auto jt = c->plan_cache.begin(); jt != c->plan_cache.end(); ++jt)
                if (jt->second.last_use < victim->second.last_use) victim = jt;
            c->plan_retired.push_back(victim->second.mem);
            c->plan_cache.erase(victim);
        }
        if (c->plan_retired.size() > 64) {   // (only a caller that keeps changing its MSM sizes gets here)
            HIPCHK(c, hipDeviceSynchronize());
            for (char *q : c->plan_retired) hipFree(q);
            c->plan_retired.clear();
        }
        // capacity: a power of two >= nbatch; small per-MSM term counts (the range-proof path: 17 .. 60 points per proof) start at 1024
        // MSMs, large ones at what ~256 k terms allow (a lone 6 179-term MSM must not build the plan of a thousand of them)
        size_t cap = 1, floor_cap = per ? (size_t)262144 / per : 1024;
        if (floor_cap > 1024) floor_cap = 1024;
        while (cap < nbatch || cap < floor_cap) cap *= 2;
        std::vector<uint32_t> nt(cap, (uint32_t)per);
        vb_plan pl;
        make_vb_plan(pl, cap, nt.data(), chunk);
        bpgpu_ctx::plan_dev pd;
        pd.cap = cap;
        pd.o1 = align_up(pl.chunks.size() * sizeof(vb_chunk) + 16);
        pd.o2 = pd.o1 + align_up((cap + 1) * 4);
        const size_t tot = pd.o2 + align_up((size_t)pl.total * 4 + 16);
        HIPCHK(c, hipMalloc((void **)&pd.mem, tot));
        if (!pl.chunks.empty()) {
            HIPCHK(c, hipMemcpy(pd.mem, pl.chunks.data(), pl.chunks.size() * sizeof(vb_chunk), hipMemcpyHostToDevice));
            HIPCHK(c, hipMemcpy(pd.mem + pd.o2, pl.term_chunk.data(), (size_t)pl.total * 4, hipMemcpyHostToDevice));
        }
        HIPCHK(c, hipMemcpy(pd.mem + pd.o1, pl.chunk_first.data(), (cap + 1) * 4, hipMemcpyHostToDevice));
        it = c->plan_cache.emplace(key, pd).first;
    }
    it->second.last_use = ++c->plan_tick;
    out->mem = it->second.mem;
    out->o1 = it->second.o1;
    out->o2 = it->second.o2;
    out->n_chunks = nbatch * ((per + chunk - 1) / chunk);
    out->total = (uint32_t)(nbatch * per);
    return BPGPU_OK;
}
static void plan_vb_uniform(arena_plan &ap, size_t nbatch, size_t per, size_t off[7], size_t tab_entries = 8, size_t chunk = BP_VB_CHUNK) {
    const size_t n_chunks = nbatch * ((per + chunk - 1) / chunk), total = nbatch * per;
    off[0] = off[1] = off[2] = 0;   // decomposition lives in the plan cache
    off[3] = ap.add(total * 32 + 16);
    off[4] = ap.add(total * tab_entries * sizeof(ge_cached) + 16);
    off[5] = ap.add(n_chunks * 64 * sizeof(ge_ext) + 16);
    off[6] = ap.add(nbatch * 64 * sizeof(ge_cached) + nbatch * sizeof(ge_ext) + 64);   // colq16 (128 B) or colc (160 B) per column, then hq
}
static int enqueue_vb_uniform(bpgpu_ctx *c, hipStream_t s, size_t nbatch, size_t per, const size_t off[7], const uint32_t *d_scalars,
                              const uint32_t *d_points, uint32_t *d_status, vb_dev &d) {
    bpgpu_ctx::plan_view pv;
    int rc = uniform_plan(c, nbatch, per, &pv);
    if (rc) return rc;
    vb_bind(c, off, d);
    d.chunks = (vb_chunk *)pv.mem;
    d.chunk_first = (uint32_t *)(pv.mem + pv.o1);
    d.term_chunk = (uint32_t *)(pv.mem + pv.o2);
    return vb_launch(c, s, pv.total, (uint32_t)pv.n_chunks, nbatch, d_scalars, d_points, d_status, d);
}

// ============================================================================
// bucket (Pippenger) path for MSMs with many variable-base terms (bucket.h)
// ============================================================================
struct bk_dev {
    uint32_t *msm_first;   // [nmsm + 1]
    fb_entry *pts;         // [total] affine Niels records
    uint32_t *rwords;      // [total][BK_RWORDS]
    uint32_t *idx;         // [nwin][total]
    bk_desc *desc;         // [nmsm * nwin][half]
    ge_ext *bsum;          // [nmsm * nwin][half]
    ge_ext *gS, *gA;       // [nmsm * nwin][leaves]: bottom level of the running-sum tree
    uint32_t *gcnt, *gcur; // [nmsm * nwin][half]: global histogram / scatter cursors of the split sort (large MSMs)
    uint32_t *colq16;      // [nmsm][64][32 words]
    ge_ext *hq;            // [nmsm]
};
static uint32_t pick_bucket_c(size_t terms_per_msm) { return terms_per_msm >= 6000 ? 12u : 8u; }
static void plan_bucket(arena_plan &ap, size_t nmsm, size_t total, bk_params prm, size_t off[12]) {
    off[0] = ap.add((nmsm + 1) * 4);
    off[1] = ap.add(total * sizeof(fb_entry) + 16);
    off[2] = ap.add(total * BK_RWORDS * 4 + 16);
    off[3] = ap.add((size_t)prm.nwin * total * 4 + 16);
    off[4] = ap.add(nmsm * prm.nwin * prm.half * sizeof(bk_desc));
    off[5] = ap.add(nmsm * prm.nwin * prm.half * sizeof(ge_ext));
    off[6] = ap.add(nmsm * 64 * 128);
    off[7] = ap.add(nmsm * sizeof(ge_ext));
    off[8] = ap.add(nmsm * prm.nwin * bk_leaves(prm) * sizeof(ge_ext));
    off[9] = ap.add(nmsm * prm.nwin * bk_leaves(prm) * sizeof(ge_ext));
    off[10] = ap.add(nmsm * prm.nwin * prm.half * 4);
    off[11] = ap.add(nmsm * prm.nwin * prm.half * 4);
}
static void bucket_bind(bpgpu_ctx *c, const size_t off[12], bk_dev &d) {
    char *a = c->arena;
    d.msm_first = (uint32_t *)(a + off[0]);
    d.pts = (fb_entry *)(a + off[1]);
    d.rwords = (uint32_t *)(a + off[2]);
    d.idx = (uint32_t *)(a + off[3]);
    d.desc = (bk_desc *)(a + off[4]);
    d.bsum = (ge_ext *)(a + off[5]);
    d.colq16 = (uint32_t *)(a + off[6]);
    d.hq = (ge_ext *)(a + off[7]);
    d.gS = (ge_ext *)(a + off[8]);
    d.gA = (ge_ext *)(a + off[9]);
    d.gcnt = (uint32_t *)(a + off[10]);
    d.gcur = (uint32_t *)(a + off[11]);
}
// counting sort of the terms of every (MSM, window) by digit + population sort of the buckets.  Few thousand terms per
// MSM: one workgroup per (MSM, window), everything in LDS; more: the terms are shared by nsub workgroups (global
// histogram, one scan workgroup, scatter through global cursors).  per_msm: the largest term count of one MSM.
static int enqueue_bucket_sort(bpgpu_ctx *c, hipStream_t s, bk_params prm, uint32_t nmsm, uint32_t total, size_t per_msm, int single, bk_dev &d,
                               const uint32_t *skip_status, uint32_t skip_div) {
    const uint32_t nbw = nmsm * prm.nwin;
    if (per_msm < 32768) {
        if (prm.lanes == 64) LAUNCH(c, s, "bk_sort", k_bk_sort<64>, nbw, 64, prm, d.msm_first, total, single, d.rwords, d.idx, d.desc, skip_status, skip_div);
        else LAUNCH(c, s, "bk_sort", k_bk_sort<256>, nbw, 256, prm, d.msm_first, total, single, d.rwords, d.idx, d.desc, skip_status, skip_div);
        return BPGPU_OK;
    }
    uint32_t nsub = (uint32_t)((per_msm + 4095) / 4096);
    if (nsub > 128) nsub = 128;
    HIPCHK(c, hipMemsetAsync(d.gcnt, 0, (size_t)nbw * prm.half * 4, s));
    for (int phase = 0; phase < 3; phase++) {
        const uint32_t grid = phase == 1 ? nbw : nbw * nsub;
        if (prm.lanes == 64)
            LAUNCH(c, s, "bk_sort", k_bk_sort_big<64>, grid, 64, phase, nsub, prm, d.msm_first, total, single, d.rwords, d.idx, d.desc, d.gcnt, d.gcur, skip_status, skip_div);
        else
            LAUNCH(c, s, "bk_sort", k_bk_sort_big<256>, grid, 256, phase, nsub, prm, d.msm_first, total, single, d.rwords, d.idx, d.desc, d.gcnt, d.gcur, skip_status, skip_div);
    }
    return BPGPU_OK;
}
// bucket sums -> window sums (running-sum tree: wide leaf level, packed upper levels) -> column sums for the Horner chain
static void enqueue_bucket_reduce(bpgpu_ctx *c, hipStream_t s, bk_params prm, uint32_t nbw, bk_dev &d) {
    const uint32_t nl = nbw * bk_leaves(prm);
    LAUNCH(c, s, "bk_leaf", k_bk_leaf, (nl + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, nl, prm, d.bsum, d.gS, d.gA);
    if (prm.c == 8) LAUNCH(c, s, "bk_tree", k_bk_tree<8>, (nbw + 63) / 64, 64, prm, nbw, d.gS, d.gA, d.colq16);
    else LAUNCH(c, s, "bk_tree", k_bk_tree<12>, (nbw + 1) / 2, 64, prm, nbw, d.gS, d.gA, d.colq16);
}
// first-term offsets of the MSMs -> device (through pinned staging)
static int bucket_upload_first(bpgpu_ctx *c, hipStream_t s, size_t nmsm, const uint32_t *n_terms, size_t uniform_per, bk_dev &d) {
    char *h = nullptr;
    int rc = pin_alloc(c, s, (nmsm + 1) * 4, &h);
    if (rc) return rc;
    uint32_t *f = (uint32_t *)h;
    uint32_t t0 = 0;
    for (size_t b = 0; b < nmsm; b++) {
        f[b] = t0;
        t0 += n_terms ? n_terms[b] : (uint32_t)uniform_per;
    }
    f[nmsm] = t0;
    HIPCHK(c, hipMemcpyAsync(d.msm_first, h, (nmsm + 1) * 4, hipMemcpyHostToDevice, s));
    return BPGPU_OK;
}
// sort -> bucket sums -> window sums -> Horner chain; leaves the MSM sums in d.hq
static int enqueue_bucket_tail(bpgpu_ctx *c, hipStream_t s, bk_params prm, size_t nmsm, size_t total, size_t per_msm, bk_dev &d) {
    const uint32_t nbw = (uint32_t)(nmsm * prm.nwin), tot32 = (uint32_t)total;
    int rcs = enqueue_bucket_sort(c, s, prm, (uint32_t)nmsm, tot32, per_msm, 0, d, nullptr, 0u);
    if (rcs) return rcs;
    const uint32_t nt = nbw * prm.half;
    const uint32_t lim = bk_chain_lim(per_msm, prm);
    LAUNCH(c, s, "bk_accum", k_bk_accum, (nt + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, nt, prm, tot32, d.desc, d.idx, d.pts, d.bsum, lim);
    const uint32_t hg = bk_heavy_groups(prm, nmsm);
    LAUNCH(c, s, "bk_heavy", k_bk_heavy, nbw * hg, 64, prm, tot32, (const bk_desc *)d.desc, (const uint32_t *)d.idx, (const fb_entry *)d.pts, d.bsum, lim, hg);
    enqueue_bucket_reduce(c, s, prm, nbw, d);
    LAUNCH(c, s, "horner_wave", k_horner_wave, (uint32_t)nmsm, 64, d.colq16, d.hq);
    return BPGPU_OK;
}
// ---- the fused chain (bucket2.h): c = 8, every MSM of the batch <= BK2_MAX_TERMS terms ----------------------------------------
static bool bucket2_applies(bpgpu_ctx *c, bk_params prm, size_t per_msm_max) { return c->bucket_chain == 0 && prm.c == BK2_C && per_msm_max <= BK2_MAX_TERMS; }
static void plan_bucket2(arena_plan &ap, size_t nmsm, size_t total, size_t off[12]) {
    const bk_params prm = bk_make(BK2_C);
    off[0] = ap.add((nmsm + 1) * 4);
    off[1] = ap.add(total * sizeof(fb_entry) + 16);
    off[2] = ap.add((size_t)BK2_NWIN * total + 16);   // digits, window-major (bk_dev::rwords)
    off[3] = off[4] = off[10] = off[11] = 0;          // no index lists, descriptors or global histograms
    off[5] = ap.add(nmsm * prm.nwin * prm.half * sizeof(ge_ext));
    off[6] = ap.add(nmsm * 64 * 128);
    off[7] = ap.add(nmsm * sizeof(ge_ext));
    off[8] = ap.add(nmsm * prm.nwin * BK2_FAST_LEAVES * sizeof(ge_ext));   // (8 leaves per window in a wide batch, 32 in a narrow one)
    off[9] = ap.add(nmsm * prm.nwin * BK2_FAST_LEAVES * sizeof(ge_ext));
}
// narrow chains end with the short-chain tail (bucket2.h: bk2_fast_v), wide batches with the one that executes fewer instructions
static bool bucket2_fast_tail(bpgpu_ctx *c, size_t nmsm) { return c->fast_tail < 0 ? nmsm < BK2_FAST_MAX_MSMS : c->fast_tail != 0; }
static uint32_t bucket2_lanes(bpgpu_ctx *c, size_t nmsm) {
    if (c->bucket_lanes) return (uint32_t)c->bucket_lanes;
    // a wide batch fills the device with one wavefront per (MSM, window) and keeps the runs long (2 081 terms: 33 additions per
    // lane, one head piece per lane); a lone MSM (32 workgroups in all) wants the shortest chain
    return nmsm >= 8 ? 64u : (nmsm >= 2 ? 128u : 256u);
}
// decode + digits -> window workgroups (sort in LDS + bucket sums) -> window sums -> Horner chain; leaves the MSM sums in d.hq
static int enqueue_bucket2(bpgpu_ctx *c, hipStream_t s, size_t nmsm, size_t total, size_t per_msm, const uint32_t *d_scalars, const uint32_t *d_points,
                           uint32_t *d_status, bk_dev &d) {
    const bk_params prm = bk_make(BK2_C);
    const uint32_t nbw = (uint32_t)(nmsm * prm.nwin), tot32 = (uint32_t)total;
    uint8_t *dig = (uint8_t *)d.rwords;
    LAUNCH(c, s, "bk_prepare", k_bk2_prepare, (tot32 + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, tot32, (uint32_t)nmsm, d.msm_first, d_scalars, d_points, d.pts, dig, d_status);
    const uint32_t lanes = bucket2_lanes(c, nmsm);
    const size_t shm = ((per_msm * 2 + 15) / 16) * 16;
    const int xcd_map = (nmsm % 8) == 0 ? 1 : 0;
    if (lanes == 64) LAUNCH_SHM(c, s, "bk_window", k_bk2_window<64>, nbw, 64, shm, (uint32_t)nmsm, xcd_map, d.msm_first, tot32, (const uint8_t *)dig, (const fb_entry *)d.pts, d.bsum);
    else if (lanes == 128) LAUNCH_SHM(c, s, "bk_window", k_bk2_window<128>, nbw, 128, shm, (uint32_t)nmsm, xcd_map, d.msm_first, tot32, (const uint8_t *)dig, (const fb_entry *)d.pts, d.bsum);
    else LAUNCH_SHM(c, s, "bk_window", k_bk2_window<256>, nbw, 256, shm, (uint32_t)nmsm, xcd_map, d.msm_first, tot32, (const uint8_t *)dig, (const fb_entry *)d.pts, d.bsum);
    if (bucket2_fast_tail(c, nmsm)) {
        const uint32_t nl = nbw * BK2_FAST_LEAVES;
        LAUNCH(c, s, "bk_leaf", k_bk2_leafv, (nl + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, nl, (const ge_ext *)d.bsum, d.gA);
    } else {
        const uint32_t nl = nbw * bk_leaves(prm);
        LAUNCH(c, s, "bk_leaf", k_bk_leaf, (nl + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, nl, prm, d.bsum, d.gS, d.gA);   // (the tree's upper level is the tail's first phase)
    }
    return BPGPU_OK;
}
// the generator half of a fused chain as ONE launch (msm_fixed.h: fb_walk_thread): partial sums -> partial[npart][nbatch]
static uint32_t fb_walk_parts(bpgpu_ctx *c, size_t nbatch, uint32_t n_gen_terms) {
    const uint32_t target = c->walk_waves ? (uint32_t)c->walk_waves : 2048u;   // (512 / 1024 / 2048 / 4096: 109 / 111 / 113 / 113 k MSMs/s, profiles/r06/cfg5_two_buffers_uncapped_and_walk_waves_ab.txt)
    if (nbatch < 32) {   // lane = slice of one MSM's generator terms: workgroups of one wavefront
        uint32_t nwg = (uint32_t)((target + nbatch - 1) / nbatch);
        const uint32_t most = (n_gen_terms + 63) / 64;
        if (nwg > most) nwg = most;
        return nwg ? nwg : 1;
    }
    const uint32_t nblk_p = (uint32_t)((nbatch + 63) / 64);
    uint32_t per = (uint32_t)(((uint64_t)n_gen_terms * nblk_p + target - 1) / target);   // generator terms per wavefront
    if (per == 0) per = 1;
    const uint32_t nslice = (n_gen_terms + per - 1) / per;
    return (nslice + 3) / 4;
}
static void enqueue_fb_walk(bpgpu_ctx *c, hipStream_t s, fb_params prm, size_t nbatch, uint32_t n_gen_terms, uint32_t nwg, const uint32_t *d_gen_scalars, const uint32_t *d_ids,
                            ge_ext *d_partial, uint32_t *d_status) {
    if (nbatch < 32) {
        LAUNCH(c, s, "fb_walk", k_fb_walk1, (uint32_t)(nwg * nbatch), 64, prm, (uint32_t)nbatch, nwg, n_gen_terms, d_gen_scalars, d_ids, (const fb_entry *)c->d_table, d_partial, d_status);
    } else {
        const uint32_t nblk_p = (uint32_t)((nbatch + 63) / 64);
        LAUNCH(c, s, "fb_walk", k_fb_walk<4>, nwg * nblk_p, 256, prm, (uint32_t)nbatch, nblk_p, nwg, n_gen_terms, d_gen_scalars, d_ids, (const fb_entry *)c->d_table, d_partial, d_status);
    }
}
static bool bucket_fits(size_t nmsm, size_t total, bk_params prm) {
    return (uint64_t)nmsm * prm.nwin * prm.half <= 0x7fffffffull && (uint64_t)total * prm.nwin <= 0x7fffffffull;
}

static int msm_batch_dev_locked(bpgpu_ctx *c, size_t nbatch, const uint32_t *n_terms, const void *d_scalars, const void *d_points,
                                void *d_out, void *d_status_bytes, hipStream_t s) {
    if (nbatch == 0) return BPGPU_OK;
    {   // many terms per MSM: bucket path (bucket.h); otherwise the table-lookup path below
        uint64_t total = 0;
        for (size_t b = 0; b < nbatch; b++) total += n_terms[b];
        const bk_params prm = bk_make(pick_bucket_c(total / nbatch));
        if (total / nbatch >= (c->bucket_min ? c->bucket_min : BK_MIN_TERMS) && bucket_fits(nbatch, total, prm)) {
            size_t per_msm = 0;
            for (size_t b = 0; b < nbatch; b++) per_msm = n_terms[b] > per_msm ? n_terms[b] : per_msm;
            const bool fused = bucket2_applies(c, prm, per_msm);
            arena_plan ap;
            size_t off[12];
            if (fused) plan_bucket2(ap, nbatch, total, off);
            else plan_bucket(ap, nbatch, total, prm, off);
            const size_t off_status = ap.add(nbatch * 4);
            int rc = arena_reserve(c, ap.total);
            if (rc) return rc;
            bk_dev d;
            bucket_bind(c, off, d);
            uint32_t *d_status = (uint32_t *)(c->arena + off_status);
            HIPCHK(c, hipMemsetAsync(d_status, 0, nbatch * 4, s));
            rc = bucket_upload_first(c, s, nbatch, n_terms, 0, d);
            if (rc) return rc;
            if (fused) {
                rc = enqueue_bucket2(c, s, nbatch, total, per_msm, (const uint32_t *)d_scalars, (const uint32_t *)d_points, d_status, d);
            } else {
                LAUNCH(c, s, "bk_prepare", k_bk_prepare, ((uint32_t)total + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, (uint32_t)total, (uint32_t)nbatch, d.msm_first,
                       (const uint32_t *)d_scalars, (const uint32_t *)d_points, d.pts, d.rwords, d_status, prm);
                rc = enqueue_bucket_tail(c, s, prm, nbatch, total, per_msm, d);
            }
            if (rc) return rc;
            if (fused) {
                if (bucket2_fast_tail(c, nbatch))
                    LAUNCH(c, s, "msm_tail", k_msm_tail_fast, (uint32_t)nbatch, 256, (uint32_t)nbatch, 1, (const ge_ext *)d.gA, 0u, (const ge_ext *)nullptr,
                           (const uint32_t *)d_status, (uint32_t *)d_out, (uint8_t *)nullptr, (uint8_t *)d_status_bytes);
                else
                    LAUNCH(c, s, "msm_tail", k_msm_tail, (uint32_t)nbatch, 64, (uint32_t)nbatch, 1, (const ge_ext *)d.gS, (const ge_ext *)d.gA, 0u, (const ge_ext *)nullptr,
                           (const uint32_t *)d_status, (uint32_t *)d_out, (uint8_t *)nullptr, (uint8_t *)d_status_bytes);
            } else {
                LAUNCH(c, s, "vb_horner", k_vb_horner, ((uint32_t)nbatch + 63) / 64, 64, (uint32_t)nbatch, d.hq, d_status, (uint32_t *)d_out);
                LAUNCH(c, s, "status_bytes", k_status_bytes, ((uint32_t)nbatch + 63) / 64, 64, (uint32_t)nbatch, d_status, (uint8_t *)d_status_bytes);
            }
            HIPCHK(c, hipGetLastError());
            return BPGPU_OK;
        }
    }
    // a few small MSMs (the boundary function called from one thread: nothing else to fill the device with): the narrow form -- second
    // tables, chunks of ~sqrt(N) terms, one tail launch (k_msm.hip: k_vb_prepare_hi / k_vb_window_hi / k_vb_tail_narrow)
    uint64_t tot_terms = 0, max_terms = 0;
    for (size_t b = 0; b < nbatch; b++) {
        tot_terms += n_terms[b];
        if (n_terms[b] > max_terms) max_terms = n_terms[b];
    }
    const bool narrow = c->msm_narrow && nbatch <= 16 && tot_terms > 0 && tot_terms <= 768;   // (same-box A/B, profiles/r06/msm_narrow_ab.txt: 29 / 147 / 542 terms 0.44 / 0.475 / 0.52 -> 0.33 / 0.38 / 0.465 ms per blocking call; 1024 terms 0.56 -> 0.59: the wavefront-per-term role then needs a second round)
    uint32_t chunk = BP_VB_CHUNK;
    if (narrow) {
        chunk = 4;
        while ((uint64_t)chunk * chunk < max_terms && chunk < BP_VB_CHUNK) chunk++;
    }
    vb_plan pl;
    make_vb_plan(pl, nbatch, n_terms, chunk);
    arena_plan ap;
    size_t off[7];
    const uint32_t lev = narrow ? 2u : 1u;   // (four table levels -- a 16-window chain, as the narrowest range-proof chains take -- were measured here: no gain,
                                             // profiles/r06/msm_narrow_four_levels_ab.txt: this call's long pole is its first launch, not the chain)
    plan_vb(ap, pl, nbatch, off, 8 * lev);
    const size_t off_status = ap.add(nbatch * 4);
    int rc = arena_reserve(c, ap.total);
    if (rc) return rc;
    uint32_t *d_status = (uint32_t *)(c->arena + off_status);
    HIPCHK(c, hipMemsetAsync(d_status, 0, nbatch * 4, s));
    vb_dev d;
    if (narrow) {
        rc = enqueue_vb(c, s, pl, nbatch, off, (const uint32_t *)d_scalars, (const uint32_t *)d_points, d_status, d, true);
        if (rc) return rc;
        const uint32_t total = pl.total, n_lb = (total + BP_BLOCK - 1) / BP_BLOCK, nt = (uint32_t)pl.chunks.size() * 64;
        ge_cached *tab_hi = d.tab + (size_t)8 * total;
        LAUNCH(c, s, "vb_prepare", k_vb_prepare_hi, n_lb + (lev - 1) * total, BP_BLOCK, total, n_lb, d.chunks, d.term_chunk, (const uint32_t *)d_scalars, (const uint32_t *)d_points,
               d.tab, d.recoded, d_status, tab_hi, lev);
        LAUNCH(c, s, "vb_window", k_vb_window_hi, (nt + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, nt, d.chunks, d.tab, d.recoded, d.part, (const ge_cached *)tab_hi, lev, total);
        LAUNCH(c, s, "vb_tail", k_vb_tail_narrow, (uint32_t)nbatch, 64, (uint32_t)nbatch, d.chunk_first, d.part, (const uint32_t *)d_status, (uint32_t *)d_out,
               (uint8_t *)d_status_bytes, lev);
        HIPCHK(c, hipGetLastError());
        return BPGPU_OK;
    }
    rc = enqueue_vb(c, s, pl, nbatch, off, (const uint32_t *)d_scalars, (const uint32_t *)d_points, d_status, d);
    if (rc) return rc;
    LAUNCH(c, s, "vb_horner", k_vb_horner, ((uint32_t)nbatch + 63) / 64, 64, (uint32_t)nbatch, d.hq, d_status, (uint32_t *)d_out);
    LAUNCH(c, s, "status_bytes", k_status_bytes, ((uint32_t)nbatch + 63) / 64, 64, (uint32_t)nbatch, d_status, (uint8_t *)d_status_bytes);
    HIPCHK(c, hipGetLastError());
    return BPGPU_OK;
}

extern "C" int bpgpu_msm_batch_dev(bpgpu_ctx *c, size_t nbatch, const uint32_t *n_terms, const void *d_scalars, const void *d_points,
                                   void *d_out, void *d_status, void *stream) {
    if (!c || (nbatch && (!n_terms || !d_out || !d_status))) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t total = 0;
    for (size_t b = 0; b < nbatch; b++) total += n_terms[b];
    if (nbatch > 0x7fffffffu / 64 || total > 0x7fffffffu / 64) return fail(c, BPGPU_ERR_INVALID_ARG, "batch too large");
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    int rc = ctx_enter(c, s);
    if (rc) return rc;
    rc = msm_batch_dev_locked(c, nbatch, n_terms, d_scalars, d_points, d_out, d_status, s);
    const int rc2 = ctx_leave(c, s);
    return rc ? rc : rc2;
}

extern "C" int bpgpu_msm_batch(bpgpu_ctx *c, size_t nbatch, const uint32_t *n_terms, const uint8_t *scalars, const uint8_t *points,
                               uint8_t *out, uint8_t *status) {
    if (!c || (nbatch && (!n_terms || !out || !status))) return BPGPU_ERR_INVALID_ARG;
    if (nbatch == 0) return BPGPU_OK;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t total = 0;
    for (size_t b = 0; b < nbatch; b++) total += n_terms[b];
    if (total && (!scalars || !points)) return BPGPU_ERR_INVALID_ARG;
    if (nbatch > 0x7fffffffu / 64 || total > 0x7fffffffu / 64) return fail(c, BPGPU_ERR_INVALID_ARG, "batch too large");
    // persistent device IO buffer + pinned staging: no allocation in steady state, one wait at the end
    const size_t sz_terms = align_up(total * 32 + 16), sz_out = align_up(nbatch * 32), sz_st = align_up(nbatch);
    hipStream_t s = c->stream;
    int rc = ctx_enter(c, s);
    if (rc) return rc;
    rc = io_reserve(c, 2 * sz_terms + sz_out + sz_st);
    if (rc) return rc;
    char *h = nullptr;
    rc = pin_alloc(c, s, 2 * sz_terms + sz_out + sz_st, &h);
    if (rc) return rc;
    char *d_s = c->io_dev, *d_p = d_s + sz_terms, *d_o = d_p + sz_terms;
    char *h_o = h + 2 * sz_terms;
    if (total) {
        memcpy(h, scalars, total * 32);
        memcpy(h + sz_terms, points, total * 32);
        HIPCHK(c, hipMemcpyAsync(d_s, h, 2 * sz_terms, hipMemcpyHostToDevice, s));
    }
    rc = msm_batch_dev_locked(c, nbatch, n_terms, d_s, d_p, d_o, d_o + sz_out, s);
    if (!rc && hipMemcpyAsync(h_o, d_o, sz_out + sz_st, hipMemcpyDeviceToHost, s) != hipSuccess) rc = fail(c, BPGPU_ERR_HIP, "D2H copy failed");
    const int rc2 = ctx_leave(c, s), rc3 = host_wait(c, s);
    if (rc || rc2 || rc3) return rc ? rc : (rc2 ? rc2 : rc3);
    memcpy(out, h_o, nbatch * 32);
    memcpy(status, h_o + sz_out, nbatch);
    return BPGPU_OK;
}

// ============================================================================
// shared-generator MSM
// ============================================================================
static uint32_t pick_splits(bpgpu_ctx *c, size_t nbatch, uint32_t npairs, bool walk_alone_in_launch = false) {
    if (c->splits) return c->splits;
    if (c->splits_hint && npairs <= 8192) {   // the pool knows how much else is in flight (pool.hip flush_dev)
        uint32_t s = c->splits_hint;
        while (s > 8 && npairs / s < 8) s -= 8;
        return s;
    }
    // >= 1024 wavefronts (one per SIMD).  A lone table walk runs 1.45x faster with 4096 (4 per SIMD hide its VALU
    // dependency stalls: 4.0 ms at 2048 wavefronts vs 2.8 ms at 4096, batch 16384), but it shares its launch with the
    // longer Horner role, and with several batches in flight the device is work-bound: every split costs one
    // more point addition per proof in the reduction (measured at 48 x 1024: 256 splits 4.35 M/s, 128: 4.5, 64:
    // 4.57, 32: 4.45) and with <= 64 partial sums the finish kernel needs no separate reduction launch.
    // Aggregated shapes (thousands of generator terms per proof, e.g. m = 16: 38950 pairs) are all table walk:
    // they keep the 4096-wavefront target (cfg3: 478 k/s vs 443 k/s).
    // A wide chain whose Horner chains run aside (rp_chain_forms) has the walk alone in launch 4 and nobody told it how much else is in
    // flight (a context on its own, >= 8192 proofs): two full rounds of three wavefronts per SIMD (a lone 16 384-proof chain ran its walk at 0.59 of the mixed-addition rate
    // with 2048 wavefronts, profiles/r03/chain16384_alone_kernel_stats.csv).
    const uint32_t nblk = (uint32_t)((nbatch + FB_BLOCK - 1) / FB_BLOCK);
    const bool lone = walk_alone_in_launch && c->busy_hint < 0;   // (a pool that said "busy" has other chains to fill the device)
    const uint32_t target = npairs > 8192 ? 4096 : (lone ? 6144 : 1024);
    uint32_t s = (target + nblk - 1) / nblk;
    s = (s + 7) & ~7u;
    if (lone && s > 64) s = 64;   // (more would need a reduction launch in front of the finish)
    while (s > 8 && npairs / s < 8) s -= 8;
    if (s < 8) s = 8;
    return s;
}

// Tree-reduce the per-split partial points (8-way per level) until at most `until` remain per proof.
// `buf` must hold nsplit*nbatch + ceil(nsplit/8)*nbatch (+ ...) points: callers reserve 2*nsplit*nbatch.
#define FB_REDUCE_GROUP 8
static void enqueue_fb_reduce(bpgpu_ctx *c, hipStream_t s, uint32_t nbatch, uint32_t nsplit, ge_ext *buf, ge_ext **out, uint32_t *nout,
                              uint32_t until = FB_REDUCE_GROUP) {
    ge_ext *cur = buf;
    uint32_t n = nsplit;
    ge_ext *next = buf + (size_t)nsplit * nbatch;
    while (n > until) {
        const uint32_t ng = (n + FB_REDUCE_GROUP - 1) / FB_REDUCE_GROUP;
        const uint32_t nt = ng * nbatch;
        LAUNCH(c, s, "fb_reduce", k_fb_reduce, (nt + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, nt, nbatch, n, (uint32_t)FB_REDUCE_GROUP, cur, next);
        cur = next;
        next = next + (size_t)ng * nbatch;
        n = ng;
    }
    *out = cur;
    *nout = n;
}

static int msm_shared_dev_locked(bpgpu_ctx *c, size_t n, size_t m, size_t nbatch, size_t n_unique, const void *d_gen_scalars,
                                 const void *d_uniq_scalars, const void *d_uniq_points, void *d_out, void *d_status_bytes,
                                 void *d_verdict, hipStream_t s, bool g_only = false, bool ct = false) {
    // ct: the constant-time walk (generator terms only: n_unique == 0) -- the prover's V, A, S, T_1, T_2
    if (nbatch == 0) return BPGPU_OK;
    if (!c->d_table) return fail(c, BPGPU_ERR_NO_GENS, "generators not loaded");
    if (ct) {
        if (n_unique) return fail(c, BPGPU_ERR_INVALID_ARG, "the constant-time path serves generator terms only");
        const int rcc = build_ct_table(c);
        if (rcc) return rcc;
    }
    if (n == 0 || m == 0 || n > c->gens_capacity || m > c->party_capacity)
        return fail(c, BPGPU_ERR_NO_GENS, "generators too small for n=%zu m=%zu", n, m);
    const uint32_t n_gen_terms = (uint32_t)((g_only ? 1 : 2) * n * m + 2);   // g_only: d_gen_scalars rows are (B_blinding, B, G(n, m))
    const fb_params prm = ct ? c->prm_ct : c->prm;
    const uint32_t npairs = n_gen_terms * prm.nwin;
    // thread / element counts are 32-bit in the kernels: refuse what does not fit (grids of <= 2^31 / 64 blocks)
    if ((uint64_t)n_gen_terms * nbatch > 0x7fffffffull || (uint64_t)nbatch * ((n_unique + BP_VB_CHUNK - 1) / BP_VB_CHUNK) * 64 > 0x7fffffffull ||
        (uint64_t)nbatch * n_unique > 0x7fffffffull / 64)
        return fail(c, BPGPU_ERR_INVALID_ARG, "batch too large for this shape");
    uint32_t *d_ids = nullptr;
    int rc = gen_ids_for(c, n, m, &d_ids, g_only);
    if (rc) return rc;
    const uint32_t nsplit = pick_splits(c, nbatch, npairs);
    const bk_params bkp = bk_make(pick_bucket_c(n_unique));
    const bool use_bucket = n_unique >= (c->bucket_min ? c->bucket_min : BK_MIN_TERMS) && bucket_fits(nbatch, nbatch * n_unique, bkp);
    arena_plan ap;
    size_t off[7], boff[12];
    const bool fused = use_bucket && !ct && bucket2_applies(c, bkp, n_unique);
    if (fused) plan_bucket2(ap, nbatch, nbatch * n_unique, boff);
    else if (use_bucket) plan_bucket(ap, nbatch, nbatch * n_unique, bkp, boff);
    // a few MSMs with a few points of their own (the mega-check of ONE proof as the caller's own MSM call: 130 generator terms + 17 points): the
    // narrow form -- second tables, ~sqrt chunks, the generator half as one launch with the recoding in registers, one tail launch
    const bool narrow_sh = c->msm_narrow && !ct && !use_bucket && n_unique > 0 && nbatch <= 16 && nbatch * n_unique <= 768;
    uint32_t sh_chunk = BP_VB_CHUNK;
    if (narrow_sh) {
        sh_chunk = 4;
        while ((size_t)sh_chunk * sh_chunk < n_unique && sh_chunk < BP_VB_CHUNK) sh_chunk++;
    }
    const uint32_t sh_lev = narrow_sh ? 2u : 1u;
    if (!fused && !use_bucket) plan_vb_uniform(ap, nbatch, n_unique, off, 8 * sh_lev, sh_chunk);
    const size_t off_status = ap.add(nbatch * 4);
    if (narrow_sh) {
        const uint32_t nwg = fb_walk_parts(c, nbatch, n_gen_terms);
        const size_t off_part = ap.add((size_t)nwg * nbatch * sizeof(ge_ext) + 16);
        rc = arena_reserve(c, ap.total);
        if (rc) return rc;
        uint32_t *d_status = (uint32_t *)(c->arena + off_status);
        ge_ext *d_part = (ge_ext *)(c->arena + off_part);
        HIPCHK(c, hipMemsetAsync(d_status, 0, nbatch * 4, s));
        hipStream_t s2 = c->msm_fork ? c->stream2 : s;
        if (s2 != s) {
            HIPCHK(c, hipEventRecord(c->fork_ev, s));
            HIPCHK(c, hipStreamWaitEvent(s2, c->fork_ev, 0));
        }
        enqueue_fb_walk(c, s2, prm, nbatch, n_gen_terms, nwg, (const uint32_t *)d_gen_scalars, d_ids, d_part, d_status);
        if (s2 != s) HIPCHK(c, hipEventRecord(c->join_ev, s2));
        bpgpu_ctx::plan_view pv;
        rc = uniform_plan(c, nbatch, n_unique, &pv, sh_chunk);
        if (rc) return rc;
        vb_dev d{};
        vb_bind(c, off, d);
        d.chunks = (vb_chunk *)pv.mem;
        d.chunk_first = (uint32_t *)(pv.mem + pv.o1);
        d.term_chunk = (uint32_t *)(pv.mem + pv.o2);
        const uint32_t total = pv.total, n_lb = (total + BP_BLOCK - 1) / BP_BLOCK, nt = (uint32_t)pv.n_chunks * 64;
        ge_cached *tab_hi = d.tab + (size_t)8 * total;
        LAUNCH(c, s, "vb_prepare", k_vb_prepare_hi, n_lb + (sh_lev - 1) * total, BP_BLOCK, total, n_lb, d.chunks, d.term_chunk, (const uint32_t *)d_uniq_scalars,
               (const uint32_t *)d_uniq_points, d.tab, d.recoded, d_status, tab_hi, sh_lev);
        LAUNCH(c, s, "vb_window", k_vb_window_hi, (nt + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, nt, d.chunks, d.tab, d.recoded, d.part, (const ge_cached *)tab_hi, sh_lev, total);
        if (s2 != s) HIPCHK(c, hipStreamWaitEvent(s, c->join_ev, 0));
        LAUNCH(c, s, "msm_tail", k_shared_tail_narrow, (uint32_t)nbatch, 128, (uint32_t)nbatch, d.chunk_first, d.part, nwg, (const ge_ext *)d_part,
               (const uint32_t *)d_status, (uint32_t *)d_out, (uint8_t *)d_verdict, (uint8_t *)d_status_bytes, sh_lev);
        HIPCHK(c, hipGetLastError());
        return BPGPU_OK;
    }
    if (fused) {   // the fused chain: generator half = one launch, tail = one launch (bucket2.h)
        const uint32_t nwg = fb_walk_parts(c, nbatch, n_gen_terms);
        const size_t off_part = ap.add((size_t)nwg * nbatch * sizeof(ge_ext) + 16);
        rc = arena_reserve(c, ap.total);
        if (rc) return rc;
        uint32_t *d_status = (uint32_t *)(c->arena + off_status);
        ge_ext *d_part = (ge_ext *)(c->arena + off_part);
        HIPCHK(c, hipMemsetAsync(d_status, 0, nbatch * 4, s));
        hipStream_t s2 = c->msm_fork ? c->stream2 : s;
        if (s2 != s) {
            HIPCHK(c, hipEventRecord(c->fork_ev, s));
            HIPCHK(c, hipStreamWaitEvent(s2, c->fork_ev, 0));
        }
        enqueue_fb_walk(c, s2, prm, nbatch, n_gen_terms, nwg, (const uint32_t *)d_gen_scalars, d_ids, d_part, d_status);
        if (s2 != s) HIPCHK(c, hipEventRecord(c->join_ev, s2));
        bk_dev bd;
        bucket_bind(c, boff, bd);
        rc = bucket_upload_first(c, s, nbatch, nullptr, n_unique, bd);
        if (rc) return rc;
        rc = enqueue_bucket2(c, s, nbatch, nbatch * n_unique, n_unique, (const uint32_t *)d_uniq_scalars, (const uint32_t *)d_uniq_points, d_status, bd);
        if (rc) return rc;
        if (s2 != s) HIPCHK(c, hipStreamWaitEvent(s, c->join_ev, 0));
        if (bucket2_fast_tail(c, nbatch))
            LAUNCH(c, s, "msm_tail", k_msm_tail_fast, (uint32_t)nbatch, 256, (uint32_t)nbatch, 1, (const ge_ext *)bd.gA, nwg, (const ge_ext *)d_part,
                   (const uint32_t *)d_status, (uint32_t *)d_out, (uint8_t *)d_verdict, (uint8_t *)d_status_bytes);
        else
            LAUNCH(c, s, "msm_tail", k_msm_tail, (uint32_t)nbatch, 64, (uint32_t)nbatch, 1, (const ge_ext *)bd.gS, (const ge_ext *)bd.gA, nwg, (const ge_ext *)d_part,
                   (const uint32_t *)d_status, (uint32_t *)d_out, (uint8_t *)d_verdict, (uint8_t *)d_status_bytes);
        HIPCHK(c, hipGetLastError());
        return BPGPU_OK;
    }
    const size_t off_digits = ap.add((size_t)npairs * nbatch * sizeof(fb_digit) + 16);
    const size_t off_partial = ap.add((size_t)2 * nsplit * nbatch * sizeof(ge_ext) + 16);
    rc = arena_reserve(c, ap.total);
    if (rc) return rc;
    uint32_t *d_status = (uint32_t *)(c->arena + off_status);
    fb_digit *d_digits = (fb_digit *)(c->arena + off_digits);
    ge_ext *d_partial = (ge_ext *)(c->arena + off_partial);
    HIPCHK(c, hipMemsetAsync(d_status, 0, nbatch * 4, s));
    // the generator-table half (recode, walk, partial reduction) on the context's second stream, beside the per-MSM points
    hipStream_t s2 = (n_unique && c->msm_fork) ? c->stream2 : s;
    if (s2 != s) {
        HIPCHK(c, hipEventRecord(c->fork_ev, s));
        HIPCHK(c, hipStreamWaitEvent(s2, c->fork_ev, 0));
    }
    const uint32_t nrec = n_gen_terms * (uint32_t)nbatch;
    const uint32_t nblk_p = (uint32_t)((nbatch + FB_BLOCK - 1) / FB_BLOCK);
    if (ct) {
        LAUNCH(c, s2, "fb_recode_ct", k_fb_recode_ct, (nrec + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, nrec, prm, (uint32_t)nbatch, n_gen_terms,
               (const uint32_t *)d_gen_scalars, d_digits);
        LAUNCH(c, s2, "fb_accum_ct", k_fb_accum_ct, nblk_p * nsplit, FB_BLOCK, prm, (uint32_t)nbatch, nblk_p, nsplit, npairs, d_ids, d_digits,
               c->d_table_ct, d_partial);
    } else {
        LAUNCH(c, s2, "fb_recode", k_fb_recode, (nrec + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, nrec, prm, (uint32_t)nbatch, n_gen_terms,
               (const uint32_t *)d_gen_scalars, d_digits, d_status);
        LAUNCH(c, s2, "fb_accum", k_fb_accum, nblk_p * nsplit, FB_BLOCK, prm, (uint32_t)nbatch, nblk_p, nsplit, npairs, d_ids, d_digits,
               c->d_table, d_partial);
    }
    ge_ext *d_red = nullptr;
    uint32_t nred = 0;
    enqueue_fb_reduce(c, s2, (uint32_t)nbatch, nsplit, d_partial, &d_red, &nred);
    if (s2 != s) HIPCHK(c, hipEventRecord(c->join_ev, s2));
    vb_dev d{};
    if (use_bucket) {   // many per-MSM points (the R1CS verifier's shape, r1cs/verifier.rs:459-491): bucket path
        bk_dev bd;
        bucket_bind(c, boff, bd);
        rc = bucket_upload_first(c, s, nbatch, nullptr, n_unique, bd);
        if (rc) return rc;
        const uint32_t total = (uint32_t)(nbatch * n_unique);   // (the fused chain left above: this is bucket.h's)
        LAUNCH(c, s, "bk_prepare", k_bk_prepare, (total + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, total, (uint32_t)nbatch, bd.msm_first,
               (const uint32_t *)d_uniq_scalars, (const uint32_t *)d_uniq_points, bd.pts, bd.rwords, d_status, bkp);
        rc = enqueue_bucket_tail(c, s, bkp, nbatch, total, n_unique, bd);
        if (rc) return rc;
        d.hq = bd.hq;
    } else if (n_unique) {
        rc = enqueue_vb_uniform(c, s, nbatch, n_unique, off, (const uint32_t *)d_uniq_scalars, (const uint32_t *)d_uniq_points, d_status, d);
        if (rc) return rc;
    }
    if (s2 != s) HIPCHK(c, hipStreamWaitEvent(s, c->join_ev, 0));
    LAUNCH(c, s, "shared_finish", k_shared_finish, ((uint32_t)nbatch + 63) / 64, 64, (uint32_t)nbatch, nred, d.hq, n_unique ? 1 : 0,
           d_red, d_status, (uint32_t *)d_out, (uint8_t *)d_verdict, (uint8_t *)d_status_bytes);   // (status bytes ride along: one launch fewer)
    HIPCHK(c, hipGetLastError());
    return BPGPU_OK;
}

extern "C" int bpgpu_msm_batch_shared_dev(bpgpu_ctx *c, size_t n, size_t m, size_t nbatch, size_t n_unique, const void *d_gen_scalars,
                                          const void *d_uniq_scalars, const void *d_uniq_points, void *d_out, void *d_status,
                                          void *stream) {
    if (!c || (nbatch && (!d_gen_scalars || !d_out || !d_status))) return BPGPU_ERR_INVALID_ARG;
    if (n_unique && (!d_uniq_scalars || !d_uniq_points)) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    int rc = ctx_enter(c, s);
    if (rc) return rc;
    rc = msm_shared_dev_locked(c, n, m, nbatch, n_unique, d_gen_scalars, d_uniq_scalars, d_uniq_points, d_out, d_status, nullptr, s);
    const int rc2 = ctx_leave(c, s);
    return rc ? rc : rc2;
}

extern "C" int bpgpu_msm_batch_shared(bpgpu_ctx *c, size_t n, size_t m, size_t nbatch, size_t n_unique, const uint8_t *gen_scalars,
                                      const uint8_t *uniq_scalars, const uint8_t *uniq_points, uint8_t *out, uint8_t *status) {
    if (!c || (nbatch && (!gen_scalars || !out || !status))) return BPGPU_ERR_INVALID_ARG;
    if (n_unique && (!uniq_scalars || !uniq_points)) return BPGPU_ERR_INVALID_ARG;
    if (nbatch == 0) return BPGPU_OK;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t ng = (2 * n * m + 2) * nbatch * 32, nu = n_unique * nbatch * 32;
    const size_t sz_g = align_up(ng + 16), sz_u = align_up(nu + 16), sz_out = align_up(nbatch * 32), sz_st = align_up(nbatch);
    hipStream_t s = c->stream;
    int rc = ctx_enter(c, s);
    if (rc) return rc;
    rc = io_reserve(c, sz_g + 2 * sz_u + sz_out + sz_st);
    if (rc) return rc;
    char *h = nullptr;
    rc = pin_alloc(c, s, sz_g + 2 * sz_u + sz_out + sz_st, &h);
    if (rc) return rc;
    char *d_g = c->io_dev, *d_us = d_g + sz_g, *d_up = d_us + sz_u, *d_o = d_up + sz_u;
    char *h_o = h + sz_g + 2 * sz_u;
    memcpy(h, gen_scalars, ng);
    if (nu) {
        memcpy(h + sz_g, uniq_scalars, nu);
        memcpy(h + sz_g + sz_u, uniq_points, nu);
    }
    HIPCHK(c, hipMemcpyAsync(d_g, h, sz_g + 2 * sz_u, hipMemcpyHostToDevice, s));
    rc = msm_shared_dev_locked(c, n, m, nbatch, n_unique, d_g, d_us, d_up, d_o, d_o + sz_out, nullptr, s);
    if (!rc && hipMemcpyAsync(h_o, d_o, sz_out + sz_st, hipMemcpyDeviceToHost, s) != hipSuccess) rc = fail(c, BPGPU_ERR_HIP, "D2H copy failed");
    const int rc2 = ctx_leave(c, s), rc3 = host_wait(c, s);
    if (rc || rc2 || rc3) return rc ? rc : (rc2 ? rc2 : rc3);
    memcpy(out, h_o, nbatch * 32);
    memcpy(status, h_o + sz_out, nbatch);
    return BPGPU_OK;
}

// ============================================================================
// generator derivation on the device (BulletproofGens::new, PedersenGens::default)
// ============================================================================
static void host_shake_chain(std::vector<uint8_t> &out, uint8_t tag, uint32_t party, size_t count) {
    // GeneratorsChain::new(label): SHAKE256("GeneratorsChain" || tag || u32le(party))  (generators.rs:64-72,186-201)
    uint32_t w[50];
    kstate st;
    st.w = w;
    st.stride = 1;
    sponge k;
    sponge_init(k, st, BP_SHAKE256_RATE);
    const uint8_t dom[15] = {'G', 'e', 'n', 'e', 'r', 'a', 't', 'o', 'r', 's', 'C', 'h', 'a', 'i', 'n'};
    const uint8_t label[5] = {tag, (uint8_t)party, (uint8_t)(party >> 8), (uint8_t)(party >> 16), (uint8_t)(party >> 24)};
    sponge_absorb(k, dom, 15);
    sponge_absorb(k, label, 5);
    sponge_finish(k, 0x1f);
    const size_t off = out.size();
    out.resize(off + 64 * count);
    sponge_squeeze(k, out.data() + off, (uint32_t)(64 * count));
}

static const uint8_t BASEPOINT_COMPRESSED[32] = {0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9,
                                                 0x61, 0xc5, 0x00, 0x51, 0x5f, 0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82,
                                                 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};

extern "C" int bpgpu_gens_create(bpgpu_ctx *c, size_t gens_capacity, size_t party_capacity) {
    if (!c || gens_capacity == 0 || party_capacity == 0) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t tot = gens_capacity * party_capacity;
    // uniform bytes: [B_blinding seed][G party 0..][H party 0..]
    std::vector<uint8_t> uni;
    {
        uint32_t w[50];
        kstate st;
        st.w = w;
        st.stride = 1;
        sponge k;
        sponge_init(k, st, BP_SHA3_512_RATE);   // hash_from_bytes::<Sha3_512>(compress(B))  (generators.rs:48)
        sponge_absorb(k, BASEPOINT_COMPRESSED, 32);
        sponge_finish(k, 0x06);
        uni.resize(64);
        sponge_squeeze(k, uni.data(), 64);
    }
    for (size_t p = 0; p < party_capacity; p++) host_shake_chain(uni, 'G', (uint32_t)p, gens_capacity);
    for (size_t p = 0; p < party_capacity; p++) host_shake_chain(uni, 'H', (uint32_t)p, gens_capacity);
    const uint32_t cnt = (uint32_t)(1 + 2 * tot);
    uint32_t *d_uni = nullptr, *d_out = nullptr;
    HIPCHK(c, hipMalloc((void **)&d_uni, uni.size()));
    HIPCHK(c, hipMalloc((void **)&d_out, (size_t)cnt * 32));
    HIPCHK(c, hipMemcpyAsync(d_uni, uni.data(), uni.size(), hipMemcpyHostToDevice, c->stream));
    LAUNCH(c, c->stream, "from_uniform", k_from_uniform, (cnt + 63) / 64, 64, cnt, d_uni, d_out);
    std::vector<uint8_t> enc((size_t)cnt * 32);
    HIPCHK(c, hipMemcpyAsync(enc.data(), d_out, enc.size(), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipFree(d_uni);
    hipFree(d_out);
    return load_gens_locked(c, gens_capacity, party_capacity, enc.data() + 32, enc.data() + 32 + tot * 32, BASEPOINT_COMPRESSED,
                            enc.data());
}

// ============================================================================
// range-proof verification
// ============================================================================

// ---- Merlin transcripts on the host (Transcript::new / append_message / challenge_bytes on the 208-byte state) ----
static void ts_to_strobe(strobe &t, uint32_t w[50], const uint8_t *state) {
    memcpy(w, state, 200);
    t.st.w = w;
    t.st.stride = 1;
    t.pos = state[200];
    t.pos_begin = state[201];
    t.cur_flags = state[202];
}
static void ts_from_strobe(uint8_t *state, const strobe &t) {
    memset(state, 0, BPGPU_TRANSCRIPT_BYTES);
    memcpy(state, t.st.w, 200);
    state[200] = (uint8_t)t.pos;
    state[201] = (uint8_t)t.pos_begin;
    state[202] = (uint8_t)t.cur_flags;
}
static bool ts_state_ok(const uint8_t *state) { return state[200] < BP_STROBE_R && state[201] <= BP_STROBE_R; }
static const uint8_t DOM_SEP[7] = {'d', 'o', 'm', '-', 's', 'e', 'p'};

extern "C" int bpgpu_transcript_new(const uint8_t *label, size_t label_len, uint8_t state[BPGPU_TRANSCRIPT_BYTES]) {
    if (!state || (label_len && !label) || label_len > 0xffffffffu) return BPGPU_ERR_INVALID_ARG;
    uint32_t w[50];
    kstate st;
    st.w = w;
    st.stride = 1;
    strobe t;
    merlin_strobe_init(t, st);
    merlin_append_message(t, DOM_SEP, 7, label, (uint32_t)label_len);   // Transcript::new: append_message(b"dom-sep", label)
    ts_from_strobe(state, t);
    return BPGPU_OK;
}
extern "C" int bpgpu_transcript_append_message(uint8_t state[BPGPU_TRANSCRIPT_BYTES], const uint8_t *label, size_t label_len,
                                               const uint8_t *msg, size_t msg_len) {
    if (!state || (label_len && !label) || (msg_len && !msg) || label_len > 0xffffffffu || msg_len > 0xffffffffu || !ts_state_ok(state))
        return BPGPU_ERR_INVALID_ARG;
    uint32_t w[50];
    strobe t;
    ts_to_strobe(t, w, state);
    merlin_append_message(t, label, (uint32_t)label_len, msg, (uint32_t)msg_len);
    ts_from_strobe(state, t);
    return BPGPU_OK;
}
extern "C" int bpgpu_transcript_challenge_bytes(uint8_t state[BPGPU_TRANSCRIPT_BYTES], const uint8_t *label, size_t label_len,
                                                uint8_t *out, size_t out_len) {
    if (!state || (label_len && !label) || (out_len && !out) || label_len > 0xffffffffu || out_len > 0xffffffffu || !ts_state_ok(state))
        return BPGPU_ERR_INVALID_ARG;
    uint32_t w[50];
    strobe t;
    ts_to_strobe(t, w, state);
    merlin_challenge_bytes(t, label, (uint32_t)label_len, out, (uint32_t)out_len);
    ts_from_strobe(state, t);
    return BPGPU_OK;
}

// thread_rng() stand-in of the provers and of everything else that needs host-side random bytes: a per-thread ChaCha20 generator
// keyed and re-keyed from the OS CSPRNG (hostrng.h).  (The verification chains need 32 bytes each: rp_shape::seed.)
static int os_random(bpgpu_ctx *c, char *dst, size_t bytes) {
    if (!bp::fast_random((uint8_t *)dst, bytes)) return fail(c, BPGPU_ERR_HIP, "getrandom failed");
    return BPGPU_OK;
}
// the caller's transcript, not yet domain-separated: the kernel applies rangeproof_domain_sep(n, m) per proof
static void strobe_init_from_state(rp_strobe_init &init, const uint8_t *state) {
    memcpy(init.w, state, 200);
    init.pos = state[200];
    init.pos_begin = state[201];
    init.cur_flags = state[202];
}

static void make_strobe_init(rp_strobe_init &init, const uint8_t *label, size_t label_len, uint64_t n, uint64_t m) {
    // Transcript::new(label) followed by rangeproof_domain_sep(n, m) (transcript.rs:44-48): identical for
    // every proof of the batch, so it is replayed once here and the kernel starts from this state
    kstate st;
    st.w = init.w;
    st.stride = 1;
    strobe t;
    merlin_strobe_init(t, st);
    const uint8_t dom[7] = {'d', 'o', 'm', '-', 's', 'e', 'p'};
    const uint8_t rp[13] = {'r', 'a', 'n', 'g', 'e', 'p', 'r', 'o', 'o', 'f', ' ', 'v', '1'};
    const uint8_t ln[1] = {'n'}, lm[1] = {'m'};
    merlin_append_message(t, dom, 7, label, (uint32_t)label_len);
    merlin_append_message(t, dom, 7, rp, 13);
    merlin_append_u64(t, ln, 1, n);
    merlin_append_u64(t, lm, 1, m);
    init.pos = t.pos;
    init.pos_begin = t.pos_begin;
    init.cur_flags = t.cur_flags;
}

// the transcript script of a shape and start position (rp_script.h), cached on the device per context.
// The cache is bounded (callers whose transcripts sit at many different STROBE positions): beyond 64 entries the least recently
// used one gives its device block to the newcomer.  Uploads come from the pinned staging buffer on the call's stream, so the
// overwrite is ordered behind every earlier launch of this context that read the old script (ctx_enter orders the streams);
// nothing is freed and nothing synchronises the device while other lanes are running.
static int script_for(bpgpu_ctx *c, hipStream_t s, uint32_t n, uint32_t m, uint32_t k, const rp_strobe_init &init, bool domsep, const rp_script_hdr **out) {
    std::vector<uint32_t> key = {n, m, k, init.pos, init.pos_begin, init.cur_flags, domsep ? 1u : 0u};
    auto it = c->script_cache.find(key);
    if (it == c->script_cache.end()) {
        const std::vector<uint32_t> img = rp_script_build(n, m, k, init.pos, init.pos_begin, init.cur_flags, domsep);
        const size_t bytes = img.size() * 4;
        bpgpu_ctx::script_ent ent;
        if (c->script_cache.size() >= 64) {
            auto lru = c->script_cache.begin();
            for (auto jt = c->script_cache.begin(); jt != c->script_cache.end(); ++jt)
                if (jt->second.last_use < lru->second.last_use) lru = jt;
            ent = lru->second;
            c->script_cache.erase(lru);
            if (ent.cap < bytes) {   // too small for the newcomer: parked until the context goes away (a few KB; bounded by the shapes a process uses)
                c->script_retired.push_back(ent.mem);
                ent.mem = nullptr;
                ent.cap = 0;
            }
        }
        if (!ent.mem) {
            const size_t cap = bytes < 8192 ? 8192 : bytes;
            HIPCHK(c, hipMalloc((void **)&ent.mem, cap));
            ent.cap = cap;
        }
        char *h = nullptr;
        const int rc = pin_alloc(c, s, bytes, &h);
        if (rc) {
            c->script_retired.push_back(ent.mem);
            return rc;
        }
        memcpy(h, img.data(), bytes);
        HIPCHK(c, hipMemcpyAsync(ent.mem, h, bytes, hipMemcpyHostToDevice, s));
        it = c->script_cache.emplace(key, ent).first;
    }
    it->second.last_use = ++c->script_tick;
    *out = (const rp_script_hdr *)it->second.mem;
    return BPGPU_OK;
}

// Where a call's transcripts start: Transcript::new(label) (label), or one caller-supplied state for the whole batch
// (shared_ts, host, 208 bytes), or one state per proof (d_ts_in, device); d_ts_out (optional): the advanced states.
struct rp_transcripts {
    const uint8_t *label = nullptr;
    size_t label_len = 0;
    const uint8_t *shared_ts = nullptr;
    const void *d_ts_in = nullptr;
    void *d_ts_out = nullptr;
    // with d_ts_in: every state of the batch sits at this STROBE position (the pool's combining queue groups its requests that
    // way), so the per-shape script replaces the byte-wise replay although the sponge words differ from proof to proof
    bool ts_uniform = false;
    uint32_t u_pos = 0, u_pos_begin = 0, u_flags = 0;
    // coalesced launches (h_segs): the label of every item, all `label_len` bytes long (`label` is item 0's).  Items whose labels differ
    // start from their own state (rp_seg::init_w); positions depend on lengths only, so the chain's script serves them all
    const uint8_t *const *seg_labels = nullptr;
};

// ---- the forms a per-proof launch chain takes -----------------------------------------------------------------------------------
// One decision table for what used to be scattered conditions.  Inputs: the context options (0 / -1 = automatic), what the pool
// said about the chain (busy_hint: 1 = other chains run beside it, 0 = it is alone on the device, -1 = nobody said), its width.
//   WIDE     (>= 2048 proofs, or split_stage3 = 1): the window sums are a launch of their own -- their register allocation instead of
//            the generator-exponent role's 252 (+1.5 ... 3 %, profiles/r03/ab_split_stage3.txt).
//   WAVE     one wavefront per Horner chain (horner_wave.h): a small batch alone is a latency matter -- up to 256 proofs always (one call
//            0.85 -> 0.62 ms), up to 2304 when the pool knows the chain is alone (one host call of 1024 / 2048 / 4096 proofs, slices of
//            <= 2048: 1.37 / 2.14 / 2.84 M/s against 1.12 / 1.93 / 2.66 with quads; at 6144, slices of 3072, quads are ahead again).
//   ASIDE    one LANE per Horner chain, as its own launch on the context's second stream right after the window sums (half the quad
//            form's instructions, twice its latency: 0.76 ms that need other work beside them).  Wide chains when other chains run beside
//            them, or from 8192 proofs on a context nobody told anything: +2.6 ... 3.6 % steady state, +2 ... 4.6 % on 20 x 1024 bursts;
//            a lone host call of 2048 / 4096 proofs loses 22 / 16 % with it (profiles/r03/ab_horner_one_lane.txt, host_call_horner_forms.txt).
//   RADIX32  with ASIDE only, option per_proof_radix = 32: 16-entry tables, 51 windows (+1 % steady, -4.5 % bursts: off by default).
//   A_OUTSIDE with ASIDE only: A, whose coefficient is 1, is added after the chain instead of going through a table and 64 windows.
// Otherwise: four lanes per Horner chain inside launch 4 (horner_quad.h).  Batch-combined calls and shape verdicts take none of these.
enum { RPC_WIDE = 1, RPC_WAVE = 2, RPC_ASIDE = 4, RPC_RADIX32 = 8, RPC_A_OUTSIDE = 16 };
static uint32_t rp_chain_forms(uint32_t horner_lanes, int split_stage3, int busy_hint, uint32_t per_proof_radix, int a_outside, size_t nbatch,
                               bool combined_or_verdict_only, bool on_second_stream) {
    const bool wide = !combined_or_verdict_only && (split_stage3 == 1 || (split_stage3 < 0 && nbatch >= 2048));
    const bool throughput = busy_hint > 0 || (busy_hint < 0 && nbatch >= 8192);
    const bool wave = horner_lanes == 64 || (horner_lanes == 0 && (nbatch <= 256 || (busy_hint == 0 && nbatch <= 2304)));
    const bool aside = wide && !wave && !on_second_stream && (horner_lanes == 1 || (horner_lanes == 0 && throughput));
    uint32_t f = 0;
    if (wide) f |= RPC_WIDE;
    if (wave) f |= RPC_WAVE;
    if (aside) f |= RPC_ASIDE;
    if (aside && per_proof_radix == 32) f |= RPC_RADIX32;
    if (aside && a_outside) f |= RPC_A_OUTSIDE;
    return f;
}
// (for tests/test_abi_and_host.py: the table is plain host logic)
extern "C" uint32_t bpgpu_internal_chain_forms(int64_t horner_lanes, int64_t split_stage3, int64_t busy_hint, int64_t per_proof_radix, int64_t a_outside,
                                               uint64_t nbatch, int combined_or_verdict_only, int on_second_stream) {
    return rp_chain_forms((uint32_t)horner_lanes, (int)split_stage3, (int)busy_hint, (uint32_t)per_proof_radix, (int)a_outside, (size_t)nbatch,
                          combined_or_verdict_only != 0, on_second_stream != 0);
}

// (tests: pin the key the library would draw for a chain's device-expanded randomness, so that a call WITHOUT rng / weight buffers can be
// compared with one that passes the expansion's bytes explicitly; key = nullptr: back to the generator)
extern "C" int bpgpu_internal_set_chain_seed(bpgpu_ctx *c, const uint8_t *key32) {
    if (!c) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    c->test_seed_set = key32 != nullptr;
    if (key32) memcpy(c->test_seed, key32, 32);
    return BPGPU_OK;
}

static int rp_verify_dev_locked(bpgpu_ctx *c, size_t n, size_t m, size_t nbatch, const void *d_proofs, size_t proof_len,
                                const void *d_commitments, const rp_transcripts &tr, const void *d_rng64, void *d_verdict,
                                void *d_msm_out, hipStream_t s, bool rlc = false, const void *d_weights64 = nullptr,
                                void *d_batch_out = nullptr, const rp_seg *h_segs = nullptr, uint32_t nseg = 0, bool dev_call = false,
                                bool reserve_only = false) {
    // reserve_only: size the arena, the status words and the work decomposition for a chain of `nbatch` proofs and return --
    // a lane of the pool's combining queue sees chains of every width and should not grow its buffers (a device-wide
    // synchronisation each time) on the way up
    // h_segs (bpgpu_pool_*, coalesced launch): the nbatch proofs are the concatenation of nseg submitted items, each with its
    // own input / output buffers (d_proofs, d_commitments, d_verdict, d_msm_out unused; d_msm_out non-null = some item wants encodings)
    // rlc: batch-combination mode (bpgpu_rangeproof_verify_rlc[_dev]); d_msm_out is unused then, d_batch_out
    // (33 bytes, optional) receives the batch verdict and the encoding of the combined point
    if (nbatch == 0) {
        if (rlc && d_batch_out) HIPCHK(c, hipMemsetAsync(d_batch_out, 0, 33, s));
        return BPGPU_OK;
    }
    if (!c->d_table) return fail(c, BPGPU_ERR_NO_GENS, "generators not loaded");
    if (nbatch > 0x7fffffffu / 64) return fail(c, BPGPU_ERR_INVALID_ARG, "batch too large");
    // ---- length-only part of RangeProof::from_bytes / InnerProductProof::from_bytes (mod.rs:505-510, ipp.rs:374-388)
    uint32_t all_verdict = 0;
    size_t k = 0;
    if (proof_len % 32 != 0 || proof_len < 7 * 32) all_verdict = BPGPU_VERDICT_FORMAT_ERROR;
    else {
        const size_t ne = (proof_len - 7 * 32) / 32;
        if (ne < 2 || (ne - 2) % 2 != 0) all_verdict = BPGPU_VERDICT_FORMAT_ERROR;
        else {
            k = (ne - 2) / 2;
            if (k >= 32) all_verdict = BPGPU_VERDICT_FORMAT_ERROR;
        }
    }
    if (tr.shared_ts && !ts_state_ok(tr.shared_ts)) return fail(c, BPGPU_ERR_INVALID_ARG, "malformed transcript state");
    if (h_segs && (all_verdict || tr.d_ts_in || tr.d_ts_out || tr.shared_ts || (rlc && d_weights64)))   // (the pool sends such items through the ordinary entry point)
        return fail(c, BPGPU_ERR_INVALID_ARG, "coalesced launches take well-formed shapes and a label only (batch-combined ones: library-drawn weights)");
    if (all_verdict) {   // every proof of the batch has the same malformed length
        if (tr.d_ts_out) {   // FormatError leaves the caller's transcript untouched
            if (tr.d_ts_in) HIPCHK(c, hipMemcpyAsync(tr.d_ts_out, tr.d_ts_in, nbatch * BPGPU_TRANSCRIPT_BYTES, hipMemcpyDeviceToDevice, s));
            else {
                char *h = nullptr;
                int rcp = pin_alloc(c, s, nbatch * BPGPU_TRANSCRIPT_BYTES, &h);
                if (rcp) return rcp;
                uint8_t st0[BPGPU_TRANSCRIPT_BYTES];
                if (tr.shared_ts) memcpy(st0, tr.shared_ts, BPGPU_TRANSCRIPT_BYTES);
                else bpgpu_transcript_new(tr.label, tr.label_len, st0);
                for (size_t b = 0; b < nbatch; b++) memcpy(h + b * BPGPU_TRANSCRIPT_BYTES, st0, BPGPU_TRANSCRIPT_BYTES);
                HIPCHK(c, hipMemcpyAsync(tr.d_ts_out, h, nbatch * BPGPU_TRANSCRIPT_BYTES, hipMemcpyHostToDevice, s));
            }
        }
        HIPCHK(c, hipMemsetAsync(d_verdict, (int)all_verdict, nbatch, s));
        if (d_msm_out) HIPCHK(c, hipMemsetAsync(d_msm_out, 0, nbatch * 32, s));
        if (rlc && d_batch_out) HIPCHK(c, hipMemsetAsync(d_batch_out, 0, 33, s));   // nothing to combine
        return BPGPU_OK;
    }
    // ---- parameter checks of verify_multiple_with_rng (mod.rs:358-366), reported per proof AFTER its format check
    uint32_t shape_verdict = 0;
    if (!(n == 8 || n == 16 || n == 32 || n == 64)) shape_verdict = BPGPU_VERDICT_INVALID_BITSIZE;
    else if (c->gens_capacity < n || c->party_capacity < m) shape_verdict = BPGPU_VERDICT_INVALID_GENERATORS_LENGTH;
    else if (m == 0 || n * m != ((size_t)1 << k)) shape_verdict = BPGPU_VERDICT_VERIFICATION_ERROR;   // ipp.rs:209
    if (!shape_verdict && k > BP_RP_MAX_K) return fail(c, BPGPU_ERR_INVALID_ARG, "n*m > 2^%d not supported", BP_RP_MAX_K);
    if (h_segs && shape_verdict) return fail(c, BPGPU_ERR_INVALID_ARG, "coalesced launches take well-formed shapes only");

    rp_shape sh;
    sh.n = (uint32_t)n;
    sh.m = (uint32_t)m;
    sh.nm = (uint32_t)(n * m);
    sh.k = (uint32_t)k;
    sh.U = (uint32_t)(4 + 2 * k + m);
    sh.proof_len = (uint32_t)proof_len;
    sh.nproofs = (uint32_t)nbatch;
    sh.shape_verdict = shape_verdict;
    uint32_t lg_m = 0;
    while (((size_t)1 << lg_m) < m) lg_m++;
    // which window table the generator terms walk: the secondary one when the shape fits it (bpgpu_gens_add_shape), else the primary
    const bool use_sec = c->sec.d_table && !shape_verdict && n <= c->sec.n && m <= c->sec.m;
    const fb_params prm = use_sec ? c->sec.prm : c->prm;
    const fb_entry *const gen_table = use_sec ? c->sec.d_table : c->d_table;
    const uint32_t n_gen_terms = shape_verdict ? 2 : (uint32_t)(2 * n * m + 2);
    const uint32_t npairs = n_gen_terms * prm.nwin;
    const rp_fields fl = rp_field_layout(sh.k, sh.m);
    uint32_t *d_ids = nullptr;
    int rc;
    if (!shape_verdict) {
        rc = use_sec ? gen_ids_for_secondary(c, n, m, &d_ids) : gen_ids_for(c, n, m, &d_ids);
        if (rc) return rc;
    }
    // which forms this chain takes (rp_chain_forms above: the decision table, with the measurements behind every line)
    const uint32_t forms = rp_chain_forms(c->horner_lanes, c->split_stage3, c->busy_hint, c->vb_radix, c->a_outside, nbatch, rlc || shape_verdict != 0,
                                          s == c->stream2);
    const bool wide = forms & RPC_WIDE, wave = forms & RPC_WAVE, aside = forms & RPC_ASIDE, r5 = forms & RPC_RADIX32, a_out = forms & RPC_A_OUTSIDE;
    uint32_t nsplit = pick_splits(c, nbatch, npairs, aside);
    // narrow chains in the wavefront-per-chain form: the walk with lane = split, a multiple of 64 splits per proof (rp_walk_narrow)
    const bool narrow_walk = wave && !wide && !rlc && nbatch <= 256 && c->narrow_walk;
    // ... and for VERY narrow ones every per-proof point gets a second table, of its 2^128 multiple, built beside the transcript by a wavefront
    // of its own (k_rp_stage1_coop's third role): the Horner chain has 32 windows, and the walk twice the splits to end with it
    const bool will_script = (!tr.d_ts_in || tr.ts_uniform) && !shape_verdict && !c->no_script;
    const bool narrow_hi = narrow_walk && will_script && c->transcript_coop && c->coop_split && !shape_verdict && (int64_t)nbatch <= (int64_t)c->narrow_hi_max;
    const uint32_t hi_levels = !narrow_hi ? 1u : ((int64_t)nbatch <= (int64_t)c->narrow_hi4_max ? 4u : 2u);   // (the very narrowest: three more tables per point, a 16-window chain)
    if (narrow_walk) {
        if (narrow_hi && nsplit < 2 * FB_BLOCK) nsplit = 2 * FB_BLOCK;
        if (hi_levels == 4 && nsplit < 4 * FB_BLOCK) nsplit = 4 * FB_BLOCK;
        nsplit = (nsplit + FB_BLOCK - 1) / FB_BLOCK * FB_BLOCK;
        while (nsplit > FB_BLOCK && npairs / nsplit < 4) nsplit -= FB_BLOCK;
    }
    // thread / element counts are 32-bit in the kernels: refuse what does not fit
    if ((uint64_t)n_gen_terms * nbatch > 0x7fffffffull || (uint64_t)nbatch * sh.U > 0x7fffffffull / 64 ||
        (uint64_t)nbatch * ((sh.U + BP_VB_CHUNK - 1) / BP_VB_CHUNK) * 64 > 0x7fffffffull || (uint64_t)(sh.nm / 4 + 1) * nbatch > 0x7fffffffull ||
        (uint64_t)((nbatch + FB_BLOCK - 1) / FB_BLOCK) * nsplit > 0x7fffffffull)
        return fail(c, BPGPU_ERR_INVALID_ARG, "batch too large for this shape");
    // batch combination with enough per-proof terms: ONE bucket MSM (bucket.h) over all proofs' weighted terms
    const size_t rlc_terms = (size_t)nbatch * sh.U;
    const bk_params bkp = bk_make(pick_bucket_c(rlc_terms));
    const bool rlc_bucket = rlc && !shape_verdict && rlc_terms >= (c->bucket_min ? c->bucket_min : BK_RLC_MIN_TERMS) && bucket_fits(1, rlc_terms, bkp);
    // Narrow chains (one wavefront per Horner chain, which adds the chunks' window sums itself): a (chunk, window) lane adding all U table
    // entries one after the other is ~50 us of a one-proof chain's launch 3.  With chunks of 8 terms a lane adds 8 entries and the Horner
    // wavefront's lane w adds the U / 8 rows of its window.  (Option narrow_chunk.)
    uint32_t narrow_q = (uint32_t)c->narrow_chunk;
    if (narrow_q == 0) narrow_q = 8;   // (A/B at U = 17, profiles/r06/narrow_chain_ab.txt: 3 / 5 / 8 / 32 -> one call 0.456 / 0.457 / 0.448 / 0.464 ms; launch 3 is then bound by its basepoint-coefficient role)
    narrow_q = narrow_q < 2 ? 2u : (narrow_q > BP_VB_CHUNK ? (uint32_t)BP_VB_CHUNK : narrow_q);
    const bool narrow_ok = !rlc && !shape_verdict && (c->horner_lanes == 0 || c->horner_lanes == 64);   // (narrow chains of this context take the wavefront-per-chain form)
    const uint32_t vb_chunk_sz = (wave && !wide && narrow_ok && nbatch <= 256) ? narrow_q : (uint32_t)BP_VB_CHUNK;
    sh.radix5 = r5 ? 1u : 0u;
    sh.a_outside = a_out ? 1u : 0u;
    sh.defer_emit = c->coop_defer_emit ? 1u : 0u;
    arena_plan ap;
    size_t off[7], boff[12];
    if (rlc_bucket) plan_bucket(ap, 1, rlc_terms, bkp, boff);
    else plan_vb_uniform(ap, nbatch, shape_verdict ? 0 : sh.U, off, narrow_hi ? 8 * hi_levels : (r5 ? 16 : 8), vb_chunk_sz);
    const size_t off_digits = ap.add((size_t)npairs * nbatch * sizeof(fb_digit) + 16);
    const size_t off_partial = ap.add((size_t)2 * nsplit * nbatch * sizeof(ge_ext) + 16);
    const size_t off_fields = ap.add((size_t)fl.count * nbatch * BP_RP_REC * 4 + 16);
    const size_t off_mv = ap.add(nbatch);
    const size_t off_segs = (h_segs && nseg > RP_SEG_INLINE) ? ap.add((size_t)nseg * sizeof(rp_seg)) : 0;
    // distinct labels among the items of a coalesced launch (usually one)
    std::vector<uint32_t> seg_label_of;
    std::vector<const uint8_t *> distinct_labels;
    if (h_segs && tr.seg_labels) {
        seg_label_of.resize(nseg);
        for (uint32_t i = 0; i < nseg; i++) {
            uint32_t j = 0;
            while (j < distinct_labels.size() && tr.label_len != 0 && memcmp(distinct_labels[j], tr.seg_labels[i], tr.label_len) != 0) j++;
            if (j == distinct_labels.size()) distinct_labels.push_back(tr.seg_labels[i]);
            seg_label_of[i] = j;
        }
    }
    const bool own_inits = distinct_labels.size() > 1;
    const size_t off_inits = own_inits ? ap.add(distinct_labels.size() * 256) : 0;
    // batch-combination mode: weights, coefficient accumulators, the column-sum reduction tree, and a batch-of-one
    // table walk (digits, partial sums, result)
    const uint32_t nsplit1 = rlc ? pick_splits(c, 1, npairs) : 0;
    const size_t n_rows0 = shape_verdict ? 0 : nbatch * ((sh.U + BP_VB_CHUNK - 1) / BP_VB_CHUNK);
    const size_t off_acc = rlc ? ap.add((size_t)n_gen_terms * 10 * 8) : 0;
    const size_t off_tree = rlc ? ap.add((n_rows0 / 8 + 64) * 64 * sizeof(ge_ext)) : 0;
    const size_t off_dig1 = rlc ? ap.add((size_t)npairs * sizeof(fb_digit) + 16) : 0;
    const size_t off_part1 = rlc ? ap.add((size_t)2 * nsplit1 * sizeof(ge_ext) + 16) : 0;
    const size_t off_res1 = rlc ? ap.add(sizeof(ge_ext) + 64) : 0;   // Horner result, then 8 result words, verdict byte, chunk bounds
    size_t arena_need = ap.total;
    if (reserve_only && narrow_ok && nbatch > 256)   // (a lane sized for wide chains also sees narrow ones, whose window sums come in more, smaller chunks)
        arena_need += (size_t)256 * ((sh.U + narrow_q - 1) / narrow_q) * 64 * sizeof(ge_ext);
    rc = arena_reserve(c, arena_need);
    if (rc) return rc;
    char *a = c->arena;
    if (!c->fin_cnt) {
        HIPCHK(c, hipMalloc((void **)&c->fin_cnt, 256 * 4));
        HIPCHK(c, hipMemset(c->fin_cnt, 0, 256 * 4));
    }
    if (c->rp_status_cap < nbatch) {
        if (c->rp_status) HIPCHK(c, hipFree(c->rp_status));
        c->rp_status = nullptr;
        c->rp_status_cap = 0;
        HIPCHK(c, hipMalloc((void **)&c->rp_status, nbatch * 4));
        c->rp_status_cap = nbatch;
        c->rp_status_dirty = true;
    }
    if (reserve_only) {
        bpgpu_ctx::plan_view pv0;
        if (shape_verdict) return BPGPU_OK;
        if (narrow_ok && narrow_q != BP_VB_CHUNK) {
            rc = uniform_plan(c, nbatch < 256 ? nbatch : 256, sh.U, &pv0, narrow_q);
            if (rc) return rc;
        }
        return (nbatch > 256 || vb_chunk_sz == BP_VB_CHUNK) ? uniform_plan(c, nbatch, sh.U, &pv0) : BPGPU_OK;
    }
    uint32_t *d_status = c->rp_status;
    fb_digit *d_digits = (fb_digit *)(a + off_digits);
    ge_ext *d_partial = (ge_ext *)(a + off_partial);
    uint32_t *d_fields = (uint32_t *)(a + off_fields);
    uint8_t *d_mv = (uint8_t *)(a + off_mv);
    // randomness the caller did not bring -- the batching challenge's rng bytes (thread_rng() in verify_multiple, mod.rs:455-470), the
    // combination weights (unpredictable to the prover) -- is expanded inside launch 1 from ONE key per launch chain (rp_shape::seed,
    // rangeproof.h): drawn here from the calling thread's generator (hostrng.h), carried in the kernel's argument block
    const uint8_t *rng_ptr = (const uint8_t *)d_rng64;
    const uint8_t *wts_ptr = (const uint8_t *)d_weights64;
    if (!rng_ptr) sh.seeded |= RP_SEED_RNG;
    if (rlc && !wts_ptr) sh.seeded |= RP_SEED_WEIGHTS;
    if (sh.seeded) {
        if (c->test_seed_set) memcpy(sh.seed, c->test_seed, 32);
        else if (!bp::fast_random((uint8_t *)sh.seed, 32)) return fail(c, BPGPU_ERR_HIP, "getrandom failed");
    }
    rp_seg_tab segtab;
    memset(&segtab, 0, sizeof segtab);
    if (own_inits) {   // Transcript::new(label) + rangeproof_domain_sep(n, m) of every distinct label: 50 sponge words each, staged ahead of the chain
        char *h = nullptr;
        rc = pin_alloc(c, s, distinct_labels.size() * 256, &h);
        if (rc) return rc;
        for (size_t j = 0; j < distinct_labels.size(); j++) {
            rp_strobe_init ij;
            make_strobe_init(ij, distinct_labels[j], tr.label_len, n, m);
            memcpy(h + j * 256, ij.w, 200);
        }
        HIPCHK(c, hipMemcpyAsync(a + off_inits, h, distinct_labels.size() * 256, hipMemcpyHostToDevice, s));
    }
    auto seg_fix = [&](rp_seg *dst) {   // (the pool's table carries no start states: they live in this chain's arena)
        for (uint32_t i = 0; i < nseg; i++) dst[i].init_w = own_inits ? (const uint32_t *)(a + off_inits + (size_t)seg_label_of[i] * 256) : nullptr;
    };
    if (h_segs && nseg <= RP_SEG_INLINE) {   // travels in the kernels' argument blocks
        segtab.n = nseg;
        memcpy(segtab.in, h_segs, (size_t)nseg * sizeof(rp_seg));
        seg_fix(segtab.in);
    } else if (h_segs) {
        char *h = nullptr;
        rc = pin_alloc(c, s, (size_t)nseg * sizeof(rp_seg), &h);
        if (rc) return rc;
        memcpy(h, h_segs, (size_t)nseg * sizeof(rp_seg));
        seg_fix((rp_seg *)h);
        HIPCHK(c, hipMemcpyAsync(a + off_segs, h, (size_t)nseg * sizeof(rp_seg), hipMemcpyHostToDevice, s));
        segtab.n = nseg;
        segtab.ext = (const rp_seg *)(a + off_segs);
    }
    rp_strobe_init init;
    uint32_t ts_flags = 0;
    if (tr.d_ts_in || tr.shared_ts) {
        ts_flags = BP_TS_DOMSEP;
        if (tr.shared_ts) strobe_init_from_state(init, tr.shared_ts);
        else {
            memset(&init, 0, sizeof init);
            if (tr.ts_uniform) {
                init.pos = tr.u_pos;
                init.pos_begin = tr.u_pos_begin;
                init.cur_flags = tr.u_flags;
            }
        }
    } else if (tr.d_ts_out) {   // label + states wanted back: start every proof from Transcript::new(label) and replay the domain separator on the device
        uint8_t st0[BPGPU_TRANSCRIPT_BYTES];
        bpgpu_transcript_new(tr.label, tr.label_len, st0);
        strobe_init_from_state(init, st0);
        ts_flags = BP_TS_DOMSEP;
    } else {
        make_strobe_init(init, tr.label, tr.label_len, n, m);
    }
    const rp_script_hdr *d_script = nullptr;
    if ((!tr.d_ts_in || tr.ts_uniform) && !shape_verdict && !c->no_script) {   // every proof starts at the position of `init`: the per-shape script replaces the byte-wise replay
        rc = script_for(c, s, sh.n, sh.m, sh.k, init, (ts_flags & BP_TS_DOMSEP) != 0, &d_script);
        if (rc) return rc;
    }
    if (c->pin_off > c->pin_marked && dev_call) {   // rng / weights / segment table / a new transcript script staged above, nothing else will be
        rc = pin_mark(c, s);
        if (rc) return rc;
    }
    if (c->rp_status_dirty) {   // (a chain whose enqueue failed half way: its status words and arrival counters were not handed back clean)
        HIPCHK(c, hipMemsetAsync(d_status, 0, c->rp_status_cap * 4, s));
        HIPCHK(c, hipMemsetAsync(c->fin_cnt, 0, 256 * 4, s));
    }
    c->rp_status_dirty = true;   // until the kernel that resets the words has been enqueued
    const uint32_t nb32 = (uint32_t)nbatch;
    // the proof-specific ("variable-base") terms: decomposition into chunks of 32 is cached per (batch, U)
    vb_dev d{};
    bk_dev bd{};
    if (rlc_bucket) bucket_bind(c, boff, bd);
    bpgpu_ctx::plan_view pv, *pd = &pv;
    if (!shape_verdict && !rlc_bucket) {
        rc = uniform_plan(c, nbatch, sh.U, &pv, vb_chunk_sz);
        if (rc) return rc;
        vb_bind(c, off, d);
        d.chunks = (vb_chunk *)pv.mem;
        d.chunk_first = (uint32_t *)(pv.mem + pv.o1);
        d.hq = (ge_ext *)((char *)d.colq16 + (size_t)nbatch * 64 * sizeof(ge_cached));
    }
    // Horner layout: quads (16 chains per wavefront, least total work) unless the caller asked for the
    // wavefront-per-chain variant (lowest latency of a single small batch)
    const bool quad = !wave;                          // column sums as cached points (both the quad and the one-lane chain read them)
    const bool one_lane = c->horner_lanes == 1;
    ge_cached *d_colc = quad ? (ge_cached *)d.colq16 : nullptr;
    const uint32_t n_tr = (nb32 + RP_BLOCK - 1) / RP_BLOCK;
    const uint32_t n_pt = shape_verdict ? 0 : (nb32 * sh.U + RP_BLOCK - 1) / RP_BLOCK;
    const bool coop = d_script && c->transcript_coop && nbatch <= 256;
    sh.coop_split = (coop && c->coop_split && !rlc && !wide && !shape_verdict && sh.k < 32) ? 1u : 0u;   // (launch 3 then carries the basepoint-coefficient role)
    const bool hi = narrow_hi && coop && sh.coop_split;
    sh.narrow_hi = hi ? hi_levels : 0u;
    ge_cached *tab_hi = hi ? d.tab + (size_t)8 * nb32 * sh.U : (ge_cached *)nullptr;
    if (coop)   // narrow chain: 32 lanes per proof for the permutations (two proofs per workgroup)
        LAUNCH(c, s, "rp_stage1", k_rp_stage1_coop, (nb32 + 1) / 2 + n_pt + (hi ? (hi_levels - 1) * nb32 * sh.U : 0u), RP_BLOCK, sh, init, (nb32 + 1) / 2, (const uint8_t *)d_proofs,
           (const uint8_t *)d_commitments, rng_ptr, d_fields, d.tab, d_status, prm, lg_m, rlc_bucket ? bd.rwords : d.recoded, d_digits,
           rlc ? wts_ptr : (const uint8_t *)nullptr, (const uint32_t *)tr.d_ts_in, (uint32_t *)tr.d_ts_out,
           rlc_bucket ? bd.pts : (fb_entry *)nullptr, rlc_bucket ? bkp.c : 0u, segtab, d_script, n_pt, tab_hi);
    else if (d_script)
        LAUNCH(c, s, "rp_stage1", k_rp_stage1<true>, n_tr + n_pt, RP_BLOCK, sh, init, n_tr, (const uint8_t *)d_proofs,
           (const uint8_t *)d_commitments, rng_ptr, d_fields, d.tab, d_status, prm, lg_m, rlc_bucket ? bd.rwords : d.recoded, d_digits,
           rlc ? wts_ptr : (const uint8_t *)nullptr, ts_flags, (const uint32_t *)tr.d_ts_in, (uint32_t *)tr.d_ts_out,
           rlc_bucket ? bd.pts : (fb_entry *)nullptr, rlc_bucket ? bkp.c : 0u, segtab, d_script);
    else
        LAUNCH(c, s, "rp_stage1", k_rp_stage1<false>, n_tr + n_pt, RP_BLOCK, sh, init, n_tr, (const uint8_t *)d_proofs,
           (const uint8_t *)d_commitments, rng_ptr, d_fields, d.tab, d_status, prm, lg_m, rlc_bucket ? bd.rwords : d.recoded, d_digits,
           rlc ? wts_ptr : (const uint8_t *)nullptr, ts_flags, (const uint32_t *)tr.d_ts_in, (uint32_t *)tr.d_ts_out,
           rlc_bucket ? bd.pts : (fb_entry *)nullptr, rlc_bucket ? bkp.c : 0u, segtab, d_script);
    if (shape_verdict) {
        HIPCHK(c, hipMemsetAsync(d_mv, 1, nbatch, s));
        LAUNCH(c, s, "rp_verdict", k_rp_verdict, (nb32 + 63) / 64, 64, nb32, d_status, d_mv, (uint8_t *)d_verdict);
        if (d_msm_out) HIPCHK(c, hipMemsetAsync(d_msm_out, 0, nbatch * 32, s));
        if (rlc && d_batch_out) HIPCHK(c, hipMemsetAsync(d_batch_out, 0, 33, s));
        HIPCHK(c, hipGetLastError());
        c->rp_status_dirty = false;
        return BPGPU_OK;
    }
    // the generator-exponent role: one index per lane for narrow chains (latency), eight in mirrored pairs otherwise (rp_expand_b8_thread: least work)
    const bool exp_single = !rlc && !wide && c->exp_single && nbatch <= 256 && sh.nm >= 4;
    const bool pairs = !rlc && !exp_single && c->exp_pairs && sh.nm >= 8;
    const uint32_t nexp = (exp_single ? sh.nm : sh.nm / (pairs ? 8 : 4)) * nb32, nwin = rlc_bucket ? 0u : (uint32_t)pd->n_chunks * 64;   // four generator indices per lane (eight in pairs)
    const uint32_t n_win = (nwin + BP_BLOCK - 1) / BP_BLOCK, n_exp = (nexp + BP_BLOCK - 1) / BP_BLOCK;
    if (rlc_bucket) {
        // ---- batch combination, bucket variant: R = sum_i rho_i MegaCheck_i with the per-proof terms as ONE MSM ----
        unsigned long long *d_acc = (unsigned long long *)(a + off_acc);
        HIPCHK(c, hipMemsetAsync(d_acc, 0, (size_t)n_gen_terms * 10 * 8, s));
        LAUNCH(c, s, "rlc_stage3", k_rlc_stage3, n_exp, BP_BLOCK, 0u, 0u, (const vb_chunk *)nullptr, (const ge_cached *)nullptr, (const uint32_t *)nullptr,
               (ge_ext *)nullptr, nexp, sh, prm, d_fields, d_status, d_acc, (nbatch % 64 == 0) ? 1 : 0);
        const uint32_t tot32 = (uint32_t)rlc_terms;
        rc = enqueue_bucket_sort(c, s, bkp, 1u, tot32, rlc_terms, 1, bd, d_status, sh.U);
        if (rc) return rc;
        fb_digit *d_dig1 = (fb_digit *)(a + off_dig1);
        ge_ext *d_part1 = (ge_ext *)(a + off_part1);
        uint32_t *d_ctl = (uint32_t *)(a + off_res1 + sizeof(ge_ext));
        const uint32_t nt = bkp.nwin * bkp.half, n_acc = (nt + BP_BLOCK - 1) / BP_BLOCK, n_sc = (n_gen_terms + BP_BLOCK - 1) / BP_BLOCK;
        const uint32_t lim = bk_chain_lim(rlc_terms, bkp);
        LAUNCH(c, s, "rlc_accum", k_rlc_accum_scalars, n_acc + n_sc, BP_BLOCK, n_acc, nt, bkp, tot32, bd.desc, bd.idx, bd.pts, bd.bsum, n_gen_terms,
               (const unsigned long long *)d_acc, d_dig1, prm, d_ctl, lim);
        const uint32_t hg = bk_heavy_groups(bkp, 1);
        LAUNCH(c, s, "bk_heavy", k_bk_heavy, bkp.nwin * hg, 64, bkp, tot32, (const bk_desc *)bd.desc, (const uint32_t *)bd.idx, (const fb_entry *)bd.pts, bd.bsum, lim, hg);
        enqueue_bucket_reduce(c, s, bkp, bkp.nwin, bd);
        LAUNCH(c, s, "rlc_stage4", k_rlc_stage4b, 1 + nsplit1, FB_BLOCK, bd.colq16, bd.hq, prm, nsplit1, npairs, d_ids, d_dig1, gen_table, d_part1);
        if (d_batch_out)
            LAUNCH(c, s, "rlc_finish", k_rlc_finish<true>, 1, 64, nsplit1, bd.hq, d_part1, nb32, d_status, (uint8_t *)d_verdict, (uint8_t *)d_batch_out, segtab);
        else
            LAUNCH(c, s, "rlc_finish", k_rlc_finish<false>, 1, 64, nsplit1, bd.hq, d_part1, nb32, d_status, (uint8_t *)d_verdict, (uint8_t *)nullptr, segtab);
        HIPCHK(c, hipGetLastError());
        c->rp_status_dirty = false;
        return BPGPU_OK;
    }
    if (rlc) {
        // ---- batch combination (rlc.h): R = sum_i rho_i MegaCheck_i ---------------------------------------
        unsigned long long *d_acc = (unsigned long long *)(a + off_acc);
        HIPCHK(c, hipMemsetAsync(d_acc, 0, (size_t)n_gen_terms * 10 * 8, s));
        LAUNCH(c, s, "rlc_stage3", k_rlc_stage3, n_win + n_exp, BP_BLOCK, n_win, nwin, d.chunks, d.tab, d.recoded, d.part, nexp, sh, prm,
               d_fields, d_status, d_acc, (nbatch % 64 == 0) ? 1 : 0);
        // column sums over ALL chunks of ALL proofs: rows of 64 window sums, tree-added 16- then 8-way until <= 8 rows
        // remain; the Horner wavefront adds those itself
        fb_digit *d_dig1 = (fb_digit *)(a + off_dig1);
        ge_ext *d_part1 = (ge_ext *)(a + off_part1), *d_hq1 = (ge_ext *)(a + off_res1);
        uint32_t *d_ctl = (uint32_t *)(a + off_res1 + sizeof(ge_ext));   // [0] unused, [1] zero, [2..4) chunk bounds {0, rows}
        ge_ext *cur = d.part, *next = (ge_ext *)(a + off_tree);
        uint32_t rows = (uint32_t)pd->n_chunks;
        bool scalars_done = false;
        const uint32_t n_sc = (n_gen_terms + BP_BLOCK - 1) / BP_BLOCK;
        while (rows > 8) {
            const uint32_t group = rows > 64 ? 16 : 8, ng = (rows + group - 1) / group, nt = ng * 64;
            if (ng <= 8) {   // last level: shares its launch with the coefficient reduction
                LAUNCH(c, s, "rlc_colsum", k_rlc_colsum_scalars, (nt + BP_BLOCK - 1) / BP_BLOCK + n_sc, BP_BLOCK, (nt + BP_BLOCK - 1) / BP_BLOCK, nt, rows,
                       group, (const ge_ext *)cur, next, n_gen_terms, (const unsigned long long *)d_acc, d_dig1, prm, d_ctl, ng);
                scalars_done = true;
            } else {
                LAUNCH(c, s, "rlc_colsum", k_fb_reduce, (nt + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, nt, 64u, rows, group, cur, next);
            }
            cur = next;
            next = next + (size_t)ng * 64;
            rows = ng;
        }
        if (!scalars_done)
            LAUNCH(c, s, "rlc_colsum", k_rlc_colsum_scalars, n_sc, BP_BLOCK, 0u, 0u, 0u, 1u, (const ge_ext *)cur, next, n_gen_terms,
                   (const unsigned long long *)d_acc, d_dig1, prm, d_ctl, rows);
        LAUNCH(c, s, "rlc_stage4", k_rp_stage4<64>, 1 + nsplit1, FB_BLOCK, 1u, d_ctl + 2, cur, (const ge_cached *)nullptr, d_hq1, prm, 1u, 1u,
               nsplit1, npairs, d_ids, d_dig1, gen_table, d_part1, 0u, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint8_t *)nullptr, rp_seg_tab{});
        if (d_batch_out)
            LAUNCH(c, s, "rlc_finish", k_rlc_finish<true>, 1, 64, nsplit1, d_hq1, d_part1, nb32, d_status, (uint8_t *)d_verdict, (uint8_t *)d_batch_out, segtab);
        else
            LAUNCH(c, s, "rlc_finish", k_rlc_finish<false>, 1, 64, nsplit1, d_hq1, d_part1, nb32, d_status, (uint8_t *)d_verdict, (uint8_t *)nullptr, segtab);
        HIPCHK(c, hipGetLastError());
        c->rp_status_dirty = false;
        return BPGPU_OK;
    }
    const bool one_chunk = pd->n_chunks == nbatch;   // U <= 32: a chunk's window sums are the MSM's column sums
    // Launch 3 holds two independent roles -- window sums and generator exponents.  Narrow chains keep them fused (one launch fewer on the
    // latency path); wide chains split them, and may move the Horner chains to the second stream (rp_chain_forms).
    bool horner_aside = false;
    if (wide) {
        const bool wide_sums = r5 || a_out;
        // After launch 1 a chain has two INDEPENDENT branches: the proof's own points (window sums -> Horner chain) and the generator terms
        // (exponents -> table walk); they meet in the finish.  With the Horner chains aside, the first branch's LAST kernel goes to the
        // second stream.  (Moving the whole branch there, or the exponents instead, were options until round 6: no gain in their A/B.)
        hipStream_t sw = s;
        // (A burst of two wide chains runs its phases in lockstep -- profiles/r06/headline_burst_kernel_sequence.txt: both window sums, then both
        // exponent launches, 470 us at two wavefronts per SIMD beside nothing but the Horner chains, then the walks.  Issuing the exponents on the second
        // stream FIRST, under the window sums, was built and measured on the same box: no difference in any bench form, cfg4 -1.3 %
        // (profiles/r06/exp_early_ab.txt) -- removed again.)
        if (r5) {
            const uint32_t nw5 = nb32 * BP_VB5_WINDOWS;
            LAUNCH(c, sw, "rp_stage3w", k_vb_window_wide<true>, (nw5 + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, nw5, sh.U, a_out ? 1u : 0u, d.tab, d.recoded, d_colc);
        } else if (a_out) {
            const uint32_t nw4 = nb32 * BP_VB_WINDOWS;
            LAUNCH(c, sw, "rp_stage3w", k_vb_window_wide<false>, (nw4 + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, nw4, sh.U, 1u, d.tab, d.recoded, d_colc);
        } else {
            LAUNCH(c, sw, "rp_stage3w", k_vb_window_colc, n_win, BP_BLOCK, nwin, d.chunks, d.tab, d.recoded, d.part, (quad && one_chunk) ? d_colc : (ge_cached *)nullptr);
        }
        if (aside) {
            if (!one_chunk && !wide_sums) {
                const uint32_t nc = nb32 * 64;
                LAUNCH(c, sw, "vb_colsum", k_vb_colsum, (nc + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, nc, d.chunk_first, d.part, (uint32_t *)nullptr, d_colc);
            }
            HIPCHK(c, hipEventRecord(c->fork_ev, s));
            HIPCHK(c, hipStreamWaitEvent(c->stream2, c->fork_ev, 0));
            const uint32_t n_hb = (nb32 + FB_BLOCK - 1) / FB_BLOCK;
            const ge_cached *extra = a_out ? d.tab : (const ge_cached *)nullptr;
            if (r5) LAUNCH(c, c->stream2, "rp_horner1", k_rp_horner_wide<true>, n_hb, FB_BLOCK, nb32, d_colc, extra, 16u * sh.U, d.hq);
            else if (a_out) LAUNCH(c, c->stream2, "rp_horner1", k_rp_horner_wide<false>, n_hb, FB_BLOCK, nb32, d_colc, extra, 8u * sh.U, d.hq);
            else LAUNCH(c, c->stream2, "rp_horner1", k_rp_horner1, n_hb, FB_BLOCK, nb32, d_colc, d.hq);
            HIPCHK(c, hipEventRecord(c->join_ev, c->stream2));
            horner_aside = true;
        }
        if (pairs && sh.nm >= (uint32_t)c->exp_w3_min_nm) LAUNCH(c, s, "rp_stage3", k_rp_exponents_w3, n_exp, BP_BLOCK, nexp, sh, prm, d_fields, d_digits, d_status);   // (aggregated shapes: three wavefronts per SIMD)
        else if (pairs) LAUNCH(c, s, "rp_stage3", k_rp_exponents<true>, n_exp, BP_BLOCK, nexp, sh, prm, d_fields, d_digits, d_status);
        else LAUNCH(c, s, "rp_stage3", k_rp_exponents<false>, n_exp, BP_BLOCK, nexp, sh, prm, d_fields, d_digits, d_status);
    } else {
        const uint32_t nrows = sh.coop_split ? nb32 : 0u, n_rows = (nrows + BP_BLOCK - 1) / BP_BLOCK;
        ge_cached *colc3 = (quad && one_chunk) ? d_colc : (ge_cached *)nullptr;
        if (exp_single) LAUNCH(c, s, "rp_stage3", k_rp_stage3<2>, n_win + n_exp + n_rows, BP_BLOCK, n_win, nwin, d.chunks, d.tab, d.recoded, d.part, colc3, nexp, sh, prm,
                               d_fields, d_digits, d_status, n_exp, nrows, lg_m, (const ge_cached *)tab_hi);
        else if (pairs) LAUNCH(c, s, "rp_stage3", k_rp_stage3<1>, n_win + n_exp + n_rows, BP_BLOCK, n_win, nwin, d.chunks, d.tab, d.recoded, d.part, colc3, nexp, sh, prm,
                               d_fields, d_digits, d_status, n_exp, nrows, lg_m, (const ge_cached *)tab_hi);
        else LAUNCH(c, s, "rp_stage3", k_rp_stage3<0>, n_win + n_exp + n_rows, BP_BLOCK, n_win, nwin, d.chunks, d.tab, d.recoded, d.part, colc3, nexp, sh, prm,
                    d_fields, d_digits, d_status, n_exp, nrows, lg_m, (const ge_cached *)tab_hi);
    }
    if (quad && !one_chunk && !horner_aside) {
        const uint32_t nc = nb32 * 64;
        LAUNCH(c, s, "vb_colsum", k_vb_colsum, (nc + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, nc, d.chunk_first, d.part, (uint32_t *)nullptr, d_colc);
    }
    const uint32_t nblk_p = (nb32 + FB_BLOCK - 1) / FB_BLOCK;
    uint32_t nparts = nsplit;   // partial sums per proof that launch 4 leaves
    // verdicts only (the crate's own call: nobody asks for the mega-check's encoding) and at most four partial sums per proof: the LAST of a proof's
    // 1 + nsplit / 64 workgroups of launch 4 to arrive adds them to the Horner result and writes the verdict -- no finish launch (option narrow_fused_finish)
    // (same-box A/B, profiles/r06/fused_finish_ab.txt: 64 threads +5 %, 256 threads +3 %, tickets and 16 threads unchanged, ONE proof per chain -2.5 % -- the
    // finisher's three additions then follow the Horner wavefront instead of overlapping with the launch of the next kernel: chains of >= 8 proofs only)
    const bool fused_finish = narrow_walk && c->narrow_fused_finish && nbatch >= 8 && !d_msm_out && nsplit / FB_BLOCK <= 4 && !(quad || horner_aside);
    if (horner_aside) {
        LAUNCH(c, s, "rp_stage4", k_rp_stage4<4>, nblk_p * nsplit, FB_BLOCK, 0u, d.chunk_first, d.part, d_colc, d.hq, prm, nb32, nblk_p,
               nsplit, npairs, d_ids, d_digits, gen_table, d_partial, 0u, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint8_t *)nullptr, rp_seg_tab{});
        HIPCHK(c, hipStreamWaitEvent(s, c->join_ev, 0));
    } else if (quad && one_lane) {
        const uint32_t n_hw = (nb32 + FB_BLOCK - 1) / FB_BLOCK;
        LAUNCH(c, s, "rp_stage4", k_rp_stage4<1>, n_hw + nblk_p * nsplit, FB_BLOCK, n_hw, d.chunk_first, d.part, d_colc, d.hq, prm, nb32, nblk_p,
               nsplit, npairs, d_ids, d_digits, gen_table, d_partial, 0u, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint8_t *)nullptr, rp_seg_tab{});
    } else if (quad) {
        const uint32_t n_hw = (nb32 + 15) / 16;
        LAUNCH(c, s, "rp_stage4", k_rp_stage4<4>, n_hw + nblk_p * nsplit, FB_BLOCK, n_hw, d.chunk_first, d.part, d_colc, d.hq, prm, nb32, nblk_p,
               nsplit, npairs, d_ids, d_digits, gen_table, d_partial, 0u, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint8_t *)nullptr, rp_seg_tab{});
    } else if (narrow_walk) {   // lane = split: a proof's partial sums are folded inside launch 4 (k_rp34.hip: rp_walk_narrow)
        nparts = nsplit / FB_BLOCK;
        LAUNCH(c, s, "rp_stage4", k_rp_stage4<64>, nb32 + nb32 * nparts, FB_BLOCK, nb32, d.chunk_first, d.part, (const ge_cached *)nullptr, d.hq,
               prm, nb32, nblk_p, nsplit, npairs, d_ids, d_digits, gen_table, d_partial, (hi ? (hi_levels == 4 ? 9u : 3u) : 1u) | (fused_finish ? 4u : 0u), c->fin_cnt, d_status,
               (uint8_t *)d_verdict, segtab);
    } else {
        LAUNCH(c, s, "rp_stage4", k_rp_stage4<64>, nb32 + nblk_p * nsplit, FB_BLOCK, nb32, d.chunk_first, d.part, (const ge_cached *)nullptr, d.hq,
               prm, nb32, nblk_p, nsplit, npairs, d_ids, d_digits, gen_table, d_partial, 0u, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint8_t *)nullptr, rp_seg_tab{});
    }
    if (fused_finish) {   // (the last workgroup of every proof has written its verdict and zeroed its status word)
        HIPCHK(c, hipGetLastError());
        c->rp_status_dirty = false;
        return BPGPU_OK;
    }
    ge_ext *d_red = nullptr;
    uint32_t nred = 0;
    enqueue_fb_reduce(c, s, nb32, nparts, d_partial, &d_red, &nred, 64);   // 8 lanes x <= 8 partials each in finish8
    if (nred <= 2 && narrow_walk) {   // one or two partial sums + the Horner result per proof: lane = proof (more: eight lanes per proof and a three-level fold, below)
        if (d_msm_out)
            LAUNCH(c, s, "finish1", k_finish1<true>, (nb32 + 63) / 64, 64, nb32, nred, d.hq, d_red, d_status, (uint32_t *)d_msm_out, (uint8_t *)d_verdict, 1, segtab);
        else
            LAUNCH(c, s, "finish1", k_finish1<false>, (nb32 + 63) / 64, 64, nb32, nred, d.hq, d_red, d_status, (uint32_t *)nullptr, (uint8_t *)d_verdict, 1, segtab);
    } else if (d_msm_out)
        LAUNCH(c, s, "finish8", k_finish8<true>, (nb32 + 7) / 8, 64, nb32, nred, d.hq, d_red, d_status, (uint32_t *)d_msm_out, (uint8_t *)d_verdict, 1, segtab);
    else
        LAUNCH(c, s, "finish8", k_finish8<false>, (nb32 + 7) / 8, 64, nb32, nred, d.hq, d_red, d_status, (uint32_t *)nullptr, (uint8_t *)d_verdict, 1, segtab);
    HIPCHK(c, hipGetLastError());
    c->rp_status_dirty = false;
    return BPGPU_OK;
}

// ---- entry points: device pointers ----------------------------------------------------------------------
static int rp_dev_call(bpgpu_ctx *c, size_t n, size_t m, size_t nbatch, const void *d_proofs, size_t proof_len, const void *d_commitments,
                       const rp_transcripts &tr, const void *d_rng64, void *d_verdict, void *d_msm_out, void *stream, bool rlc,
                       const void *d_weights64, void *d_batch_out) {
    if (!c || (nbatch && (!d_proofs || !d_verdict || (m && !d_commitments))) || (tr.label_len && !tr.label)) return BPGPU_ERR_INVALID_ARG;
    if (((uintptr_t)d_proofs | (uintptr_t)d_commitments | (uintptr_t)d_rng64 | (uintptr_t)d_weights64 | (uintptr_t)tr.d_ts_in | (uintptr_t)tr.d_ts_out) & 3)
        return fail(c, BPGPU_ERR_INVALID_ARG, "device buffers must be 4-byte aligned");
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    int rc = ctx_enter(c, s);
    if (rc) return rc;
    rc = rp_verify_dev_locked(c, n, m, nbatch, d_proofs, proof_len, d_commitments, tr, d_rng64, d_verdict, d_msm_out, s, rlc, d_weights64, d_batch_out,
                              nullptr, 0, true);
    const int rc2 = ctx_leave(c, s);
    return rc ? rc : rc2;
}

// ---- hooks of the pool (pool.hip; not part of the C ABI) -----------------------------------------------------
// can batches of this shape share a launch chain?  (malformed lengths / parameter errors take the ordinary entry point,
// which reports them per proof)
bool bpgpu_internal_rp_coalescible(bpgpu_ctx *c, size_t n, size_t m, size_t proof_len) {
    if (proof_len % 32 != 0 || proof_len < 9 * 32 || ((proof_len / 32 - 9) & 1)) return false;
    const size_t k = (proof_len / 32 - 9) / 2;
    if (k > BP_RP_MAX_K || !(n == 8 || n == 16 || n == 32 || n == 64) || m == 0) return false;
    std::lock_guard<std::mutex> lk(c->mu);
    return c->d_table && c->gens_capacity >= n && c->party_capacity >= m && n * m == ((size_t)1 << k);
}
// has everything enqueued on the context's own stream completed?
bool bpgpu_internal_idle(bpgpu_ctx *c) {
    std::lock_guard<std::mutex> lk(c->mu);
    if (hipSetDevice(c->device) != hipSuccess) return false;
    const hipError_t e = hipStreamQuery(c->stream);
    if (e != hipSuccess) (void)hipGetLastError();   // hipErrorNotReady is not an error
    return e == hipSuccess;
}
// What the pool knows about the chain it issues on this context -- busy: 1 = others run beside it, 0 = alone, -1 = nothing -- is an
// argument of the two hooks below and is forgotten when the call returns: a lane never runs on what an earlier call left behind.
// (bpgpu_internal_set_busy_hint stays for the host-pointer slices, whose worker sets it immediately before every submit.)
void bpgpu_internal_set_busy_hint(bpgpu_ctx *c, int busy) {
    std::lock_guard<std::mutex> lk(c->mu);
    c->busy_hint = busy;
}
// one coalesced launch chain over the concatenation of `nseg` items, on the context's own stream (asynchronous)
// rlc: ONE batch-combined check over all items (bpgpu_pool_rangeproof_submit_rlc_dev); any_msm then means "some item wants the 33-byte result"
int bpgpu_internal_rp_verify_segs(bpgpu_ctx *c, size_t n, size_t m, size_t proof_len, const uint8_t *const *labels, size_t label_len, const rp_seg *segs,
                                  uint32_t nseg, bool any_msm, uint32_t splits_hint, int busy, bool rlc) {
    if (!c || !segs || nseg == 0 || !labels) return BPGPU_ERR_INVALID_ARG;
    const uint8_t *label = labels[0];
    const size_t total = (size_t)segs[nseg - 1].first + segs[nseg - 1].count;
    bool any_rng_missing = false;
    for (uint32_t i = 0; i < nseg; i++) any_rng_missing = any_rng_missing || !segs[i].rng64;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    int rc = ctx_enter(c, s);
    if (rc) return rc;
    rp_transcripts tr;
    tr.label = label;
    tr.label_len = label_len;
    tr.seg_labels = labels;
    c->splits_hint = splits_hint;
    c->busy_hint = busy;
    // d_rng64: a non-null dummy keeps the library from drawing randomness nobody reads (every item brought its own)
    rc = rp_verify_dev_locked(c, n, m, total, nullptr, proof_len, nullptr, tr, any_rng_missing ? nullptr : (const void *)segs[0].rng64, nullptr,
                              (any_msm && !rlc) ? (void *)segs : nullptr, s, rlc, nullptr, (any_msm && rlc) ? (void *)segs : nullptr, segs, nseg, true);
    c->splits_hint = 0;
    c->busy_hint = -1;
    const int rc2 = ctx_leave(c, s);
    return rc ? rc : rc2;
}
// buffers of the context sized for chains of up to `nbatch_max` proofs of this shape (see rp_verify_dev_locked, reserve_only)
int bpgpu_internal_rp_reserve(bpgpu_ctx *c, size_t n, size_t m, size_t proof_len, size_t nbatch_max) {
    if (!c) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    rp_transcripts tr;
    c->busy_hint = 1;
    const int rc = rp_verify_dev_locked(c, n, m, nbatch_max, nullptr, proof_len, nullptr, tr, (const void *)c, nullptr, nullptr, c->stream, false, nullptr, nullptr,
                                        nullptr, 0, true, true);
    c->busy_hint = -1;
    return rc;
}
// the context's own stream (the pool's combining queue enqueues its staging copies around the chain)
void *bpgpu_internal_stream(bpgpu_ctx *c) { return c ? (void *)c->stream : nullptr; }
// one launch chain over device-resident inputs on the context's own stream (asynchronous): what the combining queue of the pool
// issues for a sealed buffer.  Transcripts: `shared_ts` (one 208-byte state for every proof; no states handed back), or one state
// per proof in d_ts_in / d_ts_out -- with ts_uniform all of them at STROBE position (pos, pos_begin, flags): the scripted replay.
int bpgpu_internal_rp_verify_chain(bpgpu_ctx *c, size_t n, size_t m, size_t nbatch, const void *d_proofs, size_t proof_len, const void *d_commitments,
                                   const uint8_t *shared_ts, const void *d_ts_in, void *d_ts_out, int ts_uniform, uint32_t pos, uint32_t pos_begin,
                                   uint32_t flags, const void *d_rng64, void *d_verdict, void *d_msm_out, uint32_t splits_hint, int busy) {
    if (!c || !d_proofs || !d_verdict || !d_rng64 || (shared_ts == nullptr) == (d_ts_in == nullptr)) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    int rc = ctx_enter(c, s);
    if (rc) return rc;
    rp_transcripts tr;
    tr.shared_ts = shared_ts;
    tr.d_ts_in = d_ts_in;
    tr.d_ts_out = d_ts_out;
    tr.ts_uniform = d_ts_in && ts_uniform;
    tr.u_pos = pos;
    tr.u_pos_begin = pos_begin;
    tr.u_flags = flags;
    c->splits_hint = splits_hint;
    c->busy_hint = busy;
    rc = rp_verify_dev_locked(c, n, m, nbatch, d_proofs, proof_len, d_commitments, tr, d_rng64, d_verdict, d_msm_out, s, false, nullptr, nullptr, nullptr, 0, true);
    c->splits_hint = 0;
    c->busy_hint = -1;
    const int rc2 = ctx_leave(c, s);
    return rc ? rc : rc2;
}

extern "C" int bpgpu_rangeproof_verify_batch_dev(bpgpu_ctx *c, size_t n, size_t m, size_t nbatch, const void *d_proofs, size_t proof_len,
                                                 const void *d_commitments, const uint8_t *label, size_t label_len, const void *d_rng64,
                                                 void *d_verdict, void *d_msm_out, void *stream) {
    rp_transcripts tr;
    tr.label = label;
    tr.label_len = label_len;
    return rp_dev_call(c, n, m, nbatch, d_proofs, proof_len, d_commitments, tr, d_rng64, d_verdict, d_msm_out, stream, false, nullptr, nullptr);
}

extern "C" int bpgpu_rangeproof_verify_batch_ts_dev(bpgpu_ctx *c, size_t n, size_t m, size_t nbatch, const void *d_proofs, size_t proof_len,
                                                    const void *d_commitments, const uint8_t *shared_transcript, const void *d_transcripts,
                                                    const void *d_rng64, void *d_verdict, void *d_msm_out, void *d_transcripts_out,
                                                    void *stream) {
    if (!c) return BPGPU_ERR_INVALID_ARG;
    if ((shared_transcript != nullptr) == (d_transcripts != nullptr))
        return fail(c, BPGPU_ERR_INVALID_ARG, "give exactly one of shared_transcript / d_transcripts");
    rp_transcripts tr;
    tr.shared_ts = shared_transcript;
    tr.d_ts_in = d_transcripts;
    tr.d_ts_out = d_transcripts_out;
    return rp_dev_call(c, n, m, nbatch, d_proofs, proof_len, d_commitments, tr, d_rng64, d_verdict, d_msm_out, stream, false, nullptr, nullptr);
}

extern "C" int bpgpu_rangeproof_verify_rlc_dev(bpgpu_ctx *c, size_t n, size_t m, size_t nbatch, const void *d_proofs, size_t proof_len,
                                               const void *d_commitments, const uint8_t *label, size_t label_len, const void *d_rng64,
                                               const void *d_weights64, void *d_verdict, void *d_batch_out, void *stream) {
    rp_transcripts tr;
    tr.label = label;
    tr.label_len = label_len;
    return rp_dev_call(c, n, m, nbatch, d_proofs, proof_len, d_commitments, tr, d_rng64, d_verdict, nullptr, stream, true, d_weights64, d_batch_out);
}

// ---- entry points: host pointers -------------------------------------------------------------------------
// Inputs are packed into the context's pinned staging buffer and go to the persistent device IO buffer in ONE
// asynchronous copy; results come back in one copy; one wait at the end.  No allocation in steady state.
static int rp_host_call(bpgpu_ctx *c, size_t n, size_t m, size_t nbatch, const uint8_t *proofs, size_t proof_len, const uint8_t *commitments,
                        rp_transcripts tr, const uint8_t *ts_in_host, const uint8_t *rng64, uint8_t *verdict, uint8_t *msm_out,
                        uint8_t *ts_out_host, bool rlc, const uint8_t *weights64, uint8_t *batch_out, bool submit = false) {
    if (!c || (nbatch && (!proofs || !verdict || (m && !commitments))) || (tr.label_len && !tr.label)) return BPGPU_ERR_INVALID_ARG;
    if (nbatch == 0) {
        if (batch_out) memset(batch_out, 0, 33);
        return BPGPU_OK;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t TS = BPGPU_TRANSCRIPT_BYTES;
    const size_t sz_p = align_up(nbatch * proof_len + 64), sz_c = align_up(nbatch * m * 32 + 64), sz_r = rng64 ? align_up(nbatch * 64) : 0,
                 sz_w = weights64 ? align_up(nbatch * 64) : 0, sz_ti = ts_in_host ? align_up(nbatch * TS) : 0;
    const size_t sz_in = sz_p + sz_c + sz_r + sz_w + sz_ti;
    const size_t sz_v = align_up(nbatch), sz_o = msm_out ? align_up(nbatch * 32) : 0, sz_b = rlc ? align_up(64) : 0,
                 sz_to = ts_out_host ? align_up(nbatch * TS) : 0;
    const size_t sz_out = sz_v + sz_o + sz_b + sz_to;
    hipStream_t s = c->stream;
    int rc = ctx_enter(c, s);
    if (rc) return rc;
    rc = io_reserve(c, sz_in + sz_out);
    if (rc) return rc;
    char *h = nullptr;
    rc = pin_alloc(c, s, sz_in + sz_out, &h);
    if (rc) return rc;
    char *d = c->io_dev;
    char *d_p = d, *d_c = d_p + sz_p, *d_r = d_c + sz_c, *d_w = d_r + sz_r, *d_ti = d_w + sz_w;
    char *d_v = d + sz_in, *d_o = d_v + sz_v, *d_b = d_o + sz_o, *d_to = d_b + sz_b;
    memcpy(h, proofs, nbatch * proof_len);
    if (m) memcpy(h + sz_p, commitments, nbatch * m * 32);
    if (rng64) memcpy(h + sz_p + sz_c, rng64, nbatch * 64);
    if (weights64) memcpy(h + sz_p + sz_c + sz_r, weights64, nbatch * 64);
    if (ts_in_host) memcpy(h + sz_p + sz_c + sz_r + sz_w, ts_in_host, nbatch * TS);
    HIPCHK(c, hipMemcpyAsync(d, h, sz_in, hipMemcpyHostToDevice, s));
    tr.d_ts_in = ts_in_host ? d_ti : nullptr;
    tr.d_ts_out = ts_out_host ? d_to : nullptr;
    char *h_out = h + sz_in;
    rc = rp_verify_dev_locked(c, n, m, nbatch, d_p, proof_len, d_c, tr, rng64 ? d_r : nullptr, d_v, msm_out ? d_o : nullptr, s, rlc,
                              weights64 ? d_w : nullptr, rlc ? d_b : nullptr);
    if (!rc && hipMemcpyAsync(h_out, d_v, sz_out, hipMemcpyDeviceToHost, s) != hipSuccess) rc = fail(c, BPGPU_ERR_HIP, "D2H copy failed");
    if (submit && !rc) {   // asynchronous form: leave the results in flight; bpgpu_ctx_collect (or the next call) delivers them
        const int rcl = ctx_leave(c, s);
        if (rcl) return rcl;
        c->pend.active = true;
        c->pend.h_out = h_out;
        c->pend.nbatch = nbatch;
        c->pend.off_msm = sz_v;
        c->pend.off_ts = sz_v + sz_o + sz_b;
        c->pend.verdict = verdict;
        c->pend.msm_out = msm_out;
        c->pend.ts_out = ts_out_host;
        return BPGPU_OK;
    }
    int rc2 = ctx_leave(c, s), rc3 = host_wait(c, s);
    if (rc || rc2 || rc3) return rc ? rc : (rc2 ? rc2 : rc3);
    if (rlc && (uint8_t)h_out[sz_v + sz_o] != 0) {
        // the combination is not the identity: some proof fails -- find out which, proof by proof (inputs are still on the device)
        uint8_t bo[33];
        memcpy(bo, h_out + sz_v + sz_o, 33);
        // h_out may sit in a buffer that the first pass outgrew (its nested allocations for library-drawn rng / weights)
        // and retired: it must stay alive until the verdicts below have been copied out of it
        rc = ctx_enter(c, s, true);
        if (rc) return rc;
        rc = rp_verify_dev_locked(c, n, m, nbatch, d_p, proof_len, d_c, tr, rng64 ? d_r : nullptr, d_v, nullptr, s);
        if (!rc && hipMemcpyAsync(h_out, d_v, sz_v, hipMemcpyDeviceToHost, s) != hipSuccess) rc = fail(c, BPGPU_ERR_HIP, "D2H copy failed");
        rc2 = ctx_leave(c, s);
        rc3 = host_wait(c, s);
        if (rc || rc2 || rc3) return rc ? rc : (rc2 ? rc2 : rc3);
        memcpy(h_out + sz_v + sz_o, bo, 33);
    }
    memcpy(verdict, h_out, nbatch);
    if (msm_out) memcpy(msm_out, h_out + sz_v, nbatch * 32);
    if (rlc && batch_out) memcpy(batch_out, h_out + sz_v + sz_o, 33);
    if (ts_out_host) memcpy(ts_out_host, h_out + sz_v + sz_o + sz_b, nbatch * TS);
    return BPGPU_OK;
}

extern "C" int bpgpu_rangeproof_verify_batch(bpgpu_ctx *c, size_t n, size_t m, size_t nbatch, const uint8_t *proofs, size_t proof_len,
                                             const uint8_t *commitments, const uint8_t *label, size_t label_len, const uint8_t *rng64,
                                             uint8_t *verdict, uint8_t *msm_out) {
    rp_transcripts tr;
    tr.label = label;
    tr.label_len = label_len;
    return rp_host_call(c, n, m, nbatch, proofs, proof_len, commitments, tr, nullptr, rng64, verdict, msm_out, nullptr, false, nullptr, nullptr);
}

// asynchronous forms: enqueue and return; verdict / msm_out (and the input buffers' contents are already staged) are
// filled by bpgpu_ctx_collect or implicitly by the next call on the context
extern "C" int bpgpu_rangeproof_verify_batch_submit(bpgpu_ctx *c, size_t n, size_t m, size_t nbatch, const uint8_t *proofs, size_t proof_len,
                                                    const uint8_t *commitments, const uint8_t *label, size_t label_len, const uint8_t *rng64,
                                                    uint8_t *verdict, uint8_t *msm_out) {
    rp_transcripts tr;
    tr.label = label;
    tr.label_len = label_len;
    return rp_host_call(c, n, m, nbatch, proofs, proof_len, commitments, tr, nullptr, rng64, verdict, msm_out, nullptr, false, nullptr, nullptr, true);
}
extern "C" int bpgpu_ctx_collect(bpgpu_ctx *c) {
    if (!c) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    return collect_locked(c);
}

extern "C" int bpgpu_rangeproof_verify_batch_ts(bpgpu_ctx *c, size_t n, size_t m, size_t nbatch, const uint8_t *proofs, size_t proof_len,
                                                const uint8_t *commitments, const uint8_t *transcripts, size_t transcript_stride,
                                                const uint8_t *rng64, uint8_t *verdict, uint8_t *msm_out, uint8_t *transcripts_out) {
    if (!c) return BPGPU_ERR_INVALID_ARG;
    if (!transcripts || (transcript_stride != 0 && transcript_stride != BPGPU_TRANSCRIPT_BYTES))
        return fail(c, BPGPU_ERR_INVALID_ARG, "transcripts missing, or transcript_stride neither 0 nor BPGPU_TRANSCRIPT_BYTES");
    if (transcript_stride)
        for (size_t b = 0; b < nbatch; b++)
            if (!ts_state_ok(transcripts + b * BPGPU_TRANSCRIPT_BYTES)) return fail(c, BPGPU_ERR_INVALID_ARG, "malformed transcript state %zu", b);
    rp_transcripts tr;
    if (!transcript_stride) tr.shared_ts = transcripts;
    return rp_host_call(c, n, m, nbatch, proofs, proof_len, commitments, tr, transcript_stride ? transcripts : nullptr, rng64, verdict, msm_out,
                        transcripts_out, false, nullptr, nullptr);
}

// ---- batch combination entry points (rlc.h; SURVEY 8f-3, additional to the reference's API) ---------------
extern "C" int bpgpu_rangeproof_verify_rlc(bpgpu_ctx *c, size_t n, size_t m, size_t nbatch, const uint8_t *proofs, size_t proof_len,
                                           const uint8_t *commitments, const uint8_t *label, size_t label_len, const uint8_t *rng64,
                                           const uint8_t *weights64, uint8_t *verdict, uint8_t *batch_out) {
    rp_transcripts tr;
    tr.label = label;
    tr.label_len = label_len;
    uint8_t bo[33];
    return rp_host_call(c, n, m, nbatch, proofs, proof_len, commitments, tr, nullptr, rng64, verdict, nullptr, nullptr, true, weights64,
                        batch_out ? batch_out : bo);
}

// ============================================================================
// stand-alone inner-product proofs
// ============================================================================
// working set of the inner-product / linear / audit entry points (separate from the arena, which the MSMs they call claim):
// grown on demand, never shrunk; a call that grows it first waits for everything that may still read the old block
static int ipp_reserve(bpgpu_ctx *c, size_t need) {
    if (c->ipp_cap >= need) return BPGPU_OK;
    HIPCHK(c, hipDeviceSynchronize());
    if (c->ipp_buf) HIPCHK(c, hipFree(c->ipp_buf));
    c->ipp_buf = nullptr;
    c->ipp_cap = 0;
    if (hipMalloc((void **)&c->ipp_buf, need + need / 4) != hipSuccess) return fail(c, BPGPU_ERR_HIP, "out of device memory (%zu bytes of working set)", need + need / 4);
    c->ipp_cap = need + need / 4;
    return BPGPU_OK;
}

// Transcript::new(label) [or the caller's 208-byte state] followed by innerproduct_domain_sep(n) (transcript.rs:50-53): the
// batch-invariant prefix of the stand-alone inner-product and linear proofs, replayed once on the host
static void ipp_domain_sep_state(uint8_t st0[BPGPU_TRANSCRIPT_BYTES], const uint8_t *label, size_t label_len, const uint8_t *shared_ts, size_t n,
                                 rp_strobe_init *init) {
    if (shared_ts) memcpy(st0, shared_ts, BPGPU_TRANSCRIPT_BYTES);
    else bpgpu_transcript_new(label, label_len, st0);
    uint32_t w[50];
    strobe t;
    ts_to_strobe(t, w, st0);
    const uint8_t ipp[6] = {'i', 'p', 'p', ' ', 'v', '1'}, ln[1] = {'n'};
    merlin_append_message(t, DOM_SEP, 7, ipp, 6);
    merlin_append_u64(t, ln, 1, n);
    ts_from_strobe(st0, t);
    if (init) {
        memcpy(init->w, w, 200);
        init->pos = t.pos;
        init->pos_begin = t.pos_begin;
        init->cur_flags = t.cur_flags;
    }
}

static int ipp_verify_dev_locked(bpgpu_ctx *c, size_t n, size_t nbatch, const void *d_proofs, size_t proof_len, const uint8_t *label,
                                 size_t label_len, const uint8_t *shared_ts, const void *d_Gf, const void *d_Hf, const void *d_P,
                                 const void *d_Q, const void *d_G, const void *d_H, int bases_shared, void *d_verdict, void *d_msm_out,
                                 hipStream_t s) {
    if (nbatch > 0x7fffffffu / 64) return fail(c, BPGPU_ERR_INVALID_ARG, "batch too large");
    if (shared_ts && !ts_state_ok(shared_ts)) return fail(c, BPGPU_ERR_INVALID_ARG, "malformed transcript state");
    // InnerProductProof::from_bytes, length part (ipp.rs:374-388)
    size_t k = 0;
    bool fmt = false;
    if (proof_len % 32 != 0) fmt = true;
    else {
        const size_t ne = proof_len / 32;
        if (ne < 2 || (ne - 2) % 2 != 0) fmt = true;
        else {
            k = (ne - 2) / 2;
            if (k >= 32) fmt = true;
        }
    }
    if (fmt) {
        HIPCHK(c, hipMemsetAsync(d_verdict, BPGPU_VERDICT_FORMAT_ERROR, nbatch, s));
        if (d_msm_out) HIPCHK(c, hipMemsetAsync(d_msm_out, 0, nbatch * 32, s));
        return BPGPU_OK;
    }
    ipp_shape sh;
    sh.n = (uint32_t)n;
    sh.k = (uint32_t)k;
    sh.proof_len = (uint32_t)proof_len;
    sh.nproofs = (uint32_t)nbatch;
    sh.bases_shared = bases_shared ? 1u : 0u;
    if (n == ((size_t)1 << k) && k > BP_RP_MAX_K) return fail(c, BPGPU_ERR_INVALID_ARG, "n > 2^%d not supported", BP_RP_MAX_K);
    sh.shape_verdict = (n == ((size_t)1 << k)) ? 0 : BPGPU_VERDICT_VERIFICATION_ERROR;   // ipp.rs:203-211
    const size_t n_eff = sh.shape_verdict ? 0 : n;
    sh.N = (uint32_t)(2 * n_eff + 2 * (sh.shape_verdict ? 0 : k) + 2);
    if (sh.shape_verdict) sh.n = 0;   // only the canonical-scalar check runs
    const size_t N = sh.N;
    if ((uint64_t)nbatch * N > 0x7fffffffull / 64) return fail(c, BPGPU_ERR_INVALID_ARG, "batch too large for this shape");
    // scratch: the (scalar, point) term lists, front-end status, MSM status / result; own allocation because the MSM below
    // claims the arena
    const size_t sz_terms = align_up(nbatch * N * 32 + 64), sz_st = align_up(nbatch * 4), sz_b = align_up(nbatch + 64), sz_o = align_up(nbatch * 32 + 64);
    const size_t need = 2 * sz_terms + sz_st + sz_b + sz_o;
    {
        const int rcr = ipp_reserve(c, need);
        if (rcr) return rcr;
    }
    char *d_sc = c->ipp_buf, *d_pt = d_sc + sz_terms, *d_stat = d_pt + sz_terms, *d_mst = d_stat + sz_st, *d_out = d_mst + sz_b;
    HIPCHK(c, hipMemsetAsync(d_sc, 0, 2 * sz_terms + sz_st, s));   // scalars, points, status
    rp_strobe_init init;
    {
        uint8_t st0[BPGPU_TRANSCRIPT_BYTES];
        ipp_domain_sep_state(st0, label, label_len, shared_ts, n, &init);
    }
    const uint32_t nb32 = (uint32_t)nbatch;
    LAUNCH(c, s, "ipp_prepare", k_ipp_prepare, (nb32 + RP_BLOCK - 1) / RP_BLOCK, RP_BLOCK, sh, init, (const uint8_t *)d_proofs, (const uint8_t *)d_Gf,
           (const uint8_t *)d_Hf, (const uint8_t *)d_P, (const uint8_t *)d_Q, (const uint8_t *)d_G, (const uint8_t *)d_H, (uint32_t *)d_sc,
           (uint32_t *)d_pt, (uint32_t *)d_stat);
    std::vector<uint32_t> nt(nbatch, (uint32_t)N);
    int rc = msm_batch_dev_locked(c, nbatch, nt.data(), d_sc, d_pt, d_out, d_mst, s);
    if (rc) return rc;
    LAUNCH(c, s, "ipp_verdict", k_ipp_verdict, (nb32 + 63) / 64, 64, nb32, (const uint32_t *)d_stat, (const uint8_t *)d_mst, (const uint32_t *)d_out,
           (uint8_t *)d_verdict);
    if (d_msm_out) HIPCHK(c, hipMemcpyAsync(d_msm_out, d_out, nbatch * 32, hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipGetLastError());
    return BPGPU_OK;
}

extern "C" int bpgpu_ipp_verify_batch_dev(bpgpu_ctx *c, size_t n, size_t nbatch, const void *d_proofs, size_t proof_len, const uint8_t *label,
                                          size_t label_len, const uint8_t *shared_transcript, const void *d_G_factors, const void *d_H_factors,
                                          const void *d_P, const void *d_Q, const void *d_G, const void *d_H, int bases_shared, void *d_verdict,
                                          void *d_msm_out, void *stream) {
    if (!c || (label_len && !label)) return BPGPU_ERR_INVALID_ARG;
    if (nbatch == 0) return BPGPU_OK;
    if (!d_proofs || !d_verdict || !d_P || !d_Q || (n && (!d_G_factors || !d_H_factors || !d_G || !d_H))) return BPGPU_ERR_INVALID_ARG;
    if (((uintptr_t)d_proofs | (uintptr_t)d_G_factors | (uintptr_t)d_H_factors | (uintptr_t)d_P | (uintptr_t)d_Q | (uintptr_t)d_G | (uintptr_t)d_H) & 3)
        return fail(c, BPGPU_ERR_INVALID_ARG, "device buffers must be 4-byte aligned");
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    int rc = ctx_enter(c, s);
    if (rc) return rc;
    rc = ipp_verify_dev_locked(c, n, nbatch, d_proofs, proof_len, label, label_len, shared_transcript, d_G_factors, d_H_factors, d_P, d_Q, d_G, d_H,
                               bases_shared, d_verdict, d_msm_out, s);
    const int rc2 = ctx_leave(c, s);
    return rc ? rc : rc2;
}

// InnerProductProof::from_bytes + verification_scalars (ipp.rs:198-253, 373-407) for nbatch proofs, host pointers
extern "C" int bpgpu_ipp_verification_scalars(bpgpu_ctx *c, size_t n, size_t nbatch, const uint8_t *proofs, size_t proof_len, const uint8_t *label,
                                              size_t label_len, const uint8_t *transcripts, size_t transcript_stride, uint8_t *u_sq, uint8_t *u_inv_sq,
                                              uint8_t *s_out, uint8_t *transcripts_out, uint8_t *status) {
    if (!c || (label_len && !label)) return BPGPU_ERR_INVALID_ARG;
    if (nbatch == 0) return BPGPU_OK;
    if (!proofs || !status || !u_sq || !u_inv_sq || (n && !s_out)) return BPGPU_ERR_INVALID_ARG;
    if (transcripts && transcript_stride != 0 && transcript_stride != BPGPU_TRANSCRIPT_BYTES)
        return fail(c, BPGPU_ERR_INVALID_ARG, "transcript_stride neither 0 nor BPGPU_TRANSCRIPT_BYTES");
    const bool per_proof = transcripts && transcript_stride;
    if (transcripts)
        for (size_t b = 0; b < (per_proof ? nbatch : 1); b++)
            if (!ts_state_ok(transcripts + b * BPGPU_TRANSCRIPT_BYTES)) return fail(c, BPGPU_ERR_INVALID_ARG, "malformed transcript state %zu", b);
    // InnerProductProof::from_bytes, length part (ipp.rs:374-388)
    size_t k = 0;
    bool fmt = proof_len % 32 != 0;
    if (!fmt) {
        const size_t ne = proof_len / 32;
        if (ne < 2 || (ne - 2) % 2 != 0) fmt = true;
        else {
            k = (ne - 2) / 2;
            if (k >= 32) fmt = true;
        }
    }
    // the caller's transcript as it is handed back when nothing was absorbed: from_bytes failed, or n != 2^k (ipp.rs:203-211 returns before
    // innerproduct_domain_sep)
    uint8_t st_start[BPGPU_TRANSCRIPT_BYTES];
    if (transcripts && !per_proof) memcpy(st_start, transcripts, BPGPU_TRANSCRIPT_BYTES);
    else if (!transcripts) bpgpu_transcript_new(label, label_len, st_start);
    auto untouched = [&](size_t b) {
        if (transcripts_out) memcpy(transcripts_out + b * BPGPU_TRANSCRIPT_BYTES, per_proof ? transcripts + b * BPGPU_TRANSCRIPT_BYTES : st_start, BPGPU_TRANSCRIPT_BYTES);
    };
    if (fmt) {
        memset(status, BPGPU_VERDICT_FORMAT_ERROR, nbatch);
        for (size_t b = 0; b < nbatch; b++) untouched(b);
        return BPGPU_OK;
    }
    const bool shape_bad = n != ((size_t)1 << k);   // ipp.rs:203-211 (k >= 32 was refused above)
    if (!shape_bad && k > BP_RP_MAX_K) return fail(c, BPGPU_ERR_INVALID_ARG, "n > 2^%d not supported", BP_RP_MAX_K);
    if ((uint64_t)nbatch * (n ? n : 1) > 0x7fffffffull / 64) return fail(c, BPGPU_ERR_INVALID_ARG, "batch too large for this shape");
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    ipp_shape sh;
    sh.n = (uint32_t)n;
    sh.k = (uint32_t)k;
    sh.N = 0;
    sh.proof_len = (uint32_t)proof_len;
    sh.nproofs = (uint32_t)nbatch;
    sh.shape_verdict = shape_bad ? BPGPU_VERDICT_VERIFICATION_ERROR : 0;
    sh.bases_shared = 0;
    const size_t n_eff = shape_bad ? 0 : n, TS = BPGPU_TRANSCRIPT_BYTES;
    const size_t sz_pr = align_up(nbatch * proof_len + 64), sz_ti = per_proof ? align_up(nbatch * TS) : 0;
    const size_t sz_in = sz_pr + sz_ti;
    const size_t sz_u = align_up(nbatch * (k ? k : 1) * 32), sz_s = align_up(nbatch * (n_eff ? n_eff : 1) * 32), sz_to = transcripts_out ? align_up(nbatch * TS) : 0,
                 sz_st = align_up(nbatch * 4);
    const size_t sz_out = 2 * sz_u + sz_s + sz_to + sz_st;
    const size_t sz_tab = align_up(nbatch * (k ? k : 1) * 2 * 40);
    hipStream_t s = c->stream;
    int rc = ctx_enter(c, s);
    if (rc) return rc;
    char *h = nullptr;
    do {
        rc = io_reserve(c, sz_in + sz_out + sz_tab);
        if (rc) break;
        rc = pin_alloc(c, s, sz_in + sz_out, &h);
        if (rc) break;
        char *d = c->io_dev;
        char *d_pr = d, *d_ti = d + sz_pr, *d_us = d + sz_in, *d_ui = d_us + sz_u, *d_s = d_ui + sz_u, *d_to = d_s + sz_s, *d_st = d_to + sz_to, *d_tab = d + sz_in + sz_out;
        memcpy(h, proofs, nbatch * proof_len);
        if (per_proof) memcpy(h + sz_pr, transcripts, nbatch * TS);
        if (hipMemcpyAsync(d, h, sz_in, hipMemcpyHostToDevice, s) != hipSuccess || hipMemsetAsync(d_us, 0, sz_out, s) != hipSuccess) {
            rc = fail(c, BPGPU_ERR_HIP, "staging failed");
            break;
        }
        rp_strobe_init init;
        memset(&init, 0, sizeof init);
        if (!per_proof) {   // one start state: innerproduct_domain_sep(n) replayed once on the host
            uint8_t st0[BPGPU_TRANSCRIPT_BYTES];
            ipp_domain_sep_state(st0, label, label_len, transcripts, n, &init);
        }
        const uint32_t nb32 = (uint32_t)nbatch;
        LAUNCH(c, s, "ipp_vs_front", k_ipp_vs_front, (nb32 + RP_BLOCK - 1) / RP_BLOCK, RP_BLOCK, sh, init, (const uint8_t *)d_pr, per_proof ? (const uint32_t *)d_ti : (const uint32_t *)nullptr,
               (uint32_t *)d_us, (uint32_t *)d_ui, (uint32_t *)d_tab, transcripts_out ? (uint32_t *)d_to : (uint32_t *)nullptr, (uint32_t *)d_st);
        if (n_eff) {
            const uint32_t nt = (uint32_t)(n_eff * nbatch);
            LAUNCH(c, s, "ipp_vs_s", k_ipp_vs_s, (nt + 63) / 64, 64, nt, sh, (const uint32_t *)d_tab, (const uint32_t *)d_st, (uint32_t *)d_s);
        }
        if (hipMemcpyAsync(h + sz_in, d_us, sz_out, hipMemcpyDeviceToHost, s) != hipSuccess) rc = fail(c, BPGPU_ERR_HIP, "D2H copy failed");
    } while (0);
    const int rc2 = ctx_leave(c, s), rc3 = host_wait(c, s);
    if (rc || rc2 || rc3) return rc ? rc : (rc2 ? rc2 : rc3);
    const char *ho = h + sz_in;
    memcpy(u_sq, ho, nbatch * k * 32);
    memcpy(u_inv_sq, ho + sz_u, nbatch * k * 32);
    if (n_eff) memcpy(s_out, ho + 2 * sz_u, nbatch * n_eff * 32);
    if (transcripts_out) memcpy(transcripts_out, ho + 2 * sz_u + sz_s, nbatch * TS);
    const uint32_t *st32 = (const uint32_t *)(ho + 2 * sz_u + sz_s + sz_to);
    for (size_t b = 0; b < nbatch; b++) {
        status[b] = (uint8_t)st32[b];
        // (with one start state for the batch the device replays from the state BEHIND the domain separator: where the reference never
        // got that far, the caller's state goes back as it came)
        if (status[b] == BPGPU_VERDICT_FORMAT_ERROR || shape_bad) untouched(b);
    }
    return BPGPU_OK;
}

extern "C" int bpgpu_ipp_verify_batch(bpgpu_ctx *c, size_t n, size_t nbatch, const uint8_t *proofs, size_t proof_len, const uint8_t *label,
                                      size_t label_len, const uint8_t *G_factors, const uint8_t *H_factors, const uint8_t *P,
                                      const uint8_t *Q, const uint8_t *G, const uint8_t *H, uint8_t *verdict, uint8_t *msm_out) {
    if (!c || (label_len && !label)) return BPGPU_ERR_INVALID_ARG;
    if (nbatch == 0) return BPGPU_OK;
    if (!proofs || !verdict || !P || !Q || (n && (!G_factors || !H_factors || !G || !H))) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t sz_pr = align_up(nbatch * proof_len + 64), sz_f = align_up(nbatch * n * 32 + 64), sz_pq = align_up(nbatch * 32 + 64);
    const size_t sz_in = sz_pr + 4 * sz_f + 2 * sz_pq, sz_v = align_up(nbatch), sz_o = align_up(nbatch * 32);
    hipStream_t s = c->stream;
    int rc = ctx_enter(c, s);
    if (rc) return rc;
    rc = io_reserve(c, sz_in + sz_v + sz_o);
    if (rc) return rc;
    char *h = nullptr;
    rc = pin_alloc(c, s, sz_in + sz_v + sz_o, &h);
    if (rc) return rc;
    char *d = c->io_dev;
    char *d_pr = d, *d_gf = d_pr + sz_pr, *d_hf = d_gf + sz_f, *d_g = d_hf + sz_f, *d_h = d_g + sz_f, *d_p = d_h + sz_f, *d_q = d_p + sz_pq;
    char *d_v = d + sz_in, *d_o = d_v + sz_v;
    memcpy(h, proofs, nbatch * proof_len);
    if (n) {
        memcpy(h + sz_pr, G_factors, nbatch * n * 32);
        memcpy(h + sz_pr + sz_f, H_factors, nbatch * n * 32);
        memcpy(h + sz_pr + 2 * sz_f, G, nbatch * n * 32);
        memcpy(h + sz_pr + 3 * sz_f, H, nbatch * n * 32);
    }
    memcpy(h + sz_pr + 4 * sz_f, P, nbatch * 32);
    memcpy(h + sz_pr + 4 * sz_f + sz_pq, Q, nbatch * 32);
    HIPCHK(c, hipMemcpyAsync(d, h, sz_in, hipMemcpyHostToDevice, s));
    rc = ipp_verify_dev_locked(c, n, nbatch, d_pr, proof_len, label, label_len, nullptr, d_gf, d_hf, d_p, d_q, d_g, d_h, 0, d_v, d_o, s);
    char *h_out = h + sz_in;
    if (!rc && hipMemcpyAsync(h_out, d_v, sz_v + sz_o, hipMemcpyDeviceToHost, s) != hipSuccess) rc = fail(c, BPGPU_ERR_HIP, "D2H copy failed");
    const int rc2 = ctx_leave(c, s), rc3 = host_wait(c, s);
    if (rc || rc2 || rc3) return rc ? rc : (rc2 ? rc2 : rc3);
    memcpy(verdict, h_out, nbatch);
    if (msm_out) memcpy(msm_out, h_out + sz_v, nbatch * 32);
    return BPGPU_OK;
}

// LinearProof::verify for nbatch proofs (linear.h): front end (lane = proof) -> bpgpu_msm_batch's MSM -> verdicts.
// G (n encodings), F, B are shared by the batch; b is per proof unless b_shared.
static int lin_verify_dev_locked(bpgpu_ctx *c, size_t n, size_t nbatch, const void *d_proofs, size_t proof_len, const uint8_t *label,
                                 size_t label_len, const uint8_t *shared_ts, const void *d_C, const void *d_G, const void *d_F,
                                 const void *d_B, const void *d_b, int b_shared, void *d_verdict, void *d_msm_out, void *d_ts_out,
                                 hipStream_t s) {
    if (nbatch > 0x7fffffffu / 64) return fail(c, BPGPU_ERR_INVALID_ARG, "batch too large");
    if (shared_ts && !ts_state_ok(shared_ts)) return fail(c, BPGPU_ERR_INVALID_ARG, "malformed transcript state");
    // LinearProof::from_bytes, length part (linear_proof.rs:350-366)
    size_t k = 0;
    bool fmt = false;
    if (proof_len % 32 != 0) fmt = true;
    else {
        const size_t ne = proof_len / 32;
        if (ne < 3 || (ne - 3) % 2 != 0) fmt = true;
        else {
            k = (ne - 3) / 2;
            if (k >= 32) fmt = true;
        }
    }
    if (fmt) {
        if (d_ts_out) return fail(c, BPGPU_ERR_INVALID_ARG, "proof_len is not a LinearProof length: no transcripts to return");
        HIPCHK(c, hipMemsetAsync(d_verdict, BPGPU_VERDICT_FORMAT_ERROR, nbatch, s));
        if (d_msm_out) HIPCHK(c, hipMemsetAsync(d_msm_out, 0, nbatch * 32, s));
        return BPGPU_OK;
    }
    lin_shape sh;
    sh.n = (uint32_t)n;
    sh.k = (uint32_t)k;
    sh.proof_len = (uint32_t)proof_len;
    sh.nproofs = (uint32_t)nbatch;
    sh.b_shared = b_shared ? 1u : 0u;
    if (n == ((size_t)1 << k) && k > BP_RP_MAX_K) return fail(c, BPGPU_ERR_INVALID_ARG, "n > 2^%d not supported", BP_RP_MAX_K);
    sh.shape_verdict = (n == ((size_t)1 << k)) ? 0 : BPGPU_VERDICT_VERIFICATION_ERROR;   // linear_proof.rs:263-265
    if (sh.shape_verdict) sh.n = 0;   // only the canonical-scalar check runs
    // generator-table mode: G, F, B not given = the context's bp_gens.share(0).G(n), pc_gens.B, pc_gens.B_blinding (what the
    // reference's own callers pass, linear_proof.rs:405-411): their n + 2 coefficients go through the window tables
    const bool fixed = d_G == nullptr && d_F == nullptr && d_B == nullptr;
    if (fixed) {
        if (!c->d_table) return fail(c, BPGPU_ERR_NO_GENS, "generators not loaded");
        if (!sh.shape_verdict && n > c->gens_capacity) return fail(c, BPGPU_ERR_NO_GENS, "InvalidGeneratorsLength: generators too small for n=%zu", n);
        d_G = c->d_gens + 2 * 8;
        d_F = c->d_gens + 8;
        d_B = c->d_gens;
    }
    sh.fixed = fixed ? 1u : 0u;
    sh.N = (uint32_t)(fixed ? 2 * k + 2 : sh.shape_verdict ? 4 : n + 2 * k + 4);
    const size_t N = sh.N;
    if ((uint64_t)nbatch * (n + 2 * k + 4) > 0x7fffffffull / 64) return fail(c, BPGPU_ERR_INVALID_ARG, "batch too large for this shape");
    const size_t sz_terms = align_up(nbatch * N * 32 + 64), sz_st = align_up(nbatch * 4), sz_b = align_up(nbatch + 64), sz_o = align_up(nbatch * 32 + 64);
    const size_t sz_gen = fixed ? align_up(nbatch * (sh.n + 2) * 32 + 64) : 0;
    const size_t need = 2 * sz_terms + sz_st + sz_b + sz_o + sz_gen;
    {
        const int rcr = ipp_reserve(c, need);
        if (rcr) return rcr;
    }
    char *d_sc = c->ipp_buf, *d_pt = d_sc + sz_terms, *d_stat = d_pt + sz_terms, *d_mst = d_stat + sz_st, *d_out = d_mst + sz_b;
    char *d_gen = fixed ? d_out + sz_o : nullptr;
    HIPCHK(c, hipMemsetAsync(d_sc, 0, 2 * sz_terms + sz_st, s));   // scalars, points, status
    if (fixed) HIPCHK(c, hipMemsetAsync(d_gen, 0, sz_gen, s));
    rp_strobe_init init;
    {
        uint8_t st0[BPGPU_TRANSCRIPT_BYTES];
        ipp_domain_sep_state(st0, label, label_len, shared_ts, n, &init);
    }
    const uint32_t nb32 = (uint32_t)nbatch;
    LAUNCH(c, s, "lin_prepare", k_lin_prepare, (nb32 + RP_BLOCK - 1) / RP_BLOCK, RP_BLOCK, sh, init, (const uint8_t *)d_proofs, (const uint8_t *)d_C,
           (const uint8_t *)d_b, (const uint8_t *)d_G, (const uint8_t *)d_F, (const uint8_t *)d_B, (uint32_t *)d_sc, (uint32_t *)d_pt,
           (uint32_t *)d_stat, (uint32_t *)d_ts_out, (uint32_t *)d_gen);
    int rc;
    if (fixed && sh.shape_verdict) {      // nothing to multiply: every proof already carries its verdict
        HIPCHK(c, hipMemsetAsync(d_mst, 0, sz_b + sz_o, s));
        rc = BPGPU_OK;
    } else if (fixed) {
        rc = msm_shared_dev_locked(c, n, 1, nbatch, N, d_gen, d_sc, d_pt, d_out, d_mst, nullptr, s, true);
    } else {
        std::vector<uint32_t> nt(nbatch, (uint32_t)N);
        rc = msm_batch_dev_locked(c, nbatch, nt.data(), d_sc, d_pt, d_out, d_mst, s);
    }
    if (rc) return rc;
    LAUNCH(c, s, "lin_verdict", k_ipp_verdict, (nb32 + 63) / 64, 64, nb32, (const uint32_t *)d_stat, (const uint8_t *)d_mst, (const uint32_t *)d_out,
           (uint8_t *)d_verdict);
    if (d_msm_out) HIPCHK(c, hipMemcpyAsync(d_msm_out, d_out, nbatch * 32, hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipGetLastError());
    return BPGPU_OK;
}

extern "C" int bpgpu_linear_verify_batch_dev(bpgpu_ctx *c, size_t n, size_t nbatch, const void *d_proofs, size_t proof_len, const uint8_t *label,
                                             size_t label_len, const uint8_t *shared_transcript, const void *d_C, const void *d_G,
                                             const void *d_F, const void *d_B, const void *d_b, int b_shared, void *d_verdict, void *d_msm_out,
                                             void *d_transcripts_out, void *stream) {
    if (!c || (label_len && !label)) return BPGPU_ERR_INVALID_ARG;
    if (nbatch == 0) return BPGPU_OK;
    const bool from_gens = !d_G && !d_F && !d_B;
    if (!d_proofs || !d_verdict || !d_C || (n && !d_b) || (!from_gens && (!d_F || !d_B || (n && !d_G)))) return BPGPU_ERR_INVALID_ARG;
    if (((uintptr_t)d_proofs | (uintptr_t)d_C | (uintptr_t)d_G | (uintptr_t)d_F | (uintptr_t)d_B | (uintptr_t)d_b | (uintptr_t)d_transcripts_out) & 3)
        return fail(c, BPGPU_ERR_INVALID_ARG, "device buffers must be 4-byte aligned");
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    int rc = ctx_enter(c, s);
    if (rc) return rc;
    rc = lin_verify_dev_locked(c, n, nbatch, d_proofs, proof_len, label, label_len, shared_transcript, d_C, d_G, d_F, d_B, d_b, b_shared, d_verdict,
                               d_msm_out, d_transcripts_out, s);
    const int rc2 = ctx_leave(c, s);
    return rc ? rc : rc2;
}

extern "C" int bpgpu_linear_verify_batch(bpgpu_ctx *c, size_t n, size_t nbatch, const uint8_t *proofs, size_t proof_len, const uint8_t *label,
                                         size_t label_len, const uint8_t *shared_transcript, const uint8_t *C, const uint8_t *G, const uint8_t *F,
                                         const uint8_t *B, const uint8_t *b, int b_shared, uint8_t *verdict, uint8_t *msm_out,
                                         uint8_t *transcripts_out) {
    if (!c || (label_len && !label)) return BPGPU_ERR_INVALID_ARG;
    if (nbatch == 0) return BPGPU_OK;
    const bool from_gens = !G && !F && !B;
    if (!proofs || !verdict || !C || (n && !b) || (!from_gens && (!F || !B || (n && !G)))) return BPGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nb_b = b_shared ? 1 : nbatch;
    const size_t sz_pr = align_up(nbatch * proof_len + 64), sz_c = align_up(nbatch * 32 + 64), sz_g = align_up(n * 32 + 64), sz_fb = align_up(64 + 64),
                 sz_bv = align_up(nb_b * n * 32 + 64);
    const size_t sz_in = sz_pr + sz_c + sz_g + sz_fb + sz_bv, sz_v = align_up(nbatch), sz_o = align_up(nbatch * 32);
    const size_t sz_t = transcripts_out ? align_up(nbatch * BPGPU_TRANSCRIPT_BYTES) : 0;
    hipStream_t s = c->stream;
    int rc = ctx_enter(c, s);
    if (rc) return rc;
    rc = io_reserve(c, sz_in + sz_v + sz_o + sz_t);
    if (rc) return rc;
    char *h = nullptr;
    rc = pin_alloc(c, s, sz_in + sz_v + sz_o + sz_t, &h);
    if (rc) return rc;
    char *d = c->io_dev;
    char *d_pr = d, *d_c = d_pr + sz_pr, *d_g = d_c + sz_c, *d_fb = d_g + sz_g, *d_bv = d_fb + sz_fb, *d_v = d + sz_in, *d_o = d_v + sz_v;
    char *d_t = transcripts_out ? d_o + sz_o : nullptr;
    memcpy(h, proofs, nbatch * proof_len);
    memcpy(h + sz_pr, C, nbatch * 32);
    if (n) {
        if (!from_gens) memcpy(h + sz_pr + sz_c, G, n * 32);
        memcpy(h + sz_pr + sz_c + sz_g + sz_fb, b, nb_b * n * 32);
    }
    if (!from_gens) {
        memcpy(h + sz_pr + sz_c + sz_g, F, 32);
        memcpy(h + sz_pr + sz_c + sz_g + 32, B, 32);
    }
    HIPCHK(c, hipMemcpyAsync(d, h, sz_in, hipMemcpyHostToDevice, s));
    rc = lin_verify_dev_locked(c, n, nbatch, d_pr, proof_len, label, label_len, shared_transcript, d_c, from_gens ? nullptr : d_g,
                               from_gens ? nullptr : d_fb, from_gens ? nullptr : d_fb + 32, d_bv, b_shared, d_v, d_o, d_t, s);
    char *h_out = h + sz_in;
    if (!rc && hipMemcpyAsync(h_out, d_v, sz_v + sz_o + sz_t, hipMemcpyDeviceToHost, s) != hipSuccess) rc = fail(c, BPGPU_ERR_HIP, "D2H copy failed");
    const int rc2 = ctx_leave(c, s), rc3 = host_wait(c, s);
    if (rc || rc2 || rc3) return rc ? rc : (rc2 ? rc2 : rc3);
    memcpy(verdict, h_out, nbatch);
    if (msm_out) memcpy(msm_out, h_out + sz_v, nbatch * 32);
    if (transcripts_out) memcpy(transcripts_out, h_out + sz_v + sz_o, nbatch * BPGPU_TRANSCRIPT_BYTES);
    return BPGPU_OK;
}

// ProofShare::audit_share for nshares shares of bitsize n (audit.h): front end (lane = share) -> the ragged (2n + 3, 5) pairs of
// multiscalar multiplications through bpgpu_msm_batch's engine -> Ok / Err per share.  Host pointers (the dealer's blame path).
extern "C" int bpgpu_rangeproof_audit_shares(bpgpu_ctx *c, size_t n, size_t nshares, const uint32_t *party_index, const uint8_t *shares,
                                             const uint8_t *bit_commitments, const uint8_t *poly_commitments, const uint8_t *challenges,
                                             int challenges_shared, uint8_t *verdict, uint8_t *checks_out) {
    if (!c) return BPGPU_ERR_INVALID_ARG;
    if (nshares == 0) return BPGPU_OK;
    if (!party_index || !shares || !bit_commitments || !poly_commitments || !challenges || !verdict) return BPGPU_ERR_INVALID_ARG;
    if (!(n == 8 || n == 16 || n == 32 || n == 64)) return fail(c, BPGPU_ERR_INVALID_ARG, "InvalidBitsize: n must be 8, 16, 32 or 64 (party.rs:41-43)");
    if ((uint64_t)nshares * (2 * n + 8) > 0x7fffffffull / 64) return fail(c, BPGPU_ERR_INVALID_ARG, "batch too large for this shape");
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->d_gens) return fail(c, BPGPU_ERR_NO_GENS, "generators not loaded");
    if (n > c->gens_capacity) {   // check_size (messages.rs:70-72): Err(()) for every share
        memset(verdict, BPGPU_VERDICT_VERIFICATION_ERROR, nshares);
        if (checks_out) memset(checks_out, 0, nshares * 64);
        return BPGPU_OK;
    }
    hipStream_t s = c->stream;
    int rc = ctx_enter(c, s);
    if (rc) return rc;
    const size_t share_len = 32 * (3 + 2 * n), nch = challenges_shared ? 1 : nshares;
    const size_t sz_pi = align_up(nshares * 4 + 64), sz_sh = align_up(nshares * share_len + 64), sz_bc = align_up(nshares * 96 + 64),
                 sz_pc = align_up(nshares * 64 + 64), sz_ch = align_up(nch * 96 + 64);
    const size_t sz_in = sz_pi + sz_sh + sz_bc + sz_pc + sz_ch, sz_v = align_up(nshares), sz_o = align_up(nshares * 64);
    rc = io_reserve(c, sz_in + sz_v + sz_o);
    if (rc) return rc;
    char *h = nullptr;
    rc = pin_alloc(c, s, sz_in + sz_v + sz_o, &h);
    if (rc) return rc;
    char *d = c->io_dev;
    char *d_pi = d, *d_sh = d_pi + sz_pi, *d_bc = d_sh + sz_sh, *d_pc = d_bc + sz_bc, *d_ch = d_pc + sz_pc, *d_v = d + sz_in, *d_o = d_v + sz_v;
    memcpy(h, party_index, nshares * 4);
    memcpy(h + sz_pi, shares, nshares * share_len);
    memcpy(h + sz_pi + sz_sh, bit_commitments, nshares * 96);
    memcpy(h + sz_pi + sz_sh + sz_bc, poly_commitments, nshares * 64);
    memcpy(h + sz_pi + sz_sh + sz_bc + sz_pc, challenges, nch * 96);
    HIPCHK(c, hipMemcpyAsync(d, h, sz_in, hipMemcpyHostToDevice, s));
    do {
        const size_t per = 2 * n + 8;
        const size_t sz_terms = align_up(nshares * per * 32 + 64), sz_st = align_up(nshares * 4), sz_b = align_up(2 * nshares + 64);
        const size_t need = 2 * sz_terms + sz_st + sz_b;
        rc = ipp_reserve(c, need);
        if (rc) break;
        char *d_sc = c->ipp_buf, *d_pt = d_sc + sz_terms, *d_stat = d_pt + sz_terms, *d_mst = d_stat + sz_st;
        if (hipMemsetAsync(d_sc, 0, 2 * sz_terms + sz_st, s) != hipSuccess) { rc = fail(c, BPGPU_ERR_HIP, "memset failed"); break; }
        aud_shape sh;
        sh.n = (uint32_t)n;
        sh.lg_n = 0;
        while (((size_t)1 << sh.lg_n) < n) sh.lg_n++;
        sh.nshares = (uint32_t)nshares;
        sh.gens_capacity = (uint32_t)c->gens_capacity;
        sh.party_capacity = (uint32_t)c->party_capacity;
        sh.chal_shared = challenges_shared ? 1u : 0u;
        const uint32_t ns32 = (uint32_t)nshares;
        LAUNCH(c, s, "aud_prepare", k_aud_prepare, (ns32 + 63) / 64, 64, sh, (const uint32_t *)d_pi, (const uint8_t *)d_sh, (const uint8_t *)d_bc,
               (const uint8_t *)d_pc, (const uint8_t *)d_ch, (const uint32_t *)c->d_gens, (uint32_t *)d_sc, (uint32_t *)d_pt, (uint32_t *)d_stat);
        std::vector<uint32_t> nt(2 * nshares);
        for (size_t i = 0; i < nshares; i++) {
            nt[2 * i] = (uint32_t)(2 * n + 3);     // P_check (messages.rs:128-141)
            nt[2 * i + 1] = 5;                     // t_check (:149-160)
        }
        rc = msm_batch_dev_locked(c, 2 * nshares, nt.data(), d_sc, d_pt, d_o, d_mst, s);
        if (rc) break;
        LAUNCH(c, s, "aud_verdict", k_aud_verdict, (ns32 + 63) / 64, 64, ns32, (const uint32_t *)d_stat, (const uint8_t *)d_mst, (const uint32_t *)d_o,
               (uint8_t *)d_v);
        if (hipGetLastError() != hipSuccess) rc = fail(c, BPGPU_ERR_HIP, "launch failed");
    } while (0);
    char *h_out = h + sz_in;
    if (!rc && hipMemcpyAsync(h_out, d_v, sz_v + sz_o, hipMemcpyDeviceToHost, s) != hipSuccess) rc = fail(c, BPGPU_ERR_HIP, "D2H copy failed");
    const int rc2 = ctx_leave(c, s), rc3 = host_wait(c, s);
    if (rc || rc2 || rc3) return rc ? rc : (rc2 ? rc2 : rc3);
    memcpy(verdict, h_out, nshares);
    if (checks_out) memcpy(checks_out, h_out + sz_v, nshares * 64);
    return BPGPU_OK;
}

// The rounds of InnerProductProof::create for nbatch proofs (ipp_prover.h): inputs and outputs in device memory.
// d_ts: the proofs' transcript states AFTER innerproduct_domain_sep(n), advanced in place.  The k (L, R) pairs and the
// final a, b go to d_proofs + p * proof_stride (+ 64 j, + 64 k); status_bytes (optional): BPGPU_MSM_* per proof.
// `fixed` (optional): G, H are the context's G(gn, gm), H(gn, gm) (n = gn gm) and Q = w B with w[p] given: every L_j / R_j
// is a pure generator-table MSM (msm_shared's walk) instead of a variable-base one -- no point decoding, no Horner chain.
struct ippc_fixed {
    size_t gn, gm;
    const uint32_t *w;       // [nbatch][8 words]
    uint32_t *gen_scalars;   // scratch, >= 2 nbatch (2n + 2) scalars
};
static int ippc_core(bpgpu_ctx *c, hipStream_t s, size_t n, size_t k, size_t nbatch, const void *d_a, const void *d_b, const void *d_gf, const void *d_hf,
                     const void *d_q, const void *d_G, const void *d_H, int bases_shared, uint32_t *d_ts, uint8_t *d_proofs, size_t proof_stride,
                     uint8_t *d_status_bytes, const ippc_fixed *fixed = nullptr) {
    const size_t N = n + 1, w_v = align_up(nbatch * n * 32), w_u = align_up(nbatch * 32), w_terms = align_up(2 * nbatch * N * 32 + 64),
                 w_out = align_up(2 * nbatch * 32), w_st = align_up(2 * nbatch + 64), w_status = align_up(nbatch * 4);
    const size_t need = 4 * w_v + 2 * w_u + 2 * w_terms + w_out + w_st + w_status;
    {
        const int rcr = ipp_reserve(c, need);
        if (rcr) return rcr;
    }
    char *wb = c->ipp_buf;
    uint32_t *w_a = (uint32_t *)wb, *w_b = (uint32_t *)(wb + w_v), *w_G = (uint32_t *)(wb + 2 * w_v), *w_H = (uint32_t *)(wb + 3 * w_v);
    uint32_t *w_uu = (uint32_t *)(wb + 4 * w_v), *w_ui = (uint32_t *)(wb + 4 * w_v + w_u);
    uint32_t *m_sc = (uint32_t *)(wb + 4 * w_v + 2 * w_u), *m_pt = (uint32_t *)((char *)m_sc + w_terms), *m_out = (uint32_t *)((char *)m_pt + w_terms);
    uint8_t *m_st = (uint8_t *)m_out + w_out;
    uint32_t *d_status = (uint32_t *)(m_st + w_st);
    HIPCHK(c, hipMemsetAsync(d_status, 0, nbatch * 4, s));
    ippc_shape sh;
    sh.n = (uint32_t)n;
    sh.k = (uint32_t)k;
    sh.nproofs = (uint32_t)nbatch;
    sh.bases_shared = bases_shared ? 1u : 0u;
    const uint32_t nb32 = (uint32_t)nbatch, nt32 = (uint32_t)(nbatch * n);
    LAUNCH(c, s, "ippc_init", k_ippc_init, (nt32 + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, nt32, sh, (const uint8_t *)d_a, (const uint8_t *)d_b, (const uint8_t *)d_gf,
           (const uint8_t *)d_hf, w_a, w_b, w_G, w_H, d_status);
    std::vector<uint32_t> nterms(2 * nbatch, (uint32_t)N);
    const uint32_t n_q = (nb32 + BP_BLOCK - 1) / BP_BLOCK;
    for (uint32_t j = 0; j < k; j++) {
        int rc;
        if (fixed) {
            LAUNCH(c, s, "ippc_terms", k_ippc_terms_fixed, n_q + (nt32 + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, n_q, nt32, sh, j, (const uint32_t *)w_a,
                   (const uint32_t *)w_b, (const uint32_t *)w_G, (const uint32_t *)w_H, fixed->w, fixed->gen_scalars);
            rc = msm_shared_dev_locked(c, fixed->gn, fixed->gm, 2 * nbatch, 0, fixed->gen_scalars, nullptr, nullptr, m_out, m_st, nullptr, s);
        } else {
            LAUNCH(c, s, "ippc_terms", k_ippc_terms, n_q + (nt32 + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, n_q, nt32, sh, j, (const uint32_t *)w_a, (const uint32_t *)w_b,
                   (const uint32_t *)w_G, (const uint32_t *)w_H, (const uint8_t *)d_G, (const uint8_t *)d_H, (const uint8_t *)d_q, m_sc, m_pt);
            rc = msm_batch_dev_locked(c, 2 * nbatch, nterms.data(), m_sc, m_pt, m_out, m_st, s);   // all L_j and R_j of the batch (ipp.rs:87-113)
        }
        if (rc) return rc;   // (callers drain the stream before they reuse their staging buffers)
        LAUNCH(c, s, "ippc_challenge", k_ippc_challenge, (nb32 + RP_BLOCK - 1) / RP_BLOCK, RP_BLOCK, sh, j, (const uint32_t *)m_out, (const uint8_t *)m_st,
               d_ts, w_uu, w_ui, d_proofs, (uint32_t)proof_stride, d_status);
        LAUNCH(c, s, "ippc_fold", k_ippc_fold, (nt32 + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, nt32, sh, j, (const uint32_t *)w_uu, (const uint32_t *)w_ui, w_a, w_b, w_G,
               w_H);
    }
    LAUNCH(c, s, "ippc_final", k_ippc_final, (nb32 + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, sh, (const uint32_t *)w_a, (const uint32_t *)w_b, d_proofs,
           (uint32_t)proof_stride, (const uint32_t *)d_status, d_status_bytes);
    HIPCHK(c, hipGetLastError());
    return BPGPU_OK;
}

// ============================================================================
// batched inner-product-proof creation (ipp_prover.h)
// ============================================================================
// ---- the common exit of the prover entry points -------------------------------------------------------------------
// The provers stage SECRETS (values, blindings, the witness vectors, every random scalar) in the context's persistent buffers:
// the pinned staging block, the device IO buffer, their working sets and -- as window digits of secret scalars -- the arena.
// The reference zeroizes its party state on Drop (party.rs:148-260, 308+); here every entered call, successful or not, leaves
// through prover_exit::finish: the device buffers are cleared on the stream behind the call's last copy, the call's end is
// recorded (ctx_leave), the stream is drained (host_wait), then the secret part of the staging block is cleared.  What is NOT
// cleared: registers / LDS of finished kernels and the caller's own buffers.
struct prover_exit {
    bpgpu_ctx *c;
    hipStream_t s;
    char *h = nullptr;             // the call's pinned staging block (set once allocated)
    size_t host_secret_bytes = 0;  // leading bytes of h that hold secrets
    int finish(int rc) {
        bool ok = true;
        if (c->io_dev) ok = hipMemsetAsync(c->io_dev, 0, c->io_cap, s) == hipSuccess && ok;
        if (c->rpp_buf) ok = hipMemsetAsync(c->rpp_buf, 0, c->rpp_cap, s) == hipSuccess && ok;
        if (c->ipp_buf) ok = hipMemsetAsync(c->ipp_buf, 0, c->ipp_cap, s) == hipSuccess && ok;
        if (c->arena) ok = hipMemsetAsync(c->arena, 0, c->arena_cap, s) == hipSuccess && ok;
        const int rc2 = ctx_leave(c, s), rc3 = host_wait(c, s);
        if (h && host_secret_bytes) memset(h, 0, host_secret_bytes);
        c->last_secret_bytes = (h == c->pin) ? host_secret_bytes : 0;
        if (!rc && !ok) rc = fail(c, BPGPU_ERR_HIP, "clearing the prover's working set failed");
        return rc ? rc : (rc2 ? rc2 : rc3);
    }
};
#define PX_HIPCHK(px, call)                                                                                          \
    do {                                                                                                             \
        hipError_t e_ = (call);                                                                                      \
        if (e_ != hipSuccess) return (px).finish(fail(c, BPGPU_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_))); \
    } while (0)

extern "C" int bpgpu_ipp_create_batch(bpgpu_ctx *c, size_t n, size_t nbatch, const uint8_t *label, size_t label_len, const uint8_t *shared_transcript,
                                      const uint8_t *Q, const uint8_t *G_factors, const uint8_t *H_factors, const uint8_t *G, const uint8_t *H,
                                      int bases_shared, const uint8_t *a, const uint8_t *b, uint8_t *proofs_out, uint8_t *status_out) {
    if (!c || (label_len && !label)) return BPGPU_ERR_INVALID_ARG;
    if (nbatch == 0) return BPGPU_OK;
    if (!Q || !G_factors || !H_factors || !G || !H || !a || !b || !proofs_out || !status_out) return BPGPU_ERR_INVALID_ARG;
    if (n == 0 || (n & (n - 1))) return fail(c, BPGPU_ERR_INVALID_ARG, "n must be a power of two (ipp.rs:54-59)");
    size_t k = 0;
    while (((size_t)1 << k) < n) k++;
    if (k > BP_RP_MAX_K) return fail(c, BPGPU_ERR_INVALID_ARG, "n > 2^%d not supported", BP_RP_MAX_K);
    if ((uint64_t)nbatch * 2 * (n + 1) > 0x7fffffffull / 64) return fail(c, BPGPU_ERR_INVALID_ARG, "batch too large for this shape");
    if (shared_transcript && !ts_state_ok(shared_transcript)) return fail(c, BPGPU_ERR_INVALID_ARG, "malformed transcript state");
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    int rc = ctx_enter(c, s);
    if (rc) return rc;
    prover_exit px{c, s};
    const size_t proof_len = 32 * (2 * k + 2), TS = BPGPU_TRANSCRIPT_BYTES;
    // ---- inputs: one pinned staging block -> the persistent device IO buffer
    const size_t sz_v = align_up(nbatch * n * 32 + 64), sz_q = align_up(nbatch * 32 + 64), sz_base = align_up((bases_shared ? 1 : nbatch) * n * 32 + 64),
                 sz_ts = align_up(nbatch * TS);
    const size_t sz_in = 4 * sz_v + sz_q + 2 * sz_base + sz_ts, sz_out = align_up(nbatch * proof_len) + align_up(nbatch);
    rc = io_reserve(c, sz_in + sz_out);
    if (rc) return px.finish(rc);
    char *h = nullptr;
    rc = pin_alloc(c, s, sz_in + sz_out, &h);
    if (rc) return px.finish(rc);
    px.h = h;
    px.host_secret_bytes = 2 * sz_v;   // the witness a, b
    char *d_a = c->io_dev, *d_b = d_a + sz_v, *d_gf = d_b + sz_v, *d_hf = d_gf + sz_v, *d_q = d_hf + sz_v, *d_G = d_q + sz_q, *d_H = d_G + sz_base,
         *d_ts = d_H + sz_base, *d_proofs = c->io_dev + sz_in, *d_stb = d_proofs + align_up(nbatch * proof_len);
    memcpy(h, a, nbatch * n * 32);
    memcpy(h + sz_v, b, nbatch * n * 32);
    memcpy(h + 2 * sz_v, G_factors, nbatch * n * 32);
    memcpy(h + 3 * sz_v, H_factors, nbatch * n * 32);
    memcpy(h + 4 * sz_v, Q, nbatch * 32);
    memcpy(h + 4 * sz_v + sz_q, G, (bases_shared ? 1 : nbatch) * n * 32);
    memcpy(h + 4 * sz_v + sz_q + sz_base, H, (bases_shared ? 1 : nbatch) * n * 32);
    {   // every proof's transcript after innerproduct_domain_sep(n) (transcript.rs:50-53)
        uint8_t st0[BPGPU_TRANSCRIPT_BYTES];
        ipp_domain_sep_state(st0, label, label_len, shared_transcript, n, nullptr);
        for (size_t p = 0; p < nbatch; p++) memcpy(h + 4 * sz_v + sz_q + 2 * sz_base + p * TS, st0, TS);
    }
    PX_HIPCHK(px, hipMemcpyAsync(c->io_dev, h, sz_in, hipMemcpyHostToDevice, s));
    PX_HIPCHK(px, hipMemsetAsync(d_proofs, 0, nbatch * proof_len, s));
    // ---- rounds (ippc_core: own working set, the batched MSMs claim the arena)
    rc = ippc_core(c, s, n, k, nbatch, d_a, d_b, d_gf, d_hf, d_q, d_G, d_H, bases_shared, (uint32_t *)d_ts, (uint8_t *)d_proofs, proof_len, (uint8_t *)d_stb);
    char *h_out = h + sz_in;
    if (!rc && hipMemcpyAsync(h_out, d_proofs, sz_out, hipMemcpyDeviceToHost, s) != hipSuccess) rc = fail(c, BPGPU_ERR_HIP, "D2H copy failed");
    rc = px.finish(rc);
    if (rc) return rc;
    memcpy(proofs_out, h_out, nbatch * proof_len);
    memcpy(status_out, h_out + align_up(nbatch * proof_len), nbatch);
    return BPGPU_OK;
}

// ============================================================================
// batched linear-proof creation (linear_prover.h)
// ============================================================================
extern "C" int bpgpu_linear_create_batch(bpgpu_ctx *c, size_t n, size_t nbatch, const uint8_t *label, size_t label_len, const uint8_t *shared_transcript,
                                         const uint8_t *rng, const uint8_t *C, const uint8_t *r, const uint8_t *a, const uint8_t *b, int b_shared,
                                         const uint8_t *G, const uint8_t *F, const uint8_t *B, uint8_t *proofs_out, uint8_t *status_out,
                                         uint8_t *transcripts_out) {
    if (!c || (label_len && !label)) return BPGPU_ERR_INVALID_ARG;
    if (nbatch == 0) return BPGPU_OK;
    const bool from_gens = !G && !F && !B;   // bases = the context's generators: every MSM through the window tables
    if (!C || !r || !a || !b || (!from_gens && (!G || !F || !B)) || !proofs_out || !status_out) return BPGPU_ERR_INVALID_ARG;
    if (n == 0 || (n & (n - 1))) return fail(c, BPGPU_ERR_INVALID_ARG, "InvalidInputLength: n must be a power of two (linear_proof.rs:68-70)");
    size_t k = 0;
    while (((size_t)1 << k) < n) k++;
    if (k > BP_RP_MAX_K) return fail(c, BPGPU_ERR_INVALID_ARG, "n > 2^%d not supported", BP_RP_MAX_K);
    if ((uint64_t)nbatch * (n + 4) > 0x7fffffffull / 64) return fail(c, BPGPU_ERR_INVALID_ARG, "batch too large for this shape");
    if (shared_transcript && !ts_state_ok(shared_transcript)) return fail(c, BPGPU_ERR_INVALID_ARG, "malformed transcript state");
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    if (from_gens) {
        if (!c->d_table) return fail(c, BPGPU_ERR_NO_GENS, "generators not loaded");
        if (n > c->gens_capacity) return fail(c, BPGPU_ERR_NO_GENS, "InvalidGeneratorsLength: generators too small for n=%zu", n);
    }
    int rc = ctx_enter(c, s);
    if (rc) return rc;
    prover_exit px{c, s};
    const size_t proof_len = 32 * (2 * k + 3), TS = BPGPU_TRANSCRIPT_BYTES, nd = 2 * k + 2, nb_b = b_shared ? 1 : nbatch;
    // ---- inputs: one pinned staging block -> the persistent device IO buffer
    const size_t sz_v = align_up(nbatch * n * 32 + 64), sz_bv = align_up(nb_b * n * 32 + 64), sz_q = align_up(nbatch * 32 + 64), sz_g = align_up(n * 32 + 64),
                 sz_fb = align_up(64 + 64), sz_rng = align_up(nbatch * nd * 64), sz_ts = align_up(nbatch * TS);
    const size_t sz_in = sz_v + sz_bv + 2 * sz_q + sz_g + sz_fb + sz_rng + sz_ts, sz_pr = align_up(nbatch * proof_len), sz_stb = align_up(nbatch);
    const size_t sz_out = sz_pr + sz_stb;
    rc = io_reserve(c, sz_in + sz_out);
    if (rc) return px.finish(rc);
    char *h = nullptr;
    rc = pin_alloc(c, s, sz_in + sz_out, &h);
    if (rc) return px.finish(rc);
    px.h = h;
    px.host_secret_bytes = sz_in - sz_ts;   // a, r, the random draws (and the public inputs between them); the transcript block behind is reused for the way back
    char *d_a = c->io_dev, *d_b = d_a + sz_v, *d_C = d_b + sz_bv, *d_r = d_C + sz_q, *d_G = d_r + sz_q, *d_fb = d_G + sz_g, *d_rng = d_fb + sz_fb,
         *d_ts = d_rng + sz_rng, *d_proofs = c->io_dev + sz_in, *d_stb = d_proofs + sz_pr;
    memcpy(h, a, nbatch * n * 32);
    memcpy(h + sz_v, b, nb_b * n * 32);
    memcpy(h + sz_v + sz_bv, C, nbatch * 32);
    memcpy(h + sz_v + sz_bv + sz_q, r, nbatch * 32);
    if (!from_gens) {
        memcpy(h + sz_v + sz_bv + 2 * sz_q, G, n * 32);
        memcpy(h + sz_v + sz_bv + 2 * sz_q + sz_g, F, 32);
        memcpy(h + sz_v + sz_bv + 2 * sz_q + sz_g + 32, B, 32);
    }
    const uint8_t *e_G = from_gens ? (const uint8_t *)(c->d_gens + 16) : (const uint8_t *)d_G;          // encodings the transcript absorbs
    const uint8_t *e_F = from_gens ? (const uint8_t *)(c->d_gens + 8) : (const uint8_t *)d_fb;
    const uint8_t *e_B = from_gens ? (const uint8_t *)c->d_gens : (const uint8_t *)d_fb + 32;
    {
        char *h_rng = h + sz_v + sz_bv + 2 * sz_q + sz_g + sz_fb;
        if (rng) memcpy(h_rng, rng, nbatch * nd * 64);
        else if ((rc = os_random(c, h_rng, nbatch * nd * 64)) != 0) return px.finish(rc);   // Scalar::random(&mut thread_rng())
        // every proof's transcript after innerproduct_domain_sep(n) (linear_proof.rs:73)
        uint8_t st0[BPGPU_TRANSCRIPT_BYTES];
        ipp_domain_sep_state(st0, label, label_len, shared_transcript, n, nullptr);
        for (size_t p = 0; p < nbatch; p++) memcpy(h_rng + sz_rng + p * TS, st0, TS);
    }
    PX_HIPCHK(px, hipMemcpyAsync(c->io_dev, h, sz_in, hipMemcpyHostToDevice, s));
    PX_HIPCHK(px, hipMemsetAsync(d_proofs, 0, nbatch * proof_len, s));
    do {
        // ---- working set (own allocation: the batched MSMs claim the arena)
        const size_t N = n / 2 + 2, NS = n + 2;
        const size_t w_v = align_up(nbatch * n * 32), w_u = align_up(nbatch * 32), w_d = align_up(nbatch * nd * 32),
                     w_terms = align_up(std::max(2 * nbatch * (from_gens ? NS : N), nbatch * NS) * 32 + 64), w_out = align_up(2 * nbatch * 32), w_st = align_up(2 * nbatch + 64),
                     w_status = align_up(nbatch * 4);
        const size_t need = 3 * w_v + 3 * w_u + w_d + 2 * w_terms + w_out + w_st + w_status;
        rc = ipp_reserve(c, need);
        if (rc) break;
        char *wb = c->ipp_buf;
        uint32_t *w_a = (uint32_t *)wb, *w_b = (uint32_t *)(wb + w_v), *w_G = (uint32_t *)(wb + 2 * w_v);
        uint32_t *w_r = (uint32_t *)(wb + 3 * w_v), *w_x = (uint32_t *)((char *)w_r + w_u), *w_xi = (uint32_t *)((char *)w_x + w_u);
        uint32_t *w_dr = (uint32_t *)((char *)w_xi + w_u);
        uint32_t *m_sc = (uint32_t *)((char *)w_dr + w_d), *m_pt = (uint32_t *)((char *)m_sc + w_terms), *m_out = (uint32_t *)((char *)m_pt + w_terms);
        uint8_t *m_st = (uint8_t *)m_out + w_out;
        uint32_t *d_status = (uint32_t *)(m_st + w_st);
        if (hipMemsetAsync(d_status, 0, nbatch * 4, s) != hipSuccess) { rc = fail(c, BPGPU_ERR_HIP, "memset failed"); break; }
        linc_shape sh;
        sh.n = (uint32_t)n;
        sh.k = (uint32_t)k;
        sh.nproofs = (uint32_t)nbatch;
        sh.b_shared = b_shared ? 1u : 0u;
        const uint32_t nb32 = (uint32_t)nbatch, nt32 = (uint32_t)(nbatch * n), n_q = (nb32 + BP_BLOCK - 1) / BP_BLOCK;
        LAUNCH(c, s, "linc_init", k_linc_init, (nt32 + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, nt32, sh, (const uint8_t *)d_a, (const uint8_t *)d_b, w_a, w_b, w_G,
               d_status);
        LAUNCH(c, s, "linc_public", k_linc_public, (nb32 + RP_BLOCK - 1) / RP_BLOCK, RP_BLOCK, sh, (const uint8_t *)d_C, (const uint8_t *)d_b, e_G, e_F, e_B,
               (const uint8_t *)d_r, (const uint8_t *)d_rng, (uint32_t *)d_ts, w_r, w_dr, d_status);
        std::vector<uint32_t> nterms(2 * nbatch, (uint32_t)N);
        for (uint32_t j = 0; j < k && !rc; j++) {
            if (from_gens) {   // rows of generator-table scalars (B_blinding, B, G_0..): m_sc holds [2 nbatch][n + 2] scalars
                LAUNCH(c, s, "linc_terms", k_linc_terms_fixed, n_q + (nt32 + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, n_q, nt32, sh, j, (const uint32_t *)w_a,
                       (const uint32_t *)w_b, (const uint32_t *)w_G, (const uint32_t *)w_dr, m_sc);
                rc = msm_shared_dev_locked(c, n, 1, 2 * nbatch, 0, m_sc, nullptr, nullptr, m_out, m_st, nullptr, s, true);
            } else {
                LAUNCH(c, s, "linc_terms", k_linc_terms, n_q + (nt32 + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, n_q, nt32, sh, j, (const uint32_t *)w_a, (const uint32_t *)w_b,
                       (const uint32_t *)w_G, (const uint32_t *)w_dr, e_G, e_F, e_B, m_sc, m_pt);
                rc = msm_batch_dev_locked(c, 2 * nbatch, nterms.data(), m_sc, m_pt, m_out, m_st, s);   // all L_j and R_j of the batch (:104-117)
            }
            if (rc) break;
            LAUNCH(c, s, "linc_challenge", k_linc_challenge, (nb32 + RP_BLOCK - 1) / RP_BLOCK, RP_BLOCK, sh, j, (const uint32_t *)m_out, (const uint8_t *)m_st,
                   (uint32_t *)d_ts, (const uint32_t *)w_dr, w_r, w_x, w_xi, (uint8_t *)d_proofs, (uint32_t)proof_len, d_status);
            LAUNCH(c, s, "linc_fold", k_linc_fold, (nt32 + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, nt32, sh, j, (const uint32_t *)w_x, (const uint32_t *)w_xi, w_a, w_b, w_G);
        }
        if (rc) break;
        if (from_gens) {
            LAUNCH(c, s, "linc_sterms", k_linc_sterms_fixed, n_q + (nt32 + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, n_q, nt32, sh, (const uint32_t *)w_b,
                   (const uint32_t *)w_G, (const uint32_t *)w_dr, m_sc);
            rc = msm_shared_dev_locked(c, n, 1, nbatch, 0, m_sc, nullptr, nullptr, m_out, m_st, nullptr, s, true);
        } else {
            LAUNCH(c, s, "linc_sterms", k_linc_sterms, n_q + (nt32 + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, n_q, nt32, sh, (const uint32_t *)w_b, (const uint32_t *)w_G,
                   (const uint32_t *)w_dr, e_G, e_F, e_B, m_sc, m_pt);
            std::vector<uint32_t> nts(nbatch, (uint32_t)NS);
            rc = msm_batch_dev_locked(c, nbatch, nts.data(), m_sc, m_pt, m_out, m_st, s);               // every S (:155-157)
        }
        if (rc) break;
        LAUNCH(c, s, "linc_final", k_linc_final, (nb32 + RP_BLOCK - 1) / RP_BLOCK, RP_BLOCK, sh, (const uint32_t *)m_out, (const uint8_t *)m_st, (uint32_t *)d_ts,
               (const uint32_t *)w_a, (const uint32_t *)w_dr, (const uint32_t *)w_r, (uint8_t *)d_proofs, (uint32_t)proof_len, d_status, (uint8_t *)d_stb);
        if (hipGetLastError() != hipSuccess) rc = fail(c, BPGPU_ERR_HIP, "launch failed");
    } while (0);
    char *h_out = h + sz_in;
    if (!rc && hipMemcpyAsync(h_out, d_proofs, sz_out, hipMemcpyDeviceToHost, s) != hipSuccess) rc = fail(c, BPGPU_ERR_HIP, "D2H copy failed");
    char *h_ts = h + (d_ts - c->io_dev);   // the transcript block of the staging buffer is free again: reuse it for the way back
    if (!rc && transcripts_out && hipMemcpyAsync(h_ts, d_ts, nbatch * TS, hipMemcpyDeviceToHost, s) != hipSuccess) rc = fail(c, BPGPU_ERR_HIP, "D2H copy failed");
    rc = px.finish(rc);   // (on an error too: the stream is drained before the staging buffers are reused, the secrets cleared)
    if (rc) return rc;
    memcpy(proofs_out, h_out, nbatch * proof_len);
    memcpy(status_out, h_out + sz_pr, nbatch);
    if (transcripts_out) memcpy(transcripts_out, h_ts, nbatch * TS);
    return BPGPU_OK;
}

// ============================================================================
// batched range-proof creation (rp_prover.h)
// ============================================================================
extern "C" int bpgpu_rangeproof_prove_batch(bpgpu_ctx *c, size_t n, size_t m, size_t nbatch, const uint64_t *values, const uint8_t *blindings,
                                            const uint8_t *label, size_t label_len, const uint8_t *shared_transcript, const uint8_t *rng,
                                            uint8_t *proofs_out, uint8_t *commitments_out, uint8_t *transcripts_out) {
    if (!c || (label_len && !label)) return BPGPU_ERR_INVALID_ARG;
    if (nbatch == 0) return BPGPU_OK;
    if (!values || !blindings || !proofs_out || !commitments_out) return BPGPU_ERR_INVALID_ARG;
    // the parameter checks of prove_multiple_with_rng / Party::new / Dealer::new (mod.rs:244-246, party.rs:41-53, dealer.rs:34-60)
    if (!(n == 8 || n == 16 || n == 32 || n == 64)) return fail(c, BPGPU_ERR_INVALID_ARG, "InvalidBitsize: n must be 8, 16, 32 or 64");
    if (m == 0 || (m & (m - 1))) return fail(c, BPGPU_ERR_INVALID_ARG, "InvalidAggregation: m must be a power of two");
    if (shared_transcript && !ts_state_ok(shared_transcript)) return fail(c, BPGPU_ERR_INVALID_ARG, "malformed transcript state");
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->d_table) return fail(c, BPGPU_ERR_NO_GENS, "generators not loaded");
    if (c->gens_capacity < n || c->party_capacity < m) return fail(c, BPGPU_ERR_NO_GENS, "InvalidGeneratorsLength: generators too small for n=%zu m=%zu", n, m);
    if (n < 64)
        for (size_t i = 0; i < nbatch * m; i++)
            if (values[i] >> n) return fail(c, BPGPU_ERR_INVALID_ARG, "value %zu does not fit %zu bits", i, n);   // (the reference would prove a false statement's bits silently; refuse)
    const size_t nm = n * m;
    size_t k = 0;
    while (((size_t)1 << k) < nm) k++;
    if (k > BP_RP_MAX_K) return fail(c, BPGPU_ERR_INVALID_ARG, "n*m > 2^%d not supported", BP_RP_MAX_K);
    rpp_shape sh;
    sh.n = (uint32_t)n;
    sh.m = (uint32_t)m;
    sh.nm = (uint32_t)nm;
    sh.k = (uint32_t)k;
    sh.nproofs = (uint32_t)nbatch;
    sh.proof_len = (uint32_t)(32 * (9 + 2 * k));
    sh.n_gen_terms = (uint32_t)(2 * nm + 2);
    sh.rng_per_proof = (uint32_t)(64 * (m * (2 * n + 2) + 2 * m));
    const size_t nmsm1 = nbatch * (2 + m), TS = BPGPU_TRANSCRIPT_BYTES, proof_len = sh.proof_len;
    const uint64_t gs_bytes = (uint64_t)nmsm1 * sh.n_gen_terms * 32;
    if (gs_bytes > (4ull << 30) || (uint64_t)nmsm1 * sh.n_gen_terms > 0x7fffffffull || (uint64_t)nbatch * nm > 0x7fffffffull / 64)
        return fail(c, BPGPU_ERR_INVALID_ARG, "batch too large for this shape");
    hipStream_t s = c->stream;
    int rc = ctx_enter(c, s);
    if (rc) return rc;
    prover_exit px{c, s};
    // ---- inputs through pinned staging into the persistent IO buffer
    const size_t sz_val = align_up(nbatch * m * 8), sz_bl = align_up(nbatch * m * 32), sz_rng = align_up(nbatch * sh.rng_per_proof), sz_ts = align_up(nbatch * TS);
    const size_t sz_in = sz_val + sz_bl + sz_rng + sz_ts;
    const size_t sz_pr = align_up(nbatch * proof_len), sz_cm = align_up(nbatch * m * 32);
    const size_t sz_out = sz_pr + sz_cm + sz_ts;
    rc = io_reserve(c, sz_in + sz_out);
    if (rc) return px.finish(rc);
    char *h = nullptr;
    rc = pin_alloc(c, s, sz_in + sz_out, &h);
    if (rc) return px.finish(rc);
    px.h = h;
    px.host_secret_bytes = sz_val + sz_bl + sz_rng;   // values, blindings, every random scalar
    memcpy(h, values, nbatch * m * 8);
    memcpy(h + sz_val, blindings, nbatch * m * 32);
    if (rng) memcpy(h + sz_val + sz_bl, rng, nbatch * sh.rng_per_proof);
    else {   // thread_rng() of prove_multiple (mod.rs:291-310): OS CSPRNG
        rc = os_random(c, h + sz_val + sz_bl, nbatch * sh.rng_per_proof);
        if (rc) return px.finish(rc);
    }
    uint8_t st_start[BPGPU_TRANSCRIPT_BYTES];
    {   // every proof's transcript after rangeproof_domain_sep(n, m) (transcript.rs:44-48)
        if (shared_transcript) memcpy(st_start, shared_transcript, TS);
        else bpgpu_transcript_new(label, label_len, st_start);
        uint32_t w[50];
        strobe t;
        ts_to_strobe(t, w, st_start);
        const uint8_t rp[13] = {'r', 'a', 'n', 'g', 'e', 'p', 'r', 'o', 'o', 'f', ' ', 'v', '1'}, ln[1] = {'n'}, lm[1] = {'m'};
        merlin_append_message(t, DOM_SEP, 7, rp, 13);
        merlin_append_u64(t, ln, 1, n);
        merlin_append_u64(t, lm, 1, m);
        uint8_t st1[BPGPU_TRANSCRIPT_BYTES];
        ts_from_strobe(st1, t);
        for (size_t p = 0; p < nbatch; p++) memcpy(h + sz_val + sz_bl + sz_rng + p * TS, st1, TS);
    }
    char *d_in = c->io_dev;
    const uint64_t *d_values = (const uint64_t *)d_in;
    const uint8_t *d_bl = (const uint8_t *)(d_in + sz_val), *d_rng = (const uint8_t *)(d_in + sz_val + sz_bl);
    uint32_t *d_ts = (uint32_t *)(d_in + sz_val + sz_bl + sz_rng);
    uint8_t *d_proofs = (uint8_t *)(d_in + sz_in), *d_coms = d_proofs + sz_pr;
    PX_HIPCHK(px, hipMemcpyAsync(d_in, h, sz_in, hipMemcpyHostToDevice, s));
    // ---- working set
    const size_t w_gs = align_up(gs_bytes), w_mo = align_up(nmsm1 * 32), w_ms = align_up(nmsm1 + 64), w_f = align_up(RPP_FIXED * nbatch * 32),
                 w_p = align_up(RPP_PARTY_FIELDS * nbatch * m * 32), w_v = align_up(nbatch * nm * 32), w_q = align_up(nbatch * 32), w_gen = align_up(nm * 32);
    const size_t need = w_gs + w_mo + w_ms + w_f + w_p + 10 * w_v + w_q + 2 * w_gen;
    if (c->rpp_cap < need) {   // (the old block is clean: every call clears it on its way out)
        PX_HIPCHK(px, hipDeviceSynchronize());
        if (c->rpp_buf) PX_HIPCHK(px, hipFree(c->rpp_buf));
        c->rpp_buf = nullptr;
        c->rpp_cap = 0;
        PX_HIPCHK(px, hipMalloc((void **)&c->rpp_buf, need + need / 8));
        c->rpp_cap = need + need / 8;
    }
    char *wb = c->rpp_buf;
    uint32_t *gsc = (uint32_t *)wb, *mo = (uint32_t *)(wb + w_gs);
    uint8_t *mst = (uint8_t *)mo + w_mo;
    uint32_t *fields = (uint32_t *)(mst + w_ms), *party = (uint32_t *)((char *)fields + w_f);
    char *vv = (char *)party + w_p;
    uint32_t *sL = (uint32_t *)vv, *sR = (uint32_t *)(vv + w_v), *l0 = (uint32_t *)(vv + 2 * w_v), *l1 = (uint32_t *)(vv + 3 * w_v), *r0 = (uint32_t *)(vv + 4 * w_v),
             *r1 = (uint32_t *)(vv + 5 * w_v), *avec = (uint32_t *)(vv + 6 * w_v), *bvec = (uint32_t *)(vv + 7 * w_v), *Gf = (uint32_t *)(vv + 8 * w_v),
             *Hf = (uint32_t *)(vv + 9 * w_v);
    uint32_t *qenc = (uint32_t *)(vv + 10 * w_v), *Genc = (uint32_t *)((char *)qenc + w_q), *Henc = (uint32_t *)((char *)Genc + w_gen);
    const uint32_t nb32 = (uint32_t)nbatch, nbits = (uint32_t)(nbatch * nm), n_b = (nb32 + BP_BLOCK - 1) / BP_BLOCK;
    do {
        uint32_t *d_ids = nullptr;
        rc = gen_ids_for(c, n, m, &d_ids);
        if (rc) break;
        LAUNCH(c, s, "gather32", k_gather32, ((uint32_t)(2 * nm) + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, (uint32_t)(2 * nm), (const uint32_t *)(d_ids + 2),
               (const uint32_t *)c->d_gens, Genc);   // G(n, m) then H(n, m): Henc = Genc + nm records when w_gen is exact
        if (hipMemsetAsync(gsc, 0, gs_bytes, s) != hipSuccess || hipMemsetAsync(d_proofs, 0, sz_pr, s) != hipSuccess) {
            rc = fail(c, BPGPU_ERR_HIP, "memset failed");
            break;
        }
        // (1) V_j, A, S
        LAUNCH(c, s, "rpp_commit1", k_rpp_commit1, n_b + (nbits + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, n_b, nbits, sh, d_values, d_bl, d_rng, gsc, party, sL, sR);
        rc = msm_shared_dev_locked(c, n, m, nmsm1, 0, gsc, nullptr, nullptr, mo, mst, nullptr, s, false, c->prover_ct);
        if (rc) break;
        LAUNCH(c, s, "rpp_chal1", k_rpp_chal1, (nb32 + RP_BLOCK - 1) / RP_BLOCK, RP_BLOCK, sh, (const uint32_t *)mo, d_ts, fields, d_proofs, d_coms);
        // (2) polynomials, T_1, T_2
        const uint32_t npar = (uint32_t)(nbatch * m);
        LAUNCH(c, s, "rpp_poly", k_rpp_poly, (npar + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, npar, sh, d_values, (const uint32_t *)fields, (const uint32_t *)sL,
               (const uint32_t *)sR, l0, l1, r0, r1, party);
        if (hipMemsetAsync(gsc, 0, (size_t)2 * nbatch * sh.n_gen_terms * 32, s) != hipSuccess) {
            rc = fail(c, BPGPU_ERR_HIP, "memset failed");
            break;
        }
        LAUNCH(c, s, "rpp_tcommit", k_rpp_tcommit, n_b, BP_BLOCK, sh, d_rng, gsc, party);
        rc = msm_shared_dev_locked(c, n, m, 2 * nbatch, 0, gsc, nullptr, nullptr, mo, mst, nullptr, s, false, c->prover_ct);
        if (rc) break;
        // (3) x, t_x ..., w; Q = w B; the inner-product argument's inputs
        if (hipMemsetAsync(gsc, 0, (size_t)nbatch * sh.n_gen_terms * 32, s) != hipSuccess) {
            rc = fail(c, BPGPU_ERR_HIP, "memset failed");
            break;
        }
        LAUNCH(c, s, "rpp_chal2", k_rpp_chal2, (nb32 + RP_BLOCK - 1) / RP_BLOCK, RP_BLOCK, sh, (const uint32_t *)mo, d_ts, fields, (const uint32_t *)party, gsc,
               d_proofs);
        LAUNCH(c, s, "rpp_vectors", k_rpp_vectors, (nbits + BP_BLOCK - 1) / BP_BLOCK, BP_BLOCK, nbits, sh, (const uint32_t *)fields, (const uint32_t *)l0,
               (const uint32_t *)l1, (const uint32_t *)r0, (const uint32_t *)r1, avec, bvec, Gf, Hf);
        // (4) InnerProductProof::create over G(n, m), H(n, m) with Q = w B (dealer.rs:279-293): every L_j / R_j is a generator-table
        // MSM (the Q term is (c w) on B); L, R pairs and a, b go behind the 7 fixed elements of the proof
        ippc_fixed fx;
        fx.gn = n;
        fx.gm = m;
        fx.w = fields + (size_t)RPP_W * nbatch * 8;
        fx.gen_scalars = gsc;
        rc = ippc_core(c, s, nm, k, nbatch, avec, bvec, Gf, Hf, qenc, Genc, Genc + 8 * nm, 1, d_ts, d_proofs + 224, proof_len, nullptr, &fx);
    } while (0);
    char *h_out = h + sz_in;
    if (!rc) {
        if (hipMemcpyAsync(d_coms + sz_cm, d_ts, nbatch * TS, hipMemcpyDeviceToDevice, s) != hipSuccess ||
            hipMemcpyAsync(h_out, d_proofs, sz_out, hipMemcpyDeviceToHost, s) != hipSuccess)
            rc = fail(c, BPGPU_ERR_HIP, "D2H copy failed");
    }
    rc = px.finish(rc);
    if (rc) return rc;
    memcpy(proofs_out, h_out, nbatch * proof_len);
    memcpy(commitments_out, h_out + sz_pr, nbatch * m * 32);
    if (transcripts_out) memcpy(transcripts_out, h_out + sz_pr + sz_cm, nbatch * TS);
    return BPGPU_OK;
}
