// TEST-ONLY driver of the host build of the pool's scheduler (pool.hip + fake_device.cpp, see there): plain threads hammer the
// combining queue in the shapes the crate is called in, every result is recomputed from the request's own inputs (fake_model.h).
//   threads T B S      T threads looping BLOCKING bpgpu_pool_rangeproof_verify_ts calls of B proofs, own transcript per proof, S seconds
//   tickets T Q S      T threads keeping Q single-proof tickets in flight each
//   mixed T S          T threads drawing at random: blocking calls (per-proof / shared transcripts, label form), tickets, multiscalar
//                      multiplications (shared-generator and ragged), inner-product proofs, malformed requests, multi-device spans,
//                      option changes, injected chain failures
//   destroy T          requests of every kind in flight from T threads while the pool is destroyed: every call returns, nothing
//                      reads "verified" that was not
//   flush              device-pointer batches (several shapes, labels) through submit_dev / flush / tickets
// Prints one JSON line; exit code 1 on any mismatch.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bpgpu.h"
#include "fake_model.h"

extern "C" void fake_set_model(uint64_t base_us, uint64_t item_ns, uint64_t beside_us);
extern "C" void fake_fail_next_chains(int n);
extern "C" int bpgpu_internal_pool_tune(bpgpu_pool *, const char *, int64_t);
extern "C" int bpgpu_pool_msm_batch_shared(bpgpu_pool *, size_t, size_t, size_t, size_t, const uint8_t *, const uint8_t *, const uint8_t *, uint8_t *, uint8_t *);
extern "C" int bpgpu_pool_msm_batch_shared_submit(bpgpu_pool *, size_t, size_t, size_t, size_t, const uint8_t *, const uint8_t *, const uint8_t *, uint8_t *, uint8_t *,
                                                  bpgpu_ticket **);
extern "C" int bpgpu_pool_msm_batch(bpgpu_pool *, size_t, const uint32_t *, const uint8_t *, const uint8_t *, uint8_t *, uint8_t *);
extern "C" int bpgpu_pool_ipp_verify(bpgpu_pool *, size_t, size_t, const uint8_t *, size_t, const uint8_t *, size_t, const uint8_t *, const uint8_t *, const uint8_t *,
                                     const uint8_t *, const uint8_t *, const uint8_t *, uint8_t *, uint8_t *);
extern "C" int bpgpu_pool_trace_dump(bpgpu_pool *, const char *);
extern "C" int bpgpu_internal_policy_seal(bpgpu_pool *, uint32_t, uint64_t, uint64_t, uint32_t, uint64_t, uint32_t);
extern "C" int bpgpu_internal_policy_seal_cohort(bpgpu_pool *, uint32_t, uint64_t, uint64_t, uint32_t, uint64_t, uint32_t, uint32_t, uint64_t);

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static const size_t N = 64, M = 1, PL = 672, TS = 208;
struct corpus {
    size_t count;
    std::vector<uint8_t> proofs, coms, states;
    explicit corpus(size_t c) : count(c), proofs(c * PL), coms(c * 32 * M), states(c * TS) {
        std::mt19937_64 g(12345);
        for (auto &b : proofs) b = (uint8_t)g();
        for (auto &b : coms) b = (uint8_t)g();
        for (size_t i = 0; i < c; i++) {
            uint8_t *st = &states[i * TS];
            for (size_t j = 0; j < 200; j++) st[j] = (uint8_t)g();
            static const uint8_t pos[3][3] = {{40, 7, 2}, {97, 60, 2}, {12, 0, 2}};   // three STROBE position classes
            const uint8_t *q = pos[g() % 3];
            st[200] = q[0], st[201] = q[1], st[202] = q[2];
            memset(st + 203, 0, 5);
        }
    }
};

static std::atomic<uint64_t> g_mismatch{0}, g_errors{0}, g_done{0};
static void bad(const char *what) {
    if (g_mismatch.fetch_add(1) < 5) fprintf(stderr, "MISMATCH: %s\n", what);
}

// one range proof's expectation from its inputs
static void expect_rp(const corpus &c, size_t idx, const uint8_t *ts, uint8_t *v, uint8_t *ts_out, uint8_t *msm) {
    fake_rp_result(N, M, &c.proofs[idx * PL], PL, &c.coms[idx * 32 * M], ts, v, ts_out, msm);
}

static bpgpu_pool *make_pool(int ndev, int lanes) {
    std::vector<int> devs(ndev);
    for (int i = 0; i < ndev; i++) devs[i] = i;
    bpgpu_pool *p = nullptr;
    if (bpgpu_pool_create(devs.data(), ndev, lanes, &p)) {
        fprintf(stderr, "pool_create failed\n");
        exit(2);
    }
    if (bpgpu_pool_gens_create(p, 2048, 2)) {
        fprintf(stderr, "gens failed\n");
        exit(2);
    }
    if (const char *o = getenv("BP_OPTS")) {
        std::string s = o;
        size_t q = 0;
        while (q < s.size()) {
            size_t e = s.find(',', q);
            if (e == std::string::npos) e = s.size();
            const std::string kv = s.substr(q, e - q);
            const size_t eq = kv.find('=');
            if (eq != std::string::npos && bpgpu_pool_set_option(p, kv.substr(0, eq).c_str(), atoll(kv.c_str() + eq + 1)))
                fprintf(stderr, "option %s refused: %s\n", kv.c_str(), bpgpu_pool_last_error(p));
            q = e + 1;
        }
    }
    return p;
}

struct stats {
    std::vector<float> lat;
};
static void report(bpgpu_pool *pool, const char *mode, int T, int arg, double secs, std::vector<stats> &st) {
    std::vector<float> all;
    for (auto &s : st) all.insert(all.end(), s.lat.begin(), s.lat.end());
    std::sort(all.begin(), all.end());
    auto pct = [&](double q) { return all.empty() ? 0.0 : (double)all[(size_t)(q * (all.size() - 1))]; };
    int64_t chains = 0, cproofs = 0, iss = 0, cmp = 0, dlv = 0, polls = 0;
    bpgpu_pool_get_option(pool, "stat_combined_chains", &chains);
    bpgpu_pool_get_option(pool, "stat_combined_proofs", &cproofs);
    bpgpu_pool_get_option(pool, "stat_svc_issue_us", &iss);
    bpgpu_pool_get_option(pool, "stat_svc_complete_us", &cmp);
    bpgpu_pool_get_option(pool, "stat_svc_deliver_us", &dlv);
    bpgpu_pool_get_option(pool, "stat_svc_polls", &polls);
    printf("{\"mode\": \"%s\", \"threads\": %d, \"arg\": %d, \"seconds\": %.2f, \"items\": %llu, \"rate_per_s\": %.0f, \"lat_ms\": {\"p50\": %.3f, \"p90\": %.3f, \"p99\": %.3f, "
           "\"max\": %.3f}, \"chains\": %lld, \"items_per_chain\": %.1f, \"mismatches\": %llu, \"errors\": %llu, \"svc_us_per_chain\": {\"issue\": %.1f, \"complete\": %.1f, "
           "\"deliver\": %.1f}, \"polls\": %lld}\n",
           mode, T, arg, secs, (unsigned long long)g_done.load(), (double)g_done.load() / secs, pct(0.5), pct(0.9), pct(0.99), all.empty() ? 0.0 : (double)all.back(),
           (long long)chains, chains ? (double)cproofs / (double)chains : 0.0, (unsigned long long)g_mismatch.load(), (unsigned long long)g_errors.load(),
           chains ? (double)iss / (double)chains : 0.0, chains ? (double)cmp / (double)chains : 0.0, chains ? (double)dlv / (double)chains : 0.0, (long long)polls);
}

// ---- the request kinds of `mixed` / `destroy`: each issues one call and checks it; returns false when the pool refused (being destroyed) ----
struct kinds {
    const corpus &c;
    bpgpu_pool *pool;
    std::mt19937_64 g;
    bool tolerate_errors;   // (destroy: calls may fail -- but then nothing may read as verified)
    kinds(const corpus &cc, bpgpu_pool *p, uint64_t seed, bool tol) : c(cc), pool(p), g(seed), tolerate_errors(tol) {}

    void check_rp(int rc, size_t i0, size_t B, const uint8_t *ts_used, size_t stride, const uint8_t *v, const uint8_t *ts_out, const uint8_t *msm) {
        for (size_t b = 0; b < B; b++) {
            const size_t idx = (i0 + b) % c.count;
            uint8_t ev, ets[TS], em[32];
            expect_rp(c, idx, ts_used + b * stride, &ev, ets, em);
            if (rc) {
                if (!tolerate_errors) g_errors++;
                if (v[b] != BPGPU_VERDICT_UNDECIDED && v[b] != ev) bad("after an error a verdict is neither undecided nor right");
                continue;
            }
            if (v[b] != ev) bad("verdict");
            if (ts_out && memcmp(ts_out + b * TS, ets, TS) != 0) bad("advanced transcript");
            if (msm && memcmp(msm + b * 32, em, 32) != 0) bad("encoding");
        }
    }
    void blocking_ts(size_t B, bool want_msm) {
        const size_t i0 = g() % c.count;
        std::vector<uint8_t> pr(B * PL), cm(B * 32), st(B * TS), v(B, 0), to(B * TS), ms(B * 32);
        for (size_t b = 0; b < B; b++) {
            const size_t idx = (i0 + b) % c.count;
            memcpy(&pr[b * PL], &c.proofs[idx * PL], PL);
            memcpy(&cm[b * 32], &c.coms[idx * 32], 32);
            memcpy(&st[b * TS], &c.states[idx * TS], TS);
        }
        const int rc = bpgpu_pool_rangeproof_verify_ts(pool, N, M, B, pr.data(), PL, cm.data(), st.data(), TS, nullptr, v.data(), want_msm ? ms.data() : nullptr, to.data());
        check_rp(rc, i0, B, st.data(), TS, v.data(), to.data(), want_msm ? ms.data() : nullptr);
        g_done += B;
    }
    void blocking_shared(size_t B, bool want_out, int which_label) {
        const size_t i0 = g() % c.count;
        std::vector<uint8_t> pr(B * PL), cm(B * 32), v(B, 0), to(B * TS);
        for (size_t b = 0; b < B; b++) {
            const size_t idx = (i0 + b) % c.count;
            memcpy(&pr[b * PL], &c.proofs[idx * PL], PL);
            memcpy(&cm[b * 32], &c.coms[idx * 32], 32);
        }
        const char *labels[3] = {"label-a", "label-b", "another label"};
        uint8_t st0[TS];
        bpgpu_transcript_new((const uint8_t *)labels[which_label], strlen(labels[which_label]), st0);
        int rc;
        if (want_out) rc = bpgpu_pool_rangeproof_verify_ts(pool, N, M, B, pr.data(), PL, cm.data(), st0, 0, nullptr, v.data(), nullptr, to.data());
        else rc = bpgpu_pool_rangeproof_verify(pool, N, M, B, pr.data(), PL, cm.data(), (const uint8_t *)labels[which_label], strlen(labels[which_label]), nullptr, v.data(), nullptr);
        check_rp(rc, i0, B, st0, 0, v.data(), want_out ? to.data() : nullptr, nullptr);
        g_done += B;
    }
    void malformed() {   // a length no chain takes: the ordinary entry point's answer, states handed back untouched
        const size_t B = 3, len = 640;
        std::vector<uint8_t> pr(B * len, 7), cm(B * 32, 1), st(B * TS), v(B, 0), to(B * TS);
        for (size_t b = 0; b < B; b++) memcpy(&st[b * TS], &c.states[b * TS], TS);
        const int rc = bpgpu_pool_rangeproof_verify_ts(pool, N, M, B, pr.data(), len, cm.data(), st.data(), TS, nullptr, v.data(), nullptr, to.data());
        if (rc) {
            if (!tolerate_errors) g_errors++;
            return;
        }
        for (size_t b = 0; b < B; b++)
            if (v[b] != FAKE_VERDICT_DIRECT || memcmp(&to[b * TS], &st[b * TS], TS) != 0) bad("malformed request");
        g_done += B;
    }
    void msm_shared(size_t B, size_t nu, bool ticket) {
        const size_t n = 64, m = 1, ng = 2 * n * m + 2;
        std::vector<uint8_t> gs(B * ng * 32), us(B * nu * 32 + 1), up(B * nu * 32 + 1), out(B * 32), st(B, 9);
        for (auto &x : gs) x = (uint8_t)g();
        for (auto &x : us) x = (uint8_t)g();
        for (auto &x : up) x = (uint8_t)g();
        int rc;
        if (ticket) {
            bpgpu_ticket *t = nullptr;
            rc = bpgpu_pool_msm_batch_shared_submit(pool, n, m, B, nu, gs.data(), us.data(), up.data(), out.data(), st.data(), &t);
            if (!rc) rc = bpgpu_pool_ticket_wait(pool, t);
        } else rc = bpgpu_pool_msm_batch_shared(pool, n, m, B, nu, gs.data(), us.data(), up.data(), out.data(), st.data());
        for (size_t b = 0; b < B; b++) {
            uint8_t eo[32], es;
            fake_msm_shared_result(&gs[b * ng * 32], ng, nu ? &us[b * nu * 32] : nullptr, nu ? &up[b * nu * 32] : nullptr, nu, eo, &es);
            if (rc) {
                if (st[b] != BPGPU_VERDICT_UNDECIDED && !(st[b] == es && memcmp(&out[b * 32], eo, 32) == 0) && st[b] != 9) bad("failed MSM reads as computed");
                continue;
            }
            if (st[b] != es || memcmp(&out[b * 32], eo, 32) != 0) bad("shared MSM result");
        }
        if (rc && !tolerate_errors) g_errors++;
        g_done += B;
    }
    void msm_ragged() {
        const size_t B = 1 + g() % 6;
        std::vector<uint32_t> nt(B);
        size_t total = 0;
        for (auto &x : nt) x = (uint32_t)(g() % 4 == 0 ? 0 : (g() % 3 == 0 ? 17 : 130)), total += x;
        std::vector<uint8_t> sc(total * 32 + 1), pt(total * 32 + 1), out(B * 32, 0xee), st(B, 9);
        for (auto &x : sc) x = (uint8_t)g();
        for (auto &x : pt) x = (uint8_t)g();
        const int rc = bpgpu_pool_msm_batch(pool, B, nt.data(), sc.data(), pt.data(), out.data(), st.data());
        size_t t0 = 0;
        for (size_t b = 0; b < B; b++) {
            uint8_t eo[32], es = 0;
            if (nt[b]) fake_msm_result(&sc[t0 * 32], &pt[t0 * 32], nt[b], eo, &es);
            else memset(eo, 0, 32);
            t0 += nt[b];
            if (rc) continue;
            if (st[b] != es || memcmp(&out[b * 32], eo, 32) != 0) bad("ragged MSM result");
        }
        if (rc && !tolerate_errors) g_errors++;
        g_done += B;
    }
    void ipp(size_t B, bool want_msm) {
        const size_t n = 16, len = 32 * (2 * 4 + 2);
        std::vector<uint8_t> pr(B * len), gf(B * n * 32), hf(B * n * 32), P(B * 32), Q(B * 32), G(B * n * 32), H(B * n * 32), v(B, 0), ms(B * 32);
        for (auto *vv : {&pr, &gf, &hf, &P, &Q, &G, &H})
            for (auto &x : *vv) x = (uint8_t)g();
        const char *label = "ipp label";
        const int rc = bpgpu_pool_ipp_verify(pool, n, B, pr.data(), len, (const uint8_t *)label, strlen(label), gf.data(), hf.data(), P.data(), Q.data(), G.data(), H.data(),
                                             v.data(), want_msm ? ms.data() : nullptr);
        uint8_t st0[TS];
        bpgpu_transcript_new((const uint8_t *)label, strlen(label), st0);
        for (size_t b = 0; b < B; b++) {
            uint8_t ev, em[32];
            fake_ipp_result(n, &pr[b * len], len, st0, &gf[b * n * 32], &hf[b * n * 32], &P[b * 32], &Q[b * 32], &G[b * n * 32], &H[b * n * 32], &ev, em);
            if (rc) {
                if (v[b] != BPGPU_VERDICT_UNDECIDED && v[b] != ev) bad("failed inner-product proof reads as verified");
                continue;
            }
            if (v[b] != ev || (want_msm && memcmp(&ms[b * 32], em, 32) != 0)) bad("inner-product verdict");
        }
        if (rc && !tolerate_errors) g_errors++;
        g_done += B;
    }
    void ticket_burst(size_t Q) {
        struct slot {
            bpgpu_ticket *t = nullptr;
            size_t idx = 0;
            uint8_t v = 0, ts[TS];
        };
        std::vector<slot> ring(Q);
        for (slot &s : ring) {
            s.idx = g() % c.count;
            if (bpgpu_pool_rangeproof_submit_ts(pool, N, M, 1, &c.proofs[s.idx * PL], PL, &c.coms[s.idx * 32], &c.states[s.idx * TS], TS, nullptr, &s.v, nullptr, s.ts, &s.t)) {
                s.t = nullptr;
                if (!tolerate_errors) g_errors++;
            }
        }
        for (slot &s : ring) {
            if (!s.t) continue;
            const int rc = bpgpu_pool_ticket_wait(pool, s.t);
            check_rp(rc, s.idx, 1, &c.states[s.idx * TS], TS, &s.v, s.ts, nullptr);
            g_done++;
        }
    }
    void one() {
        switch (g() % 12) {
        case 0: blocking_ts(1, false); break;
        case 1: blocking_ts(1 + g() % 40, g() % 2); break;
        case 2: blocking_shared(1 + g() % 5, false, (int)(g() % 3)); break;
        case 3: blocking_shared(1 + g() % 5, true, (int)(g() % 3)); break;
        case 4: malformed(); break;
        case 5: msm_shared(1 + g() % 3, 33, g() % 2); break;
        case 6: msm_shared(1, 0, false); break;
        case 7: msm_ragged(); break;
        case 8: ipp(1 + g() % 4, g() % 2); break;
        case 9: ticket_burst(1 + g() % 64); break;
        case 10: blocking_ts(600 + g() % 3000, false); break;   // spans buffers (and devices)
        default: blocking_ts(2, true); break;
        }
    }
};

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: pool_host_test threads T B S | tickets T Q S | mixed T S | destroy T | flush | policy\n");
        return 2;
    }
    const std::string mode = argv[1];
    const corpus C(4096);
    if (getenv("BP_MODEL")) {
        unsigned long long a = 550, b = 200, c = 100;
        sscanf(getenv("BP_MODEL"), "%llu,%llu,%llu", &a, &b, &c);
        fake_set_model(a, b, c);
    }
    if (mode == "threads" || mode == "tickets") {
        const int T = atoi(argv[2]), arg = atoi(argv[3]);
        const double secs = atof(argv[4]);
        bpgpu_pool *pool = make_pool(1, 8);
        if (getenv("BP_TRACE")) bpgpu_pool_set_option(pool, "combine_trace", 4096);
        std::vector<stats> st(T);
        std::atomic<bool> stop{false};
        std::atomic<int> ready{0};
        auto worker = [&](int t) {
            std::vector<float> &L = st[t].lat;
            L.reserve(1 << 18);
            size_t cur = ((size_t)t * 7919) % C.count;
            ready++;
            while (ready.load() < T) std::this_thread::yield();
            if (mode == "threads") {
                const size_t B = arg;
                std::vector<uint8_t> pr(B * PL), cm(B * 32), sts(B * TS), v(B), to(B * TS);
                while (!stop.load(std::memory_order_relaxed)) {
                    const size_t i0 = cur;
                    for (size_t b = 0; b < B; b++) {
                        const size_t idx = (i0 + b) % C.count;
                        memcpy(&pr[b * PL], &C.proofs[idx * PL], PL);
                        memcpy(&cm[b * 32], &C.coms[idx * 32], 32);
                        memcpy(&sts[b * TS], &C.states[idx * TS], TS);
                    }
                    cur = (cur + B) % C.count;
                    const double t0 = now_s();
                    const int rc = bpgpu_pool_rangeproof_verify_ts(pool, N, M, B, pr.data(), PL, cm.data(), sts.data(), TS, nullptr, v.data(), nullptr, to.data());
                    L.push_back((float)((now_s() - t0) * 1e3));
                    if (rc) g_errors++;
                    for (size_t b = 0; b < B; b++) {
                        uint8_t ev, ets[TS];
                        expect_rp(C, (i0 + b) % C.count, &sts[b * TS], &ev, ets, nullptr);
                        if (v[b] != ev || memcmp(&to[b * TS], ets, TS) != 0) bad("threads: result");
                    }
                    g_done += B;
                }
            } else {
                const int Q = arg;
                struct slot {
                    bpgpu_ticket *t = nullptr;
                    size_t idx = 0;
                    double t0 = 0;
                    uint8_t v[1], ts[TS];
                };
                std::vector<slot> ring(Q);
                size_t head = 0;
                auto harvest = [&](slot &s) {
                    if (bpgpu_pool_ticket_wait(pool, s.t)) g_errors++;
                    uint8_t ev, ets[TS];
                    expect_rp(C, s.idx, &C.states[s.idx * TS], &ev, ets, nullptr);
                    if (s.v[0] != ev || memcmp(s.ts, ets, TS) != 0) bad("tickets: result");
                    s.t = nullptr;
                    g_done++;
                };
                while (!stop.load(std::memory_order_relaxed)) {
                    slot &s = ring[head];
                    if (s.t) {
                        harvest(s);
                        L.push_back((float)((now_s() - s.t0) * 1e3));
                    }
                    s.idx = cur;
                    cur = (cur + 1) % C.count;
                    s.t0 = now_s();
                    if (bpgpu_pool_rangeproof_submit_ts(pool, N, M, 1, &C.proofs[s.idx * PL], PL, &C.coms[s.idx * 32], &C.states[s.idx * TS], TS, nullptr, s.v, nullptr, s.ts,
                                                        &s.t)) {
                        g_errors++;
                        s.t = nullptr;
                    }
                    head = (head + 1) % Q;
                }
                for (slot &s : ring)
                    if (s.t) harvest(s);
            }
        };
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back(worker, t);
        while (ready.load() < T) std::this_thread::yield();
        const double t1 = now_s();
        std::this_thread::sleep_for(std::chrono::duration<double>(secs));
        stop = true;
        for (auto &x : th) x.join();
        report(pool, mode.c_str(), T, arg, now_s() - t1, st);
        if (getenv("BP_TRACE")) bpgpu_pool_trace_dump(pool, getenv("BP_TRACE"));
        bpgpu_pool_destroy(pool);
    } else if (mode == "mixed") {
        const int T = atoi(argv[2]);
        const double secs = atof(argv[3]);
        bpgpu_pool *pool = make_pool(2, 4);
        std::atomic<bool> stop{false};
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                kinds k(C, pool, 1000 + t, false);
                while (!stop.load(std::memory_order_relaxed)) k.one();
            });
        // one thread changes the queue's options while the others run
        std::thread tuner([&] {
            std::mt19937_64 g(77);
            // (inflight_wide, hold_us, wide_proofs: constants of the sealing policy, moved through the library's test hook)
            const char *keys[] = {"combine_wait_us", "combine_quiet_us", "combine_inflight", "inflight_wide", "combine_max_open", "hold_us", "wide_proofs",
                                  "combine_mapped_out", "coalesce_proofs"};
            const int64_t lo[] = {5, 2, 1, 1, 1, 20, 8, 0, 64}, hi[] = {300, 60, 8, 4, 6, 800, 2000, 4096, 5120};
            while (!stop.load(std::memory_order_relaxed)) {
                const int i = (int)(g() % 9);
                const int64_t v = lo[i] + (int64_t)(g() % (uint64_t)(hi[i] - lo[i] + 1));
                if (strncmp(keys[i], "co", 2)) bpgpu_internal_pool_tune(pool, keys[i], v);
                else bpgpu_pool_set_option(pool, keys[i], v);
                std::this_thread::sleep_for(std::chrono::milliseconds(3));
            }
        });
        const double t1 = now_s();
        std::this_thread::sleep_for(std::chrono::duration<double>(secs));
        stop = true;
        for (auto &x : th) x.join();
        tuner.join();
        std::vector<stats> st;
        report(pool, "mixed", T, 0, now_s() - t1, st);
        bpgpu_pool_destroy(pool);
    } else if (mode == "failures") {   // chains that fail to issue: every piece reads UNDECIDED, the call reports the error, the queue goes on
        bpgpu_pool *pool = make_pool(1, 4);
        const int T = 8;
        std::atomic<bool> stop{false};
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                kinds k(C, pool, 5000 + t, true);
                while (!stop.load(std::memory_order_relaxed)) k.one();
            });
        for (int i = 0; i < 30; i++) {
            fake_fail_next_chains(3);
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
        stop = true;
        for (auto &x : th) x.join();
        fake_fail_next_chains(0);
        {   // and afterwards everything verifies again
            kinds k(C, pool, 1, false);
            for (int i = 0; i < 50; i++) k.one();
        }
        std::vector<stats> st;
        report(pool, "failures", T, 0, 0.3, st);
        bpgpu_pool_destroy(pool);
    } else if (mode == "destroy") {
        // Every worker is INSIDE a blocking call when the pool is destroyed (the modelled chains take >= 30 ms, the main thread waits 10 ms
        // after the last worker started): some wait in open buffers (failed at once: undecided + error), some for chains on the device
        // (delivered), some for a free buffer.  A call that has not started when bpgpu_pool_destroy begins is the caller's bug, not tested.
        const int T = atoi(argv[2]);
        fake_set_model(30000, 200, 0);
        for (int round = 0; round < 6; round++) {
            bpgpu_pool *pool = make_pool(2, 4);
            if (round & 1) bpgpu_pool_set_option(pool, "combine_max_age_us", 5000000), bpgpu_pool_set_option(pool, "combine_wait_us", 1000000),
                bpgpu_pool_set_option(pool, "combine_quiet_us", 1000000);   // (odd rounds: buffers stay OPEN until the destroy)
            std::atomic<bool> stop{false};
            std::atomic<int> started{0};
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++)
                th.emplace_back([&, t] {
                    kinds k(C, pool, 9000 + 100 * round + t, true);
                    started++;
                    while (!stop.load(std::memory_order_seq_cst)) {
                        switch ((t + round) % 6) {
                        case 0: k.blocking_ts(1, false); break;
                        case 1: k.blocking_ts(700, true); break;
                        case 2: k.msm_shared(2, 33, false); break;
                        case 3: k.msm_ragged(); break;
                        case 4: k.ipp(2, true); break;
                        default: k.blocking_shared(3, true, t % 3); break;
                        }
                    }
                });
            while (started.load() < T) std::this_thread::yield();
            for (;;) {   // until every worker is inside the pool
                int64_t inside = 0;
                bpgpu_pool_get_option(pool, "stat_active_calls", &inside);
                if (inside >= T) break;
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
            stop.store(true, std::memory_order_seq_cst);
            const double t0 = now_s();
            bpgpu_pool_destroy(pool);
            const double dt = now_s() - t0;
            for (auto &x : th) x.join();
            if (dt > 1.0) bad("bpgpu_pool_destroy took more than a second with callers inside");
        }
        printf("{\"mode\": \"destroy\", \"threads\": %d, \"rounds\": 6, \"items\": %llu, \"mismatches\": %llu}\n", T, (unsigned long long)g_done.load(),
               (unsigned long long)g_mismatch.load());
    } else if (mode == "policy") {   // when a staging buffer leaves: both sealing policies row by row (plain host logic of pool.hip, default option values)
        bpgpu_pool *pool = make_pool(1, 2);
        uint64_t bad_rows = 0, rows = 0;
        const uint64_t us = 1000;
        auto regimes = [&](int want, uint32_t r, uint64_t age_us, uint64_t quiet_us, uint32_t inflight, uint64_t items, uint32_t n_free, const char *what) {
            rows++;
            if (bpgpu_internal_policy_seal(pool, r, age_us * us, quiet_us * us, inflight, items, n_free) != want) {
                bad_rows++;
                fprintf(stderr, "policy (regimes): %s\n", what);
            }
        };
        auto cohort = [&](int want, uint32_t r, uint64_t age_us, uint64_t quiet_us, uint32_t inflight, uint64_t items, uint32_t n_free, uint32_t expected, uint64_t since_done_us,
                          const char *what) {
            rows++;
            if (bpgpu_internal_policy_seal_cohort(pool, r, age_us * us, quiet_us * us, inflight, items, n_free, expected, since_done_us * us) != want) {
                bad_rows++;
                fprintf(stderr, "policy (cohorts): %s\n", what);
            }
        };
        // ---- regimes (combine_quiet_us 20, combine_wait_us 100, combine_inflight 4, combine_wide_proofs 384, combine_inflight_wide 3, combine_hold_us 400, combine_max_age_us 1500)
        regimes(0, 5, 10, 5, 0, 0, 12, "fresh arrivals keep a buffer open");
        regimes(1, 5, 30, 25, 0, 0, 12, "nothing joined for the quiet period: leave");
        regimes(1, 5, 120, 5, 0, 0, 12, "the first request has waited combine_wait_us: leave although requests still arrive");
        regimes(0, 5, 120, 25, 4, 400, 5, "four narrow chains run: the buffer keeps filling");
        regimes(1, 5, 120, 25, 3, 300, 6, "... three: it leaves");
        regimes(0, 100, 120, 25, 3, 1500, 8, "three WIDE chains run (500 each): at most three");
        regimes(0, 100, 120, 25, 2, 1000, 9, "two wide chains run: a quiet fragment smaller than half of what runs waits for company");
        regimes(1, 250, 120, 25, 2, 1000, 9, "... half as wide as what runs: leaves");
        regimes(1, 100, 450, 5, 2, 1000, 9, "... or after combine_hold_us");
        regimes(0, 50, 200, 30, 2, 100, 1, "the last free buffer is not spent on a narrow chain while chains run");
        regimes(1, 50, 1600, 0, 6, 3000, 1, "nothing waits longer than combine_max_age_us");
        // ---- cohorts (combine_cohort_inflight 2, combine_regroup_us 60; the deadlines above; the safety net at 3 x combine_max_age_us)
        cohort(1, 1, 3, 3, 0, 0, 12, 1, 8, "the lone caller the last chain released is back: leave at once, no quiet period");
        cohort(0, 40, 12, 2, 0, 0, 12, 64, 15, "40 of the 64 callers just released are back, more arrive: wait");
        cohort(1, 64, 20, 12, 0, 0, 12, 64, 25, "all 64 back and one poll without a newcomer: leave");
        cohort(0, 64, 20, 2, 0, 0, 12, 64, 25, "all 64 back but requests still pour in: take them along");
        cohort(0, 10, 40, 25, 0, 0, 12, 64, 30, "quiet, but the group that just finished is not back yet (regrouping)");
        cohort(1, 10, 90, 25, 0, 0, 12, 64, 70, "... regroup time over: leave");
        cohort(1, 10, 110, 2, 0, 0, 12, 64, 30, "... or combine_wait_us since the first arrival");
        cohort(1, 10, 30, 25, 0, 0, 12, 0, 1000000, "no recent completion: the quiet rule alone");
        cohort(0, 200, 300, 25, 2, 800, 9, 0, 1000000, "two chains run: the buffer fills until one ends");
        cohort(0, 100, 200, 25, 1, 600, 10, 0, 1000000, "one chain of 600 runs: a quiet fragment of 100 waits for company");
        cohort(1, 300, 200, 25, 1, 600, 10, 0, 1000000, "... half its width: leaves beside it");
        cohort(1, 100, 450, 2, 1, 600, 10, 0, 1000000, "... or after combine_hold_us");
        cohort(0, 50, 2000, 0, 2, 1000, 9, 0, 1000000, "with both slots busy even combine_max_age_us does not force a third chain");
        cohort(1, 50, 4600, 0, 2, 1000, 9, 0, 1000000, "... the safety net at three times that does");
        cohort(0, 50, 200, 30, 1, 100, 1, 50, 10, "the last free buffer is not spent while a chain runs");
        printf("{\"mode\": \"policy\", \"items\": %llu, \"mismatches\": %llu, \"errors\": 0}\n", (unsigned long long)rows, (unsigned long long)bad_rows);
        bpgpu_pool_destroy(pool);
        return bad_rows ? 1 : 0;
    } else if (mode == "flush") {
        bpgpu_pool *pool = make_pool(1, 4);
        std::mt19937_64 g(5);
        const char *labels[4] = {"label-aa", "label-bb", "label-cc", "other length"};
        struct item {
            size_t i0, B, len, m;
            int label;
            std::vector<uint8_t> v, ms;
            bpgpu_ticket *t;
        };
        std::vector<item> items(60);
        int64_t chains0 = 0;
        for (int rep = 0; rep < 3; rep++) {
            bpgpu_pool_set_option(pool, "stat_reset", 1);
            for (size_t k = 0; k < items.size(); k++) {
                item &it = items[k];
                it.i0 = g() % C.count, it.B = 1 + g() % 300, it.label = (int)(g() % 4), it.len = PL, it.m = M;
                it.v.assign(it.B, 0xcc), it.ms.assign(it.B * 32, 0);
                it.t = nullptr;
                if (it.i0 + it.B > C.count) it.i0 = 0;
                if (bpgpu_pool_rangeproof_submit_dev_ex(pool, 0, N, M, it.B, &C.proofs[it.i0 * PL], PL, &C.coms[it.i0 * 32], (const uint8_t *)labels[it.label],
                                                        strlen(labels[it.label]), nullptr, it.v.data(), (k % 3 == 0) ? it.ms.data() : nullptr, nullptr, 0, (k % 2) ? &it.t : nullptr))
                    g_errors++;
            }
            if (bpgpu_pool_wait(pool)) g_errors++;
            for (item &it : items) {
                if (it.t && bpgpu_pool_ticket_wait(pool, it.t)) g_errors++;
                uint8_t st0[TS];
                bpgpu_transcript_new((const uint8_t *)labels[it.label], strlen(labels[it.label]), st0);
                for (size_t b = 0; b < it.B; b++) {
                    uint8_t ev, em[32];
                    expect_rp(C, it.i0 + b, st0, &ev, nullptr, em);
                    if (it.v[b] != ev) bad("flush: verdict");
                }
                g_done += it.B;
            }
            bpgpu_pool_get_option(pool, "stat_chains", &chains0);
        }
        int64_t cp = 0;
        bpgpu_pool_get_option(pool, "stat_chain_proofs", &cp);
        printf("{\"mode\": \"flush\", \"items\": %llu, \"chains_last_round\": %lld, \"proofs_per_chain\": %.1f, \"mismatches\": %llu, \"errors\": %llu}\n",
               (unsigned long long)g_done.load(), (long long)chains0, chains0 ? (double)cp / (double)chains0 : 0.0, (unsigned long long)g_mismatch.load(),
               (unsigned long long)g_errors.load());
        bpgpu_pool_destroy(pool);
    } else {
        fprintf(stderr, "unknown mode\n");
        return 2;
    }
    return (g_mismatch.load() || g_errors.load()) ? 1 : 0;
}
