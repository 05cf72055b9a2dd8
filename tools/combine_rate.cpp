// Rate and latency of the pool's combining queue from plain host threads (no Python in the loop):
//   threads T [B]     T threads, each looping BLOCKING bpgpu_pool_rangeproof_verify_ts calls of B proofs (default 1), every proof with
//                     its own transcript -- the reference's verify_multiple_with_rng call shape (src/range_proof/mod.rs:345-353)
//   tickets T Q       T threads, each keeping Q single-proof tickets in flight (bpgpu_pool_rangeproof_submit_ts / _ticket_wait)
//   big T NB          T threads, each looping blocking label-mode calls of NB proofs (bpgpu_pool_rangeproof_verify)
//   msm T B           T threads, each looping BLOCKING bpgpu_pool_msm_batch_shared calls of B multiscalar multiplications of config 5's
//                     shape (4098 generator terms + 2081 per-MSM points: the R1CS verifier's call, src/r1cs/verifier.rs:459-491); inputs
//                     and the ORACLE's encodings from the file named by BP_MSM_INPUTS (bench.py writes it from workload.cfg5_inputs and
//                     the committed bench_data/cfg5_expected.json): every result is compared
// BP_TRACE=path: the queue's timeline (bpgpu_pool_trace_dump) of the run is written there.
// Inputs: the file tools/combine_rate.py writes (proofs proven on fresh and on pre-bound transcripts, a few invalid, with the
// oracle's verdicts and advanced transcripts): EVERY result of every call is compared with it.
// Build: g++ -O2 -std=c++17 -pthread -I include tools/combine_rate.cpp -L bulletproofs_amd/csrc -lbpgpu -Wl,-rpath,... -o combine_rate
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "bpgpu.h"
extern "C" int bpgpu_internal_pool_state(bpgpu_pool *, char *, size_t);   // (diagnostics hook of libbpgpu.so, not part of the ABI)

struct inputs {
    uint32_t n, m, proof_len, count;
    std::vector<uint8_t> proofs, coms, states, rng, exp_v, exp_ts, fresh;
};
static bool load(const char *path, inputs &in) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    uint32_t h[4];
    if (fread(h, 4, 4, f) != 4) return false;
    in.n = h[0], in.m = h[1], in.proof_len = h[2], in.count = h[3];
    auto rd = [&](std::vector<uint8_t> &v, size_t per) {
        v.resize((size_t)in.count * per);
        return fread(v.data(), 1, v.size(), f) == v.size();
    };
    const bool ok = rd(in.proofs, in.proof_len) && rd(in.coms, 32 * in.m) && rd(in.states, 208) && rd(in.rng, 64) && rd(in.exp_v, 1) && rd(in.exp_ts, 208) && rd(in.fresh, 1);
    fclose(f);
    return ok;
}
// CPU-bandwidth throttling of the container (cgroup v2 cpu.stat, v1 cpu/cpu.stat): periods seen, periods throttled, time throttled (us).
// A pool under a CPU quota stalls for the rest of a 100 ms period once its threads have used the quota up: every caller's latency then
// shows a spike at a period boundary -- not the library's, but it must be told apart from the library's.
struct cg_stat { long long periods = -1, throttled = -1, usec = -1; };
static cg_stat read_cg() {
    cg_stat r;
    const char *paths[] = {"/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat", "/sys/fs/cgroup/cpu,cpuacct/cpu.stat"};
    for (const char *pth : paths) {
        FILE *f = fopen(pth, "r");
        if (!f) continue;
        char k[64];
        long long v;
        while (fscanf(f, "%63s %lld", k, &v) == 2) {
            if (!strcmp(k, "nr_periods")) r.periods = v;
            else if (!strcmp(k, "nr_throttled")) r.throttled = v;
            else if (!strcmp(k, "throttled_usec")) r.usec = v;
            else if (!strcmp(k, "throttled_time")) r.usec = v / 1000;
        }
        fclose(f);
        if (r.periods >= 0) break;
    }
    return r;
}
extern "C" int bpgpu_internal_pool_tune(bpgpu_pool *, const char *, int64_t);
static double g_run0 = 0;
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: combine_rate <inputs> <seconds> threads T [B] | tickets T Q | big T NB | msm T B   (env: BP_LANES, BP_W, BP_OPTS=key=val,...)\n");
        return 2;
    }
    inputs in;
    if (!load(argv[1], in)) {
        fprintf(stderr, "cannot read %s\n", argv[1]);
        return 2;
    }
    const double seconds = atof(argv[2]);
    const std::string mode = argv[3];
    const int T = atoi(argv[4]);
    const int arg2 = argc > 5 ? atoi(argv[5]) : 1;
    const int dev = 0;
    bpgpu_pool *pool = nullptr;
    const int lanes = getenv("BP_LANES") ? atoi(getenv("BP_LANES")) : 8;
    int rc = bpgpu_pool_create(&dev, 1, lanes, &pool);
    if (rc) {
        fprintf(stderr, "bpgpu_pool_create: %d\n", rc);
        return 1;
    }
    if (getenv("BP_W")) bpgpu_pool_set_option(pool, "fixed_window_bits", atoi(getenv("BP_W")));
    if (getenv("BP_TRACE")) bpgpu_pool_set_option(pool, "combine_trace", 8192);
    // msm mode: [u32 n, m, nu, count] gen_scalars | uniq_scalars | uniq_points | expected encodings (32 B each)
    struct {
        uint32_t n = 0, m = 0, nu = 0, count = 0;
        std::vector<uint8_t> gs, us, up, exp;
    } mi;
    if (mode == "msm") {
        const char *mp = getenv("BP_MSM_INPUTS");
        FILE *f = mp ? fopen(mp, "rb") : nullptr;
        uint32_t h[4];
        if (!f || fread(h, 4, 4, f) != 4) {
            fprintf(stderr, "msm mode needs BP_MSM_INPUTS\n");
            return 2;
        }
        mi.n = h[0], mi.m = h[1], mi.nu = h[2], mi.count = h[3];
        const size_t ng = 2 * (size_t)mi.n * mi.m + 2;
        mi.gs.resize(ng * 32 * mi.count), mi.us.resize((size_t)mi.nu * 32 * mi.count), mi.up.resize((size_t)mi.nu * 32 * mi.count), mi.exp.resize(32 * (size_t)mi.count);
        if (fread(mi.gs.data(), 1, mi.gs.size(), f) != mi.gs.size() || fread(mi.us.data(), 1, mi.us.size(), f) != mi.us.size() ||
            fread(mi.up.data(), 1, mi.up.size(), f) != mi.up.size() || fread(mi.exp.data(), 1, mi.exp.size(), f) != mi.exp.size()) {
            fprintf(stderr, "short msm input file\n");
            return 2;
        }
        fclose(f);
        in.n = mi.n, in.m = mi.m;   // (the generator set the pool derives below)
    }
    if (const char *o = getenv("BP_OPTS")) {
        std::string s = o;
        size_t p = 0;
        while (p < s.size()) {
            size_t e = s.find(',', p);
            if (e == std::string::npos) e = s.size();
            const std::string kv = s.substr(p, e - p);
            const size_t q = kv.find('=');
            if (q != std::string::npos && bpgpu_pool_set_option(pool, kv.substr(0, q).c_str(), atoll(kv.c_str() + q + 1)))
                fprintf(stderr, "option %s refused: %s\n", kv.c_str(), bpgpu_pool_last_error(pool));
            p = e + 1;
        }
    }
    if (const char *o = getenv("BP_TUNE")) {   // constants of the sealing policies (not options: the library's test hook), "key=value,..."
        std::string s = o;
        size_t p = 0;
        while (p < s.size()) {
            size_t e = s.find(',', p);
            if (e == std::string::npos) e = s.size();
            const std::string kv = s.substr(p, e - p);
            const size_t q = kv.find('=');
            if (q != std::string::npos && bpgpu_internal_pool_tune(pool, kv.substr(0, q).c_str(), atoll(kv.c_str() + q + 1)))
                fprintf(stderr, "tune %s refused\n", kv.c_str());
            p = e + 1;
        }
    }
    rc = bpgpu_pool_gens_create(pool, in.n, in.m);
    if (rc) {
        fprintf(stderr, "gens: %s\n", bpgpu_pool_last_error(pool));
        return 1;
    }
    const size_t PL = in.proof_len, CM = 32 * in.m;
    std::atomic<uint64_t> total{0}, mismatches{0}, errors{0};
    std::vector<std::vector<float>> lat(T), when(T);   // latency of every call / when it returned (ms since the run began)
    std::atomic<bool> stop{false};
    std::atomic<int> ready{0};
    auto check = [&](size_t idx, const uint8_t *v, const uint8_t *ts) {
        bool ok = v[0] == in.exp_v[idx];
        if (ts && in.exp_v[idx] != 2) ok = ok && memcmp(ts, &in.exp_ts[idx * 208], 208) == 0;
        if (ts && in.exp_v[idx] == 2) ok = ok && memcmp(ts, &in.states[idx * 208], 208) == 0;
        if (!ok) mismatches++;
    };
    auto worker = [&](int t) {
        std::vector<float> &L = lat[t], &Wn = when[t];
        L.reserve(1 << 20);
        Wn.reserve(1 << 20);
        uint64_t done = 0;
        size_t cur = ((size_t)t * 7919) % in.count;
        ready++;
        while (ready.load() < T) std::this_thread::yield();
        if (mode == "threads") {
            const int B = arg2;
            std::vector<uint8_t> pr(B * PL), cm(B * CM), st(B * 208), rg(B * 64), v(B), ts(B * 208);
            std::vector<size_t> idx(B);
            while (!stop.load(std::memory_order_relaxed)) {
                for (int b = 0; b < B; b++) {
                    idx[b] = cur;
                    cur = (cur + 1) % in.count;
                    memcpy(&pr[b * PL], &in.proofs[idx[b] * PL], PL);
                    memcpy(&cm[b * CM], &in.coms[idx[b] * CM], CM);
                    memcpy(&st[b * 208], &in.states[idx[b] * 208], 208);
                    memcpy(&rg[b * 64], &in.rng[idx[b] * 64], 64);
                }
                const double t0 = now_s();
                const int r = bpgpu_pool_rangeproof_verify_ts(pool, in.n, in.m, B, pr.data(), PL, cm.data(), st.data(), 208, rg.data(), v.data(), nullptr, ts.data());
                { const double e_ = now_s(); L.push_back((float)((e_ - t0) * 1e3)); Wn.push_back((float)((e_ - g_run0) * 1e3)); }
                if (r) errors++;
                else
                    for (int b = 0; b < B; b++) check(idx[b], &v[b], &ts[b * 208]);
                done += B;
            }
        } else if (mode == "tickets") {
            const int Q = arg2;
            struct slot {
                bpgpu_ticket *t = nullptr;
                size_t idx = 0;
                double t0 = 0;
                uint8_t v[1], ts[208];
            };
            std::vector<slot> ring(Q);
            size_t head = 0;
            while (!stop.load(std::memory_order_relaxed)) {
                slot &s = ring[head];
                if (s.t) {
                    if (bpgpu_pool_ticket_wait(pool, s.t)) errors++;
                    else check(s.idx, s.v, s.ts);
                    { const double e_ = now_s(); L.push_back((float)((e_ - s.t0) * 1e3)); Wn.push_back((float)((e_ - g_run0) * 1e3)); }
                    s.t = nullptr;
                    done++;
                }
                s.idx = cur;
                cur = (cur + 1) % in.count;
                s.t0 = now_s();
                if (bpgpu_pool_rangeproof_submit_ts(pool, in.n, in.m, 1, &in.proofs[s.idx * PL], PL, &in.coms[s.idx * CM], &in.states[s.idx * 208], 208, &in.rng[s.idx * 64],
                                                    s.v, nullptr, s.ts, &s.t)) {
                    errors++;
                    s.t = nullptr;
                }
                head = (head + 1) % Q;
            }
            for (slot &s : ring)
                if (s.t) {
                    if (bpgpu_pool_ticket_wait(pool, s.t)) errors++;
                    else check(s.idx, s.v, s.ts);
                    done++;
                }
        } else if (mode == "msm") {
            const size_t B = arg2, ng = 2 * (size_t)mi.n * mi.m + 2, nu = mi.nu;
            std::vector<uint8_t> gs(B * ng * 32), us(B * nu * 32), up(B * nu * 32), out(B * 32), st(B);
            size_t k = ((size_t)t * 13) % mi.count;
            while (!stop.load(std::memory_order_relaxed)) {
                std::vector<size_t> idx(B);
                for (size_t b = 0; b < B; b++) {
                    idx[b] = k;
                    k = (k + 1) % mi.count;
                    memcpy(&gs[b * ng * 32], &mi.gs[idx[b] * ng * 32], ng * 32);
                    memcpy(&us[b * nu * 32], &mi.us[idx[b] * nu * 32], nu * 32);
                    memcpy(&up[b * nu * 32], &mi.up[idx[b] * nu * 32], nu * 32);
                }
                const double t0 = now_s();
                const int r = bpgpu_pool_msm_batch_shared(pool, mi.n, mi.m, B, nu, gs.data(), us.data(), up.data(), out.data(), st.data());
                { const double e_ = now_s(); L.push_back((float)((e_ - t0) * 1e3)); Wn.push_back((float)((e_ - g_run0) * 1e3)); }
                if (r) errors++;
                else
                    for (size_t b = 0; b < B; b++)
                        if (st[b] != 0 || memcmp(&out[b * 32], &mi.exp[idx[b] * 32], 32) != 0) mismatches++;
                done += B;
            }
        } else {   // big: label-mode calls of NB proofs.  Items proven on the FRESH transcript (Transcript::new("combine-rate")) keep the
                   // oracle's verdict there; the pre-bound ones are proofs of a different statement: VerificationError (FormatError stays)
            const int NB = arg2;
            std::vector<uint8_t> pr((size_t)NB * PL), cm((size_t)NB * CM), rg((size_t)NB * 64), v(NB);
            for (int b = 0; b < NB; b++) {
                const size_t i = (cur + b) % in.count;
                memcpy(&pr[b * PL], &in.proofs[i * PL], PL);
                memcpy(&cm[b * CM], &in.coms[i * CM], CM);
                memcpy(&rg[b * 64], &in.rng[i * 64], 64);
            }
            std::vector<uint8_t> want(NB);
            for (int b = 0; b < NB; b++) {
                const size_t i = (cur + b) % in.count;
                want[b] = in.fresh[i] ? in.exp_v[i] : (in.exp_v[i] == 2 ? 2 : 1);
            }
            while (!stop.load(std::memory_order_relaxed)) {
                const double t0 = now_s();
                const int r = bpgpu_pool_rangeproof_verify(pool, in.n, in.m, NB, pr.data(), PL, cm.data(), (const uint8_t *)"combine-rate", 12, rg.data(), v.data(), nullptr);
                { const double e_ = now_s(); L.push_back((float)((e_ - t0) * 1e3)); Wn.push_back((float)((e_ - g_run0) * 1e3)); }
                if (r) errors++;
                else if (v != want) mismatches++;
                done += NB;
            }
        }
        total += done;
    };
    // watchdog: a run that does not end within its time + 20 s says where the queue stands, then gives up (the caller sees exit code 3
    // and the state on stderr instead of a silent timeout)
    std::atomic<int> phase{0};   // 0 running, 1 workers joined, 2 pool destroyed
    std::thread([&phase, pool, seconds] {
        const double limit = now_s() + seconds + 20.0;
        while (now_s() < limit) {
            if (phase.load() == 2) return;
            std::this_thread::sleep_for(std::chrono::milliseconds(50));
        }
        if (phase.load() == 2) return;
        static char buf[16384];
        fprintf(stderr, "combine_rate: WATCHDOG -- still in phase %d (0 = workers running, 1 = destroying the pool) 20 s after the run should have ended\n", phase.load());
        if (phase.load() == 0 && bpgpu_internal_pool_state(pool, buf, sizeof buf) == 0) fputs(buf, stderr);
        fflush(stderr);
        fflush(stdout);
        _Exit(3);
    }).detach();
    std::vector<std::thread> th;
    const double t0 = now_s();
    g_run0 = now_s();
    const cg_stat cg0 = read_cg();
    for (int t = 0; t < T; t++) th.emplace_back(worker, t);
    while (ready.load() < T) std::this_thread::yield();
    const double t1 = now_s();
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    stop = true;
    for (auto &x : th) x.join();
    phase = 1;
    const double t2 = now_s();
    std::vector<float> all, steady;
    for (auto &l : lat) all.insert(all.end(), l.begin(), l.end());
    const float warm_ms = getenv("BP_WARM_MS") ? (float)atof(getenv("BP_WARM_MS")) : 100.0f;   // calls that RETURNED within this time of the start are "start-up"
    for (int t = 0; t < T; t++)
        for (size_t i = 0; i < lat[t].size(); i++)
            if (when[t][i] > warm_ms) steady.push_back(lat[t][i]);
    std::sort(all.begin(), all.end());
    std::sort(steady.begin(), steady.end());
    auto pct = [&](double q) { return all.empty() ? 0.0 : (double)all[(size_t)(q * (all.size() - 1))]; };
    int64_t chains = 0, cproofs = 0, iss = 0, cmp = 0, polls = 0;
    bpgpu_pool_get_option(pool, "stat_svc_issue_us", &iss);
    bpgpu_pool_get_option(pool, "stat_svc_complete_us", &cmp);
    bpgpu_pool_get_option(pool, "stat_svc_polls", &polls);
    bpgpu_pool_get_option(pool, "stat_combined_chains", &chains);
    bpgpu_pool_get_option(pool, "stat_combined_proofs", &cproofs);
    {
        // every call above 5 x p99: how long, when it returned, which call of its thread it was
        const double lim = 5.0 * (all.empty() ? 0.0 : (double)all[(size_t)(0.99 * (all.size() - 1))]);
        std::string o = "[";
        int n_out = 0, n_start = 0;
        for (int t = 0; t < T; t++)
            for (size_t i = 0; i < lat[t].size(); i++)
                if (lat[t][i] > lim) {
                    if (when[t][i] <= warm_ms) n_start++;
                    if (n_out < 24) {
                        char buf[96];
                        snprintf(buf, sizeof buf, "%s[%.2f, %.1f, %d, %zu]", n_out ? ", " : "", lat[t][i], when[t][i], t, i);
                        o += buf;
                    }
                    n_out++;
                }
        o += "]";
        auto spct = [&](double q) { return steady.empty() ? 0.0 : (double)steady[(size_t)(q * (steady.size() - 1))]; };
        const cg_stat cg1 = read_cg();
        fprintf(stderr, "{\"outliers_above_5x_p99\": %d, \"of_them_in_first_%.0f_ms\": %d, \"listed_as_lat_ms_when_ms_thread_call\": %s, "
                        "\"steady_lat_ms\": {\"p50\": %.3f, \"p99\": %.3f, \"max\": %.3f, \"calls\": %zu}, "
                        "\"cgroup_cpu\": {\"periods\": %lld, \"periods_throttled\": %lld, \"throttled_ms\": %.1f}}\n",
                n_out, warm_ms, n_start, o.c_str(), spct(0.5), spct(0.99), steady.empty() ? 0.0 : (double)steady.back(), steady.size(),
                cg1.periods - cg0.periods, cg1.throttled - cg0.throttled, (double)(cg1.usec - cg0.usec) / 1e3);
    }
    printf("{\"mode\": \"%s\", \"threads\": %d, \"arg\": %d, \"seconds\": %.2f, \"verifications\": %llu, \"rate_per_s\": %.0f, \"calls\": %zu, \"lat_ms\": {\"p50\": %.3f, "
           "\"p90\": %.3f, \"p99\": %.3f, \"max\": %.3f}, \"chains\": %lld, \"proofs_per_chain\": %.1f, \"mismatches\": %llu, \"errors\": %llu, \"startup_s\": %.2f, \"svc\": {\"issue_us_per_chain\": %.1f, \"complete_us_per_chain\": %.1f, \"polls\": %lld}}\n",
           mode.c_str(), T, arg2, t2 - t1, (unsigned long long)total.load(), (double)total.load() / (t2 - t1), all.size(), pct(0.5), pct(0.9), pct(0.99),
           all.empty() ? 0.0 : (double)all.back(), (long long)chains, chains ? (double)cproofs / (double)chains : 0.0, (unsigned long long)mismatches.load(),
           (unsigned long long)errors.load(), t1 - t0, chains ? (double)iss / (double)chains : 0.0, chains ? (double)cmp / (double)chains : 0.0, (long long)polls);
    fflush(stdout);   // (the figures survive whatever happens during the teardown)
    if (getenv("BP_TRACE")) bpgpu_pool_trace_dump(pool, getenv("BP_TRACE"));
    bpgpu_pool_destroy(pool);
    phase = 2;
    return (mismatches.load() || errors.load()) ? 1 : 0;
}
