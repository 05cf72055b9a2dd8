// TEST-ONLY (tests/cpu_pool): what the fake back end "computes".  Every result is a function of that item's inputs only, so the
// driver can recompute it and check that each request received its own results, whichever launch chain carried it.
#ifndef BP_FAKE_MODEL_H
#define BP_FAKE_MODEL_H
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define FAKE_VERDICT_DIRECT 2   // what the fake's "ordinary entry point" reports for requests no chain can take

static inline uint64_t fake_mix(uint64_t h, const uint8_t *p, size_t n) {
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        h = (h ^ w) * 0x100000001b3ull;
        h ^= h >> 29;
    }
    for (; i < n; i++) h = (h ^ p[i]) * 0x100000001b3ull;
    return h;
}
static inline void fake_stream(uint64_t h, uint8_t *out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        h = h * 6364136223846793005ull + 1442695040888963407ull;
        out[i] = (uint8_t)(h >> 56);
    }
}
// Transcript::new(label): 200 "sponge" bytes from the label, a STROBE position that depends on the label's LENGTH only (as the real one does)
static inline void fake_transcript_new(const uint8_t *label, size_t len, uint8_t st[208]) {
    fake_stream(fake_mix(0x1234567ull + len, label, len), st, 200);
    st[200] = (uint8_t)((len * 7 + 13) % 166);
    st[201] = (uint8_t)(len % 100);
    st[202] = (uint8_t)(len & 3);
    memset(st + 203, 0, 5);
}
static inline void fake_rp_result(size_t n, size_t m, const uint8_t *proof, size_t proof_len, const uint8_t *coms, const uint8_t *ts, uint8_t *verdict, uint8_t *ts_out,
                                  uint8_t *msm) {
    uint64_t h = 0xcbf29ce484222325ull ^ (n * 131 + m);
    h = fake_mix(h, proof, proof_len);
    h = fake_mix(h, coms, m * 32);
    h = fake_mix(h, ts, 203);
    *verdict = (h % 7 == 0) ? 1 : 0;
    if (ts_out) {
        fake_stream(h ^ 0x55, ts_out, 200);
        ts_out[200] = (uint8_t)((ts[200] + 31) % 166), ts_out[201] = ts[201], ts_out[202] = ts[202];
        memset(ts_out + 203, 0, 5);
    }
    if (msm) fake_stream(h ^ 0xaa, msm, 32);
}
static inline void fake_msm_shared_result(const uint8_t *gs, size_t ng, const uint8_t *us, const uint8_t *up, size_t nu, uint8_t *out, uint8_t *status) {
    uint64_t h = fake_mix(0x9e3779b97f4a7c15ull ^ ng ^ (nu << 20), gs, ng * 32);
    if (nu) h = fake_mix(fake_mix(h, us, nu * 32), up, nu * 32);
    fake_stream(h, out, 32);
    *status = (h % 11 == 0) ? 1 : 0;
}
static inline void fake_msm_result(const uint8_t *s, const uint8_t *p, size_t nt, uint8_t *out, uint8_t *status) {
    const uint64_t h = fake_mix(fake_mix(0x7f4a7c15ull ^ nt, s, nt * 32), p, nt * 32);
    fake_stream(h, out, 32);
    *status = (h % 11 == 0) ? 1 : 0;
}
static inline void fake_ipp_result(size_t n, const uint8_t *proof, size_t proof_len, const uint8_t *st0, const uint8_t *Gf, const uint8_t *Hf, const uint8_t *P, const uint8_t *Q,
                                   const uint8_t *G, const uint8_t *H, uint8_t *verdict, uint8_t *msm) {
    uint64_t h = fake_mix(0x51ull ^ n, proof, proof_len);
    h = fake_mix(h, st0, 203);
    h = fake_mix(fake_mix(fake_mix(fake_mix(h, Gf, n * 32), Hf, n * 32), G, n * 32), H, n * 32);
    h = fake_mix(fake_mix(h, P, 32), Q, 32);
    *verdict = (h % 5 == 0) ? 1 : 0;
    if (msm) fake_stream(h ^ 0x77, msm, 32);
}
#endif
