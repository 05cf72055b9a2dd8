/* ORACLE -- TEST INFRASTRUCTURE ONLY.  See ../bp_oracle.h.
 * Restates, function by function:
 *   gens ................ /root/reference/src/generators.rs:44-53, 58-104, 157-259
 *   parse ............... src/range_proof/mod.rs:504-538, src/inner_product_proof.rs:373-407
 *   verification_scalars  src/inner_product_proof.rs:198-253
 *   verify .............. src/range_proof/mod.rs:345-452, delta 587-593
 *   sum_of_powers ....... src/util.rs:240-261
 *   prover .............. src/range_proof/{party,dealer}.rs, src/inner_product_proof.rs:38-193
 */
#include "../bp_oracle.h"
#include "ge.h"
#include "merlin.h"
#include <stdlib.h>
#include <pthread.h>
#include <time.h>
#include <malloc.h>

struct oracle_gens {
    size_t gens_capacity, party_capacity;
    ge_p3 *G, *H;            /* [party][i] */
    uint8_t *Gc, *Hc;        /* compressed */
    ge_p3 B, B_blinding;
    uint8_t Bc[32], Bbc[32];
};

static double now_s(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* ---------------- generators ---------------- */
static void generators_chain(ge_p3 *out, uint8_t *outc, uint8_t tag, uint32_t party, size_t count) {
    keccak_sponge k; shake256_init(&k);
    uint8_t label[5] = {tag, (uint8_t)party, (uint8_t)(party >> 8), (uint8_t)(party >> 16), (uint8_t)(party >> 24)};
    sponge_absorb(&k, (const uint8_t *)"GeneratorsChain", 15);
    sponge_absorb(&k, label, 5);
    for (size_t i = 0; i < count; i++) {
        uint8_t u[64]; shake256_squeeze(&k, u, 64);
        ristretto_from_uniform_bytes(&out[i], u);
        ristretto_compress(outc + 32 * i, &out[i]);
    }
}

oracle_gens *oracle_gens_new(size_t gens_capacity, size_t party_capacity) {
    ge_init();
    /* keep the per-verification scratch (up to a few MB) on the per-thread heaps: with the default
       128 KiB mmap threshold every verification mmap()s/munmap()s and the batch drivers serialise on
       the kernel's address-space lock beyond ~32 threads */
    mallopt(M_MMAP_THRESHOLD, 16 << 20);   /* (glibc caps this at 32 MiB) */
    mallopt(M_TRIM_THRESHOLD, 512 << 20);
    mallopt(M_ARENA_MAX, 512);
    oracle_gens *g = calloc(1, sizeof *g);
    g->gens_capacity = gens_capacity; g->party_capacity = party_capacity;
    size_t tot = gens_capacity * party_capacity;
    g->G = malloc((tot + 1) * sizeof(ge_p3)); g->H = malloc((tot + 1) * sizeof(ge_p3));
    g->Gc = malloc(tot * 32 + 32); g->Hc = malloc(tot * 32 + 32);
    for (size_t p = 0; p < party_capacity; p++) {
        generators_chain(g->G + p * gens_capacity, g->Gc + 32 * p * gens_capacity, 'G', (uint32_t)p, gens_capacity);
        generators_chain(g->H + p * gens_capacity, g->Hc + 32 * p * gens_capacity, 'H', (uint32_t)p, gens_capacity);
    }
    ristretto_decompress(&g->B, RISTRETTO_BASEPOINT_COMPRESSED);
    memcpy(g->Bc, RISTRETTO_BASEPOINT_COMPRESSED, 32);
    uint8_t h[64]; sha3_512(h, RISTRETTO_BASEPOINT_COMPRESSED, 32);
    ristretto_from_uniform_bytes(&g->B_blinding, h);
    ristretto_compress(g->Bbc, &g->B_blinding);
    return g;
}
void oracle_gens_free(oracle_gens *g) {
    if (!g) return;
    free(g->G); free(g->H); free(g->Gc); free(g->Hc); free(g);
}
void oracle_gens_export(const oracle_gens *g, uint8_t *G_out, uint8_t *H_out, uint8_t B[32], uint8_t Bb[32]) {
    size_t tot = g->gens_capacity * g->party_capacity;
    if (G_out) memcpy(G_out, g->Gc, tot * 32);
    if (H_out) memcpy(H_out, g->Hc, tot * 32);
    if (B) memcpy(B, g->Bc, 32);
    if (Bb) memcpy(Bb, g->Bbc, 32);
}

/* ---------------- primitives ---------------- */
int oracle_point_decompress_ok(const uint8_t in[32]) { ge_p3 p; return ristretto_decompress(&p, in) == 0; }
void oracle_from_uniform_bytes(const uint8_t in[64], uint8_t out[32]) {
    ge_p3 p; ristretto_from_uniform_bytes(&p, in); ristretto_compress(out, &p);
}
void oracle_scalar_from_wide(const uint8_t in[64], uint8_t out[32]) { sc s; sc_from_wide(&s, in); sc_tobytes(out, &s); }
void oracle_scalar_mul(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) {
    sc x, y, z; sc_from_bytes_mod_order(&x, a); sc_from_bytes_mod_order(&y, b); sc_mul(&z, &x, &y); sc_tobytes(out, &z);
}
void oracle_scalar_invert(const uint8_t a[32], uint8_t out[32]) {
    sc x, z; sc_from_bytes_mod_order(&x, a); sc_invert(&z, &x); sc_tobytes(out, &z);
}
void oracle_merlin_kat(const uint8_t *label, size_t label_len, const char *msg_label, const uint8_t *msg, size_t msg_len,
                       const char *ch_label, uint8_t *out, size_t out_len) {
    merlin_transcript t; merlin_init(&t, label, label_len);
    merlin_append_message(&t, msg_label, msg, msg_len);
    merlin_challenge_bytes(&t, ch_label, out, out_len);
}
void oracle_shake256(const uint8_t *in, size_t n, uint8_t *out, size_t out_len) {
    keccak_sponge k; shake256_init(&k); sponge_absorb(&k, in, n); shake256_squeeze(&k, out, out_len);
}
void oracle_sha3_512(const uint8_t *in, size_t n, uint8_t out[64]) { sha3_512(out, in, n); }

static __thread uint64_t last_msm_ops = 0;
uint64_t oracle_last_msm_ops(void) { return last_msm_ops; }
const char *oracle_backend(void) { return ge_backend(); }

static void msm_dispatch(ge_p3 *r, size_t n, const sc *s, const ge_p3 *p, int algo) {
    if (algo == 1) ge_msm_straus(r, n, s, p);
    else if (algo == 2) ge_msm_pippenger(r, n, s, p);
    else ge_msm_vartime(r, n, s, p);
    last_msm_ops = ge_op_counter;
}

int oracle_msm(size_t n, const uint8_t *scalars, const uint8_t *points, int algo, uint8_t out[32]) {
    ge_init();
    sc *s = malloc((n + 1) * sizeof(sc)); ge_p3 *p = malloc((n + 1) * sizeof(ge_p3));
    int bad = 0;
    for (size_t i = 0; i < n; i++) {
        sc_from_bytes_mod_order(&s[i], scalars + 32 * i);
        if (ristretto_decompress(&p[i], points + 32 * i) != 0) bad = 1;
    }
    if (bad) { memset(out, 0, 32); free(s); free(p); return 1; }
    ge_p3 r; msm_dispatch(&r, n, s, p, algo);
    ristretto_compress(out, &r);
    free(s); free(p);
    return 0;
}

/* ---------------- transcript protocol (src/transcript.rs:43-95) ---------------- */
static int is_zero32(const uint8_t *p) { uint8_t r = 0; for (int i = 0; i < 32; i++) r |= p[i]; return r == 0; }
static int validate_and_append_point(merlin_transcript *t, const char *label, const uint8_t *p) {
    if (is_zero32(p)) return -1;
    merlin_append_message(t, label, p, 32); return 0;
}
static void challenge_scalar(merlin_transcript *t, const char *label, sc *out) {
    uint8_t buf[64]; merlin_challenge_bytes(t, label, buf, 64); sc_from_wide(out, buf);
}
static void append_scalar(merlin_transcript *t, const char *label, const sc *s) {
    uint8_t b[32]; sc_tobytes(b, s); merlin_append_message(t, label, b, 32);
}

/* ---------------- parsed proof ---------------- */
typedef struct {
    const uint8_t *A, *S, *T1, *T2;
    sc t_x, t_x_blinding, e_blinding, a, b;
    size_t lg_n;
    const uint8_t *LR;       /* lg_n pairs (L_i, R_i), 64 bytes each */
} parsed_proof;

static int parse_proof(parsed_proof *pp, const uint8_t *proof, size_t len) {
    if (len % 32 != 0 || len < 7 * 32) return ORACLE_ERR_FORMAT;
    pp->A = proof; pp->S = proof + 32; pp->T1 = proof + 64; pp->T2 = proof + 96;
    if (sc_from_canonical_bytes(&pp->t_x, proof + 128)) return ORACLE_ERR_FORMAT;
    if (sc_from_canonical_bytes(&pp->t_x_blinding, proof + 160)) return ORACLE_ERR_FORMAT;
    if (sc_from_canonical_bytes(&pp->e_blinding, proof + 192)) return ORACLE_ERR_FORMAT;
    size_t ne = (len - 224) / 32;
    if (ne < 2 || (ne - 2) % 2 != 0) return ORACLE_ERR_FORMAT;
    pp->lg_n = (ne - 2) / 2;
    if (pp->lg_n >= 32) return ORACLE_ERR_FORMAT;
    pp->LR = proof + 224;
    if (sc_from_canonical_bytes(&pp->a, proof + 224 + 64 * pp->lg_n)) return ORACLE_ERR_FORMAT;
    if (sc_from_canonical_bytes(&pp->b, proof + 224 + 64 * pp->lg_n + 32)) return ORACLE_ERR_FORMAT;
    return 0;
}

/* src/util.rs:240-261 */
static void sum_of_powers(sc *r, const sc *x, size_t n) {
    sc one; sc_from_u64(&one, 1);
    if (n & (n - 1)) {  /* slow path */
        sc acc, p; sc_0(&acc); p = one;
        for (size_t i = 0; i < n; i++) { sc_add(&acc, &acc, &p); sc_mul(&p, &p, x); }
        *r = acc; return;
    }
    if (n == 0 || n == 1) { sc_from_u64(r, n); return; }
    size_t m = n; sc result, factor, t;
    sc_add(&result, &one, x); factor = *x;
    while (m > 2) {
        sc_mul(&factor, &factor, &factor);
        sc_mul(&t, &factor, &result); sc_add(&result, &result, &t);
        m /= 2;
    }
    *r = result;
}
/* src/range_proof/mod.rs:587-593 */
static void delta(sc *r, size_t n, size_t m, const sc *y, const sc *z) {
    sc sum_y, sum_2, sum_z, two, zz, a, b;
    sc_from_u64(&two, 2);
    sum_of_powers(&sum_y, y, n * m); sum_of_powers(&sum_2, &two, n); sum_of_powers(&sum_z, z, m);
    sc_mul(&zz, z, z);
    sc_sub(&a, z, &zz); sc_mul(&a, &a, &sum_y);
    sc_mul(&b, &zz, z); sc_mul(&b, &b, &sum_2); sc_mul(&b, &b, &sum_z);
    sc_sub(r, &a, &b);
}

/* Transcript replay + scalar assembly.  On success fills scalars[N] and
 * pts[N] (32-byte encodings, reference order) and *n_terms. */
static int verify_terms_core_t(const oracle_gens *g, const uint8_t *proof, size_t proof_len,
                               const uint8_t *commitments, size_t m, size_t n,
                               merlin_transcript *tp, const uint8_t rng64[64],
                               sc **scalars_out, uint8_t **pts_out, size_t *n_terms);
static int verify_terms_core(const oracle_gens *g, const uint8_t *proof, size_t proof_len,
                             const uint8_t *commitments, size_t m, size_t n,
                             const uint8_t *label, size_t label_len, const uint8_t rng64[64],
                             sc **scalars_out, uint8_t **pts_out, size_t *n_terms) {
    merlin_transcript t; merlin_init(&t, label, label_len);   /* &mut Transcript::new(label) */
    return verify_terms_core_t(g, proof, proof_len, commitments, m, n, &t, rng64, scalars_out, pts_out, n_terms);
}
/* the caller's transcript may already hold messages; it is left advanced (mod.rs:345-353: transcript: &mut Transcript) */
static int verify_terms_core_t(const oracle_gens *g, const uint8_t *proof, size_t proof_len,
                               const uint8_t *commitments, size_t m, size_t n,
                               merlin_transcript *tp, const uint8_t rng64[64],
                               sc **scalars_out, uint8_t **pts_out, size_t *n_terms) {
    parsed_proof pp;
    int rc = parse_proof(&pp, proof, proof_len);
    if (rc) return rc;
    if (!(n == 8 || n == 16 || n == 32 || n == 64)) return ORACLE_ERR_INVALID_BITSIZE;
    if (g->gens_capacity < n) return ORACLE_ERR_INVALID_GENERATORS_LENGTH;
    if (g->party_capacity < m) return ORACLE_ERR_INVALID_GENERATORS_LENGTH;

#define t (*tp)
    merlin_append_message(&t, "dom-sep", (const uint8_t *)"rangeproof v1", 13);
    merlin_append_u64(&t, "n", n); merlin_append_u64(&t, "m", m);
    for (size_t j = 0; j < m; j++) merlin_append_message(&t, "V", commitments + 32 * j, 32);
    if (validate_and_append_point(&t, "A", pp.A)) return ORACLE_ERR_VERIFICATION;
    if (validate_and_append_point(&t, "S", pp.S)) return ORACLE_ERR_VERIFICATION;
    sc y, z, zz, minus_z, x, w, c;
    challenge_scalar(&t, "y", &y); challenge_scalar(&t, "z", &z);
    sc_mul(&zz, &z, &z); sc_neg(&minus_z, &z);
    if (validate_and_append_point(&t, "T_1", pp.T1)) return ORACLE_ERR_VERIFICATION;
    if (validate_and_append_point(&t, "T_2", pp.T2)) return ORACLE_ERR_VERIFICATION;
    challenge_scalar(&t, "x", &x);
    append_scalar(&t, "t_x", &pp.t_x);
    append_scalar(&t, "t_x_blinding", &pp.t_x_blinding);
    append_scalar(&t, "e_blinding", &pp.e_blinding);
    challenge_scalar(&t, "w", &w);
    sc_from_wide(&c, rng64);

    /* verification_scalars (ipp.rs:198-253) */
    size_t nm = n * m, lg_n = pp.lg_n;
    if (lg_n >= 32 || nm != ((size_t)1 << lg_n)) return ORACLE_ERR_VERIFICATION;
    merlin_append_message(&t, "dom-sep", (const uint8_t *)"ipp v1", 6);
    merlin_append_u64(&t, "n", nm);
    sc u_sq[32], u_inv_sq[32], allinv;
    sc_from_u64(&allinv, 1);
    for (size_t i = 0; i < lg_n; i++) {
        if (validate_and_append_point(&t, "L", pp.LR + 64 * i)) return ORACLE_ERR_VERIFICATION;
        if (validate_and_append_point(&t, "R", pp.LR + 64 * i + 32)) return ORACLE_ERR_VERIFICATION;
        sc u, ui; challenge_scalar(&t, "u", &u);
        sc_invert(&ui, &u);
        sc_mul(&allinv, &allinv, &ui);
        sc_mul(&u_sq[i], &u, &u); sc_mul(&u_inv_sq[i], &ui, &ui);
    }
    sc *s = malloc((nm + 1) * sizeof(sc));
    s[0] = allinv;
    for (size_t i = 1; i < nm; i++) {
        size_t lg_i = 63 - (size_t)__builtin_clzll((unsigned long long)i);
        size_t k = (size_t)1 << lg_i;
        sc_mul(&s[i], &s[i - k], &u_sq[(lg_n - 1) - lg_i]);
    }

    size_t N = 2 * nm + 2 * lg_n + m + 6;
    sc *sc_out = malloc(N * sizeof(sc));
    uint8_t *pt_out = malloc(N * 32);
    size_t o = 0;
    sc cx, cxx, t0, t1;
    sc_from_u64(&sc_out[o], 1); memcpy(pt_out + 32 * o, pp.A, 32); o++;
    sc_out[o] = x; memcpy(pt_out + 32 * o, pp.S, 32); o++;
    sc_mul(&cx, &c, &x); sc_out[o] = cx; memcpy(pt_out + 32 * o, pp.T1, 32); o++;
    sc_mul(&cxx, &cx, &x); sc_out[o] = cxx; memcpy(pt_out + 32 * o, pp.T2, 32); o++;
    for (size_t i = 0; i < lg_n; i++) { sc_out[o] = u_sq[i]; memcpy(pt_out + 32 * o, pp.LR + 64 * i, 32); o++; }
    for (size_t i = 0; i < lg_n; i++) { sc_out[o] = u_inv_sq[i]; memcpy(pt_out + 32 * o, pp.LR + 64 * i + 32, 32); o++; }
    /* -e_blinding - c*t_x_blinding  on B_blinding */
    sc_mul(&t0, &c, &pp.t_x_blinding); sc_add(&t0, &t0, &pp.e_blinding); sc_neg(&sc_out[o], &t0);
    memcpy(pt_out + 32 * o, g->Bbc, 32); o++;
    /* basepoint scalar: w*(t_x - a*b) + c*(delta - t_x) on B */
    sc ab, dl; sc_mul(&ab, &pp.a, &pp.b); sc_sub(&t0, &pp.t_x, &ab); sc_mul(&t0, &w, &t0);
    delta(&dl, n, m, &y, &z); sc_sub(&t1, &dl, &pp.t_x); sc_mul(&t1, &c, &t1);
    sc_add(&sc_out[o], &t0, &t1); memcpy(pt_out + 32 * o, g->Bc, 32); o++;
    /* g_i = -z - a*s_i */
    for (size_t i = 0; i < nm; i++) {
        sc_mul(&t0, &pp.a, &s[i]); sc_sub(&sc_out[o], &minus_z, &t0);
        size_t party = i / n, k = i % n;
        memcpy(pt_out + 32 * o, g->Gc + 32 * (party * g->gens_capacity + k), 32); o++;
    }
    /* h_i = z + y^-i * (zz * z^j 2^k - b * s_{nm-1-i}) */
    sc y_inv, exp_y_inv, exp_z, exp_2, two;
    sc_invert(&y_inv, &y); sc_from_u64(&exp_y_inv, 1); sc_from_u64(&exp_z, 1); sc_from_u64(&two, 2);
    for (size_t j = 0; j < m; j++) {
        sc_from_u64(&exp_2, 1);
        for (size_t k = 0; k < n; k++) {
            size_t i = j * n + k;
            sc z_and_2; sc_mul(&z_and_2, &exp_2, &exp_z);
            sc_mul(&t0, &zz, &z_and_2);
            sc_mul(&t1, &pp.b, &s[nm - 1 - i]);
            sc_sub(&t0, &t0, &t1);
            sc_mul(&t0, &exp_y_inv, &t0);
            sc_add(&sc_out[o], &z, &t0);
            memcpy(pt_out + 32 * o, g->Hc + 32 * (j * g->gens_capacity + k), 32); o++;
            sc_mul(&exp_y_inv, &exp_y_inv, &y_inv);
            sc_add(&exp_2, &exp_2, &exp_2);
        }
        sc_mul(&exp_z, &exp_z, &z);
    }
    /* V_j scalars c*zz*z^j */
    sc czz; sc_mul(&czz, &c, &zz); sc_from_u64(&exp_z, 1);
    for (size_t j = 0; j < m; j++) {
        sc_mul(&sc_out[o], &czz, &exp_z); memcpy(pt_out + 32 * o, commitments + 32 * j, 32); o++;
        sc_mul(&exp_z, &exp_z, &z);
    }
    free(s);
    *scalars_out = sc_out; *pts_out = pt_out; *n_terms = N;
    return 0;
#undef t
}

int oracle_verify_terms(const oracle_gens *g, const uint8_t *proof, size_t proof_len,
                        const uint8_t *commitments, size_t m, size_t n,
                        const uint8_t *label, size_t label_len, const uint8_t rng64[64],
                        uint8_t *scalars_out, uint8_t *points_out, size_t *n_terms) {
    sc *s; uint8_t *p; size_t N;
    int rc = verify_terms_core(g, proof, proof_len, commitments, m, n, label, label_len, rng64, &s, &p, &N);
    if (rc) return rc;
    for (size_t i = 0; i < N; i++) sc_tobytes(scalars_out + 32 * i, &s[i]);
    memcpy(points_out, p, 32 * N);
    *n_terms = N;
    free(s); free(p);
    return 0;
}

static int verify_tail(const oracle_gens *g, sc *s, uint8_t *p, size_t N, size_t m, size_t n, uint8_t msm_out[32]);
int oracle_verify(const oracle_gens *g, const uint8_t *proof, size_t proof_len,
                  const uint8_t *commitments, size_t m, size_t n,
                  const uint8_t *label, size_t label_len, const uint8_t rng64[64], uint8_t msm_out[32]) {
    sc *s; uint8_t *p; size_t N;
    int rc = verify_terms_core(g, proof, proof_len, commitments, m, n, label, label_len, rng64, &s, &p, &N);
    if (rc) return rc;
    return verify_tail(g, s, p, N, m, n, msm_out);
}
/* ---- transcripts across the boundary: 208-byte state = st[200] | pos | pos_begin | cur_flags | 5 zero bytes
 * (the in-memory layout of merlin's Strobe128; include/bpgpu.h BPGPU_TRANSCRIPT_BYTES) ---- */
static void ts_load(merlin_transcript *t, const uint8_t s[208]) { memcpy(t->st, s, 200); t->pos = s[200]; t->pos_begin = s[201]; t->cur_flags = s[202]; }
static void ts_store(uint8_t s[208], const merlin_transcript *t) { memset(s, 0, 208); memcpy(s, t->st, 200); s[200] = t->pos; s[201] = t->pos_begin; s[202] = t->cur_flags; }
void oracle_transcript_new(const uint8_t *label, size_t label_len, uint8_t state[208]) {
    merlin_transcript t; merlin_init(&t, label, label_len); ts_store(state, &t);
}
void oracle_transcript_append_message(uint8_t state[208], const char *label, const uint8_t *msg, size_t n) {
    merlin_transcript t; ts_load(&t, state); merlin_append_message(&t, label, msg, n); ts_store(state, &t);
}
void oracle_transcript_challenge_bytes(uint8_t state[208], const char *label, uint8_t *out, size_t n) {
    merlin_transcript t; ts_load(&t, state); merlin_challenge_bytes(&t, label, out, n); ts_store(state, &t);
}
/* verify_multiple_with_rng(bp_gens, pc_gens, transcript, ...) with the caller's transcript: state in, advanced state out */
int oracle_verify_ts(const oracle_gens *g, const uint8_t *proof, size_t proof_len,
                     const uint8_t *commitments, size_t m, size_t n,
                     uint8_t state[208], const uint8_t rng64[64], uint8_t msm_out[32]) {
    sc *s; uint8_t *p; size_t N;
    merlin_transcript t; ts_load(&t, state);
    int rc = verify_terms_core_t(g, proof, proof_len, commitments, m, n, &t, rng64, &s, &p, &N);
    ts_store(state, &t);
    if (rc) return rc;
    return verify_tail(g, s, p, N, m, n, msm_out);
}
static int verify_tail(const oracle_gens *g, sc *s, uint8_t *p, size_t N, size_t m, size_t n, uint8_t msm_out[32]) {
    int rc;
    size_t lg_n = 0; while (((size_t)1 << lg_n) < n * m) lg_n++;
    ge_p3 *pts = malloc(N * sizeof(ge_p3));
    int bad = 0;
    size_t gen0 = 4 + 2 * lg_n, nm = n * m;
    /* the proof's own points (A, S, T_1, T_2, L_i, R_i, V_j) are decoded (mod.rs:433-443), in one batch; generator terms use the
     * cached decoded points (generators.rs holds them decoded) */
    const uint8_t **enc = malloc(N * sizeof *enc); size_t *where = malloc(N * sizeof *where); size_t nd = 0;
    for (size_t i = 0; i < N; i++) {
        if (i == gen0) pts[i] = g->B_blinding;
        else if (i == gen0 + 1) pts[i] = g->B;
        else if (i >= gen0 + 2 && i < gen0 + 2 + nm) { size_t q = i - gen0 - 2; pts[i] = g->G[(q / n) * g->gens_capacity + q % n]; }
        else if (i >= gen0 + 2 + nm && i < gen0 + 2 + 2 * nm) { size_t q = i - gen0 - 2 - nm; pts[i] = g->H[(q / n) * g->gens_capacity + q % n]; }
        else { enc[nd] = p + 32 * i; where[nd++] = i; }
    }
    {
        ge_p3 *dec = malloc((nd ? nd : 1) * sizeof(ge_p3)); int *drc = malloc((nd ? nd : 1) * sizeof(int));
        ristretto_decompress_many(dec, enc, drc, nd);
        for (size_t k = 0; k < nd; k++) { if (drc[k] != 0) bad = 1; pts[where[k]] = dec[k]; }
        free(dec); free(drc);
    }
    free(enc); free(where);
    if (bad) { free(s); free(p); free(pts); if (msm_out) memset(msm_out, 0xff, 32); return ORACLE_ERR_VERIFICATION; }
    ge_p3 r; msm_dispatch(&r, N, s, pts, 0);
    if (msm_out) ristretto_compress(msm_out, &r);
    rc = ge_is_identity(&r) ? ORACLE_OK : ORACLE_ERR_VERIFICATION;
    free(s); free(p); free(pts);
    return rc;
}

/* ---------------- prover (input synthesis only; NOT constant time) ---------------- */
static void rng_scalar(keccak_sponge *k, sc *out) { uint8_t b[64]; shake256_squeeze(k, b, 64); sc_from_wide(out, b); }
static void inner_product(sc *r, const sc *a, const sc *b, size_t n) {
    sc acc, t; sc_0(&acc);
    for (size_t i = 0; i < n; i++) { sc_mul(&t, &a[i], &b[i]); sc_add(&acc, &acc, &t); }
    *r = acc;
}
static void pedersen_commit(ge_p3 *r, const oracle_gens *g, const sc *v, const sc *bl) {
    sc ss[2] = {*v, *bl}; ge_p3 pp[2] = {g->B, g->B_blinding};
    ge_msm_straus(r, 2, ss, pp);
}

static void ipp_create(merlin_transcript *t, const ge_p3 *Q, const sc *Hf /* y^-i, G factors are 1 */,
                       ge_p3 *G, ge_p3 *H, sc *a, sc *b, size_t n, uint8_t *LR_out, sc *a_out, sc *b_out) {
    merlin_append_message(t, "dom-sep", (const uint8_t *)"ipp v1", 6);
    merlin_append_u64(t, "n", n);
    int first = 1; size_t round = 0;
    sc *sv = malloc((2 * n + 1) * sizeof(sc)); ge_p3 *pv = malloc((2 * n + 1) * sizeof(ge_p3));
    while (n != 1) {
        n /= 2;
        sc *aL = a, *aR = a + n, *bL = b, *bR = b + n;
        ge_p3 *GL = G, *GR = G + n, *HL = H, *HR = H + n;
        sc cL, cR; inner_product(&cL, aL, bR, n); inner_product(&cR, aR, bL, n);
        ge_p3 Lp, Rp;
        for (size_t i = 0; i < n; i++) {
            sv[i] = aL[i]; pv[i] = GR[i];
            if (first) sc_mul(&sv[n + i], &bR[i], &Hf[i]); else sv[n + i] = bR[i];
            pv[n + i] = HL[i];
        }
        sv[2 * n] = cL; pv[2 * n] = *Q;
        ge_msm_vartime(&Lp, 2 * n + 1, sv, pv);
        for (size_t i = 0; i < n; i++) {
            sv[i] = aR[i]; pv[i] = GL[i];
            if (first) sc_mul(&sv[n + i], &bL[i], &Hf[n + i]); else sv[n + i] = bL[i];
            pv[n + i] = HR[i];
        }
        sv[2 * n] = cR; pv[2 * n] = *Q;
        ge_msm_vartime(&Rp, 2 * n + 1, sv, pv);
        uint8_t *Lb = LR_out + 64 * round, *Rb = Lb + 32;
        ristretto_compress(Lb, &Lp); ristretto_compress(Rb, &Rp);
        merlin_append_message(t, "L", Lb, 32); merlin_append_message(t, "R", Rb, 32);
        sc u, ui; challenge_scalar(t, "u", &u); sc_invert(&ui, &u);
        for (size_t i = 0; i < n; i++) {
            sc t0, t1;
            sc_mul(&t0, &aL[i], &u); sc_mul(&t1, &ui, &aR[i]); sc_add(&aL[i], &t0, &t1);
            sc_mul(&t0, &bL[i], &ui); sc_mul(&t1, &u, &bR[i]); sc_add(&bL[i], &t0, &t1);
            sc s2[2]; ge_p3 p2[2];
            s2[0] = ui; s2[1] = u; p2[0] = GL[i]; p2[1] = GR[i];
            ge_msm_straus(&GL[i], 2, s2, p2);
            if (first) { sc_mul(&s2[0], &u, &Hf[i]); sc_mul(&s2[1], &ui, &Hf[n + i]); }
            else { s2[0] = u; s2[1] = ui; }
            p2[0] = HL[i]; p2[1] = HR[i];
            ge_msm_straus(&HL[i], 2, s2, p2);
        }
        first = 0; round++;
    }
    *a_out = a[0]; *b_out = b[0];
    free(sv); free(pv);
}

/* where prove_core leaves the per-party messages of the MPC protocol (messages.rs:23-56), for oracle_prove_shares */
typedef struct { uint8_t *bit_commitments, *poly_commitments, *shares, *challenges; } share_sink;
static share_sink *g_share_sink = NULL;   /* (test infrastructure: single-threaded use only) */
static int prove_core(const oracle_gens *g, const uint64_t *values, const uint8_t *blindings, size_t m, size_t n,
                      merlin_transcript *tp, const uint8_t *seed, size_t seed_len, uint8_t *proof_out, uint8_t *commitments_out);
int oracle_prove(const oracle_gens *g, const uint64_t *values, const uint8_t *blindings, size_t m, size_t n,
                 const uint8_t *label, size_t label_len, const uint8_t *seed, size_t seed_len,
                 uint8_t *proof_out, uint8_t *commitments_out) {
    merlin_transcript t; merlin_init(&t, label, label_len);
    return prove_core(g, values, blindings, m, n, &t, seed, seed_len, proof_out, commitments_out);
}
/* prove_multiple_with_rng on the caller's transcript (208-byte state, read and written back) */
int oracle_prove_ts(const oracle_gens *g, const uint64_t *values, const uint8_t *blindings, size_t m, size_t n,
                    uint8_t state[208], const uint8_t *seed, size_t seed_len, uint8_t *proof_out, uint8_t *commitments_out) {
    merlin_transcript t; ts_load(&t, state);
    int rc = prove_core(g, values, blindings, m, n, &t, seed, seed_len, proof_out, commitments_out);
    ts_store(state, &t);
    return rc;
}
static int prove_core(const oracle_gens *g, const uint64_t *values, const uint8_t *blindings, size_t m, size_t n,
                      merlin_transcript *tp, const uint8_t *seed, size_t seed_len, uint8_t *proof_out, uint8_t *commitments_out) {
    if (!(n == 8 || n == 16 || n == 32 || n == 64)) return ORACLE_ERR_INVALID_BITSIZE;
    if (m == 0 || (m & (m - 1))) return 5; /* InvalidAggregation */
    if (g->gens_capacity < n || g->party_capacity < m) return ORACLE_ERR_INVALID_GENERATORS_LENGTH;
    size_t nm = n * m;
    keccak_sponge rng; shake256_init(&rng); sponge_absorb(&rng, seed, seed_len);
#define t (*tp)
    merlin_append_message(&t, "dom-sep", (const uint8_t *)"rangeproof v1", 13);
    merlin_append_u64(&t, "n", n); merlin_append_u64(&t, "m", m);

    sc *vbl = malloc(m * sizeof(sc)), *a_bl = malloc(m * sizeof(sc)), *s_bl = malloc(m * sizeof(sc));
    sc *s_L = malloc(nm * sizeof(sc)), *s_R = malloc(nm * sizeof(sc));
    sc *sv = malloc((2 * n + 1) * sizeof(sc)); ge_p3 *pv = malloc((2 * n + 1) * sizeof(ge_p3));
    ge_p3 A, S; ge_identity(&A); ge_identity(&S);
    for (size_t j = 0; j < m; j++) {
        sc_from_bytes_mod_order(&vbl[j], blindings + 32 * j);
        sc v; sc_from_u64(&v, values[j]);
        ge_p3 V; pedersen_commit(&V, g, &v, &vbl[j]);
        ristretto_compress(commitments_out + 32 * j, &V);
        const ge_p3 *Gj = g->G + j * g->gens_capacity, *Hj = g->H + j * g->gens_capacity;
        rng_scalar(&rng, &a_bl[j]);
        ge_p3 Aj; ge_scalarmult(&Aj, &a_bl[j], &g->B_blinding);
        for (size_t i = 0; i < n; i++) {
            if ((values[j] >> i) & 1) ge_add(&Aj, &Aj, &Gj[i]); else ge_sub(&Aj, &Aj, &Hj[i]);
        }
        rng_scalar(&rng, &s_bl[j]);
        for (size_t i = 0; i < n; i++) rng_scalar(&rng, &s_L[j * n + i]);
        for (size_t i = 0; i < n; i++) rng_scalar(&rng, &s_R[j * n + i]);
        sv[0] = s_bl[j]; pv[0] = g->B_blinding;
        for (size_t i = 0; i < n; i++) { sv[1 + i] = s_L[j * n + i]; pv[1 + i] = Gj[i]; sv[1 + n + i] = s_R[j * n + i]; pv[1 + n + i] = Hj[i]; }
        ge_p3 Sj; ge_msm_vartime(&Sj, 2 * n + 1, sv, pv);
        if (g_share_sink) {   /* BitCommitment { V_j, A_j, S_j } (messages.rs:23-28) */
            uint8_t *bc = g_share_sink->bit_commitments + 96 * j;
            memcpy(bc, commitments_out + 32 * j, 32); ristretto_compress(bc + 32, &Aj); ristretto_compress(bc + 64, &Sj);
        }
        ge_add(&A, &A, &Aj); ge_add(&S, &S, &Sj);
    }
    for (size_t j = 0; j < m; j++) merlin_append_message(&t, "V", commitments_out + 32 * j, 32);
    uint8_t *po = proof_out;
    ristretto_compress(po, &A); ristretto_compress(po + 32, &S);
    merlin_append_message(&t, "A", po, 32); merlin_append_message(&t, "S", po + 32, 32);
    sc y, z; challenge_scalar(&t, "y", &y); challenge_scalar(&t, "z", &z);

    sc *l0 = malloc(nm * sizeof(sc)), *l1 = malloc(nm * sizeof(sc)), *r0 = malloc(nm * sizeof(sc)), *r1 = malloc(nm * sizeof(sc));
    sc *t0v = malloc(m * sizeof(sc)), *t1v = malloc(m * sizeof(sc)), *t2v = malloc(m * sizeof(sc));
    sc *t1b = malloc(m * sizeof(sc)), *t2b = malloc(m * sizeof(sc)), *ozz = malloc(m * sizeof(sc));
    ge_p3 T1, T2; ge_identity(&T1); ge_identity(&T2);
    sc zz; sc_mul(&zz, &z, &z);
    sc exp_y, offset_z, one; sc_from_u64(&exp_y, 1); sc_from_u64(&offset_z, 1); sc_from_u64(&one, 1);
    for (size_t j = 0; j < m; j++) {
        sc_mul(&ozz[j], &zz, &offset_z);
        sc exp_2; sc_from_u64(&exp_2, 1);
        for (size_t i = 0; i < n; i++) {
            size_t q = j * n + i;
            sc aL, aR, tt, tu;
            sc_from_u64(&aL, (values[j] >> i) & 1); sc_sub(&aR, &aL, &one);
            sc_sub(&l0[q], &aL, &z); l1[q] = s_L[q];
            sc_add(&tt, &aR, &z); sc_mul(&tt, &exp_y, &tt); sc_mul(&tu, &ozz[j], &exp_2); sc_add(&r0[q], &tt, &tu);
            sc_mul(&r1[q], &exp_y, &s_R[q]);
            sc_mul(&exp_y, &exp_y, &y); sc_add(&exp_2, &exp_2, &exp_2);
        }
        sc_mul(&offset_z, &offset_z, &z);
        /* t_poly = <l, r> */
        sc *ls = malloc(n * sizeof(sc)), *rs = malloc(n * sizeof(sc)), tt;
        inner_product(&t0v[j], l0 + j * n, r0 + j * n, n);
        inner_product(&t2v[j], l1 + j * n, r1 + j * n, n);
        for (size_t i = 0; i < n; i++) { sc_add(&ls[i], &l0[j * n + i], &l1[j * n + i]); sc_add(&rs[i], &r0[j * n + i], &r1[j * n + i]); }
        inner_product(&tt, ls, rs, n); sc_sub(&tt, &tt, &t0v[j]); sc_sub(&t1v[j], &tt, &t2v[j]);
        free(ls); free(rs);
        rng_scalar(&rng, &t1b[j]); rng_scalar(&rng, &t2b[j]);
        ge_p3 c1, c2; pedersen_commit(&c1, g, &t1v[j], &t1b[j]); pedersen_commit(&c2, g, &t2v[j], &t2b[j]);
        if (g_share_sink) {   /* PolyCommitment { T_1_j, T_2_j } (messages.rs:38-42) */
            ristretto_compress(g_share_sink->poly_commitments + 64 * j, &c1); ristretto_compress(g_share_sink->poly_commitments + 64 * j + 32, &c2);
        }
        ge_add(&T1, &T1, &c1); ge_add(&T2, &T2, &c2);
    }
    ristretto_compress(po + 64, &T1); ristretto_compress(po + 96, &T2);
    merlin_append_message(&t, "T_1", po + 64, 32); merlin_append_message(&t, "T_2", po + 96, 32);
    sc x; challenge_scalar(&t, "x", &x);
    sc t_x, t_x_bl, e_bl; sc_0(&t_x); sc_0(&t_x_bl); sc_0(&e_bl);
    sc *lv = malloc(nm * sizeof(sc)), *rv = malloc(nm * sizeof(sc));
    for (size_t j = 0; j < m; j++) {
        sc tt, tu;
        sc_mul(&tt, &x, &t2v[j]); sc_add(&tt, &tt, &t1v[j]); sc_mul(&tt, &x, &tt); sc_add(&tt, &tt, &t0v[j]); sc_add(&t_x, &t_x, &tt);
        sc_mul(&tt, &x, &t2b[j]); sc_add(&tt, &tt, &t1b[j]); sc_mul(&tt, &x, &tt); sc_mul(&tu, &ozz[j], &vbl[j]); sc_add(&tt, &tt, &tu); sc_add(&t_x_bl, &t_x_bl, &tt);
        sc_mul(&tt, &s_bl[j], &x); sc_add(&tt, &tt, &a_bl[j]); sc_add(&e_bl, &e_bl, &tt);
        sc txj, txbj, ebj;
        sc_mul(&txj, &x, &t2v[j]); sc_add(&txj, &txj, &t1v[j]); sc_mul(&txj, &x, &txj); sc_add(&txj, &txj, &t0v[j]);
        sc_mul(&txbj, &x, &t2b[j]); sc_add(&txbj, &txbj, &t1b[j]); sc_mul(&txbj, &x, &txbj); sc_mul(&tu, &ozz[j], &vbl[j]); sc_add(&txbj, &txbj, &tu);
        sc_mul(&ebj, &s_bl[j], &x); sc_add(&ebj, &ebj, &a_bl[j]);
        for (size_t i = 0; i < n; i++) {
            size_t q = j * n + i;
            sc_mul(&tt, &l1[q], &x); sc_add(&lv[q], &l0[q], &tt);
            sc_mul(&tt, &r1[q], &x); sc_add(&rv[q], &r0[q], &tt);
        }
        if (g_share_sink) {   /* ProofShare { t_x, t_x_blinding, e_blinding, l_vec, r_vec } (messages.rs:50-56; party.rs:268-311) */
            uint8_t *sh = g_share_sink->shares + 32 * (3 + 2 * n) * j;
            sc_tobytes(sh, &txj); sc_tobytes(sh + 32, &txbj); sc_tobytes(sh + 64, &ebj);
            for (size_t i = 0; i < n; i++) { sc_tobytes(sh + 96 + 32 * i, &lv[j * n + i]); sc_tobytes(sh + 96 + 32 * (n + i), &rv[j * n + i]); }
        }
    }
    if (g_share_sink) { sc_tobytes(g_share_sink->challenges, &y); sc_tobytes(g_share_sink->challenges + 32, &z); sc_tobytes(g_share_sink->challenges + 64, &x); }
    sc_tobytes(po + 128, &t_x); sc_tobytes(po + 160, &t_x_bl); sc_tobytes(po + 192, &e_bl);
    append_scalar(&t, "t_x", &t_x); append_scalar(&t, "t_x_blinding", &t_x_bl); append_scalar(&t, "e_blinding", &e_bl);
    sc w; challenge_scalar(&t, "w", &w);
    ge_p3 Q; ge_scalarmult(&Q, &w, &g->B);
    sc *Hf = malloc(nm * sizeof(sc)); sc y_inv; sc_invert(&y_inv, &y);
    sc_from_u64(&Hf[0], 1); for (size_t i = 1; i < nm; i++) sc_mul(&Hf[i], &Hf[i - 1], &y_inv);
    ge_p3 *Gv = malloc(nm * sizeof(ge_p3)), *Hv = malloc(nm * sizeof(ge_p3));
    for (size_t j = 0; j < m; j++) for (size_t i = 0; i < n; i++) {
        Gv[j * n + i] = g->G[j * g->gens_capacity + i]; Hv[j * n + i] = g->H[j * g->gens_capacity + i];
    }
    size_t lg_n = 0; while (((size_t)1 << lg_n) < nm) lg_n++;
    sc a_fin, b_fin;
    ipp_create(&t, &Q, Hf, Gv, Hv, lv, rv, nm, po + 224, &a_fin, &b_fin);
    sc_tobytes(po + 224 + 64 * lg_n, &a_fin); sc_tobytes(po + 224 + 64 * lg_n + 32, &b_fin);
    free(vbl); free(a_bl); free(s_bl); free(s_L); free(s_R); free(sv); free(pv);
    free(l0); free(l1); free(r0); free(r1); free(t0v); free(t1v); free(t2v); free(t1b); free(t2b); free(ozz);
    free(lv); free(rv); free(Hf); free(Gv); free(Hv);
    return 0;
#undef t
}

/* ---------------- the MPC messages and the dealer's share audit (messages.rs) ---------------- */
/* prove_multiple as its parties and dealer run it (mod.rs:234-288, party.rs, dealer.rs), also returning what they exchange:
 * bit_commitments m x 96 (V_j, A_j, S_j), poly_commitments m x 64 (T_1_j, T_2_j), shares m x 32(3 + 2n)
 * (t_x, t_x_blinding, e_blinding, l_vec, r_vec), challenges (y, z, x). */
int oracle_prove_shares(const oracle_gens *g, const uint64_t *values, const uint8_t *blindings, size_t m, size_t n,
                        const uint8_t *label, size_t label_len, const uint8_t *seed, size_t seed_len,
                        uint8_t *proof_out, uint8_t *commitments_out, uint8_t *bit_commitments, uint8_t *poly_commitments,
                        uint8_t *shares, uint8_t challenges[96]) {
    ge_init();
    share_sink sk = {bit_commitments, poly_commitments, shares, challenges};
    merlin_transcript t; merlin_init(&t, label, label_len);
    g_share_sink = &sk;
    int rc = prove_core(g, values, blindings, m, n, &t, seed, seed_len, proof_out, commitments_out);
    g_share_sink = NULL;
    return rc;
}

/* ProofShare::audit_share (messages.rs:85-167) for party j: 0 = Ok(()), 1 = Err(()).
 * out2 (optional, 64 bytes): compress(P_check), compress(t_check) -- all-zero when the share is sound. */
int oracle_audit_share(const oracle_gens *g, size_t n, size_t j, const uint8_t *share, const uint8_t bit_commitment[96],
                       const uint8_t poly_commitment[64], const uint8_t challenges[96], uint8_t *out2) {
    ge_init();
    if (out2) memset(out2, 0xff, 64);
    if (n > g->gens_capacity || j >= g->party_capacity) return 1;                           /* check_size (:57-82) */
    sc y, z, x, t_x, t_x_bl, e_bl, *l = malloc(n * sizeof(sc)), *r = malloc(n * sizeof(sc));
    int bad = sc_from_canonical_bytes(&y, challenges) | sc_from_canonical_bytes(&z, challenges + 32) | sc_from_canonical_bytes(&x, challenges + 64)
            | sc_from_canonical_bytes(&t_x, share) | sc_from_canonical_bytes(&t_x_bl, share + 32) | sc_from_canonical_bytes(&e_bl, share + 64);
    for (size_t i = 0; i < n; i++) bad |= sc_from_canonical_bytes(&l[i], share + 96 + 32 * i) | sc_from_canonical_bytes(&r[i], share + 96 + 32 * (n + i));
    int rc = 1;
    if (!bad) do {
        sc zz, minus_z, z_j, y_jn, y_jn_inv, y_inv, one, two;
        sc_from_u64(&one, 1); sc_from_u64(&two, 2);
        sc_mul(&zz, &z, &z); sc_neg(&minus_z, &z);
        z_j = one; for (size_t i = 0; i < j; i++) sc_mul(&z_j, &z_j, &z);                     /* z^j */
        y_jn = one; for (size_t i = 0; i < j * n; i++) sc_mul(&y_jn, &y_jn, &y);             /* y^(j n) */
        sc_invert(&y_jn_inv, &y_jn); sc_invert(&y_inv, &y);
        sc ip; inner_product(&ip, l, r, n);
        if (!sc_eq(&t_x, &ip)) break;                                                          /* :112-114 */
        size_t N = 2 * n + 3;
        sc *sv = malloc(N * sizeof(sc)); ge_p3 *pv = malloc(N * sizeof(ge_p3));
        int dec = 0;
        sv[0] = one; dec |= ristretto_decompress(&pv[0], bit_commitment + 32);                 /* A_j */
        sv[1] = x; dec |= ristretto_decompress(&pv[1], bit_commitment + 64);                   /* S_j */
        sc_neg(&sv[2], &e_bl); pv[2] = g->B_blinding;
        sc exp_2 = one, exp_y_inv = one;
        for (size_t i = 0; i < n; i++) {
            sc_sub(&sv[3 + i], &minus_z, &l[i]); pv[3 + i] = g->G[j * g->gens_capacity + i];   /* g_i = -z - l_i */
            sc f, t0, t1, nr;                                                                   /* h_i (:117-126) */
            sc_mul(&f, &exp_y_inv, &y_jn_inv);
            sc_neg(&nr, &r[i]); sc_mul(&t0, &f, &nr);
            sc_mul(&t1, &zz, &z_j); sc_mul(&t1, &t1, &exp_2); sc_mul(&t1, &f, &t1);
            sc_add(&sv[3 + n + i], &z, &t0); sc_add(&sv[3 + n + i], &sv[3 + n + i], &t1);
            pv[3 + n + i] = g->H[j * g->gens_capacity + i];
            sc_mul(&exp_2, &exp_2, &two); sc_mul(&exp_y_inv, &exp_y_inv, &y_inv);
        }
        /* (A_j / S_j are RistrettoPoints in BitCommitment upstream: an undecodable encoding cannot occur there; here it fails the audit) */
        ge_p3 P_check, t_check;
        int ok = !dec;
        if (ok) { ge_msm_vartime(&P_check, N, sv, pv); if (out2) ristretto_compress(out2, &P_check); ok = ge_is_identity(&P_check); }
        free(sv); free(pv);
        if (!ok) break;
        ge_p3 V_j, T1, T2;
        if (ristretto_decompress(&V_j, bit_commitment)) break;                                 /* :143 */
        if (ristretto_decompress(&T1, poly_commitment) || ristretto_decompress(&T2, poly_commitment + 32)) break;
        sc sy, s2, delta, t0, t1, s5[5]; ge_p3 p5[5];
        sum_of_powers(&sy, &y, n); sum_of_powers(&s2, &two, n);
        sc_sub(&t0, &z, &zz); sc_mul(&t0, &t0, &sy); sc_mul(&t0, &t0, &y_jn);
        sc_mul(&t1, &z, &zz); sc_mul(&t1, &t1, &s2); sc_mul(&t1, &t1, &z_j);
        sc_sub(&delta, &t0, &t1);
        sc_mul(&s5[0], &zz, &z_j); p5[0] = V_j;
        s5[1] = x; p5[1] = T1;
        sc_mul(&s5[2], &x, &x); p5[2] = T2;
        sc_sub(&s5[3], &delta, &t_x); p5[3] = g->B;
        sc_neg(&s5[4], &t_x_bl); p5[4] = g->B_blinding;
        ge_msm_vartime(&t_check, 5, s5, p5);
        if (out2) ristretto_compress(out2 + 32, &t_check);
        if (ge_is_identity(&t_check)) rc = 0;
    } while (0);
    free(l); free(r);
    return rc;
}

/* ---------------- stand-alone inner-product proof (ipp.rs:260-326, 373-407, 433-497) ---------------- */
int oracle_ipp_verify(size_t n, const uint8_t *proof, size_t proof_len, const uint8_t *label, size_t label_len,
                      const uint8_t *G_factors, const uint8_t *H_factors, const uint8_t P[32], const uint8_t Q[32],
                      const uint8_t *G, const uint8_t *H, uint8_t msm_out[32]) {
    ge_init();
    /* InnerProductProof::from_bytes */
    if (proof_len % 32 != 0) return ORACLE_ERR_FORMAT;
    size_t ne = proof_len / 32;
    if (ne < 2 || (ne - 2) % 2 != 0) return ORACLE_ERR_FORMAT;
    size_t lg_n = (ne - 2) / 2;
    if (lg_n >= 32) return ORACLE_ERR_FORMAT;
    sc a, b;
    if (sc_from_canonical_bytes(&a, proof + 64 * lg_n)) return ORACLE_ERR_FORMAT;
    if (sc_from_canonical_bytes(&b, proof + 64 * lg_n + 32)) return ORACLE_ERR_FORMAT;
    /* verification_scalars */
    if (n != ((size_t)1 << lg_n)) return ORACLE_ERR_VERIFICATION;
    merlin_transcript t; merlin_init(&t, label, label_len);
    merlin_append_message(&t, "dom-sep", (const uint8_t *)"ipp v1", 6);
    merlin_append_u64(&t, "n", n);
    sc u_sq[32], u_inv_sq[32], allinv; sc_from_u64(&allinv, 1);
    for (size_t i = 0; i < lg_n; i++) {
        if (validate_and_append_point(&t, "L", proof + 64 * i)) return ORACLE_ERR_VERIFICATION;
        if (validate_and_append_point(&t, "R", proof + 64 * i + 32)) return ORACLE_ERR_VERIFICATION;
        sc u, ui; challenge_scalar(&t, "u", &u); sc_invert(&ui, &u);
        sc_mul(&allinv, &allinv, &ui); sc_mul(&u_sq[i], &u, &u); sc_mul(&u_inv_sq[i], &ui, &ui);
    }
    sc *s = malloc((n + 1) * sizeof(sc));
    s[0] = allinv;
    for (size_t i = 1; i < n; i++) {
        size_t lg_i = 63 - (size_t)__builtin_clzll((unsigned long long)i);
        sc_mul(&s[i], &s[i - ((size_t)1 << lg_i)], &u_sq[(lg_n - 1) - lg_i]);
    }
    size_t N = 2 * n + 2 * lg_n + 2;
    sc *scal = malloc(N * sizeof(sc)); ge_p3 *pts = malloc(N * sizeof(ge_p3));
    int bad = 0; size_t o = 0;
    sc_mul(&scal[o], &a, &b); if (ristretto_decompress(&pts[o], Q)) bad = 1; o++;
    for (size_t i = 0; i < n; i++) {
        sc gf, t0; sc_from_bytes_mod_order(&gf, G_factors + 32 * i);
        sc_mul(&t0, &a, &s[i]); sc_mul(&scal[o], &t0, &gf);
        if (ristretto_decompress(&pts[o], G + 32 * i)) bad = 1; o++;
    }
    for (size_t i = 0; i < n; i++) {
        sc hf, t0; sc_from_bytes_mod_order(&hf, H_factors + 32 * i);
        sc_mul(&t0, &b, &s[n - 1 - i]); sc_mul(&scal[o], &t0, &hf);
        if (ristretto_decompress(&pts[o], H + 32 * i)) bad = 1; o++;
    }
    for (size_t i = 0; i < lg_n; i++) { sc_neg(&scal[o], &u_sq[i]); if (ristretto_decompress(&pts[o], proof + 64 * i)) bad = 1; o++; }
    for (size_t i = 0; i < lg_n; i++) { sc_neg(&scal[o], &u_inv_sq[i]); if (ristretto_decompress(&pts[o], proof + 64 * i + 32)) bad = 1; o++; }
    { sc one; sc_from_u64(&one, 1); sc_neg(&scal[o], &one); if (ristretto_decompress(&pts[o], P)) bad = 1; o++; }   /* expect_P - P */
    int rc;
    if (bad) { rc = ORACLE_ERR_VERIFICATION; if (msm_out) memset(msm_out, 0xff, 32); }
    else {
        ge_p3 r; msm_dispatch(&r, N, scal, pts, 0);
        if (msm_out) ristretto_compress(msm_out, &r);
        rc = ge_is_identity(&r) ? ORACLE_OK : ORACLE_ERR_VERIFICATION;   /* expect_P == *P  (ristretto equality) */
    }
    free(s); free(scal); free(pts);
    return rc;
}

/* InnerProductProof::from_bytes (ipp.rs:373-407) + verification_scalars (ipp.rs:198-253) on the caller's transcript: the call
 * r1cs/verifier.rs:401-404 and ipp.rs:283 make.  state: in, and advanced out when the result is Ok.  u_sq, u_inv_sq: lg_n x 32
 * bytes; s: n x 32 bytes. */
int oracle_ipp_verification_scalars(size_t n, const uint8_t *proof, size_t proof_len, uint8_t state[208], uint8_t *u_sq_out,
                                    uint8_t *u_inv_sq_out, uint8_t *s_out) {
    if (proof_len % 32 != 0) return ORACLE_ERR_FORMAT;
    size_t ne = proof_len / 32;
    if (ne < 2 || (ne - 2) % 2 != 0) return ORACLE_ERR_FORMAT;
    size_t lg_n = (ne - 2) / 2;
    if (lg_n >= 32) return ORACLE_ERR_FORMAT;
    sc a, b;
    if (sc_from_canonical_bytes(&a, proof + 64 * lg_n)) return ORACLE_ERR_FORMAT;
    if (sc_from_canonical_bytes(&b, proof + 64 * lg_n + 32)) return ORACLE_ERR_FORMAT;
    if (n != ((size_t)1 << lg_n)) return ORACLE_ERR_VERIFICATION;
    merlin_transcript t; ts_load(&t, state);
    merlin_append_message(&t, "dom-sep", (const uint8_t *)"ipp v1", 6);
    merlin_append_u64(&t, "n", n);
    sc u_sq[32], u_inv_sq[32], allinv; sc_from_u64(&allinv, 1);
    for (size_t i = 0; i < lg_n; i++) {
        /* validate_and_append_point returns Err BEFORE absorbing an identity point (transcript.rs:75-87): the caller's `&mut Transcript`
         * stays as it is at that moment -- domain separator and earlier rounds included (ipp.rs:213-222) */
        if (validate_and_append_point(&t, "L", proof + 64 * i)) { ts_store(state, &t); return ORACLE_ERR_VERIFICATION; }
        if (validate_and_append_point(&t, "R", proof + 64 * i + 32)) { ts_store(state, &t); return ORACLE_ERR_VERIFICATION; }
        sc u, ui; challenge_scalar(&t, "u", &u); sc_invert(&ui, &u);
        sc_mul(&allinv, &allinv, &ui); sc_mul(&u_sq[i], &u, &u); sc_mul(&u_inv_sq[i], &ui, &ui);
    }
    sc *s = malloc((n + 1) * sizeof(sc));
    s[0] = allinv;
    for (size_t i = 1; i < n; i++) {
        size_t lg_i = 63 - (size_t)__builtin_clzll((unsigned long long)i);
        sc_mul(&s[i], &s[i - ((size_t)1 << lg_i)], &u_sq[(lg_n - 1) - lg_i]);
    }
    for (size_t i = 0; i < lg_n; i++) { sc_tobytes(u_sq_out + 32 * i, &u_sq[i]); sc_tobytes(u_inv_sq_out + 32 * i, &u_inv_sq[i]); }
    for (size_t i = 0; i < n; i++) sc_tobytes(s_out + 32 * i, &s[i]);
    free(s);
    ts_store(state, &t);
    return ORACLE_OK;
}

/* InnerProductProof::create(...).to_bytes() (ipp.rs:38-193, 334-345) with G_factors = 1 (what the reference's callers
 * pass: range_proof/dealer.rs, ipp.rs:454) -- the oracle for the GPU's batched prover.  Returns 1 if a point does not decode. */
int oracle_ipp_create(size_t n, const uint8_t *label, size_t label_len, const uint8_t Q[32], const uint8_t *Hf,
                      const uint8_t *G, const uint8_t *H, const uint8_t *a, const uint8_t *b, uint8_t *proof_out) {
    ge_init();
    size_t lg = 0; while (((size_t)1 << lg) < n) lg++;
    ge_p3 Qp, *Gv = malloc(n * sizeof(ge_p3)), *Hv = malloc(n * sizeof(ge_p3));
    sc *av = malloc(n * sizeof(sc)), *bv = malloc(n * sizeof(sc)), *hf = malloc(n * sizeof(sc));
    int bad = ristretto_decompress(&Qp, Q) != 0;
    for (size_t i = 0; i < n; i++) {
        if (ristretto_decompress(&Gv[i], G + 32 * i)) bad = 1;
        if (ristretto_decompress(&Hv[i], H + 32 * i)) bad = 1;
        sc_from_bytes_mod_order(&av[i], a + 32 * i); sc_from_bytes_mod_order(&bv[i], b + 32 * i); sc_from_bytes_mod_order(&hf[i], Hf + 32 * i);
    }
    if (!bad) {
        merlin_transcript t; merlin_init(&t, label, label_len);
        sc af, bf;
        ipp_create(&t, &Qp, hf, Gv, Hv, av, bv, n, proof_out, &af, &bf);
        sc_tobytes(proof_out + 64 * lg, &af); sc_tobytes(proof_out + 64 * lg + 32, &bf);
    }
    free(Gv); free(Hv); free(av); free(bv); free(hf);
    return bad;
}

int oracle_ipp_test_instance(size_t n, const uint8_t *label, size_t label_len, const uint8_t *seed, size_t seed_len,
                             uint8_t *proof_out, uint8_t P_out[32], uint8_t Q_out[32], uint8_t *G_out, uint8_t *H_out,
                             uint8_t *Gf_out, uint8_t *Hf_out) {
    if (n == 0 || (n & (n - 1))) return 5;
    oracle_gens *g = oracle_gens_new(n, 1);
    keccak_sponge rng; shake256_init(&rng); sponge_absorb(&rng, seed, seed_len);
    ge_p3 Q; { uint8_t h[64]; sha3_512(h, (const uint8_t *)"test point", 10); ristretto_from_uniform_bytes(&Q, h); }
    sc *a = malloc(n * sizeof(sc)), *b = malloc(n * sizeof(sc)), *Hf = malloc(n * sizeof(sc));
    sc *sv = malloc((2 * n + 1) * sizeof(sc)); ge_p3 *pv = malloc((2 * n + 1) * sizeof(ge_p3));
    for (size_t i = 0; i < n; i++) rng_scalar(&rng, &a[i]);
    for (size_t i = 0; i < n; i++) rng_scalar(&rng, &b[i]);
    sc c; inner_product(&c, a, b, n);
    sc y_inv, one; rng_scalar(&rng, &y_inv); sc_from_u64(&one, 1);
    Hf[0] = one; for (size_t i = 1; i < n; i++) sc_mul(&Hf[i], &Hf[i - 1], &y_inv);
    for (size_t i = 0; i < n; i++) { sv[i] = a[i]; pv[i] = g->G[i]; sc_mul(&sv[n + i], &b[i], &Hf[i]); pv[n + i] = g->H[i]; }
    sv[2 * n] = c; pv[2 * n] = Q;
    ge_p3 P; ge_msm_vartime(&P, 2 * n + 1, sv, pv);
    ristretto_compress(P_out, &P); ristretto_compress(Q_out, &Q);
    memcpy(G_out, g->Gc, 32 * n); memcpy(H_out, g->Hc, 32 * n);
    for (size_t i = 0; i < n; i++) { sc_tobytes(Gf_out + 32 * i, &one); sc_tobytes(Hf_out + 32 * i, &Hf[i]); }
    ge_p3 *Gv = malloc(n * sizeof(ge_p3)), *Hv = malloc(n * sizeof(ge_p3));
    memcpy(Gv, g->G, n * sizeof(ge_p3)); memcpy(Hv, g->H, n * sizeof(ge_p3));
    size_t lg_n = 0; while (((size_t)1 << lg_n) < n) lg_n++;
    merlin_transcript t; merlin_init(&t, label, label_len);
    sc a_fin, b_fin;
    ipp_create(&t, &Q, Hf, Gv, Hv, a, b, n, proof_out, &a_fin, &b_fin);
    sc_tobytes(proof_out + 64 * lg_n, &a_fin); sc_tobytes(proof_out + 64 * lg_n + 32, &b_fin);
    free(a); free(b); free(Hf); free(sv); free(pv); free(Gv); free(Hv); oracle_gens_free(g);
    return 0;
}

/* ---------------- LinearProof (src/linear_proof.rs) ---------------- */
static void linear_public_inputs(merlin_transcript *t, size_t n, const uint8_t C[32], const uint8_t *b, const uint8_t *G,
                                 const uint8_t F[32], const uint8_t B[32]) {
    /* linear_proof.rs:73-83 (create) = :196-206 (verify) */
    merlin_append_message(t, "dom-sep", (const uint8_t *)"ipp v1", 6);
    merlin_append_u64(t, "n", n);
    merlin_append_message(t, "C", C, 32);
    for (size_t i = 0; i < n; i++) merlin_append_message(t, "b_i", b + 32 * i, 32);
    for (size_t i = 0; i < n; i++) merlin_append_message(t, "G_i", G + 32 * i, 32);
    merlin_append_message(t, "F", F, 32);
    merlin_append_message(t, "B", B, 32);
}

/* LinearProof::create(transcript, rng, C, r, a_vec, b_vec, G_vec, F, B).to_bytes() (linear_proof.rs:40-173, 322-331).
 * rng: the bytes the rng yields, 64 per Scalar::random, in draw order (s_j, t_j per round, then s_star, t_star):
 * 64 * (2 lg n + 2) bytes.  b must be canonical (it is a Vec<Scalar> upstream).  Returns 0, 5 for InvalidInputLength
 * (n not a power of two), 1 when a point does not decode. */
int oracle_linear_create(size_t n, const uint8_t state[208], const uint8_t *rng, const uint8_t C[32], const uint8_t r_in[32],
                         const uint8_t *a_in, const uint8_t *b_in, const uint8_t *G_in, const uint8_t F_in[32],
                         const uint8_t B_in[32], uint8_t *proof_out) {
    ge_init();
    if (n == 0 || (n & (n - 1))) return 5;
    merlin_transcript t; ts_load(&t, state);
    ge_p3 F, B, *G = malloc(n * sizeof(ge_p3)), *pv = malloc((n + 2) * sizeof(ge_p3));
    sc *a = malloc(n * sizeof(sc)), *b = malloc(n * sizeof(sc)), *sv = malloc((n + 2) * sizeof(sc)), r;
    int bad = ristretto_decompress(&F, F_in) != 0 || ristretto_decompress(&B, B_in) != 0;
    for (size_t i = 0; i < n; i++) {
        if (ristretto_decompress(&G[i], G_in + 32 * i)) bad = 1;
        sc_from_bytes_mod_order(&a[i], a_in + 32 * i); sc_from_bytes_mod_order(&b[i], b_in + 32 * i);
    }
    sc_from_bytes_mod_order(&r, r_in);
    if (!bad) {
        linear_public_inputs(&t, n, C, b_in, G_in, F_in, B_in);
        size_t round = 0, draw = 0, m = n;
        while (m != 1) {
            m /= 2;
            sc *aL = a, *aR = a + m, *bL = b, *bR = b + m;
            ge_p3 *GL = G, *GR = G + m;
            sc cL, cR, sj, tj;
            inner_product(&cL, aL, bR, m); inner_product(&cR, aR, bL, m);
            sc_from_wide(&sj, rng + 64 * draw++); sc_from_wide(&tj, rng + 64 * draw++);
            ge_p3 Lp, Rp;
            for (size_t i = 0; i < m; i++) { sv[i] = aL[i]; pv[i] = GR[i]; }
            sv[m] = sj; pv[m] = B; sv[m + 1] = cL; pv[m + 1] = F;
            ge_msm_vartime(&Lp, m + 2, sv, pv);                      /* L = a_L * G_R + s_j * B + c_L * F */
            for (size_t i = 0; i < m; i++) { sv[i] = aR[i]; pv[i] = GL[i]; }
            sv[m] = tj; pv[m] = B; sv[m + 1] = cR; pv[m + 1] = F;
            ge_msm_vartime(&Rp, m + 2, sv, pv);                      /* R = a_R * G_L + t_j * B + c_R * F */
            uint8_t *Lb = proof_out + 64 * round, *Rb = Lb + 32;
            ristretto_compress(Lb, &Lp); ristretto_compress(Rb, &Rp);
            merlin_append_message(&t, "L", Lb, 32); merlin_append_message(&t, "R", Rb, 32);
            sc x, xi; challenge_scalar(&t, "x_j", &x); sc_invert(&xi, &x);
            for (size_t i = 0; i < m; i++) {
                sc t0;
                sc_mul(&t0, &xi, &aR[i]); sc_add(&aL[i], &aL[i], &t0);     /* a_L + x^-1 a_R */
                sc_mul(&t0, &x, &bR[i]); sc_add(&bL[i], &bL[i], &t0);      /* b_L + x b_R */
                sc s2[2]; ge_p3 p2[2];
                sc_from_u64(&s2[0], 1); s2[1] = x; p2[0] = GL[i]; p2[1] = GR[i];
                ge_msm_straus(&GL[i], 2, s2, p2);                          /* G_L + x G_R */
            }
            sc t0, t1; sc_mul(&t0, &x, &sj); sc_mul(&t1, &xi, &tj); sc_add(&r, &r, &t0); sc_add(&r, &r, &t1);
            round++;
        }
        sc s_star, t_star; sc_from_wide(&s_star, rng + 64 * draw++); sc_from_wide(&t_star, rng + 64 * draw++);
        sc s3[3]; ge_p3 p3[3], Sp;
        s3[0] = t_star; p3[0] = B; sc_mul(&s3[1], &s_star, &b[0]); p3[1] = F; s3[2] = s_star; p3[2] = G[0];
        ge_msm_straus(&Sp, 3, s3, p3);                                     /* S = t* B + s* b_0 F + s* G_0 */
        uint8_t *Sb = proof_out + 64 * round;
        ristretto_compress(Sb, &Sp);
        merlin_append_message(&t, "S", Sb, 32);
        sc xs, a_star, r_star, t0;
        challenge_scalar(&t, "x_star", &xs);
        sc_mul(&t0, &xs, &a[0]); sc_add(&a_star, &s_star, &t0);
        sc_mul(&t0, &xs, &r); sc_add(&r_star, &t_star, &t0);
        sc_tobytes(Sb + 32, &a_star); sc_tobytes(Sb + 64, &r_star);
    }
    free(G); free(pv); free(a); free(b); free(sv);
    return bad;
}

/* LinearProof::from_bytes(proof)?.verify(transcript, C, G, F, B, b_vec) (linear_proof.rs:175-236, 240-312, 350-394).
 * msm_out (optional) = compress(expect_S - S).  Returns ORACLE_OK / ORACLE_ERR_*. */
int oracle_linear_verify(size_t n, const uint8_t *proof, size_t proof_len, const uint8_t state[208], const uint8_t C[32],
                         const uint8_t *G, const uint8_t F[32], const uint8_t B[32], const uint8_t *b_in, uint8_t msm_out[32]) {
    ge_init();
    if (proof_len % 32 != 0) return ORACLE_ERR_FORMAT;
    size_t ne = proof_len / 32;
    if (ne < 3 || (ne - 3) % 2 != 0) return ORACLE_ERR_FORMAT;
    size_t lg_n = (ne - 3) / 2;
    if (lg_n >= 32) return ORACLE_ERR_FORMAT;
    const uint8_t *Sb = proof + 64 * lg_n;
    sc a, r;
    if (sc_from_canonical_bytes(&a, Sb + 32)) return ORACLE_ERR_FORMAT;
    if (sc_from_canonical_bytes(&r, Sb + 64)) return ORACLE_ERR_FORMAT;
    merlin_transcript t; ts_load(&t, state);
    linear_public_inputs(&t, n, C, b_in, G, F, B);
    /* verification_scalars (:240-290) */
    if (n != ((size_t)1 << lg_n)) return ORACLE_ERR_VERIFICATION;
    sc *b = malloc((n + 1) * sizeof(sc)), x[32], xinv[32];
    for (size_t i = 0; i < n; i++) sc_from_bytes_mod_order(&b[i], b_in + 32 * i);
    size_t m = n;
    for (size_t j = 0; j < lg_n; j++) {
        if (validate_and_append_point(&t, "L", proof + 64 * j) || validate_and_append_point(&t, "R", proof + 64 * j + 32)) { free(b); return ORACLE_ERR_VERIFICATION; }
        challenge_scalar(&t, "x_j", &x[j]);
        m /= 2;
        for (size_t i = 0; i < m; i++) { sc t0; sc_mul(&t0, &x[j], &b[m + i]); sc_add(&b[i], &b[i], &t0); }
        sc_invert(&xinv[j], &x[j]);
    }
    sc b0 = b[0];
    merlin_append_message(&t, "S", Sb, 32);
    sc xs; challenge_scalar(&t, "x_star", &xs);
    /* subset products (:299-314) */
    sc *s = b;   /* reuse */
    sc_from_u64(&s[0], 1);
    for (size_t i = 1; i < n; i++) {
        size_t lg_i = 63 - (size_t)__builtin_clzll((unsigned long long)i);
        sc_mul(&s[i], &s[i - ((size_t)1 << lg_i)], &x[(lg_n - 1) - lg_i]);
    }
    /* expect_S - S = r B + a b_0 F - x* C - x* sum (x_j L_j + x_j^-1 R_j) + a sum s_i G_i - S  (:214-231) */
    size_t N = n + 2 * lg_n + 4, o = 0;
    sc *scal = malloc(N * sizeof(sc)), nxs, one; ge_p3 *pts = malloc(N * sizeof(ge_p3));
    int bad = 0;
    sc_neg(&nxs, &xs); sc_from_u64(&one, 1);
    scal[o] = r; if (ristretto_decompress(&pts[o], B)) bad = 1; o++;
    sc_mul(&scal[o], &a, &b0); if (ristretto_decompress(&pts[o], F)) bad = 1; o++;
    scal[o] = nxs; if (ristretto_decompress(&pts[o], C)) bad = 1; o++;
    for (size_t j = 0; j < lg_n; j++) { sc_mul(&scal[o], &nxs, &x[j]); if (ristretto_decompress(&pts[o], proof + 64 * j)) bad = 1; o++; }
    for (size_t j = 0; j < lg_n; j++) { sc_mul(&scal[o], &nxs, &xinv[j]); if (ristretto_decompress(&pts[o], proof + 64 * j + 32)) bad = 1; o++; }
    for (size_t i = 0; i < n; i++) { sc_mul(&scal[o], &a, &s[i]); if (ristretto_decompress(&pts[o], G + 32 * i)) bad = 1; o++; }
    sc_neg(&scal[o], &one); if (ristretto_decompress(&pts[o], Sb)) bad = 1; o++;
    int rc;
    if (bad) { rc = ORACLE_ERR_VERIFICATION; if (msm_out) memset(msm_out, 0xff, 32); }
    else {
        ge_p3 res; msm_dispatch(&res, N, scal, pts, 0);
        if (msm_out) ristretto_compress(msm_out, &res);
        rc = ge_is_identity(&res) ? ORACLE_OK : ORACLE_ERR_VERIFICATION;   /* expect_S == S (ristretto equality) */
    }
    free(b); free(scal); free(pts);
    return rc;
}

/* ---------------- threaded batch drivers ---------------- */
typedef struct {
    int kind, tid, threads;
    const oracle_gens *g; size_t nbatch, m, n, proof_len, label_len, seed_len, nterms;
    const uint8_t *proofs, *commitments, *label, *rng64s, *seed, *blindings, *scalars, *points;
    const uint64_t *values;
    uint8_t *verdicts, *msm_outs, *proofs_out, *commitments_out, *status;
    int algo;
} job_t;

static void *worker(void *arg) {
    job_t *j = arg;
    for (size_t i = j->tid; i < j->nbatch; i += j->threads) {
        if (j->kind == 0) {
            uint8_t tmp[32];
            int rc = oracle_verify(j->g, j->proofs + i * j->proof_len, j->proof_len, j->commitments + i * j->m * 32, j->m, j->n,
                                   j->label, j->label_len, j->rng64s + 64 * i, j->msm_outs ? j->msm_outs + 32 * i : tmp);
            j->verdicts[i] = (uint8_t)rc;
        } else if (j->kind == 1) {
            uint8_t sd[256]; size_t sl = j->seed_len > 250 ? 250 : j->seed_len;
            memcpy(sd, j->seed, sl); sd[sl] = (uint8_t)i; sd[sl + 1] = (uint8_t)(i >> 8); sd[sl + 2] = (uint8_t)(i >> 16); sd[sl + 3] = (uint8_t)(i >> 24);
            oracle_prove(j->g, j->values + i * j->m, j->blindings + i * j->m * 32, j->m, j->n, j->label, j->label_len, sd, sl + 4,
                         j->proofs_out + i * j->proof_len, j->commitments_out + i * j->m * 32);
        } else {
            j->status[i] = (uint8_t)oracle_msm(j->nterms, j->scalars + i * j->nterms * 32, j->points + i * j->nterms * 32, j->algo, j->msm_outs + 32 * i);
        }
    }
    return NULL;
}
static double run_jobs(job_t *proto, int threads) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256]; job_t jobs[256];
    double t0 = now_s();
    for (int t = 0; t < threads; t++) { jobs[t] = *proto; jobs[t].tid = t; jobs[t].threads = threads; }
    if (threads == 1) worker(&jobs[0]);
    else {
        for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, worker, &jobs[t]);
        for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    }
    return now_s() - t0;
}
double oracle_verify_batch(const oracle_gens *g, size_t nbatch, const uint8_t *proofs, size_t proof_len,
                           const uint8_t *commitments, size_t m, size_t n, const uint8_t *label, size_t label_len,
                           const uint8_t *rng64s, uint8_t *verdicts, uint8_t *msm_outs, int threads) {
    job_t j; memset(&j, 0, sizeof j);
    j.kind = 0; j.g = g; j.nbatch = nbatch; j.proofs = proofs; j.proof_len = proof_len; j.commitments = commitments;
    j.m = m; j.n = n; j.label = label; j.label_len = label_len; j.rng64s = rng64s; j.verdicts = verdicts; j.msm_outs = msm_outs;
    return run_jobs(&j, threads);
}
double oracle_prove_batch(const oracle_gens *g, size_t nbatch, const uint64_t *values, const uint8_t *blindings,
                          size_t m, size_t n, const uint8_t *label, size_t label_len, const uint8_t *seed, size_t seed_len,
                          uint8_t *proofs_out, uint8_t *commitments_out, int threads) {
    job_t j; memset(&j, 0, sizeof j);
    size_t lg = 0; while (((size_t)1 << lg) < n * m) lg++;
    j.kind = 1; j.g = g; j.nbatch = nbatch; j.values = values; j.blindings = blindings; j.m = m; j.n = n;
    j.label = label; j.label_len = label_len; j.seed = seed; j.seed_len = seed_len;
    j.proofs_out = proofs_out; j.commitments_out = commitments_out; j.proof_len = 32 * (9 + 2 * lg);
    return run_jobs(&j, threads);
}
double oracle_msm_batch(size_t nbatch, size_t n, const uint8_t *scalars, const uint8_t *points, int algo,
                        uint8_t *outs, uint8_t *status, int threads) {
    job_t j; memset(&j, 0, sizeof j);
    j.kind = 2; j.nbatch = nbatch; j.nterms = n; j.scalars = scalars; j.points = points; j.algo = algo;
    j.msm_outs = outs; j.status = status;
    ge_init();
    return run_jobs(&j, threads);
}
