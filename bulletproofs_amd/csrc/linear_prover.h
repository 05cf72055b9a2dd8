// Batched LinearProof CREATION (LinearProof::create, src/linear_proof.rs:40-173) for many independent proofs of one size n,
// every multiscalar multiplication on the GPU engine.
//
// The reference folds the generator vector every round, G'_i = G_i + x_j G_{i+n'} (linear_proof.rs:140-144: n' two-term
// multiscalar multiplications per round), and forms L_j, R_j over the folded vector (:104-117).  As in ipp_prover.h the
// generators are never folded here: a folded generator is a known combination of the ORIGINAL ones,
//     G^(j)_i = sum_{t = i mod n_j} wG_j(t) G_t ,   wG_j(t) = prod_{r < j, bit_r(t) = 1} x_r
// (bit_r(t) = the bit of t that round r consumed), so
//     L_j = sum_{t: bit_j(t) = 1} a_L[i] wG(t) G_t + s_j B + <a_L, b_R> F
//     R_j = sum_{t: bit_j(t) = 0} a_R[i] wG(t) G_t + t_j B + <a_R, b_L> F ,   i = t mod n'
// are multiscalar multiplications of n/2 + 2 terms over the original points, and the last commitment
//     S = t* B + (s* b_0) F + s* G^(k)_0 = t* B + (s* b_0) F + sum_t (s* wG(t)) G_t            (:155-157)
// is one of n + 2 terms.  Same group elements, hence byte-identical proofs; all proofs of a batch advance together.
//
// NOT constant time (the multiscalar multiplications are indexed by the secret a): like the reference's own create(),
// which calls vartime_multiscalar_mul (:104, 112, 140).
#ifndef BPGPU_LINEAR_PROVER_H
#define BPGPU_LINEAR_PROVER_H
#include "ipp_prover.h"

namespace bp {

struct linc_shape {
    uint32_t n, k;            // n = 2^k
    uint32_t nproofs;
    uint32_t b_shared;        // != 0: b holds n scalars used by every proof
};

// lane = (proof p, index t): load a, b (must be canonical: status BP_STATUS_BAD_SCALAR otherwise); all weights start at one
BP_HD void linc_init_thread(uint32_t tid, linc_shape sh, const uint8_t *a_in, const uint8_t *b_in, uint32_t *a, uint32_t *b, uint32_t *wG,
                            uint32_t *status) {
    const uint32_t p = tid / sh.n, t = tid - p * sh.n;
    sc x;
    bool ok = true;
    load_words8(x.v, a_in + (uint64_t)tid * 32);
    ok = ok && sc_is_canonical_sc(x);
    ippc_st(a + 8 * (uint64_t)tid, x);
    load_words8(x.v, b_in + (sh.b_shared ? (uint64_t)t : (uint64_t)tid) * 32);
    ok = ok && sc_is_canonical_sc(x);
    ippc_st(b + 8 * (uint64_t)tid, x);
    sc_from_u32(x, 1);
    ippc_st(wG + 8 * (uint64_t)tid, x);
    if (!ok) status_raise(status + p, BP_STATUS_BAD_SCALAR);
}

// lane = proof: the public inputs into the transcript (linear_proof.rs:73-83), the blinding r, and the 2k + 2 scalars the
// reference draws with Scalar::random (s_j, t_j per round, then s_star, t_star: 64 bytes each, in that order).
// ts[p]: the proof's transcript after innerproduct_domain_sep(n) on entry, advanced on exit.
BP_HD void linc_public_thread(uint32_t p, linc_shape sh, kstate st, const uint8_t *C, const uint8_t *b_in, const uint8_t *G, const uint8_t *F,
                              const uint8_t *B, const uint8_t *r_in, const uint8_t *rng, uint32_t *ts, uint32_t *r_out, uint32_t *draws,
                              uint32_t *status) {
    const uint32_t n = sh.n, nd = 2 * sh.k + 2;
    uint32_t *tw = ts + (uint64_t)p * BP_TS_WORDS;
    strobe t;
    t.st = st;
    for (uint32_t i = 0; i < 50; i++) ks_set32(st, i, tw[i]);
    t.pos = tw[50] & 0xffu;
    t.pos_begin = (tw[50] >> 8) & 0xffu;
    t.cur_flags = (tw[50] >> 16) & 0xffu;
    const uint8_t lC[1] = {'C'}, lb[3] = {'b', '_', 'i'}, lG[3] = {'G', '_', 'i'}, lF[1] = {'F'}, lB[1] = {'B'};
    uint32_t w[8];
    load_words8(w, C + (uint64_t)p * 32);
    merlin_append_words8(t, lC, 1, w);
    const uint8_t *bp_ = b_in + (sh.b_shared ? 0 : (uint64_t)p * n * 32);
    for (uint32_t i = 0; i < n; i++) {
        load_words8(w, bp_ + (uint64_t)i * 32);
        merlin_append_words8(t, lb, 3, w);
    }
    for (uint32_t i = 0; i < n; i++) {
        load_words8(w, G + (uint64_t)i * 32);
        merlin_append_words8(t, lG, 3, w);
    }
    load_words8(w, F);
    merlin_append_words8(t, lF, 1, w);
    load_words8(w, B);
    merlin_append_words8(t, lB, 1, w);
    for (uint32_t i = 0; i < 50; i++) tw[i] = ks_get32(st, i);
    tw[50] = rp_ts_meta(t.pos, t.pos_begin, t.cur_flags);
    tw[51] = 0;
    sc r;
    load_words8(r.v, r_in + (uint64_t)p * 32);
    if (!sc_is_canonical_sc(r)) status_raise(status + p, BP_STATUS_BAD_SCALAR);
    ippc_st(r_out + 8 * (uint64_t)p, r);
    for (uint32_t d = 0; d < nd; d++) {
        uint32_t wide[16];
        const uint8_t *src = rng + ((uint64_t)p * nd + d) * 64;
        load_words8(wide, src);
        load_words8(wide + 8, src + 32);
        sc x;
        sc_from_wide(x, wide);
        ippc_st(draws + 8 * ((uint64_t)p * nd + d), x);
    }
}

// lane = (proof p, index t), round j: the G_t term of L (MSM 2p) or R (MSM 2p + 1); each MSM has n/2 + 2 terms:
// [0, n/2) the G terms, [n/2] the B term, [n/2 + 1] the F term (linc_q_thread)
BP_HD void linc_terms_thread(uint32_t tid, linc_shape sh, uint32_t j, const uint32_t *a, const uint32_t *wG, const uint8_t *G, uint32_t *msm_sc,
                             uint32_t *msm_pt) {
    const uint32_t n = sh.n, p = tid / n, t = tid - p * n;
    const uint32_t nj = n >> j, np = nj >> 1;
    const uint32_t tt = t & (nj - 1), i = tt & (np - 1), hi = tt >= np ? 1u : 0u, pos = (t / nj) * np + i;
    const uint32_t N = n / 2 + 2;
    const uint64_t pa = (uint64_t)p * n;
    sc x, w, r;
    uint32_t pw[8];
    // a_L[i] on L when t lies in the right half (hi), a_R[i] on R when in the left half (:104-117)
    ippc_ld(x, a + 8 * (pa + (hi ? i : i + np)));
    ippc_ld(w, wG + 8 * (pa + t));
    sc_mul(r, x, w);
    const uint64_t slot = ((uint64_t)(2 * p + (hi ? 0 : 1)) * N + pos) * 8;
    ippc_st(msm_sc + slot, r);
    load_words8(pw, G + (uint64_t)t * 32);
    for (int q = 0; q < 8; q++) msm_pt[slot + q] = pw[q];
}

// lane = proof, round j: c_L = <a_L, b_R>, c_R = <a_R, b_L> (:98-99) on F; the blinding draws s_j, t_j on B
BP_HD void linc_q_thread(uint32_t p, linc_shape sh, uint32_t j, const uint32_t *a, const uint32_t *b, const uint32_t *draws, const uint8_t *F,
                         const uint8_t *B, uint32_t *msm_sc, uint32_t *msm_pt) {
    const uint32_t n = sh.n, np = (n >> j) >> 1, N = n / 2 + 2, nd = 2 * sh.k + 2;
    const uint64_t pa = (uint64_t)p * n;
    sc c, x, y, s0, s1;
    sc_0(s0);
    sc_0(s1);
    for (uint32_t i = 0; i < np; i++) {
        ippc_ld(x, a + 8 * (pa + i));
        ippc_ld(y, b + 8 * (pa + i + np));
        sc_mul(c, x, y);
        sc_add(s0, s0, c);
        ippc_ld(x, a + 8 * (pa + i + np));
        ippc_ld(y, b + 8 * (pa + i));
        sc_mul(c, x, y);
        sc_add(s1, s1, c);
    }
    uint32_t fw[8], bw[8];
    load_words8(fw, F);
    load_words8(bw, B);
    const uint64_t sl = ((uint64_t)(2 * p) * N + n / 2) * 8, sr = ((uint64_t)(2 * p + 1) * N + n / 2) * 8;
    ippc_ld(x, draws + 8 * ((uint64_t)p * nd + 2 * j));
    ippc_ld(y, draws + 8 * ((uint64_t)p * nd + 2 * j + 1));
    ippc_st(msm_sc + sl, x);          // s_j B
    ippc_st(msm_sc + sr, y);          // t_j B
    ippc_st(msm_sc + sl + 8, s0);     // c_L F
    ippc_st(msm_sc + sr + 8, s1);     // c_R F
    for (int q = 0; q < 8; q++) {
        msm_pt[sl + q] = bw[q];
        msm_pt[sr + q] = bw[q];
        msm_pt[sl + 8 + q] = fw[q];
        msm_pt[sr + 8 + q] = fw[q];
    }
}

// ---- the same terms when G, F, B are the context's generators (bp_gens.share(0).G(n), pc_gens.B, pc_gens.B_blinding): L_j, R_j and S
// are pure generator-table MSMs.  Rows 2p (L) and 2p + 1 (R) of a [2 nproofs][n + 2] scalar array in the table order
// (B_blinding, B, G_0..): every slot is written (zero where a generator does not occur).
BP_HD void linc_terms_fixed_thread(uint32_t tid, linc_shape sh, uint32_t j, const uint32_t *a, const uint32_t *wG, uint32_t *gen_scalars) {
    const uint32_t n = sh.n, p = tid / n, t = tid - p * n;
    const uint32_t nj = n >> j, np = nj >> 1;
    const uint32_t tt = t & (nj - 1), i = tt & (np - 1), hi = tt >= np ? 1u : 0u;
    const uint32_t row_len = n + 2;
    const uint64_t pa = (uint64_t)p * n;
    uint32_t *rowL = gen_scalars + (uint64_t)(2 * p) * row_len * 8, *rowR = rowL + (uint64_t)row_len * 8;
    sc x, w, r, zero;
    sc_0(zero);
    ippc_ld(x, a + 8 * (pa + (hi ? i : i + np)));
    ippc_ld(w, wG + 8 * (pa + t));
    sc_mul(r, x, w);
    ippc_st(rowL + (uint64_t)(2 + t) * 8, hi ? r : zero);
    ippc_st(rowR + (uint64_t)(2 + t) * 8, hi ? zero : r);
}
BP_HD void linc_q_fixed_thread(uint32_t p, linc_shape sh, uint32_t j, const uint32_t *a, const uint32_t *b, const uint32_t *draws, uint32_t *gen_scalars) {
    const uint32_t n = sh.n, np = (n >> j) >> 1, row_len = n + 2, nd = 2 * sh.k + 2;
    const uint64_t pa = (uint64_t)p * n;
    sc c, x, y, s0, s1;
    sc_0(s0);
    sc_0(s1);
    for (uint32_t i = 0; i < np; i++) {
        ippc_ld(x, a + 8 * (pa + i));
        ippc_ld(y, b + 8 * (pa + i + np));
        sc_mul(c, x, y);
        sc_add(s0, s0, c);
        ippc_ld(x, a + 8 * (pa + i + np));
        ippc_ld(y, b + 8 * (pa + i));
        sc_mul(c, x, y);
        sc_add(s1, s1, c);
    }
    uint32_t *rowL = gen_scalars + (uint64_t)(2 * p) * row_len * 8, *rowR = rowL + (uint64_t)row_len * 8;
    ippc_ld(x, draws + 8 * ((uint64_t)p * nd + 2 * j));
    ippc_ld(y, draws + 8 * ((uint64_t)p * nd + 2 * j + 1));
    ippc_st(rowL, x);          // s_j on B (= the table's B_blinding)
    ippc_st(rowR, y);          // t_j
    ippc_st(rowL + 8, s0);     // c_L on F (= the table's B)
    ippc_st(rowR + 8, s1);     // c_R
}
// S: row p of a [nproofs][n + 2] array
BP_HD void linc_sterms_fixed_thread(uint32_t tid, linc_shape sh, const uint32_t *wG, const uint32_t *draws, uint32_t *gen_scalars) {
    const uint32_t n = sh.n, p = tid / n, t = tid - p * n, nd = 2 * sh.k + 2;
    sc s_star, w, r;
    ippc_ld(s_star, draws + 8 * ((uint64_t)p * nd + 2 * sh.k));
    ippc_ld(w, wG + 8 * (uint64_t)tid);
    sc_mul(r, s_star, w);
    ippc_st(gen_scalars + ((uint64_t)p * (n + 2) + 2 + t) * 8, r);
}
BP_HD void linc_sq_fixed_thread(uint32_t p, linc_shape sh, const uint32_t *b, const uint32_t *draws, uint32_t *gen_scalars) {
    const uint32_t n = sh.n, nd = 2 * sh.k + 2;
    sc s_star, t_star, b0, r;
    ippc_ld(s_star, draws + 8 * ((uint64_t)p * nd + 2 * sh.k));
    ippc_ld(t_star, draws + 8 * ((uint64_t)p * nd + 2 * sh.k + 1));
    ippc_ld(b0, b + 8 * (uint64_t)p * n);
    sc_mul(r, s_star, b0);
    uint32_t *row = gen_scalars + (uint64_t)p * (n + 2) * 8;
    ippc_st(row, t_star);
    ippc_st(row + 8, r);
}

// lane = proof, after round j's MSMs: L, R -> proof bytes and transcript (:119-126), x_j and its inverse,
// r <- r + x_j s_j + x_j^-1 t_j (:148)
BP_HD void linc_challenge_thread(uint32_t p, linc_shape sh, uint32_t j, kstate st, const uint32_t *msm_out /*[2 nproofs][8]*/, const uint8_t *msm_status,
                                 uint32_t *ts, const uint32_t *draws, uint32_t *r_io, uint32_t *x_out, uint32_t *xinv_out, uint8_t *proofs,
                                 uint32_t proof_len, uint32_t *status) {
    const uint32_t nd = 2 * sh.k + 2;
    uint32_t *tw = ts + (uint64_t)p * BP_TS_WORDS;
    strobe t;
    t.st = st;
    for (uint32_t i = 0; i < 50; i++) ks_set32(st, i, tw[i]);
    t.pos = tw[50] & 0xffu;
    t.pos_begin = (tw[50] >> 8) & 0xffu;
    t.cur_flags = (tw[50] >> 16) & 0xffu;
    const uint8_t lL[1] = {'L'}, lR[1] = {'R'}, lx[3] = {'x', '_', 'j'};
    uint32_t w[8];
    uint32_t *dst = (uint32_t *)(proofs + (uint64_t)p * proof_len + 64 * j);
    if (msm_status[2 * p] | msm_status[2 * p + 1]) status_raise(status + p, msm_status[2 * p] > msm_status[2 * p + 1] ? msm_status[2 * p] : msm_status[2 * p + 1]);
    for (int q = 0; q < 8; q++) w[q] = msm_out[8 * (uint64_t)(2 * p) + q];
    for (int q = 0; q < 8; q++) dst[q] = w[q];
    merlin_append_words8(t, lL, 1, w);
    for (int q = 0; q < 8; q++) w[q] = msm_out[8 * (uint64_t)(2 * p + 1) + q];
    for (int q = 0; q < 8; q++) dst[8 + q] = w[q];
    merlin_append_words8(t, lR, 1, w);
    sc x, xi, r, sj, tj, m0;
    rp_challenge_scalar(t, lx, 3, x);
    sc_invert_safegcd(xi, x);
    ippc_st(x_out + 8 * (uint64_t)p, x);
    ippc_st(xinv_out + 8 * (uint64_t)p, xi);
    ippc_ld(r, r_io + 8 * (uint64_t)p);
    ippc_ld(sj, draws + 8 * ((uint64_t)p * nd + 2 * j));
    ippc_ld(tj, draws + 8 * ((uint64_t)p * nd + 2 * j + 1));
    sc_mul(m0, x, sj);
    sc_add(r, r, m0);
    sc_mul(m0, xi, tj);
    sc_add(r, r, m0);
    ippc_st(r_io + 8 * (uint64_t)p, r);
    for (uint32_t i = 0; i < 50; i++) tw[i] = ks_get32(st, i);
    tw[50] = rp_ts_meta(t.pos, t.pos_begin, t.cur_flags);
    tw[51] = 0;
}

// lane = (proof, t), round j: fold a, b (:133-139) and advance the generator weights (:140-144)
BP_HD void linc_fold_thread(uint32_t tid, linc_shape sh, uint32_t j, const uint32_t *x_all, const uint32_t *xinv_all, uint32_t *a, uint32_t *b,
                            uint32_t *wG) {
    const uint32_t n = sh.n, p = tid / n, t = tid - p * n;
    const uint32_t nj = n >> j, np = nj >> 1;
    const uint64_t pa = (uint64_t)p * n;
    sc x, xi, u, v, r;
    ippc_ld(x, x_all + 8 * (uint64_t)p);
    ippc_ld(xi, xinv_all + 8 * (uint64_t)p);
    if ((t & (nj - 1)) >= np) {
        ippc_ld(u, wG + 8 * (pa + t));
        sc_mul(r, u, x);
        ippc_st(wG + 8 * (pa + t), r);         // G_L + x_j G_R: the right half picks up x_j
    }
    if (t < np) {
        ippc_ld(u, a + 8 * (pa + t));
        ippc_ld(v, a + 8 * (pa + t + np));
        sc_mul(r, xi, v);
        sc_add(r, u, r);
        ippc_st(a + 8 * (pa + t), r);          // a_L[i] + x_j^-1 a_R[i]
        ippc_ld(u, b + 8 * (pa + t));
        ippc_ld(v, b + 8 * (pa + t + np));
        sc_mul(r, x, v);
        sc_add(r, u, r);
        ippc_st(b + 8 * (pa + t), r);          // b_L[i] + x_j b_R[i]
    }
}

// the terms of S (one MSM of n + 2 terms per proof).  lane = (proof, t): (s* wG(t)) G_t at slot t
BP_HD void linc_sterms_thread(uint32_t tid, linc_shape sh, const uint32_t *wG, const uint32_t *draws, const uint8_t *G, uint32_t *msm_sc, uint32_t *msm_pt) {
    const uint32_t n = sh.n, p = tid / n, t = tid - p * n, nd = 2 * sh.k + 2, N = n + 2;
    sc s_star, w, r;
    ippc_ld(s_star, draws + 8 * ((uint64_t)p * nd + 2 * sh.k));
    ippc_ld(w, wG + 8 * (uint64_t)tid);
    sc_mul(r, s_star, w);
    const uint64_t slot = ((uint64_t)p * N + t) * 8;
    ippc_st(msm_sc + slot, r);
    uint32_t pw[8];
    load_words8(pw, G + (uint64_t)t * 32);
    for (int q = 0; q < 8; q++) msm_pt[slot + q] = pw[q];
}
// lane = proof: t* B at slot n, (s* b_0) F at slot n + 1
BP_HD void linc_sq_thread(uint32_t p, linc_shape sh, const uint32_t *b, const uint32_t *draws, const uint8_t *F, const uint8_t *B, uint32_t *msm_sc,
                          uint32_t *msm_pt) {
    const uint32_t n = sh.n, nd = 2 * sh.k + 2, N = n + 2;
    sc s_star, t_star, b0, r;
    ippc_ld(s_star, draws + 8 * ((uint64_t)p * nd + 2 * sh.k));
    ippc_ld(t_star, draws + 8 * ((uint64_t)p * nd + 2 * sh.k + 1));
    ippc_ld(b0, b + 8 * (uint64_t)p * n);
    sc_mul(r, s_star, b0);
    const uint64_t slot = ((uint64_t)p * N + n) * 8;
    ippc_st(msm_sc + slot, t_star);
    ippc_st(msm_sc + slot + 8, r);
    uint32_t fw[8], bw[8];
    load_words8(fw, F);
    load_words8(bw, B);
    for (int q = 0; q < 8; q++) {
        msm_pt[slot + q] = bw[q];
        msm_pt[slot + 8 + q] = fw[q];
    }
}

// lane = proof: S -> proof bytes and transcript, x_star, a_star = s* + x* a_0, r_star = t* + x* r (:158-171)
BP_HD void linc_final_thread(uint32_t p, linc_shape sh, kstate st, const uint32_t *msm_out /*[nproofs][8]*/, const uint8_t *msm_status, uint32_t *ts,
                             const uint32_t *a, const uint32_t *draws, const uint32_t *r_in, uint8_t *proofs, uint32_t proof_len, uint32_t *status) {
    const uint32_t nd = 2 * sh.k + 2;
    uint32_t *tw = ts + (uint64_t)p * BP_TS_WORDS;
    strobe t;
    t.st = st;
    for (uint32_t i = 0; i < 50; i++) ks_set32(st, i, tw[i]);
    t.pos = tw[50] & 0xffu;
    t.pos_begin = (tw[50] >> 8) & 0xffu;
    t.cur_flags = (tw[50] >> 16) & 0xffu;
    const uint8_t lS[1] = {'S'}, lxs[6] = {'x', '_', 's', 't', 'a', 'r'};
    if (msm_status[p]) status_raise(status + p, msm_status[p]);
    uint32_t w[8];
    uint32_t *dst = (uint32_t *)(proofs + (uint64_t)p * proof_len + 64 * sh.k);
    for (int q = 0; q < 8; q++) w[q] = msm_out[8 * (uint64_t)p + q];
    for (int q = 0; q < 8; q++) dst[q] = w[q];
    merlin_append_words8(t, lS, 1, w);
    sc xs, s_star, t_star, a0, r, m0;
    rp_challenge_scalar(t, lxs, 6, xs);
    ippc_ld(s_star, draws + 8 * ((uint64_t)p * nd + 2 * sh.k));
    ippc_ld(t_star, draws + 8 * ((uint64_t)p * nd + 2 * sh.k + 1));
    ippc_ld(a0, a + 8 * (uint64_t)p * sh.n);
    ippc_ld(r, r_in + 8 * (uint64_t)p);
    sc_mul(m0, xs, a0);
    sc_add(m0, s_star, m0);
    for (int q = 0; q < 8; q++) dst[8 + q] = m0.v[q];
    sc_mul(m0, xs, r);
    sc_add(m0, t_star, m0);
    for (int q = 0; q < 8; q++) dst[16 + q] = m0.v[q];
    for (uint32_t i = 0; i < 50; i++) tw[i] = ks_get32(st, i);
    tw[50] = rp_ts_meta(t.pos, t.pos_begin, t.cur_flags);
    tw[51] = 0;
}

}  // namespace bp
#endif
