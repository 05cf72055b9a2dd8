// Stand-alone inner-product-proof verification front end (device):
// InnerProductProof::from_bytes (src/inner_product_proof.rs:373-407) + verify (ipp.rs:260-326) up to the
// multiscalar multiplication.  Lane = proof: replay the transcript (L_i, R_i -> u_i), batch-invert the
// challenges, and emit the 2n + 2k + 2 (scalar, point) terms of
//     a*b * Q + sum_i (a s_i g_i) G_i + sum_i (b s_i^-1 h_i) H_i - sum_j u_j^2 L_j - sum_j u_j^-2 R_j - P
// whose sum is the identity iff expect_P == P (ipp.rs:308-325).  The MSM itself is bpgpu_msm_batch's.
#ifndef BPGPU_IPP_H
#define BPGPU_IPP_H
#include "rangeproof.h"

namespace bp {

struct ipp_shape {
    uint32_t n, k;              // k = lg(n) as implied by the proof length
    uint32_t N;                 // terms: 2n + 2k + 2
    uint32_t proof_len, nproofs;
    uint32_t shape_verdict;     // != 0: n != 2^k (VerificationError, ipp.rs:203-211): only parse
    uint32_t bases_shared;      // != 0: G, H hold n encodings used by every proof (the reference's callers pass bp_gens.G/H)
};

// thread p.  Outputs are pre-zeroed by the host, so rejected proofs contribute identity terms.
BP_HD void ipp_prepare_thread(uint32_t p, ipp_shape sh, const rp_strobe_init &init, kstate st, const uint8_t *proofs,
                              const uint8_t *Gf, const uint8_t *Hf, const uint8_t *P, const uint8_t *Q, const uint8_t *G,
                              const uint8_t *H, uint32_t *scalars, uint32_t *points, uint32_t *status) {
    const uint32_t n = sh.n, k = sh.k;
    const uint8_t *pr = proofs + (uint64_t)p * sh.proof_len;
    sc a, b;
    load_words8(a.v, pr + 64 * k);
    load_words8(b.v, pr + 64 * k + 32);
    if (!sc_is_canonical_sc(a) || !sc_is_canonical_sc(b)) {
        status[p] = BP_VERDICT_FORMAT;
        return;
    }
    if (sh.shape_verdict) {
        status[p] = sh.shape_verdict;
        return;
    }
    strobe t;
    t.st = st;
    for (uint32_t i = 0; i < 50; i++) ks_set32(st, i, init.w[i]);
    t.pos = init.pos;
    t.pos_begin = init.pos_begin;
    t.cur_flags = init.cur_flags;
    uint32_t *sc_out = scalars + (uint64_t)p * sh.N * 8, *pt_out = points + (uint64_t)p * sh.N * 8;
    const uint8_t lL[1] = {'L'}, lR[1] = {'R'}, lu[1] = {'u'};
    sc28 um[BP_RP_MAX_K], uim[BP_RP_MAX_K], acc, inv;
    sc28_one_mont(acc);
    uint32_t w[8];
    bool verr = false;
    for (uint32_t i = 0; i < k; i++) {
        load_words8(w, pr + 64 * i);
        verr = verr || words8_zero(w);
        merlin_append_words8(t, lL, 1, w);
        for (int q = 0; q < 8; q++) pt_out[(1 + 2 * n + i) * 8 + q] = w[q];
        load_words8(w, pr + 64 * i + 32);
        verr = verr || words8_zero(w);
        merlin_append_words8(t, lR, 1, w);
        for (int q = 0; q < 8; q++) pt_out[(1 + 2 * n + k + i) * 8 + q] = w[q];
        sc u;
        rp_challenge_scalar(t, lu, 1, u);
        sc_to_mont28(um[i], u);
        uim[i] = acc;                                   // prefix product before u_i
        sc28_montmul(acc, acc, um[i]);
    }
    if (verr) {
        status_raise(status + p, BP_VERDICT_VERIFICATION);
        return;
    }
    sc28_invert_mont_safegcd(inv, acc);
    for (uint32_t ii = k; ii-- > 0;) {
        sc28 ui;
        sc28_montmul(ui, inv, uim[ii]);
        sc28_montmul(inv, inv, um[ii]);
        uim[ii] = ui;
        sc28 sq;
        sc t0;
        sc28_montsq(sq, um[ii]);
        sc_from_mont28(t0, sq);
        sc_neg(t0, t0);                                 // -u_i^2 on L_i
        store_words8(sc_out + (1 + 2 * n + ii) * 8, t0);
        sc28_montsq(sq, ui);
        sc_from_mont28(t0, sq);
        sc_neg(t0, t0);                                 // -u_i^-2 on R_i
        store_words8(sc_out + (1 + 2 * n + k + ii) * 8, t0);
    }
    sc28 am, bm, abm;
    sc t0;
    sc_to_mont28(am, a);
    sc_to_mont28(bm, b);
    sc28_montmul(abm, am, bm);
    sc_from_mont28(t0, abm);                            // a*b on Q
    store_words8(sc_out, t0);
    load_words8(w, Q + (uint64_t)p * 32);
    for (int q = 0; q < 8; q++) pt_out[q] = w[q];
    // s_i = prod_b (bit_b(i) ? u : u^-1)[k-1-b]; its inverse has the factors swapped (ipp.rs:241-250, 283).  Walked in
    // Gray-code order: from one index to the next exactly one bit b flips, so s picks up u_j^2 or u_j^-2 (j = k-1-b) and
    // 1/s the other one -- two products per index instead of 2k
    sc28 usq[BP_RP_MAX_K], uisq[BP_RP_MAX_K], s, sinv;
    sc28_one_mont(s);
    for (uint32_t j = 0; j < k; j++) {
        sc28_montsq(usq[j], um[j]);
        sc28_montsq(uisq[j], uim[j]);
        sc28_montmul(s, s, uim[j]);                      // s_0 = prod_j u_j^-1
    }
    sc28_one_mont(sinv);
    for (uint32_t j = 0; j < k; j++) sc28_montmul(sinv, sinv, um[j]);   // 1 / s_0
    for (uint32_t g = 0; g < n; g++) {
        const uint32_t i = g ^ (g >> 1);
        if (g) {
            const uint32_t bb = (uint32_t)__builtin_ctz(g), j = k - 1 - bb;
            const bool set = (i >> bb) & 1;
            sc28_montmul(s, s, set ? usq[j] : uisq[j]);
            sc28_montmul(sinv, sinv, set ? uisq[j] : usq[j]);
        }
        sc f;
        sc28 fm, r;
        load_words8(f.v, Gf + ((uint64_t)p * n + i) * 32);
        sc_to_mont28(fm, f);
        sc28_montmul(r, am, s);
        sc28_montmul(r, r, fm);
        sc_from_mont28(t0, r);                          // (a s_i) g_i on G_i
        store_words8(sc_out + (1 + i) * 8, t0);
        load_words8(f.v, Hf + ((uint64_t)p * n + i) * 32);
        sc_to_mont28(fm, f);
        sc28_montmul(r, bm, sinv);
        sc28_montmul(r, r, fm);
        sc_from_mont28(t0, r);                          // (b / s_i) h_i on H_i
        store_words8(sc_out + (1 + n + i) * 8, t0);
        const uint64_t bi = sh.bases_shared ? (uint64_t)i : (uint64_t)p * n + i;
        load_words8(w, G + bi * 32);
        for (int q = 0; q < 8; q++) pt_out[(1 + i) * 8 + q] = w[q];
        load_words8(w, H + bi * 32);
        for (int q = 0; q < 8; q++) pt_out[(1 + n + i) * 8 + q] = w[q];
    }
    // - P
    {
        sc one, m1;
        sc_from_u32(one, 1);
        sc_neg(m1, one);
        store_words8(sc_out + (1 + 2 * n + 2 * k) * 8, m1);
        load_words8(w, P + (uint64_t)p * 32);
        for (int q = 0; q < 8; q++) pt_out[(1 + 2 * n + 2 * k) * 8 + q] = w[q];
    }
}

// ---- InnerProductProof::verification_scalars alone (ipp.rs:198-253) -- bpgpu_ipp_verification_scalars ------------------
// The other caller besides InnerProductProof::verify is the R1CS verifier (r1cs/verifier.rs:401-404), which folds u_i^2,
// u_i^-2 and s_i into its own multiscalar multiplication.
// thread p: from_bytes' canonical checks, transcript replay (L_i, R_i -> u_i), one batch inversion; writes u_i^2, u_i^-2
// (canonical bytes), the Montgomery tables (u_i, u_i^-1) for ipp_vs_s_thread, the advanced transcript.
BP_HD void ipp_vs_front_thread(uint32_t p, ipp_shape sh, const rp_strobe_init &init, kstate st, const uint8_t *proofs, const uint32_t *ts_in,
                               uint32_t *u_sq, uint32_t *u_inv_sq, uint32_t *tab /*[nproofs][2k][10]*/, uint32_t *ts_out, uint32_t *status) {
    const uint32_t k = sh.k;
    const uint8_t *pr = proofs + (uint64_t)p * sh.proof_len;
    sc a, b;
    load_words8(a.v, pr + 64 * k);
    load_words8(b.v, pr + 64 * k + 32);
    rp_strobe_init none{};
    if (!sc_is_canonical_sc(a) || !sc_is_canonical_sc(b)) {
        status[p] = BP_VERDICT_FORMAT;
        rp_ts_passthrough(p, ts_in ? none : init, ts_in, ts_out);
        return;
    }
    if (sh.shape_verdict) {
        status[p] = sh.shape_verdict;
        rp_ts_passthrough(p, ts_in ? none : init, ts_in, ts_out);
        return;
    }
    strobe t;
    t.st = st;
    if (ts_in) {   // per-proof transcripts: innerproduct_domain_sep(n) is applied here
        const uint32_t *src = ts_in + (uint64_t)p * BP_TS_WORDS;
        for (uint32_t i = 0; i < 50; i++) ks_set32(st, i, src[i]);
        const uint32_t meta = src[50];
        t.pos = meta & 0xffu;
        t.pos_begin = (meta >> 8) & 0xffu;
        t.cur_flags = (meta >> 16) & 0xffu;
        const uint8_t dsep[7] = {'d', 'o', 'm', '-', 's', 'e', 'p'}, ipp[6] = {'i', 'p', 'p', ' ', 'v', '1'}, ln[1] = {'n'};
        merlin_append_message(t, dsep, 7, ipp, 6);
        merlin_append_u64(t, ln, 1, sh.n);
    } else {       // one start state for the batch: the host already applied the domain separator
        for (uint32_t i = 0; i < 50; i++) ks_set32(st, i, init.w[i]);
        t.pos = init.pos;
        t.pos_begin = init.pos_begin;
        t.cur_flags = init.cur_flags;
    }
    const uint8_t lL[1] = {'L'}, lR[1] = {'R'}, lu[1] = {'u'};
    sc28 um[BP_RP_MAX_K], uim[BP_RP_MAX_K], acc, inv;
    sc28_one_mont(acc);
    uint32_t w[8];
    bool verr = false;
    for (uint32_t i = 0; i < k; i++) {
        // validate_and_append_point returns Err BEFORE it absorbs an identity point (transcript.rs:75-87; ipp.rs:217-222): the caller's
        // transcript is handed back as of that moment -- domain separator and earlier rounds in
        load_words8(w, pr + 64 * i);
        if (words8_zero(w)) {
            if (!verr && ts_out) rp_ts_emit(p, st, rp_ts_meta(t.pos, t.pos_begin, t.cur_flags), ts_out);
            verr = true;
        }
        merlin_append_words8(t, lL, 1, w);
        load_words8(w, pr + 64 * i + 32);
        if (words8_zero(w)) {
            if (!verr && ts_out) rp_ts_emit(p, st, rp_ts_meta(t.pos, t.pos_begin, t.cur_flags), ts_out);
            verr = true;
        }
        merlin_append_words8(t, lR, 1, w);
        sc u;
        rp_challenge_scalar(t, lu, 1, u);
        sc_to_mont28(um[i], u);
        uim[i] = acc;
        sc28_montmul(acc, acc, um[i]);
    }
    if (verr) {   // an identity L_i / R_i is a VerificationError
        status[p] = BP_VERDICT_VERIFICATION;
        return;
    }
    sc28_invert_mont_safegcd(inv, acc);
    uint32_t *tb = tab + (uint64_t)p * 2 * k * 10;
    for (uint32_t ii = k; ii-- > 0;) {
        sc28 ui, sq;
        sc t0;
        sc28_montmul(ui, inv, uim[ii]);
        sc28_montmul(inv, inv, um[ii]);
        sc28_montsq(sq, um[ii]);
        sc_from_mont28(t0, sq);
        store_words8(u_sq + ((uint64_t)p * k + ii) * 8, t0);
        sc28_montsq(sq, ui);
        sc_from_mont28(t0, sq);
        store_words8(u_inv_sq + ((uint64_t)p * k + ii) * 8, t0);
#pragma unroll
        for (int q = 0; q < 10; q++) {
            tb[(2 * ii) * 10 + q] = um[ii].v[q];
            tb[(2 * ii + 1) * 10 + q] = ui.v[q];
        }
    }
    if (ts_out) {
        uint32_t *o = ts_out + (uint64_t)p * BP_TS_WORDS;
        for (uint32_t i = 0; i < 50; i++) o[i] = ks_get32(st, i);
        o[50] = rp_ts_meta(t.pos, t.pos_begin, t.cur_flags);
        o[51] = 0;
    }
}
// thread tid = i * nproofs + p: s_i = prod_b (bit_b(i) ? u : u^-1)[k-1-b]  (the closed form of the induction of ipp.rs:241-250)
BP_HD void ipp_vs_s_thread(uint32_t tid, ipp_shape sh, const uint32_t *tab, const uint32_t *status, uint32_t *s_out) {
    const uint32_t i = tid / sh.nproofs, p = tid - i * sh.nproofs, k = sh.k;
    uint32_t *dst = s_out + ((uint64_t)p * sh.n + i) * 8;
    if (status[p] != 0) {
        for (int q = 0; q < 8; q++) dst[q] = 0;
        return;
    }
    const uint32_t *tb = tab + (uint64_t)p * 2 * k * 10;
    sc28 acc;
    sc28_one_mont(acc);
    for (uint32_t bb = 0; bb < k; bb++) {
        const uint32_t j = k - 1 - bb;
        const uint32_t *f = tb + (2 * j + (((i >> bb) & 1) ? 0 : 1)) * 10;
        sc28 fm;
#pragma unroll
        for (int q = 0; q < 10; q++) fm.v[q] = f[q];
        sc28_montmul(acc, acc, fm);
    }
    sc t0;
    sc_from_mont28(t0, acc);
    store_words8(dst, t0);
}

// thread p: verdict = front-end status if set, else VerificationError when a point failed to decode or the
// difference expect_P - P is not the identity
BP_HD void ipp_verdict_thread(uint32_t p, const uint32_t *status, const uint8_t *msm_status, const uint32_t *msm_out, uint8_t *verdict) {
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) nz |= msm_out[8 * (uint64_t)p + i];
    verdict[p] = status[p] ? (uint8_t)status[p] : ((msm_status[p] != 0 || nz != 0) ? BP_VERDICT_VERIFICATION : BP_VERDICT_OK);
}

}  // namespace bp
#endif
