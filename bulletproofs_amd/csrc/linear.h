// LinearProof verification front end (device): LinearProof::from_bytes (src/linear_proof.rs:350-394) + verify
// (linear_proof.rs:175-236) + verification_scalars (:240-290) + subset_product (:299-314) up to the multiscalar
// multiplication.  Lane = proof: absorb the public inputs (C, b_i, G_i, F, B), replay the rounds (L_j, R_j -> x_j),
// absorb S -> x_star, batch-invert the challenges and emit the n + 2k + 4 (scalar, point) terms of
//     r B + (a b_0) F - x* C - sum_j (x* x_j) L_j - sum_j (x* / x_j) R_j + sum_i (a s_i) G_i - S
// whose sum is the identity iff expect_S == S (:214-236).  The reference evaluates the sum in three multiscalar
// multiplications and two scalar products; the group element is the same.  b_0, the fold of the public vector
// (b_L += x_j b_R, :279-283), equals <s, b> with the same subset products s_i that weigh the G_i, so the lane never
// holds the vector: it streams b once for the transcript and once for the dot product, walking i in Gray-code order
// (one Montgomery product per s_i: times x_j or x_j^-1 for the bit that flips).
#ifndef BPGPU_LINEAR_H
#define BPGPU_LINEAR_H
#include "rangeproof.h"

namespace bp {

struct lin_shape {
    uint32_t n, k;              // k = lg(n) as implied by the proof length
    uint32_t N;                 // terms per proof in the (scalar, point) lists: n + 2k + 4, or 2k + 2 in generator-table mode
    uint32_t proof_len, nproofs;
    uint32_t shape_verdict;     // != 0: n != 2^k (VerificationError, linear_proof.rs:263-265): only parse
    uint32_t b_shared;          // != 0: b holds n scalars used by every proof
    uint32_t fixed;             // != 0: G, F, B are the context's G(n) of party 0, B, B_blinding: their coefficients go to a row of
                                // generator-table scalars (B_blinding, B, G_0..) and only C, L_j, R_j, S stay in the lists
};

// thread p.  Outputs are pre-zeroed by the host, so rejected proofs contribute identity terms.
// Term order: B, F, C, L_0.., R_0.., G_0.., S; in generator-table mode (gen_sc != NULL) C, L_0.., R_0.., S in the lists and
// (r, a b_0, a s_0, ..) in the proof's row of gen_sc.
// ts_out (optional): the transcript as verify() leaves it (after the x_star challenge) for proofs that reach the final
// check; the start state for proofs rejected before that.
BP_HD void lin_prepare_thread(uint32_t p, lin_shape sh, const rp_strobe_init &init, kstate st, const uint8_t *proofs, const uint8_t *C,
                              const uint8_t *bvec, const uint8_t *G, const uint8_t *F, const uint8_t *B, uint32_t *scalars,
                              uint32_t *points, uint32_t *status, uint32_t *ts_out = nullptr, uint32_t *gen_sc = nullptr) {
    const uint32_t n = sh.n, k = sh.k;
    const bool fx = gen_sc != nullptr;
    const uint32_t iC = fx ? 0u : 2u, iL = iC + 1, iR = iL + k, iS = fx ? 2 * k + 1 : 3 + 2 * k + n, iG = 3 + 2 * k;
    uint32_t *grow = fx ? gen_sc + (uint64_t)p * (n + 2) * 8 : nullptr;
    const uint8_t *pr = proofs + (uint64_t)p * sh.proof_len;
    const uint8_t *Sb = pr + 64 * k;
    sc a, r;
    load_words8(a.v, Sb + 32);
    load_words8(r.v, Sb + 64);
    if (!sc_is_canonical_sc(a) || !sc_is_canonical_sc(r)) {
        status[p] = BP_VERDICT_FORMAT;
        rp_ts_passthrough(p, init, nullptr, ts_out);
        return;
    }
    if (sh.shape_verdict) {
        status[p] = sh.shape_verdict;
        rp_ts_passthrough(p, init, nullptr, ts_out);
        return;
    }
    const uint8_t *bp_ = bvec + (sh.b_shared ? 0 : (uint64_t)p * n * 32);
    strobe t;
    t.st = st;
    for (uint32_t i = 0; i < 50; i++) ks_set32(st, i, init.w[i]);
    t.pos = init.pos;
    t.pos_begin = init.pos_begin;
    t.cur_flags = init.cur_flags;
    uint32_t *sc_out = scalars + (uint64_t)p * sh.N * 8, *pt_out = points + (uint64_t)p * sh.N * 8;
    const uint8_t lC[1] = {'C'}, lb[3] = {'b', '_', 'i'}, lG[3] = {'G', '_', 'i'}, lF[1] = {'F'}, lB[1] = {'B'}, lL[1] = {'L'}, lR[1] = {'R'},
                  lx[3] = {'x', '_', 'j'}, lS[1] = {'S'}, lxs[6] = {'x', '_', 's', 't', 'a', 'r'};
    uint32_t w[8];
    // public inputs (:196-206)
    load_words8(w, C + (uint64_t)p * 32);
    merlin_append_words8(t, lC, 1, w);
    for (int q = 0; q < 8; q++) pt_out[iC * 8 + q] = w[q];
    bool fmt = false;
    for (uint32_t i = 0; i < n; i++) {
        load_words8(w, bp_ + (uint64_t)i * 32);
        fmt = fmt || sc_geq_l(w);                          // b is a Vec<Scalar> upstream: canonical by type
        merlin_append_words8(t, lb, 3, w);
    }
    if (fmt) {
        status[p] = BP_VERDICT_FORMAT;
        rp_ts_passthrough(p, init, nullptr, ts_out);
        return;
    }
    for (uint32_t i = 0; i < n; i++) {
        load_words8(w, G + (uint64_t)i * 32);
        merlin_append_words8(t, lG, 3, w);
        if (!fx)
            for (int q = 0; q < 8; q++) pt_out[(iG + i) * 8 + q] = w[q];
    }
    load_words8(w, F);
    merlin_append_words8(t, lF, 1, w);
    if (!fx)
        for (int q = 0; q < 8; q++) pt_out[1 * 8 + q] = w[q];
    load_words8(w, B);
    merlin_append_words8(t, lB, 1, w);
    if (!fx)
        for (int q = 0; q < 8; q++) pt_out[q] = w[q];
    // rounds (:271-277)
    sc28 xm[BP_RP_MAX_K], xim[BP_RP_MAX_K], acc, inv;
    sc28_one_mont(acc);
    bool verr = false;
    for (uint32_t j = 0; j < k; j++) {
        load_words8(w, pr + 64 * j);
        verr = verr || words8_zero(w);
        merlin_append_words8(t, lL, 1, w);
        for (int q = 0; q < 8; q++) pt_out[(iL + j) * 8 + q] = w[q];
        load_words8(w, pr + 64 * j + 32);
        verr = verr || words8_zero(w);
        merlin_append_words8(t, lR, 1, w);
        for (int q = 0; q < 8; q++) pt_out[(iR + j) * 8 + q] = w[q];
        sc x;
        rp_challenge_scalar(t, lx, 3, x);
        sc_to_mont28(xm[j], x);
        xim[j] = acc;                                   // prefix product before x_j
        sc28_montmul(acc, acc, xm[j]);
    }
    if (verr) {
        status_raise(status + p, BP_VERDICT_VERIFICATION);
        rp_ts_passthrough(p, init, nullptr, ts_out);
        return;
    }
    load_words8(w, Sb);
    merlin_append_words8(t, lS, 1, w);
    for (int q = 0; q < 8; q++) pt_out[iS * 8 + q] = w[q];
    sc xs;
    rp_challenge_scalar(t, lxs, 6, xs);
    if (ts_out) {
        uint32_t *o = ts_out + (uint64_t)p * BP_TS_WORDS;
        for (uint32_t i = 0; i < 50; i++) o[i] = ks_get32(st, i);
        o[50] = rp_ts_meta(t.pos, t.pos_begin, t.cur_flags);
        o[51] = 0;
    }
    sc nxs;
    sc_neg(nxs, xs);
    sc28 nxsm;
    sc_to_mont28(nxsm, nxs);
    // inverses of the challenges (:286-288), then -x* x_j on L_j and -x* / x_j on R_j
    sc28_invert_mont_safegcd(inv, acc);
    sc t0;
    for (uint32_t jj = k; jj-- > 0;) {
        sc28 xi, pm;
        sc28_montmul(xi, inv, xim[jj]);
        sc28_montmul(inv, inv, xm[jj]);
        xim[jj] = xi;
        sc28_montmul(pm, nxsm, xm[jj]);
        sc_from_mont28(t0, pm);
        store_words8(sc_out + (iL + jj) * 8, t0);
        sc28_montmul(pm, nxsm, xi);
        sc_from_mont28(t0, pm);
        store_words8(sc_out + (iR + jj) * 8, t0);
    }
    store_words8(sc_out + iC * 8, nxs);                 // -x* on C
    store_words8(fx ? grow : sc_out, r);                // r on B
    // s_i (subset products, :299-314) in Gray-code order; a s_i on G_i; b_0 = <s, b>
    sc28 am, s;
    sc_to_mont28(am, a);
    sc28_one_mont(s);
    sc b0;
    sc_0(b0);
    for (uint32_t g = 0; g < n; g++) {
        const uint32_t i = g ^ (g >> 1);
        if (g) {
            const uint32_t bb = (uint32_t)__builtin_ctz(g);                   // the bit in which i differs from its predecessor
            const bool set = (i >> bb) & 1;
            sc28_montmul(s, s, set ? xm[k - 1 - bb] : xim[k - 1 - bb]);      // bit b of i <-> challenges[lg_n - 1 - b]
        }
        sc28 pm, bi;
        sc28_montmul(pm, am, s);
        sc_from_mont28(t0, pm);
        store_words8(fx ? grow + (2 + i) * 8 : sc_out + (iG + i) * 8, t0);
        load_words8(w, bp_ + (uint64_t)i * 32);
        sc28_from_words(bi, w);
        sc28_montmul(pm, s, bi);                        // Montgomery-form s times plain b_i = plain s_i b_i
        sc_from_sc28(t0, pm);
        sc_add(b0, b0, t0);
    }
    {   // (a b_0) on F
        sc28 bm, pm;
        sc_to_mont28(bm, b0);
        sc28_montmul(pm, am, bm);
        sc_from_mont28(t0, pm);
        store_words8(fx ? grow + 8 : sc_out + 1 * 8, t0);
    }
    {   // - S
        sc one, m1;
        sc_from_u32(one, 1);
        sc_neg(m1, one);
        store_words8(sc_out + iS * 8, m1);
    }
}

}  // namespace bp
#endif
