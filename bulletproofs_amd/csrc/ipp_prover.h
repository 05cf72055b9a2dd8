// Batched inner-product-proof CREATION (InnerProductProof::create, src/inner_product_proof.rs:38-193) for many
// independent proofs of one size n, every multiscalar multiplication on the GPU engine.
//
// The reference folds the generator vectors every round: G'_i = u^-1 G_i + u G_{i+n'} (ipp.rs:127-133, 153-162: 2 n'
// two-term multiscalar multiplications per round), then forms L and R over the folded vectors (ipp.rs:87-113).  Here
// the generators are never folded: a folded generator is a known linear combination of the ORIGINAL ones,
//     G^(j)_i = sum_{t = i mod n_j} wG_j(t) G_t ,   wG_j(t) = G_factor_t * prod_{r < j} (bit_r(t) ? u_r : u_r^-1)
// (bit_r(t) = the bit of t that round r consumed; H likewise with u and u^-1 swapped), so L_j and R_j are multiscalar
// multiplications of n + 1 terms over the original points with coefficients a^(j)_i wG_j(t) / b^(j)_i wH_j(t):
//     L_j = sum_{t: bit_j(t) = 1} a_L[i] wG(t) G_t + sum_{t: bit_j(t) = 0} b_R[i] wH(t) H_t + <a_L, b_R> Q
//     R_j = sum_{t: bit_j(t) = 0} a_R[i] wG(t) G_t + sum_{t: bit_j(t) = 1} b_L[i] wH(t) H_t + <a_R, b_L> Q ,  i = t mod n'.
// Same group elements, hence byte-identical proofs; 2 k (n + 1) MSM terms per proof in 2 k batched MSMs instead of
// ~4 n scalar multiplications, and all proofs of a batch advance round by round together.
//
// NOT constant time (table lookups and bucket sorts indexed by the secret a, b): like the reference's own create(),
// which calls vartime_multiscalar_mul (ipp.rs:87, 101).  It is no replacement for the constant-time commitments of
// the range-proof parties (party.rs:119-124).
#ifndef BPGPU_IPP_PROVER_H
#define BPGPU_IPP_PROVER_H
#include "rangeproof.h"

namespace bp {

struct ippc_shape {
    uint32_t n, k;            // n = 2^k
    uint32_t nproofs;
    uint32_t bases_shared;    // != 0: G, H hold n encodings used by every proof
};

BP_HD void ippc_ld(sc &r, const uint32_t *p) {
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = p[i];
}
BP_HD void ippc_st(uint32_t *p, const sc &r) {
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = r.v[i];
}

// lane = (proof p, index t): load a, b, G_factors, H_factors (must be canonical: status 2 otherwise)
BP_HD void ippc_init_thread(uint32_t tid, ippc_shape sh, const uint8_t *a_in, const uint8_t *b_in, const uint8_t *Gf, const uint8_t *Hf,
                            uint32_t *a, uint32_t *b, uint32_t *wG, uint32_t *wH, uint32_t *status) {
    const uint32_t p = tid / sh.n;
    const uint64_t o = (uint64_t)tid * 32;
    sc x;
    bool ok = true;
    load_words8(x.v, a_in + o);   ok = ok && sc_is_canonical_sc(x);   ippc_st(a + 8 * (uint64_t)tid, x);
    load_words8(x.v, b_in + o);   ok = ok && sc_is_canonical_sc(x);   ippc_st(b + 8 * (uint64_t)tid, x);
    load_words8(x.v, Gf + o);     ok = ok && sc_is_canonical_sc(x);   ippc_st(wG + 8 * (uint64_t)tid, x);
    load_words8(x.v, Hf + o);     ok = ok && sc_is_canonical_sc(x);   ippc_st(wH + 8 * (uint64_t)tid, x);
    if (!ok) status_raise(status + p, BP_STATUS_BAD_SCALAR);
}

// lane = (proof p, index t), round j: the G_t and H_t terms of L (MSM 2p) and R (MSM 2p + 1); each MSM has n + 1 terms:
// [0, n/2) G terms, [n/2, n) H terms, [n] the Q term (ippc_q_thread)
BP_HD void ippc_terms_thread(uint32_t tid, ippc_shape sh, uint32_t j, const uint32_t *a, const uint32_t *b, const uint32_t *wG, const uint32_t *wH,
                             const uint8_t *G, const uint8_t *H, uint32_t *msm_sc, uint32_t *msm_pt) {
    const uint32_t n = sh.n, p = tid / n, t = tid - p * n;
    const uint32_t nj = n >> j, np = nj >> 1;
    const uint32_t tt = t & (nj - 1), i = tt & (np - 1), hi = tt >= np ? 1u : 0u, pos = (t / nj) * np + i;
    const uint32_t N = n + 1;
    const uint64_t pa = (uint64_t)p * n;
    sc x, w, r;
    uint32_t pw[8];
    // G_t: a_L[i] on L when t lies in the right half (hi), a_R[i] on R when in the left half
    ippc_ld(x, a + 8 * (pa + (hi ? i : i + np)));
    ippc_ld(w, wG + 8 * (pa + t));
    sc_mul(r, x, w);
    {
        const uint64_t slot = ((uint64_t)(2 * p + (hi ? 0 : 1)) * N + pos) * 8;
        ippc_st(msm_sc + slot, r);
        load_words8(pw, G + ((sh.bases_shared ? 0 : pa) + t) * 32);
        for (int q = 0; q < 8; q++) msm_pt[slot + q] = pw[q];
    }
    // H_t: b_R[i] on L when t lies in the left half, b_L[i] on R when in the right half
    ippc_ld(x, b + 8 * (pa + (hi ? i : i + np)));
    ippc_ld(w, wH + 8 * (pa + t));
    sc_mul(r, x, w);
    {
        const uint64_t slot = ((uint64_t)(2 * p + (hi ? 1 : 0)) * N + n / 2 + pos) * 8;
        ippc_st(msm_sc + slot, r);
        load_words8(pw, H + ((sh.bases_shared ? 0 : pa) + t) * 32);
        for (int q = 0; q < 8; q++) msm_pt[slot + q] = pw[q];
    }
}

// lane = proof, round j: c_L = <a_L, b_R>, c_R = <a_R, b_L> (ipp.rs:84-85) on Q
BP_HD void ippc_q_thread(uint32_t p, ippc_shape sh, uint32_t j, const uint32_t *a, const uint32_t *b, const uint8_t *Q, uint32_t *msm_sc, uint32_t *msm_pt) {
    const uint32_t n = sh.n, np = (n >> j) >> 1, N = n + 1;
    const uint64_t pa = (uint64_t)p * n;
    sc28 x, y, t;
    sc c, s0, s1;
    sc_0(s0);
    sc_0(s1);
    for (uint32_t i = 0; i < np; i++) {
        ippc_ld(c, a + 8 * (pa + i));
        sc_to_mont28(x, c);
        ippc_ld(c, b + 8 * (pa + i + np));
        sc_to_mont28(y, c);
        sc28_montmul(t, x, y);
        sc_from_mont28(c, t);
        sc_add(s0, s0, c);
        ippc_ld(c, a + 8 * (pa + i + np));
        sc_to_mont28(x, c);
        ippc_ld(c, b + 8 * (pa + i));
        sc_to_mont28(y, c);
        sc28_montmul(t, x, y);
        sc_from_mont28(c, t);
        sc_add(s1, s1, c);
    }
    uint32_t qw[8];
    load_words8(qw, Q + (uint64_t)p * 32);
    const uint64_t sl = ((uint64_t)(2 * p) * N + n) * 8, sr = ((uint64_t)(2 * p + 1) * N + n) * 8;
    ippc_st(msm_sc + sl, s0);
    ippc_st(msm_sc + sr, s1);
    for (int q = 0; q < 8; q++) {
        msm_pt[sl + q] = qw[q];
        msm_pt[sr + q] = qw[q];
    }
}

// ---- the same round when G, H are the context's generators G(n, m), H(n, m) and Q = w B (the range-proof prover,
// dealer.rs:279-293): L_j and R_j are pure generator-table MSMs.  Row 2p (L) and 2p + 1 (R) of a [2 nproofs][2 nm + 2]
// scalar array in the table order (B_blinding, B, G.., H..): every slot is written (zero where a generator does not
// occur), the Q term becomes (c w) on B.
BP_HD void ippc_terms_fixed_thread(uint32_t tid, ippc_shape sh, uint32_t j, const uint32_t *a, const uint32_t *b, const uint32_t *wG, const uint32_t *wH,
                                   uint32_t *gen_scalars) {
    const uint32_t n = sh.n, p = tid / n, t = tid - p * n;
    const uint32_t nj = n >> j, np = nj >> 1;
    const uint32_t tt = t & (nj - 1), i = tt & (np - 1), hi = tt >= np ? 1u : 0u;
    const uint32_t row_len = 2 * n + 2;
    const uint64_t pa = (uint64_t)p * n;
    uint32_t *rowL = gen_scalars + (uint64_t)(2 * p) * row_len * 8, *rowR = rowL + (uint64_t)row_len * 8;
    sc x, w, r, zero;
    sc_0(zero);
    ippc_ld(x, a + 8 * (pa + (hi ? i : i + np)));
    ippc_ld(w, wG + 8 * (pa + t));
    sc_mul(r, x, w);
    ippc_st(rowL + (uint64_t)(2 + t) * 8, hi ? r : zero);          // a_L[i] wG(t) G_t on L when t is in the right half
    ippc_st(rowR + (uint64_t)(2 + t) * 8, hi ? zero : r);          // a_R[i] wG(t) G_t on R when in the left half
    ippc_ld(x, b + 8 * (pa + (hi ? i : i + np)));
    ippc_ld(w, wH + 8 * (pa + t));
    sc_mul(r, x, w);
    ippc_st(rowL + (uint64_t)(2 + n + t) * 8, hi ? zero : r);      // b_R[i] wH(t) H_t on L when in the left half
    ippc_st(rowR + (uint64_t)(2 + n + t) * 8, hi ? r : zero);      // b_L[i] wH(t) H_t on R when in the right half
}
// lane = proof: (c_L w) and (c_R w) on B, zero on B_blinding
BP_HD void ippc_q_fixed_thread(uint32_t p, ippc_shape sh, uint32_t j, const uint32_t *a, const uint32_t *b, const uint32_t *w_all, uint32_t *gen_scalars) {
    const uint32_t n = sh.n, np = (n >> j) >> 1, row_len = 2 * n + 2;
    const uint64_t pa = (uint64_t)p * n;
    sc c, s0, s1, x, y, wq, zero;
    sc_0(s0);
    sc_0(s1);
    sc_0(zero);
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
    ippc_ld(wq, w_all + 8 * (uint64_t)p);
    sc_mul(s0, s0, wq);
    sc_mul(s1, s1, wq);
    uint32_t *rowL = gen_scalars + (uint64_t)(2 * p) * row_len * 8, *rowR = rowL + (uint64_t)row_len * 8;
    ippc_st(rowL, zero);
    ippc_st(rowR, zero);
    ippc_st(rowL + 8, s0);
    ippc_st(rowR + 8, s1);
}

// lane = proof, after round j's MSMs: L, R -> proof bytes and transcript (ipp.rs:115-119), u and u^-1
// ts: the proof's transcript state (BP_TS_WORDS words), read and written back
BP_HD void ippc_challenge_thread(uint32_t p, ippc_shape sh, uint32_t j, kstate st, const uint32_t *msm_out /*[2 nproofs][8]*/, const uint8_t *msm_status,
                                 uint32_t *ts, uint32_t *u_out, uint32_t *uinv_out, uint8_t *proofs, uint32_t proof_len, uint32_t *status) {
    uint32_t *tw = ts + (uint64_t)p * BP_TS_WORDS;
    strobe t;
    t.st = st;
    for (uint32_t i = 0; i < 50; i++) ks_set32(st, i, tw[i]);
    t.pos = tw[50] & 0xffu;
    t.pos_begin = (tw[50] >> 8) & 0xffu;
    t.cur_flags = (tw[50] >> 16) & 0xffu;
    const uint8_t lL[1] = {'L'}, lR[1] = {'R'}, lu[1] = {'u'};
    uint32_t w[8];
    uint32_t *dst = (uint32_t *)(proofs + (uint64_t)p * proof_len + 64 * j);
    if (msm_status[2 * p] | msm_status[2 * p + 1]) status_raise(status + p, msm_status[2 * p] > msm_status[2 * p + 1] ? msm_status[2 * p] : msm_status[2 * p + 1]);
    for (int q = 0; q < 8; q++) w[q] = msm_out[8 * (uint64_t)(2 * p) + q];
    for (int q = 0; q < 8; q++) dst[q] = w[q];
    merlin_append_words8(t, lL, 1, w);
    for (int q = 0; q < 8; q++) w[q] = msm_out[8 * (uint64_t)(2 * p + 1) + q];
    for (int q = 0; q < 8; q++) dst[8 + q] = w[q];
    merlin_append_words8(t, lR, 1, w);
    sc u, ui;
    rp_challenge_scalar(t, lu, 1, u);
    sc_invert_safegcd(ui, u);
    ippc_st(u_out + 8 * (uint64_t)p, u);
    ippc_st(uinv_out + 8 * (uint64_t)p, ui);
    for (uint32_t i = 0; i < 50; i++) tw[i] = ks_get32(st, i);
    tw[50] = rp_ts_meta(t.pos, t.pos_begin, t.cur_flags);
    tw[51] = 0;
}

// lane = (proof, t), round j: fold a, b (ipp.rs:121-125) and advance the generator weights
BP_HD void ippc_fold_thread(uint32_t tid, ippc_shape sh, uint32_t j, const uint32_t *u_all, const uint32_t *uinv_all, uint32_t *a, uint32_t *b,
                            uint32_t *wG, uint32_t *wH) {
    const uint32_t n = sh.n, p = tid / n, t = tid - p * n;
    const uint32_t nj = n >> j, np = nj >> 1;
    const uint64_t pa = (uint64_t)p * n;
    sc u, ui, x, y, r, s;
    ippc_ld(u, u_all + 8 * (uint64_t)p);
    ippc_ld(ui, uinv_all + 8 * (uint64_t)p);
    const bool hi = (t & (nj - 1)) >= np;
    ippc_ld(x, wG + 8 * (pa + t));
    sc_mul(r, x, hi ? u : ui);
    ippc_st(wG + 8 * (pa + t), r);
    ippc_ld(x, wH + 8 * (pa + t));
    sc_mul(r, x, hi ? ui : u);
    ippc_st(wH + 8 * (pa + t), r);
    if (t < np) {
        ippc_ld(x, a + 8 * (pa + t));
        ippc_ld(y, a + 8 * (pa + t + np));
        sc_mul(r, x, u);
        sc_mul(s, ui, y);
        sc_add(r, r, s);
        ippc_st(a + 8 * (pa + t), r);          // a_L[i] * u + u_inv * a_R[i]
        ippc_ld(x, b + 8 * (pa + t));
        ippc_ld(y, b + 8 * (pa + t + np));
        sc_mul(r, x, ui);
        sc_mul(s, u, y);
        sc_add(r, r, s);
        ippc_st(b + 8 * (pa + t), r);          // b_L[i] * u_inv + u * b_R[i]
    }
}

// lane = proof: the final a, b (ipp.rs:186-192) behind the k (L, R) pairs
BP_HD void ippc_final_thread(uint32_t p, ippc_shape sh, const uint32_t *a, const uint32_t *b, uint8_t *proofs, uint32_t proof_len) {
    uint32_t *dst = (uint32_t *)(proofs + (uint64_t)p * proof_len + 64 * sh.k);
    const uint64_t pa = (uint64_t)p * sh.n;
    for (int q = 0; q < 8; q++) {
        dst[q] = a[8 * pa + q];
        dst[8 + q] = b[8 * pa + q];
    }
}

}  // namespace bp
#endif
