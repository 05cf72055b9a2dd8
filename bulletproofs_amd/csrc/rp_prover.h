// Batched range-proof CREATION: RangeProof::prove_multiple_with_rng (src/range_proof/mod.rs:234-288) with the dealer /
// party state machine (src/range_proof/party.rs:87-297, dealer.rs:44-320) run in-line for all m parties of a proof, for
// many independent proofs of one shape (n, m) at once.  Every commitment is a multiscalar multiplication over the
// generators B_blinding, B, G(n, m), H(n, m) and goes through the fixed-base tables (bpgpu_msm_batch_shared's engine):
//     V_j = v_j B + v_blinding_j B~                                   (party.rs:55, generators.rs:39-41)
//     A   = sum_j [ a_blinding_j B~ + sum_i (a_L G - (1 - a_L) H) ]    (party.rs:99-116 summed by the dealer, dealer.rs:117-118)
//     S   = sum_j [ s_blinding_j B~ + <s_L, G> + <s_R, H> ]            (party.rs:119-124, dealer.rs:117-118)
//     T_i = (sum_j t_i,j) B + (sum_j t_i_blinding_j) B~ , i = 1, 2     (party.rs:179-187, dealer.rs:176-177)
// and the inner-product argument is the batched prover of ipp_prover.h over G(n, m), H(n, m) with H_factors y^-i
// (dealer.rs:281-293).  Summing the parties' commitments in one multiscalar multiplication yields the same group
// elements as the dealer's additions, so the proofs are byte-identical to the reference algorithm's given the same
// random scalars -- the caller supplies them as bytes, in the order the reference draws them:
//     for each party j: a_blinding, s_blinding, s_L[0..n), s_R[0..n)   (party.rs:94-97, 119-121), then
//     for each party j: t_1_blinding, t_2_blinding                     (party.rs:174-175),   64 bytes each (Scalar::random).
//
// Timing: by default the table lookups are indexed by the digits of the secrets (values, blindings, s_L, s_R).  The reference
// computes V, A, S, T_1, T_2 with its constant-time multiscalar_mul (party.rs:99-124, 179-187); the context option
// "prover_constant_time" routes exactly those MSMs through fb_accum_ct_thread (msm_fixed.h): digit-independent addresses and
// instruction stream, byte-identical results.
#ifndef BPGPU_RP_PROVER_H
#define BPGPU_RP_PROVER_H
#include "ipp_prover.h"

namespace bp {

struct rpp_shape {
    uint32_t n, m, nm, k;      // k = lg(nm)
    uint32_t nproofs;
    uint32_t proof_len;        // 32 * (9 + 2k)
    uint32_t n_gen_terms;      // 2 nm + 2: B~, B, G(n,m), H(n,m) -- the row length of the MSM scalar arrays
    uint32_t rng_per_proof;    // bytes: 64 * (m (2n + 2) + 2m)
};
// per-proof scalar store, [field][proof][8 words]
enum { RPP_Y = 0, RPP_Z, RPP_X, RPP_W, RPP_YINV, RPP_FIXED };
// per-(proof, party) scalar store, [field][proof * m + j][8 words]
enum { RPP_VBL = 0, RPP_ABL, RPP_SBL, RPP_T1B, RPP_T2B, RPP_OZZ, RPP_T0, RPP_T1, RPP_T2, RPP_PARTY_FIELDS };

BP_HD void rpp_wide(sc &r, const uint8_t *p64) {
    uint32_t w[16];
    load_words8(w, p64);
    load_words8(w + 8, p64 + 32);
    sc_from_wide(r, w);
}
BP_HD const uint8_t *rpp_rng_party(const rpp_shape &sh, const uint8_t *rng, uint32_t p, uint32_t j) {
    return rng + (uint64_t)p * sh.rng_per_proof + (uint64_t)j * 64 * (2 * sh.n + 2);
}
BP_HD uint32_t *rpp_row(uint32_t *gen_scalars, const rpp_shape &sh, uint32_t msm, uint32_t slot) {
    return gen_scalars + ((uint64_t)msm * sh.n_gen_terms + slot) * 8;
}

// lane = (proof p, party j, bit i): the bit's terms of A (MSM row 2p) and S (row 2p + 1); s_L, s_R kept for the polynomials
BP_HD void rpp_bits_thread(uint32_t tid, rpp_shape sh, const uint64_t *values, const uint8_t *rng, uint32_t *gen_scalars, uint32_t *sL, uint32_t *sR) {
    const uint32_t p = tid / sh.nm, q = tid - p * sh.nm, j = q / sh.n, i = q - j * sh.n;
    const uint64_t v = values[(uint64_t)p * sh.m + j];
    const bool bit = (v >> i) & 1;
    sc one, m1, zero, s;
    sc_from_u32(one, 1);
    sc_0(zero);
    sc_neg(m1, one);
    ippc_st(rpp_row(gen_scalars, sh, 2 * p, 2 + q), bit ? one : zero);            // a_L on G
    ippc_st(rpp_row(gen_scalars, sh, 2 * p, 2 + sh.nm + q), bit ? zero : m1);     // a_R = a_L - 1 on H
    const uint8_t *r = rpp_rng_party(sh, rng, p, j) + 128;
    rpp_wide(s, r + 64 * (uint64_t)i);
    ippc_st(rpp_row(gen_scalars, sh, 2 * p + 1, 2 + q), s);
    ippc_st(sL + 8 * (uint64_t)tid, s);
    rpp_wide(s, r + 64 * (uint64_t)(sh.n + i));
    ippc_st(rpp_row(gen_scalars, sh, 2 * p + 1, 2 + sh.nm + q), s);
    ippc_st(sR + 8 * (uint64_t)tid, s);
}

// lane = proof: the blinding terms of A and S (summed over the parties) and the V_j rows (MSM 2 nproofs + p m + j)
BP_HD void rpp_blind_thread(uint32_t p, rpp_shape sh, const uint64_t *values, const uint8_t *blindings, const uint8_t *rng, uint32_t *gen_scalars,
                            uint32_t *party) {
    const uint64_t PB = (uint64_t)sh.nproofs * sh.m;
    sc sa, ss, x;
    sc_0(sa);
    sc_0(ss);
    for (uint32_t j = 0; j < sh.m; j++) {
        const uint64_t pj = (uint64_t)p * sh.m + j;
        const uint8_t *r = rpp_rng_party(sh, rng, p, j);
        rpp_wide(x, r);
        ippc_st(party + ((uint64_t)RPP_ABL * PB + pj) * 8, x);
        sc_add(sa, sa, x);
        rpp_wide(x, r + 64);
        ippc_st(party + ((uint64_t)RPP_SBL * PB + pj) * 8, x);
        sc_add(ss, ss, x);
        uint32_t w[16];
        load_words8(w, blindings + pj * 32);
        for (int q = 8; q < 16; q++) w[q] = 0;
        sc_from_wide(x, w);                                   // Scalar given by the caller, reduced mod l
        ippc_st(party + ((uint64_t)RPP_VBL * PB + pj) * 8, x);
        const uint32_t row = 2 * sh.nproofs + (uint32_t)pj;
        ippc_st(rpp_row(gen_scalars, sh, row, 0), x);          // v_blinding on B_blinding
        sc vs;
        sc_0(vs);
        vs.v[0] = (uint32_t)values[pj];
        vs.v[1] = (uint32_t)(values[pj] >> 32);
        ippc_st(rpp_row(gen_scalars, sh, row, 1), vs);         // v on B
    }
    ippc_st(rpp_row(gen_scalars, sh, 2 * p, 0), sa);
    ippc_st(rpp_row(gen_scalars, sh, 2 * p + 1, 0), ss);
}

BP_HD void rpp_ts_load(strobe &t, kstate st, const uint32_t *tw) {
    t.st = st;
    for (uint32_t i = 0; i < 50; i++) ks_set32(st, i, tw[i]);
    t.pos = tw[50] & 0xffu;
    t.pos_begin = (tw[50] >> 8) & 0xffu;
    t.cur_flags = (tw[50] >> 16) & 0xffu;
}
BP_HD void rpp_ts_store(uint32_t *tw, const strobe &t) {
    for (uint32_t i = 0; i < 50; i++) tw[i] = ks_get32(t.st, i);
    tw[50] = rp_ts_meta(t.pos, t.pos_begin, t.cur_flags);
    tw[51] = 0;
}
BP_HD void rpp_copy8(uint32_t *dst, const uint32_t *src) {
#pragma unroll
    for (int q = 0; q < 8; q++) dst[q] = src[q];
}

// lane = proof: commitments out; transcript V_j, A, S -> y, z (mod.rs:255-259, dealer.rs:103-125)
// msm_out rows: 2p = A, 2p + 1 = S, 2 nproofs + p m + j = V_j.  The transcript state already holds rangeproof_domain_sep.
BP_HD void rpp_chal1_thread(uint32_t p, rpp_shape sh, kstate st, const uint32_t *msm_out, uint32_t *ts, uint32_t *fields, uint8_t *proofs,
                            uint8_t *commitments) {
    strobe t;
    rpp_ts_load(t, st, ts + (uint64_t)p * BP_TS_WORDS);
    const uint8_t lV[1] = {'V'}, lA[1] = {'A'}, lS[1] = {'S'}, ly[1] = {'y'}, lz[1] = {'z'};
    uint32_t w[8];
    for (uint32_t j = 0; j < sh.m; j++) {
        rpp_copy8(w, msm_out + 8 * (uint64_t)(2 * sh.nproofs + p * sh.m + j));
        rpp_copy8((uint32_t *)(commitments + ((uint64_t)p * sh.m + j) * 32), w);
        merlin_append_words8(t, lV, 1, w);
    }
    uint32_t *pr = (uint32_t *)(proofs + (uint64_t)p * sh.proof_len);
    rpp_copy8(w, msm_out + 8 * (uint64_t)(2 * p));
    rpp_copy8(pr, w);
    merlin_append_words8(t, lA, 1, w);
    rpp_copy8(w, msm_out + 8 * (uint64_t)(2 * p + 1));
    rpp_copy8(pr + 8, w);
    merlin_append_words8(t, lS, 1, w);
    sc y, z;
    rp_challenge_scalar(t, ly, 1, y);
    rp_challenge_scalar(t, lz, 1, z);
    ippc_st(fields + ((uint64_t)RPP_Y * sh.nproofs + p) * 8, y);
    ippc_st(fields + ((uint64_t)RPP_Z * sh.nproofs + p) * 8, z);
    rpp_ts_store(ts + (uint64_t)p * BP_TS_WORDS, t);
}

// lane = (proof, party): l(X) = l0 + l1 X, r(X) = r0 + r1 X and the coefficients of t(X) = <l, r> (party.rs:145-172)
BP_HD void rpp_poly_thread(uint32_t tid, rpp_shape sh, const uint64_t *values, const uint32_t *fields, const uint32_t *sL, const uint32_t *sR,
                           uint32_t *l0, uint32_t *l1, uint32_t *r0, uint32_t *r1, uint32_t *party) {
    const uint32_t p = tid / sh.m, j = tid - p * sh.m;
    const uint64_t PB = (uint64_t)sh.nproofs * sh.m, pj = tid;
    sc y, z, zz, one, exp_y, ozz, exp_2, t0, t1, t2;
    ippc_ld(y, fields + ((uint64_t)RPP_Y * sh.nproofs + p) * 8);
    ippc_ld(z, fields + ((uint64_t)RPP_Z * sh.nproofs + p) * 8);
    sc_mul(zz, z, z);
    sc_from_u32(one, 1);
    // exp_y = y^(j n): (y^n)^j, n a power of two; offset_zz = z^2 z^j
    sc yn = y;
    for (uint32_t b = 1; b < sh.n; b <<= 1) sc_mul(yn, yn, yn);
    exp_y = one;
    ozz = zz;
    for (uint32_t q = 0; q < j; q++) {
        sc_mul(exp_y, exp_y, yn);
        sc_mul(ozz, ozz, z);
    }
    ippc_st(party + ((uint64_t)RPP_OZZ * PB + pj) * 8, ozz);
    exp_2 = one;
    sc_0(t0);
    sc_0(t1);
    sc_0(t2);
    const uint64_t v = values[pj];
    for (uint32_t i = 0; i < sh.n; i++) {
        const uint64_t q = (uint64_t)p * sh.nm + (uint64_t)j * sh.n + i;
        sc aL, aR, a, b, c, d, tt, tu, ls, rs;
        sc_0(aL);
        aL.v[0] = (uint32_t)((v >> i) & 1);
        sc_sub(aR, aL, one);
        sc_sub(a, aL, z);                          // l0
        ippc_ld(b, sL + 8 * q);                    // l1
        sc_add(tt, aR, z);
        sc_mul(tt, exp_y, tt);
        sc_mul(tu, ozz, exp_2);
        sc_add(c, tt, tu);                         // r0
        ippc_ld(d, sR + 8 * q);
        sc_mul(d, exp_y, d);                       // r1
        ippc_st(l0 + 8 * q, a);
        ippc_st(l1 + 8 * q, b);
        ippc_st(r0 + 8 * q, c);
        ippc_st(r1 + 8 * q, d);
        sc_mul(tt, a, c);
        sc_add(t0, t0, tt);                        // <l0, r0>
        sc_mul(tt, b, d);
        sc_add(t2, t2, tt);                        // <l1, r1>
        sc_add(ls, a, b);
        sc_add(rs, c, d);
        sc_mul(tt, ls, rs);
        sc_add(t1, t1, tt);                        // <l0 + l1, r0 + r1>
        sc_mul(exp_y, exp_y, y);
        sc_add(exp_2, exp_2, exp_2);
    }
    sc_sub(t1, t1, t0);
    sc_sub(t1, t1, t2);                            // t1 = <l0+l1, r0+r1> - t0 - t2   (util.rs VecPoly1::inner_product)
    ippc_st(party + ((uint64_t)RPP_T0 * PB + pj) * 8, t0);
    ippc_st(party + ((uint64_t)RPP_T1 * PB + pj) * 8, t1);
    ippc_st(party + ((uint64_t)RPP_T2 * PB + pj) * 8, t2);
}

// lane = proof: T_1, T_2 rows (MSM 2p, 2p + 1): (sum_j t_i,j) on B, (sum_j t_i_blinding_j) on B_blinding
BP_HD void rpp_tcommit_thread(uint32_t p, rpp_shape sh, const uint8_t *rng, uint32_t *gen_scalars, uint32_t *party) {
    const uint64_t PB = (uint64_t)sh.nproofs * sh.m;
    sc s1, s2, b1, b2, x;
    sc_0(s1);
    sc_0(s2);
    sc_0(b1);
    sc_0(b2);
    const uint8_t *r = rng + (uint64_t)p * sh.rng_per_proof + (uint64_t)sh.m * 64 * (2 * sh.n + 2);
    for (uint32_t j = 0; j < sh.m; j++) {
        const uint64_t pj = (uint64_t)p * sh.m + j;
        rpp_wide(x, r + 128 * (uint64_t)j);
        ippc_st(party + ((uint64_t)RPP_T1B * PB + pj) * 8, x);
        sc_add(b1, b1, x);
        rpp_wide(x, r + 128 * (uint64_t)j + 64);
        ippc_st(party + ((uint64_t)RPP_T2B * PB + pj) * 8, x);
        sc_add(b2, b2, x);
        ippc_ld(x, party + ((uint64_t)RPP_T1 * PB + pj) * 8);
        sc_add(s1, s1, x);
        ippc_ld(x, party + ((uint64_t)RPP_T2 * PB + pj) * 8);
        sc_add(s2, s2, x);
    }
    ippc_st(rpp_row(gen_scalars, sh, 2 * p, 0), b1);
    ippc_st(rpp_row(gen_scalars, sh, 2 * p, 1), s1);
    ippc_st(rpp_row(gen_scalars, sh, 2 * p + 1, 0), b2);
    ippc_st(rpp_row(gen_scalars, sh, 2 * p + 1, 1), s2);
}

// lane = proof: T_1, T_2 -> x; t_x, t_x_blinding, e_blinding (party.rs:217-244 summed by the dealer, dealer.rs:262-266) -> w;
// Q = w B as MSM row p; the transcript continues with innerproduct_domain_sep(nm) for the inner-product rounds
BP_HD void rpp_chal2_thread(uint32_t p, rpp_shape sh, kstate st, const uint32_t *msm_out, uint32_t *ts, uint32_t *fields, const uint32_t *party,
                            uint32_t *gen_scalars, uint8_t *proofs) {
    const uint64_t PB = (uint64_t)sh.nproofs * sh.m;
    strobe t;
    rpp_ts_load(t, st, ts + (uint64_t)p * BP_TS_WORDS);
    const uint8_t lT1[3] = {'T', '_', '1'}, lT2[3] = {'T', '_', '2'}, lx[1] = {'x'}, lw[1] = {'w'}, ltx[3] = {'t', '_', 'x'};
    const uint8_t ltxb[12] = {'t', '_', 'x', '_', 'b', 'l', 'i', 'n', 'd', 'i', 'n', 'g'};
    const uint8_t leb[10] = {'e', '_', 'b', 'l', 'i', 'n', 'd', 'i', 'n', 'g'};
    const uint8_t ldom[7] = {'d', 'o', 'm', '-', 's', 'e', 'p'}, lipp[6] = {'i', 'p', 'p', ' ', 'v', '1'}, ln[1] = {'n'};
    uint32_t *pr = (uint32_t *)(proofs + (uint64_t)p * sh.proof_len);
    uint32_t w8[8];
    rpp_copy8(w8, msm_out + 8 * (uint64_t)(2 * p));
    rpp_copy8(pr + 16, w8);
    merlin_append_words8(t, lT1, 3, w8);
    rpp_copy8(w8, msm_out + 8 * (uint64_t)(2 * p + 1));
    rpp_copy8(pr + 24, w8);
    merlin_append_words8(t, lT2, 3, w8);
    sc x, xx, tx, txb, eb, a, b;
    rp_challenge_scalar(t, lx, 1, x);
    sc_mul(xx, x, x);
    sc_0(tx);
    sc_0(txb);
    sc_0(eb);
    for (uint32_t j = 0; j < sh.m; j++) {
        const uint64_t pj = (uint64_t)p * sh.m + j;
        // t_x_j = t0 + x (t1 + x t2)
        ippc_ld(a, party + ((uint64_t)RPP_T2 * PB + pj) * 8);
        sc_mul(a, x, a);
        ippc_ld(b, party + ((uint64_t)RPP_T1 * PB + pj) * 8);
        sc_add(a, a, b);
        sc_mul(a, x, a);
        ippc_ld(b, party + ((uint64_t)RPP_T0 * PB + pj) * 8);
        sc_add(a, a, b);
        sc_add(tx, tx, a);
        // t_x_blinding_j = z^2 z^j v_blinding + x (t1_blinding + x t2_blinding)
        ippc_ld(a, party + ((uint64_t)RPP_T2B * PB + pj) * 8);
        sc_mul(a, x, a);
        ippc_ld(b, party + ((uint64_t)RPP_T1B * PB + pj) * 8);
        sc_add(a, a, b);
        sc_mul(a, x, a);
        sc c, d;
        ippc_ld(c, party + ((uint64_t)RPP_OZZ * PB + pj) * 8);
        ippc_ld(d, party + ((uint64_t)RPP_VBL * PB + pj) * 8);
        sc_mul(c, c, d);
        sc_add(a, a, c);
        sc_add(txb, txb, a);
        // e_blinding_j = a_blinding + x s_blinding
        ippc_ld(a, party + ((uint64_t)RPP_SBL * PB + pj) * 8);
        sc_mul(a, a, x);
        ippc_ld(b, party + ((uint64_t)RPP_ABL * PB + pj) * 8);
        sc_add(a, a, b);
        sc_add(eb, eb, a);
    }
    ippc_st(pr + 32, tx);
    ippc_st(pr + 40, txb);
    ippc_st(pr + 48, eb);
    merlin_append_words8(t, ltx, 3, tx.v);
    merlin_append_words8(t, ltxb, 12, txb.v);
    merlin_append_words8(t, leb, 10, eb.v);
    sc wch, y, yinv;
    rp_challenge_scalar(t, lw, 1, wch);
    ippc_st(rpp_row(gen_scalars, sh, p, 1), wch);          // Q = w B  (dealer.rs:279)
    ippc_st(fields + ((uint64_t)RPP_X * sh.nproofs + p) * 8, x);
    ippc_st(fields + ((uint64_t)RPP_W * sh.nproofs + p) * 8, wch);
    ippc_ld(y, fields + ((uint64_t)RPP_Y * sh.nproofs + p) * 8);
    sc_invert_safegcd(yinv, y);
    ippc_st(fields + ((uint64_t)RPP_YINV * sh.nproofs + p) * 8, yinv);
    merlin_append_message(t, ldom, 7, lipp, 6);            // InnerProductProof::create: innerproduct_domain_sep(nm)
    merlin_append_u64(t, ln, 1, sh.nm);
    rpp_ts_store(ts + (uint64_t)p * BP_TS_WORDS, t);
}

// lane = (proof, index q < nm): the inner-product argument's inputs: a = l(x), b = r(x), G_factor 1, H_factor y^-q (dealer.rs:281-293)
BP_HD void rpp_vectors_thread(uint32_t tid, rpp_shape sh, const uint32_t *fields, const uint32_t *l0, const uint32_t *l1, const uint32_t *r0,
                              const uint32_t *r1, uint32_t *a_vec, uint32_t *b_vec, uint32_t *Gf, uint32_t *Hf) {
    const uint32_t p = tid / sh.nm, q = tid - p * sh.nm;
    sc x, yinv, a, b, r, one;
    ippc_ld(x, fields + ((uint64_t)RPP_X * sh.nproofs + p) * 8);
    ippc_ld(yinv, fields + ((uint64_t)RPP_YINV * sh.nproofs + p) * 8);
    ippc_ld(a, l1 + 8 * (uint64_t)tid);
    sc_mul(a, a, x);
    ippc_ld(b, l0 + 8 * (uint64_t)tid);
    sc_add(a, a, b);
    ippc_st(a_vec + 8 * (uint64_t)tid, a);
    ippc_ld(a, r1 + 8 * (uint64_t)tid);
    sc_mul(a, a, x);
    ippc_ld(b, r0 + 8 * (uint64_t)tid);
    sc_add(a, a, b);
    ippc_st(b_vec + 8 * (uint64_t)tid, a);
    sc_from_u32(one, 1);
    ippc_st(Gf + 8 * (uint64_t)tid, one);
    r = one;                                   // y^-q by square and multiply
    sc base = yinv;
    for (uint32_t e = q; e; e >>= 1) {
        if (e & 1) sc_mul(r, r, base);
        sc_mul(base, base, base);
    }
    ippc_st(Hf + 8 * (uint64_t)tid, r);
}

}  // namespace bp
#endif
