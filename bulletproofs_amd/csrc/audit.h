// The dealer's per-share audit of the multi-party range-proof protocol (device front end):
// ProofShare::audit_share (src/range_proof/messages.rs:85-167), the blame path of Dealer::receive_shares
// (src/range_proof/dealer.rs:303-335).  Lane = share: check t_x == <l, r> (:112-114) and emit the terms of the two
// multiscalar multiplications whose results must be the identity,
//     P_check = A_j + x S_j - e_blinding B~ + sum_i (-z - l_i) G_{j,i} + sum_i (z + y^-i y^-jn (z^2 z^j 2^i - r_i)) H_{j,i}   (:116-141)
//     t_check = (z^2 z^j) V_j + x T_1j + x^2 T_2j + (delta_j - t_x) B - t_x_blinding B~                                  (:143-160)
// as the ragged pair (2n + 3, 5) of bpgpu_msm_batch term lists.  Generators come from the context's loaded set.
#ifndef BPGPU_AUDIT_H
#define BPGPU_AUDIT_H
#include "rangeproof.h"

namespace bp {

struct aud_shape {
    uint32_t n, lg_n;                      // bitsize (a power of two)
    uint32_t nshares;
    uint32_t gens_capacity, party_capacity;
    uint32_t chal_shared;                  // != 0: one (y, z, x) for all shares
};

BP_HD void sc28_pow_u32(sc28 &r, const sc28 &xm, uint32_t e) {
    sc28 b = xm;
    sc28_one_mont(r);
    while (e) {
        if (e & 1) sc28_montmul(r, r, b);
        sc28_montsq(b, b);
        e >>= 1;
    }
}

// thread s.  gens: the context's encodings [B_blinding, B, G (party-major), H (party-major)], 8 words each.
// Term lists: share s owns slots [s (2n + 8), (s + 1)(2n + 8)): first the 2n + 3 terms of P_check (A_j, S_j, B~, G.., H..), then
// the 5 of t_check (V_j, T_1j, T_2j, B, B~).  Outputs are pre-zeroed by the host: a rejected share contributes identity terms.
BP_HD void aud_prepare_thread(uint32_t s, aud_shape sh, const uint32_t *party, const uint8_t *shares, const uint8_t *bit_commitments,
                              const uint8_t *poly_commitments, const uint8_t *challenges, const uint32_t *gens, uint32_t *scalars,
                              uint32_t *points, uint32_t *status) {
    const uint32_t n = sh.n, j = party[s];
    if (j >= sh.party_capacity) {          // check_size (:57-82); n <= gens_capacity is checked on the host
        status[s] = BP_VERDICT_VERIFICATION;
        return;
    }
    const uint8_t *sb = shares + (uint64_t)s * 32 * (3 + 2 * n);
    const uint8_t *ch = challenges + (sh.chal_shared ? 0 : (uint64_t)s * 96);
    sc y, z, x, t_x, t_x_bl, e_bl;
    load_words8(y.v, ch);
    load_words8(z.v, ch + 32);
    load_words8(x.v, ch + 64);
    load_words8(t_x.v, sb);
    load_words8(t_x_bl.v, sb + 32);
    load_words8(e_bl.v, sb + 64);
    bool canon = sc_is_canonical_sc(y) && sc_is_canonical_sc(z) && sc_is_canonical_sc(x) && sc_is_canonical_sc(t_x) && sc_is_canonical_sc(t_x_bl) &&
                 sc_is_canonical_sc(e_bl);
    // t_x == <l, r>
    sc acc, t0;
    sc_0(acc);
    uint32_t w[8];
    for (uint32_t i = 0; i < n; i++) {
        sc l, r;
        load_words8(l.v, sb + 96 + 32 * i);
        load_words8(r.v, sb + 96 + 32 * (n + i));
        canon = canon && sc_is_canonical_sc(l) && sc_is_canonical_sc(r);
        sc_mul(t0, l, r);
        sc_add(acc, acc, t0);
    }
    bool same = true;
    for (int q = 0; q < 8; q++) same = same && acc.v[q] == t_x.v[q];
    if (!canon || !same) {
        status[s] = BP_VERDICT_VERIFICATION;
        return;
    }
    const uint64_t base = (uint64_t)s * (2 * n + 8);
    uint32_t *sc_out = scalars + base * 8, *pt_out = points + base * 8;
    sc28 ym, zm, xm, zzm, zjm, yjnm, yjninv, yinv, one_m;
    sc_to_mont28(ym, y);
    sc_to_mont28(zm, z);
    sc_to_mont28(xm, x);
    sc28_one_mont(one_m);
    sc28_montsq(zzm, zm);
    sc28_pow_u32(zjm, zm, j);                       // z^j
    sc28_pow_u32(yjnm, ym, j * n);                  // y^(j n)
    sc28_invert_mont_safegcd(yjninv, yjnm);
    sc28_invert_mont_safegcd(yinv, ym);
    // P_check: A_j, S_j, B~
    sc one, neg;
    sc_from_u32(one, 1);
    store_words8(sc_out, one);
    load_words8(w, bit_commitments + (uint64_t)s * 96 + 32);
    for (int q = 0; q < 8; q++) pt_out[q] = w[q];
    store_words8(sc_out + 8, x);
    load_words8(w, bit_commitments + (uint64_t)s * 96 + 64);
    for (int q = 0; q < 8; q++) pt_out[8 + q] = w[q];
    sc_neg(neg, e_bl);
    store_words8(sc_out + 16, neg);
    for (int q = 0; q < 8; q++) pt_out[16 + q] = gens[q];
    // g_i = -z - l_i on G_{j,i};  h_i = z + y^-i y^-jn (z^2 z^j 2^i - r_i) on H_{j,i}
    sc mz;
    sc_neg(mz, z);
    sc28 f = yjninv, zzzj, e2;                       // f = y^-i y^-jn, e2 = z^2 z^j 2^i
    sc28_montmul(zzzj, zzm, zjm);
    e2 = zzzj;
    const uint64_t tot = (uint64_t)sh.gens_capacity * sh.party_capacity;
    const uint32_t *Gj = gens + (2 + (uint64_t)j * sh.gens_capacity) * 8, *Hj = gens + (2 + tot + (uint64_t)j * sh.gens_capacity) * 8;
    for (uint32_t i = 0; i < n; i++) {
        sc l, r;
        load_words8(l.v, sb + 96 + 32 * i);
        load_words8(r.v, sb + 96 + 32 * (n + i));
        sc_sub(t0, mz, l);
        store_words8(sc_out + (3 + i) * 8, t0);
        for (int q = 0; q < 8; q++) pt_out[(3 + i) * 8 + q] = Gj[(uint64_t)i * 8 + q];
        sc e2s, d;
        sc_from_mont28(e2s, e2);
        sc_sub(d, e2s, r);                          // z^2 z^j 2^i - r_i
        sc28 dm, hm;
        sc_to_mont28(dm, d);
        sc28_montmul(hm, f, dm);
        sc_from_mont28(t0, hm);
        sc_add(t0, z, t0);
        store_words8(sc_out + (3 + n + i) * 8, t0);
        for (int q = 0; q < 8; q++) pt_out[(3 + n + i) * 8 + q] = Hj[(uint64_t)i * 8 + q];
        sc28_montmul(f, f, yinv);
        sc_from_mont28(e2s, e2);
        sc_add(e2s, e2s, e2s);                      // times 2
        sc_to_mont28(e2, e2s);
    }
    // t_check: V_j, T_1j, T_2j, B, B~
    uint32_t *sc2 = sc_out + (2 * n + 3) * 8, *pt2 = pt_out + (2 * n + 3) * 8;
    sc_from_mont28(t0, zzzj);
    store_words8(sc2, t0);
    load_words8(w, bit_commitments + (uint64_t)s * 96);
    for (int q = 0; q < 8; q++) pt2[q] = w[q];
    store_words8(sc2 + 8, x);
    load_words8(w, poly_commitments + (uint64_t)s * 64);
    for (int q = 0; q < 8; q++) pt2[8 + q] = w[q];
    sc28 xxm;
    sc28_montsq(xxm, xm);
    sc_from_mont28(t0, xxm);
    store_words8(sc2 + 16, t0);
    load_words8(w, poly_commitments + (uint64_t)s * 64 + 32);
    for (int q = 0; q < 8; q++) pt2[16 + q] = w[q];
    // delta_j = (z - z^2) sum_of_powers(y, n) y^(jn) - z z^2 sum_of_powers(2, n) z^j   (:146-148)
    sc28 sy, s2, two_m, a_m, b_m;
    {
        sc two;
        sc_from_u32(two, 2);
        sc_to_mont28(two_m, two);
    }
    rp_sum_of_powers_pow2(sy, ym, sh.lg_n);
    rp_sum_of_powers_pow2(s2, two_m, sh.lg_n);
    sc zz_s, zmzz;
    sc_from_mont28(zz_s, zzm);
    sc_sub(zmzz, z, zz_s);
    sc_to_mont28(a_m, zmzz);
    sc28_montmul(a_m, a_m, sy);
    sc28_montmul(a_m, a_m, yjnm);
    sc28_montmul(b_m, zm, zzm);
    sc28_montmul(b_m, b_m, s2);
    sc28_montmul(b_m, b_m, zjm);
    sc da, db, delta;
    sc_from_mont28(da, a_m);
    sc_from_mont28(db, b_m);
    sc_sub(delta, da, db);
    sc_sub(t0, delta, t_x);
    store_words8(sc2 + 24, t0);
    for (int q = 0; q < 8; q++) pt2[24 + q] = gens[8 + q];     // B
    sc_neg(t0, t_x_bl);
    store_words8(sc2 + 32, t0);
    for (int q = 0; q < 8; q++) pt2[32 + q] = gens[q];         // B~
}

// thread s: Ok(()) iff the front end accepted the share, every point decoded and both results are the identity
BP_HD void aud_verdict_thread(uint32_t s, const uint32_t *status, const uint8_t *msm_status, const uint32_t *msm_out, uint8_t *verdict) {
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) nz |= msm_out[16 * (uint64_t)s + i];
    verdict[s] = (status[s] != 0 || msm_status[2 * s] != 0 || msm_status[2 * s + 1] != 0 || nz != 0) ? BP_VERDICT_VERIFICATION : BP_VERDICT_OK;
}

}  // namespace bp
#endif
