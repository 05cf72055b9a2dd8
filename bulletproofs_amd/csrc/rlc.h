// Batch combination ("random linear combination") of range-proof verifications -- SURVEY 8f-3, an ADDITIONAL
// entry point next to the per-proof one (the reference has no such call; verify_multiple checks one proof).
//
//   R = sum_i rho_i * MegaCheck_i ,   MegaCheck_i = the multiscalar multiplication of mod.rs:421-443 for proof i
//     = sum_g (sum_i rho_i s_{i,g}) P_g  +  sum_i sum_u (rho_i t_{i,u}) Q_{i,u}
//
// The 2nm+2 generator coefficients of all proofs add up in the scalar field, so the table walk runs ONCE per
// batch; only the 4+2k+m proof-specific points still cost point arithmetic per proof.  R is the identity when
// every combined proof verifies; if one does not, R != identity except with probability ~2^-252 over the
// weights (independent uniform scalars rho_i the prover never sees).  A failing batch is re-verified proof by
// proof by the caller / the host entry point.
//
// This header: accumulation of weighted coefficients over the batch.  Per generator the sum is kept as ten
// 64-bit limbs of 28 bits (lanes add their canonical values limb-wise, a wavefront combines its 64 lanes, one
// atomic add per limb per wavefront); rlc_acc_to_sc reduces it mod l.
#ifndef BPGPU_RLC_H
#define BPGPU_RLC_H
#include "sc25519.h"

namespace bp {

#define BP_VERDICT_UNDECIDED 5   // batch combination failed: verify this proof individually

// canonical scalar -> ten 28-bit limbs
BP_HD void rlc_limbs(uint64_t out[10], const sc &s) {
    sc28 t;
    sc28_from_sc(t, s);
#pragma unroll
    for (int i = 0; i < 10; i++) out[i] = t.v[i];
}

// sum of up to 2^24 canonical scalars, as ten 64-bit limb sums -> canonical scalar mod l
// value = low (252 bits) + top * 2^252,  2^252 = -c (mod l)  =>  value = low - top * c (mod l)
BP_HD void rlc_acc_to_sc(sc &r, const uint64_t acc[10]) {
    uint32_t limb[9];
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const uint64_t t = acc[i] + carry;
        limb[i] = (uint32_t)t & BP_M28;
        carry = t >> 28;
    }
    const uint64_t top = acc[9] + carry;                 // < 2^32 for up to 2^24 scalars below 2^253
    // d = low + l - top * c, limb-wise with signed carries (l = 2^252 + c: limbs C0..C4, then 0, 0, 0, 0, and bit 252)
    const uint32_t C[5] = {BP_SC28_C0, BP_SC28_C1, BP_SC28_C2, BP_SC28_C3, BP_SC28_C4};
    int64_t cy = 0;
    uint32_t d[10];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        int64_t t = (int64_t)limb[i] + cy;
        if (i < 5) t += (int64_t)C[i] - (int64_t)(top * C[i]);   // top * C[i] < 2^60
        d[i] = (uint32_t)((uint64_t)t & BP_M28);
        cy = t >> 28;                                    // arithmetic shift: floor
    }
    d[9] = (uint32_t)(1 + cy);                           // bit 252 of l, plus the last carry (0 or -1 ... stays >= 0)
    // words, then at most two conditional subtractions (0 <= d < 2l)
    uint32_t w[8];
    w[0] = d[0] | (d[1] << 28);
    w[1] = (d[1] >> 4) | (d[2] << 24);
    w[2] = (d[2] >> 8) | (d[3] << 20);
    w[3] = (d[3] >> 12) | (d[4] << 16);
    w[4] = (d[4] >> 16) | (d[5] << 12);
    w[5] = (d[5] >> 20) | (d[6] << 8);
    w[6] = (d[6] >> 24) | (d[7] << 4);
    w[7] = d[8] | (d[9] << 28);
    sc_csub_l(w, sc_geq_l(w));
    sc_csub_l(w, sc_geq_l(w));
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = w[i];
}

}  // namespace bp
#endif
