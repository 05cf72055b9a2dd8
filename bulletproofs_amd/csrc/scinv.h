// Modular inversion mod l by Bernstein-Yang "safegcd" division steps (half-delta variant), 30 steps at a
// time on the low words with a 2x2 transition matrix applied to the full 256-bit values -- the algorithm
// of eprint 2019/266, organised as in the well-known 32-bit "modinv" formulation.
//
// Why: Scalar::invert / batch_invert (curve25519-dalek; reached from mod.rs:414 and ipp.rs:226) is an
// l-2 exponentiation, ~316 Montgomery multiplications of 10x10 v_mad_u64_u32 each (~73 k instructions per
// proof, 70 % of the per-proof scalar stage).  600 division steps + 20 matrix applications cost ~18 k
// instructions and give the same canonical result, so the stage's latency drops by more than half.
//
// The step count and every branch are independent of the input (all lanes of a wavefront stay in
// lockstep; there is nothing secret here, the fixed structure is for SIMT efficiency).
// 600 = 20 x 30 >= 590 half-delta steps, the proven bound for inputs below 2^256.
#ifndef BPGPU_SCINV_H
#define BPGPU_SCINV_H
#include "sc25519.h"

namespace bp {

// signed value = sum v[i] 2^(30 i); limbs 0..7 in [0, 2^30) after an update, limb 8 carries the sign
struct s30 {
    int32_t v[9];
};
struct inv_mat {
    int32_t u, v, q, r;   // 2^30 * (f', g') = [[u, v], [q, r]] (f, g)
};
#define BP_INV_M30 0x3fffffff

// 30 half-delta division steps on the low 30 bits of (f, g); zeta = -(delta + 1/2)
BP_HD int32_t inv_divsteps30(int32_t zeta, uint32_t f, uint32_t g, inv_mat &t) {
    uint32_t u = 1, v = 0, q = 0, r = 1;
#pragma unroll
    for (int i = 0; i < 30; i++) {
        uint32_t c1 = (uint32_t)(zeta >> 31);   // all ones when delta > 0
        const uint32_t c2 = 0u - (g & 1u);      // all ones when g is odd
        // conditionally negated (f, u, v) is added to (g, q, r) when g is odd
        const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;
        g += x & c2;
        q += y & c2;
        r += z & c2;
        c1 &= c2;                               // swap case: delta > 0 and g odd
        zeta = (zeta ^ (int32_t)c1) - 1;
        f += g & c1;
        u += q & c1;
        v += r & c1;
        g >>= 1;
        u <<= 1;
        v <<= 1;
    }
    t.u = (int32_t)u;
    t.v = (int32_t)v;
    t.q = (int32_t)q;
    t.r = (int32_t)r;
    return zeta;
}

// (f, g) <- t (f, g) / 2^30 (exact)
BP_HD void inv_update_fg(s30 &f, s30 &g, const inv_mat &t) {
    int64_t cf = (int64_t)t.u * f.v[0] + (int64_t)t.v * g.v[0];
    int64_t cg = (int64_t)t.q * f.v[0] + (int64_t)t.r * g.v[0];
    cf >>= 30;
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        cf += (int64_t)t.u * f.v[i] + (int64_t)t.v * g.v[i];
        cg += (int64_t)t.q * f.v[i] + (int64_t)t.r * g.v[i];
        f.v[i - 1] = (int32_t)((uint32_t)cf & BP_INV_M30);
        g.v[i - 1] = (int32_t)((uint32_t)cg & BP_INV_M30);
        cf >>= 30;
        cg >>= 30;
    }
    f.v[8] = (int32_t)cf;
    g.v[8] = (int32_t)cg;
}

// l in 30-bit limbs, and l^-1 mod 2^30
#define BP_INV_L30 {0x1cf5d3ed, 0x20498c69, 0x2f79cd65, 0x37be77a8, 0x14, 0, 0, 0, 0x1000}
#define BP_INV_LINV30 0x2dab81e5u

// (d, e) <- t (d, e) / 2^30 mod l; keeps d, e in (-2l, l)
BP_HD void inv_update_de(s30 &d, s30 &e, const inv_mat &t) {
    const int32_t L[9] = BP_INV_L30;
    const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
    // start with a multiple of l that brings negative inputs back into range
    int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
    int64_t cd = (int64_t)t.u * d.v[0] + (int64_t)t.v * e.v[0];
    int64_t ce = (int64_t)t.q * d.v[0] + (int64_t)t.r * e.v[0];
    // and the multiple of l that clears the low 30 bits
    md -= (int32_t)((BP_INV_LINV30 * (uint32_t)cd + (uint32_t)md) & BP_INV_M30);
    me -= (int32_t)((BP_INV_LINV30 * (uint32_t)ce + (uint32_t)me) & BP_INV_M30);
    cd += (int64_t)L[0] * md;
    ce += (int64_t)L[0] * me;
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        cd += (int64_t)t.u * d.v[i] + (int64_t)t.v * e.v[i];
        ce += (int64_t)t.q * d.v[i] + (int64_t)t.r * e.v[i];
        if (L[i] != 0) {   // static after unrolling: limbs 5..7 of l are zero
            cd += (int64_t)L[i] * md;
            ce += (int64_t)L[i] * me;
        }
        d.v[i - 1] = (int32_t)((uint32_t)cd & BP_INV_M30);
        e.v[i - 1] = (int32_t)((uint32_t)ce & BP_INV_M30);
        cd >>= 30;
        ce >>= 30;
    }
    d.v[8] = (int32_t)cd;
    e.v[8] = (int32_t)ce;
}

// r = +-d mod l in canonical words: d in (-2l, l), negate when neg != 0
BP_HD void inv_normalize(sc &r, const s30 &d, int32_t neg_mask) {
    const int32_t L[9] = BP_INV_L30;
    int32_t t[9];
    // add l when negative, conditionally negate, propagate, add l again when negative
    const int32_t add1 = d.v[8] >> 31;
    int32_t carry = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        int32_t x = d.v[i] + (L[i] & add1);
        x = (x ^ neg_mask) - neg_mask;
        x += carry;
        if (i < 8) {
            carry = x >> 30;
            x &= BP_INV_M30;
        }
        t[i] = x;
    }
    const int32_t add2 = t[8] >> 31;
    carry = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        int32_t x = t[i] + (L[i] & add2) + carry;
        if (i < 8) {
            carry = x >> 30;
            x &= BP_INV_M30;
        }
        t[i] = x;
    }
    // 9 x 30 bits -> 8 x 32 bits (value < l < 2^253)
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const int lo = (32 * w) / 30, sh = (32 * w) % 30;
        uint32_t x = (uint32_t)t[lo] >> sh;
        x |= (uint32_t)t[lo + 1] << (30 - sh);
        if (30 - sh + 30 < 32 && lo + 2 < 9) x |= (uint32_t)t[lo + 2] << (60 - sh);
        r.v[w] = x;
    }
}

// r = a^-1 mod l (canonical in, canonical out; 0 -> 0 like the exponentiation)
BP_HD void sc_invert_safegcd(sc &r, const sc &a) {
    const int32_t L[9] = BP_INV_L30;
    s30 f, g, d, e;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        f.v[i] = L[i];
        d.v[i] = 0;
        e.v[i] = (i == 0);
        const int lo = (30 * i) >> 5, sh = (30 * i) & 31;
        uint32_t x = a.v[lo] >> sh;
        if (sh > 2 && lo + 1 < 8) x |= a.v[lo + 1] << (32 - sh);
        g.v[i] = (int32_t)(x & BP_INV_M30);
    }
    int32_t zeta = -1;
    for (int it = 0; it < 20; it++) {
        inv_mat t;
        zeta = inv_divsteps30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
        inv_update_de(d, e, t);
        inv_update_fg(f, g, t);
    }
    // g == 0, f == +-gcd == +-1 (or +-l when a == 0, where d == 0): a^-1 = d * f
    inv_normalize(r, d, f.v[8] >> 31);
}

// ---- the variable-time form (round 6; narrow chains, where the inversion is ~60 us of a one-proof chain's critical path) ----------------
// The same transition matrices and updates, but the 30 division steps of a batch are not taken one by one: runs of zero bits of g are
// shifted out at once (count trailing zeros), and when g is odd up to six of its low bits are cancelled by ONE multiple of f --
// w = -g f^-1 mod 2^b with f^-1 = f (2 - f^2) mod 64 (one Newton step from f^-1 = f mod 8) -- as in the well-known variable-time
// "modinv" formulation (classic delta = 1 division steps, eta = -delta).  ~7 trips of ~20 instructions per batch instead of 30 x ~20,
// and the outer loop ends when g is 0 (~18 batches on 253-bit inputs instead of a fixed 20).  Lanes of a wavefront that run it together
// diverge only inside a batch.  Nothing here is secret (a verifier's challenges); the result is the same canonical inverse.
BP_HD int32_t inv_divsteps30_var(int32_t eta, uint32_t f, uint32_t g, inv_mat &t) {
    uint32_t u = 1, v = 0, q = 0, r = 1;
    int i = 30;
    for (;;) {
        const int zeros = __builtin_ctz(g | (0xffffffffu << i));   // sentinel: never counts past the i steps that are left
        g >>= zeros;
        u <<= zeros;
        v <<= zeros;
        eta -= zeros;
        i -= zeros;
        if (i == 0) break;
        if (eta < 0) {   // delta > 0 and g odd: (f, g) <- (g, -f)
            uint32_t tmp;
            eta = -eta;
            tmp = f; f = g; g = 0u - tmp;
            tmp = u; u = q; q = 0u - tmp;
            tmp = v; v = r; r = 0u - tmp;
        }
        // cancel the low min(eta + 1, i, 6) bits of g: no more than i (the batch ends there), no more than eta + 1 (its sign flips there)
        const int limit = (eta + 1) > i ? i : (eta + 1);
        const uint32_t m = (0xffffffffu >> (32 - limit)) & 63u;
        const uint32_t w = (f * g * (f * f - 2u)) & m;
        g += f * w;
        q += u * w;
        r += v * w;
    }
    t.u = (int32_t)u;
    t.v = (int32_t)v;
    t.q = (int32_t)q;
    t.r = (int32_t)r;
    return eta;
}
// r = a^-1 mod l (canonical in, canonical out; 0 -> 0), variable time
BP_HD void sc_invert_safegcd_var(sc &r, const sc &a) {
    const int32_t L[9] = BP_INV_L30;
    s30 f, g, d, e;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        f.v[i] = L[i];
        d.v[i] = 0;
        e.v[i] = (i == 0);
        const int lo = (30 * i) >> 5, sh = (30 * i) & 31;
        uint32_t x = a.v[lo] >> sh;
        if (sh > 2 && lo + 1 < 8) x |= a.v[lo + 1] << (32 - sh);
        g.v[i] = (int32_t)(x & BP_INV_M30);
    }
    int32_t eta = -1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int it = 0; it < 26; it++) {   // (25 batches cover the proven 735-step bound of classic division steps on 256-bit inputs)
        inv_mat t;
        eta = inv_divsteps30_var(eta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
        inv_update_de(d, e, t);
        inv_update_fg(f, g, t);
        int32_t nz = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) nz |= g.v[i];
        if (nz == 0) break;
    }
    inv_normalize(r, d, f.v[8] >> 31);
}

// Montgomery form in and out (drop-in for sc28_invert_mont)
BP_HD void sc28_invert_mont_safegcd(sc28 &r, const sc28 &am) {
    sc a, ai;
    sc_from_mont28(a, am);
    sc_invert_safegcd(ai, a);
    sc_to_mont28(r, ai);
}

}  // namespace bp
#endif
