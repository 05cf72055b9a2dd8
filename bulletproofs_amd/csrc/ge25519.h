// Edwards25519 / ristretto255 group operations for one GPU lane.
//
// Replaces, for the verification hot path, what the reference obtains from
// curve25519-dalek: EdwardsPoint add/double (behind
// RistrettoPoint::optional_multiscalar_mul, src/range_proof/mod.rs:421),
// CompressedRistretto::decompress (mod.rs:433-443), compress / is_identity
// (mod.rs:447) and from_uniform_bytes (src/generators.rs:98).  Encodings follow
// RFC 9496; formulas are the complete a=-1 twisted Edwards ones, so no branch
// depends on point values (all lanes of a wavefront execute the same stream).
#ifndef BPGPU_GE25519_H
#define BPGPU_GE25519_H
#include "fe25519.h"

namespace bp {

#define BP_FE_D {{0x35978a3u, 0x0d37284u, 0x3156ebdu, 0x06a0a0eu, 0x001c029u, 0x179e898u, 0x3a03cbbu, 0x1ce7198u, 0x2e2b6ffu, 0x1480db3u}}
#define BP_FE_D2 {{0x2b2f159u, 0x1a6e509u, 0x22add7au, 0x0d4141du, 0x0038052u, 0x0f3d130u, 0x3407977u, 0x19ce331u, 0x1c56dffu, 0x0901b67u}}
#define BP_FE_SQRT_M1 {{0x20ea0b0u, 0x186c9d2u, 0x08f189du, 0x035697fu, 0x0bd0c60u, 0x1fbd7a7u, 0x2804c9eu, 0x1e16569u, 0x004fc1du, 0x0ae0c92u}}
#define BP_FE_INVSQRT_A_MINUS_D {{0x05d40eau, 0x03f6aa0u, 0x257d339u, 0x0bad20bu, 0x274bc58u, 0x001d840u, 0x13dc8ffu, 0x19442d8u, 0x05cfaffu, 0x1e1b224u}}
#define BP_FE_SQRT_AD_MINUS_ONE {{0x17b2e1bu, 0x1fda812u, 0x297afd2u, 0x060dbc2u, 0x2be7638u, 0x1f5d1fdu, 0x27e6498u, 0x11581e7u, 0x3f2b834u, 0x0dda4c6u}}
#define BP_FE_ONE_MINUS_D_SQ {{0x05fc176u, 0x1027065u, 0x2a1fc4fu, 0x1c66af1u, 0x0b20684u, 0x070dfe4u, 0x255eedfu, 0x01af332u, 0x28b2b3eu, 0x00a41cau}}
#define BP_FE_D_MINUS_ONE_SQ {{0x0ed4d20u, 0x156aa91u, 0x3332635u, 0x16580f0u, 0x34a7928u, 0x09b4eebu, 0x26997a9u, 0x048299bu, 0x3af66c2u, 0x165a2cdu}}

// extended coordinates (X:Y:Z:T), x = X/Z, y = Y/Z, xy = T/Z
struct __attribute__((aligned(16))) ge_ext {
    fe X, Y, Z, T;
};
// projective Niels form of a point, the addend format of variable-base tables
struct __attribute__((aligned(16))) ge_cached {
    fe YpX, YmX, Z, T2d;
};
// affine Niels form (Z = 1), the entry format of the fixed-base generator tables
struct ge_niels {
    fe ypx, ymx, t2d;
};

BP_HD void ge_identity(ge_ext &r) {
    fe_0(r.X);
    fe_1(r.Y);
    fe_1(r.Z);
    fe_0(r.T);
}

BP_HD void ge_to_cached(ge_cached &r, const ge_ext &p) {
    const fe d2 = BP_FE_D2;
    fe_add(r.YpX, p.Y, p.X);
    fe_carry(r.YpX);
    fe_sub(r.YmX, p.Y, p.X);
    r.Z = p.Z;
    fe_mul(r.T2d, p.T, d2);
}

// r = p + q (neg: r = p - q).  8M.  All stored operands are reduced.
// -q swaps (Y+X, Y-X) and negates T2d; the negation is folded into which of D-C / D+C plays F and G.
BP_HD void ge_add_cached(ge_ext &r, const ge_ext &p, const ge_cached &q, bool neg) {
    fe ypx, ymx, a, b, c, d, qa, qb;
    fe_select(qa, q.YmX, q.YpX, neg);
    fe_select(qb, q.YpX, q.YmX, neg);
    fe_add(ypx, p.Y, p.X);              // lazy
    fe_sub_rr(ymx, p.Y, p.X);           // lazy (no carry chain: the coordinates of p are reduced)
    fe_mul(a, ymx, qa);
    fe_mul(b, ypx, qb);
    fe_mul(c, p.T, q.T2d);
    fe_mul(d, p.Z, q.Z);
    fe_add(d, d, d);                    // lazy (2x)
    fe e, f, g, h, dmc, dpc;
    fe_sub_rr(e, b, a);                 // lazy
    fe_add(h, b, a);                    // lazy (2x)
    fe_sub(dmc, d, c);
    fe_add(dpc, d, c);                  // lazy (3x)
    fe_select(f, dmc, dpc, neg);
    fe_select(g, dpc, dmc, neg);
    // second operand = the one fe_mul premultiplies by 19: e and g serve two products each (18 instead of 27 multiplies)
    fe_mul(r.X, f, e);
    fe_mul(r.Y, h, g);
    fe_mul(r.Z, f, g);
    fe_mul(r.T, h, e);
}

// the same in two halves, so that a caller can request its next operand between them (the registers of q are free after the
// first half): front = the four products with q and the sums, back = the four products that form the result
struct ge_efgh {
    fe e, f, g, h;
};
BP_HD void ge_add_cached_front(ge_efgh &t, const ge_ext &p, const ge_cached &q, bool neg) {
    fe ypx, ymx, a, b, c, d, qa, qb;
    fe_select(qa, q.YmX, q.YpX, neg);
    fe_select(qb, q.YpX, q.YmX, neg);
    fe_add(ypx, p.Y, p.X);              // lazy
    fe_sub_rr(ymx, p.Y, p.X);           // lazy
    fe_mul(a, ymx, qa);
    fe_mul(b, ypx, qb);
    fe_mul(c, p.T, q.T2d);
    fe_mul(d, p.Z, q.Z);
    fe_add(d, d, d);                    // lazy (2x)
    fe dmc, dpc;
    fe_sub_rr(t.e, b, a);               // lazy
    fe_add(t.h, b, a);                  // lazy (2x)
    fe_sub(dmc, d, c);
    fe_add(dpc, d, c);                  // lazy (3x)
    fe_select(t.f, dmc, dpc, neg);
    fe_select(t.g, dpc, dmc, neg);
}
BP_HD void ge_add_cached_back(ge_ext &r, const ge_efgh &t) {
    fe_mul(r.X, t.f, t.e);
    fe_mul(r.Y, t.h, t.g);
    fe_mul(r.Z, t.f, t.g);
    fe_mul(r.T, t.h, t.e);
}

// mixed addition with an affine Niels point: 7M
BP_HD void ge_madd(ge_ext &r, const ge_ext &p, const ge_niels &q, bool neg) {
    fe ypx, ymx, a, b, c, d, qa, qb;
    fe_select(qa, q.ymx, q.ypx, neg);
    fe_select(qb, q.ypx, q.ymx, neg);
    fe_add(ypx, p.Y, p.X);
    fe_sub_rr(ymx, p.Y, p.X);           // lazy (no carry chain: the coordinates of p are reduced) -- round 6
    fe_mul(a, ymx, qa);
    fe_mul(b, ypx, qb);
    fe_mul(c, p.T, q.t2d);
    fe_add(d, p.Z, p.Z);
    fe e, f, g, h, dmc, dpc;
    fe_sub_rr(e, b, a);                 // lazy; D - C below keeps its carry (D is already a sum of two)
    fe_add(h, b, a);
    fe_sub(dmc, d, c);
    fe_add(dpc, d, c);
    fe_select(f, dmc, dpc, neg);
    fe_select(g, dpc, dmc, neg);
    // second operand = the one fe_mul premultiplies by 19: e and g serve two products each (18 instead of 27 multiplies)
    fe_mul(r.X, f, e);
    fe_mul(r.Y, h, g);
    fe_mul(r.Z, f, g);
    fe_mul(r.T, h, e);
}

// r = +-q for an affine Niels point: what ge_madd(identity, q, neg) computes, with its constant operands folded
// (1 multiplication instead of 7): (X:Y:Z:T) = (2e : 2h : 4 : e h), e = x-part, h = y-part of q
BP_HD void ge_from_niels(ge_ext &r, const ge_niels &q, bool neg) {
    fe qa, qb, e, h;
    fe_select(qa, q.ymx, q.ypx, neg);
    fe_select(qb, q.ypx, q.ymx, neg);
    fe_sub(e, qb, qa);                  // 2x (or -2x)
    fe_add(h, qb, qa);                  // 2y
    fe_carry(h);
    fe_add(r.X, e, e);
    fe_carry(r.X);                      // coordinates of an accumulator are kept reduced
    fe_add(r.Y, h, h);
    fe_carry(r.Y);
    fe_0(r.Z);
    r.Z.v[0] = 4;
    fe_mul(r.T, e, h);
}

BP_HD void ge_add(ge_ext &r, const ge_ext &p, const ge_ext &q) {
    ge_cached c;
    ge_to_cached(c, q);
    ge_add_cached(r, p, c, false);
}

// r = 2p : 4S + 4M (with_t = false skips T, 4S + 3M, for runs of doublings)
BP_HD void ge_dbl(ge_ext &r, const ge_ext &p, bool with_t = true) {
    fe xx, yy, zz2, s, e, f, g, h;
    fe_sq(xx, p.X);
    fe_sq(yy, p.Y);
    fe_sq(zz2, p.Z);
    fe_add(zz2, zz2, zz2);              // lazy
    fe_add(s, p.X, p.Y);                // lazy
    fe_sq(s, s);
    fe_add(h, yy, xx);                  // lazy     (= -H of dbl-2008-hwcd)
    fe_sub_rr(g, yy, xx);               // lazy     (=  G)
    fe_sub(e, s, h);                    //          (=  E)
    fe_sub(f, zz2, g);                  //          (= -F)
    fe_mul(r.X, f, e);
    fe_mul(r.Y, h, g);
    fe_mul(r.Z, f, g);
    if (with_t) fe_mul(r.T, h, e);
}

BP_HD void ge_neg(ge_ext &r, const ge_ext &p) {
    fe_neg(r.X, p.X);
    r.Y = p.Y;
    r.Z = p.Z;
    fe_neg(r.T, p.T);
}

// ristretto identity coset test (what IsIdentity amounts to, mod.rs:447)
BP_HD bool ge_is_identity(const ge_ext &p) { return fe_iszero(p.X) || fe_iszero(p.Y); }

BP_HD void ge_select(ge_ext &r, const ge_ext &a, const ge_ext &b, bool sel_b) {
    fe_select(r.X, a.X, b.X, sel_b);
    fe_select(r.Y, a.Y, b.Y, sel_b);
    fe_select(r.Z, a.Z, b.Z, sel_b);
    fe_select(r.T, a.T, b.T, sel_b);
}

// ---- ristretto255 ----------------------------------------------------------
// RFC 9496 SQRT_RATIO_M1: r = sqrt(u/v) (or sqrt(i*u/v)), non-negative; returns was_square
BP_HD bool fe_sqrt_ratio_i(fe &r, const fe &u, const fe &v) {
    const fe sqrt_m1 = BP_FE_SQRT_M1;
    fe v3, v7, t, check, neg_u, neg_u_i;
    fe_sq(v3, v);
    fe_mul(v3, v3, v);
    fe_sq(v7, v3);
    fe_mul(v7, v7, v);
    fe_mul(t, u, v7);
    fe_pow22523(t, t);
    fe_mul(t, t, v3);
    fe_mul(t, t, u);
    fe_sq(check, t);
    fe_mul(check, check, v);
    fe_neg(neg_u, u);
    fe_mul(neg_u_i, neg_u, sqrt_m1);
    const bool correct = fe_eq(check, u);
    const bool flipped = fe_eq(check, neg_u);
    const bool flipped_i = fe_eq(check, neg_u_i);
    fe ti;
    fe_mul(ti, t, sqrt_m1);
    fe_select(t, t, ti, flipped || flipped_i);
    fe_abs(t);
    r = t;
    return correct || flipped;
}

// decode 32 bytes (as 8 LE words); false = not a valid canonical encoding
BP_HD bool ristretto_decompress(ge_ext &r, const uint32_t w[8]) {
    const fe d = BP_FE_D;
    fe s, one;
    fe_1(one);
    fe_from_words(s, w);
    uint32_t chk[8];
    fe_to_words(chk, s);
    bool canonical = true;
#pragma unroll
    for (int i = 0; i < 8; i++) canonical = canonical && (chk[i] == w[i]);
    const bool s_neg = w[0] & 1;
    fe ss, u1, u2, u2s, v, I, Dx, Dy, t;
    fe_sq(ss, s);
    fe_sub(u1, one, ss);
    fe_add(u2, one, ss);
    fe_sq(u2s, u2);
    fe_sq(t, u1);
    fe_mul(t, t, d);
    fe_neg(t, t);
    fe_sub(v, t, u2s);
    fe_mul(t, v, u2s);
    const bool ok = fe_sqrt_ratio_i(I, one, t);
    fe_mul(Dx, I, u2);
    fe_mul(Dy, I, Dx);
    fe_mul(Dy, Dy, v);
    fe_add(t, s, s);
    fe_mul(r.X, t, Dx);
    fe_abs(r.X);
    fe_mul(r.Y, u1, Dy);
    fe_1(r.Z);
    fe_mul(r.T, r.X, r.Y);
    return canonical && !s_neg && ok && !fe_isneg(r.T) && !fe_iszero(r.Y);
}

// ---- the same decode with a short register footprint -------------------------------------------------------------------
// ristretto_decompress keeps s, u1, u2, v (and the square-root helper's u, v, v^3) alive across the 252-squaring chain: with the
// multiplication's own working set that is ~320 VGPRs, ONE wavefront per SIMD, and a decode launch of 2 081 wavefronts ran three
// rounds of a latency-bound chain (k_bk_prepare alone: 327 us for 64 x 2 081 points, 130 us for 2 081; profiles/r06).  Here only
// t = v u2^2 crosses the chain; afterwards the 32 input bytes are read AGAIN and s, u1, u2, v are formed a second time (five field
// operations of ~275).  BP_OPAQUE ties the second read to the chain's result so that the compiler can neither reuse the first
// computation nor hoist the second one above the chain.  Same result as ristretto_decompress, bit for bit (the CPU harness compares).
#if defined(__HIP_DEVICE_COMPILE__)
#define BP_OPAQUE2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#else
#define BP_OPAQUE2(a, b) ((void)0)
#endif
// I = 1 / sqrt(t) (or 1 / sqrt(i t)), non-negative; false when t is not a square (RFC 9496 SQRT_RATIO_M1 with u = 1)
// in two halves, so that the tail of a narrow MSM chain can run the first one -- the 252-squaring chain -- with a wavefront per field
// element (horner_wave.h: hw_invsqrt_raw, one 16-bit limb per lane) instead of one lane
BP_HD void fe_invsqrt_raw(fe &r, const fe &t) {   // r = t^3 (t^7)^((p-5)/8)
    fe v3, v7;
    fe_sq(v3, t);
    fe_mul(v3, v3, t);
    fe_sq(v7, v3);
    fe_mul(v7, v7, t);
    fe_pow22523(r, v7);
    fe_sq(v3, t);          // (formed again: v^3 does not ride through the chain)
    fe_mul(v3, v3, t);
    fe_mul(r, r, v3);
}
BP_HD bool fe_invsqrt_fix(fe &I, const fe &raw, const fe &t) {
    const fe sqrt_m1 = BP_FE_SQRT_M1;
    fe r = raw;
    fe check, one, m1, mi;
    fe_sq(check, r);
    fe_mul(check, check, t);
    fe_1(one);
    fe_neg(m1, one);
    fe_neg(mi, sqrt_m1);
    const bool correct = fe_eq(check, one);
    const bool flipped = fe_eq(check, m1);
    const bool flipped_i = fe_eq(check, mi);
    fe ri;
    fe_mul(ri, r, sqrt_m1);
    fe_select(r, r, ri, flipped || flipped_i);
    fe_abs(r);
    I = r;
    return correct || flipped;
}
BP_HD bool fe_invsqrt_i(fe &I, const fe &t) {
    fe r;
    fe_invsqrt_raw(r, t);
    return fe_invsqrt_fix(I, r, t);
}
// s, u1 = 1 - s^2, u2 = 1 + s^2, v = -d u1^2 - u2^2 from the encoding's words
BP_HD void ristretto_decode_front(fe &s, fe &u1, fe &u2, fe &v, const uint32_t w[8]) {
    const fe d = BP_FE_D;
    fe one, ss, u2s, t;
    fe_1(one);
    fe_from_words(s, w);
    fe_sq(ss, s);
    fe_sub(u1, one, ss);
    fe_add(u2, one, ss);
    fe_sq(u2s, u2);
    fe_sq(t, u1);
    fe_mul(t, t, d);
    fe_neg(t, t);
    fe_sub(v, t, u2s);
}
BP_HD bool ristretto_decompress_lp(ge_ext &r, const uint32_t *src /*8 words, read twice*/) {
    fe tin;
    bool canonical = true, s_neg;
    {
        uint32_t w[8], chk[8];
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = src[i];
        fe s, u1, u2, v, u2s;
        ristretto_decode_front(s, u1, u2, v, w);
        fe_to_words(chk, s);
#pragma unroll
        for (int i = 0; i < 8; i++) canonical = canonical && (chk[i] == w[i]);
        s_neg = w[0] & 1;
        fe_sq(u2s, u2);
        fe_mul(tin, v, u2s);
    }
    fe I;
    const bool ok = fe_invsqrt_i(I, tin);
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = src[i];
#pragma unroll
    for (int i = 0; i < 8; i++) BP_OPAQUE2(w[i], I.v[0]);
    fe s, u1, u2, v, Dx, Dy, t;
    ristretto_decode_front(s, u1, u2, v, w);
    fe_mul(Dx, I, u2);
    fe_mul(Dy, I, Dx);
    fe_mul(Dy, Dy, v);
    fe_add(t, s, s);
    fe_mul(r.X, t, Dx);
    fe_abs(r.X);
    fe_mul(r.Y, u1, Dy);
    fe_1(r.Z);
    fe_mul(r.T, r.X, r.Y);
    return canonical && !s_neg && ok && !fe_isneg(r.T) && !fe_iszero(r.Y);
}

BP_HD void ristretto_compress(uint32_t out[8], const ge_ext &p) {
    const fe sqrt_m1 = BP_FE_SQRT_M1, invsqrt_a_minus_d = BP_FE_INVSQRT_A_MINUS_D;
    fe one, u1, u2, t, I, i1, i2, zinv, den, X, Y, a, b;
    fe_1(one);
    fe_add(a, p.Z, p.Y);
    fe_sub(b, p.Z, p.Y);
    fe_mul(u1, a, b);
    fe_mul(u2, p.X, p.Y);
    fe_sq(t, u2);
    fe_mul(t, t, u1);
    fe_sqrt_ratio_i(I, one, t);
    fe_mul(i1, I, u1);
    fe_mul(i2, I, u2);
    fe_mul(zinv, i1, i2);
    fe_mul(zinv, zinv, p.T);
    fe_mul(t, p.T, zinv);
    const bool rotate = fe_isneg(t);
    fe xr, yr, dr;
    fe_mul(xr, p.Y, sqrt_m1);
    fe_mul(yr, p.X, sqrt_m1);
    fe_mul(dr, i1, invsqrt_a_minus_d);
    fe_select(X, p.X, xr, rotate);
    fe_select(Y, p.Y, yr, rotate);
    fe_select(den, i2, dr, rotate);
    fe_mul(t, X, zinv);
    fe_cneg(Y, fe_isneg(t));
    fe_sub(t, p.Z, Y);
    fe_mul(t, t, den);
    fe_abs(t);
    fe_to_words(out, t);
}

// ristretto_compress with the short register footprint (see ristretto_decompress_lp): only t = u1 u2^2 crosses the squaring chain, the
// point is read AGAIN from `src` afterwards (LDS or global memory) and u1, u2 are formed a second time.
// (front: t = u1 u2^2, the input of the inverse square root; back: the encoding from I = 1 / sqrt(t), fe_invsqrt_fix's output)
BP_HD void ristretto_compress_front(fe &tin, const ge_ext *src) {
    const ge_ext p = *src;
    fe u1, u2, a, b, t;
    fe_add(a, p.Z, p.Y);
    fe_sub(b, p.Z, p.Y);
    fe_mul(u1, a, b);
    fe_mul(u2, p.X, p.Y);
    fe_sq(t, u2);
    fe_mul(tin, t, u1);
}
BP_HD void ristretto_compress_back(uint32_t out[8], const ge_ext *src, const fe &Iin) {
    const fe sqrt_m1 = BP_FE_SQRT_M1, invsqrt_a_minus_d = BP_FE_INVSQRT_A_MINUS_D;
    fe I = Iin;
    ge_ext p = *src;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        BP_OPAQUE2(p.X.v[i], I.v[0]);
        BP_OPAQUE2(p.Y.v[i], I.v[0]);
        BP_OPAQUE2(p.Z.v[i], I.v[0]);
        BP_OPAQUE2(p.T.v[i], I.v[0]);
    }
    fe u1, u2, t, i1, i2, zinv, den, X, Y, a, b;
    fe_add(a, p.Z, p.Y);
    fe_sub(b, p.Z, p.Y);
    fe_mul(u1, a, b);
    fe_mul(u2, p.X, p.Y);
    fe_mul(i1, I, u1);
    fe_mul(i2, I, u2);
    fe_mul(zinv, i1, i2);
    fe_mul(zinv, zinv, p.T);
    fe_mul(t, p.T, zinv);
    const bool rotate = fe_isneg(t);
    fe xr, yr, dr;
    fe_mul(xr, p.Y, sqrt_m1);
    fe_mul(yr, p.X, sqrt_m1);
    fe_mul(dr, i1, invsqrt_a_minus_d);
    fe_select(X, p.X, xr, rotate);
    fe_select(Y, p.Y, yr, rotate);
    fe_select(den, i2, dr, rotate);
    fe_mul(t, X, zinv);
    fe_cneg(Y, fe_isneg(t));
    fe_sub(t, p.Z, Y);
    fe_mul(t, t, den);
    fe_abs(t);
    fe_to_words(out, t);
}
BP_HD void ristretto_compress_lp(uint32_t out[8], const ge_ext *src) {
    fe tin, I;
    ristretto_compress_front(tin, src);
    fe_invsqrt_i(I, tin);
    ristretto_compress_back(out, src, I);
}

// Elligator 2 map of RFC 9496 section 4.3.4 (one half of from_uniform_bytes)
BP_HD void ristretto_elligator(ge_ext &out, const fe &r0) {
    const fe sqrt_m1 = BP_FE_SQRT_M1, dconst = BP_FE_D, one_minus_d_sq = BP_FE_ONE_MINUS_D_SQ,
             d_minus_one_sq = BP_FE_D_MINUS_ONE_SQ, sqrt_ad_minus_one = BP_FE_SQRT_AD_MINUS_ONE;
    fe one, r, Ns, c, Dd, s, sp, Nt, t, u, W0, W1, W2, W3, ssq;
    fe_1(one);
    fe_sq(r, r0);
    fe_mul(r, r, sqrt_m1);
    fe_add(t, r, one);
    fe_mul(Ns, t, one_minus_d_sq);
    fe_neg(c, one);
    fe_mul(t, dconst, r);
    fe_sub(t, c, t);
    fe_add(u, r, dconst);
    fe_mul(Dd, t, u);
    const bool sq = fe_sqrt_ratio_i(s, Ns, Dd);
    fe_mul(sp, s, r0);
    fe_abs(sp);
    fe_neg(sp, sp);
    fe_select(s, sp, s, sq);
    fe_select(c, r, c, sq);
    fe_sub(t, r, one);
    fe_mul(t, t, c);
    fe_mul(t, t, d_minus_one_sq);
    fe_sub(Nt, t, Dd);
    fe_sq(ssq, s);
    fe_add(t, s, s);
    fe_mul(W0, t, Dd);
    fe_mul(W1, Nt, sqrt_ad_minus_one);
    fe_sub(W2, one, ssq);
    fe_add(W3, one, ssq);
    fe_mul(out.X, W0, W3);
    fe_mul(out.Y, W2, W1);
    fe_mul(out.Z, W1, W3);
    fe_mul(out.T, W0, W2);
}

// RistrettoPoint::from_uniform_bytes: 64 bytes as 16 LE words
BP_HD void ristretto_from_uniform(ge_ext &r, const uint32_t w[16]) {
    fe r1, r2;
    ge_ext p1, p2;
    fe_from_words(r1, w);
    fe_from_words(r2, w + 8);
    ristretto_elligator(p1, r1);
    ristretto_elligator(p2, r2);
    ge_add(r, p1, p2);
}

}  // namespace bp
#endif
