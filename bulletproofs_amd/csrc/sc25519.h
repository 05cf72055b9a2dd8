// Scalars modulo the ristretto255 group order
//     l = 2^252 + 27742317777372353535851937790883648493
// for one GPU lane.  Covers what the verifier does with curve25519-dalek's Scalar between the
// transcript and the multiscalar multiplication: from_bytes_mod_order_wide (src/transcript.rs:93),
// from_canonical_bytes (src/range_proof/mod.rs:519-524), mul/add/sub/neg, invert / batch_invert
// (src/inner_product_proof.rs:227, mod.rs:414).
//
// Two representations:
//   sc   : 8 x 32-bit words, canonical (< l), the wire / storage form; add, sub, neg, compare.
//   sc28 : 10 limbs of 28 bits, *lazy* (limbs <= 2^28 + 4, value < 2^254), the multiplication form.
// l is friendly to radix 2^28: l = 2^252 + c with c < 2^125, i.e. limb 9 of l is 1 and limbs 5..8
// are 0, so one Montgomery reduction step is 5 multiply-adds + 1 addition.  With R = 2^280 the
// Montgomery product of two lazy values is again < 2^254 without any conditional subtraction, and the
// 100 limb products of the schoolbook part go into independent 64-bit column sums (v_mad_u64_u32) --
// no carry chain until the final, fully parallel, two-round normalisation.  (The previous 8x32-bit
// CIOS version was latency-bound at ~2350 cycles per product on a lone wavefront.)
#ifndef BPGPU_SC25519_H
#define BPGPU_SC25519_H
#include "fe25519.h"

namespace bp {

struct sc {
    uint32_t v[8];
};
struct sc28 {
    uint32_t v[10];
};

#define BP_SC_L {{0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0u, 0u, 0u, 0x10000000u}}
#define BP_M28 0xfffffffu
// c = l - 2^252 in 28-bit limbs (limbs 0..4 of l); limb 9 of l is 1
#define BP_SC28_C0 0xcf5d3edu
#define BP_SC28_C1 0x12631a5u
#define BP_SC28_C2 0x79cd658u
#define BP_SC28_C3 0xf9dea2fu
#define BP_SC28_C4 0x00014deu
#define BP_SC28_LFACTOR 0x2547e1bu /* -l^-1 mod 2^28 */
#define BP_SC28_R {{0xcf5d3edu, 0x4305db8u, 0x676a4b2u, 0x80113d7u, 0x0622aafu, 0xfffeb21u, 0xfffffffu, 0xfffffffu, 0xfffffffu, 0x0000000u}}     /* 2^280 mod l */
#define BP_SC28_RR {{0xf8305abu, 0x8b5e723u, 0x32452e2u, 0xef1d5efu, 0x7a33f10u, 0x5c63aa9u, 0x5be65cbu, 0x73d217fu, 0xa3dceecu, 0x0000000u}}    /* 2^560 mod l */
#define BP_SC28_K536 {{0x662e943u, 0xe216d48u, 0xd14d357u, 0xee7162cu, 0xc63b992u, 0xbe65cb5u, 0x3d217f5u, 0x3dceec7u, 0xb7c309au, 0x0000000u}}  /* 2^536 mod l */

// ---- canonical 8x32 form ------------------------------------------------------
BP_HD void sc_0(sc &r) {
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
}
BP_HD void sc_from_u32(sc &r, uint32_t x) {
    sc_0(r);
    r.v[0] = x;
}
BP_HD bool sc_iszero(const sc &a) {
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r |= a.v[i];
    return r == 0;
}
// t >= l ?
BP_HD bool sc_geq_l(const uint32_t t[8]) {
    const sc l = BP_SC_L;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t d = (uint64_t)t[i] - l.v[i] - borrow;
        borrow = (uint32_t)(d >> 63);
    }
    return borrow == 0;
}
// t -= l if cond
BP_HD void sc_csub_l(uint32_t t[8], bool cond) {
    const sc l = BP_SC_L;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t d = (uint64_t)t[i] - l.v[i] - borrow;
        borrow = (uint32_t)(d >> 63);
        t[i] = cond ? (uint32_t)d : t[i];
    }
}
BP_HD bool sc_is_canonical_sc(const sc &a) { return !sc_geq_l(a.v); }

BP_HD void sc_add(sc &r, const sc &a, const sc &b) {
    uint32_t t[8], carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t s = (uint64_t)a.v[i] + b.v[i] + carry;
        t[i] = (uint32_t)s;
        carry = (uint32_t)(s >> 32);
    }
    sc_csub_l(t, sc_geq_l(t));   // a, b < l < 2^253: no carry out
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
}
BP_HD void sc_neg(sc &r, const sc &a) {
    const sc l = BP_SC_L;
    const bool z = sc_iszero(a);
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t d = (uint64_t)l.v[i] - a.v[i] - borrow;
        borrow = (uint32_t)(d >> 63);
        r.v[i] = z ? 0u : (uint32_t)d;
    }
}
BP_HD void sc_sub(sc &r, const sc &a, const sc &b) {
    sc nb;
    sc_neg(nb, b);
    sc_add(r, a, nb);
}

// ---- lazy 10x28 multiplication form ---------------------------------------------
// 256-bit words -> limbs (any 256-bit value, not necessarily < l)
BP_HD void sc28_from_words(sc28 &r, const uint32_t w[8]) {
    r.v[0] = w[0] & BP_M28;
    r.v[1] = ((w[0] >> 28) | (w[1] << 4)) & BP_M28;
    r.v[2] = ((w[1] >> 24) | (w[2] << 8)) & BP_M28;
    r.v[3] = ((w[2] >> 20) | (w[3] << 12)) & BP_M28;
    r.v[4] = ((w[3] >> 16) | (w[4] << 16)) & BP_M28;
    r.v[5] = ((w[4] >> 12) | (w[5] << 20)) & BP_M28;
    r.v[6] = ((w[5] >> 8) | (w[6] << 24)) & BP_M28;
    r.v[7] = w[6] >> 4;
    r.v[8] = w[7] & BP_M28;
    r.v[9] = w[7] >> 28;
}
BP_HD void sc28_from_sc(sc28 &r, const sc &a) { sc28_from_words(r, a.v); }

// lazy value (< 2^254) -> canonical words
BP_HD void sc_from_sc28(sc &r, const sc28 &a) {
    uint32_t t[10], carry = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const uint32_t s = a.v[i] + carry;   // <= 2^28 + 4 + 2
        t[i] = s & BP_M28;
        carry = s >> 28;
    }
    uint32_t w[8];
    w[0] = t[0] | (t[1] << 28);
    w[1] = (t[1] >> 4) | (t[2] << 24);
    w[2] = (t[2] >> 8) | (t[3] << 20);
    w[3] = (t[3] >> 12) | (t[4] << 16);
    w[4] = (t[4] >> 16) | (t[5] << 12);
    w[5] = (t[5] >> 20) | (t[6] << 8);
    w[6] = (t[6] >> 24) | (t[7] << 4);
    w[7] = t[8] | (t[9] << 28);
    sc_csub_l(w, sc_geq_l(w));   // value < 2^254 < 4l: at most three subtractions
    sc_csub_l(w, sc_geq_l(w));
    sc_csub_l(w, sc_geq_l(w));
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = w[i];
}

BP_HD void sc28_montreduce(sc28 &r, uint64_t t[20]);

// r = a * b * 2^-280 mod l, lazy in, lazy out.
// Bounds: limbs <= 2^28+4 -> products < 2^56.01, schoolbook columns < 10 * 2^56.01 < 2^59.4; the reduction
// adds < 5 * 2^56 + 2^28 + 2^32 per column -> every column stays < 2^60.  Value: (a*b + m*l) / 2^280
// < 2^228 + 2^253 for a, b < 2^254.
BP_HD void sc28_montmul(sc28 &r, const sc28 &a, const sc28 &b) {
    uint64_t t[20];
#pragma unroll
    for (int k = 0; k < 20; k++) t[k] = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
#pragma unroll
        for (int j = 0; j < 10; j++) t[i + j] += (uint64_t)a.v[i] * b.v[j];
    }
    sc28_montreduce(r, t);
}
// shared tail of the Montgomery product: reduce the 20 column sums and normalise (see sc28_montmul)
BP_HD void sc28_montreduce(sc28 &r, uint64_t t[20]) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const uint32_t m = ((uint32_t)t[i] * BP_SC28_LFACTOR) & BP_M28;
        t[i] += (uint64_t)m * BP_SC28_C0;
        t[i + 1] += (uint64_t)m * BP_SC28_C1;
        t[i + 2] += (uint64_t)m * BP_SC28_C2;
        t[i + 3] += (uint64_t)m * BP_SC28_C3;
        t[i + 4] += (uint64_t)m * BP_SC28_C4;
        t[i + 9] += m;
        t[i + 1] += t[i] >> 28;
    }
    uint32_t u[10];
#pragma unroll
    for (int k = 0; k < 10; k++) {
        uint64_t x = t[10 + k] & BP_M28;
        if (k >= 1) x += (t[10 + k - 1] >> 28) & BP_M28;
        if (k >= 2) x += t[10 + k - 2] >> 56;
        u[k] = (uint32_t)x;
    }
#pragma unroll
    for (int k = 0; k < 10; k++) r.v[k] = (u[k] & BP_M28) + (k >= 1 ? (u[k - 1] >> 28) : 0u);
}
// r = a^2 * 2^-280 mod l: 55 limb products instead of 100 (off-diagonal ones doubled)
BP_HD void sc28_montsq(sc28 &r, const sc28 &a) {
    uint64_t t[20];
#pragma unroll
    for (int k = 0; k < 20; k++) t[k] = 0;
    uint32_t a2[10];
#pragma unroll
    for (int i = 0; i < 10; i++) a2[i] = a.v[i] << 1;   // <= 2^29 + 8
#pragma unroll
    for (int i = 0; i < 10; i++) {
        t[2 * i] += (uint64_t)a.v[i] * a.v[i];
#pragma unroll
        for (int j = i + 1; j < 10; j++) t[i + j] += (uint64_t)a2[i] * a.v[j];
    }
    sc28_montreduce(r, t);
}
BP_HD void sc28_to_mont(sc28 &r, const sc28 &a) {
    const sc28 rr = BP_SC28_RR;
    sc28_montmul(r, a, rr);
}
BP_HD void sc28_from_mont(sc28 &r, const sc28 &a) {
    sc28 one;
#pragma unroll
    for (int i = 0; i < 10; i++) one.v[i] = (i == 0);
    sc28_montmul(r, a, one);
}
BP_HD void sc28_one_mont(sc28 &r) {
    const sc28 R = BP_SC28_R;
    r = R;
}
// Montgomery form in, Montgomery form out: a^(l-2).  l - 2 = 2^252 + (c - 2): after the leading 1 come
// 127 zero bits (squarings only) and a 125-bit tail, walked in 4-bit windows against a table of
// a^0..a^15 (the exponent is a public constant, so the walk is the same for every lane).
BP_HD void sc28_invert_mont(sc28 &r, const sc28 &am) {
    // nibbles of (c - 2) = 0x14def9dea2f79cd65812631a5cf5d3eb, most significant first (32 nibbles = 128 bits)
    const uint8_t nib[32] = {0x1, 0x4, 0xd, 0xe, 0xf, 0x9, 0xd, 0xe, 0xa, 0x2, 0xf, 0x7, 0x9, 0xc, 0xd, 0x6,
                             0x5, 0x8, 0x1, 0x2, 0x6, 0x3, 0x1, 0xa, 0x5, 0xc, 0xf, 0x5, 0xd, 0x3, 0xe, 0xb};
    sc28 tab[16];
    sc28_one_mont(tab[0]);
    tab[1] = am;
    for (int i = 2; i < 16; i++) sc28_montmul(tab[i], tab[i - 1], am);
    sc28 acc = am;                                      // the leading bit (2^252)
    for (int i = 0; i < 252 - 128; i++) sc28_montsq(acc, acc);
    for (int i = 0; i < 32; i++) {
        sc28_montsq(acc, acc);
        sc28_montsq(acc, acc);
        sc28_montsq(acc, acc);
        sc28_montsq(acc, acc);
        sc28_montmul(acc, acc, tab[nib[i]]);            // (nibble 0 multiplies by one: keeps the lanes uniform)
    }
    r = acc;
}

// ---- canonical-form conveniences built on sc28 -------------------------------------
BP_HD void sc_to_mont28(sc28 &r, const sc &a) {
    sc28 t;
    sc28_from_sc(t, a);
    sc28_to_mont(r, t);
}
BP_HD void sc_from_mont28(sc &r, const sc28 &am) {
    sc28 t;
    sc28_from_mont(t, am);
    sc_from_sc28(r, t);
}
BP_HD void sc_mul(sc &r, const sc &a, const sc &b) {
    sc28 x, y, t;
    sc28_from_sc(x, a);
    sc28_from_sc(y, b);
    sc28_montmul(t, x, y);   // a*b/R
    sc28_to_mont(t, t);      // a*b
    sc_from_sc28(r, t);
}
// 64 little-endian bytes (16 words) -> mod l   (Scalar::from_bytes_mod_order_wide)
BP_HD void sc_from_wide(sc &r, const uint32_t w[16]) {
    const sc28 R = BP_SC28_R, K = BP_SC28_K536;
    sc28 lo, hi, a, b;
    sc28_from_words(lo, w);
    sc28_from_words(hi, w + 8);
    sc28_montmul(a, lo, R);   // lo mod l
    sc28_montmul(b, hi, K);   // hi * 2^256 mod l
    sc x, y;
    sc_from_sc28(x, a);
    sc_from_sc28(y, b);
    sc_add(r, x, y);
}
BP_HD void sc_invert(sc &r, const sc &a) {
    sc28 am, im;
    sc_to_mont28(am, a);
    sc28_invert_mont(im, am);
    sc_from_mont28(r, im);
}

}  // namespace bp
#endif
