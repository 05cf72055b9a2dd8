// Scalars modulo the ristretto255 group order
//     l = 2^252 + 27742317777372353535851937790883648493
// for one GPU lane: 8 x 32-bit limbs, Montgomery multiplication (R = 2^256).
// Covers what the verifier does with curve25519-dalek's Scalar between the
// transcript and the multiscalar multiplication: from_bytes_mod_order_wide
// (src/transcript.rs:93), from_canonical_bytes (src/range_proof/mod.rs:519-524),
// mul/add/sub/neg, invert / batch_invert (src/inner_product_proof.rs:227,
// mod.rs:414).  Values are canonical (< l) in plain form unless a function
// says "Montgomery form".
#ifndef BPGPU_SC25519_H
#define BPGPU_SC25519_H
#include "fe25519.h"

namespace bp {

struct sc {
    uint32_t v[8];
};

#define BP_SC_L {{0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0u, 0u, 0u, 0x10000000u}}
#define BP_SC_R {{0x8d98951du, 0xd6ec3174u, 0x737dcf70u, 0xc6ef5bf4u, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0x0fffffffu}}
#define BP_SC_RR {{0x449c0f01u, 0xa40611e3u, 0x68859347u, 0xd00e1ba7u, 0x17f5be65u, 0xceec73d2u, 0x7c309a3du, 0x0399411bu}}
#define BP_SC_LFACTOR 0x12547e1bu /* -l^-1 mod 2^32 */

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

// Montgomery product a*b*R^-1 mod l (CIOS, 32-bit words); needs a*b < l*R
BP_HD void sc_montmul(sc &r, const sc &a, const sc &b) {
    const sc l = BP_SC_L;
    uint32_t t[10];
#pragma unroll
    for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t carry = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint64_t s = (uint64_t)a.v[j] * b.v[i] + t[j] + carry;
            t[j] = (uint32_t)s;
            carry = (uint32_t)(s >> 32);
        }
        uint64_t s = (uint64_t)t[8] + carry;
        t[8] = (uint32_t)s;
        t[9] = (uint32_t)(s >> 32);
        const uint32_t m = t[0] * BP_SC_LFACTOR;
        s = (uint64_t)m * l.v[0] + t[0];
        carry = (uint32_t)(s >> 32);
#pragma unroll
        for (int j = 1; j < 8; j++) {
            s = (uint64_t)m * l.v[j] + t[j] + carry;
            t[j - 1] = (uint32_t)s;
            carry = (uint32_t)(s >> 32);
        }
        s = (uint64_t)t[8] + carry;
        t[7] = (uint32_t)s;
        t[8] = t[9] + (uint32_t)(s >> 32);
    }
    sc_csub_l(t, t[8] != 0 || sc_geq_l(t));
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
}
BP_HD void sc_to_mont(sc &r, const sc &a) {
    const sc rr = BP_SC_RR;
    sc_montmul(r, a, rr);
}
BP_HD void sc_from_mont(sc &r, const sc &a) {
    sc one;
    sc_from_u32(one, 1);
    sc_montmul(r, a, one);
}
BP_HD void sc_mul(sc &r, const sc &a, const sc &b) {
    const sc rr = BP_SC_RR;
    sc t;
    sc_montmul(t, a, b);
    sc_montmul(r, t, rr);
}

// 64 little-endian bytes (16 words) -> mod l   (Scalar::from_bytes_mod_order_wide)
BP_HD void sc_from_wide(sc &r, const uint32_t w[16]) {
    const sc rr = BP_SC_RR;
    sc lo, hi, a, b;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        lo.v[i] = w[i];
        hi.v[i] = w[8 + i];
    }
    sc_montmul(a, lo, rr);   // lo * R
    sc_from_mont(a, a);      // lo mod l
    sc_montmul(b, hi, rr);   // hi * 2^256 mod l
    sc_add(r, a, b);
}

// a^(l-2) (variable time in the public exponent only)
BP_HD void sc_invert(sc &r, const sc &a) {
    const sc l = BP_SC_L, one_m = BP_SC_R;
    sc am, acc = one_m;
    sc_to_mont(am, a);
    uint32_t e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = l.v[i];
    e[0] -= 2;
    for (int i = 252; i >= 0; i--) {
        sc_montmul(acc, acc, acc);
        if ((e[i >> 5] >> (i & 31)) & 1) sc_montmul(acc, acc, am);
    }
    sc_from_mont(r, acc);
}

}  // namespace bp
#endif
