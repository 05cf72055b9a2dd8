/* ORACLE -- TEST INFRASTRUCTURE ONLY.
 * Scalars modulo the ristretto255 group order
 *   l = 2^252 + 27742317777372353535851937790883648493
 * 4 x 64-bit limbs, Montgomery multiplication (R = 2^256).  Restates the
 * behaviour of curve25519-dalek's `Scalar` as used by the reference
 * (from_canonical_bytes: src/range_proof/mod.rs:519-524;
 *  from_bytes_mod_order_wide: src/transcript.rs:93; invert/batch_invert:
 *  src/inner_product_proof.rs:227, src/range_proof/mod.rs:414).
 * Values are always kept canonical (< l) in plain (non-Montgomery) form.
 */
#ifndef ORACLE_SC_H
#define ORACLE_SC_H
#include <stdint.h>
#include <string.h>

typedef struct { uint64_t v[4]; } sc;
typedef unsigned __int128 sc_u128;

static const uint64_t SC_L[4] = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0x0ULL, 0x1000000000000000ULL};
static const uint64_t SC_LFACTOR = 0xd2b51da312547e1bULL; /* -l^-1 mod 2^64 */
static const sc SC_R = {{0xd6ec31748d98951dULL, 0xc6ef5bf4737dcf70ULL, 0xfffffffffffffffeULL, 0x0fffffffffffffffULL}};
static const sc SC_RR = {{0xa40611e3449c0f01ULL, 0xd00e1ba768859347ULL, 0xceec73d217f5be65ULL, 0x0399411b7c309a3dULL}};

static inline void sc_0(sc *r) { memset(r, 0, sizeof *r); }
static inline void sc_from_u64(sc *r, uint64_t x) { sc_0(r); r->v[0] = x; }
static inline int sc_eq(const sc *a, const sc *b) { return memcmp(a, b, sizeof *a) == 0; }
static inline int sc_iszero(const sc *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }

/* returns 1 if a >= l */
static inline int sc_geq_l(const uint64_t a[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > SC_L[i]) return 1;
        if (a[i] < SC_L[i]) return 0;
    }
    return 1;
}
static inline void sc_sub_l(uint64_t a[4]) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        sc_u128 d = (sc_u128)a[i] - SC_L[i] - borrow;
        a[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
}
static inline void sc_add(sc *r, const sc *a, const sc *b) {
    uint64_t t[4], carry = 0;
    for (int i = 0; i < 4; i++) {
        sc_u128 s = (sc_u128)a->v[i] + b->v[i] + carry;
        t[i] = (uint64_t)s; carry = (uint64_t)(s >> 64);
    }
    /* a,b < l < 2^253 so no carry out */
    if (sc_geq_l(t)) sc_sub_l(t);
    memcpy(r->v, t, 32);
}
static inline void sc_neg(sc *r, const sc *a) {
    if (sc_iszero(a)) { sc_0(r); return; }
    uint64_t borrow = 0, t[4];
    for (int i = 0; i < 4; i++) {
        sc_u128 d = (sc_u128)SC_L[i] - a->v[i] - borrow;
        t[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1;
    }
    memcpy(r->v, t, 32);
}
static inline void sc_sub(sc *r, const sc *a, const sc *b) {
    sc nb; sc_neg(&nb, b); sc_add(r, a, &nb);
}

/* Montgomery product a*b*R^-1 mod l (CIOS); requires a*b < l*R */
static inline void sc_montmul(sc *r, const sc *a, const sc *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < 4; j++) {
            sc_u128 s = (sc_u128)a->v[j] * b->v[i] + t[j] + carry;
            t[j] = (uint64_t)s; carry = (uint64_t)(s >> 64);
        }
        sc_u128 s = (sc_u128)t[4] + carry;
        t[4] = (uint64_t)s; t[5] = (uint64_t)(s >> 64);
        uint64_t m = t[0] * SC_LFACTOR;
        s = (sc_u128)m * SC_L[0] + t[0];
        carry = (uint64_t)(s >> 64);
        for (int j = 1; j < 4; j++) {
            s = (sc_u128)m * SC_L[j] + t[j] + carry;
            t[j - 1] = (uint64_t)s; carry = (uint64_t)(s >> 64);
        }
        s = (sc_u128)t[4] + carry;
        t[3] = (uint64_t)s;
        t[4] = t[5] + (uint64_t)(s >> 64);
    }
    if (t[4] || sc_geq_l(t)) sc_sub_l(t);
    memcpy(r->v, t, 32);
}
static inline void sc_mul(sc *r, const sc *a, const sc *b) {
    sc t; sc_montmul(&t, a, b); sc_montmul(r, &t, &SC_RR);
}
static inline void sc_sq(sc *r, const sc *a) { sc_mul(r, a, a); }
/* r = a*b + c */
static inline void sc_muladd(sc *r, const sc *a, const sc *b, const sc *c) {
    sc t; sc_mul(&t, a, b); sc_add(r, &t, c);
}

static inline void sc_tobytes(uint8_t out[32], const sc *a) { memcpy(out, a->v, 32); }
/* returns 0 on success, -1 if not canonical (>= l) */
static inline int sc_from_canonical_bytes(sc *r, const uint8_t in[32]) {
    memcpy(r->v, in, 32);
    return sc_geq_l(r->v) ? -1 : 0;
}
/* 512-bit little-endian -> mod l */
static inline void sc_from_wide(sc *r, const uint8_t in[64]) {
    sc lo, hi, a, b;
    memcpy(lo.v, in, 32); memcpy(hi.v, in + 32, 32);
    /* lo*R*R^-1 + hi*R^2*... : montmul(lo, RR) = lo*R ; montmul(.,1)... use:
       x = lo + hi*2^256 ;  lo mod l = montmul(montmul(lo,RR), 1)  */
    sc one; sc_from_u64(&one, 1);
    sc_montmul(&a, &lo, &SC_RR);      /* lo*R mod l */
    sc_montmul(&a, &a, &one);         /* lo mod l */
    sc_montmul(&b, &hi, &SC_RR);      /* hi*R mod l = hi*2^256 mod l */
    sc_add(r, &a, &b);
}
static inline void sc_from_bytes_mod_order(sc *r, const uint8_t in[32]) {
    uint8_t w[64]; memset(w, 0, 64); memcpy(w, in, 32); sc_from_wide(r, w);
}

/* a^(l-2) by square-and-multiply (variable time; public data only) */
static inline void sc_invert(sc *r, const sc *a) {
    uint64_t e[4]; memcpy(e, SC_L, 32); e[0] -= 2;
    sc am, acc; sc_montmul(&am, a, &SC_RR);   /* Montgomery form */
    acc = SC_R;                                /* 1 in Montgomery form */
    for (int i = 255; i >= 0; i--) {
        sc_montmul(&acc, &acc, &acc);
        if ((e[i >> 6] >> (i & 63)) & 1) sc_montmul(&acc, &acc, &am);
    }
    sc one; sc_from_u64(&one, 1);
    sc_montmul(r, &acc, &one);
}
#endif
