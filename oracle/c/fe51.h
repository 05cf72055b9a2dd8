/* ORACLE -- TEST INFRASTRUCTURE ONLY (never linked into libbpgpu.so).
 *
 * GF(2^255-19) in 5 x 51-bit limbs with 128-bit products: the same
 * representation as the reference's serial "u64_backend" of curve25519-dalek
 * (Cargo.toml:21,42 of /root/reference; the crate's source is not vendored, so
 * this is restated from the published algorithm).
 */
#ifndef ORACLE_FE51_H
#define ORACLE_FE51_H
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[5]; } fe;

#define FE_MASK51 ((uint64_t)0x7ffffffffffffULL)

static inline void fe_0(fe *h) { memset(h, 0, sizeof *h); }
static inline void fe_1(fe *h) { memset(h, 0, sizeof *h); h->v[0] = 1; }
static inline void fe_copy(fe *h, const fe *f) { *h = *f; }

static inline void fe_add(fe *h, const fe *f, const fe *g) {
    for (int i = 0; i < 5; i++) h->v[i] = f->v[i] + g->v[i];
}

/* weak reduction: limbs back below 2^51 + small */
static inline void fe_carry(fe *h) {
    uint64_t c;
    c = h->v[0] >> 51; h->v[0] &= FE_MASK51; h->v[1] += c;
    c = h->v[1] >> 51; h->v[1] &= FE_MASK51; h->v[2] += c;
    c = h->v[2] >> 51; h->v[2] &= FE_MASK51; h->v[3] += c;
    c = h->v[3] >> 51; h->v[3] &= FE_MASK51; h->v[4] += c;
    c = h->v[4] >> 51; h->v[4] &= FE_MASK51; h->v[0] += 19 * c;
}

/* h = f - g; adds 16p first so limbs stay positive (inputs < 2^54) */
static inline void fe_sub(fe *h, const fe *f, const fe *g) {
    h->v[0] = f->v[0] + 36028797018963664ULL - g->v[0]; /* 16*(2^51-19) */
    h->v[1] = f->v[1] + 36028797018963952ULL - g->v[1]; /* 16*(2^51-1)  */
    h->v[2] = f->v[2] + 36028797018963952ULL - g->v[2];
    h->v[3] = f->v[3] + 36028797018963952ULL - g->v[3];
    h->v[4] = f->v[4] + 36028797018963952ULL - g->v[4];
    fe_carry(h);
}

static inline void fe_neg(fe *h, const fe *f) {
    fe z; fe_0(&z); fe_sub(h, &z, f);
}

static inline void fe_mul(fe *h, const fe *f, const fe *g) {
    const uint64_t *a = f->v, *b = g->v;
    uint64_t b1_19 = b[1] * 19, b2_19 = b[2] * 19, b3_19 = b[3] * 19, b4_19 = b[4] * 19;
    u128 c0 = (u128)a[0] * b[0] + (u128)a[4] * b1_19 + (u128)a[3] * b2_19 + (u128)a[2] * b3_19 + (u128)a[1] * b4_19;
    u128 c1 = (u128)a[1] * b[0] + (u128)a[0] * b[1] + (u128)a[4] * b2_19 + (u128)a[3] * b3_19 + (u128)a[2] * b4_19;
    u128 c2 = (u128)a[2] * b[0] + (u128)a[1] * b[1] + (u128)a[0] * b[2] + (u128)a[4] * b3_19 + (u128)a[3] * b4_19;
    u128 c3 = (u128)a[3] * b[0] + (u128)a[2] * b[1] + (u128)a[1] * b[2] + (u128)a[0] * b[3] + (u128)a[4] * b4_19;
    u128 c4 = (u128)a[4] * b[0] + (u128)a[3] * b[1] + (u128)a[2] * b[2] + (u128)a[1] * b[3] + (u128)a[0] * b[4];
    uint64_t r0, r1, r2, r3, r4, carry;
    c1 += (uint64_t)(c0 >> 51); r0 = (uint64_t)c0 & FE_MASK51;
    c2 += (uint64_t)(c1 >> 51); r1 = (uint64_t)c1 & FE_MASK51;
    c3 += (uint64_t)(c2 >> 51); r2 = (uint64_t)c2 & FE_MASK51;
    c4 += (uint64_t)(c3 >> 51); r3 = (uint64_t)c3 & FE_MASK51;
    carry = (uint64_t)(c4 >> 51); r4 = (uint64_t)c4 & FE_MASK51;
    r0 += carry * 19;
    r1 += r0 >> 51; r0 &= FE_MASK51;
    h->v[0] = r0; h->v[1] = r1; h->v[2] = r2; h->v[3] = r3; h->v[4] = r4;
}

static inline void fe_sq(fe *h, const fe *f) {
    const uint64_t *a = f->v;
    uint64_t a3_19 = 19 * a[3], a4_19 = 19 * a[4];
    u128 c0 = (u128)a[0] * a[0] + 2 * ((u128)a[1] * a4_19 + (u128)a[2] * a3_19);
    u128 c1 = (u128)a[3] * a3_19 + 2 * ((u128)a[0] * a[1] + (u128)a[2] * a4_19);
    u128 c2 = (u128)a[1] * a[1] + 2 * ((u128)a[0] * a[2] + (u128)a[4] * a3_19);
    u128 c3 = (u128)a[4] * a4_19 + 2 * ((u128)a[0] * a[3] + (u128)a[1] * a[2]);
    u128 c4 = (u128)a[2] * a[2] + 2 * ((u128)a[0] * a[4] + (u128)a[1] * a[3]);
    uint64_t r0, r1, r2, r3, r4, carry;
    c1 += (uint64_t)(c0 >> 51); r0 = (uint64_t)c0 & FE_MASK51;
    c2 += (uint64_t)(c1 >> 51); r1 = (uint64_t)c1 & FE_MASK51;
    c3 += (uint64_t)(c2 >> 51); r2 = (uint64_t)c2 & FE_MASK51;
    c4 += (uint64_t)(c3 >> 51); r3 = (uint64_t)c3 & FE_MASK51;
    carry = (uint64_t)(c4 >> 51); r4 = (uint64_t)c4 & FE_MASK51;
    r0 += carry * 19;
    r1 += r0 >> 51; r0 &= FE_MASK51;
    h->v[0] = r0; h->v[1] = r1; h->v[2] = r2; h->v[3] = r3; h->v[4] = r4;
}

static inline void fe_sqn(fe *h, const fe *f, int n) {
    fe_sq(h, f);
    for (int i = 1; i < n; i++) fe_sq(h, h);
}

static inline void fe_frombytes(fe *h, const uint8_t s[32]) {
    uint64_t w[4];
    memcpy(w, s, 32); /* little-endian host assumed */
    h->v[0] = w[0] & FE_MASK51;
    h->v[1] = ((w[0] >> 51) | (w[1] << 13)) & FE_MASK51;
    h->v[2] = ((w[1] >> 38) | (w[2] << 26)) & FE_MASK51;
    h->v[3] = ((w[2] >> 25) | (w[3] << 39)) & FE_MASK51;
    h->v[4] = (w[3] >> 12) & FE_MASK51; /* bit 255 dropped */
}

/* canonical little-endian encoding */
static inline void fe_tobytes(uint8_t s[32], const fe *f) {
    fe t = *f;
    fe_carry(&t);
    fe_carry(&t);
    /* q = 1 iff t >= p */
    uint64_t q = (t.v[0] + 19) >> 51;
    q = (t.v[1] + q) >> 51;
    q = (t.v[2] + q) >> 51;
    q = (t.v[3] + q) >> 51;
    q = (t.v[4] + q) >> 51;
    t.v[0] += 19 * q;
    uint64_t c;
    c = t.v[0] >> 51; t.v[0] &= FE_MASK51; t.v[1] += c;
    c = t.v[1] >> 51; t.v[1] &= FE_MASK51; t.v[2] += c;
    c = t.v[2] >> 51; t.v[2] &= FE_MASK51; t.v[3] += c;
    c = t.v[3] >> 51; t.v[3] &= FE_MASK51; t.v[4] += c;
    t.v[4] &= FE_MASK51;
    uint64_t w[4];
    w[0] = t.v[0] | (t.v[1] << 51);
    w[1] = (t.v[1] >> 13) | (t.v[2] << 38);
    w[2] = (t.v[2] >> 26) | (t.v[3] << 25);
    w[3] = (t.v[3] >> 39) | (t.v[4] << 12);
    memcpy(s, w, 32);
}

static inline int fe_isneg(const fe *f) {
    uint8_t s[32]; fe_tobytes(s, f); return s[0] & 1;
}
static inline int fe_iszero(const fe *f) {
    uint8_t s[32]; fe_tobytes(s, f);
    uint8_t r = 0; for (int i = 0; i < 32; i++) r |= s[i];
    return r == 0;
}
static inline int fe_eq(const fe *f, const fe *g) {
    uint8_t a[32], b[32]; fe_tobytes(a, f); fe_tobytes(b, g);
    return memcmp(a, b, 32) == 0;
}
static inline void fe_cneg(fe *h, int b) { if (b) { fe t; fe_neg(&t, h); *h = t; } }
static inline void fe_abs(fe *h) { fe_cneg(h, fe_isneg(h)); }

/* z^(2^250-1) and z^11 : shared prefix of the inversion / sqrt chains */
static inline void fe_pow2_250m1(fe *out, fe *z11, const fe *z) {
    fe t0, t1, t2;
    fe_sq(&t0, z);                 /* 2 */
    fe_sqn(&t1, &t0, 2);           /* 8 */
    fe_mul(&t1, z, &t1);           /* 9 */
    fe_mul(&t0, &t0, &t1);         /* 11 */
    *z11 = t0;
    fe_sq(&t0, &t0);               /* 22 */
    fe_mul(&t0, &t1, &t0);         /* 31 = 2^5-1 */
    fe_sqn(&t1, &t0, 5); fe_mul(&t0, &t1, &t0);     /* 2^10-1 */
    fe_sqn(&t1, &t0, 10); fe_mul(&t1, &t1, &t0);    /* 2^20-1 */
    fe_sqn(&t2, &t1, 20); fe_mul(&t1, &t2, &t1);    /* 2^40-1 */
    fe_sqn(&t1, &t1, 10); fe_mul(&t0, &t1, &t0);    /* 2^50-1 */
    fe_sqn(&t1, &t0, 50); fe_mul(&t1, &t1, &t0);    /* 2^100-1 */
    fe_sqn(&t2, &t1, 100); fe_mul(&t1, &t2, &t1);   /* 2^200-1 */
    fe_sqn(&t1, &t1, 50); fe_mul(out, &t1, &t0);    /* 2^250-1 */
}
static inline void fe_invert(fe *out, const fe *z) {
    fe t, z11; fe_pow2_250m1(&t, &z11, z);
    fe_sqn(&t, &t, 5); fe_mul(out, &t, &z11);       /* 2^255-21 */
}
static inline void fe_pow22523(fe *out, const fe *z) {
    fe t, z11; fe_pow2_250m1(&t, &z11, z);
    fe_sqn(&t, &t, 2); fe_mul(out, &t, z);          /* 2^252-3 */
}
#endif
