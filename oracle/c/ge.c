/* ORACLE -- TEST INFRASTRUCTURE ONLY. See ge.h. */
#include "ge.h"
#include <stdlib.h>
#include "ifma4.h"   /* 4-way IFMA backend of the Straus MSM, compiled in when the target CPU has it (bench.py's -march=native build) */

static const uint8_t K_D[32] = {0xa3, 0x78, 0x59, 0x13, 0xca, 0x4d, 0xeb, 0x75, 0xab, 0xd8, 0x41, 0x41, 0x4d, 0x0a, 0x70, 0x00, 0x98, 0xe8, 0x79, 0x77, 0x79, 0x40, 0xc7, 0x8c, 0x73, 0xfe, 0x6f, 0x2b, 0xee, 0x6c, 0x03, 0x52};
static const uint8_t K_D2[32] = {0x59, 0xf1, 0xb2, 0x26, 0x94, 0x9b, 0xd6, 0xeb, 0x56, 0xb1, 0x83, 0x82, 0x9a, 0x14, 0xe0, 0x00, 0x30, 0xd1, 0xf3, 0xee, 0xf2, 0x80, 0x8e, 0x19, 0xe7, 0xfc, 0xdf, 0x56, 0xdc, 0xd9, 0x06, 0x24};
static const uint8_t K_SQRT_M1[32] = {0xb0, 0xa0, 0x0e, 0x4a, 0x27, 0x1b, 0xee, 0xc4, 0x78, 0xe4, 0x2f, 0xad, 0x06, 0x18, 0x43, 0x2f, 0xa7, 0xd7, 0xfb, 0x3d, 0x99, 0x00, 0x4d, 0x2b, 0x0b, 0xdf, 0xc1, 0x4f, 0x80, 0x24, 0x83, 0x2b};
static const uint8_t K_INVSQRT_A_MINUS_D[32] = {0xea, 0x40, 0x5d, 0x80, 0xaa, 0xfd, 0xc8, 0x99, 0xbe, 0x72, 0x41, 0x5a, 0x17, 0x16, 0x2f, 0x9d, 0x40, 0xd8, 0x01, 0xfe, 0x91, 0x7b, 0xc2, 0x16, 0xa2, 0xfc, 0xaf, 0xcf, 0x05, 0x89, 0x6c, 0x78};
static const uint8_t K_SQRT_AD_MINUS_ONE[32] = {0x1b, 0x2e, 0x7b, 0x49, 0xa0, 0xf6, 0x97, 0x7e, 0xbd, 0x54, 0x78, 0x1b, 0x0c, 0x8e, 0x9d, 0xaf, 0xfd, 0xd1, 0xf5, 0x31, 0xc9, 0xfc, 0x3c, 0x0f, 0xac, 0x48, 0x83, 0x2b, 0xbf, 0x31, 0x69, 0x37};
static const uint8_t K_ONE_MINUS_D_SQ[32] = {0x76, 0xc1, 0x5f, 0x94, 0xc1, 0x09, 0x7c, 0xe2, 0x0f, 0x35, 0x5e, 0xcd, 0x38, 0xa1, 0x81, 0x2c, 0xe4, 0xdf, 0x70, 0xbe, 0xdd, 0xab, 0x94, 0x99, 0xd7, 0xe0, 0xb3, 0xb2, 0xa8, 0x72, 0x90, 0x02};
static const uint8_t K_D_MINUS_ONE_SQ[32] = {0x20, 0x4d, 0xed, 0x44, 0xaa, 0x5a, 0xad, 0x31, 0x99, 0x19, 0x1e, 0xb0, 0x2c, 0x4a, 0x9e, 0xd2, 0xeb, 0x4e, 0x9b, 0x52, 0x2f, 0xd3, 0xdc, 0x4c, 0x41, 0x22, 0x6c, 0xf6, 0x7a, 0xb3, 0x68, 0x59};
const uint8_t RISTRETTO_BASEPOINT_COMPRESSED[32] = {0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f, 0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};

static fe C_D, C_D2, C_SQRT_M1, C_INVSQRT_A_MINUS_D, C_SQRT_AD_MINUS_ONE, C_ONE_MINUS_D_SQ, C_D_MINUS_ONE_SQ, C_ONE;
static int ge_ready = 0;
__thread uint64_t ge_op_counter = 0;

void ge_init(void) {
    if (ge_ready) return;
    fe_frombytes(&C_D, K_D); fe_frombytes(&C_D2, K_D2); fe_frombytes(&C_SQRT_M1, K_SQRT_M1);
    fe_frombytes(&C_INVSQRT_A_MINUS_D, K_INVSQRT_A_MINUS_D);
    fe_frombytes(&C_SQRT_AD_MINUS_ONE, K_SQRT_AD_MINUS_ONE);
    fe_frombytes(&C_ONE_MINUS_D_SQ, K_ONE_MINUS_D_SQ);
    fe_frombytes(&C_D_MINUS_ONE_SQ, K_D_MINUS_ONE_SQ);
    fe_1(&C_ONE);
    ge_ready = 1;
}

void ge_identity(ge_p3 *r) { fe_0(&r->X); fe_1(&r->Y); fe_1(&r->Z); fe_0(&r->T); }

static void p1p1_to_p3(ge_p3 *r, const ge_p1p1 *p) {
    fe_mul(&r->X, &p->X, &p->T);
    fe_mul(&r->Y, &p->Y, &p->Z);
    fe_mul(&r->Z, &p->Z, &p->T);
    fe_mul(&r->T, &p->X, &p->Y);
}

void ge_to_cached(ge_cached *r, const ge_p3 *p) {
    fe_add(&r->YpX, &p->Y, &p->X); fe_carry(&r->YpX);
    fe_sub(&r->YmX, &p->Y, &p->X);
    r->Z = p->Z;
    fe_mul(&r->T2d, &p->T, &C_D2);
}

static void add_cached_p1p1(ge_p1p1 *r, const ge_p3 *p, const ge_cached *q, int sub) {
    fe a, b, c, d, ypx, ymx;
    fe_add(&ypx, &p->Y, &p->X);
    fe_sub(&ymx, &p->Y, &p->X);
    if (!sub) { fe_mul(&a, &ymx, &q->YmX); fe_mul(&b, &ypx, &q->YpX); }
    else      { fe_mul(&a, &ymx, &q->YpX); fe_mul(&b, &ypx, &q->YmX); }
    fe_mul(&c, &q->T2d, &p->T);
    fe_mul(&d, &p->Z, &q->Z);
    fe_add(&d, &d, &d);
    fe_sub(&r->X, &b, &a);            /* E */
    fe_add(&r->Y, &b, &a);            /* H */
    if (!sub) { fe_add(&r->Z, &d, &c); fe_sub(&r->T, &d, &c); }  /* G, F */
    else      { fe_sub(&r->Z, &d, &c); fe_add(&r->T, &d, &c); }
    ge_op_counter++;
}
void ge_add_cached(ge_p3 *r, const ge_p3 *p, const ge_cached *q) {
    ge_p1p1 t; add_cached_p1p1(&t, p, q, 0); p1p1_to_p3(r, &t);
}
void ge_sub_cached(ge_p3 *r, const ge_p3 *p, const ge_cached *q) {
    ge_p1p1 t; add_cached_p1p1(&t, p, q, 1); p1p1_to_p3(r, &t);
}
void ge_add(ge_p3 *r, const ge_p3 *p, const ge_p3 *q) {
    ge_cached c; ge_to_cached(&c, q); ge_add_cached(r, p, &c);
}
void ge_sub(ge_p3 *r, const ge_p3 *p, const ge_p3 *q) {
    ge_cached c; ge_to_cached(&c, q); ge_sub_cached(r, p, &c);
}
void ge_neg(ge_p3 *r, const ge_p3 *p) {
    fe_neg(&r->X, &p->X); r->Y = p->Y; r->Z = p->Z; fe_neg(&r->T, &p->T);
}
void ge_dbl(ge_p3 *r, const ge_p3 *p) {
    ge_p1p1 t; fe xx, yy, b, a;
    fe_sq(&xx, &p->X); fe_sq(&yy, &p->Y);
    fe_sq(&b, &p->Z); fe_add(&b, &b, &b);
    fe_add(&a, &p->X, &p->Y); fe_sq(&a, &a);
    fe_add(&t.Y, &yy, &xx);
    fe_sub(&t.Z, &yy, &xx);
    fe_sub(&t.X, &a, &t.Y);
    fe_sub(&t.T, &b, &t.Z);
    p1p1_to_p3(r, &t);
    ge_op_counter++;
}

int ge_is_identity(const ge_p3 *p) { return fe_iszero(&p->X) || fe_iszero(&p->Y); }
int ge_ristretto_eq(const ge_p3 *p, const ge_p3 *q) {
    fe a, b;
    fe_mul(&a, &p->X, &q->Y); fe_mul(&b, &p->Y, &q->X);
    if (fe_eq(&a, &b)) return 1;
    fe_mul(&a, &p->X, &q->X); fe_mul(&b, &p->Y, &q->Y);
    return fe_eq(&a, &b);
}

void ge_scalarmult(ge_p3 *r, const sc *s, const ge_p3 *p) {
    ge_msm_straus(r, 1, s, p);
}

/* ---- ristretto255 (RFC 9496 section 4) ---- */
/* returns was_square; r = sqrt(u/v) or sqrt(i*u/v), non-negative */
static int sqrt_ratio_i(fe *r, const fe *u, const fe *v) {
    fe v3, v7, t, check, neg_u, neg_u_i;
    fe_sq(&v3, v); fe_mul(&v3, &v3, v);
    fe_sq(&v7, &v3); fe_mul(&v7, &v7, v);
    fe_mul(&t, u, &v7); fe_pow22523(&t, &t);
    fe_mul(&t, &t, &v3); fe_mul(&t, &t, u);
    fe_sq(&check, &t); fe_mul(&check, &check, v);
    fe_neg(&neg_u, u); fe_mul(&neg_u_i, &neg_u, &C_SQRT_M1);
    int correct = fe_eq(&check, u);
    int flipped = fe_eq(&check, &neg_u);
    int flipped_i = fe_eq(&check, &neg_u_i);
    if (flipped || flipped_i) fe_mul(&t, &t, &C_SQRT_M1);
    fe_abs(&t);
    *r = t;
    return correct || flipped;
}

int ristretto_decompress(ge_p3 *r, const uint8_t in[32]) {
    ge_init();
    fe s; uint8_t chk[32];
    fe_frombytes(&s, in); fe_tobytes(chk, &s);
    if (memcmp(chk, in, 32) != 0) return -1;     /* non-canonical (>= p or bit 255) */
    if (in[0] & 1) return -1;                    /* negative */
    fe ss, u1, u2, u2s, v, I, Dx, Dy, t;
    fe_sq(&ss, &s);
    fe_sub(&u1, &C_ONE, &ss);
    fe_add(&u2, &C_ONE, &ss); fe_carry(&u2);
    fe_sq(&u2s, &u2);
    fe_sq(&t, &u1); fe_mul(&t, &t, &C_D); fe_neg(&t, &t);
    fe_sub(&v, &t, &u2s);
    fe_mul(&t, &v, &u2s);
    int ok = sqrt_ratio_i(&I, &C_ONE, &t);
    fe_mul(&Dx, &I, &u2);
    fe_mul(&Dy, &I, &Dx); fe_mul(&Dy, &Dy, &v);
    fe_add(&t, &s, &s); fe_mul(&r->X, &t, &Dx); fe_abs(&r->X);
    fe_mul(&r->Y, &u1, &Dy);
    fe_1(&r->Z);
    fe_mul(&r->T, &r->X, &r->Y);
    if (!ok || fe_isneg(&r->T) || fe_iszero(&r->Y)) return -1;
    return 0;
}

/* n decodings; rc[i] as ristretto_decompress.  On IFMA builds four at a time: the prelude and the (p-5)/8 exponentiation of the
 * inverse square root -- 254 of a decoding's ~280 field multiplications -- run 4-way (one point per lane), the short tail per
 * point on the scalar backend.  Same results as n calls of ristretto_decompress. */
#ifdef ORACLE_IFMA
static fe4 fe4_pow22523(fe4 z) {
    fe4 t0, t1, t2, z11;
    t0 = fe4_sq(z);
    t1 = fe4_sq(fe4_sq(t0));
    t1 = fe4_mul(z, t1);
    t0 = fe4_mul(t0, t1);
    z11 = t0; (void)z11;
    t0 = fe4_sq(t0);
    t0 = fe4_mul(t1, t0);
    t1 = t0; for (int i = 0; i < 5; i++) t1 = fe4_sq(t1);
    t0 = fe4_mul(t1, t0);
    t1 = t0; for (int i = 0; i < 10; i++) t1 = fe4_sq(t1);
    t1 = fe4_mul(t1, t0);
    t2 = t1; for (int i = 0; i < 20; i++) t2 = fe4_sq(t2);
    t1 = fe4_mul(t2, t1);
    for (int i = 0; i < 10; i++) t1 = fe4_sq(t1);
    t0 = fe4_mul(t1, t0);
    t1 = t0; for (int i = 0; i < 50; i++) t1 = fe4_sq(t1);
    t1 = fe4_mul(t1, t0);
    t2 = t1; for (int i = 0; i < 100; i++) t2 = fe4_sq(t2);
    t1 = fe4_mul(t2, t1);
    for (int i = 0; i < 50; i++) t1 = fe4_sq(t1);
    t0 = fe4_mul(t1, t0);                       /* 2^250 - 1 */
    t0 = fe4_sq(fe4_sq(t0));
    return fe4_mul(t0, z);                      /* 2^252 - 3 */
}
static void decompress4(ge_p3 r[4], const uint8_t *in[4], int rc[4]) {
    fe s[4], one;
    fe_1(&one);
    for (int j = 0; j < 4; j++) {
        uint8_t chk[32];
        rc[j] = 0;
        fe_frombytes(&s[j], in[j]); fe_tobytes(chk, &s[j]);
        if (memcmp(chk, in[j], 32) != 0 || (in[j][0] & 1)) { rc[j] = -1; fe_0(&s[j]); }   /* non-canonical / negative: lane runs on 0 */
    }
    const fe4 S = fe4_pack(&s[0], &s[1], &s[2], &s[3]), ONE = fe4_pack(&one, &one, &one, &one), D4 = fe4_pack(&C_D, &C_D, &C_D, &C_D);
    const fe4 SS = fe4_sq(S);
    const fe4 U1 = fe4_reduce(fe4_add(ONE, fe4_neg(SS))), U2 = fe4_reduce(fe4_add(ONE, SS));
    const fe4 U2S = fe4_sq(U2);
    const fe4 TD = fe4_mul(fe4_sq(U1), D4);
    const fe4 V = fe4_reduce(fe4_add(fe4_neg(TD), fe4_neg(U2S)));                    /* -d u1^2 - u2^2 */
    const fe4 W = fe4_mul(V, U2S);
    /* sqrt_ratio_i(1, W) up to the candidate root and its check */
    const fe4 W3 = fe4_mul(fe4_sq(W), W), W7 = fe4_mul(fe4_sq(W3), W);
    const fe4 R = fe4_mul(fe4_pow22523(W7), W3);
    const fe4 CHK = fe4_mul(fe4_sq(R), W);
    fe rr[4], chk[4], u1[4], u2[4], v[4];
    fe4_unpack(&rr[0], &rr[1], &rr[2], &rr[3], &R);
    fe4_unpack(&chk[0], &chk[1], &chk[2], &chk[3], &CHK);
    fe4_unpack(&u1[0], &u1[1], &u1[2], &u1[3], &U1);
    fe4_unpack(&u2[0], &u2[1], &u2[2], &u2[3], &U2);
    fe4_unpack(&v[0], &v[1], &v[2], &v[3], &V);
    fe neg_one, neg_i;
    fe_neg(&neg_one, &one); fe_mul(&neg_i, &neg_one, &C_SQRT_M1);
    for (int j = 0; j < 4; j++) {
        fe I = rr[j], Dx, Dy, t;
        fe_carry(&I); fe_carry(&chk[j]); fe_carry(&u1[j]); fe_carry(&u2[j]); fe_carry(&v[j]);
        const int correct = fe_eq(&chk[j], &one), flipped = fe_eq(&chk[j], &neg_one), flipped_i = fe_eq(&chk[j], &neg_i);
        if (flipped || flipped_i) fe_mul(&I, &I, &C_SQRT_M1);
        fe_abs(&I);
        const int ok = correct || flipped;
        fe_mul(&Dx, &I, &u2[j]);
        fe_mul(&Dy, &I, &Dx); fe_mul(&Dy, &Dy, &v[j]);
        fe_add(&t, &s[j], &s[j]); fe_mul(&r[j].X, &t, &Dx); fe_abs(&r[j].X);
        fe_mul(&r[j].Y, &u1[j], &Dy);
        fe_1(&r[j].Z);
        fe_mul(&r[j].T, &r[j].X, &r[j].Y);
        if (!ok || fe_isneg(&r[j].T) || fe_iszero(&r[j].Y)) rc[j] = -1;
    }
}
#endif
void ristretto_decompress_many(ge_p3 *r, const uint8_t *const *in, int *rc, size_t n) {
    ge_init();
    size_t i = 0;
#ifdef ORACLE_IFMA
    for (; i + 4 <= n; i += 4) decompress4(r + i, (const uint8_t **)(in + i), rc + i);
    if (i < n) {   /* ragged tail: pad with the last input */
        ge_p3 t[4]; const uint8_t *pin[4]; int prc[4];
        for (size_t j = 0; j < 4; j++) pin[j] = in[i + j < n ? i + j : n - 1];
        decompress4(t, pin, prc);
        for (size_t j = 0; i + j < n; j++) { r[i + j] = t[j]; rc[i + j] = prc[j]; }
        i = n;
    }
#endif
    for (; i < n; i++) rc[i] = ristretto_decompress(&r[i], in[i]);
}

void ristretto_compress(uint8_t out[32], const ge_p3 *p) {
    ge_init();
    fe u1, u2, t, I, i1, i2, zinv, den, X, Y, a, b;
    fe_add(&a, &p->Z, &p->Y); fe_sub(&b, &p->Z, &p->Y); fe_mul(&u1, &a, &b);
    fe_mul(&u2, &p->X, &p->Y);
    fe_sq(&t, &u2); fe_mul(&t, &t, &u1);
    sqrt_ratio_i(&I, &C_ONE, &t);
    fe_mul(&i1, &I, &u1); fe_mul(&i2, &I, &u2);
    fe_mul(&zinv, &i1, &i2); fe_mul(&zinv, &zinv, &p->T);
    X = p->X; Y = p->Y; den = i2;
    fe_mul(&t, &p->T, &zinv);
    if (fe_isneg(&t)) {
        fe_mul(&X, &p->Y, &C_SQRT_M1);
        fe_mul(&Y, &p->X, &C_SQRT_M1);
        fe_mul(&den, &i1, &C_INVSQRT_A_MINUS_D);
    }
    fe_mul(&t, &X, &zinv);
    if (fe_isneg(&t)) fe_neg(&Y, &Y);
    fe_sub(&t, &p->Z, &Y); fe_mul(&t, &t, &den); fe_abs(&t);
    fe_tobytes(out, &t);
}

static void elligator(ge_p3 *out, const fe *r0) {
    fe r, Ns, c, Dd, s, sp, Nt, t, u, W0, W1, W2, W3, ssq;
    fe_sq(&r, r0); fe_mul(&r, &r, &C_SQRT_M1);
    fe_add(&t, &r, &C_ONE); fe_mul(&Ns, &t, &C_ONE_MINUS_D_SQ);
    fe_neg(&c, &C_ONE);
    fe_mul(&t, &C_D, &r); fe_sub(&t, &c, &t);
    fe_add(&u, &r, &C_D); fe_mul(&Dd, &t, &u);
    int sq = sqrt_ratio_i(&s, &Ns, &Dd);
    fe_mul(&sp, &s, r0); fe_abs(&sp); fe_neg(&sp, &sp);
    if (!sq) { s = sp; c = r; }
    fe_sub(&t, &r, &C_ONE); fe_mul(&t, &t, &c); fe_mul(&t, &t, &C_D_MINUS_ONE_SQ);
    fe_sub(&Nt, &t, &Dd);
    fe_sq(&ssq, &s);
    fe_add(&t, &s, &s); fe_mul(&W0, &t, &Dd);
    fe_mul(&W1, &Nt, &C_SQRT_AD_MINUS_ONE);
    fe_sub(&W2, &C_ONE, &ssq);
    fe_add(&W3, &C_ONE, &ssq);
    fe_mul(&out->X, &W0, &W3);
    fe_mul(&out->Y, &W2, &W1);
    fe_mul(&out->Z, &W1, &W3);
    fe_mul(&out->T, &W0, &W2);
}

void ristretto_from_uniform_bytes(ge_p3 *r, const uint8_t b[64]) {
    ge_init();
    fe r1, r2; ge_p3 p1, p2;
    fe_frombytes(&r1, b);        /* bit 255 cleared, reduced implicitly */
    fe_frombytes(&r2, b + 32);
    elligator(&p1, &r1); elligator(&p2, &r2);
    ge_add(r, &p1, &p2);
}

/* ---- multiscalar multiplication ---- */
static void sc_naf5(int8_t naf[257], const sc *s) {
    uint64_t x[5]; memcpy(x, s->v, 32); x[4] = 0;
    memset(naf, 0, 257);
    int pos = 0; uint64_t carry = 0;
    while (pos < 256) {
        int idx = pos >> 6, bit = pos & 63;
        uint64_t buf = bit < 59 ? (x[idx] >> bit) : ((x[idx] >> bit) | (x[idx + 1] << (64 - bit)));
        uint64_t window = carry + (buf & 31);
        if ((window & 1) == 0) { pos += 1; continue; }
        if (window < 16) { carry = 0; naf[pos] = (int8_t)window; }
        else { carry = 1; naf[pos] = (int8_t)((int)window - 32); }
        pos += 5;
    }
    if (carry) naf[256] = 1;
}

#ifdef ORACLE_IFMA
const char *ge_backend(void) { return "avx512-ifma 4-way (parallel formulas) for the Straus MSM, u64 5x51 elsewhere"; }
/* the same Straus walk (width-5 NAF, 8 odd multiples per point) on the vector backend */
void ge_msm_straus(ge_p3 *r, size_t n, const sc *scalars, const ge_p3 *points) {
    ge_init();
    ge4_init(&C_D2);
    ge_op_counter = 0;
    int8_t (*nafs)[257] = malloc(n * 257 + 1);
    ge4_cached (*tab)[8] = aligned_alloc(32, (n ? n : 1) * sizeof(ge4_cached[8]));
    for (size_t i = 0; i < n; i++) {
        sc_naf5(nafs[i], &scalars[i]);
        ge4 cur = ge4_from_p3(&points[i]);
        ge4 p2 = ge4_dbl(&cur);
        const ge4_cached c2 = ge4_to_cached(&p2);
        tab[i][0] = ge4_to_cached(&cur);
        for (int j = 1; j < 8; j++) { cur = ge4_add_cached(&cur, &c2, 0); tab[i][j] = ge4_to_cached(&cur); }
    }
    ge_p3 id; ge_identity(&id);
    ge4 acc = ge4_from_p3(&id);
    int started = 0;
    for (int b = 256; b >= 0; b--) {
        if (started) { acc = ge4_dbl(&acc); ge_op_counter++; }
        for (size_t i = 0; i < n; i++) {
            int d = nafs[i][b];
            if (d > 0) { acc = ge4_add_cached(&acc, &tab[i][d >> 1], 0); started = 1; ge_op_counter++; }
            else if (d < 0) { acc = ge4_add_cached(&acc, &tab[i][(-d) >> 1], 1); started = 1; ge_op_counter++; }
        }
    }
    ge4_to_p3(r, &acc);
    free(nafs); free(tab);
}
#else
const char *ge_backend(void) { return "u64 5x51 serial"; }
void ge_msm_straus(ge_p3 *r, size_t n, const sc *scalars, const ge_p3 *points) {
    ge_init();
    ge_op_counter = 0;
    int8_t (*nafs)[257] = malloc(n * 257);
    ge_cached (*tab)[8] = malloc(n * sizeof(ge_cached[8]));
    for (size_t i = 0; i < n; i++) {
        sc_naf5(nafs[i], &scalars[i]);
        ge_p3 p2, cur = points[i];
        ge_dbl(&p2, &points[i]);
        ge_cached c2; ge_to_cached(&c2, &p2);
        ge_to_cached(&tab[i][0], &cur);
        for (int j = 1; j < 8; j++) { ge_add_cached(&cur, &cur, &c2); ge_to_cached(&tab[i][j], &cur); }
    }
    ge_p3 acc; ge_identity(&acc);
    int started = 0;
    for (int b = 256; b >= 0; b--) {
        if (started) ge_dbl(&acc, &acc);
        for (size_t i = 0; i < n; i++) {
            int d = nafs[i][b];
            if (d > 0) { ge_add_cached(&acc, &acc, &tab[i][d >> 1]); started = 1; }
            else if (d < 0) { ge_sub_cached(&acc, &acc, &tab[i][(-d) >> 1]); started = 1; }
        }
    }
    *r = acc;
    free(nafs); free(tab);
}
#endif

void ge_msm_pippenger(ge_p3 *r, size_t n, const sc *scalars, const ge_p3 *points) {
    ge_init();
    ge_op_counter = 0;
    int w = n < 500 ? 6 : (n < 800 ? 7 : 8);
    int cols = (256 + w - 1) / w + 1;
    int nb = 1 << (w - 1);
    int16_t *digits = malloc(n * cols * sizeof(int16_t));
    ge_cached *cp = malloc(n * sizeof(ge_cached));
    for (size_t i = 0; i < n; i++) {
        uint64_t x[5]; memcpy(x, scalars[i].v, 32); x[4] = 0;
        int64_t carry = 0;
        for (int c = 0; c < cols; c++) {
            int off = c * w, idx = off >> 6, bit = off & 63;
            uint64_t bits = 0;
            if (idx < 4) {
                bits = x[idx] >> bit;
                if (bit + w > 64) bits |= x[idx + 1] << (64 - bit);
                bits &= ((uint64_t)1 << w) - 1;
            }
            int64_t coef = carry + (int64_t)bits;
            carry = (coef + nb) >> w;
            digits[i * cols + c] = (int16_t)(coef - (carry << w));
        }
        ge_to_cached(&cp[i], &points[i]);
    }
    ge_p3 *buckets = malloc(nb * sizeof(ge_p3));
    ge_p3 total; ge_identity(&total);
    for (int c = cols - 1; c >= 0; c--) {
        for (int b = 0; b < nb; b++) ge_identity(&buckets[b]);
        for (size_t i = 0; i < n; i++) {
            int d = digits[i * cols + c];
            if (d > 0) ge_add_cached(&buckets[d - 1], &buckets[d - 1], &cp[i]);
            else if (d < 0) ge_sub_cached(&buckets[-d - 1], &buckets[-d - 1], &cp[i]);
        }
        ge_p3 inter = buckets[nb - 1], sum = buckets[nb - 1];
        for (int b = nb - 2; b >= 0; b--) {
            ge_add(&inter, &inter, &buckets[b]);
            ge_add(&sum, &sum, &inter);
        }
        for (int k = 0; k < w; k++) ge_dbl(&total, &total);
        ge_add(&total, &total, &sum);
    }
    *r = total;
    free(digits); free(cp); free(buckets);
}

void ge_msm_vartime(ge_p3 *r, size_t n, const sc *scalars, const ge_p3 *points) {
    if (n < 190) ge_msm_straus(r, n, scalars, points);
    else ge_msm_pippenger(r, n, scalars, points);
}
