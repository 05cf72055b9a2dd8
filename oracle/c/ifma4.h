/* ORACLE -- TEST INFRASTRUCTURE ONLY (never linked into libbpgpu.so).
 *
 * 4-way vectorised GF(2^255-19) and Edwards point arithmetic on AVX-512 IFMA (vpmadd52luq / vpmadd52huq, 256-bit vectors):
 * the CPU baseline's counterpart of the reference dependency's SIMD backends (curve25519-dalek "avx2_backend" /
 * IFMA backend, /root/reference/README.md:69-84, Cargo.toml:42) -- the "parallel formulas" of Hisil-Wong-Carter-Dawson: the
 * four coordinates (X : Y : Z : T) of a point sit in the four 64-bit lanes of a vector, the four independent field
 * multiplications of each half of a point operation are ONE vector multiplication.  Restated from the published technique
 * (the crate's source is not vendored); same radix as fe51.h (5 limbs of 51 bits), so lanes convert by packing.
 * Compiled only when the build targets a CPU with the instructions (bench.py's cpu_baseline rebuilds the oracle with
 * -march=native on the box it times); otherwise ge_msm_straus stays on the scalar u64 backend.  Results are bit-identical
 * (tests/test_oracle.py::test_vector_backend_equals_scalar_backend).
 */
#ifndef ORACLE_IFMA4_H
#define ORACLE_IFMA4_H
#if defined(__AVX512IFMA__) && defined(__AVX512VL__) && !defined(ORACLE_NO_IFMA)
#define ORACLE_IFMA 1
#include <immintrin.h>
#include "fe51.h"

typedef struct { __m256i v[5]; } fe4;   /* lane j of v[i] = limb i of field element j */

#define FE4_MASK _mm256_set1_epi64x((long long)FE_MASK51)

static inline fe4 fe4_pack(const fe *a, const fe *b, const fe *c, const fe *d) {
    fe4 r;
    for (int i = 0; i < 5; i++) r.v[i] = _mm256_set_epi64x((long long)d->v[i], (long long)c->v[i], (long long)b->v[i], (long long)a->v[i]);
    return r;
}
static inline void fe4_unpack(fe *a, fe *b, fe *c, fe *d, const fe4 *x) {
    for (int i = 0; i < 5; i++) {
        uint64_t t[4];
        _mm256_storeu_si256((__m256i *)t, x->v[i]);
        a->v[i] = t[0]; b->v[i] = t[1]; c->v[i] = t[2]; d->v[i] = t[3];
    }
}
/* parallel carry: any limbs < 2^64 in -> limbs < 2^51 + 19 * 2^13 out (the bound the multiplier's inputs need: < 2^52) */
static inline fe4 fe4_reduce(fe4 x) {
    const __m256i m = FE4_MASK;
    __m256i c0 = _mm256_srli_epi64(x.v[0], 51), c1 = _mm256_srli_epi64(x.v[1], 51), c2 = _mm256_srli_epi64(x.v[2], 51),
            c3 = _mm256_srli_epi64(x.v[3], 51), c4 = _mm256_srli_epi64(x.v[4], 51);
    __m256i c4_19 = _mm256_add_epi64(_mm256_add_epi64(_mm256_slli_epi64(c4, 4), _mm256_slli_epi64(c4, 1)), c4);
    fe4 r;
    r.v[0] = _mm256_add_epi64(_mm256_and_si256(x.v[0], m), c4_19);
    r.v[1] = _mm256_add_epi64(_mm256_and_si256(x.v[1], m), c0);
    r.v[2] = _mm256_add_epi64(_mm256_and_si256(x.v[2], m), c1);
    r.v[3] = _mm256_add_epi64(_mm256_and_si256(x.v[3], m), c2);
    r.v[4] = _mm256_add_epi64(_mm256_and_si256(x.v[4], m), c3);
    return r;
}
static inline fe4 fe4_add(fe4 a, fe4 b) {   /* lazy: reduce before it feeds a multiplication */
    fe4 r;
    for (int i = 0; i < 5; i++) r.v[i] = _mm256_add_epi64(a.v[i], b.v[i]);
    return r;
}
/* 8p - x, lazy (x limbs < 2^54): limbs stay positive */
static inline fe4 fe4_neg(fe4 x) {
    const __m256i b0 = _mm256_set1_epi64x(18014398509481832LL) /* 8 (2^51 - 19) */, b = _mm256_set1_epi64x(18014398509481976LL) /* 8 (2^51 - 1) */;
    fe4 r;
    r.v[0] = _mm256_sub_epi64(b0, x.v[0]);
    for (int i = 1; i < 5; i++) r.v[i] = _mm256_sub_epi64(b, x.v[i]);
    return r;
}
/* lanes permuted by the immediate of vpermq */
#define fe4_perm(x, imm) ({ fe4 r_; const fe4 x_ = (x); r_.v[0] = _mm256_permute4x64_epi64(x_.v[0], (imm)); r_.v[1] = _mm256_permute4x64_epi64(x_.v[1], (imm)); \
    r_.v[2] = _mm256_permute4x64_epi64(x_.v[2], (imm)); r_.v[3] = _mm256_permute4x64_epi64(x_.v[3], (imm)); r_.v[4] = _mm256_permute4x64_epi64(x_.v[4], (imm)); r_; })
#define PERM(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
/* lane j of the result = lane j of (mask bit j set ? b : a); mask over the four 64-bit lanes */
#define LANES32(m) ((((m) & 1) ? 3 : 0) | (((m) & 2) ? 12 : 0) | (((m) & 4) ? 48 : 0) | (((m) & 8) ? 192 : 0))
#define fe4_blend(a, b, lanemask) ({ fe4 r_; const fe4 a_ = (a), b_ = (b); \
    r_.v[0] = _mm256_blend_epi32(a_.v[0], b_.v[0], LANES32(lanemask)); r_.v[1] = _mm256_blend_epi32(a_.v[1], b_.v[1], LANES32(lanemask)); \
    r_.v[2] = _mm256_blend_epi32(a_.v[2], b_.v[2], LANES32(lanemask)); r_.v[3] = _mm256_blend_epi32(a_.v[3], b_.v[3], LANES32(lanemask)); \
    r_.v[4] = _mm256_blend_epi32(a_.v[4], b_.v[4], LANES32(lanemask)); r_; })

/* h = f * g, limbs of f, g < 2^52; output limbs < 2^51 + 2^14 */
static inline fe4 fe4_mul(fe4 f, fe4 g) {
    const __m256i z = _mm256_setzero_si256();
    __m256i lo[9], hi[9];
    for (int k = 0; k < 9; k++) { lo[k] = z; hi[k] = z; }
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 5; j++) {
            lo[i + j] = _mm256_madd52lo_epu64(lo[i + j], f.v[i], g.v[j]);
            hi[i + j] = _mm256_madd52hi_epu64(hi[i + j], f.v[i], g.v[j]);
        }
    /* the high halves sit at 2^52 = 2 * 2^51: z[k] = lo[k] + 2 hi[k-1], k = 0..9 */
    __m256i t[10];
    t[0] = lo[0];
    for (int k = 1; k < 9; k++) t[k] = _mm256_add_epi64(lo[k], _mm256_slli_epi64(hi[k - 1], 1));
    t[9] = _mm256_slli_epi64(hi[8], 1);
    /* 2^255 = 19: r[k] = t[k] + 19 t[k+5] */
    fe4 r;
    for (int k = 0; k < 5; k++) {
        const __m256i u = t[k + 5];
        r.v[k] = _mm256_add_epi64(t[k], _mm256_add_epi64(_mm256_add_epi64(_mm256_slli_epi64(u, 4), _mm256_slli_epi64(u, 1)), u));
    }
    return fe4_reduce(r);
}

/* h = f^2: 15 + 15 multiply-adds instead of 25 + 25 (cross products accumulated once, doubled afterwards) */
static inline fe4 fe4_sq(fe4 f) {
    const __m256i z = _mm256_setzero_si256();
    __m256i dlo[9], dhi[9], clo[9], chi[9];
    for (int k = 0; k < 9; k++) { dlo[k] = z; dhi[k] = z; clo[k] = z; chi[k] = z; }
    for (int i = 0; i < 5; i++) {
        dlo[2 * i] = _mm256_madd52lo_epu64(dlo[2 * i], f.v[i], f.v[i]);
        dhi[2 * i] = _mm256_madd52hi_epu64(dhi[2 * i], f.v[i], f.v[i]);
        for (int j = i + 1; j < 5; j++) {
            clo[i + j] = _mm256_madd52lo_epu64(clo[i + j], f.v[i], f.v[j]);
            chi[i + j] = _mm256_madd52hi_epu64(chi[i + j], f.v[i], f.v[j]);
        }
    }
    __m256i lo[9], hi[9], t[10];
    for (int k = 0; k < 9; k++) {
        lo[k] = _mm256_add_epi64(dlo[k], _mm256_slli_epi64(clo[k], 1));
        hi[k] = _mm256_add_epi64(dhi[k], _mm256_slli_epi64(chi[k], 1));
    }
    t[0] = lo[0];
    for (int k = 1; k < 9; k++) t[k] = _mm256_add_epi64(lo[k], _mm256_slli_epi64(hi[k - 1], 1));
    t[9] = _mm256_slli_epi64(hi[8], 1);
    fe4 r;
    for (int k = 0; k < 5; k++) {
        const __m256i u = t[k + 5];
        r.v[k] = _mm256_add_epi64(t[k], _mm256_add_epi64(_mm256_add_epi64(_mm256_slli_epi64(u, 4), _mm256_slli_epi64(u, 1)), u));
    }
    return fe4_reduce(r);
}

/* ---- points: lanes (X, Y, Z, T), limbs reduced (< 2^52) ------------------------------------------------------- */
typedef struct { fe4 p; } ge4;          /* extended coordinates */
typedef struct { fe4 c; } ge4_cached;   /* (Y+X, Y-X, 2Z, 2dT) */

static inline ge4 ge4_from_p3(const ge_p3 *p) { ge4 r; r.p = fe4_pack(&p->X, &p->Y, &p->Z, &p->T); return r; }
static inline void ge4_to_p3(ge_p3 *r, const ge4 *p) {
    fe4_unpack(&r->X, &r->Y, &r->Z, &r->T, &p->p);
    fe_carry(&r->X); fe_carry(&r->Y); fe_carry(&r->Z); fe_carry(&r->T);
}
/* (X, Y, Z, T) -> (Y+X, Y-X, Z, T), reduced */
static inline fe4 ge4_diffsum(fe4 p) {
    const fe4 a = fe4_perm(p, PERM(1, 1, 2, 3));                 /* (Y, Y, Z, T) */
    const fe4 x = fe4_perm(p, PERM(0, 0, 0, 0));                 /* (X, X, X, X) */
    fe4 s = fe4_blend(x, fe4_neg(x), 2);                          /* (X, -X, X, X) */
    fe4 zero;
    for (int i = 0; i < 5; i++) zero.v[i] = _mm256_setzero_si256();
    s = fe4_blend(s, zero, 12);                                   /* (X, -X, 0, 0) */
    return fe4_reduce(fe4_add(a, s));
}
static fe4 GE4_CACHE_CONST;   /* (1, 1, 2, 2d) */
static int ge4_ready = 0;
static inline void ge4_init(const fe *d2) {
    if (ge4_ready) return;
    fe one, two;
    fe_1(&one);
    fe_0(&two); two.v[0] = 2;
    GE4_CACHE_CONST = fe4_pack(&one, &one, &two, d2);
    ge4_ready = 1;
}
static inline ge4_cached ge4_to_cached(const ge4 *p) {
    ge4_cached r;
    r.c = fe4_mul(ge4_diffsum(p->p), GE4_CACHE_CONST);
    return r;
}
/* the second half shared by addition and doubling: given the four products / squares it needs, form
 * U = (E, G, F, E), V = (F, H, G, H) and return U * V = (X3, Y3, Z3, T3) */
static inline ge4 ge4_add_cached(const ge4 *p, const ge4_cached *q, int sub) {
    /* q as stored: (Y2+X2, Y2-X2, 2 Z2, 2d T2); subtracting = its first two lanes swapped and C = 2d T1 T2 negated */
    const fe4 qc = sub ? fe4_perm(q->c, PERM(1, 0, 2, 3)) : q->c;
    const fe4 prod = fe4_mul(ge4_diffsum(p->p), qc);              /* (PP, MM, D, C) */
    const fe4 p0 = fe4_perm(prod, PERM(0, 2, 2, 0));               /* (PP, D, D, PP) */
    const fe4 p1 = fe4_perm(prod, PERM(1, 3, 3, 1));               /* (MM, C, C, MM) */
    const fe4 q0 = fe4_perm(prod, PERM(2, 0, 2, 0));               /* (D, PP, D, PP) */
    const fe4 q1 = fe4_perm(prod, PERM(3, 1, 3, 1));               /* (C, MM, C, MM) */
    const fe4 n1 = fe4_neg(p1), m1 = fe4_neg(q1);
    /* E = PP - MM, H = PP + MM, F = D - C, G = D + C  (C -> -C when subtracting) */
    const fe4 U = fe4_reduce(fe4_add(p0, sub ? fe4_blend(n1, p1, 4) : fe4_blend(n1, p1, 2)));   /* (E, G, F, E) */
    const fe4 V = fe4_reduce(fe4_add(q0, sub ? fe4_blend(q1, m1, 4) : fe4_blend(q1, m1, 1)));   /* (F, H, G, H) */
    ge4 r;
    r.p = fe4_mul(U, V);
    return r;
}
static inline ge4 ge4_dbl(const ge4 *p) {
    const fe4 u = fe4_perm(p->p, PERM(0, 1, 2, 0));                /* (X, Y, Z, X) */
    const fe4 w = fe4_perm(p->p, PERM(1, 1, 1, 1));                /* (Y, Y, Y, Y) */
    const fe4 t = fe4_reduce(fe4_blend(u, fe4_add(u, w), 8));      /* (X, Y, Z, X+Y) */
    const fe4 sq = fe4_sq(t);                                      /* (A, B, Zs, S) */
    const fe4 A4 = fe4_perm(sq, PERM(0, 0, 0, 0)), B4 = fe4_perm(sq, PERM(1, 1, 1, 1));
    const fe4 S4 = fe4_perm(sq, PERM(3, 3, 3, 3)), Z4 = fe4_perm(sq, PERM(2, 2, 2, 2));
    const fe4 nB = fe4_neg(B4), nS = fe4_neg(S4), C4 = fe4_add(Z4, Z4);
    fe4 zero;
    for (int i = 0; i < 5; i++) zero.v[i] = _mm256_setzero_si256();
    /* E = A + B - S, G = A - B, F = 2 Zs + A - B, H = A + B */
    fe4 U = fe4_add(A4, fe4_blend(B4, nB, 6));                      /* + (B, -B, -B, B) */
    U = fe4_add(U, fe4_blend(fe4_blend(nS, zero, 2), C4, 4));       /* + (-S, 0, 2Zs, -S)  -> (E, G, F, E) */
    fe4 V = fe4_add(A4, fe4_blend(B4, nB, 5));                      /* + (-B, B, -B, B) */
    V = fe4_add(V, fe4_blend(zero, C4, 1));                         /* + (2Zs, 0, 0, 0)    -> (F, H, G, H) */
    ge4 r;
    r.p = fe4_mul(fe4_reduce(U), fe4_reduce(V));
    return r;
}
#endif /* IFMA */
#endif
