/* ORACLE -- TEST INFRASTRUCTURE ONLY.
 * Edwards25519 points in extended coordinates + ristretto255 encode/decode/
 * Elligator (RFC 9496), and the variable-time multiscalar multiplication
 * with the same algorithm split curve25519-dalek 2.x uses behind
 * RistrettoPoint::optional_multiscalar_mul (the call at
 * /root/reference/src/range_proof/mod.rs:421): Straus with width-5 NAF below
 * 190 terms, Pippenger (w = 6/7/8) at and above.
 */
#ifndef ORACLE_GE_H
#define ORACLE_GE_H
#include "fe51.h"
#include "sc.h"
#include <stddef.h>

typedef struct { fe X, Y, Z, T; } ge_p3;       /* extended */
typedef struct { fe X, Y, Z, T; } ge_p1p1;     /* completed */
typedef struct { fe YpX, YmX, Z, T2d; } ge_cached; /* projective Niels */

void ge_init(void);                            /* load constants (idempotent) */
void ge_identity(ge_p3 *r);
void ge_add(ge_p3 *r, const ge_p3 *p, const ge_p3 *q);
void ge_sub(ge_p3 *r, const ge_p3 *p, const ge_p3 *q);
void ge_dbl(ge_p3 *r, const ge_p3 *p);
void ge_neg(ge_p3 *r, const ge_p3 *p);
void ge_to_cached(ge_cached *r, const ge_p3 *p);
void ge_add_cached(ge_p3 *r, const ge_p3 *p, const ge_cached *q);
void ge_sub_cached(ge_p3 *r, const ge_p3 *p, const ge_cached *q);
void ge_scalarmult(ge_p3 *r, const sc *s, const ge_p3 *p);
int ge_is_identity(const ge_p3 *p);            /* ristretto coset test X==0 || Y==0 */
int ge_ristretto_eq(const ge_p3 *p, const ge_p3 *q);

int ristretto_decompress(ge_p3 *r, const uint8_t s[32]);   /* 0 ok, -1 invalid */
void ristretto_compress(uint8_t s[32], const ge_p3 *p);
void ristretto_from_uniform_bytes(ge_p3 *r, const uint8_t b[64]);
extern const uint8_t RISTRETTO_BASEPOINT_COMPRESSED[32];

/* sum s_i * P_i ; algorithm chosen like the reference's dependency */
void ge_msm_vartime(ge_p3 *r, size_t n, const sc *scalars, const ge_p3 *points);
void ge_msm_straus(ge_p3 *r, size_t n, const sc *scalars, const ge_p3 *points);
const char *ge_backend(void);
void ristretto_decompress_many(ge_p3 *r, const uint8_t *const *in, int *rc, size_t n);   /* n x ristretto_decompress (4-way on SIMD builds) */   /* which field backend this build's Straus MSM uses */
void ge_msm_pippenger(ge_p3 *r, size_t n, const sc *scalars, const ge_p3 *points);
/* number of point additions+doublings the last ge_msm_* call on this thread did */
extern __thread uint64_t ge_op_counter;
#endif
