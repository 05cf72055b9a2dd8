/* ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's range-proof verification path
 * and of the arithmetic it calls.  It is the checker for the HIP engine and the
 * "port" CPU baseline of bench.py.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load liboracle.so; the product
 * (bulletproofs_amd / libbpgpu.so) never does.
 *
 * Pinning: checked against the reference's 16 golden proofs
 * (/root/reference/tests/range_proof.rs:16-95 -> tests/golden/rangeproof_v1.json),
 * the Merlin and generator known-answer values of SURVEY.md Appendix A, and an
 * independent pure-Python twin (oracle/py/bp_twin.py).  The reference itself is
 * Rust with un-vendored dependencies (curve25519-dalek ^2, merlin ^2, sha3 0.8:
 * Cargo.toml:21-31) and cannot be built here, so there is no oracle/_ref.
 *
 * Error codes mirror ProofError (/root/reference/src/errors.rs:12-54).
 */
#ifndef BP_ORACLE_H
#define BP_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_OK 0
#define ORACLE_ERR_VERIFICATION 1
#define ORACLE_ERR_FORMAT 2
#define ORACLE_ERR_INVALID_BITSIZE 3
#define ORACLE_ERR_INVALID_GENERATORS_LENGTH 4

typedef struct oracle_gens oracle_gens;

/* BulletproofGens::new (src/generators.rs:157-204) + PedersenGens::default (44-53) */
oracle_gens *oracle_gens_new(size_t gens_capacity, size_t party_capacity);
void oracle_gens_free(oracle_gens *g);
/* compressed encodings, party-major: G[party*gens_capacity + i] */
void oracle_gens_export(const oracle_gens *g, uint8_t *G_out, uint8_t *H_out, uint8_t B[32], uint8_t B_blinding[32]);

/* group / hash primitives */
int oracle_point_decompress_ok(const uint8_t in[32]);
void oracle_from_uniform_bytes(const uint8_t in[64], uint8_t out[32]);
void oracle_scalar_from_wide(const uint8_t in[64], uint8_t out[32]);
void oracle_scalar_mul(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]);
void oracle_scalar_invert(const uint8_t a[32], uint8_t out[32]);
void oracle_merlin_kat(const uint8_t *label, size_t label_len, const char *msg_label,
                       const uint8_t *msg, size_t msg_len, const char *ch_label, uint8_t *out, size_t out_len);
void oracle_shake256(const uint8_t *in, size_t n, uint8_t *out, size_t out_len);
void oracle_sha3_512(const uint8_t *in, size_t n, uint8_t out[64]);

/* vartime_multiscalar_mul on compressed inputs; out = compress(sum).
 * returns 1 when some point fails to decode (the Option::None case), else 0.
 * algo: 0 = reference split (Straus < 190 <= Pippenger), 1 = Straus, 2 = Pippenger. */
int oracle_msm(size_t n, const uint8_t *scalars, const uint8_t *points, int algo, uint8_t out[32]);
uint64_t oracle_last_msm_ops(void);
const char *oracle_backend(void);   /* "u64 5x51 serial", or the SIMD backend a -march=native build selected */

/* verify_multiple_with_rng (src/range_proof/mod.rs:345-452).
 * rng64: the 64 bytes the rng would yield to Scalar::random for the batching
 * challenge c (mod.rs:396).  msm_out (optional) receives compress(mega_check).
 * Returns ORACLE_OK or an ORACLE_ERR_* code. */
int oracle_verify(const oracle_gens *g, const uint8_t *proof, size_t proof_len,
                  const uint8_t *commitments, size_t m, size_t n,
                  const uint8_t *label, size_t label_len, const uint8_t rng64[64],
                  uint8_t msm_out[32]);
/* The same call with the caller's transcript (mod.rs:345-353 takes `transcript: &mut Transcript`, which may already
 * hold application messages and is left advanced).  state: 208 bytes = Strobe128 { state[200], pos, pos_begin,
 * cur_flags } + 5 zero bytes, read and written back.  oracle_transcript_* = Transcript::new / append_message /
 * challenge_bytes on that representation. */
int oracle_verify_ts(const oracle_gens *g, const uint8_t *proof, size_t proof_len,
                     const uint8_t *commitments, size_t m, size_t n,
                     uint8_t state[208], const uint8_t rng64[64], uint8_t msm_out[32]);
void oracle_transcript_new(const uint8_t *label, size_t label_len, uint8_t state[208]);
void oracle_transcript_append_message(uint8_t state[208], const char *label, const uint8_t *msg, size_t n);
void oracle_transcript_challenge_bytes(uint8_t state[208], const char *label, uint8_t *out, size_t n);
/* same transcript replay + scalar assembly, but returns the MSM terms instead:
 * scalars_out / points_out: N x 32 bytes, N = 2nm + 2lg(nm) + m + 6, order of mod.rs:422-443 */
int oracle_verify_terms(const oracle_gens *g, const uint8_t *proof, size_t proof_len,
                        const uint8_t *commitments, size_t m, size_t n,
                        const uint8_t *label, size_t label_len, const uint8_t rng64[64],
                        uint8_t *scalars_out, uint8_t *points_out, size_t *n_terms);

/* Non-constant-time prover used only to synthesise inputs
 * (prove_multiple_with_rng, src/range_proof/mod.rs:234-288; party.rs; dealer.rs;
 * inner_product_proof.rs:38-193).  Randomness = SHAKE256(seed) stream.
 * proof_out: 32*(9+2lg(nm)) bytes; commitments_out: m*32 bytes. */
int oracle_prove(const oracle_gens *g, const uint64_t *values, const uint8_t *blindings, size_t m, size_t n,
                 const uint8_t *label, size_t label_len, const uint8_t *seed, size_t seed_len,
                 uint8_t *proof_out, uint8_t *commitments_out);

int oracle_prove_ts(const oracle_gens *g, const uint64_t *values, const uint8_t *blindings, size_t m, size_t n,
                    uint8_t state[208], const uint8_t *seed, size_t seed_len, uint8_t *proof_out, uint8_t *commitments_out);

/* Stand-alone inner-product proof (src/inner_product_proof.rs).
 * oracle_ipp_verify = InnerProductProof::from_bytes(proof)?.verify(n, &mut Transcript::new(label), G_factors,
 * H_factors, &P, &Q, &G, &H) (ipp.rs:260-326, 373-407); points compressed, scalars 32-byte canonical;
 * msm_out (optional) = compress(expect_P - P).  Returns ORACLE_OK / ORACLE_ERR_*.
 * oracle_ipp_test_instance = the reference's own test_helper_create(n) (ipp.rs:433-497) with a SHAKE256(seed)
 * rng: fills proof (32*(2 lg n + 2) bytes), P, Q, G[n], H[n], G_factors[n] (= 1), H_factors[n] (= y^-i). */
int oracle_ipp_verify(size_t n, const uint8_t *proof, size_t proof_len, const uint8_t *label, size_t label_len,
                      const uint8_t *G_factors, const uint8_t *H_factors, const uint8_t P[32], const uint8_t Q[32],
                      const uint8_t *G, const uint8_t *H, uint8_t msm_out[32]);
/* InnerProductProof::create(&mut Transcript::new(label), &Q, G_factors = 1, H_factors, G, H, a, b).to_bytes()
 * (ipp.rs:38-193): proof_out = 32 * (2 lg n + 2) bytes. */
/* InnerProductProof::from_bytes + verification_scalars (ipp.rs:198-253) on the caller's transcript */
int oracle_ipp_verification_scalars(size_t n, const uint8_t *proof, size_t proof_len, uint8_t state[208], uint8_t *u_sq,
                                    uint8_t *u_inv_sq, uint8_t *s);
int oracle_ipp_create(size_t n, const uint8_t *label, size_t label_len, const uint8_t Q[32], const uint8_t *Hf,
                      const uint8_t *G, const uint8_t *H, const uint8_t *a, const uint8_t *b, uint8_t *proof_out);
int oracle_ipp_test_instance(size_t n, const uint8_t *label, size_t label_len, const uint8_t *seed, size_t seed_len,
                             uint8_t *proof_out, uint8_t P_out[32], uint8_t Q_out[32], uint8_t *G_out, uint8_t *H_out,
                             uint8_t *Gf_out, uint8_t *Hf_out);

/* The MPC messages of prove_multiple (messages.rs:23-56; party.rs, dealer.rs) and the dealer's per-share audit
 * (ProofShare::audit_share, messages.rs:85-167 -- the blame path of Dealer::receive_shares, dealer.rs:303-335).
 * oracle_prove_shares: the same proof bytes and commitments as oracle_prove, plus per party j: bit_commitments
 * (V_j, A_j, S_j: 96 bytes), poly_commitments (T_1_j, T_2_j: 64), shares (t_x, t_x_blinding, e_blinding, l_vec[n],
 * r_vec[n]: 32 (3 + 2n)), and the challenges (y, z, x).  Not thread-safe (test infrastructure).
 * Pinning: the reference holds no fixed vector for these messages (its MPC tests are random round trips, mod.rs:726-841);
 * the share export is part of the prover that is pinned through the 16 golden proofs' verifier (its proofs verify), and
 * the audit is pinned by consistency: the honest shares of such a proof audit to two identity points, the shares of a party
 * that committed to an out-of-range value do not (the reference's detect_dishonest_party_during_aggregation scenario).
 * oracle_audit_share: 0 = Ok(()), 1 = Err(()); out2 (optional, 64 bytes) = compress(P_check), compress(t_check). */
int oracle_prove_shares(const oracle_gens *g, const uint64_t *values, const uint8_t *blindings, size_t m, size_t n,
                        const uint8_t *label, size_t label_len, const uint8_t *seed, size_t seed_len,
                        uint8_t *proof_out, uint8_t *commitments_out, uint8_t *bit_commitments, uint8_t *poly_commitments,
                        uint8_t *shares, uint8_t challenges[96]);
int oracle_audit_share(const oracle_gens *g, size_t n, size_t j, const uint8_t *share, const uint8_t bit_commitment[96],
                       const uint8_t poly_commitment[64], const uint8_t challenges[96], uint8_t *out2);

/* LinearProof (src/linear_proof.rs; public type, `pub use` lib.rs:36): proves <a, b> = c for secret a and public b.
 * oracle_linear_create = LinearProof::create(transcript, rng, &C, r, a, b, G, &F, &B).to_bytes() (linear_proof.rs:40-173):
 * state = the caller's transcript (208-byte form, not written back); rng = the bytes the rng yields, 64 per
 * Scalar::random in draw order (s_j, t_j per round; s_star, t_star): 64 * (2 lg n + 2); proof_out: 32 * (2 lg n + 3).
 * Returns 0; 5 = InvalidInputLength (n not a power of two); 1 = a point does not decode.
 * oracle_linear_verify = LinearProof::from_bytes(proof)?.verify(transcript, &C, &G, &F, &B, b) (:175-236, 350-394);
 * msm_out (optional) = compress(expect_S - S).  No reference test holds a fixed vector for this type (its tests are
 * random round trips, :397-488): parity here is pinned by the independent Python twin only. */
int oracle_linear_create(size_t n, const uint8_t state[208], const uint8_t *rng, const uint8_t C[32], const uint8_t r[32],
                         const uint8_t *a, const uint8_t *b, const uint8_t *G, const uint8_t F[32], const uint8_t B[32],
                         uint8_t *proof_out);
int oracle_linear_verify(size_t n, const uint8_t *proof, size_t proof_len, const uint8_t state[208], const uint8_t C[32],
                         const uint8_t *G, const uint8_t F[32], const uint8_t B[32], const uint8_t *b, uint8_t msm_out[32]);

/* Batch drivers (independent proofs, equal shape), `threads` worker threads.
 * verdicts[i] = error code.  Returns wall seconds. */
double oracle_verify_batch(const oracle_gens *g, size_t nbatch, const uint8_t *proofs, size_t proof_len,
                           const uint8_t *commitments, size_t m, size_t n,
                           const uint8_t *label, size_t label_len, const uint8_t *rng64s,
                           uint8_t *verdicts, uint8_t *msm_outs, int threads);
double oracle_prove_batch(const oracle_gens *g, size_t nbatch, const uint64_t *values, const uint8_t *blindings,
                          size_t m, size_t n, const uint8_t *label, size_t label_len,
                          const uint8_t *seed, size_t seed_len, uint8_t *proofs_out, uint8_t *commitments_out,
                          int threads);
double oracle_msm_batch(size_t nbatch, size_t n, const uint8_t *scalars, const uint8_t *points, int algo,
                        uint8_t *outs, uint8_t *status, int threads);
#ifdef __cplusplus
}
#endif
#endif
