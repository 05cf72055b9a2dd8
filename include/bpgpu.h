/* bpgpu.h -- C ABI of the MI355X-native Bulletproofs verification engine
 * (libbpgpu.so).
 *
 * The reference (dalek-cryptography/bulletproofs, Rust) has no FFI layer: its
 * verification hot path is one call into a dependency,
 *     RistrettoPoint::optional_multiscalar_mul(scalars, points)
 * made at src/range_proof/mod.rs:421-445 (range proofs),
 * src/inner_product_proof.rs:308-319 (stand-alone inner-product proofs) and
 * src/r1cs/verifier.rs:459-491 (R1CS).  This header is the boundary a Rust
 * maintainer would bind with `extern "C"` to move that call -- and, one level
 * up, whole batches of RangeProof::verify_multiple -- onto the GPU.  Every
 * entry point below names the reference interface it replaces.  INTEGRATION.md
 * shows the Rust-side binding.
 *
 * Conventions
 *   - plain pointers and sizes only; caller owns every buffer; nothing is
 *     retained after a call returns (except what *_load / *_create upload).
 *   - scalars : 32 bytes little-endian, canonical (< l), as Scalar::as_bytes().
 *   - points  : 32 bytes, CompressedRistretto encodings (RFC 9496).
 *   - return  : BPGPU_OK or a negative BPGPU_ERR_* code; per-item results in
 *     status / verdict byte arrays.  The library never aborts the process.
 *   - entry points without suffix take HOST pointers and do H2D/D2H themselves;
 *     `_dev` twins take DEVICE pointers plus a hipStream_t (as void*; NULL =
 *     the context's own stream), enqueue asynchronously and return.
 *   - one context serves one device; calls on one context are serialised by an
 *     internal mutex (the reference's calls are &self and re-entrant; use one
 *     context per host thread for concurrency).  A context owns ONE set of device
 *     scratch buffers: work enqueued on it is ordered.  If consecutive `_dev` calls
 *     on one context name different streams, the library makes the later stream
 *     wait (hipStreamWaitEvent) for the earlier call's kernels, so results stay
 *     correct -- but only one context per stream gives overlap.
 *   - throughput: a batch is a chain of a few dependent launches, several of them
 *     narrow, so one stream cannot fill the device.  Keep many batches in flight:
 *     one context per HIP stream (the generator tables are shared by the contexts
 *     of a process), ~48 streams, and GPU_MAX_HW_QUEUES=16 in the environment
 *     (ROCm's default of 4 hardware queues serialises the streams).
 */
#ifndef BPGPU_H
#define BPGPU_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BPGPU_OK 0
#define BPGPU_ERR_INVALID_ARG (-1)
#define BPGPU_ERR_HIP (-2)         /* a HIP runtime call failed; see bpgpu_last_error */
#define BPGPU_ERR_NO_GENS (-3)     /* generators not loaded, or too few for the request */
#define BPGPU_ERR_NO_DEVICE (-4)
#define BPGPU_ERR_BAD_GENERATOR (-5) /* a generator encoding passed to bpgpu_gens_load does not decode */
#define BPGPU_ERR_HW_QUEUES (-6)   /* bpgpu_pool_create: GPU_MAX_HW_QUEUES is set to a value the pool's lanes cannot overlap on */

/* per-MSM status (msm entry points) */
#define BPGPU_MSM_OK 0
#define BPGPU_MSM_BAD_POINT 1      /* some point failed to decode: optional_multiscalar_mul -> None */
#define BPGPU_MSM_BAD_SCALAR 2     /* some scalar is not canonical */

/* per-proof verdicts: ProofError of src/errors.rs:12-54 */
#define BPGPU_VERDICT_OK 0
#define BPGPU_VERDICT_VERIFICATION_ERROR 1
#define BPGPU_VERDICT_FORMAT_ERROR 2
#define BPGPU_VERDICT_INVALID_BITSIZE 3
#define BPGPU_VERDICT_INVALID_GENERATORS_LENGTH 4

typedef struct bpgpu_ctx bpgpu_ctx;

/* ---- context ------------------------------------------------------------- */
int bpgpu_version(void);
/* device: HIP device ordinal.  Fails with BPGPU_ERR_NO_DEVICE when no GPU is usable
 * (there is no CPU fallback). */
int bpgpu_ctx_create(int device, bpgpu_ctx **out);
void bpgpu_ctx_destroy(bpgpu_ctx *ctx);
const char *bpgpu_last_error(bpgpu_ctx *ctx);
/* Tunables (set before bpgpu_gens_*):
 *   "fixed_window_bits"     window W of the generator tables, 2..20; 0 (default) = the W with the fewest windows whose
 *                           table (n_gens * ceil(255/W) * 2^(W-1) * 128 bytes) fits fixed_table_max_bytes
 *   "fixed_table_max_bytes" HBM budget of the tables (default 160 GiB of the MI355X's 288 GB; a context settles for a
 *                           smaller window when the allocation fails)
 *   "fixed_splits"          workgroups the generator terms of one proof block are split over (0 = auto)
 *   "bucket_min_terms"      variable-base terms per MSM from which the bucket (Pippenger) path is taken instead of the
 *                           table-lookup one (default 1536; the batch-combined check, one MSM over all proofs' terms,
 *                           switches at 32768 terms unless this option is set; a huge value disables the bucket
 *                           path, 1 forces it).  Results are bit-identical either way.
 *   "host_sync_blocking"    1: host-pointer entry points wait for their results on a blocking event (the calling
 *                           thread sleeps: right for many host threads, one context each); 0 (default): spin-wait
 *   "prover_constant_time"  1: the prover's secret-dependent commitments through the constant-time walk (bpgpu_rangeproof_prove_batch)
 *   "transcript_script"     0: byte-wise transcript replay instead of the per-shape script (csrc/rp_script.h; for A/B only)
 *   "horner_lanes"          lanes per Horner chain of the proof-specific terms in the range-proof path:
 *                           1 (one lane per proof: least work, longest chain; on chains of >= 2048 proofs it runs on the context's
 *                           second stream beside the table walk), 4 (a quad per proof: twice the instructions, half the latency),
 *                           64 (one wavefront per chain: lowest latency of a single small batch),
 *                           0 = auto (default): 64 up to 256 proofs (a small batch alone is a latency matter: 0.61 instead of 0.85 ms
 *                           for a single proof); 1 on chains of >= 8192 proofs, and on chains of >= 2048 when the pool knows that
 *                           other chains run beside them; 4 otherwise
 *   "a_outside"             1 (default): on chains of >= 2048 proofs A, whose coefficient is 1, is added after the Horner chain instead of
 *                           going through a table and the window sums; 0: like every other point (for A/B)
 *   "per_proof_radix"       radix of the proofs' own points: 0 / 16 (default), 32 (16-entry tables, 51 windows; takes effect on chains
 *                           of >= 2048 proofs; measured +1 % steady, -4 % on bursts)
 *   "split_stage3"          -1 (default): the window sums as their own launch on chains of >= 2048 proofs; 0 / 1: never / always
 *   "transcript_coop"       1 (default): launch chains of up to 256 proofs replay their transcripts 32 lanes per proof (Keccak-f[1600]
 *                           with one state word per lane: one blocking call of 1 proof 0.62 -> 0.53 ms); 0: one lane per proof everywhere
 *   narrow chains (up to 256 proofs: the crate's own call shape, the combining queue's small chains; round 6):
 *   "coop_split"            1 (default): launch 1 stages the proof, the script and its masks in LDS, the 5 + k challenges are reduced on
 *                           5 + k lanes, the k + 1 inversions run on k + 1 lanes at once, the basepoint coefficients are a role of launch 3;
 *                           0: the group's leader does all of it (k_rp_stage1_coop 227 -> 157 us for one proof)
 *   "exp_single"            1 (default): the generator-exponent role with one index per lane (launch 3: 73 -> 33 us); 0: as wide chains
 *   "narrow_chunk"          per-proof points per (chunk, window) lane of launch 3: 0 = 8 (default), 2 .. 32; 32 = one chunk
 *   "narrow_walk"           1 (default): the table walk with lane = split, a proof's partial sums folded inside launch 4 (finish 40 -> 5 us)
 *   "narrow_hi_max"         chains of up to this many proofs (default 32; 0: never) give every per-proof point a second table, of its
 *                           2^128 multiple (a wavefront per point, beside the transcript), and run a 32-window Horner chain
 *   "narrow_hi4_max"        chains of up to this many proofs (default 4; 0 = never) take THREE more tables per point (2^64 P, 2^128 P, 2^192 P) and a
 *                           16-window chain (one blocking call 0.315 -> 0.29 ms)
 *   "narrow_fused_finish"   1 (default): verdict-only calls, chains of 8 .. 256 proofs: the last workgroup of a proof in launch 4 adds up its pieces
 *                           and writes the verdict (no finish launch); 0: k_finish1
 *   "msm_narrow"            1 (default): bpgpu_msm_batch with at most 16 MSMs of at most 768 terms in all (the boundary function called from one
 *                           thread with a Straus-size MSM) runs three launches: decode beside a wavefront per term that builds the table of
 *                           2^128 P, window sums in ~sqrt(N) chunks, one tail (32-window Horner chain + encoding); 0: the batch form
 *   "msm_fork"              1 (default): bpgpu_msm_batch_shared's generator half on the context's second stream beside the per-MSM points; 0: one stream
 *   "bucket_chain"          0 (default): MSMs of up to 6144 variable-base terms take the fused bucket chain (csrc/bucket2.h: decode, one LDS
 *                           sort + accumulate workgroup per (MSM, window), one tail launch); 1: bucket.h's chain everywhere (for A/B)
 *   "bucket_lanes"          lanes of a (MSM, window) workgroup of the fused chain: 0 = by batch width (default), 64, 128, 256
 *   "bucket_fast_tail"      -1 (default): batches of fewer than 48 MSMs end with the short-chain tail (leaves of 4 buckets, shuffle sums);
 *                           0 / 1: never / always
 *   "exponent_pairs"        1 (default): the generator-exponent role handles index i together with nm-1-i (they share s_i and s_i^-1: -24 % of
 *                           the role's Montgomery products); 0: four consecutive indices per lane (for A/B)
 *   "fb_walk_waves"         wavefronts the generator half of a fused chain is cut into (0 = 2048)
 * get_option additionally answers "fixed_table_bytes" and the effective "fixed_window_bits".
 * Returns BPGPU_ERR_INVALID_ARG for unknown keys. */
int bpgpu_ctx_set_option(bpgpu_ctx *ctx, const char *key, int64_t value);
int bpgpu_ctx_get_option(bpgpu_ctx *ctx, const char *key, int64_t *value);
int bpgpu_synchronize(bpgpu_ctx *ctx);

/* ---- generators ------------------------------------------------------------
 * Replace BulletproofGens::new(gens_capacity, party_capacity)
 * (src/generators.rs:157-204) and PedersenGens::default() (generators.rs:44-53)
 * as held by a verifier.  Both build the fixed-base window tables in HBM. */
/* derive the generators on the device exactly as the reference does
 * (SHAKE256("GeneratorsChain" || 'G'/'H' || u32le(party)) -> from_uniform_bytes;
 *  B = basepoint, B_blinding = hash_from_bytes::<Sha3_512>(compress(B))). */
int bpgpu_gens_create(bpgpu_ctx *ctx, size_t gens_capacity, size_t party_capacity);
/* or load them from encodings the caller already has; G/H party-major:
 * G[(party * gens_capacity + i) * 32], as BulletproofGens::G_vec[party][i]. */
int bpgpu_gens_load(bpgpu_ctx *ctx, size_t gens_capacity, size_t party_capacity,
                    const uint8_t *G, const uint8_t *H, const uint8_t B[32], const uint8_t B_blinding[32]);
/* read back the encodings (any pointer may be NULL) */
int bpgpu_gens_export(bpgpu_ctx *ctx, uint8_t *G, uint8_t *H, uint8_t B[32], uint8_t B_blinding[32]);
/* A service that verifies TWO shapes (say m = 16 and m = 1 proofs) on one generator set: BulletproofGens::new(gens_capacity,
 * party_capacity) serves every n <= gens_capacity, m <= party_capacity (src/generators.rs:157-259), but the window table that replaces
 * the doublings is sized for the whole set -- at (64, 16) that is W = 16, and m = 1 proofs walked it 9 % slower than through their own
 * W = 20 table.  This adds a SECOND table over the sub-set (n2, m2) of the loaded generators and re-balances both windows under the
 * context's one budget ("fixed_table_max_bytes"): the pair minimising nwin(W1) / nwin(W1 alone) + nwin(W2) / nwin(W2 alone) -- under
 * 160 GiB, (64, 16) + (64, 1): W = 15 (73 GB) + W = 19 (61 GB).  Range proofs with n <= n2 and m <= m2 then walk the secondary table
 * (results are the same group elements: bit-identical verdicts and encodings).  Options (get): "secondary_window_bits",
 * "secondary_table_bytes", "secondary_shape_n" / "_m"; "fixed_window_bits" reports the re-balanced primary window.  Calling it again
 * replaces the secondary shape; bpgpu_gens_create / _load drop it. */
int bpgpu_gens_add_shape(bpgpu_ctx *ctx, size_t n2, size_t m2);

/* ---- multiscalar multiplication ---------------------------------------------
 * bpgpu_msm_batch: nbatch independent calls of
 *   RistrettoPoint::optional_multiscalar_mul(scalars_b, points_b.decompress())
 * (trait curve25519_dalek::traits::VartimeMultiscalarMul; call sites
 * range_proof/mod.rs:421, inner_product_proof.rs:308, r1cs/verifier.rs:459).
 *   n_terms[b]  : number of (scalar, point) pairs of MSM b (host array, may be 0)
 *   scalars     : sum(n_terms) x 32 bytes, MSM after MSM
 *   points      : sum(n_terms) x 32 bytes
 *   out         : nbatch x 32 bytes = compress(sum_i scalars[i] * points[i]);
 *                 all-zero when status[b] != 0
 *   status      : nbatch bytes, BPGPU_MSM_* */
int bpgpu_msm_batch(bpgpu_ctx *ctx, size_t nbatch, const uint32_t *n_terms,
                    const uint8_t *scalars, const uint8_t *points, uint8_t *out, uint8_t *status);
int bpgpu_msm_batch_dev(bpgpu_ctx *ctx, size_t nbatch, const uint32_t *n_terms_host,
                        const void *d_scalars, const void *d_points, void *d_out, void *d_status, void *stream);

/* bpgpu_msm_batch_shared: the verification "mega-check" shape
 * (range_proof/mod.rs:421-443): every MSM of the batch is
 *     sum_{g < 2nm+2} gen_scalars[b][g] * Gen_g  +  sum_{u < n_unique} uniq_scalars[b][u] * uniq_points[b][u]
 * with Gen = (B_blinding, B, G(n,m)..., H(n,m)...) taken from the loaded
 * generators in the reference's order (mod.rs:439-442).  The generator terms
 * use the precomputed tables; results are bit-identical to bpgpu_msm_batch on
 * the same terms.  Requires n <= gens_capacity, m <= party_capacity. */
int bpgpu_msm_batch_shared(bpgpu_ctx *ctx, size_t n, size_t m, size_t nbatch, size_t n_unique,
                           const uint8_t *gen_scalars, const uint8_t *uniq_scalars, const uint8_t *uniq_points,
                           uint8_t *out, uint8_t *status);
int bpgpu_msm_batch_shared_dev(bpgpu_ctx *ctx, size_t n, size_t m, size_t nbatch, size_t n_unique,
                               const void *d_gen_scalars, const void *d_uniq_scalars, const void *d_uniq_points,
                               void *d_out, void *d_status, void *stream);

/* ---- Merlin transcripts across the boundary ------------------------------------------
 * The reference's verifiers take `transcript: &mut Transcript` (range_proof/mod.rs:345-353,
 * inner_product_proof.rs:260-270): it may already hold application messages and is left advanced.
 * A transcript crosses this ABI as its 208-byte state, which is the in-memory image of merlin's
 * Strobe128 { state: [u8; 200] (align 8), pos: u8, pos_begin: u8, cur_flags: u8 }:
 *   [0,200) sponge bytes, [200] pos, [201] pos_begin, [202] cur_flags, [203,208) zero.
 * The three helpers run on the host (no GPU needed) and are Transcript::new / append_message /
 * challenge_bytes of src/transcript.rs's dependency (merlin 2.x) on that representation. */
#define BPGPU_TRANSCRIPT_BYTES 208
int bpgpu_transcript_new(const uint8_t *label, size_t label_len, uint8_t state[BPGPU_TRANSCRIPT_BYTES]);
int bpgpu_transcript_append_message(uint8_t state[BPGPU_TRANSCRIPT_BYTES], const uint8_t *label, size_t label_len,
                                    const uint8_t *msg, size_t msg_len);
int bpgpu_transcript_challenge_bytes(uint8_t state[BPGPU_TRANSCRIPT_BYTES], const uint8_t *label, size_t label_len,
                                     uint8_t *out, size_t out_len);

/* ---- range-proof verification ---------------------------------------------------
 * nbatch independent calls of
 *   RangeProof::from_bytes(proof)?.verify_multiple_with_rng(bp_gens, pc_gens,
 *        &mut Transcript::new(label), &commitments, n, rng)
 * (src/range_proof/mod.rs:345-452, 504-538) with all proofs of one shape (n, m).
 *   proofs       : nbatch x proof_len bytes, proof_len = 32*(9 + 2*lg(n*m)) for valid input
 *   commitments  : nbatch x m x 32 bytes (value commitments V_j)
 *   label        : Merlin transcript label shared by the batch
 *   rng64        : nbatch x 64 bytes = what the rng would hand Scalar::random for the
 *                  batching challenge c (mod.rs:396); NULL = the library's thread_rng(): ONE 32-byte key per launch chain from a
 *                  per-thread ChaCha20 generator keyed by the OS CSPRNG, expanded on the device -- proof p's 64 bytes are block p
 *                  of ChaCha20(key) (nothing per proof is drawn or copied on the host)
 *   verdict      : nbatch bytes, BPGPU_VERDICT_*  (Ok(()) == 0)
 *   msm_out      : optional nbatch x 32 bytes, compress(mega_check) for parity tests */
int bpgpu_rangeproof_verify_batch(bpgpu_ctx *ctx, size_t n, size_t m, size_t nbatch,
                                  const uint8_t *proofs, size_t proof_len, const uint8_t *commitments,
                                  const uint8_t *label, size_t label_len, const uint8_t *rng64,
                                  uint8_t *verdict, uint8_t *msm_out);
int bpgpu_rangeproof_verify_batch_dev(bpgpu_ctx *ctx, size_t n, size_t m, size_t nbatch,
                                      const void *d_proofs, size_t proof_len, const void *d_commitments,
                                      const uint8_t *label, size_t label_len, const void *d_rng64,
                                      void *d_verdict, void *d_msm_out, void *stream);

/* Asynchronous form of bpgpu_rangeproof_verify_batch for a host thread that keeps several contexts busy: the inputs are
 * staged (the caller may reuse its input buffers at once), the work is enqueued, the call returns.  `verdict` / `msm_out`
 * are filled by bpgpu_ctx_collect(ctx) -- or implicitly by the next call made on this context -- and must stay valid
 * until then.  One submitted call per context at a time.  A single thread cycling over ~32 contexts reaches the
 * throughput of one thread per context (tools/archive/host_api_rate.py --pipelined). */
int bpgpu_rangeproof_verify_batch_submit(bpgpu_ctx *ctx, size_t n, size_t m, size_t nbatch,
                                         const uint8_t *proofs, size_t proof_len, const uint8_t *commitments,
                                         const uint8_t *label, size_t label_len, const uint8_t *rng64,
                                         uint8_t *verdict, uint8_t *msm_out);
int bpgpu_ctx_collect(bpgpu_ctx *ctx);

/* The same verification on caller-supplied transcripts (the full `&mut Transcript` semantics of mod.rs:345-353):
 *   transcripts       : transcript_stride == 0: ONE state (208 bytes) every proof of the batch starts from;
 *                       transcript_stride == BPGPU_TRANSCRIPT_BYTES: nbatch states, one per proof
 *   transcripts_out   : optional nbatch x 208 bytes: each proof's transcript as verify_multiple_with_rng leaves it
 *                       (after the last inner-product challenge).  A proof rejected by from_bytes or by the parameter
 *                       checks (FormatError, InvalidBitsize, InvalidGeneratorsLength) gets its input state back; a proof
 *                       with an identity A / S / T_1 / T_2 / L_i / R_i gets the state as of that message (nothing of it
 *                       absorbed: validate_and_append_point, transcript.rs:75-87); n m != 2^k leaves the state as of the
 *                       `w` challenge (verification_scalars, ipp.rs:203-211) -- on every path the reference's own state.
 * `_dev`: `shared_transcript` is a HOST pointer to one state (or NULL), `d_transcripts` a device pointer to nbatch
 * states (or NULL); exactly one of the two must be given. */
int bpgpu_rangeproof_verify_batch_ts(bpgpu_ctx *ctx, size_t n, size_t m, size_t nbatch,
                                     const uint8_t *proofs, size_t proof_len, const uint8_t *commitments,
                                     const uint8_t *transcripts, size_t transcript_stride, const uint8_t *rng64,
                                     uint8_t *verdict, uint8_t *msm_out, uint8_t *transcripts_out);
int bpgpu_rangeproof_verify_batch_ts_dev(bpgpu_ctx *ctx, size_t n, size_t m, size_t nbatch,
                                         const void *d_proofs, size_t proof_len, const void *d_commitments,
                                         const uint8_t *shared_transcript, const void *d_transcripts, const void *d_rng64,
                                         void *d_verdict, void *d_msm_out, void *d_transcripts_out, void *stream);

/* ---- batch-combined verification (ADDITIONAL entry point; not a call of the reference) ---------------
 * The reference verifies one proof per MSM (verify_multiple checks ONE aggregated proof: mod.rs:345-452).
 * For a batch of independent proofs the standard next step is a random linear combination:
 *     R = sum_i rho_i * MegaCheck_i ,  MegaCheck_i = the multiscalar multiplication of mod.rs:421-443 for proof i,
 * with one weight rho_i = Scalar::from_bytes_mod_order_wide(weights64[i]) per proof.  The 2nm+2 generator
 * coefficients of all proofs add up in the scalar field, so the table walk runs once per batch.  R is the
 * identity when every combined proof verifies; if one does not, R != identity except with probability ~2^-252
 * over the weights, which must be unpredictable to the provers (NULL = drawn by the library as for rng64: uniform 512-bit strings
 * expanded on the device from a per-chain key).  Weights a caller passes are valid at any width; SHORT ones (e.g. 128 bits,
 * zero-extended) give proof i's A term the bare weight as its coefficient, and those equal-length scalars crowd single buckets of
 * the combined MSM -- which the bucket stage handles (crowded buckets are summed 64 lanes at a time: +4 % per combination,
 * measured at 4096 proofs).
 *   verdict   : nbatch bytes.  Proofs rejected by the parser / point decoder get their BPGPU_VERDICT_* code and
 *               are left out of the combination.  The others get 0 when R is the identity.  Otherwise:
 *               - bpgpu_rangeproof_verify_rlc re-verifies the batch proof by proof (same rng64) and returns
 *                 exactly the verdicts of bpgpu_rangeproof_verify_batch;
 *               - the asynchronous _dev variant marks them BPGPU_VERDICT_UNDECIDED and leaves that to the caller.
 *   batch_out : optional 33 bytes: [0] = 0 if R is the identity else 1; [1..33) = compress(R) (parity tests) */
#define BPGPU_VERDICT_UNDECIDED 5
int bpgpu_rangeproof_verify_rlc(bpgpu_ctx *ctx, size_t n, size_t m, size_t nbatch,
                                const uint8_t *proofs, size_t proof_len, const uint8_t *commitments,
                                const uint8_t *label, size_t label_len, const uint8_t *rng64,
                                const uint8_t *weights64, uint8_t *verdict, uint8_t *batch_out);
int bpgpu_rangeproof_verify_rlc_dev(bpgpu_ctx *ctx, size_t n, size_t m, size_t nbatch,
                                    const void *d_proofs, size_t proof_len, const void *d_commitments,
                                    const uint8_t *label, size_t label_len, const void *d_rng64,
                                    const void *d_weights64, void *d_verdict, void *d_batch_out, void *stream);

/* ---- stand-alone inner-product proofs -------------------------------------------
 * nbatch independent calls of
 *   InnerProductProof::from_bytes(proof)?.verify(n, &mut Transcript::new(label), G_factors, H_factors, &P, &Q, &G, &H)
 * (src/inner_product_proof.rs:260-326, 373-407), all of one size n.
 *   proofs     : nbatch x proof_len bytes, proof_len = 32*(2*lg(n) + 2) for valid input
 *   G_factors, H_factors : nbatch x n x 32 bytes (the iterators of verify())
 *   P, Q       : nbatch x 32 bytes; G, H : nbatch x n x 32 bytes (compressed points)
 *   verdict    : nbatch bytes, BPGPU_VERDICT_* (an undecodable point counts as VerificationError)
 *   msm_out    : optional nbatch x 32 bytes, compress(expect_P - P) for parity tests */
int bpgpu_ipp_verify_batch(bpgpu_ctx *ctx, size_t n, size_t nbatch, const uint8_t *proofs, size_t proof_len,
                           const uint8_t *label, size_t label_len, const uint8_t *G_factors, const uint8_t *H_factors,
                           const uint8_t *P, const uint8_t *Q, const uint8_t *G, const uint8_t *H,
                           uint8_t *verdict, uint8_t *msm_out);

/* Device-pointer variant.  bases_shared != 0: every proof of the batch uses the SAME generator vectors, as the
 * reference's callers do (G = bp_gens.G(n, m), H = bp_gens.H(n, m)): d_G and d_H then hold n x 32 bytes instead of
 * nbatch x n x 32.  transcript: shared_transcript (host, 208 bytes, may hold earlier messages) if not NULL, else
 * Transcript::new(label). */
int bpgpu_ipp_verify_batch_dev(bpgpu_ctx *ctx, size_t n, size_t nbatch, const void *d_proofs, size_t proof_len,
                               const uint8_t *label, size_t label_len, const uint8_t *shared_transcript,
                               const void *d_G_factors, const void *d_H_factors, const void *d_P, const void *d_Q,
                               const void *d_G, const void *d_H, int bases_shared,
                               void *d_verdict, void *d_msm_out, void *stream);

/* ---- linear proofs (LinearProof, `pub use` src/lib.rs:36) --------------------------------
 * nbatch independent calls of
 *   LinearProof::from_bytes(proof)?.verify(&mut transcript, &C, &G, &F, &B, b_vec)
 * (src/linear_proof.rs:175-236, 240-312, 350-394), all of one size n = G.len() = b_vec.len(): the proof that
 * <a, b> = c for a committed secret a and the public b.  The three multiscalar multiplications and the comparison of
 * verify() (:207-236) run as ONE multiscalar multiplication of n + 2 lg(n) + 4 terms per proof whose result must be
 * the identity; transcript replay, the fold of b and the subset products run on the device (csrc/linear.h).
 *   proofs  : nbatch x proof_len bytes, proof_len = 32*(2*lg(n) + 3) for valid input (L_j, R_j pairs, S, a, r)
 *   transcript : shared_transcript (host, 208 bytes, may hold earlier messages) if not NULL, else Transcript::new(label);
 *             every proof starts from it
 *   transcripts_out : optional nbatch x 208 bytes: each proof's transcript as verify() leaves it (after the x_star
 *             challenge, :208) when the proof reaches the final check; for proofs rejected earlier the state right after
 *             innerproduct_domain_sep(n) (:196)
 *   C       : nbatch x 32 bytes (compressed commitments)
 *   G       : n x 32 bytes, F, B : 32 bytes each -- compressed points shared by the batch (the reference's callers pass
 *             bp_gens.share(0).G(n), pedersen B and B_blinding: linear_proof.rs:405-411); the encodings given here are
 *             what the transcript absorbs (G_i.compress(), :201-205), so they must be canonical
 *             G = F = B = NULL: the bases are the context's generators -- G = bp_gens.share(0).G(n), F = pc_gens.B,
 *             B = pc_gens.B_blinding -- and their n + 2 coefficients go through the fixed-base window tables instead of
 *             per-call point tables (the fast path; needs bpgpu_gens_create / _load with gens_capacity >= n)
 *   b       : nbatch x n x 32 bytes canonical scalars, or n x 32 when b_shared != 0
 *   verdict : nbatch bytes, BPGPU_VERDICT_* (FormatError: proof length, non-canonical a, r or b_i; VerificationError:
 *             n != 2^lg_n, an identity L_j / R_j, an undecodable point, or expect_S != S)
 *   msm_out : optional nbatch x 32 bytes, compress(expect_S - S) for parity tests */
int bpgpu_linear_verify_batch(bpgpu_ctx *ctx, size_t n, size_t nbatch, const uint8_t *proofs, size_t proof_len,
                              const uint8_t *label, size_t label_len, const uint8_t *shared_transcript,
                              const uint8_t *C, const uint8_t *G, const uint8_t *F, const uint8_t *B,
                              const uint8_t *b, int b_shared, uint8_t *verdict, uint8_t *msm_out, uint8_t *transcripts_out);
int bpgpu_linear_verify_batch_dev(bpgpu_ctx *ctx, size_t n, size_t nbatch, const void *d_proofs, size_t proof_len,
                                  const uint8_t *label, size_t label_len, const uint8_t *shared_transcript,
                                  const void *d_C, const void *d_G, const void *d_F, const void *d_B,
                                  const void *d_b, int b_shared, void *d_verdict, void *d_msm_out, void *d_transcripts_out,
                                  void *stream);

/* nbatch independent calls of
 *   LinearProof::create(&mut transcript, &mut rng, &C, r, a_vec, b_vec, G_vec, &F, &B).to_bytes()
 * (src/linear_proof.rs:40-173), all of one size n (a power of two) over the same G, F, B.  Per round the reference forms
 * L_j, R_j with two (n'+2)-term multiscalar multiplications (:104-117) and folds the generators with n' two-term ones
 * (:140-144); here all proofs advance round by round together and every L_j / R_j / S is one multiscalar
 * multiplication over the ORIGINAL points (csrc/linear_prover.h): the same group elements, so the proofs are
 * byte-identical to the reference algorithm's given the same transcript, inputs and randomness.  Variable time, like
 * the reference's own create() (vartime_multiscalar_mul).
 *   rng     : nbatch x 64*(2 lg n + 2) bytes: what the rng would yield to Scalar::random, 64 bytes per draw, in the
 *             reference's draw order (s_j, t_j per round, then s_star, t_star); NULL = the OS CSPRNG
 *   C, r    : nbatch x 32 bytes: the commitment (absorbed into the transcript as given) and its blinding factor
 *   a       : nbatch x n x 32 canonical scalars (secret); b : nbatch x n x 32, or n x 32 when b_shared != 0 (public)
 *   G       : n x 32 bytes; F, B : 32 bytes (compressed points, shared by the batch); G = F = B = NULL: the context's
 *             generators as in bpgpu_linear_verify_batch -- every L_j, R_j, S is then a pure window-table MSM
 *   proofs_out : nbatch x 32*(2 lg n + 3) bytes; status_out : nbatch bytes (BPGPU_MSM_OK, or why no proof was made:
 *             an undecodable point or a non-canonical scalar); transcripts_out : optional nbatch x 208 bytes, each
 *             proof's transcript as create() leaves it */
int bpgpu_linear_create_batch(bpgpu_ctx *ctx, size_t n, size_t nbatch, const uint8_t *label, size_t label_len,
                              const uint8_t *shared_transcript, const uint8_t *rng, const uint8_t *C, const uint8_t *r,
                              const uint8_t *a, const uint8_t *b, int b_shared, const uint8_t *G, const uint8_t *F,
                              const uint8_t *B, uint8_t *proofs_out, uint8_t *status_out, uint8_t *transcripts_out);

/* ---- the dealer's share audit of the multi-party range-proof protocol ---------------------------
 * nshares independent calls of
 *   ProofShare::audit_share(&self, bp_gens, pc_gens, j, &bit_commitment, &bit_challenge, &poly_commitment, &poly_challenge)
 * (src/range_proof/messages.rs:85-167): what Dealer::receive_shares runs for every party when the aggregated proof does
 * not verify (src/range_proof/dealer.rs:303-335), to name the parties whose shares are malformed.  Per share: the check
 * t_x == <l_vec, r_vec> and two multiscalar multiplications (2n + 3 and 5 terms, messages.rs:128-141, 149-160) whose
 * results must be the identity; generators are the context's (bpgpu_gens_create / _load).
 *   n           : bitsize of the shares (l_vec.len(): 8, 16, 32 or 64); n > gens_capacity fails every share (check_size)
 *   party_index : nshares x uint32, the party position j of each share (j >= party_capacity fails that share)
 *   shares      : nshares x 32*(3 + 2n) bytes: t_x, t_x_blinding, e_blinding, l_vec[n], r_vec[n] (canonical scalars;
 *                 a non-canonical one fails the share -- upstream they are Scalars by type)
 *   bit_commitments  : nshares x 96 bytes: V_j, A_j, S_j (compressed; A_j, S_j are RistrettoPoints upstream: an
 *                 undecodable encoding fails the share here)
 *   poly_commitments : nshares x 64 bytes: T_1_j, T_2_j
 *   challenges  : nshares x 96 bytes (y, z, x per share), or 96 bytes when challenges_shared != 0
 *   verdict     : nshares bytes: BPGPU_VERDICT_OK = Ok(()), BPGPU_VERDICT_VERIFICATION_ERROR = Err(())
 *   checks_out  : optional nshares x 64 bytes: compress(P_check), compress(t_check) (parity tests) */
int bpgpu_rangeproof_audit_shares(bpgpu_ctx *ctx, size_t n, size_t nshares, const uint32_t *party_index,
                                  const uint8_t *shares, const uint8_t *bit_commitments, const uint8_t *poly_commitments,
                                  const uint8_t *challenges, int challenges_shared, uint8_t *verdict, uint8_t *checks_out);

/* ---- batched inner-product-proof creation (prover side) --------------------------------
 * nbatch independent calls of
 *   InnerProductProof::create(&mut transcript, &Q, G_factors, H_factors, G_vec, H_vec, a_vec, b_vec).to_bytes()
 * (src/inner_product_proof.rs:38-193), all of one size n (a power of two): per round the reference forms L and R with
 * two (2n'+1)-term multiscalar multiplications (ipp.rs:87-113) and folds the generators with 2n' two-term ones
 * (ipp.rs:127-178); here all proofs advance round by round together and every L_j / R_j is one multiscalar
 * multiplication over the ORIGINAL points (csrc/ipp_prover.h), which yields the same group elements: proofs are
 * byte-identical to the reference algorithm's.
 *   transcript  : shared_transcript (host, 208 bytes, may hold earlier messages) if not NULL, else Transcript::new(label);
 *                 innerproduct_domain_sep(n) is applied here, as create() does
 *   Q           : nbatch x 32;  G_factors, H_factors, a, b : nbatch x n x 32 (canonical scalars)
 *   G, H        : nbatch x n x 32, or n x 32 when bases_shared != 0 (compressed points)
 *   proofs_out  : nbatch x 32 * (2 lg n + 2) bytes: L_1 R_1 ... L_k R_k a b   (InnerProductProof::to_bytes, ipp.rs:334-345)
 *   status      : nbatch bytes, BPGPU_MSM_* (a point that does not decode / a non-canonical scalar: that proof is void)
 * VARIABLE TIME in the secret vectors a, b -- as the reference's create(), which calls vartime_multiscalar_mul -- and
 * therefore no replacement for the constant-time commitments of the range-proof parties (party.rs:119-124). */
int bpgpu_ipp_create_batch(bpgpu_ctx *ctx, size_t n, size_t nbatch, const uint8_t *label, size_t label_len,
                           const uint8_t *shared_transcript, const uint8_t *Q, const uint8_t *G_factors,
                           const uint8_t *H_factors, const uint8_t *G, const uint8_t *H, int bases_shared,
                           const uint8_t *a, const uint8_t *b, uint8_t *proofs_out, uint8_t *status);

/* ---- batched range-proof creation (prover side) ------------------------------------------
 * nbatch independent calls of
 *   RangeProof::prove_multiple_with_rng(bp_gens, pc_gens, &mut transcript, &values, &blindings, n, rng)
 * (src/range_proof/mod.rs:234-288, running the dealer / party protocol of party.rs and dealer.rs in-line), all of one
 * shape (n, m = values per proof): every commitment (V_j, A, S, T_1, T_2, Q) is a multiscalar multiplication over the
 * generator tables, the inner-product argument is bpgpu_ipp_create_batch's (csrc/rp_prover.h).  Byte-identical to the
 * reference algorithm given the same random scalars.
 *   values      : nbatch x m u64 (each < 2^n);  blindings : nbatch x m x 32 bytes (Scalars, reduced mod l)
 *   transcript  : shared_transcript (host, 208 bytes, may hold earlier messages) if not NULL, else Transcript::new(label)
 *   rng         : nbatch x 64 * (m * (2n + 2) + 2m) bytes = what the rng would hand Scalar::random, in the order the
 *                 reference draws: per party j: a_blinding, s_blinding, s_L[0..n), s_R[0..n); then per party j:
 *                 t_1_blinding, t_2_blinding.  NULL = OS CSPRNG (the thread_rng() of prove_multiple).
 *   proofs_out  : nbatch x 32 * (9 + 2 lg(n m)) bytes (RangeProof::to_bytes);  commitments_out : nbatch x m x 32 bytes
 *   transcripts_out : optional nbatch x 208 bytes: each proof's transcript as the prover leaves it
 * Returns BPGPU_ERR_INVALID_ARG for the reference's InvalidBitsize / InvalidAggregation (m not a power of two) / a value
 * that does not fit n bits, BPGPU_ERR_NO_GENS for InvalidGeneratorsLength.
 * Timing: by default VARIABLE TIME in the secrets (values, blindings, s_L, s_R) -- the window-table walk is indexed by their
 * digits; for provers whose GPU an adversary cannot observe.  With the context option "prover_constant_time" = 1 the
 * secret-dependent commitments V_j, A, S, T_1, T_2 -- the ones the reference computes with its constant-time
 * multiscalar_mul (party.rs:99-124, 179-187; generators.rs:39-41) -- take a small-window table walk whose addresses and
 * instruction stream do not depend on the scalars (every (generator, window) pair reads all 8 table entries, selects by masks,
 * always adds; csrc/msm_fixed.h).  Proofs are byte-identical either way.  The inner-product rounds stay variable-time, as in the
 * reference (vartime_multiscalar_mul at ipp.rs:87-178).  Cost: ~2x the prover's time at (64, 1).
 * Secrets at rest: like the reference's parties (zeroize on Drop, party.rs:148-260), every prover entry point
 * (bpgpu_rangeproof_prove_batch, bpgpu_ipp_create_batch, bpgpu_linear_create_batch) clears what it staged before it returns,
 * on success and on every error path: the context's device IO buffer, the provers' working sets and the MSM arena (window
 * digits of secret scalars) on the stream behind the last copy, then the secret part of the pinned host staging block.
 * Not cleared: the caller's own buffers. */
int bpgpu_rangeproof_prove_batch(bpgpu_ctx *ctx, size_t n, size_t m, size_t nbatch, const uint64_t *values,
                                 const uint8_t *blindings, const uint8_t *label, size_t label_len,
                                 const uint8_t *shared_transcript, const uint8_t *rng,
                                 uint8_t *proofs_out, uint8_t *commitments_out, uint8_t *transcripts_out);

/* InnerProductProof::from_bytes + InnerProductProof::verification_scalars (src/inner_product_proof.rs:198-253, 373-407) for
 * nbatch proofs: what the R1CS verifier calls (src/r1cs/verifier.rs:401-404) before it builds ITS multiscalar multiplication
 * (r1cs/verifier.rs:459-491, served by bpgpu_msm_batch_shared), and what InnerProductProof::verify uses at ipp.rs:283.
 *   proofs      : nbatch x proof_len bytes, InnerProductProof::to_bytes (L_0 R_0 .. L_{k-1} R_{k-1} a b)
 *   transcript  : transcripts == NULL: Transcript::new(label);  transcript_stride == 0: ONE 208-byte state shared by the
 *                 batch;  == BPGPU_TRANSCRIPT_BYTES: one state per proof.  innerproduct_domain_sep(n) is applied by the call.
 *   u_sq, u_inv_sq : nbatch x k x 32 bytes (k = lg n): u_i^2 and u_i^-2 in creation order;  s : nbatch x n x 32 bytes
 *   transcripts_out (optional) : the advanced states of the proofs with status 0 (others: unspecified)
 *   status      : nbatch bytes: 0, BPGPU_VERDICT_VERIFICATION_ERROR (n != 2^k, or an identity L_i / R_i:
 *                 validate_and_append_point), BPGPU_VERDICT_FORMAT_ERROR (malformed length, a / b not canonical); the outputs
 *                 of a rejected proof are zero */
int bpgpu_ipp_verification_scalars(bpgpu_ctx *ctx, size_t n, size_t nbatch, const uint8_t *proofs, size_t proof_len,
                                   const uint8_t *label, size_t label_len, const uint8_t *transcripts, size_t transcript_stride,
                                   uint8_t *u_sq, uint8_t *u_inv_sq, uint8_t *s, uint8_t *transcripts_out, uint8_t *status);

/* ---- pool: the scheduler (any number of proofs per call, any number of devices) ------------
 * The reference's call shape is ONE call for as many proofs as the caller has: a loop over
 * RangeProof::verify_multiple (src/range_proof/mod.rs:457-470), from one thread or many.  A bpgpu_ctx is one launch
 * chain on one stream and cannot fill a device by itself; a pool owns `lanes_per_device` contexts on each of `ndev`
 * devices (the generator tables are built once per device and shared by its lanes) and does the scheduling that the
 * benchmark used to do by hand:
 *   - bpgpu_pool_rangeproof_verify / bpgpu_pool_rangeproof_verify_ts (HOST pointers, blocking, any nbatch, ANY NUMBER OF THREADS AT
 *     ONCE): the requests go through the pool's combining queue (declared below): calls that arrive close together share launch
 *     chains, a large call spans several chains and -- proofs being independent units -- takes the contiguous shard
 *     [nbatch d / ndev, nbatch (d+1) / ndev) of every device; every piece's verdicts land at its offset of the caller's buffer: the
 *     "final gather" of SURVEY 8e is that host-side placement, no collective is involved.  Verdicts are exactly those of
 *     bpgpu_rangeproof_verify_batch[_ts] on the whole batch.  (Option "host_path_combining" = 0 restores round 3's path for the label
 *     form: one call at a time, slices staged by "host_workers" threads per device.)
 *   - bpgpu_pool_rangeproof_submit_dev (DEVICE pointers on device `dev_index` of the pool, asynchronous): the batch is
 *     queued; bpgpu_pool_flush -- or the pool itself once "auto_flush_items" batches wait (default: one per lane) --
 *     packs consecutive queued batches of one shape (n, m, proof_len, label) into coalesced launch chains of about
 *     "coalesce_proofs" proofs (default 5120) and issues them on the lanes round-robin.  A burst of small batches is
 *     thereby served as a few wide chains instead of many narrow ones (20 x 1024 proofs from an idle device: 4.1 M
 *     verifications/s as twenty chains, 5.9 - 6.2 M as two); every batch still gets its own verdict (and msm_out) buffer filled.
 *     Input buffers must be complete on the device when the batch is submitted and stay valid until bpgpu_pool_wait returns (the
 *     pool's streams are not ordered against the caller's) -- or use bpgpu_pool_rangeproof_submit_dev_ex below: a producer stream
 *     the chain waits for, and a ticket per batch.  Batches whose length or parameters are malformed are passed to
 *     bpgpu_rangeproof_verify_batch_dev unchanged, which reports them per proof.  When a chain cannot be issued, the verdict bytes
 *     of the batches it carried are set to BPGPU_VERDICT_UNDECIDED (never left at 0 = "verified") and the flush reports the error.
 * devices: HIP ordinals, one entry per shard (an ordinal may repeat: two shards on one GPU -- what the one-GPU tests
 * do).  lanes_per_device: 0 = 32 (plus the combining queue's own lanes: environment BPGPU_COMBINE_LANES, default 12).  More than 4
 * lanes need 8..16 hardware queues: export GPU_MAX_HW_QUEUES=16 before the process's FIRST HIP call (the ROCm runtime reads it
 * then).  libbpgpu sets the variable when it is loaded if it is unset -- which only helps when nothing initialised HIP earlier;
 * bpgpu_pool_create therefore returns BPGPU_ERR_HW_QUEUES when it finds another value, and, when the value is the library's own,
 * after timing sixteen single-wavefront kernels that spin 1 ms each on sixteen streams (~3 ms in all, best of three) and finding fewer than 6 of them overlapping.
 * bpgpu_pool_last_error is per calling thread: what the last pool call OF THAT THREAD reported.
 * Options (bpgpu_pool_set_option; one line each, every one of them is flipped by a test -- tests/test_abi_and_host.py::
 * test_every_pool_option_is_documented_and_settable, tests/cpu_pool):
 *   "coalesce_proofs"      5120   target width of a coalesced launch chain
 *   "max_chain_proofs"     16384  no chain wider than this (a lane's arena: ~55 KB per proof)
 *   "pair_limit_proofs"    24576  a flush of up to this many proofs is issued as at most two chains
 *   "latency_proofs"       6144   a host call, or a flush on an idle device, of up to this many proofs is alone: its chains take the latency forms
 *   "auto_flush_items"     0      flush by itself once this many items wait on a device (0 = the number of lanes)
 *   "auto_flush_proofs"    0      ... or once this many proofs wait (0 = off)
 *   "slice_proofs"         0      host-pointer calls of the round-3 host path: proofs per slice (0 = automatic)
 *   "host_workers"         0      ... and its worker threads (set before the first host-pointer call)
 *   "host_path_combining"  1      bpgpu_pool_rangeproof_verify through the combining queue (0 = the round-3 slicing workers)
 *   "rlc_isolate"          0      1 = a chain carries at most one batch-combined batch: a bad proof leaves only its own batch undecided
 *   "combine_wait_us"      100    a staging buffer leaves at the latest this long after its first request arrived ...
 *   "combine_quiet_us"     20     ... or when nothing has joined it for this long
 *   "combine_max_age_us"   1500   ... and no request waits longer than this for its chain to be issued
 *   "combine_inflight"     4      deadlines seal buffers only while fewer chains than this run: beyond, load widens the chains
 *   "combine_busy_chains"  2      a chain issued beside this many others takes the throughput forms
 *   "combine_max_open"     4      transcript-position classes with a staging buffer of their own
 *   "combine_mapped_out"   1024   chains up to this wide write their results straight into pinned host memory
 *   "combine_msm_bytes"    32 MiB staging block of a multiscalar-multiplication class: MSMs per chain = this / bytes per MSM
 *   "combine_trace"        0      ring size of the timeline records (bpgpu_pool_trace_dump)
 *   "stat_reset"           -      zero all statistics
 * Any other key is forwarded to every lane context (set those before bpgpu_pool_gens_*).  When a staging buffer leaves is decided per
 * kind of work, as measured (profiles/r05/combine_policy_ab.txt): range proofs by the deadlines above in two regimes, multiscalar multiplications and inner-product
 * proofs in cohorts (a buffer leaves when the group the last chain released is back); the constants of both policies are not options.
 * Removed in round 6 because their own A/B refuted them (the tables stay under profiles/r05/): "plan_by_work", "plan_min_chain_proofs",
 * "stagger_chains", "combine_policy", "combine_mapped_in"; context keys "split_stage1", "fork_early".
 * Read-only statistics (bpgpu_pool_get_option): "stat_chains", "stat_chain_proofs" (launch chains issued by flushes and the proofs they
 * carried), "stat_last_splits", "stat_combined_chains" / "_proofs" / "_requests", "stat_svc_issue_us" / "_complete_us" / "_polls" /
 * "_deliver_us" (the combining queue's service threads), "stat_active_calls", "combine_lanes". */
typedef struct bpgpu_pool bpgpu_pool;
int bpgpu_pool_create(const int *devices, int ndev, int lanes_per_device, bpgpu_pool **out);
void bpgpu_pool_destroy(bpgpu_pool *pool);
const char *bpgpu_pool_last_error(bpgpu_pool *pool);
int bpgpu_pool_set_option(bpgpu_pool *pool, const char *key, int64_t value);
int bpgpu_pool_get_option(bpgpu_pool *pool, const char *key, int64_t *value);
int bpgpu_pool_devices(bpgpu_pool *pool);
int bpgpu_pool_lanes(bpgpu_pool *pool);
/* lane `lane` of shard `dev_index` (for bpgpu_profile_*, bpgpu_ctx_get_option); owned by the pool */
bpgpu_ctx *bpgpu_pool_lane(bpgpu_pool *pool, int dev_index, int lane);
/* BulletproofGens::new / PedersenGens::default (or the caller's encodings) on every device of the pool */
int bpgpu_pool_gens_create(bpgpu_pool *pool, size_t gens_capacity, size_t party_capacity);
int bpgpu_pool_gens_load(bpgpu_pool *pool, size_t gens_capacity, size_t party_capacity, const uint8_t *G, const uint8_t *H,
                         const uint8_t B[32], const uint8_t B_blinding[32]);
/* bpgpu_gens_add_shape on every lane of every device (call it while the pool is idle) */
int bpgpu_pool_gens_add_shape(bpgpu_pool *pool, size_t n2, size_t m2);
/* arguments as bpgpu_rangeproof_verify_batch */
int bpgpu_pool_rangeproof_verify(bpgpu_pool *pool, size_t n, size_t m, size_t nbatch, const uint8_t *proofs, size_t proof_len,
                                 const uint8_t *commitments, const uint8_t *label, size_t label_len, const uint8_t *rng64,
                                 uint8_t *verdict, uint8_t *msm_out);
/* arguments as bpgpu_rangeproof_verify_batch_dev, without the stream */
int bpgpu_pool_rangeproof_submit_dev(bpgpu_pool *pool, int dev_index, size_t n, size_t m, size_t nbatch, const void *d_proofs,
                                     size_t proof_len, const void *d_commitments, const uint8_t *label, size_t label_len,
                                     const void *d_rng64, void *d_verdict, void *d_msm_out);
/* RangeProof::verify_multiple_with_rng in the reference's own call shape (src/range_proof/mod.rs:345-353, 455-470): a FEW proofs
 * per call -- one is fine --, each with its own `transcript: &mut Transcript`, BLOCKING, from ANY number of threads at once.
 * Calls that arrive close together share a launch chain (the pool's combining queue: one staging buffer per class of requests
 * -- shape and STROBE position of the transcripts --, which leaves as one chain when it is full ("coalesce_proofs"), when its
 * first proof has waited "combine_wait_us" (default 100) or when nothing joined it for "combine_quiet_us" (default 20); every
 * caller is woken when ITS proofs are done, not when the pool drains).  Arguments as bpgpu_rangeproof_verify_batch_ts:
 *   transcripts / transcript_stride : BPGPU_TRANSCRIPT_BYTES: one 208-byte state per proof (may have absorbed application data:
 *                 the normal use of Merlin); 0: ONE state shared by the nbatch proofs (e.g. bpgpu_transcript_new(label))
 *   transcripts_out (optional, nbatch x 208; may alias `transcripts` when the stride is 208): the advanced states, as
 *                 bpgpu_rangeproof_verify_batch_ts leaves them (a FormatError proof's state is handed back untouched)
 *   rng64 (optional): 64 bytes per proof for the batching challenge; NULL: drawn per calling thread from a ChaCha20 generator
 *                 keyed by the OS (the thread_rng() of verify_multiple)
 * Verdicts, encodings and states are bit-identical to bpgpu_rangeproof_verify_batch_ts on the same inputs.  Requests of any
 * size are accepted (a large one spans several chains and, on a multi-device pool, takes a contiguous shard per device).
 * bpgpu_pool_rangeproof_verify (label instead of transcripts) is the same call with Transcript::new(label) for every proof.
 * On a non-zero return no verdict byte of the call reads 0: proofs whose chain failed carry BPGPU_VERDICT_UNDECIDED.
 *
 * bpgpu_pool_rangeproof_submit_ts: the non-blocking form.  Returns once the inputs have been copied (they may be reused at
 * once); verdict / msm_out / transcripts_out must stay valid until bpgpu_pool_ticket_wait(ticket) has returned -- it blocks
 * until THIS request is complete (not the pool), returns the request's error code and frees the ticket; every ticket must be
 * waited for exactly once.  bpgpu_pool_ticket_done: 1 when wait would not block.  One thread can thereby keep thousands of
 * single-proof requests in flight. */
typedef struct bpgpu_ticket bpgpu_ticket;
int bpgpu_pool_rangeproof_verify_ts(bpgpu_pool *pool, size_t n, size_t m, size_t nbatch, const uint8_t *proofs, size_t proof_len,
                                    const uint8_t *commitments, const uint8_t *transcripts, size_t transcript_stride,
                                    const uint8_t *rng64, uint8_t *verdict, uint8_t *msm_out, uint8_t *transcripts_out);
int bpgpu_pool_rangeproof_submit_ts(bpgpu_pool *pool, size_t n, size_t m, size_t nbatch, const uint8_t *proofs, size_t proof_len,
                                    const uint8_t *commitments, const uint8_t *transcripts, size_t transcript_stride,
                                    const uint8_t *rng64, uint8_t *verdict, uint8_t *msm_out, uint8_t *transcripts_out,
                                    bpgpu_ticket **ticket);
int bpgpu_pool_ticket_done(bpgpu_pool *pool, bpgpu_ticket *ticket);
int bpgpu_pool_ticket_wait(bpgpu_pool *pool, bpgpu_ticket *ticket);
/* ---- the boundary function itself through the pool: one multiscalar multiplication (or a few) per call, from any thread ------------
 * SURVEY 8b's drop-in point is RistrettoPoint::optional_multiscalar_mul / vartime_multiscalar_mul as the crate calls it: ONE
 * multiscalar multiplication per call (src/range_proof/mod.rs:421-445, src/r1cs/verifier.rs:459-491, src/inner_product_proof.rs:308-319,
 * src/linear_proof.rs:217-225, src/range_proof/messages.rs:128-149), from whatever thread verifies.  One such call cannot fill a device
 * (bpgpu_msm_batch_shared on a context of its own: 0.77 ms per 6 179-term MSM, and threads do not add up beyond one context each), so
 * these go through the same combining queue as the range proofs: calls that arrive close together share a launch chain, every caller
 * is woken when ITS results are there.  BLOCKING, ANY NUMBER OF THREADS AT ONCE; results bit-identical to the bpgpu_msm_batch* /
 * bpgpu_ipp_verify_batch call on a context.
 *   bpgpu_pool_msm_batch_shared : arguments as bpgpu_msm_batch_shared (the mega-check shape: 2nm+2 generator scalars in the order of
 *                                 mod.rs:421-443 + n_unique (scalar, point) pairs per MSM); MSMs of one (n, m, n_unique) share chains
 *   bpgpu_pool_msm_batch_shared_submit : the non-blocking form; out / status must stay valid until bpgpu_pool_ticket_wait(ticket)
 *   bpgpu_pool_msm_batch        : arguments as bpgpu_msm_batch (ragged batch); MSMs of equal length share chains, so a call is placed
 *                                 stretch by stretch of equal n_terms; an MSM of 0 terms is the identity (encoding 0, status 0)
 *   bpgpu_pool_ipp_verify       : arguments as bpgpu_ipp_verify_batch (InnerProductProof::verify, ipp.rs:260-326); proofs of one
 *                                 (n, proof_len, label) share chains
 *   bpgpu_pool_msm_batch_shared_submit_dev : the same call for inputs that are already in HBM (device pointers, arguments as
 *                                 bpgpu_msm_batch_shared_dev): the batch leaves at once as ONE chain on the next lane of pool device
 *                                 `dev_index`, behind `producer_stream` if have_producer; ticket (optional) as for
 *                                 bpgpu_pool_rangeproof_submit_dev_ex; nothing is combined -- device-resident batches are as wide as
 *                                 their owner made them.  The pool's lanes are the (context, stream) pairs a caller would otherwise
 *                                 manage by hand.
 * On a non-zero return no status / verdict byte of the call reads 0: items whose chain failed carry BPGPU_VERDICT_UNDECIDED.
 * Option "combine_msm_bytes" (default 32 MiB): staging block per chain of a multiscalar-multiplication class -- items per chain =
 * that / input bytes per MSM (cfg5's shape, 263 KB per MSM: 127 per chain). */
int bpgpu_pool_msm_batch_shared(bpgpu_pool *pool, size_t n, size_t m, size_t nbatch, size_t n_unique, const uint8_t *gen_scalars,
                                const uint8_t *uniq_scalars, const uint8_t *uniq_points, uint8_t *out, uint8_t *status);
int bpgpu_pool_msm_batch_shared_submit(bpgpu_pool *pool, size_t n, size_t m, size_t nbatch, size_t n_unique, const uint8_t *gen_scalars,
                                       const uint8_t *uniq_scalars, const uint8_t *uniq_points, uint8_t *out, uint8_t *status,
                                       bpgpu_ticket **ticket);
int bpgpu_pool_msm_batch(bpgpu_pool *pool, size_t nbatch, const uint32_t *n_terms, const uint8_t *scalars, const uint8_t *points,
                         uint8_t *out, uint8_t *status);
int bpgpu_pool_msm_batch_shared_submit_dev(bpgpu_pool *pool, int dev_index, size_t n, size_t m, size_t nbatch, size_t n_unique,
                                           const void *d_gen_scalars, const void *d_uniq_scalars, const void *d_uniq_points, void *d_out,
                                           void *d_status, void *producer_stream, int have_producer, bpgpu_ticket **ticket);
int bpgpu_pool_ipp_verify(bpgpu_pool *pool, size_t n, size_t nbatch, const uint8_t *proofs, size_t proof_len, const uint8_t *label,
                          size_t label_len, const uint8_t *G_factors, const uint8_t *H_factors, const uint8_t *P, const uint8_t *Q,
                          const uint8_t *G, const uint8_t *H, uint8_t *verdict, uint8_t *msm_out);
/* The combining queue's timeline (set option "combine_trace" = ring size first): one JSON object per line -- every launch chain (opened,
 * sealed, issue begin / end, completion seen, delivery begin / end, buffer free; CLOCK_MONOTONIC ns) and every eighth request per thread
 * (submitted, slots reserved, inputs written, delivered, woken).  tools/combine_timeline.py turns it into "where does a request wait". */
int bpgpu_pool_trace_dump(bpgpu_pool *pool, const char *path);
/* bpgpu_pool_rangeproof_submit_dev with a completion contract (a service that consumes batch k while batch k+1 is queued, with no
 * device-wide synchronisation anywhere):
 *   producer_stream / have_producer : have_producer != 0: the batch's input buffers are complete when the work queued so far on
 *                 `producer_stream` (a hipStream_t; NULL = the legacy default stream) is -- the chain that carries the batch waits
 *                 for exactly that point on the device (an event recorded there now); 0: complete at submission, as before
 *   ticket (optional): handle of THIS batch.  bpgpu_pool_ticket_stream_wait(pool, ticket, consumer_stream) makes a consumer stream
 *                 wait, on the device, until the batch's verdicts (and encodings) are written -- it issues the pending chains of
 *                 that device first if they have not left; bpgpu_pool_ticket_wait blocks the host until then, returns the
 *                 batch's error code and frees the ticket (required exactly once per ticket); bpgpu_pool_ticket_done polls.
 * On an error of the chain that carried (part of) the batch its verdict bytes read BPGPU_VERDICT_UNDECIDED, never 0. */
int bpgpu_pool_rangeproof_submit_dev_ex(bpgpu_pool *pool, int dev_index, size_t n, size_t m, size_t nbatch, const void *d_proofs,
                                        size_t proof_len, const void *d_commitments, const uint8_t *label, size_t label_len,
                                        const void *d_rng64, void *d_verdict, void *d_msm_out, void *producer_stream,
                                        int have_producer, bpgpu_ticket **ticket);
int bpgpu_pool_ticket_stream_wait(bpgpu_pool *pool, bpgpu_ticket *ticket, void *consumer_stream);
/* bpgpu_rangeproof_verify_rlc_dev through the pool (SURVEY 8f-3's ADDITIONAL batch-combined check, no counterpart in the crate): consecutive
 * submitted batches of one shape are combined by ONE identity check per launch chain (~coalesce_proofs proofs: from 32 768 per-proof terms
 * the bucket MSM) instead of one per batch -- a service with batches of 1 024 gets the rate of batches of 5 120.  Weights: drawn by the
 * library per chain (OS CSPRNG).  d_verdict: per proof 0 / the front end's status / BPGPU_VERDICT_UNDECIDED when the CHAIN's combination is
 * not the identity -- some proof of some batch of that chain fails; which one is for the caller to find by resubmitting the undecided
 * batches through bpgpu_pool_rangeproof_submit_dev[_ex].  d_batch_out (optional, 33 bytes, 4-byte aligned): the verdict byte and the
 * combined point of the chain that carried the batch.  producer_stream / have_producer / ticket as bpgpu_pool_rangeproof_submit_dev_ex. */
int bpgpu_pool_rangeproof_submit_rlc_dev(bpgpu_pool *pool, int dev_index, size_t n, size_t m, size_t nbatch, const void *d_proofs,
                                         size_t proof_len, const void *d_commitments, const uint8_t *label, size_t label_len,
                                         const void *d_rng64, void *d_verdict, void *d_batch_out, void *producer_stream,
                                         int have_producer, bpgpu_ticket **ticket);
/* The final gather for callers that keep verdicts on the devices (north star: "... only for the final identity-check gather"): the
 * verdict bytes of every shard -- part[d]: device memory on pool device d, bytes[d] bytes -- to ONE buffer d_dst on pool device `root`,
 * shard after shard, by peer copies (xGMI between the GPUs of a node), ordered on the device behind everything the pool has issued on the
 * source device so far (flush first).  Asynchronous on `stream` (hipStream_t of the root device, NULL = default stream).  Nothing else of
 * a verification ever crosses devices: proofs are independent units (src/range_proof/mod.rs:455-470 verifies them one by one). */
int bpgpu_pool_gather_dev(bpgpu_pool *pool, int root, const void *const *part, const size_t *bytes, void *d_dst, void *stream);
int bpgpu_pool_flush(bpgpu_pool *pool);   /* issue everything queued; returns without waiting */
int bpgpu_pool_wait(bpgpu_pool *pool);    /* flush, then wait until every lane is idle */

/* ---- instrumentation -----------------------------------------------------------
 * When enabled, every kernel launch is bracketed by HIP events on its stream;
 * bpgpu_profile_report writes one line per kernel: "name launches total_ms". */
int bpgpu_profile_enable(bpgpu_ctx *ctx, int on);
int bpgpu_profile_reset(bpgpu_ctx *ctx);
int bpgpu_profile_report(bpgpu_ctx *ctx, char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
