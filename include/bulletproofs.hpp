// bulletproofs.hpp -- host-side mirror (C++17, header-only) of the reference crate's public
// verification surface, implemented over the C ABI of libbpgpu.so (include/bpgpu.h).
//
// Mirrors, with the same names, argument meaning and error behaviour:
//   ProofError ............ src/errors.rs:12-54
//   PedersenGens .......... src/generators.rs:30-53        (default() = basepoint + hashed blinding base)
//   BulletproofGens ....... src/generators.rs:157-204      (new(gens_capacity, party_capacity))
//   Transcript ............ merlin::Transcript: new / append_message / append_u64 / challenge_bytes, held as its
//                           208-byte STROBE state; verifiers take it by reference and leave it advanced
//   RangeProof ............ src/range_proof/mod.rs:59-76, from_bytes 504-538, to_bytes 487-500,
//                           verify_single[_with_rng] 316-342, verify_multiple[_with_rng] 345-470,
//                           prove_single/multiple_with_rng 115-288 (variable time on the GPU)
//   LinearProof ........... src/linear_proof.rs: create 40-173 (variable time on the GPU), from_bytes 350-394,
//                           to_bytes 322-331, verify 175-236
// plus verify_batch, the batched entry point this engine exists for.  No arithmetic happens on the
// host: parsing checks lengths and scalar canonicity (so from_bytes fails where the reference's does)
// and everything else is one call into the GPU library.  There is no CPU fallback.
#ifndef BULLETPROOFS_HPP
#define BULLETPROOFS_HPP
#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <variant>
#include <vector>

#include "bpgpu.h"

namespace bulletproofs {

enum class ProofError {
    VerificationError = BPGPU_VERDICT_VERIFICATION_ERROR,
    FormatError = BPGPU_VERDICT_FORMAT_ERROR,
    InvalidBitsize = BPGPU_VERDICT_INVALID_BITSIZE,
    InvalidGeneratorsLength = BPGPU_VERDICT_INVALID_GENERATORS_LENGTH,
};

// Result<(), ProofError>
struct Status {
    bool ok_;
    ProofError err_;
    static Status Ok() { return {true, ProofError::VerificationError}; }
    static Status Err(ProofError e) { return {false, e}; }
    bool is_ok() const { return ok_; }
    ProofError unwrap_err() const {
        if (ok_) throw std::logic_error("unwrap_err on Ok");
        return err_;
    }
    bool operator==(const Status &o) const { return ok_ == o.ok_ && (ok_ || err_ == o.err_); }
};

using CompressedRistretto = std::array<uint8_t, 32>;
using ScalarBytes = std::array<uint8_t, 32>;

class GpuError : public std::runtime_error {
  public:
    using std::runtime_error::runtime_error;
};

class PedersenGens {
  public:
    CompressedRistretto B{}, B_blinding{};
    bool operator==(const PedersenGens &o) const { return B == o.B && B_blinding == o.B_blinding; }
};

class BulletproofGens {
  public:
    size_t gens_capacity, party_capacity;
    // BulletproofGens::new(gens_capacity, party_capacity): generators are derived on `device`
    BulletproofGens(size_t gens_capacity_, size_t party_capacity_, int device = 0)
        : gens_capacity(gens_capacity_), party_capacity(party_capacity_) {
        bpgpu_ctx *c = nullptr;
        if (bpgpu_ctx_create(device, &c) != BPGPU_OK) throw GpuError("bpgpu_ctx_create failed: no usable GPU (there is no CPU fallback)");
        ctx_.reset(c, bpgpu_ctx_destroy);
        if (bpgpu_gens_create(c, gens_capacity, party_capacity) != BPGPU_OK) throw GpuError(bpgpu_last_error(c));
    }
    bpgpu_ctx *ctx() const { return ctx_.get(); }
    // PedersenGens::default() as held by this verifier
    PedersenGens pedersen() const {
        PedersenGens pc;
        if (bpgpu_gens_export(ctx_.get(), nullptr, nullptr, pc.B.data(), pc.B_blinding.data()) != BPGPU_OK) throw GpuError(bpgpu_last_error(ctx_.get()));
        return pc;
    }
    // PedersenGens::commit(value, blinding) (generators.rs:38-42) with this context's bases, compressed (variable time on the GPU)
    CompressedRistretto commit(const ScalarBytes &value, const ScalarBytes &blinding) const {
        const PedersenGens pc = pedersen();
        uint8_t sc[64], pt[64], status = 0;
        std::memcpy(sc, value.data(), 32);
        std::memcpy(sc + 32, blinding.data(), 32);
        std::memcpy(pt, pc.B.data(), 32);
        std::memcpy(pt + 32, pc.B_blinding.data(), 32);
        const uint32_t nt = 2;
        CompressedRistretto out{};
        if (bpgpu_msm_batch(ctx_.get(), 1, &nt, sc, pt, out.data(), &status) != BPGPU_OK) throw GpuError(bpgpu_last_error(ctx_.get()));
        if (status != 0) throw std::invalid_argument("commit: scalars must be canonical");
        return out;
    }
    // BulletproofGens::increase_capacity (generators.rs:177-204): no-op unless larger; the device tables are rebuilt
    void increase_capacity(size_t new_capacity) {
        if (gens_capacity >= new_capacity) return;
        if (bpgpu_gens_create(ctx_.get(), new_capacity, party_capacity) != BPGPU_OK) throw GpuError(bpgpu_last_error(ctx_.get()));
        gens_capacity = new_capacity;
    }
    // the aggregated iterators G(n, m) / H(n, m) (generators.rs:207-259): the first n generators of each of the first m parties
    std::vector<CompressedRistretto> G(size_t n, size_t m) const { return slice(true, n, m, 0); }
    std::vector<CompressedRistretto> H(size_t n, size_t m) const { return slice(false, n, m, 0); }
    // share(j).G(n) / share(j).H(n) (generators.rs:168-175, 262-292)
    std::vector<CompressedRistretto> share_G(size_t j, size_t n) const { return slice(true, n, 1, j); }
    std::vector<CompressedRistretto> share_H(size_t j, size_t n) const { return slice(false, n, 1, j); }

    // The verifier multiplies by pc_gens.B / B_blinding (mod.rs:439-440); the device tables hold the bases of this
    // BulletproofGens.  Any other PedersenGens would silently verify a different statement: refuse it.
    void check_pedersen(const PedersenGens &pc) const {
        if (!(pc == pedersen())) throw std::invalid_argument("pc_gens differs from the Pedersen bases in the device tables (load custom bases with bpgpu_gens_load)");
    }

  private:
    std::vector<CompressedRistretto> slice(bool g, size_t n, size_t m, size_t first_party) const {
        if (n > gens_capacity || first_party + m > party_capacity) throw std::out_of_range("generators: n or party index beyond capacity");
        const size_t tot = gens_capacity * party_capacity;
        std::vector<uint8_t> flat(tot * 32);
        if (bpgpu_gens_export(ctx_.get(), g ? flat.data() : nullptr, g ? nullptr : flat.data(), nullptr, nullptr) != BPGPU_OK)
            throw GpuError(bpgpu_last_error(ctx_.get()));
        std::vector<CompressedRistretto> out(n * m);
        for (size_t j = 0; j < m; j++)
            for (size_t i = 0; i < n; i++) std::memcpy(out[j * n + i].data(), &flat[((first_party + j) * gens_capacity + i) * 32], 32);
        return out;
    }
    std::shared_ptr<bpgpu_ctx> ctx_;
};

// merlin::Transcript, held as its 208-byte STROBE-128 state (bpgpu.h BPGPU_TRANSCRIPT_BYTES).  It may absorb
// application messages before it is handed to a verifier, and a verifier leaves it advanced, as
// verify_multiple_with_rng(&mut transcript, ...) does (mod.rs:345-353).
class Transcript {
  public:
    explicit Transcript(const std::string &label) : Transcript(reinterpret_cast<const uint8_t *>(label.data()), label.size()) {}
    Transcript(const uint8_t *label, size_t n) : fresh_label_(label, label + n), fresh_(true) {
        if (bpgpu_transcript_new(label, n, state_.data()) != BPGPU_OK) throw std::invalid_argument("bpgpu_transcript_new");
    }
    void append_message(const std::string &label, const uint8_t *msg, size_t n) {
        if (bpgpu_transcript_append_message(state_.data(), reinterpret_cast<const uint8_t *>(label.data()), label.size(), msg, n) != BPGPU_OK)
            throw std::invalid_argument("bpgpu_transcript_append_message");
        fresh_ = false;
    }
    void append_u64(const std::string &label, uint64_t x) {
        uint8_t b[8];
        for (int i = 0; i < 8; i++) b[i] = static_cast<uint8_t>(x >> (8 * i));
        append_message(label, b, 8);
    }
    void challenge_bytes(const std::string &label, uint8_t *out, size_t n) {
        if (bpgpu_transcript_challenge_bytes(state_.data(), reinterpret_cast<const uint8_t *>(label.data()), label.size(), out, n) != BPGPU_OK)
            throw std::invalid_argument("bpgpu_transcript_challenge_bytes");
        fresh_ = false;
    }
    const std::array<uint8_t, BPGPU_TRANSCRIPT_BYTES> &state() const { return state_; }
    std::array<uint8_t, BPGPU_TRANSCRIPT_BYTES> &state_mut() {
        fresh_ = false;
        return state_;
    }
    // the label while the transcript is still exactly Transcript::new(label) (what the batch-combined entry point needs)
    bool is_fresh() const { return fresh_; }
    const std::vector<uint8_t> &label() const { return fresh_label_; }

  private:
    std::array<uint8_t, BPGPU_TRANSCRIPT_BYTES> state_{};
    std::vector<uint8_t> fresh_label_;
    bool fresh_;
};

namespace detail {
inline bool scalar_is_canonical(const uint8_t *s) {   // Scalar::from_canonical_bytes
    static const uint8_t L[32] = {0xed, 0xd3, 0xf5, 0x5c, 0x1a, 0x63, 0x12, 0x58, 0xd6, 0x9c, 0xf7, 0xa2, 0xde, 0xf9, 0xde, 0x14,
                                  0,    0,    0,    0,    0,    0,    0,    0,    0,    0,    0,    0,    0,    0,    0,    0x10};
    for (int i = 31; i >= 0; i--) {
        if (s[i] < L[i]) return true;
        if (s[i] > L[i]) return false;
    }
    return false;
}
}  // namespace detail

class RangeProof {
  public:
    // Result<RangeProof, ProofError> of RangeProof::from_bytes (mod.rs:504-538 + ipp.rs:373-407)
    static std::variant<RangeProof, ProofError> from_bytes(const uint8_t *slice, size_t len) {
        if (len % 32 != 0 || len < 7 * 32) return ProofError::FormatError;
        for (int i = 4; i < 7; i++)
            if (!detail::scalar_is_canonical(slice + 32 * i)) return ProofError::FormatError;
        const size_t ne = (len - 7 * 32) / 32;
        if (ne < 2 || (ne - 2) % 2 != 0) return ProofError::FormatError;
        const size_t lg_n = (ne - 2) / 2;
        if (lg_n >= 32) return ProofError::FormatError;
        if (!detail::scalar_is_canonical(slice + len - 64) || !detail::scalar_is_canonical(slice + len - 32)) return ProofError::FormatError;
        RangeProof p;
        p.bytes_.assign(slice, slice + len);
        return p;
    }
    static std::variant<RangeProof, ProofError> from_bytes(const std::vector<uint8_t> &v) { return from_bytes(v.data(), v.size()); }

    // RangeProof::prove_multiple_with_rng (mod.rs:234-288) on the GPU: (proof, value commitments).  `transcript` is &mut
    // (left advanced).  rng_bytes: the bytes the rng would hand Scalar::random, in the reference's draw order --
    // 64 * (m (2n + 2) + 2m) of them -- or nullptr for the OS CSPRNG.  VARIABLE TIME in the secrets (see bpgpu.h).
    static std::pair<RangeProof, std::vector<CompressedRistretto>> prove_multiple_with_rng(const BulletproofGens &bp_gens, const PedersenGens &pc_gens,
                                                                                           Transcript &transcript, const std::vector<uint64_t> &values,
                                                                                           const std::vector<ScalarBytes> &blindings, size_t n,
                                                                                           const uint8_t *rng_bytes = nullptr) {
        bp_gens.check_pedersen(pc_gens);
        const size_t m = values.size();
        if (blindings.size() != m) throw std::invalid_argument("prove_multiple: WrongNumBlindingFactors");
        size_t k = 0;
        while ((size_t(1) << k) < n * m) k++;
        RangeProof p;
        p.bytes_.resize(32 * (9 + 2 * k));
        std::vector<CompressedRistretto> vc(m);
        std::vector<uint8_t> bl(32 * m);
        for (size_t j = 0; j < m; j++) std::memcpy(&bl[32 * j], blindings[j].data(), 32);
        std::array<uint8_t, BPGPU_TRANSCRIPT_BYTES> in = transcript.state();
        const int rc = bpgpu_rangeproof_prove_batch(bp_gens.ctx(), n, m, 1, values.data(), bl.data(), nullptr, 0, in.data(), rng_bytes, p.bytes_.data(),
                                                    m ? vc[0].data() : nullptr, transcript.state_mut().data());
        if (rc != BPGPU_OK) throw GpuError(bpgpu_last_error(bp_gens.ctx()));
        return {p, vc};
    }
    static std::pair<RangeProof, CompressedRistretto> prove_single_with_rng(const BulletproofGens &bp_gens, const PedersenGens &pc_gens, Transcript &transcript,
                                                                            uint64_t v, const ScalarBytes &v_blinding, size_t n,
                                                                            const uint8_t *rng_bytes = nullptr) {
        auto r = prove_multiple_with_rng(bp_gens, pc_gens, transcript, {v}, {v_blinding}, n, rng_bytes);
        return {r.first, r.second[0]};
    }
    // prove_multiple / prove_single (mod.rs:290-311, 141-158): thread_rng() -> the OS CSPRNG
    static std::pair<RangeProof, std::vector<CompressedRistretto>> prove_multiple(const BulletproofGens &bp_gens, const PedersenGens &pc_gens,
                                                                                  Transcript &transcript, const std::vector<uint64_t> &values,
                                                                                  const std::vector<ScalarBytes> &blindings, size_t n) {
        return prove_multiple_with_rng(bp_gens, pc_gens, transcript, values, blindings, n, nullptr);
    }
    static std::pair<RangeProof, CompressedRistretto> prove_single(const BulletproofGens &bp_gens, const PedersenGens &pc_gens, Transcript &transcript,
                                                                   uint64_t v, const ScalarBytes &v_blinding, size_t n) {
        return prove_single_with_rng(bp_gens, pc_gens, transcript, v, v_blinding, n, nullptr);
    }
    const std::vector<uint8_t> &to_bytes() const { return bytes_; }

    // verify_multiple_with_rng: rng64 = the 64 bytes the rng would hand Scalar::random (mod.rs:396)
    // (`transcript` is `&mut Transcript`: it may hold earlier messages and is left advanced)
    Status verify_multiple_with_rng(const BulletproofGens &bp_gens, const PedersenGens &pc_gens, Transcript &transcript,
                                    const std::vector<CompressedRistretto> &value_commitments, size_t n, const uint8_t *rng64) const {
        bp_gens.check_pedersen(pc_gens);
        uint8_t verdict = 0;
        std::array<uint8_t, BPGPU_TRANSCRIPT_BYTES> in = transcript.state();
        const int rc = bpgpu_rangeproof_verify_batch_ts(bp_gens.ctx(), n, value_commitments.size(), 1, bytes_.data(), bytes_.size(),
                                                        value_commitments.empty() ? nullptr : value_commitments[0].data(), in.data(),
                                                        BPGPU_TRANSCRIPT_BYTES, rng64, &verdict, nullptr, transcript.state_mut().data());
        if (rc != BPGPU_OK) throw GpuError(bpgpu_last_error(bp_gens.ctx()));
        return verdict == 0 ? Status::Ok() : Status::Err(static_cast<ProofError>(verdict));
    }
    Status verify_multiple_with_rng(const BulletproofGens &bp_gens, const PedersenGens &pc_gens, Transcript &&transcript,
                                    const std::vector<CompressedRistretto> &value_commitments, size_t n, const uint8_t *rng64) const {
        return verify_multiple_with_rng(bp_gens, pc_gens, transcript, value_commitments, n, rng64);
    }
    // verify_multiple: thread_rng() -> the library draws from the OS CSPRNG
    Status verify_multiple(const BulletproofGens &bp_gens, const PedersenGens &pc_gens, Transcript &transcript,
                           const std::vector<CompressedRistretto> &value_commitments, size_t n) const {
        return verify_multiple_with_rng(bp_gens, pc_gens, transcript, value_commitments, n, nullptr);
    }
    Status verify_multiple(const BulletproofGens &bp_gens, const PedersenGens &pc_gens, Transcript &&transcript,
                           const std::vector<CompressedRistretto> &value_commitments, size_t n) const {
        return verify_multiple_with_rng(bp_gens, pc_gens, transcript, value_commitments, n, nullptr);
    }
    Status verify_single_with_rng(const BulletproofGens &bp_gens, const PedersenGens &pc_gens, Transcript &transcript,
                                  const CompressedRistretto &V, size_t n, const uint8_t *rng64) const {
        return verify_multiple_with_rng(bp_gens, pc_gens, transcript, {V}, n, rng64);
    }
    Status verify_single_with_rng(const BulletproofGens &bp_gens, const PedersenGens &pc_gens, Transcript &&transcript,
                                  const CompressedRistretto &V, size_t n, const uint8_t *rng64) const {
        return verify_multiple_with_rng(bp_gens, pc_gens, transcript, {V}, n, rng64);
    }
    Status verify_single(const BulletproofGens &bp_gens, const PedersenGens &pc_gens, Transcript &transcript,
                         const CompressedRistretto &V, size_t n) const {
        return verify_multiple_with_rng(bp_gens, pc_gens, transcript, {V}, n, nullptr);
    }
    Status verify_single(const BulletproofGens &bp_gens, const PedersenGens &pc_gens, Transcript &&transcript,
                         const CompressedRistretto &V, size_t n) const {
        return verify_multiple_with_rng(bp_gens, pc_gens, transcript, {V}, n, nullptr);
    }

    // Batched form: proofs[i].verify_multiple(bp_gens, pc_gens, &mut transcript.clone(), &commitments[i], n)
    // for all i in one GPU pass (`transcript` itself is not advanced).  All proofs must share the aggregation size m
    // and the byte length.
    static std::vector<Status> verify_batch(const BulletproofGens &bp_gens, const PedersenGens &pc_gens, const Transcript &transcript,
                                            const std::vector<std::vector<uint8_t>> &proofs,
                                            const std::vector<std::vector<CompressedRistretto>> &commitments, size_t n,
                                            const uint8_t *rng64 = nullptr) {
        const size_t nb = proofs.size();
        std::vector<Status> out;
        if (nb == 0) return out;
        bp_gens.check_pedersen(pc_gens);
        const size_t m = commitments.at(0).size(), len = proofs[0].size();
        std::vector<uint8_t> flat(nb * len), vs(nb * m * 32), verdict(nb);
        for (size_t i = 0; i < nb; i++) {
            if (proofs[i].size() != len || commitments.at(i).size() != m) throw std::invalid_argument("verify_batch: proofs of one call share (m, length)");
            std::memcpy(&flat[i * len], proofs[i].data(), len);
            for (size_t j = 0; j < m; j++) std::memcpy(&vs[(i * m + j) * 32], commitments[i][j].data(), 32);
        }
        const int rc = bpgpu_rangeproof_verify_batch_ts(bp_gens.ctx(), n, m, nb, flat.data(), len, vs.data(), transcript.state().data(), 0, rng64,
                                                        verdict.data(), nullptr, nullptr);
        if (rc != BPGPU_OK) throw GpuError(bpgpu_last_error(bp_gens.ctx()));
        for (size_t i = 0; i < nb; i++) out.push_back(verdict[i] == 0 ? Status::Ok() : Status::Err(static_cast<ProofError>(verdict[i])));
        return out;
    }

    // Same verdicts through the batch-combined check (bpgpu_rangeproof_verify_rlc, no counterpart in the crate): one
    // identity test for the whole batch when every proof verifies, per-proof re-verification inside the call when
    // not.  weights64 = nullptr draws the combination weights from the OS CSPRNG.
    // This entry point replays every proof's transcript from its label: `transcript` must be a fresh Transcript(label).
    static std::vector<Status> verify_batch_combined(const BulletproofGens &bp_gens, const PedersenGens &pc_gens, const Transcript &transcript,
                                                     const std::vector<std::vector<uint8_t>> &proofs,
                                                     const std::vector<std::vector<CompressedRistretto>> &commitments, size_t n,
                                                     const uint8_t *rng64 = nullptr, const uint8_t *weights64 = nullptr) {
        const size_t nb = proofs.size();
        std::vector<Status> out;
        if (nb == 0) return out;
        bp_gens.check_pedersen(pc_gens);
        if (!transcript.is_fresh()) throw std::invalid_argument("verify_batch_combined needs a fresh Transcript(label); use verify_batch for pre-bound transcripts");
        const size_t m = commitments.at(0).size(), len = proofs[0].size();
        std::vector<uint8_t> flat(nb * len), vs(nb * m * 32), verdict(nb);
        for (size_t i = 0; i < nb; i++) {
            if (proofs[i].size() != len || commitments.at(i).size() != m) throw std::invalid_argument("verify_batch_combined: proofs of one call share (m, length)");
            std::memcpy(&flat[i * len], proofs[i].data(), len);
            for (size_t j = 0; j < m; j++) std::memcpy(&vs[(i * m + j) * 32], commitments[i][j].data(), 32);
        }
        const int rc = bpgpu_rangeproof_verify_rlc(bp_gens.ctx(), n, m, nb, flat.data(), len, vs.data(), transcript.label().data(),
                                                   transcript.label().size(), rng64, weights64, verdict.data(), nullptr);
        if (rc != BPGPU_OK) throw GpuError(bpgpu_last_error(bp_gens.ctx()));
        for (size_t i = 0; i < nb; i++) out.push_back(verdict[i] == 0 ? Status::Ok() : Status::Err(static_cast<ProofError>(verdict[i])));
        return out;
    }

  private:
    std::vector<uint8_t> bytes_;
};

// LinearProof (src/linear_proof.rs, `pub use` lib.rs:36): <a, b> = c for a committed secret a and a public b.
// from_bytes :350-394, to_bytes :322-331, serialized_size :318-320, verify :175-236.  The reference's verify takes no
// generator object; here the device context that runs the check is passed explicitly (any BulletproofGens holds one).
class LinearProof {
  public:
    static std::variant<LinearProof, ProofError> from_bytes(const uint8_t *slice, size_t len) {
        if (len % 32 != 0) return ProofError::FormatError;
        const size_t ne = len / 32;
        if (ne < 3 || (ne - 3) % 2 != 0 || (ne - 3) / 2 >= 32) return ProofError::FormatError;
        if (!detail::scalar_is_canonical(slice + len - 64) || !detail::scalar_is_canonical(slice + len - 32)) return ProofError::FormatError;
        LinearProof p;
        p.bytes_.assign(slice, slice + len);
        return p;
    }
    static std::variant<LinearProof, ProofError> from_bytes(const std::vector<uint8_t> &v) { return from_bytes(v.data(), v.size()); }
    const std::vector<uint8_t> &to_bytes() const { return bytes_; }
    size_t serialized_size() const { return bytes_.size(); }

    // LinearProof::create(transcript, rng, &C, r, a_vec, b_vec, G_vec, &F, &B) (:40-173) on the GPU, variable time like the
    // reference's.  rng_bytes: the 64 * (2 lg n + 2) bytes the rng would yield to Scalar::random (nullptr = OS CSPRNG).
    static std::variant<LinearProof, ProofError> create(bpgpu_ctx *ctx, Transcript &transcript, const uint8_t *rng_bytes, const CompressedRistretto &C,
                                                        const ScalarBytes &r, const std::vector<ScalarBytes> &a_vec, const std::vector<ScalarBytes> &b_vec,
                                                        const std::vector<CompressedRistretto> &G_vec, const CompressedRistretto &F,
                                                        const CompressedRistretto &B) {
        const size_t n = b_vec.size();
        if (G_vec.size() != n) return ProofError::InvalidGeneratorsLength;                                        // :61-63
        if (a_vec.size() != n || n == 0 || (n & (n - 1))) throw std::invalid_argument("InvalidInputLength");       // :64-70
        size_t lg = 0;
        while ((size_t(1) << lg) < n) lg++;
        LinearProof p;
        p.bytes_.resize(32 * (2 * lg + 3));
        uint8_t status = 0;
        std::array<uint8_t, BPGPU_TRANSCRIPT_BYTES> in = transcript.state();
        const int rc = bpgpu_linear_create_batch(ctx, n, 1, nullptr, 0, in.data(), rng_bytes, C.data(), r.data(), a_vec[0].data(), b_vec[0].data(), 0,
                                                 G_vec[0].data(), F.data(), B.data(), p.bytes_.data(), &status, transcript.state_mut().data());
        if (rc != BPGPU_OK) throw GpuError(bpgpu_last_error(ctx));
        if (status != 0) throw std::invalid_argument("an input point does not decode or a scalar is not canonical");
        return p;
    }

    // LinearProof::verify(&self, transcript, C, G, F, B, b_vec); the transcript is left advanced
    Status verify(bpgpu_ctx *ctx, Transcript &transcript, const CompressedRistretto &C, const std::vector<CompressedRistretto> &G,
                  const CompressedRistretto &F, const CompressedRistretto &B, const std::vector<ScalarBytes> &b_vec) const {
        if (G.size() != b_vec.size()) return Status::Err(ProofError::InvalidGeneratorsLength);   // :189-191
        uint8_t verdict = 0;
        std::array<uint8_t, BPGPU_TRANSCRIPT_BYTES> in = transcript.state();
        const int rc = bpgpu_linear_verify_batch(ctx, b_vec.size(), 1, bytes_.data(), bytes_.size(), nullptr, 0, in.data(), C.data(),
                                                 G.empty() ? nullptr : G[0].data(), F.data(), B.data(), b_vec.empty() ? nullptr : b_vec[0].data(), 0,
                                                 &verdict, nullptr, transcript.state_mut().data());
        if (rc != BPGPU_OK) throw GpuError(bpgpu_last_error(ctx));
        return verdict == 0 ? Status::Ok() : Status::Err(static_cast<ProofError>(verdict));
    }
    // many proofs of one size over the same G, F, B in one GPU pass; b_vecs holds one public vector per proof.
    // `transcript` is cloned per proof and not advanced.
    static std::vector<Status> verify_batch(bpgpu_ctx *ctx, const Transcript &transcript, const std::vector<LinearProof> &proofs,
                                            const std::vector<CompressedRistretto> &Cs, const std::vector<CompressedRistretto> &G,
                                            const CompressedRistretto &F, const CompressedRistretto &B,
                                            const std::vector<std::vector<ScalarBytes>> &b_vecs) {
        const size_t nb = proofs.size(), n = G.size();
        std::vector<Status> out;
        if (nb == 0) return out;
        if (Cs.size() != nb || b_vecs.size() != nb) throw std::invalid_argument("one commitment and one public vector per proof");
        const size_t pl = proofs[0].bytes_.size();
        std::vector<uint8_t> pb, cb, bb, verdict(nb);
        for (size_t i = 0; i < nb; i++) {
            if (proofs[i].bytes_.size() != pl || b_vecs[i].size() != n) throw std::invalid_argument("proofs of a batch must have one size");
            pb.insert(pb.end(), proofs[i].bytes_.begin(), proofs[i].bytes_.end());
            cb.insert(cb.end(), Cs[i].begin(), Cs[i].end());
            for (const auto &x : b_vecs[i]) bb.insert(bb.end(), x.begin(), x.end());
        }
        const int rc = bpgpu_linear_verify_batch(ctx, n, nb, pb.data(), pl, nullptr, 0, transcript.state().data(), cb.data(),
                                                 n ? G[0].data() : nullptr, F.data(), B.data(), n ? bb.data() : nullptr, 0, verdict.data(), nullptr,
                                                 nullptr);
        if (rc != BPGPU_OK) throw GpuError(bpgpu_last_error(ctx));
        for (uint8_t v : verdict) out.push_back(v == 0 ? Status::Ok() : Status::Err(static_cast<ProofError>(v)));
        return out;
    }

  private:
    std::vector<uint8_t> bytes_;
};

}  // namespace bulletproofs
#endif
