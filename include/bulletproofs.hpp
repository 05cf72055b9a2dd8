// bulletproofs.hpp -- host-side mirror (C++17, header-only) of the reference crate's public
// verification surface, implemented over the C ABI of libbpgpu.so (include/bpgpu.h).
//
// Mirrors, with the same names, argument meaning and error behaviour:
//   ProofError ............ src/errors.rs:12-54
//   PedersenGens .......... src/generators.rs:30-53        (default() = basepoint + hashed blinding base)
//   BulletproofGens ....... src/generators.rs:157-204      (new(gens_capacity, party_capacity))
//   Transcript ............ merlin::Transcript::new(label)  (the engine replays it from the label)
//   RangeProof ............ src/range_proof/mod.rs:59-76, from_bytes 504-538, to_bytes 487-500,
//                           verify_single[_with_rng] 316-342, verify_multiple[_with_rng] 345-470
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

  private:
    std::shared_ptr<bpgpu_ctx> ctx_;
};

// merlin::Transcript as far as this path needs it: a fresh transcript named by its label
class Transcript {
  public:
    explicit Transcript(const std::string &label) : label_(label.begin(), label.end()) {}
    Transcript(const uint8_t *label, size_t n) : label_(label, label + n) {}
    const std::vector<uint8_t> &label() const { return label_; }

  private:
    std::vector<uint8_t> label_;
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
    const std::vector<uint8_t> &to_bytes() const { return bytes_; }

    // verify_multiple_with_rng: rng64 = the 64 bytes the rng would hand Scalar::random (mod.rs:396)
    Status verify_multiple_with_rng(const BulletproofGens &bp_gens, const PedersenGens &, const Transcript &transcript,
                                    const std::vector<CompressedRistretto> &value_commitments, size_t n, const uint8_t *rng64) const {
        uint8_t verdict = 0;
        const int rc = bpgpu_rangeproof_verify_batch(bp_gens.ctx(), n, value_commitments.size(), 1, bytes_.data(), bytes_.size(),
                                                     value_commitments.empty() ? nullptr : value_commitments[0].data(), transcript.label().data(),
                                                     transcript.label().size(), rng64, &verdict, nullptr);
        if (rc != BPGPU_OK) throw GpuError(bpgpu_last_error(bp_gens.ctx()));
        return verdict == 0 ? Status::Ok() : Status::Err(static_cast<ProofError>(verdict));
    }
    // verify_multiple: thread_rng() -> the library draws from the OS CSPRNG
    Status verify_multiple(const BulletproofGens &bp_gens, const PedersenGens &pc_gens, const Transcript &transcript,
                           const std::vector<CompressedRistretto> &value_commitments, size_t n) const {
        return verify_multiple_with_rng(bp_gens, pc_gens, transcript, value_commitments, n, nullptr);
    }
    Status verify_single_with_rng(const BulletproofGens &bp_gens, const PedersenGens &pc_gens, const Transcript &transcript,
                                  const CompressedRistretto &V, size_t n, const uint8_t *rng64) const {
        return verify_multiple_with_rng(bp_gens, pc_gens, transcript, {V}, n, rng64);
    }
    Status verify_single(const BulletproofGens &bp_gens, const PedersenGens &pc_gens, const Transcript &transcript,
                         const CompressedRistretto &V, size_t n) const {
        return verify_multiple_with_rng(bp_gens, pc_gens, transcript, {V}, n, nullptr);
    }

    // Batched form: proofs[i].verify_multiple(bp_gens, pc_gens, &mut Transcript::new(label), &commitments[i], n)
    // for all i in one GPU pass.  All proofs must share the aggregation size m and the byte length.
    static std::vector<Status> verify_batch(const BulletproofGens &bp_gens, const PedersenGens &, const Transcript &transcript,
                                            const std::vector<std::vector<uint8_t>> &proofs,
                                            const std::vector<std::vector<CompressedRistretto>> &commitments, size_t n,
                                            const uint8_t *rng64 = nullptr) {
        const size_t nb = proofs.size();
        std::vector<Status> out;
        if (nb == 0) return out;
        const size_t m = commitments.at(0).size(), len = proofs[0].size();
        std::vector<uint8_t> flat(nb * len), vs(nb * m * 32), verdict(nb);
        for (size_t i = 0; i < nb; i++) {
            if (proofs[i].size() != len || commitments.at(i).size() != m) throw std::invalid_argument("verify_batch: proofs of one call share (m, length)");
            std::memcpy(&flat[i * len], proofs[i].data(), len);
            for (size_t j = 0; j < m; j++) std::memcpy(&vs[(i * m + j) * 32], commitments[i][j].data(), 32);
        }
        const int rc = bpgpu_rangeproof_verify_batch(bp_gens.ctx(), n, m, nb, flat.data(), len, vs.data(), transcript.label().data(),
                                                     transcript.label().size(), rng64, verdict.data(), nullptr);
        if (rc != BPGPU_OK) throw GpuError(bpgpu_last_error(bp_gens.ctx()));
        for (size_t i = 0; i < nb; i++) out.push_back(verdict[i] == 0 ? Status::Ok() : Status::Err(static_cast<ProofError>(verdict[i])));
        return out;
    }

    // Same verdicts through the batch-combined check (bpgpu_rangeproof_verify_rlc, no counterpart in the crate): one
    // identity test for the whole batch when every proof verifies, per-proof re-verification inside the call when
    // not.  weights64 = nullptr draws the combination weights from the OS CSPRNG.
    static std::vector<Status> verify_batch_combined(const BulletproofGens &bp_gens, const PedersenGens &, const Transcript &transcript,
                                                     const std::vector<std::vector<uint8_t>> &proofs,
                                                     const std::vector<std::vector<CompressedRistretto>> &commitments, size_t n,
                                                     const uint8_t *rng64 = nullptr, const uint8_t *weights64 = nullptr) {
        const size_t nb = proofs.size();
        std::vector<Status> out;
        if (nb == 0) return out;
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

}  // namespace bulletproofs
#endif
