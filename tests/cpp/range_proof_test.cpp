// C++ rendition of the reference's tests/range_proof.rs::deserialize_and_verify (lines 16-95) on top of
// include/bulletproofs.hpp (the host-side mirror of the crate API over the C ABI).  The golden hex vectors
// come from tests/golden/rangeproof_v1.json via the generated golden_vectors.inc.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "bulletproofs.hpp"
#include "golden_vectors.inc"   // GOLDEN_PROOFS[4][4], GOLDEN_VC[8], GOLDEN_LABEL

using namespace bulletproofs;

static std::vector<uint8_t> hex_decode(const char *h) {
    std::vector<uint8_t> out;
    for (size_t i = 0; h[i] && h[i + 1]; i += 2) out.push_back((uint8_t)std::stoul(std::string(h + i, 2), nullptr, 16));
    return out;
}
#define CHECK(cond)                                                         \
    do {                                                                    \
        if (!(cond)) {                                                      \
            std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            return 1;                                                       \
        }                                                                   \
    } while (0)

int main() {
    // let pc_gens = PedersenGens::default(); let bp_gens = BulletproofGens::new(64, 8);
    BulletproofGens bp_gens(64, 8);
    PedersenGens pc_gens = bp_gens.pedersen();
    std::vector<CompressedRistretto> vc;
    for (int k = 0; k < 8; k++) {
        auto b = hex_decode(GOLDEN_VC[k]);
        CompressedRistretto c;
        std::memcpy(c.data(), b.data(), 32);
        vc.push_back(c);
    }
    for (int i = 0; i < 4; i++) {
        for (int j = 0; j < 4; j++) {
            const size_t n = 8u << i, m = 1u << j;
            auto parsed = RangeProof::from_bytes(hex_decode(GOLDEN_PROOFS[i][j]));
            CHECK(std::holds_alternative<RangeProof>(parsed));   // .expect("Rangeproof deserialization failed")
            const RangeProof &proof = std::get<RangeProof>(parsed);
            Transcript transcript(GOLDEN_LABEL);
            std::vector<CompressedRistretto> vs(vc.begin(), vc.begin() + m);
            CHECK(proof.verify_multiple(bp_gens, pc_gens, transcript, vs, n) == Status::Ok());
            // and the negative directions the reference only covers indirectly
            Transcript other("other label");
            CHECK(proof.verify_multiple(bp_gens, pc_gens, other, vs, n) == Status::Err(ProofError::VerificationError));
            CHECK(proof.verify_multiple(bp_gens, pc_gens, Transcript(GOLDEN_LABEL), vs, 12) == Status::Err(ProofError::InvalidBitsize));
            if (m == 1) CHECK(proof.verify_single(bp_gens, pc_gens, Transcript(GOLDEN_LABEL), vc[0], n) == Status::Ok());
            // `transcript` is &mut: the first call left it advanced, so it no longer matches a second verification
            CHECK(!transcript.is_fresh());
            CHECK(proof.verify_multiple(bp_gens, pc_gens, transcript, vs, n) == Status::Err(ProofError::VerificationError));
            // a transcript that absorbed an application message first is a different statement
            Transcript bound(GOLDEN_LABEL);
            bound.append_message("app", reinterpret_cast<const uint8_t *>("ctx"), 3);
            CHECK(proof.verify_multiple(bp_gens, pc_gens, bound, vs, n) == Status::Err(ProofError::VerificationError));
        }
    }
    // from_bytes error behaviour (mod.rs:505-524)
    auto good = hex_decode(GOLDEN_PROOFS[0][0]);
    CHECK(std::get<ProofError>(RangeProof::from_bytes(good.data(), good.size() - 1)) == ProofError::FormatError);
    CHECK(std::get<ProofError>(RangeProof::from_bytes(good.data(), 6 * 32)) == ProofError::FormatError);
    auto bad = good;
    std::memset(&bad[4 * 32], 0xff, 32);
    CHECK(std::get<ProofError>(RangeProof::from_bytes(bad)) == ProofError::FormatError);
    // too few generators
    BulletproofGens small(8, 1);
    auto p16 = std::get<RangeProof>(RangeProof::from_bytes(hex_decode(GOLDEN_PROOFS[1][0])));
    CHECK(p16.verify_single(small, small.pedersen(), Transcript(GOLDEN_LABEL), vc[0], 16) == Status::Err(ProofError::InvalidGeneratorsLength));
    // a PedersenGens other than the one in the device tables is refused, not silently ignored
    {
        PedersenGens other_pc = pc_gens;
        other_pc.B = vc[0];
        bool threw = false;
        try {
            (void)p16.verify_single(bp_gens, other_pc, Transcript(GOLDEN_LABEL), vc[0], 16);
        } catch (const std::invalid_argument &) {
            threw = true;
        }
        CHECK(threw);
    }
    // prove on the GPU, verify on the GPU: the README example of the reference (create and verify a 32-bit range proof)
    {
        Transcript prover_transcript("doctest example");
        ScalarBytes blinding{};
        blinding[0] = 7;
        auto made = RangeProof::prove_single_with_rng(bp_gens, pc_gens, prover_transcript, 1037578891ull, blinding, 32);
        Transcript verifier_transcript("doctest example");
        CHECK(made.first.verify_single(bp_gens, pc_gens, verifier_transcript, made.second, 32) == Status::Ok());
        CHECK(verifier_transcript.state() == prover_transcript.state());
        CHECK(made.first.verify_single(bp_gens, pc_gens, Transcript("another"), made.second, 32) == Status::Err(ProofError::VerificationError));
    }
    // batched form
    std::vector<std::vector<uint8_t>> proofs(5, hex_decode(GOLDEN_PROOFS[3][0]));
    proofs[2][130] ^= 1;
    std::vector<std::vector<CompressedRistretto>> coms(5, std::vector<CompressedRistretto>{vc[0]});
    auto res = RangeProof::verify_batch(bp_gens, pc_gens, Transcript(GOLDEN_LABEL), proofs, coms, 64);
    for (int k = 0; k < 5; k++) CHECK(res[k] == (k == 2 ? Status::Err(ProofError::VerificationError) : Status::Ok()));
    // the batch-combined entry point gives the same verdicts (per-proof fallback inside when the combination fails)
    auto res2 = RangeProof::verify_batch_combined(bp_gens, pc_gens, Transcript(GOLDEN_LABEL), proofs, coms, 64);
    for (int k = 0; k < 5; k++) CHECK(res2[k] == res[k]);
    proofs[2][130] ^= 1;
    auto res3 = RangeProof::verify_batch_combined(bp_gens, pc_gens, Transcript(GOLDEN_LABEL), proofs, coms, 64);
    for (int k = 0; k < 5; k++) CHECK(res3[k] == Status::Ok());
    std::printf("deserialize_and_verify: ok\n");
    return 0;
}
