"""The reference's own integration test, tests/range_proof.rs::deserialize_and_verify (lines 16-95),
written against the host-side mirrors of the crate API: Python (bulletproofs_amd.api) and C++
(include/bulletproofs.hpp, compiled here with g++ and linked against libbpgpu.so)."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_deserialize_and_verify(golden):
    from bulletproofs_amd import BulletproofGens, RangeProof, Transcript, VerificationError, InvalidBitsize, FormatError
    # proofs[i][j] has bitsize n = 8 << i, aggregation size m = 1 << j
    proofs = [[golden["cases"][4 * i + j]["proof"] for j in range(4)] for i in range(4)]
    vc = [bytes.fromhex(v) for v in golden["value_commitments"]]
    bp_gens = BulletproofGens(64, 8)
    pc_gens = bp_gens.pedersen()
    for i in range(4):
        for j in range(4):
            n, m = 8 << i, 1 << j
            proof = RangeProof.from_bytes(bytes.fromhex(proofs[i][j]))
            transcript = Transcript(b"Deserialize-And-Verify Test")
            assert proof.verify_multiple(bp_gens, pc_gens, transcript, vc[0:m], n) is None   # == Ok(())
            with pytest.raises(VerificationError):
                proof.verify_multiple(bp_gens, pc_gens, Transcript(b"another label"), vc[0:m], n)
            before = transcript.state
            with pytest.raises(InvalidBitsize):
                proof.verify_multiple(bp_gens, pc_gens, transcript, vc[0:m], 24)
            assert transcript.state == before      # the bitsize check precedes any transcript operation (mod.rs:358-360)
    with pytest.raises(FormatError):
        RangeProof.from_bytes(bytes.fromhex(proofs[0][0])[:-1])
    res = RangeProof.verify_batch(bp_gens, pc_gens, Transcript(b"Deserialize-And-Verify Test"),
                                  [bytes.fromhex(proofs[3][0])] * 3, [[vc[0]], [vc[1]], [vc[0]]], 64)
    assert res[0] is None and res[1] == VerificationError() and res[2] is None
    # the batch-combined entry point returns the same verdicts (clean batch: one identity check; otherwise the fallback)
    clean = RangeProof.verify_batch_combined(bp_gens, pc_gens, Transcript(b"Deserialize-And-Verify Test"),
                                             [bytes.fromhex(proofs[3][0])] * 4, [[vc[0]]] * 4, 64)
    assert clean == [None] * 4
    res2 = RangeProof.verify_batch_combined(bp_gens, pc_gens, Transcript(b"Deserialize-And-Verify Test"),
                                            [bytes.fromhex(proofs[3][0])] * 3, [[vc[0]], [vc[1]], [vc[0]]], 64)
    assert res2[0] is None and res2[1] == VerificationError() and res2[2] is None


def test_cpp_mirror_deserialize_and_verify(golden, tmp_path):
    inc = tmp_path / "golden_vectors.inc"
    with open(inc, "w") as f:
        f.write("static const char *GOLDEN_PROOFS[4][4] = {\n")
        for i in range(4):
            f.write("  {" + ", ".join('"%s"' % golden["cases"][4 * i + j]["proof"] for j in range(4)) + "},\n")
        f.write("};\nstatic const char *GOLDEN_VC[8] = {" + ", ".join('"%s"' % v for v in golden["value_commitments"]) + "};\n")
        f.write('static const char *GOLDEN_LABEL = "%s";\n' % golden["transcript_label"])
    exe = tmp_path / "range_proof_test"
    libdir = os.path.join(ROOT, "bulletproofs_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I", str(tmp_path),
                           os.path.join(ROOT, "tests", "cpp", "range_proof_test.cpp"), "-o", str(exe),
                           "-L", libdir, "-lbpgpu", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    assert "deserialize_and_verify: ok" in out.stdout
