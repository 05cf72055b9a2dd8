"""Option `transcript_coop` (default on; csrc/keccak.h: keccak_f1600_masked_coop, csrc/rangeproof.h: rp_transcript_scripted_coop, k_rp_stage1_coop):
launch chains of up to 256 proofs replay their transcripts 32 lanes per proof -- the group's leader walks the per-shape script, all
lanes run the Keccak-f[1600] rounds together (one 64-bit state word per lane, exchanges by ds_bpermute).  Everything must be
bit-identical to the default path (lane = proof) and to the oracle: verdicts, mega-check encodings, advanced transcripts."""
import hashlib

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair():
    import bulletproofs_amd as bp
    a, b = bp.Context(0), bp.Context(0)
    a.set_option("transcript_coop", 0)     # lane = proof (what wide chains always take)
    b.set_option("transcript_coop", 1)     # the default
    a.gens_create(64, 8)
    b.gens_create(64, 8)
    yield a, b
    a.close()
    b.close()


def test_golden_shapes_and_ragged_widths(pair, oracle, oracle_gens_64_8, golden):
    ref, coop = pair
    label, vc = golden["label"], golden["vc_bytes"]
    for case in golden["cases"]:
        n, m = case["n"], case["m"]
        pr = bytes.fromhex(case["proof"])
        bad = bytearray(pr)
        bad[128] ^= 1                      # wrong t_x: runs the whole transcript, fails the check
        fmt = bytearray(pr)
        fmt[160:192] = b"\xff" * 32        # FormatError: the group permutes a state nobody reads
        ident = bytearray(pr)
        ident[224:256] = bytes(32)         # L_0 = identity encoding: VerificationError raised by the transcript role
        for nb in (1, 2, 3, 5, 64):
            batch = [[pr, bytes(bad), bytes(fmt), bytes(ident), pr][i % 5] for i in range(nb)]
            proofs, coms = b"".join(batch), vc[:32 * m] * nb
            rng = hashlib.shake_256(b"coop-%d-%d-%d" % (n, m, nb)).digest(64 * nb)
            v0, e0 = ref.rangeproof_verify_batch(n, m, proofs, len(pr), coms, label, rng, want_msm=True)
            v1, e1 = coop.rangeproof_verify_batch(n, m, proofs, len(pr), coms, label, rng, want_msm=True)
            assert v1 == v0 and e1 == e0, (n, m, nb)
            assert list(v1) == [[0, 1, 2, 1, 0][i % 5] for i in range(nb)]
        rc, enc = oracle.verify(oracle_gens_64_8, bytes(bad), vc[:32 * m], n, label, rng[64:128])[:2]
        assert rc == 1 and enc == e1[32:64]


def test_widths_around_the_limit_and_caller_transcripts(pair, oracle, oracle_gens_64_8):
    from bulletproofs_amd import workload as wl
    ref, coop = pair
    fx = wl.load_fixture("cfg2_n64_m1")
    for nb in (255, 256, 257):             # 257: wider than the option's limit -> the default launch
        proofs, coms = wl.tile_batch(fx, nb, first=11)
        pb = bytearray(proofs)
        pb[100 * fx.proof_len + 128] ^= 1
        proofs = bytes(pb)
        rng = hashlib.shake_256(b"coop-w%d" % nb).digest(64 * nb)
        v0, e0 = ref.rangeproof_verify_batch(fx.n, fx.m, proofs, fx.proof_len, coms, fx.label, rng, want_msm=True)
        v1, e1 = coop.rangeproof_verify_batch(fx.n, fx.m, proofs, fx.proof_len, coms, fx.label, rng, want_msm=True)
        assert v1 == v0 and e1 == e0 and [i for i in range(nb) if v1[i]] == [100]
    # one transcript per proof, all at one STROBE position (what the combining queue hands a chain), states in and out
    n, m, nb = 32, 2, 9
    pl = oracle.proof_len(n, m)
    states, proofs, coms = [], b"", b""
    for i in range(nb):
        st = oracle.transcript_append_message(oracle.transcript_new(b"coop app"), b"session", hashlib.shake_256(b"s%d" % i).digest(24))
        vals = [int.from_bytes(hashlib.shake_256(b"cv%d-%d" % (i, j)).digest(4), "little") for j in range(m)]
        bl = b"".join(hashlib.shake_256(b"cb%d-%d" % (i, j)).digest(31) + b"\x00" for j in range(m))
        pr, cm, _ = oracle.prove_ts(oracle_gens_64_8, vals, bl, n, st, b"seed%d" % i)
        states.append(st)
        proofs += pr
        coms += cm
    pb = bytearray(proofs)
    pb[4 * pl + 128] ^= 1
    proofs = bytes(pb)
    rng = hashlib.shake_256(b"coop-ts").digest(64 * nb)
    r0 = ref.rangeproof_verify_batch_ts(n, m, proofs, pl, coms, b"".join(states), rng, want_msm=True, want_transcripts=True)
    r1 = coop.rangeproof_verify_batch_ts(n, m, proofs, pl, coms, b"".join(states), rng, want_msm=True, want_transcripts=True)
    assert r1 == r0 and list(r1[0]) == [0, 0, 0, 0, 1, 0, 0, 0, 0]
    for i in (0, 4, 8):
        rc, emsm, est = oracle.verify_ts(oracle_gens_64_8, proofs[pl * i:pl * (i + 1)], coms[32 * m * i:32 * m * (i + 1)], n, states[i], rng[64 * i:64 * i + 64])
        assert r1[0][i] == rc and r1[1][32 * i:32 * i + 32] == emsm and r1[2][208 * i:208 * (i + 1)] == est
