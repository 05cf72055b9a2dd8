"""GPU parity for bpgpu_rangeproof_audit_shares (ProofShare::audit_share, src/range_proof/messages.rs:85-167 -- the blame path
of Dealer::receive_shares, dealer.rs:303-335) against the oracle's restatement, and the reference's own scenario
detect_dishonest_party_during_aggregation (src/range_proof/mod.rs:726-799)."""
import hashlib

import pytest

from test_device_code_on_cpu import _share_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import bulletproofs_amd as bp
    c = bp.Context(0)
    c.gens_create(64, 4)
    yield c
    c.close()


@pytest.mark.parametrize("n,m", [(8, 1), (16, 4), (32, 2), (64, 4)])
def test_audit_shares_matches_oracle(ctx, oracle, n, m):
    g = oracle.Gens(64, 4)
    idx, sh, bc, pc, chal, expect = _share_cases(oracle, g, n, m, b"gaud-%d-%d" % (n, m))
    verdict, chk = ctx.rangeproof_audit_shares(n, idx, b"".join(sh), b"".join(bc), b"".join(pc), chal, want_checks=True)
    assert list(verdict) == expect
    for k in range(len(idx)):
        rc, out = oracle.audit_share(g, n, idx[k], sh[k], bc[k], pc[k], chal)
        assert verdict[k] == rc, k
        for h in (0, 1):
            if out[32 * h:32 * h + 32] != b"\xff" * 32:
                assert chk[64 * k + 32 * h:64 * k + 32 * h + 32] == out[32 * h:32 * h + 32], (k, h)   # bit-exact, also the non-identity points


def test_detect_dishonest_party_during_aggregation(ctx, oracle):
    """mod.rs:726-799: four parties, n = 32; parties 1 and 3 commit to 64-bit values.  The aggregated proof does not verify and the
    audit names exactly [1, 3] -- in one batched call, mixed with the shares of an honest aggregation that has other challenges."""
    n, m = 32, 4
    g = oracle.Gens(64, 4)
    rnd = hashlib.shake_256(b"dishonest").digest(8 * 8)
    u = lambda i: int.from_bytes(rnd[8 * i:8 * i + 8], "little")
    bl = b"".join(hashlib.shake_256(b"dbl%d" % i).digest(31) + b"\x00" for i in range(m))
    bad = oracle.prove_shares(g, [u(0) & 0xffffffff, u(1) | (1 << 63), u(2) & 0xffffffff, u(3) | (1 << 62)], bl, n, b"AggregatedRangeProofTest", b"s1")
    good = oracle.prove_shares(g, [u(4) & 0xffffffff, u(5) & 0xffffffff, u(6) & 0xffffffff, u(7) & 0xffffffff], bl, n, b"AggregatedRangeProofTest", b"s2")
    pl = len(bad["proof"])
    v = ctx.rangeproof_verify_batch(n, m, bad["proof"] + good["proof"], pl, bad["commitments"] + good["commitments"], b"AggregatedRangeProofTest")
    assert list(v) == [1, 0]                                        # the dealer's own check of the aggregated proof (dealer.rs:303-312)
    chal = bad["challenges"] * m + good["challenges"] * m          # per-share challenges: two aggregations in one call
    verdict = ctx.rangeproof_audit_shares(n, list(range(m)) * 2, bad["shares"] + good["shares"], bad["bit_commitments"] + good["bit_commitments"],
                                          bad["poly_commitments"] + good["poly_commitments"], chal)
    bad_shares = [j for j in range(m) if verdict[j] != 0]
    assert bad_shares == [1, 3] and list(verdict[m:]) == [0] * m   # MPCError::MalformedProofShares { bad_shares: [1, 3] }
    sl = 32 * (3 + 2 * n)
    for j in range(m):
        assert oracle.audit_share(g, n, j, bad["shares"][sl * j:sl * (j + 1)], bad["bit_commitments"][96 * j:96 * j + 96],
                                  bad["poly_commitments"][64 * j:64 * j + 64], bad["challenges"])[0] == verdict[j]


def test_audit_shares_parameter_checks(ctx, oracle):
    import bulletproofs_amd as bp
    g = oracle.Gens(64, 4)
    r = oracle.prove_shares(g, [5], bytes(32), 8, b"x", b"s")
    with pytest.raises(bp.BpgpuError):                              # not a bitsize (party.rs:41-43)
        ctx.rangeproof_audit_shares(12, [0], bytes(32 * 27), r["bit_commitments"], r["poly_commitments"], r["challenges"])
    small = bp.Context(0)
    small.gens_create(8, 1)
    r16 = oracle.prove_shares(g, [5], bytes(32), 16, b"x", b"s")
    assert list(small.rangeproof_audit_shares(16, [0], r16["shares"], r16["bit_commitments"], r16["poly_commitments"], r16["challenges"])) == [1]   # check_size
    assert list(small.rangeproof_audit_shares(8, [0], r["shares"], r["bit_commitments"], r["poly_commitments"], r["challenges"])) == [0]
    small.close()
