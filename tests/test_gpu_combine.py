"""The pool's combining queue (bpgpu_pool_rangeproof_verify_ts / _submit_ts, include/bpgpu.h): the reference's OWN call shape --
RangeProof::verify_multiple_with_rng(bp_gens, pc_gens, transcript: &mut Transcript, ...) for ONE proof, blocking, from any number
of threads (src/range_proof/mod.rs:345-353, 455-470) -- served by launch chains that the calls share.  Whatever the queue does
(which calls end up in which chain, scripted or byte-wise replay, one device or two), every verdict byte, every 32-byte
mega-check encoding and every advanced transcript must equal oracle.verify_ts on the same inputs."""
import hashlib
import os
import threading

import pytest

pytestmark = pytest.mark.gpu

from test_gpu_bench_config import _tamper  # noqa: E402

TS = 208


def _state(oracle, kind, i=0):
    """Transcripts as applications hand them over: fresh (kind 0), or with messages of their own already absorbed --
    kinds 1..: different amounts, i.e. different STROBE positions."""
    st = oracle.transcript_new(b"combine-test v1")
    if kind == 0:
        return st
    st = oracle.transcript_append_message(st, b"session", hashlib.shake_256(b"sess%d" % i).digest(16))
    if kind >= 2:
        st = oracle.transcript_append_message(st, b"ctx", bytes([i & 0xff]) * (3 * kind))
    if kind >= 3:
        st, _ = oracle.transcript_challenge_bytes(st, b"binding", 16)
    return st


def _make(oracle, gens, n, m, count, kinds, seed):
    """`count` proofs of shape (n, m), proof i proven on a transcript of kind kinds[i % len(kinds)]; a few tampered."""
    items = []
    for i in range(count):
        st = _state(oracle, kinds[i % len(kinds)], i)
        vals = [int.from_bytes(hashlib.shake_256(b"%s-v%d-%d" % (seed, i, j)).digest(8), "little") % (1 << n) for j in range(m)]
        bl = b"".join(hashlib.shake_256(b"%s-b%d-%d" % (seed, i, j)).digest(31) + b"\x00" for j in range(m))
        pr, cm, _ = oracle.prove_ts(gens, vals, bl, n, st, b"%s%d" % (seed, i))
        pr = bytearray(pr)
        if i % 11 == 3:
            pr[130] ^= 1                      # wrong t_x: VerificationError
        if i % 17 == 5:
            pr[165:192] = b"\xff" * 27        # non-canonical t_x_blinding: FormatError, transcript untouched
        if i % 23 == 7:
            st = _state(oracle, 3, 9999)      # right proof, wrong history
        rng = hashlib.shake_256(b"%s-r%d" % (seed, i)).digest(64)
        items.append((bytes(pr), cm, st, rng))
    return items


def _expect(oracle, gens, n, items):
    return [oracle.verify_ts(gens, pr, cm, n, st, rng) for pr, cm, st, rng in items]


def _check(i, got, exp, st_in):
    v, msm, ts = got
    rc, emsm, est = exp
    assert v[0] == rc, (i, v[0], rc)
    if rc in (0, 1) and emsm != b"\xff" * 32:
        assert msm == emsm, i
    if rc == 2:
        assert ts == st_in, i                 # from_bytes failed: the caller's transcript was never touched
    assert ts == est, i                       # (also after an early Err: the state as of the offending message, tests/test_gpu_transcript_stop.py)


@pytest.fixture(scope="module")
def pool64():
    import bulletproofs_amd as bp
    p = bp.Pool((0,), 8, fixed_window_bits=16)
    p.gens_create(64, 4)
    yield p
    p.close()


def test_single_proof_calls_from_64_threads_own_transcripts_vs_oracle(oracle, pool64):
    """64 host threads, each looping blocking ONE-proof calls with its own transcript (half of them with application messages
    absorbed: three STROBE position classes) -- the literal verify_multiple_with_rng shape.  Everything == oracle, and the calls
    did share chains."""
    n, m, per_thread, T = 64, 1, 5, 64
    gens = oracle.Gens(64, 4)
    items = _make(oracle, gens, n, m, T * per_thread, [0, 1, 0, 2, 0, 3, 0, 1], b"c64")
    exp = _expect(oracle, gens, n, items)
    pl = oracle.proof_len(n, m)
    pool64.set_option("stat_reset", 1)
    got = [None] * len(items)
    errs = []

    def worker(t):
        try:
            for j in range(per_thread):
                i = t * per_thread + j
                pr, cm, st, rng = items[i]
                got[i] = pool64.rangeproof_verify_ts(n, m, pr, pl, cm, st, rng, want_msm=True, want_transcripts=True)
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    for i, (g, e) in enumerate(zip(got, exp)):
        _check(i, g, e, items[i][2])
    assert sum(1 for e in exp if e[0] == 0) > len(items) // 2
    chains, reqs, proofs = (pool64.get_option("stat_combined_" + k) for k in ("chains", "requests", "proofs"))
    assert reqs == len(items) and proofs == sum(1 for it in items if len(it[0]) == pl)
    assert chains < reqs / 2, (chains, reqs)      # calls were combined (round 6's narrow chain is a third shorter: Python threads meet in it less often than the 3 per chain of round 5)


def test_many_position_classes_fall_back_to_bytewise_chain(oracle, pool64):
    """More STROBE position classes in flight than `combine_max_open`: the overflow shares a catch-all chain that replays the
    transcripts byte by byte.  Same results."""
    n, m = 32, 2
    gens = oracle.Gens(64, 4)
    pl = oracle.proof_len(n, m)
    items = []
    for i in range(60):
        st = oracle.transcript_new(b"classes")
        st = oracle.transcript_append_message(st, b"pad", b"\x07" * (i % 12))      # 12 different positions
        pr, cm, _ = oracle.prove_ts(gens, [i + 1, 2 ** 31 + i], bytes(64), n, st, b"cl%d" % i)
        items.append((pr, cm, st, hashlib.shake_256(b"clr%d" % i).digest(64)))
    exp = _expect(oracle, gens, n, items)
    pool64.set_option("combine_max_open", 2)
    try:
        tickets = [pool64.submit_ts(n, m, pr, pl, cm, st, rng, want_msm=True, want_transcripts=True) for pr, cm, st, rng in items]
        got = [t.wait() for t in tickets]
    finally:
        pool64.set_option("combine_max_open", 4)
    for i, (g, e) in enumerate(zip(got, exp)):
        _check(i, g, e, items[i][2])
    assert all(e[0] == 0 for e in exp)
    # one request holding proofs at several positions: cut into stretches of one position each
    proofs = b"".join(it[0] for it in items[:30])
    coms = b"".join(it[1] for it in items[:30])
    states = b"".join(it[2] for it in items[:30])
    rng = b"".join(it[3] for it in items[:30])
    v, msm, ts = pool64.rangeproof_verify_ts(n, m, proofs, pl, coms, states, rng, want_msm=True, want_transcripts=True)
    for i in range(30):
        _check(i, (v[i:i + 1], msm[32 * i:32 * i + 32], ts[TS * i:TS * i + TS]), exp[i], items[i][2])


def test_tickets_one_thread_keeps_hundreds_of_requests_in_flight(oracle, pool64):
    n, m = 64, 1
    gens = oracle.Gens(64, 4)
    pl = oracle.proof_len(n, m)
    items = _make(oracle, gens, n, m, 120, [0, 2], b"tk")
    exp = _expect(oracle, gens, n, items)
    # every request three times: 360 tickets outstanding at once, from one thread
    tickets = [pool64.submit_ts(n, m, pr, pl, cm, st, rng, want_msm=True, want_transcripts=True) for _ in range(3) for pr, cm, st, rng in items]
    got = [t.wait() for t in tickets]
    for r in range(3):
        for i in range(len(items)):
            _check(i, got[r * len(items) + i], exp[i], items[i][2])
    assert all(t.done() for t in tickets)
    # library-drawn batching challenges (rng64 = NULL): same verdicts
    tickets = [pool64.submit_ts(n, m, pr, pl, cm, st, None, want_transcripts=False) for pr, cm, st, _ in items]
    assert [t.wait()[0] for t in tickets] == [e[0] for e in exp]


def test_large_calls_from_two_threads_overlap_and_match_oracle(oracle):
    """Two threads, each ONE blocking call of 6 000 label-mode proofs (bpgpu_pool_rangeproof_verify: Transcript::new(label) for every
    proof), ~5 % tampered; then the same proofs as one shared transcript with the advanced states handed back."""
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    fx = wl.load_fixture("cfg2_n64_m1")
    nb = 6000
    pool = bp.Pool((0,), 8, fixed_window_bits=16)
    pool.gens_create(64, 1)
    gens = oracle.Gens(64, 1)
    data = []
    for t in range(2):
        proofs, coms = wl.tile_batch(fx, nb)
        proofs, coms, bad = _tamper(proofs, coms, fx.proof_len, fx.m, nb, 100 + t)
        rng = hashlib.shake_256(b"two%d" % t).digest(64 * nb)
        data.append((proofs, coms, rng))
    out = [None, None]

    def run(t):
        out[t] = pool.rangeproof_verify(fx.n, fx.m, data[t][0], fx.proof_len, data[t][1], fx.label, data[t][2], want_msm=True)

    th = [threading.Thread(target=run, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for t in range(2):
        _, ev, em = oracle.verify_batch(gens, data[t][0], data[t][1], fx.m, fx.n, fx.label, data[t][2], threads=os.cpu_count() or 1)
        assert out[t][0] == ev
        for i in range(nb):
            if ev[i] in (0, 1) and em[32 * i:32 * i + 32] != b"\xff" * 32:
                assert out[t][1][32 * i:32 * i + 32] == em[32 * i:32 * i + 32], (t, i)
    # one shared start state, states handed back: proof i's advanced state == the oracle's for the same proof
    st0 = oracle.transcript_new(fx.label)
    k = 700
    v, ts = pool.rangeproof_verify_ts(fx.n, fx.m, data[0][0][:k * fx.proof_len], fx.proof_len, data[0][1][:k * 32 * fx.m], st0, data[0][2][:64 * k])
    assert v == out[0][0][:k]
    for i in (0, 1, 17, 333, k - 1):
        rc, _, est = oracle.verify_ts(gens, data[0][0][i * fx.proof_len:(i + 1) * fx.proof_len], data[0][1][32 * i:32 * i + 32], fx.n, st0, data[0][2][64 * i:64 * i + 64])
        assert v[i] == rc and (rc == 2 or ts[TS * i:TS * i + TS] == est), i
    pool.close()


def test_requests_no_chain_can_take_are_reported_per_proof(oracle, pool64):
    """Malformed lengths, an invalid bit size, too few generators: the ordinary entry point answers, proof by proof, in the
    reference's order of checks (mod.rs:358-366, 505-510); a malformed transcript state is an argument error."""
    import bulletproofs_amd as bp
    n, m = 64, 1
    gens = oracle.Gens(64, 4)
    pl = oracle.proof_len(n, m)
    st = oracle.transcript_new(b"odd")
    pr, cm, _ = oracle.prove_ts(gens, [77], bytes(32), n, st, b"odd")
    v, ts = pool64.rangeproof_verify_ts(n, m, pr[:-32], pl - 32, cm, st)            # truncated proof
    assert v == b"\x02" and ts == st
    v, _ = pool64.rangeproof_verify_ts(24, m, pr, pl, cm, st)                        # InvalidBitsize
    assert v == b"\x03"
    v, _ = pool64.rangeproof_verify_ts(n, 8, pr, pl, cm * 8, st)                     # party capacity 4 < 8
    assert v == b"\x04"
    bad = bytearray(st)
    bad[200] = 200
    with pytest.raises(bp.BpgpuError):
        pool64.rangeproof_verify_ts(n, m, pr, pl, cm, bytes(bad))
    assert pool64.rangeproof_verify_ts(n, m, b"", pl, b"", st, want_transcripts=False) == b""
    v, ts = pool64.rangeproof_verify_ts(n, m, pr, pl, cm, st)
    assert v == b"\x00"


def test_two_shards_on_one_gpu(oracle):
    """A pool of two devices (the one GPU twice): a large request takes a contiguous shard per device, small ones alternate."""
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    fx = wl.load_fixture("cfg2_n64_m1")
    nb = 3001
    pool = bp.Pool((0, 0), 4, fixed_window_bits=16)
    pool.gens_create(64, 1)
    gens = oracle.Gens(64, 1)
    proofs, coms = wl.tile_batch(fx, nb)
    proofs, coms, bad = _tamper(proofs, coms, fx.proof_len, fx.m, nb, 5)
    rng = hashlib.shake_256(b"shards").digest(64 * nb)
    st0 = oracle.transcript_new(fx.label)
    v, msm, ts = pool.rangeproof_verify_ts(fx.n, fx.m, proofs, fx.proof_len, coms, st0 * nb, rng, want_msm=True)
    _, ev, em = oracle.verify_batch(gens, proofs, coms, fx.m, fx.n, fx.label, rng, threads=os.cpu_count() or 1)
    assert v == ev
    for i in range(0, nb, 97):
        rc, emsm, est = oracle.verify_ts(gens, proofs[i * fx.proof_len:(i + 1) * fx.proof_len], coms[32 * i:32 * i + 32], fx.n, st0, rng[64 * i:64 * i + 64])
        assert rc == v[i] and (rc == 2 or ts[TS * i:TS * i + TS] == est)
    singles = [pool.submit_ts(fx.n, fx.m, proofs[i * fx.proof_len:(i + 1) * fx.proof_len], fx.proof_len, coms[32 * i:32 * i + 32], st0, rng[64 * i:64 * i + 64],
                              want_transcripts=False) for i in range(40)]
    assert b"".join(t.wait() for t in singles) == ev[:40]
    pool.close()
