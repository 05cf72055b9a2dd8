"""One BulletproofGens serves any (n <= capacity, m <= parties) under any label (src/generators.rs:157-259), and callers do not sort
their submissions.  A flush groups the pending device batches by SHAPE (not by adjacency) and lets batches that differ only in their
label share a launch chain (labels of one length share every transcript position; each batch starts from its own state,
rp_seg::init_w): 40 alternating (64, 1) / (64, 16) batches under four labels become a handful of chains, and every verdict and
mega-check encoding equals the oracle's under the batch's own label."""
import hashlib
import os

import pytest

pytestmark = pytest.mark.gpu


def _planted(oracle, gens, n, m, label, count, seed):
    """`count` valid proofs under `label` (the fixtures only hold proofs under their own label)"""
    out = []
    for i in range(count):
        vals = [int.from_bytes(hashlib.shake_256(b"%s-v%d-%d" % (seed, i, j)).digest(8), "little") % (1 << n) for j in range(m)]
        bl = b"".join(hashlib.shake_256(b"%s-b%d-%d" % (seed, i, j)).digest(31) + b"\x00" for j in range(m))
        out.append(oracle.prove(gens, vals, bl, n, label, b"%s-%d" % (seed, i)))
    return out


def test_alternating_shapes_and_four_labels_share_chains_vs_oracle(oracle):
    import torch
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    f1, f16 = wl.load_fixture("cfg2_n64_m1"), wl.load_fixture("cfg3_n64_m16")
    assert f1.label == f16.label
    base = f1.label
    labels = [base, bytes(reversed(base)), bytes((c ^ 0x20) for c in base), b"#" * len(base)]
    assert len(set(labels)) == 4
    dev = torch.device("cuda", 0)
    pool = bp.Pool((0,), 8, fixed_table_max_bytes=24 << 30)
    pool.gens_create(64, 16)
    pool.gens_add_shape(64, 1)
    pool.set_option("auto_flush_items", 1000)
    gens = oracle.Gens(64, 16)
    to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    items = []
    for i in range(40):
        fx, nb = (f1, 256) if i % 2 == 0 else (f16, 64)
        label = labels[(i // 2) % 4]
        proofs, coms = wl.tile_batch(fx, nb, first=(i * 37) % fx.count)
        pb, cb = bytearray(proofs), bytearray(coms)
        # a few proofs that are VALID under this batch's label (the fixture's proofs are valid under the first label only)
        for j, (pr, cm) in enumerate(_planted(oracle, gens, fx.n, fx.m, label, 2, b"co%d" % i)):
            at = 3 + 11 * j
            pb[at * fx.proof_len:(at + 1) * fx.proof_len] = pr
            cb[at * 32 * fx.m:(at + 1) * 32 * fx.m] = cm
        pb[(nb - 1) * fx.proof_len + 129] ^= 2          # one tampered scalar per batch
        proofs, coms = bytes(pb), bytes(cb)
        rng = hashlib.shake_256(b"co-rng%d" % i).digest(64 * nb)
        items.append(dict(fx=fx, nb=nb, label=label, proofs=proofs, coms=coms, rng=rng, d_p=to_dev(proofs), d_c=to_dev(coms), d_r=to_dev(rng),
                          d_v=torch.full((nb,), 255, dtype=torch.uint8, device=dev), d_m=torch.full((nb, 32), 255, dtype=torch.uint8, device=dev)))
    torch.cuda.synchronize()
    pool.set_option("stat_reset", 1)
    for d in items:
        fx = d["fx"]
        pool.submit_dev(0, fx.n, fx.m, d["nb"], d["d_p"].data_ptr(), fx.proof_len, d["d_c"].data_ptr(), d["label"], d["d_r"].data_ptr(), d["d_v"].data_ptr(),
                        d["d_m"].data_ptr())
    pool.wait()
    chains, chain_proofs = pool.get_option("stat_chains"), pool.get_option("stat_chain_proofs")
    assert chain_proofs == 20 * 256 + 20 * 64
    # by proof count (one chain's worth), cut in two because it carries several chains' worth of work: 3200 + 1920 single proofs, one chain of
    # the 1280 aggregated ones
    assert chains == 3, chains
    n_ok = 0
    for i, d in enumerate(items):
        fx = d["fx"]
        _, ev, em = oracle.verify_batch(gens, d["proofs"], d["coms"], fx.m, fx.n, d["label"], d["rng"], threads=os.cpu_count() or 1)
        v, msm = bytes(d["d_v"].cpu().numpy()), bytes(d["d_m"].cpu().numpy().reshape(-1))
        assert v == ev, i
        for q in range(d["nb"]):
            if em[32 * q:32 * q + 32] != b"\xff" * 32:
                assert msm[32 * q:32 * q + 32] == em[32 * q:32 * q + 32], (i, q)
        assert ev[3] == 0 and ev[14] == 0 and ev[-1] == 1
        n_ok += sum(1 for x in ev if x == 0)
        if d["label"] != base:
            assert sum(1 for x in ev if x == 0) == 2            # only the planted proofs verify under a foreign label
    assert n_ok > 20 * 2 + 5 * 250
    pool.close()


def test_batch_combined_item_wider_than_a_chain_is_refused(oracle):
    """a batch-combined batch is ONE identity check = one chain; one that max_chain_proofs cannot hold used to be cut, every
    piece after the first writing its 33-byte result past the caller's buffer.  Now refused at submission; what fits still runs whole."""
    import torch
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    fx = wl.load_fixture("cfg2_n64_m1")
    dev = torch.device("cuda", 0)
    pool = bp.Pool((0,), 4, fixed_window_bits=14)
    pool.gens_create(64, 1)
    pool.set_option("max_chain_proofs", 64)
    nb = 300
    proofs, coms = wl.tile_batch(fx, nb)
    to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_p, d_c = to_dev(proofs), to_dev(coms)
    d_v = torch.full((nb,), 255, dtype=torch.uint8, device=dev)
    d_b = torch.full((6, 36), 255, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    with pytest.raises(bp.BpgpuError):
        pool.submit_rlc_dev(0, fx.n, fx.m, nb, d_p.data_ptr(), fx.proof_len, d_c.data_ptr(), fx.label, None, d_v.data_ptr(), d_b[0].data_ptr())
    # pieces of <= 64 proofs: each its own combination, each its own 33 bytes, nothing written outside them
    off = 0
    for i in range(5):
        cnt = min(64, nb - off)
        pool.submit_rlc_dev(0, fx.n, fx.m, cnt, d_p.data_ptr() + off * fx.proof_len, fx.proof_len, d_c.data_ptr() + off * 32, fx.label, None, d_v.data_ptr() + off,
                            d_b[i].data_ptr())
        off += cnt
    pool.wait()
    assert off == nb and bytes(d_v.cpu().numpy()) == bytes(nb)
    b = d_b.cpu().numpy()
    for i in range(5):
        assert bytes(b[i][:33]) == bytes(33) and bytes(b[i][33:]) == b"\xff" * 3
    assert bytes(b[5]) == b"\xff" * 36
    pool.close()


def test_rlc_isolate_keeps_a_bad_proof_s_damage_in_its_own_batch(oracle):
    """Batch-combined batches of unrelated submitters share one identity check per chain by default: one bad proof leaves all of them
    undecided.  With the pool option rlc_isolate every batch is its own combination: only the batch with the bad proof comes back undecided."""
    import torch
    import bulletproofs_amd as bp
    from bulletproofs_amd import workload as wl
    fx = wl.load_fixture("cfg2_n64_m1")
    dev = torch.device("cuda", 0)
    to_dev = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    sizes = [200, 300, 150]
    total = sum(sizes)
    proofs, coms = wl.tile_batch(fx, total, first=3)
    pb = bytearray(proofs)
    pb[(200 + 17) * fx.proof_len + 129] ^= 4            # a tampered scalar in the second batch
    d_p, d_c = to_dev(bytes(pb)), to_dev(coms)
    for isolate, expect_und in ((0, [0, 1, 2]), (1, [1])):
        pool = bp.Pool((0,), 4, fixed_window_bits=14)
        pool.gens_create(64, 1)
        pool.set_option("auto_flush_items", 1000)
        pool.set_option("rlc_isolate", isolate)
        d_v = torch.full((total,), 255, dtype=torch.uint8, device=dev)
        d_b = torch.full((3, 36), 255, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        off = 0
        for i, nb in enumerate(sizes):
            pool.submit_rlc_dev(0, fx.n, fx.m, nb, d_p.data_ptr() + off * fx.proof_len, fx.proof_len, d_c.data_ptr() + off * 32, fx.label, None, d_v.data_ptr() + off,
                                d_b[i].data_ptr())
            off += nb
        pool.wait()
        assert pool.get_option("stat_chains") == (3 if isolate else 1)
        v, b = bytes(d_v.cpu().numpy()), d_b.cpu().numpy()
        off = 0
        for i, nb in enumerate(sizes):
            if i in expect_und:
                assert v[off:off + nb] == b"\x05" * nb and b[i][0] != 0, (isolate, i)
            else:
                assert v[off:off + nb] == bytes(nb) and bytes(b[i][:33]) == bytes(33), (isolate, i)
            off += nb
        pool.close()
