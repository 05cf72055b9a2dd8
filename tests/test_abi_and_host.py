"""Host-side checks that need no GPU: the C-ABI library loads and exports every symbol declared in
include/bpgpu.h, refuses to run without a device (no CPU fallback), and the product package never
touches oracle/."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.build()
    import bulletproofs_amd as bp
    return bp


def test_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "bpgpu.h")).read()
    declared = sorted(set(re.findall(r"\b(bpgpu_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 18
    L = C.CDLL(built.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    from bulletproofs_amd import _lib
    assert sorted(_lib.EXPORTS) == declared
    assert built.lib().bpgpu_version() >= 100


def test_no_cpu_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(built.BpgpuError, match="NO_DEVICE"):
        built.Context(0)


def test_pool_refuses_without_device_and_with_too_few_hardware_queues(built):
    """bpgpu_pool_create: no CPU fallback either; and it fails loudly (BPGPU_ERR_HW_QUEUES) when GPU_MAX_HW_QUEUES was set to a
    value on which its lanes would serialise.  The library itself asks for 16 queues when the variable is unset."""
    code = ("import os, sys; sys.path.insert(0, %r); import ctypes as C; import bulletproofs_amd as bp; L = bp.lib();"
            "g = C.CDLL(None).getenv; g.restype = C.c_char_p; print((g(b'GPU_MAX_HW_QUEUES') or b'None').decode()); h = C.c_void_p(); d = (C.c_int * 1)(0);"
            "print(L.bpgpu_pool_create(d, 1, 32, C.byref(h))); print(L.bpgpu_pool_create(d, 1, 2, C.byref(h)) in (0, -4));"
            "print(L.bpgpu_pool_create(d, 0, 2, C.byref(h)))" % ROOT)
    env = dict(os.environ)
    env.pop("GPU_MAX_HW_QUEUES", None)
    out = subprocess.check_output([sys.executable, "-c", code], env=env).decode().split()
    assert out[0] == "16" and out[1] in ("0", "-4") and out[2] == "True" and out[3] == "-1"     # unset -> the library set 16
    env["GPU_MAX_HW_QUEUES"] = "4"
    out = subprocess.check_output([sys.executable, "-c", code], env=env).decode().split()
    assert out[0] == "4" and out[1] == "-6" and out[2] == "True"                                # 32 lanes on 4 queues: refused; 2 lanes: fine
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(built.BpgpuError, match="NO_DEVICE"):
            built.Pool((0,), 8)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "bulletproofs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in txt and "liboracle" not in txt and "bp_twin" not in txt, f
                assert not re.search(r'#include\s+"[^"]*oracle', txt), f
    out = subprocess.check_output(["ldd", os.path.join(pkg, "csrc", "libbpgpu.so")]).decode()
    assert "liboracle" not in out


def test_workload_fixtures_and_figures():
    sys.path.insert(0, ROOT)
    from bulletproofs_amd import workload as wl
    fx = wl.load_fixture("cfg2_n64_m1")
    assert (fx.n, fx.m, fx.count, fx.proof_len) == (64, 1, 8192, 672) and fx.label == b"AggregateRangeProofBenchmark"
    p, c = wl.tile_batch(fx, 1030, first=8188)
    assert len(p) == 1030 * 672 and p[4 * 672:5 * 672] == fx.proofs[:672] and len(c) == 1030 * 32
    assert len(set(fx.proofs[i * 672:(i + 1) * 672] for i in range(fx.count))) == fx.count   # all distinct
    assert wl.load_fixture("cfg3_n64_m16").count >= 256 and wl.load_fixture("cfg4_n64_m32").count >= 512
    # SURVEY.md Appendix B / section 8d figures
    assert [wl.msm_terms(*s) for s in ((32, 1), (64, 1), (64, 16), (64, 32))] == [81, 147, 2090, 4156]
    assert [wl.reference_point_ops(N) for N in (147, 2090, 4156, 6179)] == [7704, 77608, 145786, 212545]
    assert [wl.algorithmic_bytes_per_verification(*s) for s in ((64, 1), (64, 16), (64, 32))] == [5280, 68192, 134880]
    for total, world in ((4096, 8), (10, 4), (3, 8)):
        r = [wl.shard_range(total, world, k) for k in range(world)]
        assert r[0][0] == 0 and r[-1][1] == total and all(r[i][1] == r[i + 1][0] for i in range(world - 1))
        assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_fixture_proofs_verify_with_oracle(oracle):
    sys.path.insert(0, ROOT)
    from bulletproofs_amd import workload as wl
    for name in ("cfg1_n32_m1", "cfg3_n64_m16", "cfg4_n64_m32"):
        fx = wl.load_fixture(name)
        g = oracle.Gens(fx.n, fx.m)
        cnt = min(fx.count, 4)
        _, v, _ = oracle.verify_batch(g, fx.proofs[:cnt * fx.proof_len], fx.commitments[:cnt * 32 * fx.m], fx.m, fx.n, fx.label,
                                      bytes(range(64)) * cnt, threads=2)
        assert v == bytes(cnt)


def test_transcript_helpers_match_merlin_kat_and_oracle(oracle):
    """bpgpu_transcript_new / append_message / challenge_bytes run on the host (no GPU): the Merlin known-answer vector of
    SURVEY.md Appendix A, and state-for-state agreement with the oracle's Merlin across rate-boundary crossings."""
    sys.path.insert(0, ROOT)
    from bulletproofs_amd._lib import transcript_new, transcript_append_message, transcript_challenge_bytes, TRANSCRIPT_BYTES
    s = transcript_new(b"test protocol")
    assert len(s) == TRANSCRIPT_BYTES == 208 and s[203:] == bytes(5)
    s = transcript_append_message(s, b"some label", b"some data")
    s, ch = transcript_challenge_bytes(s, b"challenge", 32)
    assert ch.hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    a, b = transcript_new(b""), oracle.transcript_new(b"")
    assert a == b
    for i, n in enumerate((0, 1, 31, 165, 166, 167, 500)):
        msg = bytes((7 * i + j) & 0xff for j in range(n))
        a, b = transcript_append_message(a, b"lbl%d" % i, msg), oracle.transcript_append_message(b, b"lbl%d" % i, msg)
        assert a == b
        (a, ca), (b, cb) = transcript_challenge_bytes(a, b"ch", 64 + n), oracle.transcript_challenge_bytes(b, b"ch", 64 + n)
        assert a == b and ca == cb
    # malformed states are refused, not executed
    import pytest as _pt
    from bulletproofs_amd import BpgpuError
    bad = bytearray(a)
    bad[200] = 200
    with _pt.raises(BpgpuError):
        transcript_append_message(bytes(bad), b"x", b"y")


def test_chain_form_decision_table():
    """rp_chain_forms (csrc/bpgpu.hip): which forms a per-proof launch chain takes, as a function of the context options, the pool's
    hint and the chain's width -- plain host logic, checked here row by row (the measurements behind each row: DESIGN.md section 4)."""
    import ctypes as C
    import bulletproofs_amd as bp
    L = bp.lib()
    f = L.bpgpu_internal_chain_forms
    f.restype = C.c_uint32
    f.argtypes = [C.c_int64] * 5 + [C.c_uint64, C.c_int, C.c_int]
    WIDE, WAVE, ASIDE, R32, AOUT = 1, 2, 4, 8, 16

    def forms(nbatch, lanes=0, split3=-1, busy=-1, radix=0, a_out=1, other=0, s2=0):
        return f(lanes, split3, busy, radix, a_out, nbatch, other, s2)
    # a context nobody told anything: wavefront chains up to 256 proofs, quads to 8191 (wide from 2048), one-lane chains aside from 8192
    assert forms(1) == WAVE and forms(256) == WAVE and forms(257) == 0 and forms(2047) == 0
    assert forms(2048) == WIDE and forms(8191) == WIDE
    assert forms(8192) == WIDE | ASIDE | AOUT
    # the pool says other chains run beside this one: the throughput forms from 2048 proofs
    assert forms(2048, busy=1) == WIDE | ASIDE | AOUT and forms(1024, busy=1) == 0 and forms(200, busy=1) == WAVE
    # the pool says the chain is alone: wavefront chains up to 2304 proofs, quads above -- never the one-lane form
    assert forms(2048, busy=0) == WIDE | WAVE and forms(2304, busy=0) == WIDE | WAVE and forms(2305, busy=0) == WIDE
    assert forms(16384, busy=0) == WIDE
    # explicit options win; radix 32 and A outside exist only with the one-lane chain aside
    assert forms(3000, lanes=1) == WIDE | ASIDE | AOUT and forms(3000, lanes=1, a_out=0) == WIDE | ASIDE
    assert forms(3000, lanes=1, radix=32) == WIDE | ASIDE | R32 | AOUT and forms(3000, lanes=4, radix=32, busy=1) == WIDE
    assert forms(3000, lanes=64, busy=1) == WIDE | WAVE and forms(100, lanes=1) == 0 and forms(100, lanes=4) == 0
    assert forms(100000, split3=0) == 0 and forms(100000, split3=0, lanes=1) == 0
    # a forced split on a small busy batch must not combine the wavefront chain with the one-lane launch
    assert forms(100, split3=1, busy=1) == WIDE | WAVE
    # batch-combined calls / verdict-only shapes, and a caller on the context's second stream: never wide / aside
    assert forms(50000, other=1, busy=1) == 0 and forms(50000, s2=1, busy=1) == WIDE


def test_window_pair_policy_for_two_shapes_under_one_budget():
    """bpgpu_gens_add_shape's choice is plain host logic (pick_window_pair, bpgpu.hip): the pair of windows that minimises
    nwin(W1) / nwin(W1 alone) + nwin(W2) / nwin(W2 alone) among those whose two tables fit the budget together."""
    import ctypes as C
    import bulletproofs_amd as bp
    L = bp.lib()
    L.bpgpu_internal_window_pair.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.bpgpu_internal_window_pair.restype = None

    def pair(s1, s2, budget, w1_fixed=0):
        a, b = C.c_uint32(), C.c_uint32()
        L.bpgpu_internal_window_pair(2 + 2 * s1[0] * s1[1], 2 + 2 * s2[0] * s2[1], budget, w1_fixed, C.byref(a), C.byref(b))
        return a.value, b.value

    def tbytes(shape, W):
        return (2 + 2 * shape[0] * shape[1]) * -(-255 // W) * (1 << (W - 1)) * 128

    GiB = 1 << 30
    # the service of VERDICT r03 #8: m = 16 and m = 1 proofs under the default 160 GiB -- alone they take W = 16 (138 GB) and W = 20 (113 GB)
    assert pair((64, 16), (64, 1), 160 * GiB) == (15, 19)
    nw = lambda W: -(-255 // W)
    f = lambda w1, w2: nw(w1) / 16 + nw(w2) / 13                      # relative walk lengths against the alone-optima (16 and 13 windows)
    fits = [(a, b) for a in range(4, 21) for b in range(4, 21) if tbytes((64, 16), a) + tbytes((64, 1), b) <= 160 * GiB]
    assert (15, 19) in fits and all(f(15, 19) <= f(a, b) + 1e-12 for a, b in fits)
    assert pair((64, 16), (64, 1), 250 * GiB) == (16, 20)            # room for both alone-optima
    assert pair((64, 16), (64, 1), 160 * GiB, 16) == (16, 17)        # primary window pinned by the caller: the rest goes to the secondary
    for budget in (8 * GiB, 96 * GiB, 160 * GiB):
        for s1, s2 in (((64, 16), (64, 1)), ((64, 32), (64, 1)), ((64, 8), (32, 1))):
            w1, w2 = pair(s1, s2, budget)
            assert tbytes(s1, w1) + tbytes(s2, w2) <= budget


def test_hot_path_kernels_use_no_scratch_memory_and_the_build_gate_knows_every_exception():
    """Read from the shipped code objects (tools/kernel_resources.py): the kernels of a wide launch chain -- window sums, generator
    exponents, table walk, Horner chains, partial-sum reduction, finish -- have no private segment and no spilled registers; every kernel
    that does is on __graft_entry__.SCRATCH_ALLOW with ceilings it respects (build() fails otherwise); launch 1's scripted variant stays
    at or below the figures this round reached (round 3: 1152 B / 346 spills / 926 KB of code)."""
    import glob
    import __graft_entry__ as ge
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as kr
    objs = sorted(glob.glob(os.path.join(ROOT, "bulletproofs_amd", "csrc", "build", "*.o")))
    if not objs:
        ge.build()
        objs = sorted(glob.glob(os.path.join(ROOT, "bulletproofs_amd", "csrc", "build", "*.o")))
    assert kr.check(objs, ge.SCRATCH_ALLOW) == []
    ks = {k["name"]: k for o in objs for k in kr.kernels_of(o)}
    for hot in ("void k_vb_window_wide<false>", "void k_rp_exponents<true>", "void k_rp_exponents<false>", "void k_rp_stage3<0>", "void k_rp_stage3<1>", "void k_rp_stage3<2>",
                "void k_bk2_window<64>", "void k_bk2_window<256>", "k_bk2_leafv", "k_msm_tail", "k_msm_tail_fast", "void k_rp_stage4<4>", "void k_rp_horner_wide<false>", "k_fb_reduce",
                "void k_finish8<false>", "void k_finish1<false>", "k_vb_window_colc", "void k_rp_stage4<64>"):
        assert ks[hot]["scratch"] == 0 and ks[hot]["spill_vgpr"] == 0, hot
    s1 = ks["void k_rp_stage1<true>"]
    assert s1["scratch"] <= 360 and s1["spill_vgpr"] <= 118 and s1["code_bytes"] <= 320 * 1024
    assert ks["void k_rp_exponents<false>"]["vgpr"] <= 168         # three wavefronts per SIMD would fit (measured: two are better on bursts)
    assert ks["void k_rp_exponents<true>"]["vgpr"] <= 256          # the paired form: pinned at two wavefronts per SIMD, four running products live
    assert ks["void k_bk2_window<64>"]["vgpr"] <= 168 and ks["k_bk2_prepare"]["vgpr"] <= 168   # three wavefronts per SIMD: the fused bucket chain's wide launches


def test_flush_plan_how_pending_batches_are_cut_into_launch_chains():
    """plan_flush (csrc/pool.hip): the pool's decision how many launch chains a flush of T pending proofs becomes -- plain host logic,
    row by row (measurements: DESIGN 2a / 4b, profiles/r03/coalesce_sweep_after_horner_aside.txt, profiles/r04/ab_rlc_burst_chains.txt)."""
    import ctypes as C
    import bulletproofs_amd as bp
    L = bp.lib()
    f = L.bpgpu_internal_plan_flush
    f.restype = None
    f.argtypes = [C.c_uint64] * 5 + [C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]

    def plan(T, rlc=0, one=0, coalesce=5120, pair=24576, mx=1 << 20, lanes=64):
        ch, per, hint = C.c_uint64(), C.c_uint64(), C.c_uint32()
        f(T, coalesce, pair, mx, lanes, rlc, one, C.byref(ch), C.byref(per), C.byref(hint))
        return ch.value, per.value, hint.value
    assert plan(1024)[:2] == (1, 1024) and plan(1)[:2] == (1, 1)
    assert plan(8 * 1024)[:2] == (2, 4096)                       # 8 x 1024: two of 4096
    assert plan(20 * 1024)[:2] == (2, 10240)                     # the driver's form: a burst that fits two chains takes two
    assert plan(24576)[:2] == (2, 12288) and plan(24577)[0] == 5 # ... up to pair_limit_proofs
    assert plan(40 * 1024)[:2] == (8, 5120)                      # 40 x 1024: eight of 5120
    assert plan(20 * 1024, rlc=1)[:2] == (4, 5120)               # batch-combined bursts keep the chains of coalesce_proofs
    assert plan(8 * 1024, rlc=1)[:2] == (2, 4096)
    assert plan(20 * 1024, one=1)[:2] == (1, 20480)              # a chain's worth goes out as it is
    assert plan(10**6, lanes=8)[0] == 8 and plan(10**6, lanes=8, mx=65536) == (16, 65536, 16)   # one chain per lane; chains never exceed max_chain_proofs
    assert plan(20 * 1024, pair=0)[:2] == (4, 5120) and plan(20 * 1024, coalesce=2048)[:2] == (2, 10240)
    for T in (1, 63, 64, 1024, 5120, 20480, 10**5, 10**6):       # every proof is in exactly one chain; the hint stays in [16, 64], multiples of 8
        for rlc in (0, 1):
            ch, per, hint = plan(T, rlc)
            assert (ch - 1) * per < T <= ch * per and 16 <= hint <= 64 and hint % 8 == 0
    assert plan(20 * 1024)[2] == 56 and plan(40 * 1024)[2] == 32 and plan(10**6)[2] == 16
    # ---- table-walk work in (64,1)-proof equivalents (what split_lone_heavy weighs; planning chains in PROPORTION to it was an option until
    # round 6: measured worse on BASELINE configs 3 / 4, profiles/r05/plan_by_work_ab.txt) ----
    we, go = L.bpgpu_internal_work_equiv, L.bpgpu_internal_flush_group_order
    we.restype = C.c_uint64
    we.argtypes = [C.c_uint64] * 3
    go.restype = None
    go.argtypes = [C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint64)]
    assert we(1024, 64, 1) == 1024 and we(20 * 1024, 64, 1) == 20480                 # single 64-bit proofs: nothing changes
    assert we(256, 64, 16) == 4037 and we(512, 64, 32) == 16140                      # an m = 16 proof walks 2050 generator terms, a single one 130
    # ---- by proof count, but a LONE chain with two chains' worth of work is cut in two ----
    sl = L.bpgpu_internal_split_lone_heavy
    sl.restype = None
    sl.argtypes = [C.c_uint64] * 6 + [C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]

    def split(T, Tw, rlc=0, one=0, lanes=64):
        ch, per, _ = plan(T, rlc, one, lanes=lanes)
        co, po = C.c_uint64(), C.c_uint64()
        sl(ch, per, T, Tw, 5120, lanes, rlc, one, C.byref(co), C.byref(po))
        return co.value, po.value
    assert split(20 * 256, 20 * we(256, 64, 16)) == (2, 2560)          # config 3 as the driver runs it: two chains of 2560 instead of one of 5120
    assert split(20 * 512, 20 * we(512, 64, 32)) == (2, 5120)          # config 4: already two chains by proof count
    assert split(4096, 4096) == (1, 4096) and split(5120, 5120) == (1, 5120)   # single 64-bit proofs: one chain's worth stays one chain
    assert split(1024, we(1024, 64, 16)) == (2, 512)                   # a small aggregated burst with two chains' worth of work
    assert split(512, we(512, 64, 4)) == (1, 512) and split(100, 10**6) == (1, 100)   # ... not below two chains' worth, never chains of a few proofs
    assert split(5120, 10**6, rlc=1) == (1, 5120) and split(5120, 10**6, one=1) == (1, 5120) and split(5120, 10**6, lanes=1) == (1, 5120)
    # ---- grouping: alternating submissions of two shapes (and of several labels of one length) become runs, order kept inside a run ----

    def order(keys):
        a = (C.c_uint64 * len(keys))(*keys)
        o = (C.c_uint64 * len(keys))()
        go(a, len(keys), o)
        return list(o)
    assert order([0, 1, 0, 1, 0, 1]) == [0, 2, 4, 1, 3, 5]
    assert order([7, 7, 7]) == [0, 1, 2] and order([]) == [] and order([3, 2, 1]) == [0, 1, 2]
    assert order([5, 9, 9, 5, 2, 9]) == [0, 3, 1, 2, 5, 4]


def test_integration_md_binds_every_entry_point_of_the_header():
    """INTEGRATION.md section 1 is the reference-side binding a maintainer would add: every prototype of include/bpgpu.h has its Rust
    `extern "C"` declaration there (tools/gen_rust_extern.py derives them from the header)."""
    import re
    import subprocess
    import sys
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "gen_rust_extern.py"), "--missing"], text=True)
    assert out.strip() == "", "INTEGRATION.md lacks:\n" + out
    hdr = open(os.path.join(ROOT, "include", "bpgpu.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    declared = set(re.findall(r"fn (bpgpu_[a-z0-9_]+)", doc))
    assert len(declared) >= 66 and all(re.search(r"\b%s\s*\(" % n, hdr) for n in declared)   # and nothing there that the header does not have


def _pool_option_table():
    """the documented pool options of include/bpgpu.h: [(key, default)]"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "include", "bpgpu.h")).read()
    sec = txt[txt.index(" * Options (bpgpu_pool_set_option;"):txt.index(" * Any other key is forwarded to every lane context")]
    return re.findall(r'^ \*   "([a-z_]+)"\s+(\S+)', sec, re.M)


def test_every_pool_option_is_documented_and_settable():
    """the pool's option surface is what include/bpgpu.h lists, one line per key, at most 20 keys; every key the
    scheduler's bpgpu_pool_set_option handles is in that table and vice versa (the option branches round 5's own A/B refuted are gone)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "bulletproofs_amd", "csrc", "pool.hip")).read()
    body = src[src.index("int bpgpu_pool_set_option(bpgpu_pool *p, const char *key, int64_t value) {"):src.index("int bpgpu_internal_pool_tune(")]
    handled = set(re.findall(r'!strcmp\(key, "([a-z_]+)"\)', body))
    documented = [k for k, _ in _pool_option_table()]
    assert len(documented) == len(set(documented)) <= 20, documented
    assert handled == set(documented), (sorted(handled - set(documented)), sorted(set(documented) - handled))
    for gone in ("plan_by_work", "plan_min_chain_proofs", "stagger_chains", "combine_policy", "combine_mapped_in", "split_stage1", "fork_early"):
        assert '"%s")' % gone not in src and '"%s")' % gone not in open(os.path.join(root, "bulletproofs_amd", "csrc", "bpgpu.hip")).read(), gone
