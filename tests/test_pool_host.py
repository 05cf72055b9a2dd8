"""The pool's scheduler on the CPU: bulletproofs_amd/csrc/pool.hip compiled with g++ against a fake HIP runtime and a fake back end
(tests/cpu_pool/), driven by plain threads -- once optimised (behaviour), once under ThreadSanitizer (data races, lock order).
Every result the queue hands back is recomputed from the request's own inputs (tests/cpu_pool/fake_model.h), so a piece delivered
to the wrong request, a chain that ran on half-written inputs or a buffer recycled too early all show up as mismatches."""
import json
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
DIR = os.path.join(HERE, "cpu_pool")


@pytest.fixture(scope="module")
def built():
    subprocess.check_call(["make", "-s", "-j2"], cwd=DIR, stdout=subprocess.DEVNULL)
    return os.path.join(DIR, "build")


def run(exe, args, env=None, timeout=300):
    e = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    e.update(env or {})
    p = subprocess.run([exe] + args, capture_output=True, text=True, timeout=timeout, env=e)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert line, "no result line:\n" + p.stderr[-2000:]
    return p, json.loads(line[-1])


@pytest.mark.parametrize("mode", [["threads", "24", "1", "0.5"], ["threads", "6", "40", "0.5"], ["tickets", "8", "64", "0.8"], ["mixed", "10", "2.0"], ["failures"],
                                  ["destroy", "12"], ["flush"]])
def test_scheduler_under_thread_sanitizer(built, mode):
    """blocking calls, tickets, every request kind at once with options changing underneath, chains that fail to issue, destroy
    with every caller inside, device-pointer flushes: results == the model's, and ThreadSanitizer reports nothing"""
    p, d = run(os.path.join(built, "pool_host_test_tsan"), mode)
    assert "WARNING: ThreadSanitizer" not in p.stderr, p.stderr[:6000]
    assert p.returncode == 0, p.stderr[-2000:]
    assert d["mismatches"] == 0 and d.get("errors", 0) == 0
    assert d["items"] > 0


def test_queue_machinery_keeps_up_with_a_ticket_load(built):
    """16 threads x 128 single-proof tickets against the modelled device (0.55 ms + 0.2 us per proof + 0.1 ms per chain beside it): the
    queue's own machinery -- slot reservation, sealing, issue, delivery -- must not be what bounds the rate, and the chains must not
    collapse into a convoy of narrow ones (round 4 on the GPU: 360 k/s at 108 proofs per chain, one thread issuing AND delivering)"""
    best = None
    for _ in range(3):   # (a shared 8-core container: take the best of three short runs)
        p, d = run(os.path.join(built, "pool_host_test"), ["tickets", "16", "128", "0.8"])
        assert p.returncode == 0 and d["mismatches"] == 0 and d["errors"] == 0
        if best is None or d["rate_per_s"] > best["rate_per_s"]:
            best = d
    assert best["rate_per_s"] > 300e3, best
    assert best["items_per_chain"] > 60, best
    assert best["svc_us_per_chain"]["deliver"] < 400, best


def test_one_blocking_caller_pays_only_the_chain(built):
    """one thread, one proof per blocking call: latency = the modelled chain (0.55 ms + the fake stream thread's ~50 us timer slack) + the
    queue's part (measured ~45 us: opening the buffer 14, quiet detection 17-30, issue 1, completion seen -> woken 8); bounded generously:
    a shared container, and timing asserted in a test suite must not flake"""
    best = 1e9
    for _ in range(3):
        p, d = run(os.path.join(built, "pool_host_test"), ["threads", "1", "1", "0.4"])
        assert p.returncode == 0 and d["mismatches"] == 0
        best = min(best, d["lat_ms"]["p50"])
    assert best < 1.0, best   # (0.65 stand-alone here; a lost doorbell would show as combine_max_age_us = 1.5 ms)


def test_wide_regime_policy_is_plain_host_logic(built):
    """policy_seal through the driver's option plumbing is covered above; here: the timeline dump is well-formed"""
    out = os.path.join(built, "trace_test.jsonl")
    p, d = run(os.path.join(built, "pool_host_test"), ["tickets", "4", "32", "0.3"], env={"BP_TRACE": out})
    assert p.returncode == 0
    chains = reqs = 0
    for ln in open(out):
        r = json.loads(ln)
        if "chain" in r:
            c = r["chain"]
            assert c["t_open"] <= c["t_seal"] <= c["t_issue0"] <= c["t_issue1"] <= c["t_done"] <= c["t_free"]
            chains += 1
        else:
            q = r["req"]
            assert q["t_submit"] <= q["t_reserved"] <= q["t_written"] <= q["t_delivered"]
            reqs += 1
    assert chains > 10 and reqs > 10


@pytest.mark.parametrize("mode", [["tickets", "8", "64", "0.6"], ["mixed", "10", "1.0"], ["threads", "48", "3", "0.6"]])
def test_a_buffer_opened_slowly_is_never_sealed_on_its_predecessors_word(built, mode):
    """Round 5's first build published a reopened buffer's state (`st = CB_OPEN`) BEFORE its new reservation word; in between, the service
    thread could read the previous incarnation's word -- sealed, with that chain's width in it -- and seal the new incarnation at the old
    width: callers went on reserving, `written` passed K, the buffer waited for `written == K` forever (3 of 23 runs of the first GPU
    sweep; caught by combine_rate's watchdog).  BP_TEST_DELAY=1:200 (host-test builds only) holds the opener between the two stores for
    200 us: with the old order these runs hang within a second, with the word published first they complete."""
    p, d = run(os.path.join(built, "pool_host_test"), mode, env={"BP_TEST_DELAY": "1:200"}, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    assert d["mismatches"] == 0 and d.get("errors", 0) == 0 and d["items"] > 0


@pytest.mark.parametrize("point", list(range(1, 13)))
def test_every_publication_point_of_the_queue_held_open(built, point):
    """BP_TEST_DELAY=point:150 holds a thread for 150 us at one of twelve publication points of the combining queue (a reopened buffer between
    its two stores, a freed buffer before the waiter check, a finished chain between `st = CB_DONE` and its futex word, the service thread
    between sealing and `st = CB_SEALED`, a caller between its reservation, its record, its inputs and its count, the filler before it seals,
    the delivery thread before its last access to a ticket, a caller about to wait for a buffer, the service thread around its sleeping flag):
    tickets and every kind of request at once complete with every result right, whichever window is the wide one."""
    for mode in (["tickets", "8", "64", "0.4"], ["mixed", "10", "0.6"]):
        p, d = run(os.path.join(built, "pool_host_test"), mode, env={"BP_TEST_DELAY": "%d:150" % point}, timeout=120)
        assert p.returncode == 0, (point, mode, p.stderr[-1500:])
        assert d["mismatches"] == 0 and d.get("errors", 0) == 0 and d["items"] > 0, (point, mode)


def test_when_a_staging_buffer_leaves_both_policies_row_by_row(built):
    """policy_seal (the two regimes) and policy_seal_cohort (cohorts), plain host logic of pool.hip, against twenty-six stated situations:
    quiet / deadline / chains in flight / the throughput regime's "half of what runs" / the last free buffer / the safety net; a lone
    caller that is back leaves at once, a group that is not all back is waited for, a fragment beside a wide chain waits for company."""
    p, d = run(os.path.join(built, "pool_host_test"), ["policy"])
    assert p.returncode == 0 and d["mismatches"] == 0 and d["items"] == 26, p.stderr[-2000:]
