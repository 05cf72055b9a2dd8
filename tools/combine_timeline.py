#!/usr/bin/env python3
"""Summarise a timeline written by bpgpu_pool_trace_dump (option "combine_trace"): where a request of the combining queue spends its
time, stage by stage, and what the launch chains looked like.  Usage: combine_timeline.py trace.jsonl"""
import json
import sys


def pct(v, q):
    v = sorted(v)
    return v[int(q * (len(v) - 1))] if v else 0.0


def main(path):
    chains, reqs = {}, []
    for ln in open(path):
        d = json.loads(ln)
        if "chain" in d:
            c = d["chain"]
            chains[(c["dev"], c["buf"], c["epoch"])] = c
        else:
            reqs.append(d["req"])
    us = lambda a, b: (b - a) / 1e3
    rows = {k: [] for k in ("submit->reserved", "reserved->written", "written->sealed (waiting for the buffer to leave)", "sealed->issue begins", "issue (host enqueue)",
                            "issue end->completion seen (device + polling)", "completion seen->delivered", "delivered->woken", "TOTAL submit->delivered")}
    n_join = 0
    for r in reqs:
        c = chains.get((r["dev"], r["buf"], r["epoch"]))
        if not c or not r["t_delivered"]:
            continue
        n_join += 1
        rows["submit->reserved"].append(us(r["t_submit"], r["t_reserved"]))
        rows["reserved->written"].append(us(r["t_reserved"], r["t_written"]))
        rows["written->sealed (waiting for the buffer to leave)"].append(us(r["t_written"], max(c["t_seal"], r["t_written"])))
        rows["sealed->issue begins"].append(us(max(c["t_seal"], r["t_written"]), c["t_issue0"]))
        rows["issue (host enqueue)"].append(us(c["t_issue0"], c["t_issue1"]))
        rows["issue end->completion seen (device + polling)"].append(us(c["t_issue1"], c["t_done"]))
        rows["completion seen->delivered"].append(us(c["t_done"], r["t_delivered"]))
        if r["t_woken"]:
            rows["delivered->woken"].append(us(r["t_woken"], r["t_delivered"]))
        rows["TOTAL submit->delivered"].append(us(r["t_submit"], r["t_delivered"]))
    print("requests sampled: %d (joined with their chain: %d); chains recorded: %d" % (len(reqs), n_join, len(chains)))
    print("%-58s %9s %9s %9s   (microseconds)" % ("stage", "p50", "p90", "mean"))
    for k, v in rows.items():
        if v:
            print("%-58s %9.1f %9.1f %9.1f" % (k, pct(v, .5), pct(v, .9), sum(v) / len(v)))
    cs = list(chains.values())
    if cs:
        K = [c["K"] for c in cs]
        print("chains: width p10 / p50 / p90 / mean = %d / %d / %d / %.1f; in flight when sealed p50 = %d" % (pct(K, .1), pct(K, .5), pct(K, .9), sum(K) / len(K),
                                                                                                         pct([c["inflight_at_seal"] for c in cs], .5)))
        print("        open->sealed p50 %.1f us; sealed->issued p50 %.1f; on device (issue end->done seen) p50 %.1f; done->buffer free p50 %.1f" % (
            pct([us(c["t_open"], c["t_seal"]) for c in cs], .5), pct([us(c["t_seal"], c["t_issue1"]) for c in cs], .5),
            pct([us(c["t_issue1"], c["t_done"]) for c in cs], .5), pct([us(c["t_done"], c["t_free"]) for c in cs if c["t_free"]], .5)))
        dl = [us(c["t_deliv0"], c["t_deliv1"]) for c in cs if c["t_deliv1"]]
        if dl:
            print("        delivery thread: p50 %.1f us per chain (%.0f ns per ticket piece)" % (pct(dl, .5), 1e3 * sum(dl) / max(1, sum(c["n_async"] for c in cs if c["t_deliv1"]))))


if __name__ == "__main__":
    main(sys.argv[1])
