#!/usr/bin/env python3
"""Extracts the reference's known-answer vectors into a JSON fixture.

Source: /root/reference/tests/range_proof.rs:16-95 (`deserialize_and_verify`):
16 hex proofs created with bulletproofs v1.0.0, proofs[i][j] <-> n = 8<<i,
m = 1<<j, the 8 value commitments vc[0..8], the transcript label
b"Deserialize-And-Verify Test" and gens BulletproofGens::new(64, 8).
These are test DATA (hex strings), reproduced with attribution; no reference
code is copied.  Run in the authoring container only (the GPU box has no
/root/reference):  python tests/golden/extract_reference_vectors.py
"""
import json, re, os, sys

SRC = "/root/reference/tests/range_proof.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rangeproof_v1.json")

text = open(SRC).read()
body = text[text.index("fn deserialize_and_verify"):text.index("fn generate_test_vectors")]
proofs = re.findall(r'b"([0-9a-f]{400,})"', body)
vcs = re.findall(r'hex::decode\("([0-9a-f]{64})"\)', body)
label = re.search(r'Transcript::new\(b"([^"]+)"\)', body).group(1)
assert len(proofs) == 16 and len(vcs) == 8, (len(proofs), len(vcs))
cases = []
for i in range(4):
    for j in range(4):
        n, m = 8 << i, 1 << j
        p = proofs[4 * i + j]
        k = (n * m).bit_length() - 1
        assert len(p) == 2 * 32 * (9 + 2 * k), (n, m, len(p))
        cases.append({"n": n, "m": m, "proof": p})
json.dump({"source": "dalek-cryptography/bulletproofs tests/range_proof.rs:16-95 (v1.0.0 vectors)",
           "transcript_label": label, "gens_capacity": 64, "party_capacity": 8,
           "value_commitments": vcs, "cases": cases, "expect": "verify_multiple == Ok(())"},
          open(OUT, "w"), indent=1)
print("wrote", OUT, len(cases), "cases")
