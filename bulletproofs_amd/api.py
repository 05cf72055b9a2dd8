"""Host-side mirror (Python) of the reference crate's verification API over the C ABI -- same names,
argument meaning and error behaviour as src/lib.rs:34-45 exports for this path (RangeProof,
BulletproofGens, PedersenGens, ProofError, Transcript).  The C++ twin is include/bulletproofs.hpp.
No arithmetic happens here: parsing checks lengths / scalar canonicity, everything else is one call
into libbpgpu.so.  There is no CPU fallback."""
from ._lib import Context, transcript_new, transcript_append_message, transcript_challenge_bytes

_L = 2**252 + 27742317777372353535851937790883648493


class ProofError(Exception):
    """src/errors.rs:12-54"""
    code = None

    def __eq__(self, other):
        return isinstance(other, ProofError) and type(self) is type(other)

    def __hash__(self):
        return hash(type(self))


class VerificationError(ProofError):
    code = 1


class FormatError(ProofError):
    code = 2


class InvalidBitsize(ProofError):
    code = 3


class InvalidGeneratorsLength(ProofError):
    code = 4


_BY_CODE = {c.code: c for c in (VerificationError, FormatError, InvalidBitsize, InvalidGeneratorsLength)}


class PedersenGens:
    """src/generators.rs:30-53; PedersenGens::default() as held by a BulletproofGens on the device."""

    def __init__(self, B, B_blinding):
        self.B, self.B_blinding = B, B_blinding


class BulletproofGens:
    """BulletproofGens::new(gens_capacity, party_capacity) (src/generators.rs:157-166), derived on the GPU."""

    def __init__(self, gens_capacity, party_capacity, device=0, **ctx_options):
        self.gens_capacity, self.party_capacity = gens_capacity, party_capacity
        self.ctx = Context(device, **ctx_options)
        self.ctx.gens_create(gens_capacity, party_capacity)

    def pedersen(self):
        _, _, B, Bb = self.ctx.gens_export()
        return PedersenGens(B, Bb)

    def increase_capacity(self, new_capacity):
        """generators.rs:177-204: extends every party's chain to new_capacity (no-op if not larger); the device tables are rebuilt."""
        if self.gens_capacity >= new_capacity:
            return
        # the Pedersen bases the context holds now (custom ones if the caller went through Context.gens_load) must survive the rebuild
        _, _, B0, Bb0 = self.ctx.gens_export()
        self.ctx.gens_create(new_capacity, self.party_capacity)
        G1, H1, B1, Bb1 = self.ctx.gens_export()
        if (B0, Bb0) != (B1, Bb1):          # custom bases were loaded: keep them, with the extended G / H chains
            self.ctx.gens_load(new_capacity, self.party_capacity, G1, H1, B0, Bb0)
        self.gens_capacity = new_capacity
        self.__dict__.pop("_pc", None)

    def _flat(self):
        G, H, _, _ = self.ctx.gens_export()
        return G, H

    def G(self, n, m):
        """the aggregated iterator of generators.rs:207-232: the first n generators of each of the first m parties, compressed"""
        G, _ = self._flat()
        c = self.gens_capacity
        return [G[32 * (j * c + i):32 * (j * c + i) + 32] for j in range(m) for i in range(n)]

    def H(self, n, m):
        _, H = self._flat()
        c = self.gens_capacity
        return [H[32 * (j * c + i):32 * (j * c + i) + 32] for j in range(m) for i in range(n)]

    def share(self, j):
        """BulletproofGens::share(j) (generators.rs:168-175): .G(n) / .H(n) of party j"""
        return BulletproofGensShare(self, j)


    def commit(self, value, blinding):
        """PedersenGens::commit(value, blinding) (generators.rs:38-42) with this context's bases: value B + blinding B_blinding,
        compressed.  value: int or 32-byte scalar; blinding: 32-byte scalar.  (Variable time on the GPU, unlike the reference's
        constant-time multiscalar_mul.)"""
        v = value.to_bytes(32, "little") if isinstance(value, int) else bytes(value)
        _, _, B, Bb = self.ctx.gens_export()
        out, st = self.ctx.msm_batch([2], v + bytes(blinding), B + Bb)
        if st != bytes(1):
            raise ValueError("scalars must be canonical")
        return out

    def _check_pedersen(self, pc_gens):
        """The verifier multiplies by pc_gens.B / B_blinding (mod.rs:439-440); the device tables hold the bases this
        BulletproofGens was created with.  A different PedersenGens must be loaded with Context.gens_load."""
        if pc_gens is None:
            return
        if not hasattr(self, "_pc"):
            self._pc = self.pedersen()
        if (bytes(pc_gens.B), bytes(pc_gens.B_blinding)) != (self._pc.B, self._pc.B_blinding):
            raise ValueError("pc_gens differs from the Pedersen bases held in the device tables (use Context.gens_load for custom bases)")


class BulletproofGensShare:
    """generators.rs:262-292"""

    def __init__(self, gens, share):
        self.gens, self.share_index = gens, share

    def G(self, n):
        G, _ = self.gens._flat()
        o = self.share_index * self.gens.gens_capacity
        return [G[32 * (o + i):32 * (o + i) + 32] for i in range(n)]

    def H(self, n):
        _, H = self.gens._flat()
        o = self.share_index * self.gens.gens_capacity
        return [H[32 * (o + i):32 * (o + i) + 32] for i in range(n)]


class Transcript:
    """merlin::Transcript: held as its 208-byte STROBE state (include/bpgpu.h BPGPU_TRANSCRIPT_BYTES), so a transcript
    that already absorbed application messages can be handed to the verifier, which leaves it advanced exactly as
    verify_multiple_with_rng(&mut transcript, ...) does (mod.rs:345-353)."""

    def __init__(self, label, _state=None):
        self.state = _state if _state is not None else transcript_new(bytes(label))
        self.fresh_label = bytes(label) if _state is None else None   # set while the transcript is exactly Transcript::new(label)

    def clone(self):
        t = Transcript(None, self.state)
        t.fresh_label = self.fresh_label
        return t

    def append_message(self, label, message):
        self.state = transcript_append_message(self.state, bytes(label), bytes(message))
        self.fresh_label = None

    def append_u64(self, label, x):
        self.append_message(label, int(x).to_bytes(8, "little"))

    def challenge_bytes(self, label, n):
        self.state, out = transcript_challenge_bytes(self.state, bytes(label), n)
        self.fresh_label = None
        return out


class RangeProof:
    def __init__(self, raw):
        self._raw = raw

    @staticmethod
    def from_bytes(b):
        """src/range_proof/mod.rs:504-538 + src/inner_product_proof.rs:373-407; raises FormatError."""
        b = bytes(b)
        if len(b) % 32 != 0 or len(b) < 7 * 32:
            raise FormatError()
        ne = (len(b) - 7 * 32) // 32
        if ne < 2 or (ne - 2) % 2 != 0 or (ne - 2) // 2 >= 32:
            raise FormatError()
        for off in (128, 160, 192, len(b) - 64, len(b) - 32):
            if int.from_bytes(b[off:off + 32], "little") >= _L:
                raise FormatError()
        return RangeProof(b)

    def to_bytes(self):
        return self._raw

    @staticmethod
    def prove_multiple_with_rng(bp_gens, pc_gens, transcript, values, blindings, n, rng_bytes=None):
        """RangeProof::prove_multiple_with_rng (mod.rs:234-288) on the GPU (variable time: see include/bpgpu.h).  values: ints,
        blindings: 32-byte scalars; rng_bytes: the bytes the rng would hand Scalar::random, in the reference's draw order
        (None = OS CSPRNG).  Returns (RangeProof, [commitment bytes]); `transcript` is left advanced."""
        bp_gens._check_pedersen(pc_gens)
        m = len(values)
        proofs, coms, ts = bp_gens.ctx.rangeproof_prove_batch(n, m, list(values), b"".join(blindings), transcript=transcript.state, rng=rng_bytes,
                                                            want_transcripts=True)
        transcript.state = ts
        transcript.fresh_label = None
        return RangeProof(proofs), [coms[32 * j:32 * j + 32] for j in range(m)]

    @staticmethod
    def prove_single_with_rng(bp_gens, pc_gens, transcript, v, v_blinding, n, rng_bytes=None):
        proof, coms = RangeProof.prove_multiple_with_rng(bp_gens, pc_gens, transcript, [v], [v_blinding], n, rng_bytes)
        return proof, coms[0]

    @staticmethod
    def prove_multiple(bp_gens, pc_gens, transcript, values, blindings, n):
        """RangeProof::prove_multiple (mod.rs:290-311): prove_multiple_with_rng with thread_rng() -- here the OS CSPRNG"""
        return RangeProof.prove_multiple_with_rng(bp_gens, pc_gens, transcript, values, blindings, n, None)

    @staticmethod
    def prove_single(bp_gens, pc_gens, transcript, v, v_blinding, n):
        """RangeProof::prove_single (mod.rs:141-158)"""
        return RangeProof.prove_single_with_rng(bp_gens, pc_gens, transcript, v, v_blinding, n, None)

    def verify_multiple_with_rng(self, bp_gens, pc_gens, transcript, value_commitments, n, rng64):
        """Ok(()) -> returns None; Err(e) -> raises e.  rng64 = the 64 bytes Scalar::random(rng) would draw (mod.rs:396);
        None = thread_rng()."""
        m = len(value_commitments)
        bp_gens._check_pedersen(pc_gens)
        v, ts = bp_gens.ctx.rangeproof_verify_batch_ts(n, m, self._raw, len(self._raw), b"".join(value_commitments), transcript.state, rng64,
                                                       want_transcripts=True)
        transcript.state = ts   # &mut Transcript: left advanced
        transcript.fresh_label = None
        if v[0] != 0:
            raise _BY_CODE[v[0]]()

    def verify_multiple(self, bp_gens, pc_gens, transcript, value_commitments, n):
        return self.verify_multiple_with_rng(bp_gens, pc_gens, transcript, value_commitments, n, None)

    def verify_single_with_rng(self, bp_gens, pc_gens, transcript, V, n, rng64):
        return self.verify_multiple_with_rng(bp_gens, pc_gens, transcript, [V], n, rng64)

    def verify_single(self, bp_gens, pc_gens, transcript, V, n):
        return self.verify_multiple_with_rng(bp_gens, pc_gens, transcript, [V], n, None)

    @staticmethod
    def verify_batch(bp_gens, pc_gens, transcript, proofs, commitments, n, rng64=None):
        """proofs[i].verify_multiple(bp_gens, pc_gens, &mut transcript.clone(), &commitments[i], n) for all i in one GPU pass
        -> list of None / ProofError.  `transcript` itself is not advanced."""
        if not proofs:
            return []
        bp_gens._check_pedersen(pc_gens)
        raw = [p.to_bytes() if isinstance(p, RangeProof) else bytes(p) for p in proofs]
        m, ln = len(commitments[0]), len(raw[0])
        assert all(len(r) == ln for r in raw) and all(len(c) == m for c in commitments)
        v = bp_gens.ctx.rangeproof_verify_batch_ts(n, m, b"".join(raw), ln, b"".join(b"".join(c) for c in commitments), transcript.state, rng64)
        if len(raw) == 1:
            v = bytes(v)
        return [None if x == 0 else _BY_CODE[x]() for x in v]

    @staticmethod
    def verify_batch_combined(bp_gens, pc_gens, transcript, proofs, commitments, n, rng64=None, weights64=None):
        """The same verdicts through the batch-combined check (bpgpu_rangeproof_verify_rlc; no counterpart in the crate): one
        identity test per batch when every proof verifies, per-proof re-verification inside the call when not.  This entry
        point replays each proof's transcript from its label: `transcript` must be a fresh Transcript(label)."""
        if not proofs:
            return []
        bp_gens._check_pedersen(pc_gens)
        label = transcript.fresh_label
        if label is None:
            raise ValueError("verify_batch_combined needs a fresh Transcript(label); use verify_batch for pre-bound transcripts")
        raw = [p.to_bytes() if isinstance(p, RangeProof) else bytes(p) for p in proofs]
        m, ln = len(commitments[0]), len(raw[0])
        assert all(len(r) == ln for r in raw) and all(len(c) == m for c in commitments)
        v, _, _ = bp_gens.ctx.rangeproof_verify_rlc(n, m, b"".join(raw), ln, b"".join(b"".join(c) for c in commitments), label, rng64,
                                                    weights64)
        return [None if x == 0 else _BY_CODE[x]() for x in v]


class LinearProof:
    """src/linear_proof.rs (`pub use` lib.rs:36): the proof that <a, b> = c for a committed secret vector a and a public
    vector b.  Verification runs on the device (bpgpu_linear_verify_batch)."""

    def __init__(self, raw):
        self._raw = raw

    @staticmethod
    def from_bytes(b):
        """linear_proof.rs:350-394; raises FormatError."""
        b = bytes(b)
        ne = len(b) // 32
        if len(b) % 32 != 0 or ne < 3 or (ne - 3) % 2 != 0 or (ne - 3) // 2 >= 32:
            raise FormatError()
        for off in (len(b) - 64, len(b) - 32):
            if int.from_bytes(b[off:off + 32], "little") >= _L:
                raise FormatError()
        return LinearProof(b)

    def to_bytes(self):
        return self._raw

    def serialized_size(self):
        return len(self._raw)

    @staticmethod
    def create(transcript, rng_bytes, C, r, a_vec, b_vec, G_vec, F, B, ctx):
        """LinearProof::create(transcript, rng, &C, r, a_vec, b_vec, G_vec, &F, &B) (linear_proof.rs:40-173) on the GPU (variable
        time, like the reference's).  rng_bytes: the 64 * (2 lg n + 2) bytes the rng would yield to Scalar::random, or None for
        the OS CSPRNG.  The transcript is left advanced.  Raises ValueError for InvalidInputLength / InvalidGeneratorsLength."""
        c = getattr(ctx, "ctx", ctx)
        n = len(b_vec)
        if len(G_vec) != n:
            raise InvalidGeneratorsLength()
        if len(a_vec) != n or n == 0 or n & (n - 1):
            raise ValueError("InvalidInputLength")
        proofs, status, ts = c.linear_create_batch(n, bytes(C), bytes(r), b"".join(bytes(x) for x in a_vec), b"".join(bytes(x) for x in b_vec),
                                                   b"".join(bytes(g) for g in G_vec), bytes(F), bytes(B), transcript=transcript.state, rng=rng_bytes,
                                                   want_transcripts=True)
        if status[0] != 0:
            raise ValueError("an input point does not decode or a scalar is not canonical")
        transcript.state = ts
        transcript.fresh_label = None
        return LinearProof(proofs)

    def verify(self, transcript, C, G, F, B, b_vec, ctx):
        """LinearProof::verify(&self, transcript, C, G, F, B, b_vec) (linear_proof.rs:175-236): Ok(()) -> None, Err(e) -> raises e.
        C, F, B and the entries of G are 32-byte compressed points, b_vec 32-byte canonical scalars; ctx: the Context (or a
        BulletproofGens) whose device runs the check.  The transcript is left advanced as the reference leaves it when the proof
        is accepted; after an Err its state is unspecified (upstream has absorbed part of the public inputs at that point, this
        engine returns the state right after innerproduct_domain_sep(n) -- INTEGRATION.md 3c)."""
        c = getattr(ctx, "ctx", ctx)
        if len(G) != len(b_vec):
            raise InvalidGeneratorsLength()
        v, ts = c.linear_verify_batch(len(b_vec), self._raw, len(self._raw), bytes(C), b"".join(bytes(g) for g in G), bytes(F), bytes(B),
                                      b"".join(bytes(x) for x in b_vec), transcript=transcript.state, want_transcripts=True)
        transcript.state = ts
        transcript.fresh_label = None
        if v[0] != 0:
            raise _BY_CODE[v[0]]()

    @staticmethod
    def verify_batch(ctx, transcript, proofs, Cs, G, F, B, b_vecs):
        """proofs[i].verify(&mut transcript.clone(), &Cs[i], G, F, B, b_vecs[i]) for all i in one GPU pass -> list of None /
        ProofError.  b_vecs: one vector per proof, or a single vector (list of scalars) shared by all."""
        c = getattr(ctx, "ctx", ctx)
        if not proofs:
            return []
        raw = [p.to_bytes() if isinstance(p, LinearProof) else bytes(p) for p in proofs]
        ln, n = len(raw[0]), len(G)
        assert all(len(r) == ln for r in raw)
        shared = len(b_vecs) == n and isinstance(b_vecs[0], (bytes, bytearray))
        b = b"".join(bytes(x) for x in b_vecs) if shared else b"".join(b"".join(bytes(x) for x in v) for v in b_vecs)
        v = c.linear_verify_batch(n, b"".join(raw), ln, b"".join(bytes(x) for x in Cs), b"".join(bytes(g) for g in G), bytes(F), bytes(B), b,
                                  transcript=transcript.state)
        return [None if x == 0 else _BY_CODE[x]() for x in v]
