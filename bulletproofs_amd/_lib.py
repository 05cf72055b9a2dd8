"""ctypes binding of libbpgpu.so (the C ABI declared in include/bpgpu.h).

There is no CPU fallback: if the HIP library is missing or no GPU is usable,
everything here raises.  Nothing under oracle/ is ever imported.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libbpgpu.so")

_lib = None

EXPORTS = [
    "bpgpu_version", "bpgpu_ctx_create", "bpgpu_ctx_destroy", "bpgpu_last_error", "bpgpu_ctx_set_option",
    "bpgpu_ctx_get_option",
    "bpgpu_synchronize", "bpgpu_gens_create", "bpgpu_gens_load", "bpgpu_gens_export",
    "bpgpu_msm_batch", "bpgpu_msm_batch_dev", "bpgpu_msm_batch_shared", "bpgpu_msm_batch_shared_dev",
    "bpgpu_rangeproof_verify_batch", "bpgpu_rangeproof_verify_batch_dev", "bpgpu_ipp_verify_batch",
    "bpgpu_rangeproof_verify_rlc", "bpgpu_rangeproof_verify_rlc_dev",
    "bpgpu_profile_enable", "bpgpu_profile_reset", "bpgpu_profile_report",
    "bpgpu_transcript_new", "bpgpu_transcript_append_message", "bpgpu_transcript_challenge_bytes",
    "bpgpu_rangeproof_verify_batch_ts", "bpgpu_rangeproof_verify_batch_ts_dev", "bpgpu_ipp_verify_batch_dev",
    "bpgpu_ipp_create_batch", "bpgpu_rangeproof_prove_batch", "bpgpu_rangeproof_verify_batch_submit", "bpgpu_ctx_collect",
    "bpgpu_linear_verify_batch", "bpgpu_linear_verify_batch_dev", "bpgpu_linear_create_batch",
    "bpgpu_rangeproof_audit_shares", "bpgpu_ipp_verification_scalars",
    "bpgpu_pool_create", "bpgpu_pool_destroy", "bpgpu_pool_last_error", "bpgpu_pool_set_option", "bpgpu_pool_get_option",
    "bpgpu_pool_devices", "bpgpu_pool_lanes", "bpgpu_pool_lane", "bpgpu_pool_gens_create", "bpgpu_pool_gens_load",
    "bpgpu_pool_rangeproof_verify", "bpgpu_pool_rangeproof_submit_dev", "bpgpu_pool_flush", "bpgpu_pool_wait",
    "bpgpu_pool_rangeproof_verify_ts", "bpgpu_pool_rangeproof_submit_ts", "bpgpu_pool_ticket_done", "bpgpu_pool_ticket_wait",
    "bpgpu_pool_rangeproof_submit_dev_ex", "bpgpu_pool_ticket_stream_wait", "bpgpu_pool_rangeproof_submit_rlc_dev",
    "bpgpu_gens_add_shape", "bpgpu_pool_gens_add_shape", "bpgpu_pool_gather_dev",
    "bpgpu_pool_msm_batch_shared", "bpgpu_pool_msm_batch_shared_submit", "bpgpu_pool_msm_batch", "bpgpu_pool_ipp_verify", "bpgpu_pool_trace_dump",
    "bpgpu_pool_msm_batch_shared_submit_dev",
]

TRANSCRIPT_BYTES = 208


class BpgpuError(RuntimeError):
    pass


def lib():
    """Load libbpgpu.so (built by __graft_entry__.build()); raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels bundle their own libamdhip64; if libbpgpu.so pulled in /opt/rocm's copy first, a
    # later `import torch` in the same process would bring up a second HIP runtime and find no GPU.  Loading
    # torch first makes both share one runtime (torch is plumbing here: device buffers, streams, RCCL).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise BpgpuError("libbpgpu.so not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` -- "
                         "there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, sz, u8p, i = C.c_void_p, C.c_size_t, C.c_char_p, C.c_int
    L.bpgpu_version.restype = i
    L.bpgpu_ctx_create.argtypes = [i, C.POINTER(vp)]
    L.bpgpu_ctx_destroy.argtypes = [vp]
    L.bpgpu_ctx_destroy.restype = None
    L.bpgpu_last_error.argtypes = [vp]
    L.bpgpu_last_error.restype = C.c_char_p
    L.bpgpu_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.bpgpu_ctx_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
    L.bpgpu_synchronize.argtypes = [vp]
    missing = [n for n in EXPORTS if not hasattr(L, n)]
    if missing:
        raise BpgpuError("libbpgpu.so lacks symbols declared in include/bpgpu.h: %s" % missing)
    L.bpgpu_gens_create.argtypes = [vp, sz, sz]
    L.bpgpu_gens_load.argtypes = [vp, sz, sz, u8p, u8p, u8p, u8p]
    L.bpgpu_gens_export.argtypes = [vp, u8p, u8p, u8p, u8p]
    L.bpgpu_msm_batch.argtypes = [vp, sz, C.POINTER(C.c_uint32), u8p, u8p, u8p, u8p]
    L.bpgpu_msm_batch_dev.argtypes = [vp, sz, C.POINTER(C.c_uint32), vp, vp, vp, vp, vp]
    L.bpgpu_msm_batch_shared.argtypes = [vp, sz, sz, sz, sz, u8p, u8p, u8p, u8p, u8p]
    L.bpgpu_msm_batch_shared_dev.argtypes = [vp, sz, sz, sz, sz, vp, vp, vp, vp, vp, vp]
    L.bpgpu_rangeproof_verify_batch.argtypes = [vp, sz, sz, sz, u8p, sz, u8p, u8p, sz, u8p, u8p, u8p]
    L.bpgpu_rangeproof_verify_batch_dev.argtypes = [vp, sz, sz, sz, vp, sz, vp, u8p, sz, vp, vp, vp, vp]
    L.bpgpu_rangeproof_verify_rlc.argtypes = [vp, sz, sz, sz, u8p, sz, u8p, u8p, sz, u8p, u8p, u8p, u8p]
    L.bpgpu_rangeproof_verify_rlc_dev.argtypes = [vp, sz, sz, sz, vp, sz, vp, u8p, sz, vp, vp, vp, vp, vp]
    L.bpgpu_ipp_verify_batch.argtypes = [vp, sz, sz, u8p, sz, u8p, sz, u8p, u8p, u8p, u8p, u8p, u8p, u8p, u8p]
    L.bpgpu_transcript_new.argtypes = [u8p, sz, u8p]
    L.bpgpu_transcript_append_message.argtypes = [u8p, u8p, sz, u8p, sz]
    L.bpgpu_transcript_challenge_bytes.argtypes = [u8p, u8p, sz, u8p, sz]
    L.bpgpu_rangeproof_verify_batch_ts.argtypes = [vp, sz, sz, sz, u8p, sz, u8p, u8p, sz, u8p, u8p, u8p, u8p]
    L.bpgpu_rangeproof_verify_batch_ts_dev.argtypes = [vp, sz, sz, sz, vp, sz, vp, u8p, vp, vp, vp, vp, vp, vp]
    L.bpgpu_ipp_verify_batch_dev.argtypes = [vp, sz, sz, vp, sz, u8p, sz, u8p, vp, vp, vp, vp, vp, vp, i, vp, vp, vp]
    L.bpgpu_ipp_create_batch.argtypes = [vp, sz, sz, u8p, sz, u8p, u8p, u8p, u8p, u8p, u8p, i, u8p, u8p, u8p, u8p]
    L.bpgpu_rangeproof_prove_batch.argtypes = [vp, sz, sz, sz, C.POINTER(C.c_uint64), u8p, u8p, sz, u8p, u8p, u8p, u8p, u8p]
    L.bpgpu_rangeproof_verify_batch_submit.argtypes = [vp, sz, sz, sz, u8p, sz, u8p, u8p, sz, u8p, u8p, u8p]
    L.bpgpu_ctx_collect.argtypes = [vp]
    L.bpgpu_linear_verify_batch.argtypes = [vp, sz, sz, u8p, sz, u8p, sz, u8p, u8p, u8p, u8p, u8p, u8p, i, u8p, u8p, u8p]
    L.bpgpu_rangeproof_audit_shares.argtypes = [vp, sz, sz, C.POINTER(C.c_uint32), u8p, u8p, u8p, u8p, i, u8p, u8p]
    L.bpgpu_linear_create_batch.argtypes = [vp, sz, sz, u8p, sz, u8p, u8p, u8p, u8p, u8p, u8p, i, u8p, u8p, u8p, u8p, u8p, u8p]
    L.bpgpu_linear_verify_batch_dev.argtypes = [vp, sz, sz, vp, sz, u8p, sz, u8p, vp, vp, vp, vp, vp, i, vp, vp, vp, vp]
    L.bpgpu_ipp_verification_scalars.argtypes = [vp, sz, sz, u8p, sz, u8p, sz, u8p, sz, u8p, u8p, u8p, u8p, u8p]
    L.bpgpu_pool_create.argtypes = [C.POINTER(C.c_int), i, i, C.POINTER(vp)]
    L.bpgpu_pool_destroy.argtypes = [vp]
    L.bpgpu_pool_destroy.restype = None
    L.bpgpu_pool_last_error.argtypes = [vp]
    L.bpgpu_pool_last_error.restype = C.c_char_p
    L.bpgpu_pool_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.bpgpu_pool_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
    L.bpgpu_pool_devices.argtypes = [vp]
    L.bpgpu_pool_lanes.argtypes = [vp]
    L.bpgpu_pool_lane.argtypes = [vp, i, i]
    L.bpgpu_pool_lane.restype = vp
    L.bpgpu_pool_gens_create.argtypes = [vp, sz, sz]
    L.bpgpu_gens_add_shape.argtypes = [vp, sz, sz]
    L.bpgpu_pool_gens_add_shape.argtypes = [vp, sz, sz]
    L.bpgpu_pool_gens_load.argtypes = [vp, sz, sz, u8p, u8p, u8p, u8p]
    L.bpgpu_pool_rangeproof_verify.argtypes = [vp, sz, sz, sz, u8p, sz, u8p, u8p, sz, u8p, u8p, u8p]
    L.bpgpu_pool_rangeproof_submit_dev.argtypes = [vp, i, sz, sz, sz, vp, sz, vp, u8p, sz, vp, vp, vp]
    L.bpgpu_pool_rangeproof_verify_ts.argtypes = [vp, sz, sz, sz, u8p, sz, u8p, u8p, sz, u8p, u8p, u8p, u8p]
    L.bpgpu_pool_rangeproof_submit_ts.argtypes = [vp, sz, sz, sz, u8p, sz, u8p, u8p, sz, u8p, u8p, u8p, u8p, C.POINTER(vp)]
    L.bpgpu_pool_rangeproof_submit_dev_ex.argtypes = [vp, i, sz, sz, sz, vp, sz, vp, u8p, sz, vp, vp, vp, vp, i, C.POINTER(vp)]
    L.bpgpu_pool_ticket_stream_wait.argtypes = [vp, vp, vp]
    L.bpgpu_pool_gather_dev.argtypes = [vp, i, C.POINTER(vp), C.POINTER(sz), vp, vp]
    L.bpgpu_pool_rangeproof_submit_rlc_dev.argtypes = [vp, i, sz, sz, sz, vp, sz, vp, u8p, sz, vp, vp, vp, vp, i, C.POINTER(vp)]
    L.bpgpu_pool_msm_batch_shared.argtypes = [vp, sz, sz, sz, sz, u8p, u8p, u8p, u8p, u8p]
    L.bpgpu_pool_msm_batch_shared_submit.argtypes = [vp, sz, sz, sz, sz, u8p, u8p, u8p, u8p, u8p, C.POINTER(vp)]
    L.bpgpu_pool_msm_batch.argtypes = [vp, sz, C.POINTER(C.c_uint32), u8p, u8p, u8p, u8p]
    L.bpgpu_pool_ipp_verify.argtypes = [vp, sz, sz, u8p, sz, u8p, sz, u8p, u8p, u8p, u8p, u8p, u8p, u8p, u8p]
    L.bpgpu_pool_msm_batch_shared_submit_dev.argtypes = [vp, i, sz, sz, sz, sz, vp, vp, vp, vp, vp, vp, i, C.POINTER(vp)]
    L.bpgpu_pool_trace_dump.argtypes = [vp, C.c_char_p]
    L.bpgpu_pool_ticket_done.argtypes = [vp, vp]
    L.bpgpu_pool_ticket_wait.argtypes = [vp, vp]
    L.bpgpu_pool_flush.argtypes = [vp]
    L.bpgpu_pool_wait.argtypes = [vp]
    L.bpgpu_profile_enable.argtypes = [vp, i]
    L.bpgpu_profile_reset.argtypes = [vp]
    L.bpgpu_profile_report.argtypes = [vp, C.c_char_p, sz]
    _lib = L
    return L


def transcript_new(label):
    """merlin::Transcript::new(label) as its 208-byte state (host-side helper of the library; needs no GPU)."""
    st = C.create_string_buffer(TRANSCRIPT_BYTES)
    if lib().bpgpu_transcript_new(label, len(label), st) != 0:
        raise BpgpuError("bpgpu_transcript_new failed")
    return st.raw


def transcript_append_message(state, label, msg):
    st = C.create_string_buffer(bytes(state), TRANSCRIPT_BYTES)
    if lib().bpgpu_transcript_append_message(st, label, len(label), msg, len(msg)) != 0:
        raise BpgpuError("bpgpu_transcript_append_message failed (malformed state?)")
    return st.raw


def transcript_challenge_bytes(state, label, n):
    st = C.create_string_buffer(bytes(state), TRANSCRIPT_BYTES)
    out = C.create_string_buffer(max(n, 1))
    if lib().bpgpu_transcript_challenge_bytes(st, label, len(label), out, n) != 0:
        raise BpgpuError("bpgpu_transcript_challenge_bytes failed (malformed state?)")
    return st.raw, out.raw[:n]


ERR_NAMES = {0: "OK", -1: "INVALID_ARG", -2: "HIP", -3: "NO_GENS", -4: "NO_DEVICE", -5: "BAD_GENERATOR", -6: "HW_QUEUES"}


class Context:
    """Owns one bpgpu_ctx (one GPU)."""

    def __init__(self, device=0, fixed_window_bits=None, fixed_splits=None, fixed_table_max_bytes=None, horner_lanes=None):
        self._L = lib()
        h = C.c_void_p()
        rc = self._L.bpgpu_ctx_create(device, C.byref(h))
        if rc != 0:
            raise BpgpuError("bpgpu_ctx_create(device=%d) failed: %s (no CPU fallback)" % (device, ERR_NAMES.get(rc, rc)))
        self.h = h
        if fixed_window_bits is not None:
            self.set_option("fixed_window_bits", fixed_window_bits)
        if fixed_splits is not None:
            self.set_option("fixed_splits", fixed_splits)
        if fixed_table_max_bytes is not None:
            self.set_option("fixed_table_max_bytes", fixed_table_max_bytes)
        if horner_lanes is not None:
            self.set_option("horner_lanes", horner_lanes)

    def close(self):
        if getattr(self, "h", None):
            self._L.bpgpu_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise BpgpuError("%s: %s" % (ERR_NAMES.get(rc, rc), self._L.bpgpu_last_error(self.h).decode()))

    def set_option(self, key, value):
        self._chk(self._L.bpgpu_ctx_set_option(self.h, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_int64(0)
        self._chk(self._L.bpgpu_ctx_get_option(self.h, key.encode(), C.byref(v)))
        return v.value

    def synchronize(self):
        self._chk(self._L.bpgpu_synchronize(self.h))

    # ---- generators ----
    def gens_create(self, gens_capacity, party_capacity):
        self._chk(self._L.bpgpu_gens_create(self.h, gens_capacity, party_capacity))
        self.gens_capacity, self.party_capacity = gens_capacity, party_capacity

    def gens_add_shape(self, n2, m2):
        """a second, larger-window table for the smaller shape (n2, m2) (bpgpu_gens_add_shape)"""
        self._chk(self._L.bpgpu_gens_add_shape(self.h, n2, m2))

    def gens_load(self, gens_capacity, party_capacity, G, H, B, B_blinding):
        assert len(G) == len(H) == 32 * gens_capacity * party_capacity
        self._chk(self._L.bpgpu_gens_load(self.h, gens_capacity, party_capacity, G, H, B, B_blinding))
        self.gens_capacity, self.party_capacity = gens_capacity, party_capacity

    def gens_export(self):
        tot = self.gens_capacity * self.party_capacity
        G, H = C.create_string_buffer(32 * tot), C.create_string_buffer(32 * tot)
        B, Bb = C.create_string_buffer(32), C.create_string_buffer(32)
        self._chk(self._L.bpgpu_gens_export(self.h, G, H, B, Bb))
        return G.raw, H.raw, B.raw, Bb.raw

    # ---- MSM ----
    def msm_batch(self, n_terms, scalars, points):
        nb = len(n_terms)
        tot = sum(n_terms)
        assert len(scalars) == len(points) == 32 * tot
        nt = (C.c_uint32 * max(nb, 1))(*n_terms)
        out, st = C.create_string_buffer(32 * max(nb, 1)), C.create_string_buffer(max(nb, 1))
        self._chk(self._L.bpgpu_msm_batch(self.h, nb, nt, scalars, points, out, st))
        return out.raw[:32 * nb], st.raw[:nb]

    def msm_batch_shared(self, n, m, nbatch, n_unique, gen_scalars, uniq_scalars, uniq_points):
        assert len(gen_scalars) == 32 * (2 * n * m + 2) * nbatch
        assert len(uniq_scalars) == len(uniq_points) == 32 * n_unique * nbatch
        out, st = C.create_string_buffer(32 * max(nbatch, 1)), C.create_string_buffer(max(nbatch, 1))
        self._chk(self._L.bpgpu_msm_batch_shared(self.h, n, m, nbatch, n_unique, gen_scalars, uniq_scalars, uniq_points, out, st))
        return out.raw[:32 * nbatch], st.raw[:nbatch]

    # ---- range proofs ----
    def rangeproof_verify_batch(self, n, m, proofs, proof_len, commitments, label, rng64=None, want_msm=False):
        nb = len(proofs) // proof_len if proof_len else 0
        assert len(proofs) == nb * proof_len and len(commitments) == 32 * m * nb
        assert rng64 is None or len(rng64) == 64 * nb
        verdict = C.create_string_buffer(max(nb, 1))
        msm = C.create_string_buffer(32 * max(nb, 1)) if want_msm else None
        self._chk(self._L.bpgpu_rangeproof_verify_batch(self.h, n, m, nb, proofs, proof_len, commitments, label, len(label),
                                                        rng64, verdict, msm))
        return (verdict.raw[:nb], msm.raw[:32 * nb]) if want_msm else verdict.raw[:nb]

    def rangeproof_verify_batch_submit(self, n, m, proofs, proof_len, commitments, label, rng64=None):
        """Asynchronous bpgpu_rangeproof_verify_batch: returns at once; collect() returns the verdict bytes."""
        nb = len(proofs) // proof_len if proof_len else 0
        assert len(proofs) == nb * proof_len and len(commitments) == 32 * m * nb
        self._pending = (C.create_string_buffer(max(nb, 1)), nb)
        self._chk(self._L.bpgpu_rangeproof_verify_batch_submit(self.h, n, m, nb, proofs, proof_len, commitments, label, len(label), rng64,
                                                               self._pending[0], None))

    def collect(self):
        self._chk(self._L.bpgpu_ctx_collect(self.h))
        buf, nb = self._pending
        self._pending = None
        return buf.raw[:nb]

    def rangeproof_verify_batch_ts(self, n, m, proofs, proof_len, commitments, transcripts, rng64=None, want_msm=False,
                                   want_transcripts=False):
        """bpgpu_rangeproof_verify_batch_ts: `transcripts` is ONE 208-byte state shared by the batch, or nbatch of them.
        Returns verdict bytes [, msm encodings] [, advanced states]."""
        nb = len(proofs) // proof_len if proof_len else 0
        assert len(proofs) == nb * proof_len and len(commitments) == 32 * m * nb
        assert len(transcripts) in (TRANSCRIPT_BYTES, TRANSCRIPT_BYTES * nb)
        stride = 0 if (len(transcripts) == TRANSCRIPT_BYTES and nb != 1) else TRANSCRIPT_BYTES
        verdict = C.create_string_buffer(max(nb, 1))
        msm = C.create_string_buffer(32 * max(nb, 1)) if want_msm else None
        tso = C.create_string_buffer(TRANSCRIPT_BYTES * max(nb, 1)) if want_transcripts else None
        self._chk(self._L.bpgpu_rangeproof_verify_batch_ts(self.h, n, m, nb, proofs, proof_len, commitments, transcripts, stride,
                                                           rng64, verdict, msm, tso))
        out = [verdict.raw[:nb]]
        if want_msm:
            out.append(msm.raw[:32 * nb])
        if want_transcripts:
            out.append(tso.raw[:TRANSCRIPT_BYTES * nb])
        return out[0] if len(out) == 1 else tuple(out)

    def rangeproof_verify_rlc(self, n, m, proofs, proof_len, commitments, label, rng64=None, weights64=None):
        """Batch-combined verification (include/bpgpu.h): returns (verdict bytes, batch_ok, 32-byte encoding of the
        combined point).  On a failing combination the verdicts come from the per-proof path (automatic fallback)."""
        nb = len(proofs) // proof_len if proof_len else 0
        assert len(proofs) == nb * proof_len and len(commitments) == 32 * m * nb
        assert rng64 is None or len(rng64) == 64 * nb
        assert weights64 is None or len(weights64) == 64 * nb
        verdict = C.create_string_buffer(max(nb, 1))
        bo = C.create_string_buffer(33)
        self._chk(self._L.bpgpu_rangeproof_verify_rlc(self.h, n, m, nb, proofs, proof_len, commitments, label, len(label),
                                                      rng64, weights64, verdict, bo))
        return verdict.raw[:nb], bo.raw[0] == 0, bo.raw[1:33]

    # ---- stand-alone inner-product proofs ----
    def ipp_verify_batch(self, n, proofs, proof_len, label, Gf, Hf, P, Q, G, H, want_msm=False):
        nb = len(proofs) // proof_len if proof_len else 0
        assert len(Gf) == len(Hf) == len(G) == len(H) == 32 * n * nb and len(P) == len(Q) == 32 * nb
        verdict = C.create_string_buffer(max(nb, 1))
        msm = C.create_string_buffer(32 * max(nb, 1)) if want_msm else None
        self._chk(self._L.bpgpu_ipp_verify_batch(self.h, n, nb, proofs, proof_len, label, len(label), Gf, Hf, P, Q, G, H, verdict, msm))
        return (verdict.raw[:nb], msm.raw[:32 * nb]) if want_msm else verdict.raw[:nb]

    def ipp_verification_scalars(self, n, proofs, proof_len, label=b"", transcripts=None, want_transcripts=False):
        """InnerProductProof::verification_scalars for len(proofs) / proof_len proofs (bpgpu_ipp_verification_scalars): returns
        (u_sq bytes, u_inv_sq bytes, s bytes, status bytes[, advanced transcripts])."""
        nb = len(proofs) // proof_len if proof_len else 0
        k = max(0, (proof_len // 32 - 2) // 2)
        stride = 0
        if transcripts is not None:
            assert len(transcripts) in (TRANSCRIPT_BYTES, TRANSCRIPT_BYTES * nb)
            stride = 0 if (len(transcripts) == TRANSCRIPT_BYTES and nb != 1) else TRANSCRIPT_BYTES
        us, ui = C.create_string_buffer(32 * max(k * nb, 1)), C.create_string_buffer(32 * max(k * nb, 1))
        s_, st = C.create_string_buffer(32 * max(n * nb, 1)), C.create_string_buffer(max(nb, 1))
        tso = C.create_string_buffer(TRANSCRIPT_BYTES * max(nb, 1)) if want_transcripts else None
        self._chk(self._L.bpgpu_ipp_verification_scalars(self.h, n, nb, proofs, proof_len, label, len(label), transcripts, stride, us, ui, s_, tso, st))
        out = (us.raw[:32 * k * nb], ui.raw[:32 * k * nb], s_.raw[:32 * n * nb], st.raw[:nb])
        return out + (tso.raw[:TRANSCRIPT_BYTES * nb],) if want_transcripts else out

    def rangeproof_audit_shares(self, n, party_index, shares, bit_commitments, poly_commitments, challenges, want_checks=False):
        """ProofShare::audit_share for len(party_index) shares (bpgpu_rangeproof_audit_shares); challenges: 96 bytes shared or per share."""
        ns = len(party_index)
        assert len(shares) == ns * 32 * (3 + 2 * n) and len(bit_commitments) == 96 * ns and len(poly_commitments) == 64 * ns
        assert len(challenges) in (96, 96 * ns)
        shared = 1 if (len(challenges) == 96 and ns != 1) else 0
        pi = (C.c_uint32 * max(ns, 1))(*party_index)
        verdict = C.create_string_buffer(max(ns, 1))
        chk = C.create_string_buffer(64 * max(ns, 1)) if want_checks else None
        self._chk(self._L.bpgpu_rangeproof_audit_shares(self.h, n, ns, pi, shares, bit_commitments, poly_commitments, challenges, shared, verdict, chk))
        return (verdict.raw[:ns], chk.raw[:64 * ns]) if want_checks else verdict.raw[:ns]

    def linear_verify_batch(self, n, proofs, proof_len, Cs, G, F, B, b, label=b"", transcript=None, want_msm=False, want_transcripts=False):
        """LinearProof::verify for len(Cs) / 32 proofs (bpgpu_linear_verify_batch).  G: n points shared by the batch; b: per-proof
        vectors (nb * n scalars) or one shared vector (n scalars)."""
        nb = len(Cs) // 32
        assert len(proofs) == nb * proof_len and len(b) in (32 * n, 32 * n * nb)
        assert (G is None and F is None and B is None) or (len(G) == 32 * n and len(F) == len(B) == 32)   # None: the context's generators
        shared = 1 if (len(b) == 32 * n and nb != 1) else 0
        verdict = C.create_string_buffer(max(nb, 1))
        msm = C.create_string_buffer(32 * max(nb, 1)) if want_msm else None
        tso = C.create_string_buffer(TRANSCRIPT_BYTES * max(nb, 1)) if want_transcripts else None
        self._chk(self._L.bpgpu_linear_verify_batch(self.h, n, nb, proofs, proof_len, label, len(label), transcript, Cs, G, F, B, b, shared,
                                                    verdict, msm, tso))
        out = (verdict.raw[:nb],) + ((msm.raw[:32 * nb],) if want_msm else ()) + ((tso.raw[:TRANSCRIPT_BYTES * nb],) if want_transcripts else ())
        return out if len(out) > 1 else out[0]

    def linear_create_batch(self, n, Cs, rs, a, b, G, F, B, label=b"", transcript=None, rng=None, want_transcripts=False):
        """LinearProof::create for len(Cs) / 32 proofs (bpgpu_linear_create_batch): returns (proofs bytes, status bytes[, transcripts])."""
        nb = len(Cs) // 32
        assert len(rs) == 32 * nb and len(a) == 32 * n * nb and len(b) in (32 * n, 32 * n * nb)
        assert (G is None and F is None and B is None) or (len(G) == 32 * n and len(F) == len(B) == 32)   # None: the context's generators
        shared = 1 if (len(b) == 32 * n and nb != 1) else 0
        lg = n.bit_length() - 1
        assert rng is None or len(rng) == 64 * (2 * lg + 2) * nb
        pl = 32 * (2 * lg + 3)
        out, st = C.create_string_buffer(pl * max(nb, 1)), C.create_string_buffer(max(nb, 1))
        tso = C.create_string_buffer(TRANSCRIPT_BYTES * max(nb, 1)) if want_transcripts else None
        self._chk(self._L.bpgpu_linear_create_batch(self.h, n, nb, label, len(label), transcript, rng, Cs, rs, a, b, shared, G, F, B, out, st, tso))
        res = (out.raw[:pl * nb], st.raw[:nb])
        return res + (tso.raw[:TRANSCRIPT_BYTES * nb],) if want_transcripts else res

    def ipp_create_batch(self, n, Q, Gf, Hf, G, H, a, b, label=b"", transcript=None):
        """InnerProductProof::create for nbatch proofs (bpgpu_ipp_create_batch).  G/H of n*32 bytes = bases shared by the batch.
        Returns (proofs bytes, status bytes)."""
        nb = len(Q) // 32
        assert len(a) == len(b) == len(Gf) == len(Hf) == 32 * n * nb and len(G) == len(H) and len(G) in (32 * n, 32 * n * nb)
        shared = 1 if (len(G) == 32 * n and nb != 1) else 0
        pl = 32 * (2 * (n.bit_length() - 1) + 2)
        out, st = C.create_string_buffer(pl * max(nb, 1)), C.create_string_buffer(max(nb, 1))
        self._chk(self._L.bpgpu_ipp_create_batch(self.h, n, nb, label, len(label), transcript, Q, Gf, Hf, G, H, shared, a, b, out, st))
        return out.raw[:pl * nb], st.raw[:nb]

    def rangeproof_prove_batch(self, n, m, values, blindings, label=b"", transcript=None, rng=None, want_transcripts=False):
        """RangeProof::prove_multiple_with_rng for len(values) / m proofs (bpgpu_rangeproof_prove_batch): returns
        (proofs bytes, commitments bytes[, final transcript states])."""
        nb = len(values) // m
        assert len(values) == nb * m and len(blindings) == 32 * nb * m
        per = 64 * (m * (2 * n + 2) + 2 * m)
        assert rng is None or len(rng) == per * nb
        pl = 32 * (9 + 2 * ((n * m).bit_length() - 1))
        va = (C.c_uint64 * max(len(values), 1))(*values)
        proofs, coms = C.create_string_buffer(pl * max(nb, 1)), C.create_string_buffer(32 * m * max(nb, 1))
        tso = C.create_string_buffer(TRANSCRIPT_BYTES * max(nb, 1)) if want_transcripts else None
        self._chk(self._L.bpgpu_rangeproof_prove_batch(self.h, n, m, nb, va, blindings, label, len(label), transcript, rng, proofs, coms, tso))
        out = (proofs.raw[:pl * nb], coms.raw[:32 * m * nb])
        return out + (tso.raw[:TRANSCRIPT_BYTES * nb],) if want_transcripts else out

    # ---- instrumentation ----
    def profile_enable(self, on=True):
        self._chk(self._L.bpgpu_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        self._chk(self._L.bpgpu_profile_reset(self.h))

    def profile_report(self):
        buf = C.create_string_buffer(1 << 16)
        self._chk(self._L.bpgpu_profile_report(self.h, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, n, ms = line.split()
            out[name] = (int(n), float(ms))
        return out


def _profile_report_of(L, h):
    buf = C.create_string_buffer(1 << 16)
    if L.bpgpu_profile_report(h, buf, len(buf)) != 0:
        raise BpgpuError("bpgpu_profile_report failed")
    out = {}
    for line in buf.value.decode().splitlines():
        name, n, ms = line.split()
        out[name] = (int(n), float(ms))
    return out


class Ticket:
    """One request in flight in the pool's combining queue (bpgpu_pool_rangeproof_submit_ts)."""

    def __init__(self, pool, h, nb, verdict, msm, ts_out):
        self.pool, self.h, self.nb, self._v, self._m, self._t = pool, h, nb, verdict, msm, ts_out

    def done(self):
        return self.h is None or self.pool._L.bpgpu_pool_ticket_done(self.pool.h, self.h) == 1

    def wait(self):
        if self.h is not None:
            h, self.h = self.h, None
            self.pool._chk(self.pool._L.bpgpu_pool_ticket_wait(self.pool.h, h))
        out = [self._v.raw[:self.nb]]
        if self._m is not None:
            out.append(self._m.raw[:32 * self.nb])
        if self._t is not None:
            out.append(self._t.raw[:TRANSCRIPT_BYTES * self.nb])
        return tuple(out) if len(out) > 1 else out[0]


class DevTicket:
    """One submitted device-pointer batch (bpgpu_pool_rangeproof_submit_dev_ex)."""

    def __init__(self, pool, h):
        self.pool, self.h = pool, h

    def stream_wait(self, stream):
        """make `stream` (raw hipStream_t as int; 0 = default stream) wait on the device for this batch's verdicts"""
        self.pool._chk(self.pool._L.bpgpu_pool_ticket_stream_wait(self.pool.h, self.h, stream or None))

    def done(self):
        return self.h is None or self.pool._L.bpgpu_pool_ticket_done(self.pool.h, self.h) == 1

    def wait(self):
        if self.h is not None:
            h, self.h = self.h, None
            self.pool._chk(self.pool._L.bpgpu_pool_ticket_wait(self.pool.h, h))


class Pool:
    """Owns one bpgpu_pool: `lanes` contexts on each of `devices` (include/bpgpu.h, "pool").  The scheduler of the library:
    one synchronous call for any number of proofs from host memory (verify), or asynchronous device-pointer batches that the
    pool coalesces into wide launch chains (submit_dev / flush / wait)."""

    def __init__(self, devices=(0,), lanes=0, **options):
        self._L = lib()
        h = C.c_void_p()
        devs = (C.c_int * len(devices))(*devices)
        rc = self._L.bpgpu_pool_create(devs, len(devices), lanes, C.byref(h))
        if rc != 0:
            raise BpgpuError("bpgpu_pool_create(devices=%s) failed: %s (no CPU fallback)" % (list(devices), ERR_NAMES.get(rc, rc)))
        self.h = h
        for k, v in options.items():
            if v is not None:
                self.set_option(k, v)

    def close(self):
        if getattr(self, "h", None):
            self._L.bpgpu_pool_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise BpgpuError("%s: %s" % (ERR_NAMES.get(rc, rc), self._L.bpgpu_pool_last_error(self.h).decode()))

    def set_option(self, key, value):
        self._chk(self._L.bpgpu_pool_set_option(self.h, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_int64(0)
        self._chk(self._L.bpgpu_pool_get_option(self.h, key.encode(), C.byref(v)))
        return v.value

    @property
    def n_devices(self):
        return self._L.bpgpu_pool_devices(self.h)

    @property
    def n_lanes(self):
        return self._L.bpgpu_pool_lanes(self.h)

    def gens_create(self, gens_capacity, party_capacity):
        self._chk(self._L.bpgpu_pool_gens_create(self.h, gens_capacity, party_capacity))

    def gens_load(self, gens_capacity, party_capacity, G, H, B, B_blinding):
        assert len(G) == len(H) == 32 * gens_capacity * party_capacity
        self._chk(self._L.bpgpu_pool_gens_load(self.h, gens_capacity, party_capacity, G, H, B, B_blinding))

    def gens_add_shape(self, n2, m2):
        """a second, larger-window table for the smaller shape (n2, m2), both windows re-balanced under one budget (bpgpu_pool_gens_add_shape)"""
        self._chk(self._L.bpgpu_pool_gens_add_shape(self.h, n2, m2))

    def rangeproof_verify(self, n, m, proofs, proof_len, commitments, label, rng64=None, want_msm=False):
        """ONE call, any number of proofs, host memory in and out (bpgpu_pool_rangeproof_verify)."""
        nb = len(proofs) // proof_len if proof_len else 0
        assert len(proofs) == nb * proof_len and len(commitments) == 32 * m * nb
        assert rng64 is None or len(rng64) == 64 * nb
        verdict = C.create_string_buffer(max(nb, 1))
        msm = C.create_string_buffer(32 * max(nb, 1)) if want_msm else None
        self._chk(self._L.bpgpu_pool_rangeproof_verify(self.h, n, m, nb, proofs, proof_len, commitments, label, len(label), rng64, verdict, msm))
        return (verdict.raw[:nb], msm.raw[:32 * nb]) if want_msm else verdict.raw[:nb]

    def rangeproof_verify_ts(self, n, m, proofs, proof_len, commitments, transcripts, rng64=None, want_msm=False, want_transcripts=True):
        """The reference's call shape (bpgpu_pool_rangeproof_verify_ts): blocking, any number of threads at once; `transcripts`
        is ONE 208-byte state for all proofs or one per proof.  Returns (verdict, [msm,] [transcripts_out])."""
        nb = len(proofs) // proof_len if proof_len else 0
        assert len(proofs) == nb * proof_len and len(commitments) == 32 * m * nb
        assert len(transcripts) in (TRANSCRIPT_BYTES, TRANSCRIPT_BYTES * nb)
        stride = TRANSCRIPT_BYTES if (len(transcripts) == TRANSCRIPT_BYTES * nb and nb != 1) or (nb == 1 and want_transcripts) else 0
        assert rng64 is None or len(rng64) == 64 * nb
        verdict = C.create_string_buffer(max(nb, 1))
        msm = C.create_string_buffer(32 * max(nb, 1)) if want_msm else None
        ts_out = C.create_string_buffer(TRANSCRIPT_BYTES * max(nb, 1)) if want_transcripts else None
        self._chk(self._L.bpgpu_pool_rangeproof_verify_ts(self.h, n, m, nb, proofs, proof_len, commitments, transcripts, stride, rng64, verdict, msm, ts_out))
        out = [verdict.raw[:nb]]
        if want_msm:
            out.append(msm.raw[:32 * nb])
        if want_transcripts:
            out.append(ts_out.raw[:TRANSCRIPT_BYTES * nb])
        return tuple(out) if len(out) > 1 else out[0]

    def submit_ts(self, n, m, proofs, proof_len, commitments, transcripts, rng64=None, want_msm=False, want_transcripts=True):
        """The non-blocking form (bpgpu_pool_rangeproof_submit_ts).  Returns a Ticket; ticket.wait() -> as rangeproof_verify_ts."""
        nb = len(proofs) // proof_len if proof_len else 0
        assert len(proofs) == nb * proof_len and len(commitments) == 32 * m * nb
        assert len(transcripts) in (TRANSCRIPT_BYTES, TRANSCRIPT_BYTES * nb)
        stride = TRANSCRIPT_BYTES if (len(transcripts) == TRANSCRIPT_BYTES * nb and nb != 1) or (nb == 1 and want_transcripts) else 0
        verdict = C.create_string_buffer(max(nb, 1))
        msm = C.create_string_buffer(32 * max(nb, 1)) if want_msm else None
        ts_out = C.create_string_buffer(TRANSCRIPT_BYTES * max(nb, 1)) if want_transcripts else None
        t = C.c_void_p()
        self._chk(self._L.bpgpu_pool_rangeproof_submit_ts(self.h, n, m, nb, proofs, proof_len, commitments, transcripts, stride, rng64, verdict, msm, ts_out,
                                                          C.byref(t)))
        return Ticket(self, t, nb, verdict, msm, ts_out)

    def submit_dev(self, dev_index, n, m, nbatch, d_proofs, proof_len, d_commitments, label, d_rng64, d_verdict, d_msm_out=None):
        """queue a device-resident batch (raw device pointers as ints); see flush / wait"""
        self._chk(self._L.bpgpu_pool_rangeproof_submit_dev(self.h, dev_index, n, m, nbatch, d_proofs, proof_len, d_commitments, label, len(label),
                                                           d_rng64, d_verdict, d_msm_out))

    def submit_dev_ex(self, dev_index, n, m, nbatch, d_proofs, proof_len, d_commitments, label, d_rng64, d_verdict, d_msm_out=None, producer_stream=None,
                      want_ticket=True):
        """submit_dev with a completion contract (bpgpu_pool_rangeproof_submit_dev_ex): producer_stream = raw hipStream_t (int; 0 = the
        legacy default stream) the inputs are produced on, or None; returns a DevTicket (or None)."""
        t = C.c_void_p()
        self._chk(self._L.bpgpu_pool_rangeproof_submit_dev_ex(self.h, dev_index, n, m, nbatch, d_proofs, proof_len, d_commitments, label, len(label), d_rng64,
                                                              d_verdict, d_msm_out, producer_stream or None, 0 if producer_stream is None else 1,
                                                              C.byref(t) if want_ticket else None))
        return DevTicket(self, t) if want_ticket else None

    def submit_rlc_dev(self, dev_index, n, m, nbatch, d_proofs, proof_len, d_commitments, label, d_rng64, d_verdict, d_batch_out=None, producer_stream=None,
                       want_ticket=False):
        """queue a device-resident batch for the batch-combined check (bpgpu_pool_rangeproof_submit_rlc_dev): one identity check per launch chain"""
        t = C.c_void_p()
        self._chk(self._L.bpgpu_pool_rangeproof_submit_rlc_dev(self.h, dev_index, n, m, nbatch, d_proofs, proof_len, d_commitments, label, len(label), d_rng64,
                                                               d_verdict, d_batch_out, producer_stream or None, 0 if producer_stream is None else 1,
                                                               C.byref(t) if want_ticket else None))
        return DevTicket(self, t) if want_ticket else None

    def gather_dev(self, root, parts, sizes, d_dst, stream=None):
        """bpgpu_pool_gather_dev: every shard's device-resident verdict bytes to one buffer on pool device `root` (raw device pointers as ints)"""
        n = self.n_devices
        assert len(parts) == len(sizes) == n
        self._chk(self._L.bpgpu_pool_gather_dev(self.h, root, (C.c_void_p * n)(*parts), (C.c_size_t * n)(*sizes), d_dst, stream or None))

    def flush(self):
        self._chk(self._L.bpgpu_pool_flush(self.h))

    def wait(self):
        self._chk(self._L.bpgpu_pool_wait(self.h))

    # ---- the boundary function through the combining queue (blocking, any number of threads) ----
    def msm_batch_shared(self, n, m, nbatch, n_unique, gen_scalars, uniq_scalars, uniq_points):
        """optional_multiscalar_mul in the mega-check shape, one call per MSM (or a few) from any thread (bpgpu_pool_msm_batch_shared)."""
        assert len(gen_scalars) == 32 * (2 * n * m + 2) * nbatch
        assert len(uniq_scalars) == len(uniq_points) == 32 * n_unique * nbatch
        out, st = C.create_string_buffer(32 * max(nbatch, 1)), C.create_string_buffer(max(nbatch, 1))
        self._chk(self._L.bpgpu_pool_msm_batch_shared(self.h, n, m, nbatch, n_unique, gen_scalars, uniq_scalars, uniq_points, out, st))
        return out.raw[:32 * nbatch], st.raw[:nbatch]

    def msm_batch(self, n_terms, scalars, points):
        """a ragged batch of multiscalar multiplications (bpgpu_pool_msm_batch): MSMs of equal length share launch chains."""
        nb = len(n_terms)
        assert len(scalars) == len(points) == 32 * sum(n_terms)
        nt = (C.c_uint32 * max(nb, 1))(*n_terms)
        out, st = C.create_string_buffer(32 * max(nb, 1)), C.create_string_buffer(max(nb, 1))
        self._chk(self._L.bpgpu_pool_msm_batch(self.h, nb, nt, scalars, points, out, st))
        return out.raw[:32 * nb], st.raw[:nb]

    def msm_shared_submit_dev(self, dev_index, n, m, nbatch, n_unique, d_gen_scalars, d_uniq_scalars, d_uniq_points, d_out, d_status, producer_stream=None,
                              want_ticket=False):
        """device-resident MSM batch on the next lane of pool device dev_index (bpgpu_pool_msm_batch_shared_submit_dev; raw device pointers as ints)"""
        t = C.c_void_p()
        self._chk(self._L.bpgpu_pool_msm_batch_shared_submit_dev(self.h, dev_index, n, m, nbatch, n_unique, d_gen_scalars, d_uniq_scalars or None, d_uniq_points or None,
                                                                 d_out, d_status, producer_stream or None, 0 if producer_stream is None else 1,
                                                                 C.byref(t) if want_ticket else None))
        return DevTicket(self, t) if want_ticket else None

    def ipp_verify(self, n, proofs, proof_len, label, Gf, Hf, P, Q, G, H, want_msm=False):
        """InnerProductProof::verify for len(proofs) / proof_len proofs through the queue (bpgpu_pool_ipp_verify)."""
        nb = len(proofs) // proof_len if proof_len else 0
        assert len(Gf) == len(Hf) == len(G) == len(H) == 32 * n * nb and len(P) == len(Q) == 32 * nb
        verdict = C.create_string_buffer(max(nb, 1))
        msm = C.create_string_buffer(32 * max(nb, 1)) if want_msm else None
        self._chk(self._L.bpgpu_pool_ipp_verify(self.h, n, nb, proofs, proof_len, label, len(label), Gf, Hf, P, Q, G, H, verdict, msm))
        return (verdict.raw[:nb], msm.raw[:32 * nb]) if want_msm else verdict.raw[:nb]

    def trace_dump(self, path):
        self._chk(self._L.bpgpu_pool_trace_dump(self.h, path.encode()))

    # ---- instrumentation (per lane context) ----
    def _lanes(self, every=1):
        for d in range(self.n_devices):
            for l in range(0, self.n_lanes, every):
                yield self._L.bpgpu_pool_lane(self.h, d, l)

    def profile_enable(self, on=True, every=1):
        for h in self._lanes():
            self._L.bpgpu_profile_enable(h, 0)
        if on:
            for h in self._lanes(every):
                self._L.bpgpu_profile_enable(h, 1)

    def profile_reset(self):
        for h in self._lanes():
            self._L.bpgpu_profile_reset(h)

    def profile_report(self):
        out = {}
        for h in self._lanes():
            for name, (cnt, ms) in _profile_report_of(self._L, h).items():
                o = out.get(name, (0, 0.0))
                out[name] = (o[0] + cnt, o[1] + ms)
        return out
