"""Pure-Python big-int twin of the Bulletproofs range-proof verifier (and a
non-constant-time prover used only to synthesise test inputs).

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product path (the
``bulletproofs_amd`` package, ``libbpgpu.so``) may import or call this file.
It exists so that two independent restatements (this one and the C oracle in
``oracle/c``) can be cross-checked against each other and against the
reference's own golden vectors (``/root/reference/tests/range_proof.rs:16-95``).

What it restates (reference file:line):
  * verify_multiple_with_rng ........ src/range_proof/mod.rs:345-452
  * RangeProof::from_bytes .......... src/range_proof/mod.rs:504-538
  * delta ........................... src/range_proof/mod.rs:587-593
  * verification_scalars ............ src/inner_product_proof.rs:198-253
  * InnerProductProof::from_bytes ... src/inner_product_proof.rs:373-407
  * InnerProductProof::create ....... src/inner_product_proof.rs:38-193
  * TranscriptProtocol .............. src/transcript.rs:43-95
  * PedersenGens / BulletproofGens .. src/generators.rs:44-53, 58-104, 157-259
  * exp_iter / sum_of_powers ........ src/util.rs:44-67, 240-261
  * prover (dealer/party) ........... src/range_proof/{party,dealer}.rs

The arithmetic below the crate (curve25519-dalek ^2, merlin ^2, sha3 0.8;
Cargo.toml:21-31) is not in /root/reference; it is restated from the public
specifications: ristretto255 = RFC 9496, Merlin = STROBE-128 over
Keccak-f[1600].
"""
import hashlib

# ----------------------------------------------------------------------------
# field GF(2^255-19)
# ----------------------------------------------------------------------------
P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493
D = (-121665 * pow(121666, P - 2, P)) % P
SQRT_M1 = pow(2, (P - 1) // 4, P)


def inv(x):
    return pow(x, P - 2, P)


def is_neg(x):
    return (x % P) & 1


def fabs(x):
    x %= P
    return P - x if x & 1 else x


def sqrt_ratio_i(u, v):
    """RFC 9496 SQRT_RATIO_M1.  Returns (was_square, r)."""
    u %= P
    v %= P
    v3 = v * v % P * v % P
    v7 = v3 * v3 % P * v % P
    r = u * v3 % P * pow(u * v7 % P, (P - 5) // 8, P) % P
    check = v * r % P * r % P
    correct = check == u
    flipped = check == (P - u) % P
    flipped_i = check == (P - u) * SQRT_M1 % P
    if flipped or flipped_i:
        r = r * SQRT_M1 % P
    r = fabs(r)
    return (correct or flipped), r


def invsqrt(x):
    return sqrt_ratio_i(1, x)


# derived constants (checked against SURVEY.md Appendix A in tests)
ONE_MINUS_D_SQ = (1 - D * D) % P
D_MINUS_ONE_SQ = (D - 1) * (D - 1) % P
_ok, INVSQRT_A_MINUS_D = invsqrt((-1 - D) % P)
assert _ok
# sqrt(a*d - 1) with a = -1: the RFC 9496 constant is the ODD root.
_ok, _r = sqrt_ratio_i((-D - 1) % P, 1)
assert _ok
SQRT_AD_MINUS_ONE = _r if _r & 1 else P - _r

# ----------------------------------------------------------------------------
# Edwards points, extended coordinates (X, Y, Z, T)
# ----------------------------------------------------------------------------
IDENT = (0, 1, 1, 0)


def pt_add(p, q):
    X1, Y1, Z1, T1 = p
    X2, Y2, Z2, T2 = q
    A = (Y1 - X1) * (Y2 - X2) % P
    B = (Y1 + X1) * (Y2 + X2) % P
    C = 2 * D * T1 % P * T2 % P
    Dd = 2 * Z1 * Z2 % P
    E = B - A
    F = Dd - C
    G = Dd + C
    H = B + A
    return (E * F % P, G * H % P, F * G % P, E * H % P)


def pt_neg(p):
    return ((-p[0]) % P, p[1], p[2], (-p[3]) % P)


def pt_dbl(p):
    return pt_add(p, p)


def pt_mul(s, p):
    s %= L
    acc = IDENT
    for bit in bin(s)[2:] if s else "":
        acc = pt_dbl(acc)
        if bit == "1":
            acc = pt_add(acc, p)
    return acc


def msm(scalars, points):
    """Plain sum of scalar multiples (Straus with 4-bit windows)."""
    tables = []
    for p in points:
        t = [IDENT, p]
        for _ in range(14):
            t.append(pt_add(t[-1], p))
        tables.append(t)
    ss = [s % L for s in scalars]
    acc = IDENT
    for w in range(63, -1, -1):
        for _ in range(4):
            acc = pt_dbl(acc)
        for s, t in zip(ss, tables):
            d = (s >> (4 * w)) & 15
            if d:
                acc = pt_add(acc, t[d])
    return acc


def pt_is_identity(p):
    """Ristretto identity coset test."""
    return p[0] % P == 0 or p[1] % P == 0


def pt_eq(p, q):
    return (p[0] * q[1] - p[1] * q[0]) % P == 0 or (p[0] * q[0] - p[1] * q[1]) % P == 0


# ----------------------------------------------------------------------------
# ristretto255
# ----------------------------------------------------------------------------
def decompress(b):
    s = int.from_bytes(b, "little")
    if s >= P or (s & 1):
        return None
    ss = s * s % P
    u1 = (1 - ss) % P
    u2 = (1 + ss) % P
    u2_sqr = u2 * u2 % P
    v = (-(D * u1 % P * u1) - u2_sqr) % P
    ok, I = invsqrt(v * u2_sqr % P)
    Dx = I * u2 % P
    Dy = I * Dx % P * v % P
    x = fabs(2 * s * Dx % P)
    y = u1 * Dy % P
    t = x * y % P
    if (not ok) or is_neg(t) or y == 0:
        return None
    return (x, y, 1, t)


def compress(p):
    X, Y, Z, T = p
    u1 = (Z + Y) * (Z - Y) % P
    u2 = X * Y % P
    _, I = invsqrt(u1 * u2 % P * u2 % P)
    i1 = I * u1 % P
    i2 = I * u2 % P
    z_inv = i1 * i2 % P * T % P
    den_inv = i2
    if is_neg(T * z_inv % P):
        X, Y = Y * SQRT_M1 % P, X * SQRT_M1 % P
        den_inv = i1 * INVSQRT_A_MINUS_D % P
    if is_neg(X * z_inv % P):
        Y = (-Y) % P
    s = fabs(den_inv * (Z - Y) % P)
    return s.to_bytes(32, "little")


def elligator(r0):
    r = SQRT_M1 * r0 % P * r0 % P
    Ns = (r + 1) * ONE_MINUS_D_SQ % P
    c = P - 1
    Dd = (c - D * r) % P * ((r + D) % P) % P
    sq, s = sqrt_ratio_i(Ns, Dd)
    s_prime = (-fabs(s * r0 % P)) % P
    if not sq:
        s = s_prime
        c = r
    Nt = (c * (r - 1) % P * D_MINUS_ONE_SQ - Dd) % P
    s_sq = s * s % P
    W0 = 2 * s * Dd % P
    W1 = Nt * SQRT_AD_MINUS_ONE % P
    W2 = (1 - s_sq) % P
    W3 = (1 + s_sq) % P
    return (W0 * W3 % P, W2 * W1 % P, W1 * W3 % P, W0 * W2 % P)


def from_uniform_bytes(b):
    assert len(b) == 64
    lo = int.from_bytes(b[:32], "little") & ((1 << 255) - 1)
    hi = int.from_bytes(b[32:], "little") & ((1 << 255) - 1)
    return pt_add(elligator(lo % P), elligator(hi % P))


BASEPOINT_COMPRESSED = bytes.fromhex(
    "e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76")
BASEPOINT = decompress(BASEPOINT_COMPRESSED)

# ----------------------------------------------------------------------------
# Keccak-f[1600], STROBE-128, Merlin
# ----------------------------------------------------------------------------
_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61],
        [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M64 = (1 << 64) - 1


def _rol(x, n):
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _M64 if n else x


def keccak_f(state):
    """state: bytearray(200), permuted in place."""
    A = [[int.from_bytes(state[8 * (x + 5 * y):8 * (x + 5 * y) + 8], "little")
          for y in range(5)] for x in range(5)]
    for rnd in range(24):
        C = [A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4] for x in range(5)]
        Dl = [C[(x - 1) % 5] ^ _rol(C[(x + 1) % 5], 1) for x in range(5)]
        A = [[A[x][y] ^ Dl[x] for y in range(5)] for x in range(5)]
        B = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                B[y][(2 * x + 3 * y) % 5] = _rol(A[x][y], _ROT[x][y])
        A = [[B[x][y] ^ ((~B[(x + 1) % 5][y]) & B[(x + 2) % 5][y]) for y in range(5)]
             for x in range(5)]
        A[0][0] ^= _RC[rnd]
    for x in range(5):
        for y in range(5):
            state[8 * (x + 5 * y):8 * (x + 5 * y) + 8] = A[x][y].to_bytes(8, "little")


class Strobe128:
    R = 166
    FLAG_I, FLAG_A, FLAG_C, FLAG_T, FLAG_M, FLAG_K = 1, 2, 4, 8, 16, 32

    def __init__(self, label):
        st = bytearray(200)
        st[0:6] = bytes([1, self.R + 2, 1, 0, 1, 96])
        st[6:18] = b"STROBEv1.0.2"
        keccak_f(st)
        self.st = st
        self.pos = 0
        self.pos_begin = 0
        self.cur_flags = 0
        self.meta_ad(label, False)

    def clone(self):
        o = Strobe128.__new__(Strobe128)
        o.st = bytearray(self.st)
        o.pos, o.pos_begin, o.cur_flags = self.pos, self.pos_begin, self.cur_flags
        return o

    def _run_f(self):
        self.st[self.pos] ^= self.pos_begin
        self.st[self.pos + 1] ^= 0x04
        self.st[self.R + 1] ^= 0x80
        keccak_f(self.st)
        self.pos = 0
        self.pos_begin = 0

    def _absorb(self, data):
        for b in data:
            self.st[self.pos] ^= b
            self.pos += 1
            if self.pos == self.R:
                self._run_f()

    def _squeeze(self, n):
        out = bytearray(n)
        for i in range(n):
            out[i] = self.st[self.pos]
            self.st[self.pos] = 0
            self.pos += 1
            if self.pos == self.R:
                self._run_f()
        return bytes(out)

    def _begin_op(self, flags, more):
        if more:
            assert flags == self.cur_flags
            return
        assert not (flags & self.FLAG_T)
        old_begin = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self._absorb(bytes([old_begin, flags]))
        force_f = bool(flags & (self.FLAG_C | self.FLAG_K))
        if force_f and self.pos != 0:
            self._run_f()

    def meta_ad(self, data, more):
        self._begin_op(self.FLAG_M | self.FLAG_A, more)
        self._absorb(data)

    def ad(self, data, more):
        self._begin_op(self.FLAG_A, more)
        self._absorb(data)

    def prf(self, n, more=False):
        self._begin_op(self.FLAG_I | self.FLAG_A | self.FLAG_C, more)
        return self._squeeze(n)


class Transcript:
    """merlin::Transcript (the subset the verifier and prover use)."""

    def __init__(self, label):
        self.strobe = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def clone(self):
        o = Transcript.__new__(Transcript)
        o.strobe = self.strobe.clone()
        return o

    def append_message(self, label, msg):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(len(msg).to_bytes(4, "little"), True)
        self.strobe.ad(msg, False)

    def append_u64(self, label, x):
        self.append_message(label, x.to_bytes(8, "little"))

    def challenge_bytes(self, label, n):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(n.to_bytes(4, "little"), True)
        return self.strobe.prf(n)

    # --- TranscriptProtocol (src/transcript.rs:43-95) ---
    def rangeproof_domain_sep(self, n, m):
        self.append_message(b"dom-sep", b"rangeproof v1")
        self.append_u64(b"n", n)
        self.append_u64(b"m", m)

    def innerproduct_domain_sep(self, n):
        self.append_message(b"dom-sep", b"ipp v1")
        self.append_u64(b"n", n)

    def append_scalar(self, label, s):
        self.append_message(label, (s % L).to_bytes(32, "little"))

    def append_point(self, label, pb):
        self.append_message(label, pb)

    def validate_and_append_point(self, label, pb):
        if pb == bytes(32):
            raise VerificationError()
        self.append_message(label, pb)

    def challenge_scalar(self, label):
        return int.from_bytes(self.challenge_bytes(label, 64), "little") % L


# ----------------------------------------------------------------------------
# errors, generators
# ----------------------------------------------------------------------------
class ProofError(Exception):
    pass


class VerificationError(ProofError):
    pass


class FormatError(ProofError):
    pass


class InvalidBitsize(ProofError):
    pass


class InvalidGeneratorsLength(ProofError):
    pass


class PedersenGens:
    def __init__(self):
        self.B = BASEPOINT
        self.B_blinding = from_uniform_bytes(hashlib.sha3_512(BASEPOINT_COMPRESSED).digest())

    def commit(self, v, blinding):
        return pt_add(pt_mul(v, self.B), pt_mul(blinding, self.B_blinding))


def generators_chain(label, count):
    stream = hashlib.shake_256(b"GeneratorsChain" + label).digest(64 * count)
    return [from_uniform_bytes(stream[64 * i:64 * i + 64]) for i in range(count)]


class BulletproofGens:
    def __init__(self, gens_capacity, party_capacity):
        self.gens_capacity = gens_capacity
        self.party_capacity = party_capacity
        self.G_vec = []
        self.H_vec = []
        for i in range(party_capacity):
            self.G_vec.append(generators_chain(b"G" + i.to_bytes(4, "little"), gens_capacity))
            self.H_vec.append(generators_chain(b"H" + i.to_bytes(4, "little"), gens_capacity))

    def G(self, n, m):
        return [self.G_vec[j][i] for j in range(m) for i in range(n)]

    def H(self, n, m):
        return [self.H_vec[j][i] for j in range(m) for i in range(n)]


# ----------------------------------------------------------------------------
# proof parsing
# ----------------------------------------------------------------------------
def _canonical_scalar(b):
    s = int.from_bytes(b, "little")
    if s >= L:
        raise FormatError()
    return s


class RangeProof:
    def __init__(self):
        self.A = self.S = self.T_1 = self.T_2 = None
        self.t_x = self.t_x_blinding = self.e_blinding = 0
        self.L_vec = []
        self.R_vec = []
        self.a = self.b = 0

    @staticmethod
    def from_bytes(b):
        if len(b) % 32 != 0 or len(b) < 7 * 32:
            raise FormatError()
        pr = RangeProof()
        pr.A, pr.S, pr.T_1, pr.T_2 = (b[32 * i:32 * i + 32] for i in range(4))
        pr.t_x = _canonical_scalar(b[128:160])
        pr.t_x_blinding = _canonical_scalar(b[160:192])
        pr.e_blinding = _canonical_scalar(b[192:224])
        ipp = b[224:]
        ne = len(ipp) // 32
        if ne < 2 or (ne - 2) % 2 != 0:
            raise FormatError()
        lg_n = (ne - 2) // 2
        if lg_n >= 32:
            raise FormatError()
        for i in range(lg_n):
            pr.L_vec.append(ipp[64 * i:64 * i + 32])
            pr.R_vec.append(ipp[64 * i + 32:64 * i + 64])
        pr.a = _canonical_scalar(ipp[64 * lg_n:64 * lg_n + 32])
        pr.b = _canonical_scalar(ipp[64 * lg_n + 32:64 * lg_n + 64])
        return pr

    def to_bytes(self):
        out = self.A + self.S + self.T_1 + self.T_2
        out += self.t_x.to_bytes(32, "little") + self.t_x_blinding.to_bytes(32, "little")
        out += self.e_blinding.to_bytes(32, "little")
        for l, r in zip(self.L_vec, self.R_vec):
            out += l + r
        out += self.a.to_bytes(32, "little") + self.b.to_bytes(32, "little")
        return out


def sum_of_powers(x, n):
    if n & (n - 1):
        return sum(pow(x, i, L) for i in range(n)) % L
    if n in (0, 1):
        return n
    m = n
    result = (1 + x) % L
    factor = x
    while m > 2:
        factor = factor * factor % L
        result = (result + factor * result) % L
        m //= 2
    return result


def delta(n, m, y, z):
    sum_y = sum_of_powers(y, n * m)
    sum_2 = sum_of_powers(2, n)
    sum_z = sum_of_powers(z, m)
    return ((z - z * z) * sum_y - z * z * z * sum_2 * sum_z) % L


def sc_inv(x):
    return pow(x, L - 2, L)


def verification_scalars(pr, n, transcript):
    lg_n = len(pr.L_vec)
    if lg_n >= 32 or n != (1 << lg_n):
        raise VerificationError()
    transcript.innerproduct_domain_sep(n)
    ch = []
    for Lp, Rp in zip(pr.L_vec, pr.R_vec):
        transcript.validate_and_append_point(b"L", Lp)
        transcript.validate_and_append_point(b"R", Rp)
        ch.append(transcript.challenge_scalar(b"u"))
    ch_inv = [sc_inv(u) for u in ch]
    allinv = 1
    for u in ch_inv:
        allinv = allinv * u % L
    ch_sq = [u * u % L for u in ch]
    ch_inv_sq = [u * u % L for u in ch_inv]
    s = [allinv]
    for i in range(1, n):
        lg_i = i.bit_length() - 1
        k = 1 << lg_i
        s.append(s[i - k] * ch_sq[(lg_n - 1) - lg_i] % L)
    return ch_sq, ch_inv_sq, s


def verification_msm_terms(pr, bp_gens, pc_gens, transcript, commitments, n, c):
    """Transcript replay + scalar assembly of verify_multiple_with_rng
    (src/range_proof/mod.rs:345-443).  Returns (scalars, point_bytes_or_points)
    where generator points are given as extended tuples and proof points as
    32-byte encodings, in the reference's order."""
    m = len(commitments)
    if n not in (8, 16, 32, 64):
        raise InvalidBitsize()
    if bp_gens.gens_capacity < n or bp_gens.party_capacity < m:
        raise InvalidGeneratorsLength()
    transcript.rangeproof_domain_sep(n, m)
    for V in commitments:
        transcript.append_point(b"V", V)
    transcript.validate_and_append_point(b"A", pr.A)
    transcript.validate_and_append_point(b"S", pr.S)
    y = transcript.challenge_scalar(b"y")
    z = transcript.challenge_scalar(b"z")
    zz = z * z % L
    minus_z = (-z) % L
    transcript.validate_and_append_point(b"T_1", pr.T_1)
    transcript.validate_and_append_point(b"T_2", pr.T_2)
    x = transcript.challenge_scalar(b"x")
    transcript.append_scalar(b"t_x", pr.t_x)
    transcript.append_scalar(b"t_x_blinding", pr.t_x_blinding)
    transcript.append_scalar(b"e_blinding", pr.e_blinding)
    w = transcript.challenge_scalar(b"w")
    c %= L
    x_sq, x_inv_sq, s = verification_scalars(pr, n * m, transcript)
    a, b = pr.a, pr.b
    powers_of_2 = [pow(2, i, L) for i in range(n)]
    concat_z_and_2 = [pow(z, j, L) * p2 % L for j in range(m) for p2 in powers_of_2]
    g = [(minus_z - a * s_i) % L for s_i in s]
    y_inv = sc_inv(y)
    h = []
    exp_y_inv = 1
    for i in range(n * m):
        s_i_inv = s[n * m - 1 - i]
        h.append((z + exp_y_inv * (zz * concat_z_and_2[i] - b * s_i_inv)) % L)
        exp_y_inv = exp_y_inv * y_inv % L
    vscal = [c * zz % L * pow(z, j, L) % L for j in range(m)]
    basepoint_scalar = (w * (pr.t_x - a * b) + c * (delta(n, m, y, z) - pr.t_x)) % L
    scalars = [1, x, c * x % L, c * x % L * x % L] + x_sq + x_inv_sq
    scalars += [(-pr.e_blinding - c * pr.t_x_blinding) % L, basepoint_scalar] + g + h + vscal
    points = [pr.A, pr.S, pr.T_1, pr.T_2] + pr.L_vec + pr.R_vec
    points += [pc_gens.B_blinding, pc_gens.B] + bp_gens.G(n, m) + bp_gens.H(n, m)
    points += list(commitments)
    assert len(points) == len(scalars)
    return scalars, points


def verify_multiple(pr, bp_gens, pc_gens, transcript, commitments, n, c=1):
    """Returns the 32-byte encoding of the mega-check MSM (all-zero == Ok).
    Raises VerificationError when a point fails to decode."""
    scalars, points = verification_msm_terms(pr, bp_gens, pc_gens, transcript, commitments, n, c)
    dec = []
    for p in points:
        if isinstance(p, (bytes, bytearray)):
            q = decompress(bytes(p))
            if q is None:
                raise VerificationError()
            dec.append(q)
        else:
            dec.append(p)
    return compress(msm(scalars, dec))


# ----------------------------------------------------------------------------
# prover (NOT constant time; test-input synthesis only)
# ----------------------------------------------------------------------------
class ShakeRng:
    """Deterministic byte source: SHAKE256(seed) stream."""

    def __init__(self, seed):
        self.seed = seed
        self.off = 0

    def bytes(self, n):
        out = hashlib.shake_256(self.seed).digest(self.off + n)[self.off:]
        self.off += n
        return out

    def scalar(self):
        return int.from_bytes(self.bytes(64), "little") % L


def inner_product(a, b):
    return sum(x * y for x, y in zip(a, b)) % L


def ipp_create(transcript, Q, G_factors, H_factors, G, H, a, b):
    n = len(G)
    transcript.innerproduct_domain_sep(n)
    L_vec, R_vec = [], []
    first = True
    G = list(G)
    H = list(H)
    a = list(a)
    b = list(b)
    while n != 1:
        n //= 2
        aL, aR, bL, bR = a[:n], a[n:], b[:n], b[n:]
        GL, GR, HL, HR = G[:n], G[n:], H[:n], H[n:]
        cL = inner_product(aL, bR)
        cR = inner_product(aR, bL)
        if first:
            Ls = [x * g % L for x, g in zip(aL, G_factors[n:2 * n])] + \
                 [x * h % L for x, h in zip(bR, H_factors[0:n])] + [cL]
            Rs = [x * g % L for x, g in zip(aR, G_factors[0:n])] + \
                 [x * h % L for x, h in zip(bL, H_factors[n:2 * n])] + [cR]
        else:
            Ls = aL + bR + [cL]
            Rs = aR + bL + [cR]
        Lp = compress(msm(Ls, GR + HL + [Q]))
        Rp = compress(msm(Rs, GL + HR + [Q]))
        L_vec.append(Lp)
        R_vec.append(Rp)
        transcript.append_point(b"L", Lp)
        transcript.append_point(b"R", Rp)
        u = transcript.challenge_scalar(b"u")
        u_inv = sc_inv(u)
        na, nb, nG, nH = [], [], [], []
        for i in range(n):
            na.append((aL[i] * u + u_inv * aR[i]) % L)
            nb.append((bL[i] * u_inv + u * bR[i]) % L)
            if first:
                nG.append(msm([u_inv * G_factors[i] % L, u * G_factors[n + i] % L], [GL[i], GR[i]]))
                nH.append(msm([u * H_factors[i] % L, u_inv * H_factors[n + i] % L], [HL[i], HR[i]]))
            else:
                nG.append(msm([u_inv, u], [GL[i], GR[i]]))
                nH.append(msm([u, u_inv], [HL[i], HR[i]]))
        a, b, G, H = na, nb, nG, nH
        first = False
    return L_vec, R_vec, a[0], b[0]


def prove_multiple(bp_gens, pc_gens, transcript, values, blindings, n, rng):
    """RangeProof::prove_multiple_with_rng (src/range_proof/mod.rs:234-288) with
    the dealer/party MPC run in-line.  Returns (RangeProof, [V bytes])."""
    m = len(values)
    assert n in (8, 16, 32, 64) and m & (m - 1) == 0 and m >= 1
    transcript.rangeproof_domain_sep(n, m)
    Vs = [compress(pc_gens.commit(v, bl)) for v, bl in zip(values, blindings)]
    parties = []
    A = IDENT
    S = IDENT
    for j, v in enumerate(values):
        Gj, Hj = bp_gens.G_vec[j][:n], bp_gens.H_vec[j][:n]
        a_blinding = rng.scalar()
        Aj = pt_mul(a_blinding, pc_gens.B_blinding)
        for i in range(n):
            Aj = pt_add(Aj, Gj[i] if (v >> i) & 1 else pt_neg(Hj[i]))
        s_blinding = rng.scalar()
        s_L = [rng.scalar() for _ in range(n)]
        s_R = [rng.scalar() for _ in range(n)]
        Sj = msm([s_blinding] + s_L + s_R, [pc_gens.B_blinding] + Gj + Hj)
        A = pt_add(A, Aj)
        S = pt_add(S, Sj)
        parties.append(dict(v=v, a_blinding=a_blinding, s_blinding=s_blinding, s_L=s_L, s_R=s_R))
    for V in Vs:
        transcript.append_point(b"V", V)
    A_b, S_b = compress(A), compress(S)
    transcript.append_point(b"A", A_b)
    transcript.append_point(b"S", S_b)
    y = transcript.challenge_scalar(b"y")
    z = transcript.challenge_scalar(b"z")
    T1 = IDENT
    T2 = IDENT
    for j, pt in enumerate(parties):
        v = pt["v"]
        offset_y = pow(y, j * n, L)
        offset_z = pow(z, j, L)
        offset_zz = z * z % L * offset_z % L
        l0, l1, r0, r1 = [], [], [], []
        exp_y = offset_y
        exp_2 = 1
        for i in range(n):
            a_L = (v >> i) & 1
            a_R = (a_L - 1) % L
            l0.append((a_L - z) % L)
            l1.append(pt["s_L"][i])
            r0.append((exp_y * (a_R + z) + offset_zz * exp_2) % L)
            r1.append(exp_y * pt["s_R"][i] % L)
            exp_y = exp_y * y % L
            exp_2 = exp_2 * 2 % L
        t0 = inner_product(l0, r0)
        t2 = inner_product(l1, r1)
        t1 = (inner_product([(p + q) % L for p, q in zip(l0, l1)],
                            [(p + q) % L for p, q in zip(r0, r1)]) - t0 - t2) % L
        t_1_blinding = rng.scalar()
        t_2_blinding = rng.scalar()
        T1 = pt_add(T1, pc_gens.commit(t1, t_1_blinding))
        T2 = pt_add(T2, pc_gens.commit(t2, t_2_blinding))
        pt.update(offset_zz=offset_zz, l0=l0, l1=l1, r0=r0, r1=r1, t0=t0, t1=t1, t2=t2,
                  t_1_blinding=t_1_blinding, t_2_blinding=t_2_blinding)
    T1_b, T2_b = compress(T1), compress(T2)
    transcript.append_point(b"T_1", T1_b)
    transcript.append_point(b"T_2", T2_b)
    x = transcript.challenge_scalar(b"x")
    t_x = t_x_blinding = e_blinding = 0
    l_vec, r_vec = [], []
    for pt, bl in zip(parties, blindings):
        t_x += pt["t0"] + x * (pt["t1"] + x * pt["t2"])
        t_x_blinding += pt["offset_zz"] * bl + x * (pt["t_1_blinding"] + x * pt["t_2_blinding"])
        e_blinding += pt["a_blinding"] + pt["s_blinding"] * x
        l_vec += [(p + q * x) % L for p, q in zip(pt["l0"], pt["l1"])]
        r_vec += [(p + q * x) % L for p, q in zip(pt["r0"], pt["r1"])]
    t_x %= L
    t_x_blinding %= L
    e_blinding %= L
    transcript.append_scalar(b"t_x", t_x)
    transcript.append_scalar(b"t_x_blinding", t_x_blinding)
    transcript.append_scalar(b"e_blinding", e_blinding)
    w = transcript.challenge_scalar(b"w")
    Q = pt_mul(w, pc_gens.B)
    G_factors = [1] * (n * m)
    y_inv = sc_inv(y)
    H_factors = [pow(y_inv, i, L) for i in range(n * m)]
    Lv, Rv, a, b = ipp_create(transcript, Q, G_factors, H_factors,
                              bp_gens.G(n, m), bp_gens.H(n, m), l_vec, r_vec)
    pr = RangeProof()
    pr.A, pr.S, pr.T_1, pr.T_2 = A_b, S_b, T1_b, T2_b
    pr.t_x, pr.t_x_blinding, pr.e_blinding = t_x, t_x_blinding, e_blinding
    pr.L_vec, pr.R_vec, pr.a, pr.b = Lv, Rv, a, b
    return pr, Vs


def ipp_verify(proof_bytes, n, transcript, G_factors, H_factors, P, Q, G, H):
    """InnerProductProof::from_bytes + verify (src/inner_product_proof.rs:260-326, 373-407).
    Points are extended tuples.  Returns compress(expect_P - P) (all-zero == Ok)."""
    b = proof_bytes
    if len(b) % 32 != 0 or len(b) // 32 < 2 or (len(b) // 32 - 2) % 2:
        raise FormatError()
    lg_n = (len(b) // 32 - 2) // 2
    if lg_n >= 32:
        raise FormatError()
    pr = RangeProof()
    pr.L_vec = [b[64 * i:64 * i + 32] for i in range(lg_n)]
    pr.R_vec = [b[64 * i + 32:64 * i + 64] for i in range(lg_n)]
    pr.a = _canonical_scalar(b[64 * lg_n:64 * lg_n + 32])
    pr.b = _canonical_scalar(b[64 * lg_n + 32:64 * lg_n + 64])
    u_sq, u_inv_sq, s = verification_scalars(pr, n, transcript)
    scalars = [pr.a * pr.b % L] + [pr.a * s[i] % L * G_factors[i] % L for i in range(n)] + \
              [pr.b * s[n - 1 - i] % L * H_factors[i] % L for i in range(n)] + [(-u) % L for u in u_sq] + [(-u) % L for u in u_inv_sq] + [L - 1]
    Ls, Rs = [decompress(x) for x in pr.L_vec], [decompress(x) for x in pr.R_vec]
    if any(p is None for p in Ls + Rs):
        raise VerificationError()
    return compress(msm(scalars, [Q] + list(G) + list(H) + Ls + Rs + [P]))


# ----------------------------------------------------------------------------
# LinearProof (src/linear_proof.rs): <a, b> = c for secret a, public b
# ----------------------------------------------------------------------------
def _linear_public_inputs(transcript, n, C, b_vec, G, F, B):
    """linear_proof.rs:73-83 / 196-206; G, F, B are extended tuples, C bytes"""
    transcript.innerproduct_domain_sep(n)
    transcript.append_point(b"C", C)
    for x in b_vec:
        transcript.append_scalar(b"b_i", x)
    for p in G:
        transcript.append_point(b"G_i", compress(p))
    transcript.append_point(b"F", compress(F))
    transcript.append_point(b"B", compress(B))


def linear_create(transcript, rng, C, r, a_vec, b_vec, G_vec, F, B):
    """LinearProof::create(...).to_bytes() (linear_proof.rs:40-173, 322-331); rng.scalar() = Scalar::random"""
    n = len(b_vec)
    assert len(G_vec) == n and len(a_vec) == n and n & (n - 1) == 0 and n
    _linear_public_inputs(transcript, n, C, b_vec, G_vec, F, B)
    a, b, G = list(a_vec), list(b_vec), list(G_vec)
    out = b""
    while n != 1:
        n //= 2
        aL, aR, bL, bR, GL, GR = a[:n], a[n:], b[:n], b[n:], G[:n], G[n:]
        cL, cR = inner_product(aL, bR), inner_product(aR, bL)
        sj, tj = rng.scalar(), rng.scalar()
        Lp = compress(msm(aL + [sj, cL], GR + [B, F]))
        Rp = compress(msm(aR + [tj, cR], GL + [B, F]))
        out += Lp + Rp
        transcript.append_point(b"L", Lp)
        transcript.append_point(b"R", Rp)
        x = transcript.challenge_scalar(b"x_j")
        xi = sc_inv(x)
        a = [(aL[i] + xi * aR[i]) % L for i in range(n)]
        b = [(bL[i] + x * bR[i]) % L for i in range(n)]
        G = [msm([1, x], [GL[i], GR[i]]) for i in range(n)]
        r = (r + x * sj + xi * tj) % L
    s_star, t_star = rng.scalar(), rng.scalar()
    S = compress(msm([t_star, s_star * b[0] % L, s_star], [B, F, G[0]]))
    transcript.append_point(b"S", S)
    x_star = transcript.challenge_scalar(b"x_star")
    a_star = (s_star + x_star * a[0]) % L
    r_star = (t_star + x_star * r) % L
    return out + S + a_star.to_bytes(32, "little") + r_star.to_bytes(32, "little")


def linear_verify(proof_bytes, transcript, C, G, F, B, b_vec):
    """LinearProof::from_bytes + verify (linear_proof.rs:175-236, 240-312, 350-394), computed the way the reference does:
    separate sums, then expect_S compared with S.  Returns compress(expect_S - S) (all-zero == Ok)."""
    pb = proof_bytes
    if len(pb) % 32 != 0 or len(pb) // 32 < 3 or (len(pb) // 32 - 3) % 2:
        raise FormatError()
    lg_n = (len(pb) // 32 - 3) // 2
    if lg_n >= 32:
        raise FormatError()
    Lv = [pb[64 * i:64 * i + 32] for i in range(lg_n)]
    Rv = [pb[64 * i + 32:64 * i + 64] for i in range(lg_n)]
    Sb = pb[64 * lg_n:64 * lg_n + 32]
    a = _canonical_scalar(pb[64 * lg_n + 32:64 * lg_n + 64])
    r = _canonical_scalar(pb[64 * lg_n + 64:64 * lg_n + 96])
    n = len(b_vec)
    if len(G) != n:
        raise InvalidGeneratorsLength()
    _linear_public_inputs(transcript, n, C, b_vec, G, F, B)
    if n != (1 << lg_n):
        raise VerificationError()
    b = list(b_vec)
    ch = []
    m = n
    for Lb, Rb in zip(Lv, Rv):
        transcript.validate_and_append_point(b"L", Lb)
        transcript.validate_and_append_point(b"R", Rb)
        x = transcript.challenge_scalar(b"x_j")
        ch.append(x)
        m //= 2
        b = [(b[i] + x * b[m + i]) % L for i in range(m)]
    ch_inv = [sc_inv(x) for x in ch]
    b0 = b[0]
    transcript.append_point(b"S", Sb)
    x_star = transcript.challenge_scalar(b"x_star")
    Ls, Rs = [decompress(x) for x in Lv], [decompress(x) for x in Rv]
    if any(p is None for p in Ls + Rs):
        raise VerificationError()
    LR = msm(ch + ch_inv, Ls + Rs) if lg_n else msm([0], [B])
    s = [1]
    for i in range(1, n):
        lg_i = i.bit_length() - 1
        s.append(s[i - (1 << lg_i)] * ch[(lg_n - 1) - lg_i] % L)
    G0 = msm(s, list(G))
    Sp, Cp = decompress(Sb), decompress(C)
    if Sp is None or Cp is None:
        raise VerificationError()
    expect = msm([r, a * b0 % L, (-x_star) % L, (-x_star) % L, a], [B, F, Cp, LR, G0])
    return compress(pt_add(expect, pt_neg(Sp)))
