"""TEST INFRASTRUCTURE (oracle): restatement of the deterministic rng behind the reference's golden vectors.

`tests/range_proof.rs:108-113` of the reference draws the eight blinding factors of its value commitments from
    let mut test_rng = ChaChaRng::from_seed([24u8; 32]);   Scalar::random(&mut test_rng)
with rand_chacha 0.2 (Cargo.toml:38; not vendored in /root/reference).  Published algorithm, restated here:
  * ChaChaRng = ChaCha20 (20 rounds), key = the 32 seed bytes, 64-bit block counter starting at 0 in state words 12..13,
    64-bit stream id 0 in words 14..15 (the "djb" layout); the generator emits the keystream blocks in order and
    `fill_bytes` consumes its 32-bit output words in order, little-endian;
  * Scalar::random (curve25519-dalek 2.x): 64 bytes from the rng -> Scalar::from_bytes_mod_order_wide.
Pinned by the reference itself: commit(j, r_j) must equal the committed `vc[j]` (tests/test_oracle.py)."""
import struct

MASK = 0xFFFFFFFF


def _rotl(x, n):
    return ((x << n) | (x >> (32 - n))) & MASK


def _qr(x, a, b, c, d):
    x[a] = (x[a] + x[b]) & MASK; x[d] = _rotl(x[d] ^ x[a], 16)
    x[c] = (x[c] + x[d]) & MASK; x[b] = _rotl(x[b] ^ x[c], 12)
    x[a] = (x[a] + x[b]) & MASK; x[d] = _rotl(x[d] ^ x[a], 8)
    x[c] = (x[c] + x[d]) & MASK; x[b] = _rotl(x[b] ^ x[c], 7)


def chacha20_block(key32, counter, stream=0):
    st = list(struct.unpack("<4I", b"expand 32-byte k")) + list(struct.unpack("<8I", key32)) + [counter & MASK, (counter >> 32) & MASK, stream & MASK, (stream >> 32) & MASK]
    x = st[:]
    for _ in range(10):
        _qr(x, 0, 4, 8, 12); _qr(x, 1, 5, 9, 13); _qr(x, 2, 6, 10, 14); _qr(x, 3, 7, 11, 15)
        _qr(x, 0, 5, 10, 15); _qr(x, 1, 6, 11, 12); _qr(x, 2, 7, 8, 13); _qr(x, 3, 4, 9, 14)
    return struct.pack("<16I", *[(a + b) & MASK for a, b in zip(x, st)])


class ChaChaRng:
    """rand_chacha::ChaChaRng::from_seed(seed) as far as fill_bytes of whole words goes"""

    def __init__(self, seed32):
        assert len(seed32) == 32
        self.key, self.counter, self.buf = bytes(seed32), 0, b""

    def fill_bytes(self, n):
        while len(self.buf) < n:
            self.buf += chacha20_block(self.key, self.counter)
            self.counter += 1
        out, self.buf = self.buf[:n], self.buf[n:]
        return out


L = 2 ** 252 + 27742317777372353535851937790883648493


def scalar_random(rng):
    """Scalar::random: from_bytes_mod_order_wide of 64 rng bytes; returned as 32 canonical little-endian bytes"""
    return (int.from_bytes(rng.fill_bytes(64), "little") % L).to_bytes(32, "little")


def golden_blindings():
    """the r_j of tests/range_proof.rs:108-113: vc[j] = j B + r_j B~, j = 0..7"""
    rng = ChaChaRng(bytes([24]) * 32)
    return [scalar_random(rng) for _ in range(8)]
