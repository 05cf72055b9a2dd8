// Keccak-f[1600], SHAKE256 / SHA3-512 and the Merlin transcript (STROBE-128)
// for host and device.
//
// Replaces, on the verification path, the reference's use of merlin ^2
// (src/transcript.rs:43-95: append_message / append_u64 / challenge_bytes) and
// of sha3 0.8 (src/generators.rs:48, 64-72).  The 200-byte sponge state is
// accessed through `kstate`, an array of 50 u32 words with a stride: on the
// device it points into LDS with stride = workgroup size (word w of lane t at
// w*stride + t: conflict-free, and byte positions that are uniform across the
// wavefront need no register indexing); on the host the stride is 1.
#ifndef BPGPU_KECCAK_H
#define BPGPU_KECCAK_H
#include "fe25519.h"

namespace bp {

struct kstate {
    uint32_t *w;       // word i at w[i * stride]
    uint32_t stride;
};

BP_HD uint32_t ks_get32(const kstate &s, uint32_t i) { return s.w[i * s.stride]; }
BP_HD void ks_set32(const kstate &s, uint32_t i, uint32_t v) { s.w[i * s.stride] = v; }
BP_HD void ks_xor8(const kstate &s, uint32_t pos, uint32_t b) { s.w[(pos >> 2) * s.stride] ^= b << (8 * (pos & 3)); }
BP_HD uint32_t ks_get8(const kstate &s, uint32_t pos) { return (s.w[(pos >> 2) * s.stride] >> (8 * (pos & 3))) & 0xffu; }
BP_HD void ks_clear8(const kstate &s, uint32_t pos) { s.w[(pos >> 2) * s.stride] &= ~(0xffu << (8 * (pos & 3))); }
BP_HD void ks_zero(const kstate &s) {
    for (uint32_t i = 0; i < 50; i++) ks_set32(s, i, 0);
}

// 64-bit rotate by a compile-time amount, written on 32-bit halves: each half is one funnel shift
// (v_alignbit_b32, full rate) instead of two quarter-rate 64-bit shifts and an OR
BP_HD uint64_t rotl64(uint64_t x, int n) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    if (n >= 32) {
        const uint32_t t = lo;
        lo = hi;
        hi = t;
        n -= 32;
    }
    if (n == 0) return ((uint64_t)hi << 32) | lo;
    const uint32_t nlo = (lo << n) | (hi >> (32 - n));
    const uint32_t nhi = (hi << n) | (lo >> (32 - n));
    return ((uint64_t)nhi << 32) | nlo;
}

// 24 rounds on 25 lanes held in registers (all indices static after unrolling)
BP_HD void keccak_f1600_lanes(uint64_t a[25]) {
    const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    for (int r = 0; r < 24; r++) {
        uint64_t c[5], d[5], b[25];
#pragma unroll
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
#pragma unroll
        for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
        // rho + pi : b[y, 2x+3y] = rot(a[x, y])
        b[0] = a[0];
        b[10] = rotl64(a[1], 1);   b[20] = rotl64(a[2], 62);  b[5] = rotl64(a[3], 28);   b[15] = rotl64(a[4], 27);
        b[16] = rotl64(a[5], 36);  b[1] = rotl64(a[6], 44);   b[11] = rotl64(a[7], 6);   b[21] = rotl64(a[8], 55);
        b[6] = rotl64(a[9], 20);   b[7] = rotl64(a[10], 3);   b[17] = rotl64(a[11], 10); b[2] = rotl64(a[12], 43);
        b[12] = rotl64(a[13], 25); b[22] = rotl64(a[14], 39); b[23] = rotl64(a[15], 41); b[8] = rotl64(a[16], 45);
        b[18] = rotl64(a[17], 15); b[3] = rotl64(a[18], 21);  b[13] = rotl64(a[19], 8);  b[14] = rotl64(a[20], 18);
        b[24] = rotl64(a[21], 2);  b[9] = rotl64(a[22], 61);  b[19] = rotl64(a[23], 56); b[4] = rotl64(a[24], 14);
#pragma unroll
        for (int y = 0; y < 25; y += 5) {
#pragma unroll
            for (int x = 0; x < 5; x++) a[y + x] = b[y + x] ^ ((~b[y + (x + 1) % 5]) & b[y + (x + 2) % 5]);
        }
        a[0] ^= RC[r];
    }
}

BP_HD void keccak_f1600(const kstate &s) {
#ifdef BP_EXP_NOKECCAK   // timing experiments only (tools/archive/stage1_breakdown.py)
    ks_set32(s, 0, ks_get32(s, 0) + 1);
    return;
#endif
    uint64_t a[25];
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = (uint64_t)ks_get32(s, 2 * i) | ((uint64_t)ks_get32(s, 2 * i + 1) << 32);
    keccak_f1600_lanes(a);
#pragma unroll
    for (int i = 0; i < 25; i++) {
        ks_set32(s, 2 * i, (uint32_t)a[i]);
        ks_set32(s, 2 * i + 1, (uint32_t)(a[i] >> 32));
    }
}

// ONE out-of-line copy per kernel for the byte-wise STROBE paths: they reach the permutation from ~10 inlined framing sites
// (strobe_run_f below), and 24 unrolled rounds are ~50 KB of machine code each time -- round 3's launch 1 was 926 KB.  The
// scripted replay of the range-proof path (rp_transcript_scripted) has a single call site and keeps its copy inline.
BP_HD_NOINLINE void keccak_f1600_outofline(uint32_t *w, uint32_t stride) {
    kstate s;
    s.w = w;
    s.stride = stride;
    keccak_f1600(s);
}

// the permutation with a constant XOR mask folded into the load of the first `nmask` state words (rp_script.h: all framing bytes
// of a transcript span at once; the mask is uniform across the wavefront)
BP_HD void keccak_f1600_masked(const kstate &s, const uint32_t *mask, uint32_t nmask) {
    uint64_t a[25];
#pragma unroll
    for (int i = 0; i < 25; i++) {
        uint32_t lo = ks_get32(s, 2 * i), hi = ks_get32(s, 2 * i + 1);
        if ((uint32_t)(2 * i) < nmask) lo ^= mask[2 * i];
        if ((uint32_t)(2 * i + 1) < nmask) hi ^= mask[2 * i + 1];
        a[i] = (uint64_t)lo | ((uint64_t)hi << 32);
    }
#ifdef BP_EXP_NOKECCAK
    a[0] += 1;
#else
    keccak_f1600_lanes(a);
#endif
#pragma unroll
    for (int i = 0; i < 25; i++) {
        ks_set32(s, 2 * i, (uint32_t)a[i]);
        ks_set32(s, 2 * i + 1, (uint32_t)(a[i] >> 32));
    }
}

// ---- Keccak-f[1600] on 25 lanes of a 32-lane group: one 64-bit state word per lane -------------------------------------------
// For NARROW launch chains (a call of a few proofs, the combining queue's small chains) launch 1 is one lane replaying one
// transcript: ~16 permutations x 24 rounds x ~270 dependent instructions, 13 us each at a lone wavefront's issue rate -- while
// 63 lanes of the wavefront idle.  Here lane j = x + 5 y of a half-wavefront holds a[x, y]; a round is four exchange phases
// (nine 64-bit fetches from other lanes: ds_bpermute_b32 pairs) and a handful of ALU operations per lane:
//   P1  c = XOR of the lane's column (4 fetches)            P2  a ^= c[x-1] ^ rot(c[x+1], 1) (2 fetches)
//   P3  b[pi(j)] = rot(a, rho[j]): rotate, then every lane fetches from its source (1)     P4  a = b ^ (~b[x+1] & b[x+2]) (2), iota
// (round 6: P2 and P3 are one stage, see kc_p23)
// Seven times the lane-instructions of the serial form, a quarter of its latency: for latency-bound chains only (option
// "transcript_coop").  The phase functions take the fetch as a functor, so that the CPU harness runs the same bodies on a snapshot
// of the 25 lane values (tests/cpu_harness: keccak_coop_host).
BP_HD uint32_t kc_rho(uint32_t j) {
    const uint8_t R[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    return R[j];
}
BP_HD uint32_t kc_pi_src(uint32_t i) {   // b[i] = rot(a[kc_pi_src(i)])
    const uint8_t S[25] = {0, 6, 12, 18, 24, 3, 9, 10, 16, 22, 1, 7, 13, 19, 20, 4, 5, 11, 17, 23, 2, 8, 14, 15, 21};
    return S[i];
}
BP_HD uint64_t kc_rc(uint32_t r) {
    const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    return RC[r];
}
BP_HD uint64_t kc_rotl_var(uint64_t x, uint32_t n) { return n ? (x << n) | (x >> (64 - n)) : x; }
// G: uint64_t operator()(uint64_t mine, uint32_t src_lane, int which): the value `mine` as lane src_lane (0 .. 24) of the group holds it
// (`which` names the exchanged quantity for the host twin, whose lanes are array slots: 0 = the state word, 1 = the column parity)
// Round 6: THREE exchange stages per round instead of four -- theta's second half rides on the pi fetch: lane j fetches its source's
// state word AND the two column parities that source would have fetched (parities are the same in every row: taken from row 0), applies
// theta and the source's rho rotation itself.  Nine 64-bit fetches per round as before, one dependent LDS round trip (~120 cycles of a
// lone wavefront) fewer.
template <class G>
BP_HD uint64_t kc_p1(uint64_t a, uint32_t x, uint32_t y, G &g) {
    uint64_t c = a;
#pragma unroll
    for (uint32_t d = 1; d < 5; d++) c ^= g(a, x + 5 * ((y + d) % 5), 0);
    return c;
}
// b[j] = rot(a[src] ^ c[src.x - 1] ^ rot(c[src.x + 1], 1), rho[src]), src = kc_pi_src(j)
template <class G>
BP_HD uint64_t kc_p23(uint64_t a, uint64_t c, uint32_t src, uint32_t rho_src, G &g) {
    const uint32_t sx = src % 5;
    const uint64_t as = g(a, src, 0), cm = g(c, (sx + 4) % 5, 1), cp = g(c, (sx + 1) % 5, 1);
    return kc_rotl_var(as ^ cm ^ ((cp << 1) | (cp >> 63)), rho_src);
}
template <class G>
BP_HD uint64_t kc_p4(uint64_t b, uint32_t x, uint32_t y, uint32_t j, uint32_t r, G &g) {
    const uint64_t b1 = g(b, (x + 1) % 5 + 5 * y, 0), b2 = g(b, (x + 2) % 5 + 5 * y, 0);
    uint64_t a = b ^ (~b1 & b2);
    if (j == 0) a ^= kc_rc(r);
    return a;
}
#if defined(__HIP_DEVICE_COMPILE__)
struct kc_fetch_dev {
    uint32_t base;   // first lane of this group in the wavefront (0 or 32)
    __device__ __forceinline__ uint64_t operator()(uint64_t mine, uint32_t src, int) const {
        const int addr = (int)((base + src) << 2);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)(uint32_t)mine);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)(uint32_t)(mine >> 32));
        return ((uint64_t)hi << 32) | lo;
    }
};
// all 64 lanes of the wavefront call this together (the exchanges are wavefront-wide instructions); lanes 25 .. 31 of a group
// run along on lane 24's coordinates and keep nothing.  st: the GROUP's sponge state (the same pointer in all its lanes, stride 1),
// written by the group's leader before and read by it after: the caller brackets the call with workgroup barriers.
__device__ __forceinline__ void keccak_f1600_masked_coop(const kstate &st, const uint32_t *mask, uint32_t nmask, uint32_t lane) {
    const uint32_t jj = lane & 31, j = jj < 25 ? jj : 24, x = j % 5, y = j / 5;
    const uint32_t src = kc_pi_src(j), rho_src = kc_rho(src);
    kc_fetch_dev g;
    g.base = lane & 32;
    uint32_t lo = st.w[(2 * j) * st.stride], hi = st.w[(2 * j + 1) * st.stride];
    if (2 * j < nmask) lo ^= mask[2 * j];
    if (2 * j + 1 < nmask) hi ^= mask[2 * j + 1];
    uint64_t a = ((uint64_t)hi << 32) | lo;
#ifdef BP_EXP_NOKECCAK   // timing experiments only
    a += 1;
#else
#pragma unroll 1
    for (uint32_t r = 0; r < 24; r++) {
        const uint64_t c = kc_p1(a, x, y, g);
        const uint64_t b = kc_p23(a, c, src, rho_src, g);
        a = kc_p4(b, x, y, j, r, g);
    }
#endif
    if (jj < 25) {
        st.w[(2 * j) * st.stride] = (uint32_t)a;
        st.w[(2 * j + 1) * st.stride] = (uint32_t)(a >> 32);
    }
}
#else
// host twin (CPU harness): the same phase functions, every phase on a snapshot of the 25 lanes' values
struct kc_fetch_host {
    const uint64_t *snap[2];
    uint64_t operator()(uint64_t, uint32_t src, int which) const { return snap[which][src]; }
};
inline void keccak_f1600_masked_coop(const kstate &st, const uint32_t *mask, uint32_t nmask, uint32_t) {
    uint64_t a[25], c[25], t[25];
    for (uint32_t j = 0; j < 25; j++) {
        uint32_t lo = st.w[(2 * j) * st.stride], hi = st.w[(2 * j + 1) * st.stride];
        if (2 * j < nmask) lo ^= mask[2 * j];
        if (2 * j + 1 < nmask) hi ^= mask[2 * j + 1];
        a[j] = ((uint64_t)hi << 32) | lo;
    }
    for (uint32_t r = 0; r < 24; r++) {
        kc_fetch_host g;
        g.snap[0] = a;
        g.snap[1] = nullptr;
        for (uint32_t j = 0; j < 25; j++) c[j] = kc_p1(a[j], j % 5, j / 5, g);
        g.snap[1] = c;
        for (uint32_t j = 0; j < 25; j++) t[j] = kc_p23(a[j], c[j], kc_pi_src(j), kc_rho(kc_pi_src(j)), g);
        g.snap[0] = t;
        for (uint32_t j = 0; j < 25; j++) c[j] = kc_p4(t[j], j % 5, j / 5, j, r, g);
        for (uint32_t j = 0; j < 25; j++) a[j] = c[j];
    }
    for (uint32_t j = 0; j < 25; j++) {
        st.w[(2 * j) * st.stride] = (uint32_t)a[j];
        st.w[(2 * j + 1) * st.stride] = (uint32_t)(a[j] >> 32);
    }
}
#endif

// ---- plain sponges (generator derivation) -------------------------------------
struct sponge {
    kstate st;
    uint32_t pos, rate;
};
BP_HD void sponge_init(sponge &k, kstate st, uint32_t rate) {
    k.st = st;
    k.pos = 0;
    k.rate = rate;
    ks_zero(st);
}
BP_HD void sponge_absorb(sponge &k, const uint8_t *in, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) {
        ks_xor8(k.st, k.pos++, in[i]);
        if (k.pos == k.rate) {
            keccak_f1600(k.st);
            k.pos = 0;
        }
    }
}
BP_HD void sponge_finish(sponge &k, uint32_t suffix) {
    ks_xor8(k.st, k.pos, suffix);
    ks_xor8(k.st, k.rate - 1, 0x80);
    keccak_f1600(k.st);
    k.pos = 0;
}
BP_HD void sponge_squeeze(sponge &k, uint8_t *out, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) {
        if (k.pos == k.rate) {
            keccak_f1600(k.st);
            k.pos = 0;
        }
        out[i] = (uint8_t)ks_get8(k.st, k.pos++);
    }
}
#define BP_SHAKE256_RATE 136
#define BP_SHA3_512_RATE 72

// ---- STROBE-128 as used by Merlin -------------------------------------------
#define BP_STROBE_R 166
#define BP_FLAG_I 1
#define BP_FLAG_A 2
#define BP_FLAG_C 4
#define BP_FLAG_T 8
#define BP_FLAG_M 16
#define BP_FLAG_K 32

struct strobe {
    kstate st;
    uint32_t pos, pos_begin, cur_flags;
};

BP_HD void strobe_run_f(strobe &t) {
    ks_xor8(t.st, t.pos, t.pos_begin);
    ks_xor8(t.st, t.pos + 1, 0x04);
    ks_xor8(t.st, BP_STROBE_R + 1, 0x80);
#ifdef BP_KECCAK_OUTOFLINE   // (k_rp1.hip: the byte-wise variant of launch 1; the prover / inner-product kernels keep their inlined copies and registers)
    keccak_f1600_outofline(t.st.w, t.st.stride);
#else
    keccak_f1600(t.st);
#endif
    t.pos = 0;
    t.pos_begin = 0;
}
BP_HD void strobe_absorb1(strobe &t, uint32_t b) {
    ks_xor8(t.st, t.pos++, b);
    if (t.pos == BP_STROBE_R) strobe_run_f(t);
}
BP_HD void strobe_absorb(strobe &t, const uint8_t *d, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) strobe_absorb1(t, d[i]);
}
// absorb n32 little-endian words (messages that already sit in registers).  A word that does not
// straddle the rate boundary is XORed into (at most two) state words at once instead of four
// byte-wide read-modify-writes; positions are uniform across the wavefront.
BP_HD void strobe_absorb_words(strobe &t, const uint32_t *w, uint32_t n32) {
    for (uint32_t i = 0; i < n32; i++) {
        const uint32_t x = w[i];
        if (t.pos + 4 <= BP_STROBE_R) {
            const uint32_t wi = t.pos >> 2, sh = (t.pos & 3) * 8;
            t.st.w[wi * t.st.stride] ^= x << sh;
            if (sh) t.st.w[(wi + 1) * t.st.stride] ^= x >> (32 - sh);
            t.pos += 4;
            if (t.pos == BP_STROBE_R) strobe_run_f(t);
        } else {
            strobe_absorb1(t, x & 0xff);
            strobe_absorb1(t, (x >> 8) & 0xff);
            strobe_absorb1(t, (x >> 16) & 0xff);
            strobe_absorb1(t, x >> 24);
        }
    }
}
BP_HD uint32_t strobe_squeeze1(strobe &t) {
    const uint32_t b = ks_get8(t.st, t.pos);
    ks_clear8(t.st, t.pos++);
    if (t.pos == BP_STROBE_R) strobe_run_f(t);
    return b;
}
// squeeze one little-endian word (read, then zero the squeezed bytes, as STROBE's PRF does)
BP_HD uint32_t strobe_squeeze_word(strobe &t) {
    if (t.pos + 4 <= BP_STROBE_R) {
        const uint32_t wi = t.pos >> 2, sh = (t.pos & 3) * 8;
        uint32_t *w0 = &t.st.w[wi * t.st.stride];
        uint32_t x;
        if (sh) {
            uint32_t *w1 = &t.st.w[(wi + 1) * t.st.stride];
            x = (*w0 >> sh) | (*w1 << (32 - sh));
            *w0 &= (1u << sh) - 1u;
            *w1 &= 0xffffffffu << sh;
        } else {
            x = *w0;
            *w0 = 0;
        }
        t.pos += 4;
        if (t.pos == BP_STROBE_R) strobe_run_f(t);
        return x;
    }
    uint32_t x = strobe_squeeze1(t);
    x |= strobe_squeeze1(t) << 8;
    x |= strobe_squeeze1(t) << 16;
    x |= strobe_squeeze1(t) << 24;
    return x;
}
BP_HD void strobe_begin_op(strobe &t, uint32_t flags, bool more) {
    if (more) return;   // continuation of the current operation (flags must equal cur_flags)
    const uint32_t old_begin = t.pos_begin;
    t.pos_begin = t.pos + 1;
    t.cur_flags = flags;
    strobe_absorb1(t, old_begin);
    strobe_absorb1(t, flags);
    if ((flags & (BP_FLAG_C | BP_FLAG_K)) && t.pos != 0) strobe_run_f(t);
}
BP_HD void strobe_meta_ad(strobe &t, const uint8_t *d, uint32_t n, bool more) {
    strobe_begin_op(t, BP_FLAG_M | BP_FLAG_A, more);
    strobe_absorb(t, d, n);
}
BP_HD void strobe_ad(strobe &t, const uint8_t *d, uint32_t n, bool more) {
    strobe_begin_op(t, BP_FLAG_A, more);
    strobe_absorb(t, d, n);
}
BP_HD void strobe_meta_len(strobe &t, uint32_t n) {
    strobe_begin_op(t, BP_FLAG_M | BP_FLAG_A, true);
    strobe_absorb1(t, n & 0xff);
    strobe_absorb1(t, (n >> 8) & 0xff);
    strobe_absorb1(t, (n >> 16) & 0xff);
    strobe_absorb1(t, n >> 24);
}

// Strobe128::new(b"Merlin v1.0")
BP_HD void merlin_strobe_init(strobe &t, kstate st) {
    t.st = st;
    ks_zero(st);
    const uint8_t hdr[18] = {1, BP_STROBE_R + 2, 1, 0, 1, 96, 'S', 'T', 'R', 'O', 'B', 'E', 'v', '1', '.', '0', '.', '2'};
    for (uint32_t i = 0; i < 18; i++) ks_xor8(st, i, hdr[i]);
    keccak_f1600(st);
    t.pos = 0;
    t.pos_begin = 0;
    t.cur_flags = 0;
    const uint8_t proto[11] = {'M', 'e', 'r', 'l', 'i', 'n', ' ', 'v', '1', '.', '0'};
    strobe_meta_ad(t, proto, 11, false);
}
// Transcript::append_message(label, msg)
BP_HD void merlin_append_message(strobe &t, const uint8_t *label, uint32_t label_len, const uint8_t *msg, uint32_t n) {
    strobe_meta_ad(t, label, label_len, false);
    strobe_meta_len(t, n);
    strobe_ad(t, msg, n, false);
}
// append_message with a 32-byte message held as 8 LE words
BP_HD void merlin_append_words8(strobe &t, const uint8_t *label, uint32_t label_len, const uint32_t w[8]) {
    strobe_meta_ad(t, label, label_len, false);
    strobe_meta_len(t, 32);
    strobe_begin_op(t, BP_FLAG_A, false);
    strobe_absorb_words(t, w, 8);
}
BP_HD void merlin_append_u64(strobe &t, const uint8_t *label, uint32_t label_len, uint64_t x) {
    uint8_t b[8];
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(x >> (8 * i));
    merlin_append_message(t, label, label_len, b, 8);
}
// Transcript::challenge_bytes(label, 64) as 16 LE words
BP_HD void merlin_challenge_words16(strobe &t, const uint8_t *label, uint32_t label_len, uint32_t out[16]) {
    strobe_meta_ad(t, label, label_len, false);
    strobe_meta_len(t, 64);
    strobe_begin_op(t, BP_FLAG_I | BP_FLAG_A | BP_FLAG_C, false);
    for (uint32_t i = 0; i < 16; i++) out[i] = strobe_squeeze_word(t);
}
BP_HD void merlin_challenge_bytes(strobe &t, const uint8_t *label, uint32_t label_len, uint8_t *out, uint32_t n) {
    strobe_meta_ad(t, label, label_len, false);
    strobe_meta_len(t, n);
    strobe_begin_op(t, BP_FLAG_I | BP_FLAG_A | BP_FLAG_C, false);
    for (uint32_t i = 0; i < n; i++) out[i] = (uint8_t)strobe_squeeze1(t);
}

}  // namespace bp
#endif
