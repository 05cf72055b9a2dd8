// GF(2^255-19) for CDNA4 VALU lanes: 10 limbs of 25.5 bits in 32-bit VGPRs,
// 64-bit column accumulators fed by v_mad_u64_u32.
//
// This is the arithmetic the reference reaches through curve25519-dalek's
// FieldElement (u64 / AVX2 backends; Cargo.toml:21,42 of the reference) when it
// calls RistrettoPoint::optional_multiscalar_mul (src/range_proof/mod.rs:421).
// The representation is chosen for the GPU, not translated: 32-bit limbs keep a
// field element in 10 VGPRs, the limb products are 32x32->64 multiply-adds
// (one v_mad_u64_u32 each, no carry chains inside the column sums) and
// additions are carry-free.
//
// Limb discipline ("reduced" vs "lazy"):
//   reduced: even limbs < 2^26 + 2^19, odd limbs < 2^25 + 2^19 (output of mul/sq/carry/sub)
//   lazy   : limb-wise sum of up to three reduced values (< 3*2^26 + small)
//   fe_mul / fe_sq accept lazy inputs (column sums stay below 2^64, see fe_mul),
//   fe_sub accepts lazy inputs and returns a reduced value.
// The header compiles for the device (hipcc) and for the host (g++) so that the
// identical code is unit-tested on CPU (tests/cpu_harness); with BP_FE_CHECK the
// host build asserts the discipline on every multiplication.
#ifndef BPGPU_FE25519_H
#define BPGPU_FE25519_H
#include <stdint.h>

#if defined(__HIPCC__)
#define BP_HD __host__ __device__ __forceinline__
#define BP_HD_NOINLINE __host__ __device__ inline __attribute__((noinline))   // one copy of a big body (I-cache)
#else
#define BP_HD inline
#define BP_HD_NOINLINE inline
#endif

#ifdef BP_FE_CHECK
#include <assert.h>
#define BP_ASSERT(x) assert(x)
#else
#define BP_ASSERT(x) ((void)0)
#endif

namespace bp {

// (8-byte aligned so that whole-element loads and stores can use 64/128-bit memory operations)
struct __attribute__((aligned(8))) fe {
    uint32_t v[10];
};

#define BP_M26 0x3ffffffu
#define BP_M25 0x1ffffffu

BP_HD void fe_0(fe &h) {
#pragma unroll
    for (int i = 0; i < 10; i++) h.v[i] = 0;
}
BP_HD void fe_1(fe &h) {
    fe_0(h);
    h.v[0] = 1;
}

// h = f + g, limb-wise, no carry (lazy)
BP_HD void fe_add(fe &h, const fe &f, const fe &g) {
#pragma unroll
    for (int i = 0; i < 10; i++) h.v[i] = f.v[i] + g.v[i];
}

// weak reduction of 32-bit limbs: result is "reduced"
BP_HD void fe_carry(fe &h) {
    uint32_t c;
    c = h.v[0] >> 26; h.v[0] &= BP_M26; h.v[1] += c;
    c = h.v[1] >> 25; h.v[1] &= BP_M25; h.v[2] += c;
    c = h.v[2] >> 26; h.v[2] &= BP_M26; h.v[3] += c;
    c = h.v[3] >> 25; h.v[3] &= BP_M25; h.v[4] += c;
    c = h.v[4] >> 26; h.v[4] &= BP_M26; h.v[5] += c;
    c = h.v[5] >> 25; h.v[5] &= BP_M25; h.v[6] += c;
    c = h.v[6] >> 26; h.v[6] &= BP_M26; h.v[7] += c;
    c = h.v[7] >> 25; h.v[7] &= BP_M25; h.v[8] += c;
    c = h.v[8] >> 26; h.v[8] &= BP_M26; h.v[9] += c;
    c = h.v[9] >> 25; h.v[9] &= BP_M25; h.v[0] += 19 * c;
}

// h = f - g (inputs may be lazy); adds 8p so every limb difference stays positive.
// 8p limbs: 8*(2^26-19), 8*(2^25-1), 8*(2^26-1), ...  (g limbs < 2^28 / 2^27)
BP_HD void fe_sub(fe &h, const fe &f, const fe &g) {
    h.v[0] = f.v[0] + 0x1fffff68u - g.v[0];
    h.v[1] = f.v[1] + 0x0ffffff8u - g.v[1];
    h.v[2] = f.v[2] + 0x1ffffff8u - g.v[2];
    h.v[3] = f.v[3] + 0x0ffffff8u - g.v[3];
    h.v[4] = f.v[4] + 0x1ffffff8u - g.v[4];
    h.v[5] = f.v[5] + 0x0ffffff8u - g.v[5];
    h.v[6] = f.v[6] + 0x1ffffff8u - g.v[6];
    h.v[7] = f.v[7] + 0x0ffffff8u - g.v[7];
    h.v[8] = f.v[8] + 0x1ffffff8u - g.v[8];
    h.v[9] = f.v[9] + 0x0ffffff8u - g.v[9];
    fe_carry(h);
}

#ifdef BP_FE_CHECK
inline void fe_check_reduced(const fe &f) {
    for (int i = 0; i < 10; i++) BP_ASSERT(f.v[i] < ((i & 1) ? 0x2080000u : 0x4080000u));  // 2^25(26) + 2^19
}
#else
BP_HD void fe_check_reduced(const fe &) {}
#endif
// h = f - g for REDUCED f and g, without the carry chain: f + 2p - g, limb-wise.  The result is a lazy value (every limb <= reduced +
// 2^27 / 2^26: within the "three reduced values" bound fe_mul / fe_sq accept, also as the operand that is premultiplied by 19).
// 30 instructions less than fe_sub; profiles/r06/microbench_madd.txt: +4 .. 5 % on a mixed addition with two of its three
// subtractions in this form.  2p limbs: 2 (2^26 - 19), 2 (2^25 - 1), 2 (2^26 - 1), ...
BP_HD void fe_sub_rr(fe &h, const fe &f, const fe &g) {
    fe_check_reduced(f);
    fe_check_reduced(g);
    h.v[0] = f.v[0] + 0x7ffffdau - g.v[0];
    h.v[1] = f.v[1] + 0x3fffffeu - g.v[1];
    h.v[2] = f.v[2] + 0x7fffffeu - g.v[2];
    h.v[3] = f.v[3] + 0x3fffffeu - g.v[3];
    h.v[4] = f.v[4] + 0x7fffffeu - g.v[4];
    h.v[5] = f.v[5] + 0x3fffffeu - g.v[5];
    h.v[6] = f.v[6] + 0x7fffffeu - g.v[6];
    h.v[7] = f.v[7] + 0x3fffffeu - g.v[7];
    h.v[8] = f.v[8] + 0x7fffffeu - g.v[8];
    h.v[9] = f.v[9] + 0x3fffffeu - g.v[9];
}

BP_HD void fe_neg(fe &h, const fe &f) {
    fe z;
    fe_0(z);
    fe_sub(h, z, f);
}

// branch-free select: h = b ? g : f
BP_HD void fe_select(fe &h, const fe &f, const fe &g, bool b) {
#pragma unroll
    for (int i = 0; i < 10; i++) h.v[i] = b ? g.v[i] : f.v[i];
}

// carry the ten 64-bit column sums into reduced 32-bit limbs
BP_HD void fe_reduce_columns(fe &h, uint64_t c[10]) {
    c[1] += c[0] >> 26; h.v[0] = (uint32_t)c[0] & BP_M26;
    c[2] += c[1] >> 25; h.v[1] = (uint32_t)c[1] & BP_M25;
    c[3] += c[2] >> 26; h.v[2] = (uint32_t)c[2] & BP_M26;
    c[4] += c[3] >> 25; h.v[3] = (uint32_t)c[3] & BP_M25;
    c[5] += c[4] >> 26; h.v[4] = (uint32_t)c[4] & BP_M26;
    c[6] += c[5] >> 25; h.v[5] = (uint32_t)c[5] & BP_M25;
    c[7] += c[6] >> 26; h.v[6] = (uint32_t)c[6] & BP_M26;
    c[8] += c[7] >> 25; h.v[7] = (uint32_t)c[7] & BP_M25;
    c[9] += c[8] >> 26; h.v[8] = (uint32_t)c[8] & BP_M26;
    uint64_t top = c[9] >> 25; h.v[9] = (uint32_t)c[9] & BP_M25;
    uint64_t t = (uint64_t)h.v[0] + 19 * top;   // top < 2^39
    h.v[0] = (uint32_t)t & BP_M26;
    h.v[1] += (uint32_t)(t >> 26);               // < 2^25 + 2^18
}

#ifdef BP_FE_CHECK
inline void fe_check_lazy(const fe &f) {
    for (int i = 0; i < 10; i++) BP_ASSERT(f.v[i] <= ((i & 1) ? 0x6100000u : 0xc200000u));  // 3*2^25(26) + slack
}
#else
BP_HD void fe_check_lazy(const fe &) {}
#endif

// h = f * g.  Term f_i*g_j lands in column (i+j) mod 10, times 19 when i+j >= 10
// (2^255 = 19) and times 2 when i and j are both odd (25.5-bit radix).
// Bound: |f_i|,|g_j| <= 3*2^26 -> 2f <= 2^28.6, 19g <= 2^31.9 (fits u32);
// column sum <= 10 * 2^28.6 * 2^31.9 < 2^63.9.
BP_HD void fe_mul(fe &h, const fe &f, const fe &g) {
    fe_check_lazy(f);
    fe_check_lazy(g);
    uint32_t g19[10], f2[10];
#pragma unroll
    for (int i = 0; i < 10; i++) {
        g19[i] = 19u * g.v[i];
        f2[i] = 2u * f.v[i];
    }
    uint64_t c[10];
#pragma unroll
    for (int k = 0; k < 10; k++) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
#pragma unroll
        for (int j = 0; j < 10; j++) {
            const int k = i + j;
            const uint32_t a = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
            const uint32_t b = (k >= 10) ? g19[j] : g.v[j];
            c[k % 10] += (uint64_t)a * b;
        }
    }
    fe_reduce_columns(h, c);
}

// h = f^2 (55 limb products instead of 100)
BP_HD void fe_sq(fe &h, const fe &f) {
    fe_check_lazy(f);
    uint32_t f19[10], f2[10], f4[10];
#pragma unroll
    for (int i = 0; i < 10; i++) {
        f19[i] = 19u * f.v[i];
        f2[i] = 2u * f.v[i];
        f4[i] = 4u * f.v[i];
    }
    uint64_t c[10];
#pragma unroll
    for (int k = 0; k < 10; k++) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
#pragma unroll
        for (int j = i; j < 10; j++) {
            const int k = i + j;
            const bool odd2 = (i & 1) && (j & 1);
            // multiplicity: 2 for off-diagonal, another 2 when both odd
            const uint32_t a = (i == j) ? (odd2 ? f2[i] : f.v[i]) : (odd2 ? f4[i] : f2[i]);
            const uint32_t b = (k >= 10) ? f19[j] : f.v[j];
            c[k % 10] += (uint64_t)a * b;
        }
    }
    fe_reduce_columns(h, c);
}

BP_HD void fe_sqn(fe &h, const fe &f, int n) {
    fe_sq(h, f);
    for (int i = 1; i < n; i++) fe_sq(h, h);
}

// load 32 little-endian bytes given as 8 u32 words; bit 255 is ignored
BP_HD void fe_from_words(fe &h, const uint32_t w[8]) {
    h.v[0] = w[0] & BP_M26;
    h.v[1] = ((w[0] >> 26) | (w[1] << 6)) & BP_M25;
    h.v[2] = ((w[1] >> 19) | (w[2] << 13)) & BP_M26;
    h.v[3] = ((w[2] >> 13) | (w[3] << 19)) & BP_M25;
    h.v[4] = (w[3] >> 6) & BP_M26;
    h.v[5] = w[4] & BP_M25;
    h.v[6] = ((w[4] >> 25) | (w[5] << 7)) & BP_M26;
    h.v[7] = ((w[5] >> 19) | (w[6] << 13)) & BP_M25;
    h.v[8] = ((w[6] >> 12) | (w[7] << 20)) & BP_M26;
    h.v[9] = (w[7] >> 6) & BP_M25;
}

// canonical (fully reduced mod p) little-endian encoding as 8 u32 words
BP_HD void fe_to_words(uint32_t w[8], const fe &f) {
    fe t = f;
    fe_carry(t);
    fe_carry(t);
    // q = 1 iff t >= p  (t < 2^255 + small)
    uint32_t q = (t.v[0] + 19) >> 26;
    q = (t.v[1] + q) >> 25; q = (t.v[2] + q) >> 26; q = (t.v[3] + q) >> 25; q = (t.v[4] + q) >> 26;
    q = (t.v[5] + q) >> 25; q = (t.v[6] + q) >> 26; q = (t.v[7] + q) >> 25; q = (t.v[8] + q) >> 26;
    q = (t.v[9] + q) >> 25;
    t.v[0] += 19 * q;
    uint32_t c;
    c = t.v[0] >> 26; t.v[0] &= BP_M26; t.v[1] += c;
    c = t.v[1] >> 25; t.v[1] &= BP_M25; t.v[2] += c;
    c = t.v[2] >> 26; t.v[2] &= BP_M26; t.v[3] += c;
    c = t.v[3] >> 25; t.v[3] &= BP_M25; t.v[4] += c;
    c = t.v[4] >> 26; t.v[4] &= BP_M26; t.v[5] += c;
    c = t.v[5] >> 25; t.v[5] &= BP_M25; t.v[6] += c;
    c = t.v[6] >> 26; t.v[6] &= BP_M26; t.v[7] += c;
    c = t.v[7] >> 25; t.v[7] &= BP_M25; t.v[8] += c;
    c = t.v[8] >> 26; t.v[8] &= BP_M26; t.v[9] += c;
    t.v[9] &= BP_M25;
    w[0] = t.v[0] | (t.v[1] << 26);
    w[1] = (t.v[1] >> 6) | (t.v[2] << 19);
    w[2] = (t.v[2] >> 13) | (t.v[3] << 13);
    w[3] = (t.v[3] >> 19) | (t.v[4] << 6);
    w[4] = t.v[5] | (t.v[6] << 25);
    w[5] = (t.v[6] >> 7) | (t.v[7] << 19);
    w[6] = (t.v[7] >> 13) | (t.v[8] << 12);
    w[7] = (t.v[8] >> 20) | (t.v[9] << 6);
}

BP_HD bool fe_isneg(const fe &f) {
    uint32_t w[8];
    fe_to_words(w, f);
    return w[0] & 1;
}
BP_HD bool fe_iszero(const fe &f) {
    uint32_t w[8];
    fe_to_words(w, f);
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r |= w[i];
    return r == 0;
}
BP_HD bool fe_eq(const fe &f, const fe &g) {
    fe d;
    fe_sub(d, f, g);
    return fe_iszero(d);
}
// h = b ? -h : h
BP_HD void fe_cneg(fe &h, bool b) {
    fe n;
    fe_neg(n, h);
    fe_select(h, h, n, b);
}
BP_HD void fe_abs(fe &h) { fe_cneg(h, fe_isneg(h)); }

// z^(2^250-1) and z^11: shared prefix of inversion and of the (p-5)/8 power
BP_HD void fe_pow2_250m1(fe &out, fe &z11, const fe &z) {
    fe t0, t1, t2;
    fe_sq(t0, z);
    fe_sqn(t1, t0, 2);
    fe_mul(t1, z, t1);
    fe_mul(t0, t0, t1);
    z11 = t0;
    fe_sq(t0, t0);
    fe_mul(t0, t1, t0);
    fe_sqn(t1, t0, 5);   fe_mul(t0, t1, t0);
    fe_sqn(t1, t0, 10);  fe_mul(t1, t1, t0);
    fe_sqn(t2, t1, 20);  fe_mul(t1, t2, t1);
    fe_sqn(t1, t1, 10);  fe_mul(t0, t1, t0);
    fe_sqn(t1, t0, 50);  fe_mul(t1, t1, t0);
    fe_sqn(t2, t1, 100); fe_mul(t1, t2, t1);
    fe_sqn(t1, t1, 50);  fe_mul(out, t1, t0);
}
BP_HD void fe_invert(fe &out, const fe &z) {
    fe t, z11;
    fe_pow2_250m1(t, z11, z);
    fe_sqn(t, t, 5);
    fe_mul(out, t, z11);
}
BP_HD void fe_pow22523(fe &out, const fe &z) {
    fe t, z11;
    fe_pow2_250m1(t, z11, z);
    // two plain squarings, not fe_sqn(t, t, 2): with the in-place two-step loop here the compiler emitted twice the code for the whole
    // chain and 301 instead of 134 registers -- every kernel that decodes or encodes a point carried it (tools/kernel_resources.py)
    fe_sq(t, t);
    fe_sq(t, t);
    fe_mul(out, t, z);
}

}  // namespace bp
#endif
