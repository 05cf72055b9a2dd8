// Variable-base multiscalar multiplication, per-lane bodies.
//
// This is the engine behind bpgpu_msm_batch: the drop-in for
// RistrettoPoint::vartime_multiscalar_mul / optional_multiscalar_mul
// (callers: src/range_proof/mod.rs:421, src/inner_product_proof.rs:308,
// src/r1cs/verifier.rs:459 of the reference) for arbitrary points.
//
// Decomposition (every stage is embarrassingly parallel; stages meet in HBM):
//   vb_prepare : lane = term      decode point, build {1..8}P in projective
//                                 Niels form, recode the scalar to signed
//                                 radix-16 digits (s + 0x88..8 trick)
//   vb_window  : lane = (chunk,w) Pippenger-style column sum with the bucket
//                                 step replaced by the 8-entry table lookup:
//                                 W[chunk][w] = sum_k d_{k,w} * P_k
//   vb_colsum  : lane = (msm,w)   add the chunks' column sums
//   vb_horner  : lane = msm       sum_w 16^w * W[w]  (252 doublings), encode
//
// The bodies are plain functions of (thread index, pointers) so the same code
// runs under the CPU harness in tests/cpu_harness.
#ifndef BPGPU_MSM_VB_H
#define BPGPU_MSM_VB_H
#include "ge25519.h"

namespace bp {

#define BP_VB_CHUNK 32      // terms per (chunk) work item
#define BP_VB_WINDOWS 64    // signed radix-16 digits of a canonical scalar

#define BP_STATUS_OK 0
#define BP_STATUS_BAD_POINT 1
#define BP_STATUS_BAD_SCALAR 2

struct vb_chunk {
    uint32_t msm;     // which MSM of the batch
    uint32_t first;   // first global term index
    uint32_t count;   // terms in this chunk (<= BP_VB_CHUNK)
    uint32_t pad;
};

// l = 2^252 + 27742317777372353535851937790883648493, little-endian words
#define BP_L_WORDS {0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0u, 0u, 0u, 0x10000000u}

// raise a per-item status word to at least `code` (deterministic when lanes disagree: max wins)
BP_HD void status_raise(uint32_t *status, uint32_t code) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMax(status, code);
#else
    if (*status < code) *status = code;
#endif
}

BP_HD bool sc_is_canonical(const uint32_t s[8]) {
    const uint32_t l[8] = BP_L_WORDS;
    // s < l  <=>  borrow out of s - l
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t d = (uint64_t)s[i] - l[i] - borrow;
        borrow = (uint32_t)(d >> 63);
    }
    return borrow != 0;
}

// signed radix-16 recoding: r = s + 0x8888...8; digit_w = nibble_w(r) - 8 in [-8, 7]
BP_HD void sc_recode16(uint32_t r[8], const uint32_t s[8]) {
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t t = (uint64_t)s[i] + 0x88888888u + carry;
        r[i] = (uint32_t)t;
        carry = (uint32_t)(t >> 32);
    }
}
BP_HD int sc_digit16(const uint32_t r[8], int w) {
    return (int)((r[w >> 3] >> ((w & 7) * 4)) & 15u) - 8;
}

// signed radix-32 recoding (the wide range-proof chains): r = s + sum_{w<51} 16 * 32^w, digit_w = bits [5w, 5w+5) of r, minus 16,
// in [-16, 15]; 51 digits cover a canonical scalar (r < 2^255), still 8 words
#define BP_VB5_WINDOWS 51
BP_HD void sc_recode32(uint32_t r[8], const uint32_t s[8]) {
    const uint32_t bias[8] = {0x21084210u, 0x08421084u, 0x42108421u, 0x10842108u, 0x84210842u, 0x21084210u, 0x08421084u, 0x42108421u};
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t t = (uint64_t)s[i] + bias[i] + carry;
        r[i] = (uint32_t)t;
        carry = (uint32_t)(t >> 32);
    }
}
BP_HD int sc_digit32(const uint32_t *r /*[8]*/, uint32_t w) {
    const uint32_t bit = 5 * w, i = bit >> 5, sh = bit & 31;
    const uint64_t v = ((uint64_t)(i < 7 ? r[i + 1] : 0u) << 32) | r[i];
    return (int)((uint32_t)(v >> sh) & 31u) - 16;
}

// ---- stage 1 ---------------------------------------------------------------
// multiples 1P .. 8P of a decoded point, projective Niels form
BP_HD void vb_build_table(ge_cached *out /*[8]*/, const ge_ext &p) {
    ge_cached c1, ck;
    ge_to_cached(c1, p);
    out[0] = c1;
    ge_ext cur = p;
    for (int k = 1; k < 8; k++) {
        ge_add_cached(cur, cur, c1, false);
        ge_to_cached(ck, cur);
        out[k] = ck;
    }
}

// multiples 1P .. 16P (radix-32 digits); only_first: just 1P (a point whose coefficient is known to be 1)
BP_HD void vb_build_table16(ge_cached *out /*[16]*/, const ge_ext &p, bool only_first) {
    ge_cached c1, ck;
    ge_to_cached(c1, p);
    out[0] = c1;
    if (only_first) return;
    ge_ext cur = p;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int k = 1; k < 16; k++) {
        ge_add_cached(cur, cur, c1, false);
        ge_to_cached(ck, cur);
        out[k] = ck;
    }
}

// thread t < n_terms_total
BP_HD void vb_prepare_thread(uint32_t t, const vb_chunk *chunks, const uint32_t *term_chunk,
                             const uint32_t *scalars, const uint32_t *points,
                             ge_cached *tab /*[t][8]*/, uint32_t *recoded /*[t][8]*/, uint32_t *status) {
    uint32_t sw[8], pw[8], rw[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        sw[i] = scalars[8 * (uint64_t)t + i];
        pw[i] = points[8 * (uint64_t)t + i];
    }
    const uint32_t msm = chunks[term_chunk[t]].msm;
    ge_ext p;
    const bool ok = ristretto_decompress(p, pw);
    const bool canon = sc_is_canonical(sw);
    if (!canon) status_raise(status + msm, BP_STATUS_BAD_SCALAR);
    else if (!ok) status_raise(status + msm, BP_STATUS_BAD_POINT);
    sc_recode16(rw, sw);
#pragma unroll
    for (int i = 0; i < 8; i++) recoded[8 * (uint64_t)t + i] = rw[i];
    vb_build_table(tab + 8 * (uint64_t)t, p);
}

// ---- stage 2 ---------------------------------------------------------------
// thread = chunk * 64 + w
// colc (optional, only when every MSM has exactly ONE chunk): the window sum IS the column sum; it is
// written as a cached point to colc[msm][w] (input of the quad Horner chain, horner_quad.h) instead of part.
// skip_status (optional): MSMs whose status word is set contribute the identity (batch-combination mode,
// where the window sums of all proofs are added together afterwards).
// tab_hi (optional; very narrow chains): the tables of the points' 2^128 multiples (hi_levels = 2), or of their 2^64, 2^128 and 2^192 multiples
// one after the other, lv_stride entries each (hi_levels = 4) -- windows 64 / levels .. 63 select from them, and the chain that follows adds the
// sums of windows w + 64 / levels, ... to window w's (hw_colsum_horner_msm)
BP_HD void vb_window_thread(uint32_t tid, const vb_chunk *chunks, const ge_cached *tab,
                            const uint32_t *recoded, ge_ext *part /*[chunk][64]*/, ge_cached *colc = nullptr,
                            const uint32_t *skip_status = nullptr, const ge_cached *tab_hi = nullptr, uint32_t hi_levels = 2, uint64_t lv_stride = 0) {
    const uint32_t c = tid >> 6, w = tid & 63;
    if (tab_hi) {
        const uint32_t lv = w / (64u / hi_levels);
        if (lv) tab = tab_hi + (uint64_t)(lv - 1) * lv_stride;
    }
    const vb_chunk ch = chunks[c];
    ge_ext acc;
    ge_identity(acc);
    const uint32_t count = (skip_status && skip_status[ch.msm] != 0) ? 0u : ch.count;
    for (uint32_t k = 0; k < count; k++) {
        const uint64_t t = (uint64_t)ch.first + k;
        const uint32_t word = recoded[8 * t + (w >> 3)];
        const int d = (int)((word >> ((w & 7) * 4)) & 15u) - 8;
        if (d != 0) {
            const int a = d < 0 ? -d : d;
            const ge_cached q = tab[8 * t + (a - 1)];
            ge_add_cached(acc, acc, q, d < 0);
        }
    }
    if (colc) {
        ge_cached cc;
        ge_to_cached(cc, acc);
        colc[(uint64_t)ch.msm * 64 + w] = cc;
    } else {
        part[tid] = acc;
    }
}

// ---- stage 3 ---------------------------------------------------------------
// thread = msm * 64 + w ; chunks of one msm are contiguous: [chunk_first[msm], chunk_first[msm+1])
// col    (optional): the column sum as an extended point (input of the one-lane Horner chain)
// colq16 (optional): the same point for the wavefront-cooperative chain (horner_wave.h):
//                    [msm][w][4][8 words] = canonical encodings of (Y-X, Y+X, Z, 2dT), i.e. 16 u16 limbs each
BP_HD void vb_colsum_acc(ge_ext &acc, uint32_t b, uint32_t w, const uint32_t *chunk_first, const ge_ext *part) {
    const uint32_t c0 = chunk_first[b], c1 = chunk_first[b + 1];
    ge_identity(acc);
    if (c1 > c0) acc = part[(uint64_t)c0 * 64 + w];
    for (uint32_t c = c0 + 1; c < c1; c++) {
        const ge_ext q = part[(uint64_t)c * 64 + w];
        ge_add(acc, acc, q);
    }
}
// 32 words: canonical encodings of (Y-X, Y+X, Z, 2dT) = 4 x 16 u16 limbs for the wavefront Horner chain
BP_HD void vb_encode_colq16(uint32_t *o, const ge_ext &acc) {
    ge_cached cc;
    ge_to_cached(cc, acc);
    fe_to_words(o, cc.YmX);
    fe_to_words(o + 8, cc.YpX);
    fe_to_words(o + 16, cc.Z);
    fe_to_words(o + 24, cc.T2d);
}
BP_HD void vb_colsum_thread(uint32_t tid, const uint32_t *chunk_first, const ge_ext *part, ge_ext *col, uint32_t *colq16,
                            ge_cached *colc = nullptr) {
    ge_ext acc;
    vb_colsum_acc(acc, tid >> 6, tid & 63, chunk_first, part);
    if (col) col[tid] = acc;
    if (colq16) vb_encode_colq16(colq16 + (uint64_t)tid * 32, acc);
    if (colc) {
        ge_cached cc;
        ge_to_cached(cc, acc);
        colc[tid] = cc;
    }
}

// ---- stage 4 ---------------------------------------------------------------
// Horner over the 64 column sums of one MSM: acc = sum_w 16^w col[w]
BP_HD void vb_horner_point(ge_ext &acc, const ge_ext *col /*64 entries*/) {
    acc = col[BP_VB_WINDOWS - 1];
    for (int w = BP_VB_WINDOWS - 2; w >= 0; w--) {
        ge_dbl(acc, acc, false);
        ge_dbl(acc, acc, false);
        ge_dbl(acc, acc, false);
        ge_dbl(acc, acc, true);
        const ge_ext q = col[w];
        ge_add(acc, acc, q);
    }
}

// ---- window sums and chain of the wide range-proof chains ---------------------------------------------------------------------
// R5: signed radix 32 (16-entry tables tab[t][16], 51 windows, sc_recode32) instead of radix 16 (8-entry tables, 64 windows).
// thread = msm * windows + w: window sum over the msm's points [k0, U) (all in one lane: a wide chain has lanes enough); written as
// a cached point to colc[msm][w] (stride 64)
template <bool R5>
BP_HD void vb_window_wide_thread(uint32_t tid, uint32_t U, uint32_t k0, const ge_cached *tab, const uint32_t *recoded, ge_cached *colc) {
    const uint32_t nw = R5 ? BP_VB5_WINDOWS : BP_VB_WINDOWS, ne = R5 ? 16 : 8;
    const uint32_t msm = tid / nw, w = tid - msm * nw;
    ge_ext acc;
    ge_identity(acc);
    // software-pipelined without extra registers: the table entry of point k+1 is requested between the two halves of the addition of
    // point k, into the registers the first half has just released
    const uint64_t t0 = (uint64_t)msm * U;
    int d = 0;
    ge_cached q;
    if (k0 < U) {
        d = R5 ? sc_digit32(recoded + 8 * (t0 + k0), w) : sc_digit16(recoded + 8 * (t0 + k0), (int)w);
        const int a = d < 0 ? -d : d;
        q = tab[ne * (t0 + k0) + (a ? a - 1 : 0)];
    }
    for (uint32_t k = k0; k < U; k++) {
        ge_efgh mid;
        const bool on = d != 0;
        if (on) ge_add_cached_front(mid, acc, q, d < 0);
        int dn = 0;
        if (k + 1 < U) {
            dn = R5 ? sc_digit32(recoded + 8 * (t0 + k + 1), w) : sc_digit16(recoded + 8 * (t0 + k + 1), (int)w);
            const int a = dn < 0 ? -dn : dn;
            q = tab[ne * (t0 + k + 1) + (a ? a - 1 : 0)];
        }
        if (on) ge_add_cached_back(acc, mid);
        d = dn;
    }
    ge_cached cc;
    ge_to_cached(cc, acc);
    colc[(uint64_t)msm * 64 + w] = cc;
}
// thread = msm: sum_w radix^w colc[msm][w], plus `extra` (optional: one more cached point per msm at extra[msm * extra_stride] --
// the point whose coefficient is 1)
template <bool R5>
BP_HD void vb_horner_wide_thread(uint32_t b, uint32_t nmsm, const ge_cached *colc, const ge_cached *extra, uint64_t extra_stride, ge_ext *out) {
    if (b >= nmsm) return;
    const int nw = R5 ? BP_VB5_WINDOWS : BP_VB_WINDOWS;
    const ge_cached *col = colc + (uint64_t)b * 64;
    ge_ext acc;
    ge_identity(acc);
    ge_add_cached(acc, acc, col[nw - 1], false);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int w = nw - 2; w >= 0; w--) {
        const ge_cached q = col[w];   // issued early: independent of the doublings
        ge_dbl(acc, acc, false);
        ge_dbl(acc, acc, false);
        ge_dbl(acc, acc, false);
        if (R5) ge_dbl(acc, acc, false);
        ge_dbl(acc, acc, true);
        ge_add_cached(acc, acc, q, false);
    }
    if (extra) ge_add_cached(acc, acc, extra[(uint64_t)b * extra_stride], false);
    out[b] = acc;
}

// thread = msm: the same chain from cached column sums (what launch 3 writes for the quad chain), result to out[msm]
BP_HD void vb_horner_cached_thread(uint32_t b, uint32_t nmsm, const ge_cached *colc, ge_ext *out) {
    if (b >= nmsm) return;
    const ge_cached *col = colc + (uint64_t)b * BP_VB_WINDOWS;
    ge_ext acc;
    ge_identity(acc);
    ge_add_cached(acc, acc, col[BP_VB_WINDOWS - 1], false);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int w = BP_VB_WINDOWS - 2; w >= 0; w--) {
        const ge_cached q = col[w];   // issued early: independent of the doublings
        ge_dbl(acc, acc, false);
        ge_dbl(acc, acc, false);
        ge_dbl(acc, acc, false);
        ge_dbl(acc, acc, true);
        ge_add_cached(acc, acc, q, false);
    }
    out[b] = acc;
}

// thread = msm
// `pre` (optional): Horner results already computed by the wavefront-cooperative kernel
BP_HD void vb_horner_thread(uint32_t b, const ge_ext *col, const ge_ext *pre, const uint32_t *status, uint32_t *out /*[msm][8]*/,
                            ge_ext *out_ext /*optional [msm]*/) {
    ge_ext acc;
    if (pre) acc = pre[b];
    else vb_horner_point(acc, col + (uint64_t)b * 64);
    uint32_t w[8];
    ristretto_compress(w, acc);
    const bool bad = status[b] != 0;
#pragma unroll
    for (int i = 0; i < 8; i++) out[8 * (uint64_t)b + i] = bad ? 0u : w[i];
    if (out_ext) out_ext[b] = acc;
}

}  // namespace bp
#endif
