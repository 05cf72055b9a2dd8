// Wavefront-cooperative Horner chain: ONE WAVEFRONT PER MSM, one 16-bit limb per lane.
//
// The last stage of every MSM,  R = sum_w 16^w C_w  over the 64 column sums of the variable-base
// part, is 252 doublings + 64 additions that are sequential per MSM.  With one lane per MSM a batch
// of 1024 proofs keeps 16 wavefronts busy for ~0.7 ms while 98 % of the chip idles, and because at
// most ~4 kernels run concurrently on the device this chain bounds the batch-1024 throughput.
//
// Layout: the 64 lanes of a wavefront are 4 rows of 16 lanes; row r holds coordinate r of the running
// point (X, Y, Z, T), lane k of a row holds limb k of that coordinate in radix 2^16
// (2^256 = 38 mod p).  A field multiplication is 16 steps of {broadcast limb s of f inside the row
// (through LDS: one store, four ds_read_b128), rotate g by s inside the row (DPP row_ror), multiply-accumulate into a 64-bit column
// sum}; the carry normalisation is two rounds of {split, rotate by one lane, add} -- O(1) depth instead
// of a 16-step carry chain.  The four field multiplications of a point operation run in the four rows
// at once; coordinates move between rows with one LDS store + one ds_read_b128 (a field element is ONE VGPR
// per lane).  A doubling costs 2 multiplication slots (~100 instructions each) instead of 8 x ~200.
//
// Written against wavevec.h, so the identical code runs lane-exact on the host (tests/cpu_harness).
#ifndef BPGPU_HORNER_WAVE_H
#define BPGPU_HORNER_WAVE_H
#include "msm_vb.h"
#include "wavevec.h"

namespace bp {

// ---- radix-2^16 field arithmetic, one limb per lane ---------------------------------------------
// "small" = every limb <= 2^16 + 2^15 (output of hw_mul / hw_norm); hw_mul accepts limbs <= 2^21.

// one carry round: limb k keeps its low 16 bits and receives the overflow of limb k-1
// (limb 15's overflow re-enters limb 0 times 38).  In: limbs < 2^26.  Out: limbs <= 2^16 + 38 * 2^10.
WV_FN wu32 hw_norm(const wu32 &l, const wu32 &k) {
    const wu32 c = wv_row_ror<1>(l >> 16);
    return (l & 0xffffu) + wv_select(k < 1u, c * 38u, c);
}

template <int S>
struct hw_mul_steps {
    static WV_FN void run(wu64 acc[4], const wu32 fl[16], const wu32 &g, const wu32 &g38, const wu32 &k) {
        // limb (k-S) mod 16 of g, x38 when it wrapped: two zero-filling row shifts OR-ed together (no select)
        const wu32 v = wv_row_shr0<S>(g) | wv_row_shl0<16 - S>(g38);
        acc[S & 3] = wv_mad64(acc[S & 3], fl[S], v);                                          // four independent accumulation chains
        hw_mul_steps<S + 1>::run(acc, fl, g, g38, k);
    }
};
template <>
struct hw_mul_steps<16> {
    static WV_FN void run(wu64 *, const wu32 *, const wu32 &, const wu32 &, const wu32 &) {}
};

// column k = sum_{i+j=k} f_i g_j + 38 sum_{i+j=k+16} f_i g_j ; inputs <= 2^21 -> column < 2^52
// The 16 limbs of f reach every lane of the row through LDS (wv_row_gather16); g is rotated with DPP.
WV_FN wu32 hw_mul(const wv_ctx &cx, const wu32 &f, const wu32 &g, const wu32 &k) {
    wu32 fl[16];
    wv_row_gather16(cx, f, fl);
    wu64 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = wv_widen(wv_splat(0));
    hw_mul_steps<0>::run(acc, fl, g, g * 38u, k);
    const wu64 sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    const wu32 lo = wv_lo32(sum), hi = wv_hi32(sum);          // hi < 2^20
    const wu32 b1 = wv_row_ror<1>(lo >> 16), b2 = wv_row_ror<2>(hi);
    const wu32 l = (lo & 0xffffu) + wv_select(k < 1u, b1 * 38u, b1) + wv_select(k < 2u, b2 * 38u, b2);   // < 2^26
    return hw_norm(l, k);
}

// a - b + 8p, then one carry round.  b limbs must be <= 2^18 - 8; a limbs <= 2^21.
WV_FN wu32 hw_sub(const wu32 &a, const wu32 &b, const wu32 &k) {
    // 8p, limb-wise: 8 * (0xffed, 0xffff x 14, 0x7fff)
    const wu32 bias = wv_select(k == 0u, wv_splat(0x7ff68u), wv_select(k == 15u, wv_splat(0x3fff8u), wv_splat(0x7fff8u)));
    return hw_norm(a + bias - b, k);
}

// second half shared by doubling and addition: every row holds e, f, g, h; row r forms
// X3 = e*f (r=0), Y3 = h*g (r=1), Z3 = g*f (r=2), T3 = e*h (r=3)
WV_FN wu32 hw_point_finish(const wv_ctx &cx, const wu32 &e, const wu32 &f, const wu32 &g, const wu32 &h, const wu32 &row, const wu32 &k) {
    const wu32 a = wv_select(row == 0u || row == 3u, e, wv_select(row == 1u, h, g));
    const wu32 b = wv_select(row == 0u || row == 2u, f, wv_select(row == 1u, g, h));
    return hw_mul(cx, a, b, k);
}

// c (small) <- 2 * point
WV_FN wu32 hw_dbl(const wv_ctx &cx, const wu32 &c, const wu32 &row, const wu32 &k) {
    wu32 r4[4];
    wv_rows4(cx, c, r4);                                      // X, Y, Z, T limbs k in every row
    const wu32 in = wv_select(row == 3u, r4[0] + r4[1], c);   // rows: X, Y, Z, X+Y
    const wu32 sq = hw_mul(cx, in, in, k);                    // XX, YY, ZZ, (X+Y)^2
    wv_rows4(cx, sq, r4);
    const wu32 xx = r4[0], yy = r4[1], zz = r4[2], s = r4[3];
    const wu32 h = yy + xx;                                   // <= 2^17.6
    const wu32 g = hw_sub(yy, xx, k);
    const wu32 e = hw_sub(s, h, k);
    const wu32 f = hw_sub(zz + zz, g, k);
    return hw_point_finish(cx, e, f, g, h, row, k);
}

// c (small) <- point + Q, q = this lane's limb of Q in cached row order (Y-X | Y+X | Z | 2dT), canonical limbs
WV_FN wu32 hw_add_cached(const wv_ctx &cx, const wu32 &c, const wu32 &q, const wu32 &row, const wu32 &k) {
    wu32 r4[4];
    wv_rows4(cx, c, r4);
    const wu32 x = r4[0], y = r4[1];
    const wu32 in = wv_select(row == 0u, hw_sub(y, x, k), wv_select(row == 1u, y + x, c));
    const wu32 m = hw_mul(cx, in, q, k);                      // A | B | Z1*Z2 | C
    wv_rows4(cx, m, r4);
    const wu32 a = r4[0], b = r4[1], cc = r4[3];
    const wu32 d = r4[2] + r4[2];
    const wu32 e = hw_sub(b, a, k);
    const wu32 h = b + a;
    const wu32 f = hw_sub(d, cc, k);
    const wu32 g = d + cc;                                    // <= 2^18.2
    return hw_point_finish(cx, e, f, g, h, row, k);
}

// The whole chain for one MSM.  colq16: this MSM's 64 column sums, [w][4][16] u16 limbs of the
// canonical encodings of (Y-X, Y+X, Z, 2dT).  Returns the running point (row r = coordinate r).
// nwin: 64, or 32 for a very narrow chain whose points come with a second table of their 2^128 multiples (hw_point_shift: the upper 32
// digits of a coefficient then select from that table and are added into the columns of the lower 32 -- half the dependent doublings)
WV_FN wu32 hw_horner(const wv_ctx &cx, const uint16_t *colq16, int nwin = BP_VB_WINDOWS) {
    const wu32 lane = wv_lane();
    const wu32 k = lane & 15u, row = lane >> 4;
    // identity (0 : 1 : 1 : 0)
    wu32 c = wv_select((k == 0u) && (row == 1u || row == 2u), wv_splat(1), wv_splat(0));
    for (int w = nwin - 1; w >= 0; w--) {
        if (w != nwin - 1) {
            c = hw_dbl(cx, c, row, k);
            c = hw_dbl(cx, c, row, k);
            c = hw_dbl(cx, c, row, k);
            c = hw_dbl(cx, c, row, k);
        }
        const wu32 q = wv_load_u16(colq16, lane + (uint32_t)(w * 64));
        c = hw_add_cached(cx, c, q, row, k);
    }
    return c;
}

// The same chain over 32 window sums of 8 bits (the fused bucket chain, bucket2.h): colq8[w][4][16] u16 limbs, eight doublings
// between additions -- 32 additions instead of the 64 of the radix-16 form, half of which added the identity.
WV_FN wu32 hw_horner8(const wv_ctx &cx, const uint16_t *colq8) {
    const wu32 lane = wv_lane();
    const wu32 k = lane & 15u, row = lane >> 4;
    wu32 c = wv_select((k == 0u) && (row == 1u || row == 2u), wv_splat(1), wv_splat(0));
    for (int w = 31; w >= 0; w--) {
        if (w != 31) {
            for (int i = 0; i < 8; i++) c = hw_dbl(cx, c, row, k);
        }
        const wu32 q = wv_load_u16(colq8, lane + (uint32_t)(w * 64));
        c = hw_add_cached(cx, c, q, row, k);
    }
    return c;
}

// 2^n * point: c holds limb k of coordinate `row` (X, Y, Z; row 3 is not read by a doubling)
WV_FN wu32 hw_shift(const wv_ctx &cx, wu32 c, int n) {
    const wu32 lane = wv_lane();
    const wu32 k = lane & 15u, row = lane >> 4;
    for (int i = 0; i < n; i++) c = hw_dbl(cx, c, row, k);
    return c;
}

// the running point in cached row order (Y-X | Y+X | Z | 2dT), "small" limbs: one multiplication slot -- rows 0..2 by 1 (which normalises
// them), row 3 by 2d.  d2l: this lane's limb of 2d.
WV_FN wu32 hw_to_cached(const wv_ctx &cx, const wu32 &c, const wu32 &d2l, const wu32 &row, const wu32 &k) {
    wu32 r4[4];
    wv_rows4(cx, c, r4);
    const wu32 x = r4[0], y = r4[1];
    const wu32 in = wv_select(row == 0u, hw_sub(y, x, k), wv_select(row == 1u, y + x, c));
    const wu32 f = wv_select(row == 3u, d2l, wv_select(k == 0u, wv_splat(1), wv_splat(0)));
    return hw_mul(cx, in, f, k);
}

// ---- the inverse-square-root chain with a wavefront per field element ---------------------------------------------------------------
// ge25519.h: fe_invsqrt_raw -- r = t^3 (t^7)^((p-5)/8), 254 squarings + 14 multiplications in sequence -- is what a lone MSM's canonical
// encoding waits for at the end of its chain: 143 us in one lane (a lane alone on its SIMD issues an instruction every ~9 cycles).  With
// one 16-bit limb per lane a multiplication is 16 multiply-accumulate steps instead of 100 (hw_mul): the same chain in ~50 us.  All four
// rows compute the same value (the layout is the Horner chain's; a row per coordinate has nothing to do here).
WV_FN wu32 hw_sqn(const wv_ctx &cx, wu32 x, int n, const wu32 &k) {
    for (int i = 0; i < n; i++) x = hw_mul(cx, x, x, k);
    return x;
}
// t: canonical 16-bit limbs (or "small"); returns "small" limbs
WV_FN wu32 hw_invsqrt_raw(const wv_ctx &cx, const wu32 &t, const wu32 &k) {
    const wu32 v3 = hw_mul(cx, hw_mul(cx, t, t, k), t, k);
    const wu32 z = hw_mul(cx, hw_mul(cx, v3, v3, k), t, k);   // t^7
    // z^(2^250 - 1): fe25519.h fe_pow2_250m1, step for step
    wu32 t0 = hw_mul(cx, z, z, k);
    wu32 t1 = hw_sqn(cx, t0, 2, k);
    t1 = hw_mul(cx, z, t1, k);
    t0 = hw_mul(cx, t0, t1, k);
    t0 = hw_mul(cx, t0, t0, k);
    t0 = hw_mul(cx, t1, t0, k);
    t1 = hw_sqn(cx, t0, 5, k);
    t0 = hw_mul(cx, t1, t0, k);
    t1 = hw_sqn(cx, t0, 10, k);
    t1 = hw_mul(cx, t1, t0, k);
    wu32 t2 = hw_sqn(cx, t1, 20, k);
    t1 = hw_mul(cx, t2, t1, k);
    t1 = hw_sqn(cx, t1, 10, k);
    t0 = hw_mul(cx, t1, t0, k);
    t1 = hw_sqn(cx, t0, 50, k);
    t1 = hw_mul(cx, t1, t0, k);
    t2 = hw_sqn(cx, t1, 100, k);
    t1 = hw_mul(cx, t2, t1, k);
    t1 = hw_sqn(cx, t1, 50, k);
    wu32 r = hw_mul(cx, t1, t0, k);
    r = hw_sqn(cx, r, 2, k);
    r = hw_mul(cx, r, z, k);                                  // z^((p-5)/8)
    return hw_mul(cx, r, v3, k);
}

// one lane's 16 lazy limbs (<= 2^17) -> 10 x 25.5-bit field element (lazy, limb 0 may exceed 2^26 by 19+38*small)
BP_HD void hw_limbs_to_fe(fe &out, const uint32_t l[16]) {
    uint32_t t[16], carry = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t s = l[i] + carry;
        t[i] = s & 0xffffu;
        carry = s >> 16;
    }
    // carry * 2^256 = carry * 38: fold into the low limbs (carry <= 2)
    uint32_t s = t[0] + 38u * carry;
    t[0] = s & 0xffffu;
    carry = s >> 16;
#pragma unroll
    for (int i = 1; i < 16; i++) {
        const uint32_t u = t[i] + carry;
        t[i] = u & 0xffffu;
        carry = u >> 16;
    }
    // (a second overflow is impossible: the value is now < 2^256)
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = t[2 * i] | (t[2 * i + 1] << 16);
    const uint32_t top = w[7] >> 31;        // 2^255 = 19
    w[7] &= 0x7fffffffu;
    fe_from_words(out, w);
    out.v[0] += 19u * top;
}

// Driver for one MSM (= one wavefront on the device, one lockstep emulation on the host):
// runs the chain and writes the result as an extended point (4 field elements, lazy limbs).
#if defined(__HIPCC__) && !defined(__HIP_DEVICE_COMPILE__)
__device__ void hw_horner_msm(const uint16_t *colq16, ge_ext *out);   // host pass of hipcc: declarations only
__device__ void hw_horner8_msm(const uint16_t *colq8, uint32_t *lds128, ge_ext *out);
__device__ void hw_invsqrt_raw_fe(const uint16_t *t16, uint32_t *lds128, fe *out);
__device__ void hw_colsum_horner_msm(uint32_t b, const uint32_t *chunk_first, const ge_ext *part, ge_ext *out, int nlev = 1);
__device__ void hw_point_shift(const ge_ext &p, int n, ge_ext *out);
__device__ void hw_ristretto_decode(ge_ext &r, const uint32_t w[8]);
__device__ void hw_shift_table8(const ge_ext &p, int n, ge_cached *out8);
#elif defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void hw_horner_msm(const uint16_t *colq16, ge_ext *out) {
    __shared__ __attribute__((aligned(16))) uint32_t hw_lds[128];
    wv_ctx cx;
    cx.lds = hw_lds;
    const wu32 c = hw_horner(cx, colq16);
    const uint32_t lane = wv_lane();
    uint32_t limbs[16];
    wv_row_gather16(cx, c, limbs);
    if ((lane & 15u) == 0) {
        fe r;
        hw_limbs_to_fe(r, limbs);
        ((fe *)out)[lane >> 4] = r;
    }
}
// Fused variant: the wavefront first forms its MSM's 64 column sums itself (lane w = window w: add the
// chunks' partial sums, re-encode as 16-bit limbs into LDS) and then runs the chain from LDS -- one launch
// and no [msm][64][128 B] round trip through HBM.
// nlev = 2 / 4 (very narrow chains: tables of the 2^128, or 2^64 / 2^128 / 2^192, multiples of every point): 64 / nlev columns -- lane w < 64 / nlev
// adds the sums of lanes w + 64 / nlev, ... (the upper digits' window sums, formed from those tables) to its own before the chain, which then
// has 64 / nlev windows
__device__ __forceinline__ void hw_colsum_horner_msm(uint32_t b, const uint32_t *chunk_first, const ge_ext *part, ge_ext *out, int nlev = 1) {
    __shared__ __attribute__((aligned(16))) uint32_t hw_lds2[128];
    __shared__ __attribute__((aligned(16))) uint32_t hw_colq[64 * 32];
    const uint32_t lane = wv_lane(), wl = 64u / (uint32_t)nlev;
    {
        ge_ext acc;
        vb_colsum_acc(acc, b, lane, chunk_first, part);
        if (nlev > 1) {   // (wavefront-uniform)
            ge_ext *x = (ge_ext *)hw_colq;   // <= 48 x 160 bytes of the 8 KB, overwritten by the encodings below
            if (lane >= wl) x[lane - wl] = acc;
            WV_LDS_ORDER();
            if (lane < wl) {
#pragma unroll 1
                for (uint32_t j = 0; j + 1 < (uint32_t)nlev; j++) {
                    const ge_ext q = x[lane + j * wl];
                    ge_add(acc, acc, q);
                }
            }
            WV_LDS_ORDER();
        }
        if (lane < wl) vb_encode_colq16(hw_colq + lane * 32, acc);
    }
    WV_LDS_ORDER();
    wv_ctx cx;
    cx.lds = hw_lds2;
    const wu32 c = hw_horner(cx, (const uint16_t *)hw_colq, (int)wl);
    uint32_t limbs[16];
    wv_row_gather16(cx, c, limbs);
    if ((lane & 15u) == 0) {
        fe r;
        hw_limbs_to_fe(r, limbs);
        ((fe *)out)[lane >> 4] = r;
    }
}
// q = 2^n p with the whole wavefront (every lane holds p; q is written to `out`, LDS or global, by four lanes -- one coordinate each)
__device__ __forceinline__ void hw_point_shift(const ge_ext &p, int n, ge_ext *out) {
    __shared__ __attribute__((aligned(16))) uint32_t hw_lds3[128];
    __shared__ __attribute__((aligned(16))) uint32_t hw_pt[32];
    const uint32_t lane = wv_lane(), k = lane & 15u, row = lane >> 4;
    if (k == 0) {
        uint32_t w[8];
        fe sel;
#pragma unroll
        for (int i = 0; i < 10; i++) sel.v[i] = row == 0 ? p.X.v[i] : (row == 1 ? p.Y.v[i] : (row == 2 ? p.Z.v[i] : p.T.v[i]));   // (limb-wise: no indexed copy of a struct)
        fe_to_words(w, sel);
#pragma unroll
        for (int i = 0; i < 8; i++) hw_pt[8 * row + i] = w[i];
    }
    WV_LDS_ORDER();
    wv_ctx cx;
    cx.lds = hw_lds3;
    const wu32 c = hw_shift(cx, wv_load_u16((const uint16_t *)hw_pt, lane), n);
    uint32_t limbs[16];
    wv_row_gather16(cx, c, limbs);
    if (k == 0) {
        fe r;
        hw_limbs_to_fe(r, limbs);
        ((fe *)out)[row] = r;
    }
}
// out8[e] = (e + 1) 2^n p as cached points, e = 0 .. 7 (what vb_build_table makes of 2^n p in one lane: ~25 us; here ~10), the whole
// chain in the wavefront's layout -- no trip through one lane between the doublings and the table
__device__ __forceinline__ void hw_store_cached(const wv_ctx &cx, const wu32 &ce, ge_cached *out, uint32_t row, uint32_t k) {
    uint32_t limbs[16];
    wv_row_gather16(cx, ce, limbs);
    if (k == 0) {
        fe r;
        hw_limbs_to_fe(r, limbs);
        fe_carry(r);   // (stored operands are reduced: ge_add_cached subtracts them without a carry chain)
        if (row == 0) out->YmX = r;
        else if (row == 1) out->YpX = r;
        else if (row == 2) out->Z = r;
        else out->T2d = r;
    }
}
__device__ __forceinline__ void hw_shift_table8(const ge_ext &p, int n, ge_cached *out8) {
    __shared__ __attribute__((aligned(16))) uint32_t hw_lds4[128];
    __shared__ __attribute__((aligned(16))) uint32_t hw_pt4[40];
    const uint32_t lane = wv_lane(), k = lane & 15u, row = lane >> 4;
    if (k == 0) {
        uint32_t w[8];
        fe sel;
        const fe d2 = BP_FE_D2;
#pragma unroll
        for (int i = 0; i < 10; i++) sel.v[i] = row == 0 ? p.X.v[i] : (row == 1 ? p.Y.v[i] : (row == 2 ? p.Z.v[i] : d2.v[i]));   // (row 3 of a doubling's input is not read: it carries 2d to LDS)
        fe_to_words(w, sel);
#pragma unroll
        for (int i = 0; i < 8; i++) hw_pt4[8 * row + i] = w[i];
    }
    WV_LDS_ORDER();
    wv_ctx cx;
    cx.lds = hw_lds4;
    const wu32 d2l = wv_load_u16((const uint16_t *)hw_pt4, 48u + k);
    wu32 cur = hw_shift(cx, wv_load_u16((const uint16_t *)hw_pt4, lane), n);
    const wu32 c1 = hw_to_cached(cx, cur, d2l, row, k);
    hw_store_cached(cx, c1, out8, row, k);
#pragma unroll 1
    for (int e = 1; e < 8; e++) {
        cur = hw_add_cached(cx, cur, c1, row, k);
        const wu32 ce = hw_to_cached(cx, cur, d2l, row, k);
        hw_store_cached(cx, ce, out8 + e, row, k);
    }
}
// the 8-bit-window chain: colq8 in LDS (or global memory), scratch `lds128` = 128 words of LDS owned by the wavefront
__device__ __forceinline__ void hw_horner8_msm(const uint16_t *colq8, uint32_t *lds128, ge_ext *out) {
    wv_ctx cx;
    cx.lds = lds128;
    const wu32 c = hw_horner8(cx, colq8);
    const uint32_t lane = wv_lane();
    uint32_t limbs[16];
    wv_row_gather16(cx, c, limbs);
    if ((lane & 15u) == 0) {
        fe r;
        hw_limbs_to_fe(r, limbs);
        ((fe *)out)[lane >> 4] = r;
    }
}
// t16: the 16 canonical limbs of t (8 words of fe_to_words) in LDS; out (LDS): fe_invsqrt_raw(t), written by lane 0
__device__ __forceinline__ void hw_invsqrt_raw_fe(const uint16_t *t16, uint32_t *lds128, fe *out) {
    wv_ctx cx;
    cx.lds = lds128;
    const uint32_t lane = wv_lane(), k = lane & 15u;
    const wu32 r = hw_invsqrt_raw(cx, wv_load_u16(t16, k), k);
    uint32_t limbs[16];
    wv_row_gather16(cx, r, limbs);
    if (lane == 0) {
        fe o;
        hw_limbs_to_fe(o, limbs);
        *out = o;
    }
}
// RFC 9496 decode with the whole wavefront: every lane forms the cheap ends (s, u1, u2, v and t = v u2^2 before; the sign fix-ups, x, y
// after), the 254-squaring chain between them runs one limb per lane (~57 us instead of ~120 in one lane).  The point only -- whether
// the encoding was valid is the decode role's business (rp_points_thread reports it); every lane returns the same r.
__device__ __forceinline__ void hw_ristretto_decode(ge_ext &r, const uint32_t w[8]) {
    __shared__ __attribute__((aligned(16))) uint32_t hd_l128[128];
    __shared__ __attribute__((aligned(16))) uint32_t hd_tw[8];
    __shared__ fe hd_raw;
    fe s, u1, u2, v, u2s, tin;
    ristretto_decode_front(s, u1, u2, v, w);
    fe_sq(u2s, u2);
    fe_mul(tin, v, u2s);
    if (wv_lane() == 0) {
        uint32_t tw[8];
        fe_to_words(tw, tin);
#pragma unroll
        for (int i = 0; i < 8; i++) hd_tw[i] = tw[i];
    }
    WV_LDS_ORDER();
    hw_invsqrt_raw_fe((const uint16_t *)hd_tw, hd_l128, &hd_raw);
    WV_LDS_ORDER();
    const fe raw = hd_raw;
    fe I, Dx, Dy, t;
    (void)fe_invsqrt_fix(I, raw, tin);
    fe_mul(Dx, I, u2);
    fe_mul(Dy, I, Dx);
    fe_mul(Dy, Dy, v);
    fe_add(t, s, s);
    fe_mul(r.X, t, Dx);
    fe_abs(r.X);
    fe_mul(r.Y, u1, Dy);
    fe_1(r.Z);
    fe_mul(r.T, r.X, r.Y);
}
#else
inline void hw_colsum_horner_msm(uint32_t b, const uint32_t *chunk_first, const ge_ext *part, ge_ext *out, int nlev = 1);
inline void hw_point_shift(const ge_ext &p, int n, ge_ext *out) {
    uint32_t pt[32];
    fe_to_words(pt, p.X); fe_to_words(pt + 8, p.Y); fe_to_words(pt + 16, p.Z); fe_to_words(pt + 24, p.T);
    wv_ctx cx{0};
    const wu32 c = hw_shift(cx, wv_load_u16((const uint16_t *)pt, wv_lane()), n);
    wu32 limbs[16];
    wv_row_gather16(cx, c, limbs);
    for (int row = 0; row < 4; row++) {
        uint32_t l[16];
        for (int i = 0; i < 16; i++) l[i] = limbs[i].l[row * 16];
        fe r;
        hw_limbs_to_fe(r, l);
        ((fe *)out)[row] = r;
    }
}
inline void hw_invsqrt_raw_fe(const uint16_t *t16, uint32_t *, fe *out) {
    wv_ctx cx{0};
    const wu32 lane = wv_lane(), k = lane & 15u;
    const wu32 r = hw_invsqrt_raw(cx, wv_load_u16(t16, k), k);
    wu32 limbs[16];
    wv_row_gather16(cx, r, limbs);
    uint32_t l[16];
    for (int i = 0; i < 16; i++) l[i] = limbs[i].l[0];
    hw_limbs_to_fe(*out, l);
}
inline void hw_shift_table8(const ge_ext &p, int n, ge_cached *out8) {
    uint32_t pt[32];
    const fe d2 = BP_FE_D2;
    fe_to_words(pt, p.X); fe_to_words(pt + 8, p.Y); fe_to_words(pt + 16, p.Z); fe_to_words(pt + 24, d2);
    wv_ctx cx{0};
    const wu32 lane = wv_lane(), k = lane & 15u, row = lane >> 4;
    const wu32 d2l = wv_load_u16((const uint16_t *)pt, k + 48u);
    wu32 cur = hw_shift(cx, wv_load_u16((const uint16_t *)pt, lane), n);
    const wu32 c1 = hw_to_cached(cx, cur, d2l, row, k);
    for (int e = 0; e < 8; e++) {
        if (e) cur = hw_add_cached(cx, cur, c1, row, k);
        const wu32 ce = e ? hw_to_cached(cx, cur, d2l, row, k) : c1;
        wu32 limbs[16];
        wv_row_gather16(cx, ce, limbs);
        for (int r = 0; r < 4; r++) {
            uint32_t l[16];
            for (int i = 0; i < 16; i++) l[i] = limbs[i].l[r * 16];
            fe f;
            hw_limbs_to_fe(f, l);
            fe_carry(f);
            if (r == 0) out8[e].YmX = f;
            else if (r == 1) out8[e].YpX = f;
            else if (r == 2) out8[e].Z = f;
            else out8[e].T2d = f;
        }
    }
}
inline void hw_ristretto_decode(ge_ext &r, const uint32_t w[8]) {
    fe s, u1, u2, v, u2s, tin, raw;
    ristretto_decode_front(s, u1, u2, v, w);
    fe_sq(u2s, u2);
    fe_mul(tin, v, u2s);
    uint32_t tw[8];
    fe_to_words(tw, tin);
    hw_invsqrt_raw_fe((const uint16_t *)tw, nullptr, &raw);
    fe I, Dx, Dy, t;
    (void)fe_invsqrt_fix(I, raw, tin);
    fe_mul(Dx, I, u2);
    fe_mul(Dy, I, Dx);
    fe_mul(Dy, Dy, v);
    fe_add(t, s, s);
    fe_mul(r.X, t, Dx);
    fe_abs(r.X);
    fe_mul(r.Y, u1, Dy);
    fe_1(r.Z);
    fe_mul(r.T, r.X, r.Y);
}
inline void hw_horner8_msm(const uint16_t *colq8, uint32_t *, ge_ext *out) {
    wv_ctx cx{0};
    const wu32 c = hw_horner8(cx, colq8);
    wu32 limbs[16];
    wv_row_gather16(cx, c, limbs);
    for (int row = 0; row < 4; row++) {
        uint32_t l[16];
        for (int i = 0; i < 16; i++) l[i] = limbs[i].l[row * 16];
        fe r;
        hw_limbs_to_fe(r, l);
        ((fe *)out)[row] = r;
    }
}
inline void hw_horner_msm(const uint16_t *colq16, ge_ext *out, int nwin = BP_VB_WINDOWS) {
    wv_ctx cx{0};
    const wu32 c = hw_horner(cx, colq16, nwin);
    wu32 limbs[16];
    wv_row_gather16(cx, c, limbs);
    for (int row = 0; row < 4; row++) {
        uint32_t l[16];
        for (int i = 0; i < 16; i++) l[i] = limbs[i].l[row * 16];
        fe r;
        hw_limbs_to_fe(r, l);
        ((fe *)out)[row] = r;
    }
}
inline void hw_colsum_horner_msm(uint32_t b, const uint32_t *chunk_first, const ge_ext *part, ge_ext *out, int nlev) {
    uint32_t colq[64 * 32];
    const uint32_t wl = 64u / (uint32_t)nlev;
    for (uint32_t w = 0; w < wl; w++) {
        ge_ext acc;
        vb_colsum_acc(acc, b, w, chunk_first, part);
        for (uint32_t j = 1; j < (uint32_t)nlev; j++) {
            ge_ext hi;
            vb_colsum_acc(hi, b, w + j * wl, chunk_first, part);
            ge_add(acc, acc, hi);
        }
        vb_encode_colq16(colq + w * 32, acc);
    }
    hw_horner_msm((const uint16_t *)colq, out, (int)wl);
}
#endif

}  // namespace bp
#endif
