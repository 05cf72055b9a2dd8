// Four-lane Horner chain: ONE QUAD (4 adjacent lanes) PER MSM, one coordinate per lane.
//
// The tail of every MSM,  R = sum_w 16^w C_w  over its 64 column sums, is 252 doublings + 64 additions that
// are sequential per MSM.  horner_wave.h spends a whole wavefront on one chain (lowest latency, but 64 lanes x
// 16-bit limbs cost ~10x the instructions of the plain field arithmetic); with many batches in flight the
// device is bound by total VALU work, not by the latency of one chain, and this layout is the better trade:
// lane r of a quad holds coordinate r of the running point (X, Y, Z, T) in the ordinary 10 x 25.5-bit limbs,
// the four field multiplications of each half of a point operation run in the four lanes at once
// ("parallel formulas", as the AVX2 backend of curve25519-dalek does with 4 x 64-bit SIMD lanes), and
// coordinates move between the lanes with DPP quad_perm moves (one instruction per limb, no LDS).
// One wavefront = 16 chains; ~2 multiplications + ~170 cheap instructions per point operation.
//
// Doubling uses E = 2XY = 2TZ (T Z = X Y for every extended point), so lane 3 needs only Z:
//   (X^2, Y^2, Z^2, T Z)  ->  H = X^2 + Y^2, G = Y^2 - X^2, E = 2 T Z, F = 2 Z^2 - G
//   X3 = E F, Y3 = H G, Z3 = G F, T3 = E H            (same E, F, G, H as ge_dbl in ge25519.h)
// Addition of a cached point (Y2+X2, Y2-X2, Z2, 2d T2), one member per lane:
//   (Y1+X1, Y1-X1, Z1, T1) * cached = (B, A, Z1 Z2, C);  D = 2 Z1 Z2, E = B - A, H = B + A, F = D - C, G = D + C
//
// Written against a tiny "quad value" layer so that the identical code runs on the host (tests/cpu_harness),
// where a quad value is an array of four field elements.
#ifndef BPGPU_HORNER_QUAD_H
#define BPGPU_HORNER_QUAD_H
#include "msm_vb.h"

namespace bp {

#if defined(__HIP_DEVICE_COMPILE__)
typedef fe qfe;   // this lane's field element
#define HQ_FN __device__ __forceinline__
// lane i of every quad receives the value of lane P_i of the same quad
template <int P0, int P1, int P2, int P3>
HQ_FN qfe hq_perm(const qfe &x) {
    qfe r;
#pragma unroll
    for (int i = 0; i < 10; i++)
        r.v[i] = (uint32_t)__builtin_amdgcn_update_dpp((int)x.v[i], (int)x.v[i], P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xf, 0xf, false);
    return r;
}
// lanes whose bit is set in MASK (bit r = lane r of the quad) take a, the others b
template <int MASK>
HQ_FN qfe hq_sel(const qfe &a, const qfe &b) {
    const bool take_a = (MASK >> (__lane_id() & 3)) & 1;
    qfe r;
    fe_select(r, b, a, take_a);
    return r;
}
HQ_FN qfe hq_add(const qfe &a, const qfe &b) { qfe r; fe_add(r, a, b); return r; }
HQ_FN qfe hq_sub(const qfe &a, const qfe &b) { qfe r; fe_sub(r, a, b); return r; }
HQ_FN qfe hq_mul(const qfe &a, const qfe &b) { qfe r; fe_mul(r, a, b); return r; }
#else
struct qfe {
    fe l[4];
};
#define HQ_FN inline
template <int P0, int P1, int P2, int P3>
HQ_FN qfe hq_perm(const qfe &x) {
    qfe r;
    r.l[0] = x.l[P0];
    r.l[1] = x.l[P1];
    r.l[2] = x.l[P2];
    r.l[3] = x.l[P3];
    return r;
}
template <int MASK>
HQ_FN qfe hq_sel(const qfe &a, const qfe &b) {
    qfe r;
    for (int i = 0; i < 4; i++) r.l[i] = ((MASK >> i) & 1) ? a.l[i] : b.l[i];
    return r;
}
HQ_FN qfe hq_add(const qfe &a, const qfe &b) { qfe r; for (int i = 0; i < 4; i++) fe_add(r.l[i], a.l[i], b.l[i]); return r; }
HQ_FN qfe hq_sub(const qfe &a, const qfe &b) { qfe r; for (int i = 0; i < 4; i++) fe_sub(r.l[i], a.l[i], b.l[i]); return r; }
HQ_FN qfe hq_mul(const qfe &a, const qfe &b) {
    qfe r;
    for (int i = 0; i < 4; i++) {
        fe_check_lazy(a.l[i]);
        fe_check_lazy(b.l[i]);
        fe_mul(r.l[i], a.l[i], b.l[i]);
    }
    return r;
}
#endif

// second half shared by doubling and addition: lanes form X3 = E F, Y3 = H G, Z3 = G F, T3 = E H.
// ee must be valid in lanes 0 and 3, ff in lanes 0 and 2, gg in lanes 1 and 2, hh in lanes 1 and 3.
HQ_FN qfe hq_finish(const qfe &ee, const qfe &ff, const qfe &gg, const qfe &hh) {
    const qfe a = hq_sel<0x9>(ee, hq_sel<0x2>(hh, gg));   // (E, H, G, E)
    const qfe b = hq_sel<0x5>(ff, hq_sel<0x2>(gg, hh));   // (F, G, F, H)
    return hq_mul(a, b);
}

// c <- 2 c     (c: reduced limbs)
HQ_FN qfe hq_dbl(const qfe &c) {
    const qfe m = hq_mul(c, hq_perm<0, 1, 2, 2>(c));       // (X^2, Y^2, Z^2, T Z)
    const qfe xx = hq_perm<0, 0, 0, 0>(m), yy = hq_perm<1, 1, 1, 1>(m);
    const qfe hh = hq_add(xx, yy);                          // lazy 2x
    const qfe gg = hq_sub(yy, xx);
    const qfe t1 = hq_perm<3, 3, 3, 3>(m);                  // T Z
    const qfe t2 = hq_perm<2, 2, 2, 2>(m);                  // Z^2
    const qfe ee = hq_add(t1, t1);                          // lazy 2x
    const qfe ff = hq_sub(hq_add(t2, t2), gg);
    return hq_finish(ee, ff, gg, hh);
}

// c <- c + Q,  q = this lane's member of Q as a ge_cached (Y+X | Y-X | Z | 2dT), reduced limbs
HQ_FN qfe hq_add_cached(const qfe &c, const qfe &q) {
    const qfe x = hq_perm<0, 0, 0, 0>(c), y = hq_perm<1, 1, 1, 1>(c);
    const qfe in = hq_sel<0x1>(hq_add(y, x), hq_sel<0x2>(hq_sub(y, x), c));   // (Y+X, Y-X, Z, T)
    const qfe m = hq_mul(in, q);                            // (B, A, Z1 Z2, C)
    const qfe pb = hq_perm<0, 0, 0, 0>(m), pa = hq_perm<1, 1, 1, 1>(m);
    const qfe pz = hq_perm<2, 2, 2, 2>(m), pc = hq_perm<3, 3, 3, 3>(m);
    const qfe dd = hq_add(pz, pz);                          // lazy 2x
    const qfe ee = hq_sub(pb, pa);
    const qfe hh = hq_add(pb, pa);                          // lazy 2x
    const qfe ff = hq_sub(dd, pc);
    const qfe gg = hq_add(dd, pc);                          // lazy 3x
    return hq_finish(ee, ff, gg, hh);
}

// The chain for one MSM; colc = its 64 column sums as cached points ([w] -> 4 field elements).
// load_q(w) must return this lane's coordinate of column sum w.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void hq_horner_msm(uint32_t b, uint32_t nmsm, const ge_cached *colc, ge_ext *out) {
    const uint32_t r = __lane_id() & 3;
    const bool live = b < nmsm;
    const fe *col = (const fe *)(colc + (uint64_t)(live ? b : 0) * BP_VB_WINDOWS) + r;   // coordinate r of window 0
    qfe c;
    fe_0(c);
    c.v[0] = (r == 1 || r == 2) ? 1u : 0u;                 // identity (0 : 1 : 1 : 0)
#pragma unroll 1
    for (int w = BP_VB_WINDOWS - 1; w >= 0; w--) {
        const qfe q = col[4 * w];                           // issued early: independent of the doublings
        if (w != BP_VB_WINDOWS - 1) {
#pragma unroll 1
            for (int i = 0; i < 4; i++) c = hq_dbl(c);
        }
        c = hq_add_cached(c, q);
    }
    if (live) ((fe *)(out + b))[r] = c;
}
#elif !defined(__HIPCC__)
inline void hq_horner_msm(uint32_t b, uint32_t nmsm, const ge_cached *colc, ge_ext *out) {
    if (b >= nmsm) return;
    const fe *col = (const fe *)(colc + (uint64_t)b * BP_VB_WINDOWS);
    qfe c;
    for (int r = 0; r < 4; r++) {
        fe_0(c.l[r]);
        c.l[r].v[0] = (r == 1 || r == 2) ? 1u : 0u;
    }
    for (int w = BP_VB_WINDOWS - 1; w >= 0; w--) {
        qfe q;
        for (int r = 0; r < 4; r++) q.l[r] = col[4 * w + r];
        if (w != BP_VB_WINDOWS - 1)
            for (int i = 0; i < 4; i++) c = hq_dbl(c);
        c = hq_add_cached(c, q);
    }
    for (int r = 0; r < 4; r++) ((fe *)(out + b))[r] = c.l[r];
}
#else
__device__ void hq_horner_msm(uint32_t b, uint32_t nmsm, const ge_cached *colc, ge_ext *out);   // host pass of hipcc
#endif

}  // namespace bp
#endif
