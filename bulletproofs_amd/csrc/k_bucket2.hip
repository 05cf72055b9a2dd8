// k_bucket2.hip: kernels of the fused bucket chain (bucket2.h).
#include <hip/hip_runtime.h>
#include "kernels.h"

using namespace bp;

// lane = term.  Three wavefronts per SIMD (<= 168 registers): the decode is one dependent chain of ~265 field operations, a launch
// of 2 081 wavefronts has to be resident at once or it runs in rounds of that chain's latency (the old k_bk_prepare: 327 registers,
// one wavefront per SIMD, three rounds).  The compiler parks ten words across the squaring chain (outside its loops).
__global__ void __launch_bounds__(BP_BLOCK, 3) k_bk2_prepare(uint32_t total, uint32_t nbatch, const uint32_t *msm_first, const uint32_t *scalars,
                                                              const uint32_t *points, fb_entry *pts, uint8_t *dig, uint32_t *status) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < total) bk2_prepare_thread(t, total, nbatch, msm_first, scalars, points, pts, dig, status);
}

// workgroup = (MSM b, window w).  Workgroups are handed to the eight XCDs round-robin (blockIdx.x mod 8); with xcd_map the 32
// windows of one MSM are consecutive workgroups of ONE XCD, so that the MSM's point records (128 B x terms: 266 kB at 2 081) are
// fetched into one L2 once instead of into all eight.  Dynamic LDS: the 16-bit list, 2 bytes per term of the largest MSM.
template <int LANES>
__global__ void __launch_bounds__(LANES, 3) k_bk2_window(uint32_t nmsm, int xcd_map, const uint32_t *msm_first, uint32_t total, const uint8_t *dig,
                                                       const fb_entry *pts, ge_ext *bsum) {
    extern __shared__ __attribute__((aligned(16))) uint16_t s_list[];
    __shared__ uint32_t s_cnt[BK2_HALF], s_off[BK2_HALF + 4], s_tmp[BK2_HALF];
    __shared__ ge_ext s_head[LANES];
    const uint32_t lane = threadIdx.x, x = blockIdx.x;
    uint32_t b, w;
    if (xcd_map) {
        const uint32_t xcd = x & 7u, slot = x >> 3;
        b = (slot / BK2_NWIN) * 8u + xcd;
        w = slot % BK2_NWIN;
    } else {
        b = x / BK2_NWIN;
        w = x % BK2_NWIN;
    }
    bk2_seg sg;
    sg.first = msm_first[b];
    sg.count = msm_first[b + 1] - sg.first;
    sg.lanes = LANES;
    sg.dig_w = dig + (uint64_t)w * total;
    bk2_lds l;
    l.cnt = s_cnt;
    l.off = s_off;
    l.tmp = s_tmp;
    l.list = s_list;
    l.head = s_head;
    bk2_w0(lane, sg, l);
    __syncthreads();
    bk2_w1(lane, sg, l);
    __syncthreads();
    {
        uint32_t *src = s_cnt, *dst = s_tmp;
#pragma unroll 1
        for (uint32_t s = 1; s < BK2_HALF; s <<= 1) {
            bk2_w2_step(lane, s, sg, src, dst);
            __syncthreads();
            uint32_t *t = src;
            src = dst;
            dst = t;
        }
    }
    bk2_w2_fin(lane, sg, l);
    __syncthreads();
    bk2_w3(lane, sg, l);
    __syncthreads();
    ge_ext *bsum_w = bsum + ((uint64_t)b * BK2_NWIN + w) * BK2_HALF;
    bk2_tail tl;
    bk2_w4(lane, sg, l, pts + sg.first, bsum_w, tl);
    __syncthreads();
    bk2_w5(lane, sg, l, bsum_w, tl);
}
template __global__ void k_bk2_window<64>(uint32_t, int, const uint32_t *, uint32_t, const uint8_t *, const fb_entry *, ge_ext *);
template __global__ void k_bk2_window<128>(uint32_t, int, const uint32_t *, uint32_t, const uint8_t *, const fb_entry *, ge_ext *);
template __global__ void k_bk2_window<256>(uint32_t, int, const uint32_t *, uint32_t, const uint8_t *, const fb_entry *, ge_ext *);

// ---- the generator half as one launch (msm_fixed.h: fb_walk_thread) ------------------------------------------------------------
// workgroup = WAVES wavefronts = WAVES slices of the generator terms for one block of 64 MSMs (lane = MSM); blockIdx.x = wg * nblk_p + pblk.
// The wavefronts fold their sums through LDS; wavefront 0 writes partial[wg][p].
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 3) k_fb_walk(fb_params prm, uint32_t nproofs, uint32_t nblk_p, uint32_t nwg, uint32_t n_gen_terms,
                                                         const uint32_t *gen_scalars, const uint32_t *gen_ids, const fb_entry *table, ge_ext *partial,
                                                         uint32_t *status) {
    __shared__ ge_ext xch[(WAVES / 2) * 64];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t pblk = blockIdx.x % nblk_p, wg = blockIdx.x / nblk_p;
    const uint32_t p = pblk * 64 + lane;
    const bool live = p < nproofs;
    ge_ext acc;
    uint32_t g0, g1;
    fb_walk_slice(g0, g1, wg * WAVES + wave, nwg * WAVES, n_gen_terms);
    if (live) fb_walk_thread(acc, p, g0, g1, prm, n_gen_terms, gen_scalars, gen_ids, table, status);
    else ge_identity(acc);
#pragma unroll 1
    for (uint32_t half = WAVES / 2; half >= 1; half >>= 1) {
        fb_walk_fold_store(wave, lane, half, acc, xch);
        __syncthreads();
        fb_walk_fold_add(wave, lane, half, acc, xch);
        __syncthreads();
    }
    if (wave == 0 && live) partial[(uint64_t)wg * nproofs + p] = acc;
}
template __global__ void k_fb_walk<4>(fb_params, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t *, const uint32_t *, const fb_entry *, ge_ext *, uint32_t *);
template __global__ void k_fb_walk<8>(fb_params, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t *, const uint32_t *, const fb_entry *, ge_ext *, uint32_t *);

// few MSMs: lane = slice of the generator terms of ONE MSM; blockIdx.x = wg * nproofs + p; the wavefront folds its 64 sums in six steps
__global__ void __launch_bounds__(64, 3) k_fb_walk1(fb_params prm, uint32_t nproofs, uint32_t nwg, uint32_t n_gen_terms, const uint32_t *gen_scalars,
                                                  const uint32_t *gen_ids, const fb_entry *table, ge_ext *partial, uint32_t *status) {
    __shared__ ge_ext xch[32];
    const uint32_t lane = threadIdx.x, p = blockIdx.x % nproofs, wg = blockIdx.x / nproofs;
    ge_ext acc;
    uint32_t g0, g1;
    fb_walk_slice(g0, g1, wg * 64 + lane, nwg * 64, n_gen_terms);
    fb_walk_thread(acc, p, g0, g1, prm, n_gen_terms, gen_scalars, gen_ids, table, status);
#pragma unroll 1
    for (uint32_t step = 32; step >= 1; step >>= 1) {
        fb_walk1_fold(lane, step, acc, xch, true);
        __syncthreads();
        fb_walk1_fold(lane, step, acc, xch, false);
        __syncthreads();
    }
    if (lane == 0) partial[(uint64_t)wg * nproofs + p] = acc;
}

// ---- the chain's tail (bucket2.h: bk2_tail_*): workgroup = one wavefront = MSM ------------------------------------------------------
__global__ void __launch_bounds__(64) k_msm_tail(uint32_t nmsm, int have_bucket, const ge_ext *gS, const ge_ext *gA, uint32_t npart, const ge_ext *partial,
                                                  const uint32_t *status, uint32_t *out_words, uint8_t *verdict, uint8_t *status_bytes) {
    __shared__ __attribute__((aligned(16))) uint32_t s_colq8[BK2_NWIN * 32];
    __shared__ __attribute__((aligned(16))) uint32_t s_hw[128];
    __shared__ ge_ext s_xch[32];
    __shared__ ge_ext s_fin[2];   // [0] Horner result, [1] the sum that is encoded
    __shared__ fe s_tin, s_raw;
    __shared__ __attribute__((aligned(16))) uint32_t s_tw[8];
    const uint32_t lane = threadIdx.x, b = blockIdx.x;
    if (have_bucket) bk2_tail_t1(lane, b, gS, gA, s_colq8);
    ge_ext acc;
    bk2_tail_t2(lane, b, nmsm, npart, partial, acc);
#pragma unroll 1
    for (uint32_t step = 32; step >= 1; step >>= 1) {
        fb_walk1_fold(lane, step, acc, s_xch, true);
        __syncthreads();
        fb_walk1_fold(lane, step, acc, s_xch, false);
        __syncthreads();
    }
    if (have_bucket) {
        hw_horner8_msm((const uint16_t *)s_colq8, s_hw, &s_fin[0]);
        __syncthreads();
    }
    if (lane == 0) {
        if (have_bucket) {
            const ge_ext h = s_fin[0];
            ge_add(acc, acc, h);
        }
        s_fin[1] = acc;
    }
    __syncthreads();
    if (out_words) {   // the encoding's inverse square root: the wavefront's chain between two steps of lane 0
        if (lane == 0) bk2_tail_t4a(&s_fin[1], &s_tin, s_tw);
        __syncthreads();
        hw_invsqrt_raw_fe((const uint16_t *)s_tw, s_hw, &s_raw);
        __syncthreads();
    }
    if (lane == 0) bk2_tail_t4b(b, &s_fin[1], &s_raw, &s_tin, status, out_words, verdict, status_bytes);
}

// ---- the tail of a narrow chain (bucket2.h: bk2_leafv_thread, bk2_fast_v4): leaves of 4 buckets; workgroup = 256 lanes = MSM -------------
__global__ void __launch_bounds__(BP_BLOCK) k_bk2_leafv(uint32_t nthreads, const ge_ext *bsum, ge_ext *gV) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nthreads) bk2_leafv_thread(t, bsum, gV);
}
// r = the point held by lane (this lane + delta) of the wavefront (its own beyond the last lane)
__device__ __forceinline__ void ge_shfl_down(ge_ext &r, const ge_ext &p, uint32_t delta) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        r.X.v[i] = __shfl_down(p.X.v[i], delta);
        r.Y.v[i] = __shfl_down(p.Y.v[i], delta);
        r.Z.v[i] = __shfl_down(p.Z.v[i], delta);
        r.T.v[i] = __shfl_down(p.T.v[i], delta);
    }
}
__global__ void __launch_bounds__(256) k_msm_tail_fast(uint32_t nmsm, int have_bucket, const ge_ext *gV, uint32_t npart, const ge_ext *partial,
                                                        const uint32_t *status, uint32_t *out_words, uint8_t *verdict, uint8_t *status_bytes) {
    __shared__ __attribute__((aligned(16))) uint32_t s_colq8[BK2_NWIN * 32];
    __shared__ __attribute__((aligned(16))) uint32_t s_hw[128];
    __shared__ ge_ext s_fin[3];   // [0] Horner result, [1] the sum that is encoded, [2] the generator half
    __shared__ fe s_tin, s_raw;
    __shared__ __attribute__((aligned(16))) uint32_t s_tw[8];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, b = blockIdx.x;
    if (have_bucket) {
        const uint32_t w = tid >> 3, l = tid & 7u;
        ge_ext V, q;
        bk2_fast_v4(w, l, b, gV, V);
#pragma unroll 1
        for (uint32_t step = 4; step >= 1; step >>= 1) {   // (every lane adds: the sums of lanes l >= step are not used)
            ge_shfl_down(q, V, step);
            ge_add(V, V, q);
        }
        if (l == 0) vb_encode_colq16(s_colq8 + w * 32, V);
    }
    __syncthreads();
    if (wave == 0 && have_bucket) {
        hw_horner8_msm((const uint16_t *)s_colq8, s_hw, &s_fin[0]);
    } else if (wave == 1) {
        ge_ext acc, q;
        bk2_tail_t2(lane, b, nmsm, npart, partial, acc);
#pragma unroll 1
        for (uint32_t step = 32; step >= 1; step >>= 1) {
            ge_shfl_down(q, acc, step);
            ge_add(acc, acc, q);
        }
        if (lane == 0) s_fin[2] = acc;
    }
    __syncthreads();
    if (tid == 0) {
        ge_ext acc = s_fin[2];
        if (have_bucket) {
            const ge_ext h = s_fin[0];
            ge_add(acc, acc, h);
        }
        s_fin[1] = acc;
        if (out_words) bk2_tail_t4a(&s_fin[1], &s_tin, s_tw);
    }
    __syncthreads();
    if (out_words && wave == 0) hw_invsqrt_raw_fe((const uint16_t *)s_tw, s_hw, &s_raw);
    __syncthreads();
    if (tid == 0) bk2_tail_t4b(b, &s_fin[1], &s_raw, &s_tin, status, out_words, verdict, status_bytes);
}

// The tail of bpgpu_msm_batch_shared in the narrow small-MSM form (k_msm.hip: k_vb_prepare_hi / k_vb_window_hi; option msm_narrow): wavefront 0 adds the chunks' rows and runs the 32-window chain over the caller's own points,
// wavefront 1 adds the generator half's partial sums (k_fb_walk*: one launch on the second stream) with six shuffle steps; then the encoding
__global__ void __launch_bounds__(128) k_shared_tail_narrow(uint32_t nmsm, const uint32_t *chunk_first, const ge_ext *part, uint32_t npart, const ge_ext *partial,
                                                             const uint32_t *status, uint32_t *out_words, uint8_t *verdict, uint8_t *status_bytes, uint32_t levels) {
    __shared__ ge_ext s_fin[3];   // [0] the chain's result, [1] the sum that is encoded, [2] the generator half
    __shared__ fe s_tin, s_raw;
    __shared__ __attribute__((aligned(16))) uint32_t s_tw[8];
    __shared__ __attribute__((aligned(16))) uint32_t s_hw[128];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, b = blockIdx.x;
    if (wave == 0) {
        hw_colsum_horner_msm(b, chunk_first, part, &s_fin[0], (int)levels);
    } else {
        ge_ext acc, q;
        bk2_tail_t2(lane, b, nmsm, npart, partial, acc);
#pragma unroll 1
        for (uint32_t step = 32; step >= 1; step >>= 1) {
            ge_shfl_down(q, acc, step);
            ge_add(acc, acc, q);
        }
        if (lane == 0) s_fin[2] = acc;
    }
    __syncthreads();
    if (tid == 0) {
        ge_ext acc = s_fin[2];
        const ge_ext h = s_fin[0];
        ge_add(acc, acc, h);
        s_fin[1] = acc;
        if (out_words) bk2_tail_t4a(&s_fin[1], &s_tin, s_tw);
    }
    __syncthreads();
    if (out_words && wave == 0) hw_invsqrt_raw_fe((const uint16_t *)s_tw, s_hw, &s_raw);
    __syncthreads();
    if (tid == 0) bk2_tail_t4b(b, &s_fin[1], &s_raw, &s_tin, status, out_words, verdict, status_bytes);
}

