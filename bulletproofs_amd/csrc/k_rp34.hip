// k_rp34.hip: HIP kernels of libbpgpu.so (gfx950); thin __global__ wrappers around the per-lane bodies in the headers.
#include <hip/hip_runtime.h>
#include "kernels.h"

using namespace bp;

// launch 3: [0, n_win) per-chunk window sums of the proof-specific points  ||  the 2nm generator exponents  ||  (narrow chains with
// option coop_split: nthreads_rows = proofs, else 0) the B_blinding / B coefficients, lane = proof
// Launch 3 of a narrow chain under load (the ticket regime: five chains of ~150 proofs in flight) runs 88 us instead of 33; its 260 registers (the
// basepoint-coefficient role's working set) allow one wavefront per SIMD.  Capping it at 256 / 168 registers (two / three wavefronts per SIMD, 8
// registers spilled) was measured: no difference in any call-shape row (profiles/r06/stage3_waves_ab.txt) -- the launches queue behind each other,
// not behind their own register files.  Left uncapped.
#ifndef BP_STAGE3_WAVES
#define BP_STAGE3_WAVES 1
#endif
// FORM: 0 four indices per lane, 1 eight in mirrored pairs (least work), 2 one index per lane (least latency: chains of <= 256 proofs)
template <int FORM>
__global__ void __launch_bounds__(BP_BLOCK, FORM == 1 ? 1 : BP_STAGE3_WAVES) k_rp_stage3(uint32_t n_win, uint32_t nthreads_win, const vb_chunk *chunks, const ge_cached *tab,
                                                         const uint32_t *recoded, ge_ext *part, ge_cached *colc, uint32_t nthreads_exp,
                                                         rp_shape sh, fb_params prm, const uint32_t *fields, fb_digit *digits,
                                                         const uint32_t *status, uint32_t n_exp, uint32_t nthreads_rows, uint32_t lg_m, const ge_cached *tab_hi) {
    if (blockIdx.x < n_win) {
        const uint32_t tid = blockIdx.x * BP_BLOCK + threadIdx.x;
        if (tid < nthreads_win) vb_window_thread(tid, chunks, tab, recoded, part, colc, nullptr, tab_hi, sh.narrow_hi, (uint64_t)8 * sh.nproofs * sh.U);
    } else if (blockIdx.x < n_win + n_exp) {
        const uint32_t tid = (blockIdx.x - n_win) * BP_BLOCK + threadIdx.x;
        if (tid < nthreads_exp) {
            if (FORM == 1) rp_expand_b8_thread(tid, sh, prm, fields, digits, status);
            else if (FORM == 2) rp_expand_b1_thread(tid, sh, prm, fields, digits, status);
            else rp_expand_b4_thread(tid, sh, prm, fields, digits, status);
        }
    } else {
        const uint32_t tid = (blockIdx.x - n_win - n_exp) * BP_BLOCK + threadIdx.x;
        if (tid < nthreads_rows) rp_rows_thread(tid, sh, prm, lg_m, fields, digits, status);
    }
}
template __global__ void k_rp_stage3<0>(uint32_t, uint32_t, const vb_chunk *, const ge_cached *, const uint32_t *, ge_ext *, ge_cached *, uint32_t, rp_shape, fb_params,
                                        const uint32_t *, fb_digit *, const uint32_t *, uint32_t, uint32_t, uint32_t, const ge_cached *);
template __global__ void k_rp_stage3<1>(uint32_t, uint32_t, const vb_chunk *, const ge_cached *, const uint32_t *, ge_ext *, ge_cached *, uint32_t, rp_shape, fb_params,
                                        const uint32_t *, fb_digit *, const uint32_t *, uint32_t, uint32_t, uint32_t, const ge_cached *);
template __global__ void k_rp_stage3<2>(uint32_t, uint32_t, const vb_chunk *, const ge_cached *, const uint32_t *, ge_ext *, ge_cached *, uint32_t, rp_shape, fb_params,
                                        const uint32_t *, fb_digit *, const uint32_t *, uint32_t, uint32_t, uint32_t, const ge_cached *);

// the generator exponents alone (wide chains, where the window sums are a launch of their own): the role's own register allocation.
// Two wavefronts per SIMD although 166 registers would allow three: with three, the kernel itself runs 200 instead of 290 us, but on
// 20 x 1024 bursts the one-lane Horner chains beside it slow down by as much (1.06 -> 1.4 ms) and they are the longer path: -12 %
// (profiles/r04/ab_exponents_*.txt).  Issue priority for the lane-serial roles (s_setprio) was measured too: neutral, not kept.
template <bool PAIRS>
__global__ void __attribute__((amdgpu_waves_per_eu(2, 2))) __launch_bounds__(BP_BLOCK) k_rp_exponents(uint32_t nthreads_exp, rp_shape sh, fb_params prm, const uint32_t *fields,
                                                                                                 fb_digit *digits, const uint32_t *status) {
    const uint32_t tid = blockIdx.x * BP_BLOCK + threadIdx.x;
    if (tid < nthreads_exp) {
        if (PAIRS) rp_expand_b8_thread(tid, sh, prm, fields, digits, status);   // (round 6: mirrored pairs of indices share s_i and s_i^-1)
        else rp_expand_b4_thread(tid, sh, prm, fields, digits, status);
    }
}
template __global__ void k_rp_exponents<false>(uint32_t, rp_shape, fb_params, const uint32_t *, fb_digit *, const uint32_t *);
template __global__ void k_rp_exponents<true>(uint32_t, rp_shape, fb_params, const uint32_t *, fb_digit *, const uint32_t *);
// The paired form at THREE wavefronts per SIMD (168 registers, 27 spilled) for aggregated shapes (nm >= 1024: the role is thousands of indices per
// proof and a larger share of the chain): same-box A/B of the 20-step bursts, profiles/r06/exponent_waves_ab.txt -- m = 32 +1.9 %, m = 16 +0.7 %; the
// single-proof shape loses (bursts 6.25 -> 5.96 M/s: the one-lane Horner chains beside it) and keeps two.
__global__ void __attribute__((amdgpu_waves_per_eu(3, 3))) __launch_bounds__(BP_BLOCK) k_rp_exponents_w3(uint32_t nthreads_exp, rp_shape sh, fb_params prm, const uint32_t *fields,
                                                                                                    fb_digit *digits, const uint32_t *status) {
    const uint32_t tid = blockIdx.x * BP_BLOCK + threadIdx.x;
    if (tid < nthreads_exp) rp_expand_b8_thread(tid, sh, prm, fields, digits, status);
}

// the one-lane Horner chains as their own launch (wide chains: issued on the context's second stream as soon as the window sums
// exist, so that their ~1 ms of dependent instructions run beside the generator exponents and the table walk instead of after them)
__global__ void __launch_bounds__(FB_BLOCK) k_rp_horner1(uint32_t nproofs, const ge_cached *colc, ge_ext *hq) {
    vb_horner_cached_thread(blockIdx.x * FB_BLOCK + threadIdx.x, nproofs, colc, hq);
}

// the same with the point of coefficient 1 (`extra`: A as a cached point, one per proof) added after the chain; R5 = radix-32 column sums
template <bool R5>
__global__ void __launch_bounds__(FB_BLOCK) k_rp_horner_wide(uint32_t nproofs, const ge_cached *colc, const ge_cached *extra, uint32_t extra_stride, ge_ext *hq) {
    vb_horner_wide_thread<R5>(blockIdx.x * FB_BLOCK + threadIdx.x, nproofs, colc, extra, extra_stride, hq);
}
template __global__ void k_rp_horner_wide<false>(uint32_t, const ge_cached *, const ge_cached *, uint32_t, ge_ext *);
template __global__ void k_rp_horner_wide<true>(uint32_t, const ge_cached *, const ge_cached *, uint32_t, ge_ext *);

// launch 4: [0, n_hw) the Horner chains of the proof-specific terms -- HL = lanes per chain.  4: one quad of lanes per proof, 16
// proofs per wavefront, from cached column sums (horner_quad.h); 1: one lane per proof (least work -- about half the quad's
// instructions -- and twice its latency: the form for wide chains, msm_vb.h); 64: one wavefront per proof, which forms
// its column sums itself (horner_wave.h)  ||  the fixed-base table walk (block -> (split, proof block) as in
// k_fb_accum)
// The walk of a NARROW chain (walk_form 1; nsplit a multiple of 64): workgroup L = (proof, group of 64 splits), lane = split -- the 64 partial
// sums of a proof sit in ONE wavefront and are folded through LDS (six levels) while the Horner wavefronts are still running, instead
// of in the finish kernel behind them (8 lanes x 8 partial sums in sequence + three levels: 40 us of a one-proof chain).  With thread =
// proof (the wide form below) a one-proof chain had 64 workgroups of one active lane each.
// The fused finish (walk_form bit 2; verdicts only): every workgroup of launch 4 that has left its result for proof p -- the Horner wavefront and the
// nsplit / 64 walk workgroups -- counts itself in; the one that finds all the others there adds the pieces up, tests for the identity, writes the
// verdict, hands the status word back zeroed and the counter too.  Nobody waits for anybody: no finish launch, nothing to deadlock.
__device__ __forceinline__ void rp_fused_finish(uint32_t p, uint32_t nproofs, uint32_t nparts, const ge_ext *hq, const ge_ext *partial, uint32_t *fin_cnt, uint32_t *status,
                                                uint8_t *verdict, const rp_seg_tab &segs) {
    // (called by ONE lane, after its workgroup's own result is in global memory)
    __threadfence();
    const uint32_t before = atomicAdd(fin_cnt + p, 1u);
    if (before != nparts) return;   // (1 + nparts arrivals per proof)
    __threadfence();
    ge_ext acc = hq[p];
#pragma unroll 1
    for (uint32_t j = 0; j < nparts; j++) {
        const ge_ext q = partial[(uint64_t)j * nproofs + p];
        ge_add(acc, acc, q);
    }
    if (segs.n) {
        const rp_seg sg = rp_seg_lookup(segs, p);
        shared_finish_tail(p, p - sg.first, acc, status, nullptr, sg.verdict);
    } else {
        shared_finish_tail(p, acc, status, nullptr, verdict);
    }
    status[p] = 0;
    fin_cnt[p] = 0;
}
__device__ __forceinline__ void rp_walk_narrow(uint32_t L, fb_params prm, uint32_t nproofs, uint32_t nsplit, uint32_t npairs, const uint32_t *gen_ids,
                                               const fb_digit *digits, const fb_entry *table, ge_ext *partial, bool fused, const ge_ext *hq, uint32_t *fin_cnt,
                                               uint32_t *status, uint8_t *verdict, const rp_seg_tab &segs) {
    __shared__ ge_ext xch[FB_BLOCK];
    const uint32_t nsg = nsplit / FB_BLOCK, p = L / nsg, sg = L - p * nsg, lane = threadIdx.x, split = sg * FB_BLOCK + lane;
    const uint32_t per = (npairs + nsplit - 1) / nsplit;
    const uint32_t q0 = split * per, q1 = (q0 + per < npairs) ? q0 + per : npairs;
    ge_ext acc;
    fb_accum_point(acc, p, q0 < npairs ? q0 : npairs, q1, prm, nproofs, gen_ids, digits, table);
#pragma unroll 1
    for (uint32_t step = FB_BLOCK / 2; step >= 1; step >>= 1) {
        xch[lane] = acc;
        __syncthreads();
        if (lane < step) {
            const ge_ext q = xch[lane + step];
            ge_add(acc, acc, q);
        }
        __syncthreads();
    }
    if (lane == 0) {
        partial[(uint64_t)sg * nproofs + p] = acc;
        if (fused) rp_fused_finish(p, nproofs, nsg, hq, partial, fin_cnt, status, verdict, segs);
    }
}

template <int HL>
__global__ void __launch_bounds__(FB_BLOCK) k_rp_stage4(uint32_t n_hw, const uint32_t *chunk_first, const ge_ext *part, const ge_cached *colc,
                                                         ge_ext *hq, fb_params prm, uint32_t nproofs, uint32_t nblk_p, uint32_t nsplit,
                                                         uint32_t npairs, const uint32_t *gen_ids, const fb_digit *digits,
                                                         const fb_entry *table, ge_ext *partial, uint32_t walk_form, uint32_t *fin_cnt, uint32_t *status,
                                                         uint8_t *verdict, rp_seg_tab segs) {
    if (blockIdx.x < n_hw) {
            if (HL == 4) hq_horner_msm(blockIdx.x * 16 + (threadIdx.x >> 2), nproofs, colc, hq);
        else if (HL == 1) vb_horner_cached_thread(blockIdx.x * FB_BLOCK + threadIdx.x, nproofs, colc, hq);
        else {
            hw_colsum_horner_msm(blockIdx.x, chunk_first, part, hq + blockIdx.x, (walk_form & 8u) ? 4 : ((walk_form & 2u) ? 2 : 1));   // (bit 1 / bit 3: 32 / 16 windows, the upper digits' sums folded in)
            if (walk_form & 4u) {   // (bit 2: the fused finish; four lanes wrote the result's coordinates)
                __threadfence();
                __syncthreads();
                if (threadIdx.x == 0) rp_fused_finish(blockIdx.x, nproofs, nsplit / FB_BLOCK, hq, partial, fin_cnt, status, verdict, segs);
            }
        }
        return;
    }
    if (HL == 64 && (walk_form & 1u)) {
        rp_walk_narrow(blockIdx.x - n_hw, prm, nproofs, nsplit, npairs, gen_ids, digits, table, partial, (walk_form & 4u) != 0, hq, fin_cnt, status, verdict, segs);
        return;
    }
    const uint32_t L = blockIdx.x - n_hw;
    uint32_t split, pblk;
    if ((nsplit & 7) == 0) {
        const uint32_t r = L & 7, rest = L >> 3;
        pblk = rest % nblk_p;
        split = r + 8 * (rest / nblk_p);
    } else {
        pblk = L % nblk_p;
        split = L / nblk_p;
    }
    const uint32_t p = pblk * FB_BLOCK + threadIdx.x;
    const uint32_t per = (npairs + nsplit - 1) / nsplit;
    const uint32_t q0 = split * per, q1 = (q0 + per < npairs) ? q0 + per : npairs;
    if (p < nproofs) fb_accum_thread(p, split, q0 < npairs ? q0 : npairs, q1, prm, nproofs, gen_ids, digits, table, partial);
}

template __global__ void k_rp_stage4<4>(uint32_t, const uint32_t *, const ge_ext *, const ge_cached *, ge_ext *, fb_params, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t *, const fb_digit *, const fb_entry *, ge_ext *, uint32_t, uint32_t *, uint32_t *, uint8_t *, rp_seg_tab);
template __global__ void k_rp_stage4<1>(uint32_t, const uint32_t *, const ge_ext *, const ge_cached *, ge_ext *, fb_params, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t *, const fb_digit *, const fb_entry *, ge_ext *, uint32_t, uint32_t *, uint32_t *, uint8_t *, rp_seg_tab);
template __global__ void k_rp_stage4<64>(uint32_t, const uint32_t *, const ge_ext *, const ge_cached *, ge_ext *, fb_params, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t *, const fb_digit *, const fb_entry *, ge_ext *, uint32_t, uint32_t *, uint32_t *, uint8_t *, rp_seg_tab);
