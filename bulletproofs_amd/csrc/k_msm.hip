// k_msm.hip: HIP kernels of libbpgpu.so (gfx950); thin __global__ wrappers around the per-lane bodies in the headers.
#include <hip/hip_runtime.h>
#include "kernels.h"

using namespace bp;

__global__ void __launch_bounds__(BP_BLOCK) k_vb_prepare(uint32_t total, const vb_chunk *chunks, const uint32_t *term_chunk,
                                                          const uint32_t *scalars, const uint32_t *points, ge_cached *tab,
                                                          uint32_t *recoded, uint32_t *status) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < total) vb_prepare_thread(t, chunks, term_chunk, scalars, points, tab, recoded, status);
}

__global__ void __launch_bounds__(BP_BLOCK) k_vb_window(uint32_t nthreads, const vb_chunk *chunks, const ge_cached *tab,
                                                         const uint32_t *recoded, ge_ext *part) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < nthreads) vb_window_thread(tid, chunks, tab, recoded, part);
}

// ---- a FEW small MSMs per call (option msm_narrow; <= 16 MSMs, <= 768 terms: optional_multiscalar_mul in its Straus-size regime, mod.rs:421,
// ipp.rs:308, one call from one thread): three launches instead of six, the playbook of the narrow range-proof chain --
//   1  lane = term: decode + 8-entry table (vb_prepare_thread)  ||  one WAVEFRONT per term: wavefront-cooperative decode, 128 cooperative
//      doublings, the table of Q = 2^128 P (hw_ristretto_decode, hw_shift_table8)
//   2  window sums in chunks of ~sqrt(N) terms; windows 32 .. 63 select from Q's tables
//   3  one wavefront per MSM: adds the chunks' rows, folds window w + 32 into window w, a 32-window Horner chain (124 dependent doublings
//      instead of 252), then the encoding with its inverse square root as a wavefront chain, the status byte
// (levels = 4, calls of <= 256 terms: three such wavefronts per term -- 2^64 P, 2^128 P, 2^192 P -- and a 16-window chain)
__global__ void __launch_bounds__(BP_BLOCK) k_vb_prepare_hi(uint32_t total, uint32_t n_lane_blocks, const vb_chunk *chunks, const uint32_t *term_chunk,
                                                             const uint32_t *scalars, const uint32_t *points, ge_cached *tab, uint32_t *recoded,
                                                             uint32_t *status, ge_cached *tab_hi, uint32_t levels) {
    if (blockIdx.x < n_lane_blocks) {
        const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
        if (t < total) vb_prepare_thread(t, chunks, term_chunk, scalars, points, tab, recoded, status);
    } else {
        const uint32_t idx = blockIdx.x - n_lane_blocks, lv = idx / total + 1, t = idx - (lv - 1) * total;
        uint32_t pw[8];
#pragma unroll
        for (int i = 0; i < 8; i++) pw[i] = points[8 * (uint64_t)t + i];
        ge_ext pt;
        hw_ristretto_decode(pt, pw);   // (an undecodable point is reported by the lane role; its tables are never used)
        hw_shift_table8(pt, (int)(lv * (256u / levels)), tab_hi + 8 * ((uint64_t)(lv - 1) * total + t));
    }
}
__global__ void __launch_bounds__(BP_BLOCK) k_vb_window_hi(uint32_t nthreads, const vb_chunk *chunks, const ge_cached *tab, const uint32_t *recoded, ge_ext *part,
                                                            const ge_cached *tab_hi, uint32_t levels, uint32_t total) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < nthreads) vb_window_thread(tid, chunks, tab, recoded, part, nullptr, nullptr, tab_hi, levels, (uint64_t)8 * total);
}
__global__ void __launch_bounds__(64) k_vb_tail_narrow(uint32_t nbatch, const uint32_t *chunk_first, const ge_ext *part, const uint32_t *status, uint32_t *out_words,
                                                        uint8_t *status_bytes, uint32_t levels) {
    __shared__ ge_ext s_fin;
    __shared__ fe s_tin, s_raw;
    __shared__ __attribute__((aligned(16))) uint32_t s_tw[8];
    __shared__ __attribute__((aligned(16))) uint32_t s_hw[128];
    const uint32_t b = blockIdx.x;
    hw_colsum_horner_msm(b, chunk_first, part, &s_fin, (int)levels);
    __syncthreads();
    if (threadIdx.x == 0) bk2_tail_t4a(&s_fin, &s_tin, s_tw);
    __syncthreads();
    hw_invsqrt_raw_fe((const uint16_t *)s_tw, s_hw, &s_raw);
    __syncthreads();
    if (threadIdx.x == 0) bk2_tail_t4b(b, &s_fin, &s_raw, &s_tin, status, out_words, nullptr, status_bytes);
}

// the window-sum role of launch 3 as a launch of its own (experiment "split_stage3": its own register budget -- 128 VGPRs, four
// wavefronts per SIMD -- instead of the 252 of the role-fused kernel, whose generator-exponent role sets the allocation)
__global__ void __launch_bounds__(BP_BLOCK) k_vb_window_colc(uint32_t nthreads, const vb_chunk *chunks, const ge_cached *tab,
                                                              const uint32_t *recoded, ge_ext *part, ge_cached *colc) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < nthreads) vb_window_thread(tid, chunks, tab, recoded, part, colc);
}

// window sums of a wide range-proof chain (msm_vb.h): lane = (proof, window), points [k0, U) of the proof; R5 = radix 32
template <bool R5>
__global__ void __launch_bounds__(BP_BLOCK) k_vb_window_wide(uint32_t nthreads, uint32_t U, uint32_t k0, const ge_cached *tab, const uint32_t *recoded,
                                                              ge_cached *colc) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < nthreads) vb_window_wide_thread<R5>(tid, U, k0, tab, recoded, colc);
}
template __global__ void k_vb_window_wide<false>(uint32_t, uint32_t, uint32_t, const ge_cached *, const uint32_t *, ge_cached *);
template __global__ void k_vb_window_wide<true>(uint32_t, uint32_t, uint32_t, const ge_cached *, const uint32_t *, ge_cached *);

__global__ void __launch_bounds__(BP_BLOCK) k_vb_colsum(uint32_t nthreads, const uint32_t *chunk_first, const ge_ext *part,
                                                         uint32_t *colq16, ge_cached *colc) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < nthreads) vb_colsum_thread(tid, chunk_first, part, nullptr, colq16, colc);
}

// wavefront-cooperative Horner chain (horner_wave.h): one 64-lane workgroup = one wavefront = one MSM
__global__ void __launch_bounds__(64) k_horner_wave(const uint32_t *colq16, ge_ext *hq) {
    const uint32_t b = blockIdx.x;
    hw_horner_msm((const uint16_t *)(colq16 + (uint64_t)b * 64 * 32), hq + b);
}

__global__ void __launch_bounds__(64) k_vb_horner(uint32_t nbatch, const ge_ext *hq, const uint32_t *status, uint32_t *out) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nbatch) vb_horner_thread(b, nullptr, hq, status, out, nullptr);
}

__global__ void __launch_bounds__(64) k_status_bytes(uint32_t n, const uint32_t *status, uint8_t *out) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n) out[b] = (uint8_t)status[b];
}

__global__ void __launch_bounds__(64) k_fb_base(fb_params prm, const uint32_t *gens, ge_ext *base, uint32_t *bad) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < prm.n_gens) fb_base_thread(g, prm, gens, base, bad);
}

__global__ void __launch_bounds__(64) k_fb_fill(fb_params prm, const ge_ext *base, fb_entry *table) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < prm.n_gens * prm.nwin) fb_fill_thread(tid, prm, base, table);
}

__global__ void __launch_bounds__(64) k_fb_norm(uint64_t n_groups, uint64_t n_entries, fb_entry *table) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < n_groups) fb_norm_thread(gid, n_entries, table);
}

__global__ void __launch_bounds__(BP_BLOCK) k_fb_recode(uint32_t nthreads, fb_params prm, uint32_t nproofs, uint32_t n_gen_terms,
                                                         const uint32_t *gen_scalars, fb_digit *digits, uint32_t *status) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < nthreads) fb_recode_thread(tid, prm, nproofs, n_gen_terms, gen_scalars, digits, status);
}

// grid: 1-D, nblk_p * nsplit blocks of FB_BLOCK lanes (lane = proof).  Block L serves
// split = (L % 8) + 8 * ((L / 8) / nblk_p) so that the blocks the dispatcher places on
// one XCD (L % 8) share the same slices of the table in that XCD's L2.
__global__ void __launch_bounds__(FB_BLOCK) k_fb_accum(fb_params prm, uint32_t nproofs, uint32_t nblk_p, uint32_t nsplit,
                                                        uint32_t npairs, const uint32_t *gen_ids, const fb_digit *digits,
                                                        const fb_entry *table, ge_ext *partial) {
    const uint32_t L = blockIdx.x;
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

// constant-time twins (msm_fixed.h): same grids
__global__ void __launch_bounds__(BP_BLOCK) k_fb_recode_ct(uint32_t nthreads, fb_params prm, uint32_t nproofs, uint32_t n_gen_terms,
                                                            const uint32_t *gen_scalars, fb_digit *digits) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < nthreads) fb_recode_ct_thread(tid, prm, nproofs, n_gen_terms, gen_scalars, digits);
}
__global__ void __launch_bounds__(FB_BLOCK) k_fb_accum_ct(fb_params prm, uint32_t nproofs, uint32_t nblk_p, uint32_t nsplit, uint32_t npairs,
                                                           const uint32_t *__restrict__ gen_ids, const fb_digit *__restrict__ digits,
                                                           const fb_entry *__restrict__ table, ge_ext *__restrict__ partial) {
    const uint32_t pblk = blockIdx.x % nblk_p, split = blockIdx.x / nblk_p;
    const uint32_t p = pblk * FB_BLOCK + threadIdx.x;
    const uint32_t per = (npairs + nsplit - 1) / nsplit;
    const uint32_t q0 = split * per, q1 = (q0 + per < npairs) ? q0 + per : npairs;
    if (p < nproofs) fb_accum_ct_thread(p, split, q0 < npairs ? q0 : npairs, q1, prm, nproofs, gen_ids, digits, table, partial);
}

__global__ void __launch_bounds__(BP_BLOCK) k_fb_reduce(uint32_t nthreads, uint32_t nproofs, uint32_t nsplit, uint32_t group,
                                                         const ge_ext *partial, ge_ext *out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < nthreads) fb_reduce_thread(tid, nproofs, nsplit, group, partial, out);
}

__global__ void __launch_bounds__(64) k_shared_finish(uint32_t nproofs, uint32_t nsplit, const ge_ext *hq, int have_unique,
                                                       const ge_ext *partial, const uint32_t *status, uint32_t *out_words,
                                                       uint8_t *verdict, uint8_t *status_bytes) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nproofs) {
        shared_finish_thread(p, nproofs, nsplit, nullptr, have_unique != 0, hq, partial, status, out_words, verdict);
        if (status_bytes) status_bytes[p] = (uint8_t)status[p];   // (what k_status_bytes did as a launch of its own)
    }
}

// One launch for the whole tail: 8 lanes per proof add up the per-split partial sums (and the Horner result),
// a 3-level exchange through LDS folds them, lane 0 tests / compresses.  Replaces two fb_reduce launches and
// shared_finish; `reset_status` hands the status words back zeroed for the next call on this context.
template <bool WITH_OUT>   // WITH_OUT = false: verdicts only (no compression code, a third of the registers)
__global__ void __launch_bounds__(64) k_finish8(uint32_t nproofs, uint32_t nsplit, const ge_ext *hq, const ge_ext *partial, uint32_t *status,
                                                 uint32_t *out_words, uint8_t *verdict, int reset_status, rp_seg_tab segs) {
    __shared__ ge_ext xch[64];
    const uint32_t lane = threadIdx.x, j = lane & 7, p = blockIdx.x * 8 + (lane >> 3);
    const bool live = p < nproofs;
    ge_ext acc;
    if (live) shared_finish8_gather(acc, p, j, nproofs, nsplit, hq, partial);
    else ge_identity(acc);
#pragma unroll 1
    for (uint32_t step = 4; step >= 1; step >>= 1) {
        xch[lane] = acc;
        __syncthreads();
        if (j < step) {
            const ge_ext q = xch[lane + step];
            ge_add(acc, acc, q);
        }
        __syncthreads();
    }
    if (live && j == 0) {
        if (segs.n) {   // coalesced launch: the verdict goes to the buffers of the item the proof came with
            const rp_seg sg = rp_seg_lookup(segs, p);
            shared_finish_tail(p, p - sg.first, acc, status, WITH_OUT ? sg.msm_out : nullptr, sg.verdict);
        } else {
            shared_finish_tail(p, acc, status, WITH_OUT ? out_words : nullptr, verdict);
        }
        if (reset_status) status[p] = 0;
    }
}

// the tail of a narrow chain whose walk folded its partial sums itself (rp_walk_narrow): lane = proof, Horner result + one partial sum
template <bool WITH_OUT>
__global__ void __launch_bounds__(64) k_finish1(uint32_t nproofs, uint32_t nparts, const ge_ext *hq, const ge_ext *partial, uint32_t *status, uint32_t *out_words,
                                                 uint8_t *verdict, int reset_status, rp_seg_tab segs) {
    const uint32_t p = blockIdx.x * 64 + threadIdx.x;
    if (p >= nproofs) return;
    ge_ext acc = hq[p];
#pragma unroll 1
    for (uint32_t j = 0; j < nparts; j++) {
        const ge_ext q = partial[(uint64_t)j * nproofs + p];
        ge_add(acc, acc, q);
    }
    if (segs.n) {
        const rp_seg sg = rp_seg_lookup(segs, p);
        shared_finish_tail(p, p - sg.first, acc, status, WITH_OUT ? sg.msm_out : nullptr, sg.verdict);
    } else {
        shared_finish_tail(p, acc, status, WITH_OUT ? out_words : nullptr, verdict);
    }
    if (reset_status) status[p] = 0;
}
template __global__ void k_finish1<true>(uint32_t, uint32_t, const ge_ext *, const ge_ext *, uint32_t *, uint32_t *, uint8_t *, int, rp_seg_tab);
template __global__ void k_finish1<false>(uint32_t, uint32_t, const ge_ext *, const ge_ext *, uint32_t *, uint32_t *, uint8_t *, int, rp_seg_tab);

// generator derivation: 64 uniform bytes -> RistrettoPoint::from_uniform_bytes -> encoding
__global__ void __launch_bounds__(64) k_from_uniform(uint32_t n, const uint32_t *uniform, uint32_t *out) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    uint32_t w[16], o[8];
    for (int i = 0; i < 16; i++) w[i] = uniform[16 * (uint64_t)g + i];
    ge_ext r;
    ristretto_from_uniform(r, w);
    ristretto_compress(o, r);
    for (int i = 0; i < 8; i++) out[8 * (uint64_t)g + i] = o[i];
}

template __global__ void k_finish8<true>(uint32_t, uint32_t, const ge_ext *, const ge_ext *, uint32_t *, uint32_t *, uint8_t *, int, rp_seg_tab);
template __global__ void k_finish8<false>(uint32_t, uint32_t, const ge_ext *, const ge_ext *, uint32_t *, uint32_t *, uint8_t *, int, rp_seg_tab);
