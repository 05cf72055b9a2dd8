// k_rlc.hip: HIP kernels of libbpgpu.so (gfx950); thin __global__ wrappers around the per-lane bodies in the headers.
#include <hip/hip_runtime.h>
#include "kernels.h"

using namespace bp;

// ---- batch combination (rlc.h) --------------------------------------------------------------------------
// sum over the wavefront of a value below 2^28 per lane: four DPP prefix steps inside each row of 16 lanes
// (row sums < 2^32), then the four row totals are read to scalars and added in 64 bits
__device__ __forceinline__ uint64_t wave_sum_u28(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);   // row_shr:1, zero fill
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);   // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);   // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);   // row_shr:8
    return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)x, 15) + (uint32_t)__builtin_amdgcn_readlane((int)x, 31) +
           (uint32_t)__builtin_amdgcn_readlane((int)x, 47) + (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
// add one scalar per lane into the batch accumulator of generator row `row` (ten 64-bit limb sums).
// uniform: all 64 lanes of the wavefront hold contributions to the SAME row -> one atomic per limb per wavefront
__device__ __forceinline__ void rlc_accumulate(unsigned long long *acc, uint32_t row, const sc &v, bool active, bool uniform) {
    uint64_t l[10];
    rlc_limbs(l, v);
    if (uniform) {
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const uint64_t t = wave_sum_u28(active ? (uint32_t)l[i] : 0u);
            if (__lane_id() == 0) atomicAdd(acc + (uint64_t)row * 10 + i, (unsigned long long)t);
        }
    } else if (active) {
#pragma unroll
        for (int i = 0; i < 10; i++) atomicAdd(acc + (uint64_t)row * 10 + i, (unsigned long long)l[i]);
    }
}

// launch 2 of the combined mode: [0, n_win) window sums of the proof-specific points (rejected proofs skipped)
// ||  the weighted generator coefficients, summed over the batch into acc[row][10]
__global__ void __launch_bounds__(BP_BLOCK) k_rlc_stage3(uint32_t n_win, uint32_t nthreads_win, const vb_chunk *chunks, const ge_cached *tab,
                                                          const uint32_t *recoded, ge_ext *part, uint32_t nthreads_exp, rp_shape sh,
                                                          fb_params prm, const uint32_t *fields, const uint32_t *status,
                                                          unsigned long long *acc, int uniform) {
    if (blockIdx.x < n_win) {
        const uint32_t tid = blockIdx.x * BP_BLOCK + threadIdx.x;
        if (tid < nthreads_win) vb_window_thread(tid, chunks, tab, recoded, part, nullptr, status);
        return;
    }
    const uint32_t tid = (blockIdx.x - n_win) * BP_BLOCK + threadIdx.x;
    const bool valid = tid < nthreads_exp;
    const uint32_t B = sh.nproofs;
    const uint32_t t4 = valid ? tid / B : 0, p = valid ? tid - t4 * B : 0;
    sc g[4], h[4];
    for (int j = 0; j < 4; j++) {
        sc_0(g[j]);
        sc_0(h[j]);
    }
    if (valid) rp_expand_b4_thread(tid, sh, prm, fields, nullptr, status, g, h);
#pragma unroll 1
    for (uint32_t j = 0; j < 4; j++) {
        rlc_accumulate(acc, 2 + 4 * t4 + j, g[j], valid, uniform != 0);
        rlc_accumulate(acc, 2 + sh.nm + 4 * t4 + j, h[j], valid, uniform != 0);
    }
    // the B_blinding (row 0) and B (row 1) coefficients were left in the ROW0/ROW1 fields by launch 1; the lanes
    // of the first index group add them (a proof rejected since then contributes nothing)
    const bool row_lane = valid && t4 == 0;
    if (!uniform || t4 == 0) {   // uniform mode: t4 is the same in all 64 lanes, so whole wavefronts take this branch
#pragma unroll 1
        for (uint32_t row = 0; row < 2; row++) {
            sc r;
            sc_0(r);
            if (row_lane && status[p] == 0) rp_load(r, fields, B, RPF_ROW0 + row, p);
            rlc_accumulate(acc, row, r, row_lane, uniform != 0);
        }
    }
}

// (k_rlc_colsum_scalars below runs this as the second role of the last column-sum launch)
// lane g: reduce the accumulated coefficient of generator row g mod l, recode it for the table walk (batch of 1)
// (lane 0 also initialises the small control block of the batch-of-one tail: verdict byte, a zero status word and
// the chunk bounds {0, rows} of the final column sums)
__device__ __forceinline__ void rlc_scalars_lane(uint32_t g, uint32_t n_rows, const unsigned long long *acc, fb_digit *digits, fb_params prm,
                                                 uint32_t *ctl, uint32_t rows) {
    if (g == 0) {
        ctl[0] = 0;
        ctl[1] = 0;
        ctl[2] = 0;
        ctl[3] = rows;
    }
    if (g >= n_rows) return;
    uint64_t a[10];
#pragma unroll
    for (int i = 0; i < 10; i++) a[i] = acc[(uint64_t)g * 10 + i];
    sc v;
    rlc_acc_to_sc(v, a);
    fb_recode(digits + (uint64_t)g * prm.nwin, 1, v.v, prm);
}
// one level of the column-sum tree (blocks [0, n_red))  ||  the combined generator coefficients
__global__ void __launch_bounds__(BP_BLOCK) k_rlc_colsum_scalars(uint32_t n_red, uint32_t nthreads, uint32_t rows_in, uint32_t group,
                                                                  const ge_ext *in, ge_ext *out, uint32_t n_rows, const unsigned long long *acc,
                                                                  fb_digit *digits, fb_params prm, uint32_t *ctl, uint32_t rows_out) {
    if (blockIdx.x < n_red) {
        const uint32_t tid = blockIdx.x * BP_BLOCK + threadIdx.x;
        if (tid < nthreads) fb_reduce_thread(tid, 64, rows_in, group, in, out);
    } else {
        rlc_scalars_lane((blockIdx.x - n_red) * BP_BLOCK + threadIdx.x, n_rows, acc, digits, prm, ctl, rows_out);
    }
}

// bucket variant (bucket.h) of the per-proof terms: blocks [0, n_acc) accumulate the buckets of the ONE multiscalar
// multiplication over all proofs' weighted terms  ||  the combined generator coefficients (as above)
__global__ void __launch_bounds__(BP_BLOCK) k_rlc_accum_scalars(uint32_t n_acc, uint32_t nthreads, bk_params bk, uint32_t total, const bk_desc *desc,
                                                                 const uint32_t *idx, const fb_entry *pts, ge_ext *bsum, uint32_t n_rows,
                                                                 const unsigned long long *acc, fb_digit *digits, fb_params prm, uint32_t *ctl, uint32_t lim) {
    if (blockIdx.x < n_acc) {
        const uint32_t tid = blockIdx.x * BP_BLOCK + threadIdx.x;
        if (tid >= nthreads) return;
        const uint32_t bw = tid / bk.half, r = tid - bw * bk.half;
        bk_accum_thread(bw, r, bk, desc, idx + (uint64_t)bw * total, pts, bsum, lim);   // one MSM: bw = window
    } else {
        rlc_scalars_lane((blockIdx.x - n_acc) * BP_BLOCK + threadIdx.x, n_rows, acc, digits, prm, ctl, 0u);
    }
}
// block 0: the Horner chain over the window sums (radix-16 column sums written by k_bk_tree)  ||  the table walk
// for a batch of one (block -> split as in k_fb_accum with one proof block)
__global__ void __launch_bounds__(FB_BLOCK) k_rlc_stage4b(const uint32_t *colq16, ge_ext *hq, fb_params prm, uint32_t nsplit, uint32_t npairs,
                                                           const uint32_t *gen_ids, const fb_digit *digits, const fb_entry *table, ge_ext *partial) {
    if (blockIdx.x == 0) {
        hw_horner_msm((const uint16_t *)colq16, hq);
        return;
    }
    const uint32_t split = blockIdx.x - 1;
    const uint32_t per = (npairs + nsplit - 1) / nsplit;
    const uint32_t q0 = split * per, q1 = (q0 + per < npairs) ? q0 + per : npairs;
    if (threadIdx.x == 0) fb_accum_thread(0, split, q0 < npairs ? q0 : npairs, q1, prm, 1, gen_ids, digits, table, partial);
}

// Tail of the combined check, ONE wavefront: the 64 lanes add up the batch-of-one table walk's partial points
// (and the Horner result), fold them through LDS, lane 0 tests the identity (WITH_OUT: and encodes R); then
// every proof's verdict is written: the front end's status if set, else 0 when R is the identity, UNDECIDED
// otherwise.  Status words are handed back zeroed.
template <bool WITH_OUT>
__global__ void __launch_bounds__(64) k_rlc_finish(uint32_t nsplit, const ge_ext *hq, const ge_ext *partial, uint32_t nproofs, uint32_t *status,
                                                    uint8_t *verdict, uint8_t *batch_out, rp_seg_tab segs) {
    // segs (bpgpu_pool_rangeproof_submit_rlc_dev: several submitted batches combined by ONE check): every proof's verdict goes to its own
    // batch's buffer, and every batch that asked for it (rp_seg::msm_out doubles as its 33-byte batch_out) gets the chain's result
    __shared__ ge_ext xch[64];
    __shared__ uint32_t res[9];
    const uint32_t lane = threadIdx.x;
    ge_ext acc;
    bool have = false;
    for (uint32_t sp = lane; sp < nsplit; sp += 64) {
        const ge_ext q = partial[sp];
        if (have) ge_add(acc, acc, q);
        else acc = q;
        have = true;
    }
    if (lane == 63) {
        const ge_ext q = hq[0];
        if (have) ge_add(acc, acc, q);
        else acc = q;
        have = true;
    }
    if (!have) ge_identity(acc);
#pragma unroll 1
    for (uint32_t step = 32; step >= 1; step >>= 1) {
        xch[lane] = acc;
        __syncthreads();
        if (lane < step) {
            const ge_ext q = xch[lane + step];
            ge_add(acc, acc, q);
        }
        __syncthreads();
    }
    if (lane == 0) {
        const uint32_t zero = 0;
        uint8_t bv = 0;
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        shared_finish_tail(0, acc, &zero, WITH_OUT ? w : nullptr, &bv);
        res[8] = bv;
        for (int i = 0; i < 8; i++) res[i] = w[i];
    }
    __syncthreads();
    const uint8_t bv = (uint8_t)res[8];
    for (uint32_t p = lane; p < nproofs; p += 64) {
        const uint8_t v = status[p] ? (uint8_t)status[p] : (bv ? (uint8_t)BP_VERDICT_UNDECIDED : (uint8_t)0);
        if (segs.n) {
            const rp_seg sg = rp_seg_lookup(segs, p);
            sg.verdict[p - sg.first] = v;
        } else {
            verdict[p] = v;
        }
        status[p] = 0;
    }
    if (WITH_OUT && lane < 33) {
        const uint8_t b = lane == 0 ? bv : (uint8_t)(res[(lane - 1) >> 2] >> (8 * ((lane - 1) & 3)));
        if (segs.n) {
            if (segs.ext) {
                for (uint32_t i = 0; i < segs.n; i++)
                    if (segs.ext[i].msm_out) ((uint8_t *)segs.ext[i].msm_out)[lane] = b;
            } else {
#pragma unroll
                for (uint32_t i = 0; i < RP_SEG_INLINE; i++)   // (static indices: the argument block is read with scalar loads)
                    if (i < segs.n && segs.in[i].msm_out) ((uint8_t *)segs.in[i].msm_out)[lane] = b;
            }
        } else {
            batch_out[lane] = b;
        }
    }
}

template __global__ void k_rlc_finish<true>(uint32_t, const ge_ext *, const ge_ext *, uint32_t, uint32_t *, uint8_t *, uint8_t *, rp_seg_tab);
template __global__ void k_rlc_finish<false>(uint32_t, const ge_ext *, const ge_ext *, uint32_t, uint32_t *, uint8_t *, uint8_t *, rp_seg_tab);
