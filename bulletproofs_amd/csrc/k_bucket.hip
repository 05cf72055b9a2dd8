// k_bucket.hip: kernels of the bucket (Pippenger) multiscalar multiplication (bucket.h).
#include <hip/hip_runtime.h>
#include "kernels.h"

using namespace bp;

__global__ void __launch_bounds__(BP_BLOCK) k_bk_prepare(uint32_t total, uint32_t nbatch, const uint32_t *msm_first, const uint32_t *scalars,
                                                          const uint32_t *points, fb_entry *pts, uint32_t *rwords, uint32_t *status, bk_params prm) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < total) bk_prepare_thread(t, nbatch, msm_first, scalars, points, pts, rwords, status, prm);
}

// workgroup = (MSM b, window w), blockIdx.x = b * nwin + w.  single != 0: ONE MSM over all `total` terms (msm_first
// unused) -- the batch-combination mode, where skip_status / skip_div leave the terms of rejected proofs out.
template <int LANES>
__global__ void __launch_bounds__(LANES) k_bk_sort(bk_params prm, const uint32_t *msm_first, uint32_t total, int single,
                                                    const uint32_t *rwords, uint32_t *idx, bk_desc *desc, const uint32_t *skip_status,
                                                    uint32_t skip_div) {
    __shared__ uint32_t s_cnt[2048], s_off[2048], s_part[256], s_hist2[256];
    const uint32_t bw = blockIdx.x, b = bw / prm.nwin, lane = threadIdx.x;
    bk_seg sg;
    sg.w = bw - b * prm.nwin;
    sg.first = single ? 0u : msm_first[b];
    sg.count = single ? total : msm_first[b + 1] - sg.first;
    sg.sub = 0;
    sg.nsub = 1;
    sg.skip_status = skip_status;
    sg.skip_div = skip_div ? skip_div : 1u;
    bk_lds l;
    l.cnt = s_cnt;
    l.off = s_off;
    l.part = s_part;
    l.hist2 = s_hist2;
    uint32_t *idx_w = idx + (uint64_t)sg.w * total;
    bk_sort_p0(lane, prm, l);
    __syncthreads();
    bk_sort_p1(lane, prm, sg, rwords, l);
    __syncthreads();
    bk_sort_p2(lane, prm, l);
    __syncthreads();
    bk_sort_p3(lane, prm, l);
    __syncthreads();
    bk_sort_p4(lane, prm, l);
    __syncthreads();
    bk_sort_p5(lane, l);
    __syncthreads();
    bk_sort_p6(lane, prm, sg, l, desc + (uint64_t)bw * prm.half);
    __syncthreads();
    bk_sort_p7(lane, prm, sg, rwords, l, idx_w);
}
template __global__ void k_bk_sort<64>(bk_params, const uint32_t *, uint32_t, int, const uint32_t *, uint32_t *, bk_desc *, const uint32_t *, uint32_t);
template __global__ void k_bk_sort<256>(bk_params, const uint32_t *, uint32_t, int, const uint32_t *, uint32_t *, bk_desc *, const uint32_t *, uint32_t);

// ---- the same sort for large MSMs, three launches: histogram (nsub workgroups per (MSM, window)), scan, scatter ----
// phase 0: blockIdx.x = bw * nsub + sub: local histogram -> global gcnt[bw][half] (zeroed by the host)
// phase 1: blockIdx.x = bw: scan, population sort, descriptors; cursors -> gcur[bw][half]
// phase 2: blockIdx.x = bw * nsub + sub: scatter through the global cursors
template <int LANES>
__global__ void __launch_bounds__(LANES) k_bk_sort_big(int phase, uint32_t nsub, bk_params prm, const uint32_t *msm_first, uint32_t total, int single,
                                                        const uint32_t *rwords, uint32_t *idx, bk_desc *desc, uint32_t *gcnt, uint32_t *gcur,
                                                        const uint32_t *skip_status, uint32_t skip_div) {
    __shared__ uint32_t s_cnt[2048], s_off[2048], s_part[256], s_hist2[256];
    const uint32_t lane = threadIdx.x;
    const uint32_t bw = phase == 1 ? blockIdx.x : blockIdx.x / nsub, b = bw / prm.nwin;
    bk_seg sg;
    sg.w = bw - b * prm.nwin;
    sg.first = single ? 0u : msm_first[b];
    sg.count = single ? total : msm_first[b + 1] - sg.first;
    sg.sub = phase == 1 ? 0u : blockIdx.x - bw * nsub;
    sg.nsub = phase == 1 ? 1u : nsub;
    sg.skip_status = skip_status;
    sg.skip_div = skip_div ? skip_div : 1u;
    bk_lds l;
    l.cnt = s_cnt;
    l.off = s_off;
    l.part = s_part;
    l.hist2 = s_hist2;
    if (phase == 0) {
        bk_sort_p0(lane, prm, l);
        __syncthreads();
        bk_sort_p1(lane, prm, sg, rwords, l);
        __syncthreads();
        bk_sort_merge(lane, prm, l, gcnt + (uint64_t)bw * prm.half);
    } else if (phase == 1) {
        bk_sort_load(lane, prm, l, gcnt + (uint64_t)bw * prm.half);
        __syncthreads();
        bk_sort_p2(lane, prm, l);
        __syncthreads();
        bk_sort_p3(lane, prm, l);
        __syncthreads();
        bk_sort_p4(lane, prm, l);
        __syncthreads();
        bk_sort_p5(lane, l);
        __syncthreads();
        bk_sort_p6(lane, prm, sg, l, desc + (uint64_t)bw * prm.half);
        __syncthreads();
        bk_sort_publish(lane, prm, l, gcur + (uint64_t)bw * prm.half);
    } else {
        l.cnt = gcur + (uint64_t)bw * prm.half;
        bk_sort_p7(lane, prm, sg, rwords, l, idx + (uint64_t)sg.w * total);
    }
}
template __global__ void k_bk_sort_big<64>(int, uint32_t, bk_params, const uint32_t *, uint32_t, int, const uint32_t *, uint32_t *, bk_desc *, uint32_t *, uint32_t *, const uint32_t *, uint32_t);
template __global__ void k_bk_sort_big<256>(int, uint32_t, bk_params, const uint32_t *, uint32_t, int, const uint32_t *, uint32_t *, bk_desc *, uint32_t *, uint32_t *, const uint32_t *, uint32_t);

__global__ void __launch_bounds__(BP_BLOCK) k_bk_accum(uint32_t nthreads, bk_params prm, uint32_t total, const bk_desc *desc, const uint32_t *idx,
                                                        const fb_entry *pts, ge_ext *bsum, uint32_t lim) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= nthreads) return;
    const uint32_t bw = tid / prm.half, r = tid - bw * prm.half, w = bw % prm.nwin;
    bk_accum_thread(bw, r, prm, desc, idx + (uint64_t)w * total, pts, bsum, lim);
}

// what crowded buckets hold beyond the lanes' share (bucket.h, stage 3b): G wavefronts per (MSM, window), blockIdx.x = bw * G + g
__global__ void __launch_bounds__(64) k_bk_heavy(bk_params prm, uint32_t total, const bk_desc *desc, const uint32_t *idx, const fb_entry *pts, ge_ext *bsum,
                                                  uint32_t lim, uint32_t G) {
    __shared__ uint32_t s_n[1], s_list[BK_HEAVY_MAX];
    __shared__ ge_ext s_xch[64];
    const uint32_t bw = blockIdx.x / G, g = blockIdx.x - bw * G, lane = threadIdx.x, w = bw % prm.nwin;
    bk_heavy_lds l;
    l.n = s_n;
    l.list = s_list;
    l.xch = s_xch;
    bk_heavy_h0(lane, l);
    __syncthreads();
    bk_heavy_h1(lane, bw, g, G, prm, desc, lim, l);
    __syncthreads();
    const uint32_t n = s_n[0] < BK_HEAVY_MAX ? s_n[0] : BK_HEAVY_MAX;
    const uint32_t *idx_w = idx + (uint64_t)w * total;
#pragma unroll 1
    for (uint32_t i = 0; i < n; i++) {
        bk_heavy_h2(lane, bw, i, prm, desc, lim, idx_w, pts, l);
        __syncthreads();
#pragma unroll 1
        for (uint32_t step = 32; step >= 1; step >>= 1) {
            bk_heavy_h3(lane, step, l);
            __syncthreads();
        }
        bk_heavy_h4(lane, bw, i, prm, desc, l, bsum);
        __syncthreads();
    }
}

// bottom level of the running-sum tree: lane = (MSM, window, leaf)
__global__ void __launch_bounds__(BP_BLOCK) k_bk_leaf(uint32_t nthreads, bk_params prm, const ge_ext *bsum, ge_ext *gS, ge_ext *gA) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < nthreads) bk_leaf_thread(tid, prm, bsum, gS, gA);
}

// upper levels, packed: c = 8: 8 leaves per window -> one lane per window (64 windows per wavefront);
// c = 12: 256 leaves per window -> 32 + 4 + 1 nodes, two windows per wavefront, levels exchanged through LDS.
// The root's A is the window sum; it goes to the MSM's radix-16 column sums (colq16[b][64][32 words]).
template <int C>
__global__ void __launch_bounds__(64) k_bk_tree(bk_params prm, uint32_t nbw, const ge_ext *gS, const ge_ext *gA, uint32_t *colq16) {
    const uint32_t lane = threadIdx.x;
    if (C == 8) {
        const uint32_t bw = blockIdx.x * 64 + lane;
        if (bw >= nbw) return;
        ge_ext S, A;
        bk_combine(S, A, gS, gA, bw * 8, 8, 1, 16);
        const uint32_t b = bw / prm.nwin, w = bw - b * prm.nwin;
        bk_emit_columns(w, prm, A, colq16 + (uint64_t)b * 64 * 32);
    } else {
        __shared__ ge_ext lS[64], lA[64];
        const uint32_t wi = lane >> 5, g = lane & 31, bw = blockIdx.x * 2 + wi;
        const bool live = bw < nbw;
        ge_ext S, A;
        if (live) {
            bk_combine(S, A, gS, gA, bw * 256 + g * 8, 8, 1, 8);          // 256 leaves of 8 buckets -> 32 nodes of 64
            lS[lane] = S;
            lA[lane] = A;
        }
        __syncthreads();
        if (live && g < 4) bk_combine(S, A, lS, lA, wi * 32 + g * 8, 8, 1, 64);   // -> 4 nodes of 512
        __syncthreads();
        if (live && g < 4) {
            lS[wi * 32 + g * 8] = S;
            lA[wi * 32 + g * 8] = A;
        }
        __syncthreads();
        if (live && g == 0) {
            bk_combine(S, A, lS, lA, wi * 32, 4, 8, 512);                  // -> the window sum
            const uint32_t b = bw / prm.nwin, w = bw - b * prm.nwin;
            bk_emit_columns(w, prm, A, colq16 + (uint64_t)b * 64 * 32);
        }
    }
}
template __global__ void k_bk_tree<8>(bk_params, uint32_t, const ge_ext *, const ge_ext *, uint32_t *);
template __global__ void k_bk_tree<12>(bk_params, uint32_t, const ge_ext *, const ge_ext *, uint32_t *);
