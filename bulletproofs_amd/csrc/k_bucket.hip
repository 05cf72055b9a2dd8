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

__global__ void __launch_bounds__(BP_BLOCK) k_bk_accum(uint32_t nthreads, bk_params prm, uint32_t total, const bk_desc *desc, const uint32_t *idx,
                                                        const fb_entry *pts, ge_ext *bsum) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= nthreads) return;
    const uint32_t bw = tid / prm.half, r = tid - bw * prm.half, w = bw % prm.nwin;
    bk_accum_thread(bw, r, prm, desc, idx + (uint64_t)w * total, pts, bsum);
}

// workgroup = (MSM, window): running-sum tree over the window's buckets, then the window sum goes to the MSM's
// radix-16 column sums (colq16[b][64][32 words]) for the Horner chain
template <int LANES>
__global__ void __launch_bounds__(LANES) k_bk_reduce(bk_params prm, const ge_ext *bsum, uint32_t *colq16) {
    __shared__ ge_ext S[LANES], A[LANES];
    const uint32_t bw = blockIdx.x, lane = threadIdx.x;
    bk_reduce_leaf(lane, prm, bsum + (uint64_t)bw * prm.half, S, A);
    __syncthreads();
    uint32_t nodes = LANES, stride = 1, width = prm.half / LANES;
    while (nodes > 1) {
        const uint32_t k = bk_reduce_fanout(nodes), groups = nodes / k;
        if (lane < groups) bk_reduce_node(lane, k, stride, width, S, A);
        __syncthreads();
        nodes = groups;
        stride *= k;
        width *= k;
    }
    if (lane == 0) {
        const uint32_t b = bw / prm.nwin, w = bw - b * prm.nwin;
        const ge_ext sum = A[0];
        bk_emit_columns(w, prm, sum, colq16 + (uint64_t)b * 64 * 32);
    }
}
template __global__ void k_bk_reduce<64>(bk_params, const ge_ext *, uint32_t *);
template __global__ void k_bk_reduce<256>(bk_params, const ge_ext *, uint32_t *);
