// k_rpp.hip: kernels of the batched range-proof prover (rp_prover.h).
#include <hip/hip_runtime.h>
#include "kernels.h"

using namespace bp;

// blocks [0, n_b): lane = proof (blinding terms, V rows)  ||  lane = (proof, party, bit)
__global__ void __launch_bounds__(BP_BLOCK) k_rpp_commit1(uint32_t n_b, uint32_t nthreads, rpp_shape sh, const uint64_t *values, const uint8_t *blindings,
                                                           const uint8_t *rng, uint32_t *gen_scalars, uint32_t *party, uint32_t *sL, uint32_t *sR) {
    if (blockIdx.x < n_b) {
        const uint32_t p = blockIdx.x * BP_BLOCK + threadIdx.x;
        if (p < sh.nproofs) rpp_blind_thread(p, sh, values, blindings, rng, gen_scalars, party);
    } else {
        const uint32_t tid = (blockIdx.x - n_b) * BP_BLOCK + threadIdx.x;
        if (tid < nthreads) rpp_bits_thread(tid, sh, values, rng, gen_scalars, sL, sR);
    }
}

__global__ void __launch_bounds__(RP_BLOCK) k_rpp_chal1(rpp_shape sh, const uint32_t *msm_out, uint32_t *ts, uint32_t *fields, uint8_t *proofs,
                                                         uint8_t *commitments) {
    __shared__ uint32_t lds[50 * RP_BLOCK];
    const uint32_t p = blockIdx.x * RP_BLOCK + threadIdx.x;
    kstate st;
    st.w = lds + threadIdx.x;
    st.stride = RP_BLOCK;
    if (p < sh.nproofs) rpp_chal1_thread(p, sh, st, msm_out, ts, fields, proofs, commitments);
}

__global__ void __launch_bounds__(BP_BLOCK) k_rpp_poly(uint32_t nthreads, rpp_shape sh, const uint64_t *values, const uint32_t *fields, const uint32_t *sL,
                                                        const uint32_t *sR, uint32_t *l0, uint32_t *l1, uint32_t *r0, uint32_t *r1, uint32_t *party) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < nthreads) rpp_poly_thread(tid, sh, values, fields, sL, sR, l0, l1, r0, r1, party);
}

__global__ void __launch_bounds__(BP_BLOCK) k_rpp_tcommit(rpp_shape sh, const uint8_t *rng, uint32_t *gen_scalars, uint32_t *party) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < sh.nproofs) rpp_tcommit_thread(p, sh, rng, gen_scalars, party);
}

__global__ void __launch_bounds__(RP_BLOCK) k_rpp_chal2(rpp_shape sh, const uint32_t *msm_out, uint32_t *ts, uint32_t *fields, const uint32_t *party,
                                                         uint32_t *gen_scalars, uint8_t *proofs) {
    __shared__ uint32_t lds[50 * RP_BLOCK];
    const uint32_t p = blockIdx.x * RP_BLOCK + threadIdx.x;
    kstate st;
    st.w = lds + threadIdx.x;
    st.stride = RP_BLOCK;
    if (p < sh.nproofs) rpp_chal2_thread(p, sh, st, msm_out, ts, fields, party, gen_scalars, proofs);
}

__global__ void __launch_bounds__(BP_BLOCK) k_rpp_vectors(uint32_t nthreads, rpp_shape sh, const uint32_t *fields, const uint32_t *l0, const uint32_t *l1,
                                                           const uint32_t *r0, const uint32_t *r1, uint32_t *a_vec, uint32_t *b_vec, uint32_t *Gf, uint32_t *Hf) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < nthreads) rpp_vectors_thread(tid, sh, fields, l0, l1, r0, r1, a_vec, b_vec, Gf, Hf);
}

// dst[t] = src[ids[t]] for 32-byte records (the generator encodings of an (n, m) proof out of the loaded set)
__global__ void __launch_bounds__(BP_BLOCK) k_gather32(uint32_t count, const uint32_t *ids, const uint32_t *src, uint32_t *dst) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint32_t *s = src + 8 * (uint64_t)ids[t];
    for (int q = 0; q < 8; q++) dst[8 * (uint64_t)t + q] = s[q];
}
