// k_linc.hip: kernels of the batched LinearProof prover (linear_prover.h).
#include <hip/hip_runtime.h>
#include "kernels.h"

using namespace bp;

__global__ void __launch_bounds__(BP_BLOCK) k_linc_init(uint32_t nthreads, linc_shape sh, const uint8_t *a_in, const uint8_t *b_in, uint32_t *a, uint32_t *b,
                                                         uint32_t *wG, uint32_t *status) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < nthreads) linc_init_thread(tid, sh, a_in, b_in, a, b, wG, status);
}

// lane = proof: public inputs into the transcript (n b_i and n G_i messages: the long, narrow part of a proof), r, the draws
__global__ void __launch_bounds__(RP_BLOCK) k_linc_public(linc_shape sh, const uint8_t *C, const uint8_t *b_in, const uint8_t *G, const uint8_t *F,
                                                           const uint8_t *B, const uint8_t *r_in, const uint8_t *rng, uint32_t *ts, uint32_t *r_out,
                                                           uint32_t *draws, uint32_t *status) {
    __shared__ uint32_t lds[50 * RP_BLOCK];
    const uint32_t p = blockIdx.x * RP_BLOCK + threadIdx.x;
    kstate st;
    st.w = lds + threadIdx.x;
    st.stride = RP_BLOCK;
    if (p < sh.nproofs) linc_public_thread(p, sh, st, C, b_in, G, F, B, r_in, rng, ts, r_out, draws, status);
}

// blocks [0, n_q): the B and F terms (lane = proof: the two inner products)  ||  the G_t terms (lane = (proof, t))
__global__ void __launch_bounds__(BP_BLOCK) k_linc_terms(uint32_t n_q, uint32_t nthreads, linc_shape sh, uint32_t j, const uint32_t *a, const uint32_t *b,
                                                          const uint32_t *wG, const uint32_t *draws, const uint8_t *G, const uint8_t *F, const uint8_t *B,
                                                          uint32_t *msm_sc, uint32_t *msm_pt) {
    if (blockIdx.x < n_q) {
        const uint32_t p = blockIdx.x * BP_BLOCK + threadIdx.x;
        if (p < sh.nproofs) linc_q_thread(p, sh, j, a, b, draws, F, B, msm_sc, msm_pt);
    } else {
        const uint32_t tid = (blockIdx.x - n_q) * BP_BLOCK + threadIdx.x;
        if (tid < nthreads) linc_terms_thread(tid, sh, j, a, wG, G, msm_sc, msm_pt);
    }
}

__global__ void __launch_bounds__(RP_BLOCK) k_linc_challenge(linc_shape sh, uint32_t j, const uint32_t *msm_out, const uint8_t *msm_status, uint32_t *ts,
                                                              const uint32_t *draws, uint32_t *r_io, uint32_t *x, uint32_t *xinv, uint8_t *proofs,
                                                              uint32_t proof_len, uint32_t *status) {
    __shared__ uint32_t lds[50 * RP_BLOCK];
    const uint32_t p = blockIdx.x * RP_BLOCK + threadIdx.x;
    kstate st;
    st.w = lds + threadIdx.x;
    st.stride = RP_BLOCK;
    if (p < sh.nproofs) linc_challenge_thread(p, sh, j, st, msm_out, msm_status, ts, draws, r_io, x, xinv, proofs, proof_len, status);
}

__global__ void __launch_bounds__(BP_BLOCK) k_linc_fold(uint32_t nthreads, linc_shape sh, uint32_t j, const uint32_t *x, const uint32_t *xinv, uint32_t *a,
                                                         uint32_t *b, uint32_t *wG) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < nthreads) linc_fold_thread(tid, sh, j, x, xinv, a, b, wG);
}

// the n + 2 terms of S: blocks [0, n_q): t* B and (s* b_0) F (lane = proof)  ||  (s* wG(t)) G_t (lane = (proof, t))
__global__ void __launch_bounds__(BP_BLOCK) k_linc_sterms(uint32_t n_q, uint32_t nthreads, linc_shape sh, const uint32_t *b, const uint32_t *wG,
                                                           const uint32_t *draws, const uint8_t *G, const uint8_t *F, const uint8_t *B, uint32_t *msm_sc,
                                                           uint32_t *msm_pt) {
    if (blockIdx.x < n_q) {
        const uint32_t p = blockIdx.x * BP_BLOCK + threadIdx.x;
        if (p < sh.nproofs) linc_sq_thread(p, sh, b, draws, F, B, msm_sc, msm_pt);
    } else {
        const uint32_t tid = (blockIdx.x - n_q) * BP_BLOCK + threadIdx.x;
        if (tid < nthreads) linc_sterms_thread(tid, sh, wG, draws, G, msm_sc, msm_pt);
    }
}

__global__ void __launch_bounds__(RP_BLOCK) k_linc_final(linc_shape sh, const uint32_t *msm_out, const uint8_t *msm_status, uint32_t *ts, const uint32_t *a,
                                                          const uint32_t *draws, const uint32_t *r, uint8_t *proofs, uint32_t proof_len, uint32_t *status,
                                                          uint8_t *status_out) {
    __shared__ uint32_t lds[50 * RP_BLOCK];
    const uint32_t p = blockIdx.x * RP_BLOCK + threadIdx.x;
    kstate st;
    st.w = lds + threadIdx.x;
    st.stride = RP_BLOCK;
    if (p >= sh.nproofs) return;
    linc_final_thread(p, sh, st, msm_out, msm_status, ts, a, draws, r, proofs, proof_len, status);
    if (status_out) status_out[p] = (uint8_t)status[p];
}

// generator-table mode (bases = the context's generators): rows of table scalars instead of (scalar, point) lists
__global__ void __launch_bounds__(BP_BLOCK) k_linc_terms_fixed(uint32_t n_q, uint32_t nthreads, linc_shape sh, uint32_t j, const uint32_t *a, const uint32_t *b,
                                                                const uint32_t *wG, const uint32_t *draws, uint32_t *gen_scalars) {
    if (blockIdx.x < n_q) {
        const uint32_t p = blockIdx.x * BP_BLOCK + threadIdx.x;
        if (p < sh.nproofs) linc_q_fixed_thread(p, sh, j, a, b, draws, gen_scalars);
    } else {
        const uint32_t tid = (blockIdx.x - n_q) * BP_BLOCK + threadIdx.x;
        if (tid < nthreads) linc_terms_fixed_thread(tid, sh, j, a, wG, gen_scalars);
    }
}

__global__ void __launch_bounds__(BP_BLOCK) k_linc_sterms_fixed(uint32_t n_q, uint32_t nthreads, linc_shape sh, const uint32_t *b, const uint32_t *wG,
                                                                 const uint32_t *draws, uint32_t *gen_scalars) {
    if (blockIdx.x < n_q) {
        const uint32_t p = blockIdx.x * BP_BLOCK + threadIdx.x;
        if (p < sh.nproofs) linc_sq_fixed_thread(p, sh, b, draws, gen_scalars);
    } else {
        const uint32_t tid = (blockIdx.x - n_q) * BP_BLOCK + threadIdx.x;
        if (tid < nthreads) linc_sterms_fixed_thread(tid, sh, wG, draws, gen_scalars);
    }
}
