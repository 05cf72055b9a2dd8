// k_ippc.hip: kernels of the batched inner-product-proof prover (ipp_prover.h).
#include <hip/hip_runtime.h>
#include "kernels.h"

using namespace bp;

__global__ void __launch_bounds__(BP_BLOCK) k_ippc_init(uint32_t nthreads, ippc_shape sh, const uint8_t *a_in, const uint8_t *b_in, const uint8_t *Gf,
                                                         const uint8_t *Hf, uint32_t *a, uint32_t *b, uint32_t *wG, uint32_t *wH, uint32_t *status) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < nthreads) ippc_init_thread(tid, sh, a_in, b_in, Gf, Hf, a, b, wG, wH, status);
}

// blocks [0, n_q): the Q terms (lane = proof: the two inner products)  ||  the G_t / H_t terms (lane = (proof, t))
__global__ void __launch_bounds__(BP_BLOCK) k_ippc_terms(uint32_t n_q, uint32_t nthreads, ippc_shape sh, uint32_t j, const uint32_t *a, const uint32_t *b,
                                                          const uint32_t *wG, const uint32_t *wH, const uint8_t *G, const uint8_t *H, const uint8_t *Q,
                                                          uint32_t *msm_sc, uint32_t *msm_pt) {
    if (blockIdx.x < n_q) {
        const uint32_t p = blockIdx.x * BP_BLOCK + threadIdx.x;
        if (p < sh.nproofs) ippc_q_thread(p, sh, j, a, b, Q, msm_sc, msm_pt);
    } else {
        const uint32_t tid = (blockIdx.x - n_q) * BP_BLOCK + threadIdx.x;
        if (tid < nthreads) ippc_terms_thread(tid, sh, j, a, b, wG, wH, G, H, msm_sc, msm_pt);
    }
}

// the same for G, H = the context's generators and Q = w B: rows of generator-table scalars instead of (scalar, point) lists
__global__ void __launch_bounds__(BP_BLOCK) k_ippc_terms_fixed(uint32_t n_q, uint32_t nthreads, ippc_shape sh, uint32_t j, const uint32_t *a, const uint32_t *b,
                                                                const uint32_t *wG, const uint32_t *wH, const uint32_t *w_all, uint32_t *gen_scalars) {
    if (blockIdx.x < n_q) {
        const uint32_t p = blockIdx.x * BP_BLOCK + threadIdx.x;
        if (p < sh.nproofs) ippc_q_fixed_thread(p, sh, j, a, b, w_all, gen_scalars);
    } else {
        const uint32_t tid = (blockIdx.x - n_q) * BP_BLOCK + threadIdx.x;
        if (tid < nthreads) ippc_terms_fixed_thread(tid, sh, j, a, b, wG, wH, gen_scalars);
    }
}

__global__ void __launch_bounds__(RP_BLOCK) k_ippc_challenge(ippc_shape sh, uint32_t j, const uint32_t *msm_out, const uint8_t *msm_status, uint32_t *ts,
                                                              uint32_t *u, uint32_t *uinv, uint8_t *proofs, uint32_t proof_len, uint32_t *status) {
    __shared__ uint32_t lds[50 * RP_BLOCK];
    const uint32_t p = blockIdx.x * RP_BLOCK + threadIdx.x;
    kstate st;
    st.w = lds + threadIdx.x;
    st.stride = RP_BLOCK;
    if (p < sh.nproofs) ippc_challenge_thread(p, sh, j, st, msm_out, msm_status, ts, u, uinv, proofs, proof_len, status);
}

__global__ void __launch_bounds__(BP_BLOCK) k_ippc_fold(uint32_t nthreads, ippc_shape sh, uint32_t j, const uint32_t *u, const uint32_t *uinv, uint32_t *a,
                                                         uint32_t *b, uint32_t *wG, uint32_t *wH) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < nthreads) ippc_fold_thread(tid, sh, j, u, uinv, a, b, wG, wH);
}

__global__ void __launch_bounds__(BP_BLOCK) k_ippc_final(ippc_shape sh, const uint32_t *a, const uint32_t *b, uint8_t *proofs, uint32_t proof_len,
                                                          const uint32_t *status, uint8_t *status_out) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= sh.nproofs) return;
    ippc_final_thread(p, sh, a, b, proofs, proof_len);
    if (status_out) status_out[p] = (uint8_t)status[p];
}
