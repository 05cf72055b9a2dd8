// k_ipp.hip: HIP kernels of libbpgpu.so (gfx950); thin __global__ wrappers around the per-lane bodies in the headers.
#include <hip/hip_runtime.h>
#include "kernels.h"

using namespace bp;

__global__ void __launch_bounds__(RP_BLOCK) k_ipp_prepare(ipp_shape sh, rp_strobe_init init, const uint8_t *proofs, const uint8_t *Gf,
                                                           const uint8_t *Hf, const uint8_t *P, const uint8_t *Q, const uint8_t *G,
                                                           const uint8_t *H, uint32_t *scalars, uint32_t *points, uint32_t *status) {
    __shared__ uint32_t lds[50 * RP_BLOCK];
    const uint32_t p = blockIdx.x * RP_BLOCK + threadIdx.x;
    kstate st;
    st.w = lds + threadIdx.x;
    st.stride = RP_BLOCK;
    if (p < sh.nproofs) ipp_prepare_thread(p, sh, init, st, proofs, Gf, Hf, P, Q, G, H, scalars, points, status);
}

__global__ void __launch_bounds__(64) k_ipp_verdict(uint32_t n, const uint32_t *status, const uint8_t *msm_status, const uint32_t *msm_out,
                                                     uint8_t *verdict) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) ipp_verdict_thread(p, status, msm_status, msm_out, verdict);
}

// InnerProductProof::verification_scalars alone (ipp.h): front end lane = proof, then one lane per (index, proof) for s_i
__global__ void __launch_bounds__(RP_BLOCK) k_ipp_vs_front(ipp_shape sh, rp_strobe_init init, const uint8_t *proofs, const uint32_t *ts_in, uint32_t *u_sq,
                                                            uint32_t *u_inv_sq, uint32_t *tab, uint32_t *ts_out, uint32_t *status) {
    __shared__ uint32_t lds[50 * RP_BLOCK];
    const uint32_t p = blockIdx.x * RP_BLOCK + threadIdx.x;
    kstate st;
    st.w = lds + threadIdx.x;
    st.stride = RP_BLOCK;
    if (p < sh.nproofs) ipp_vs_front_thread(p, sh, init, st, proofs, ts_in, u_sq, u_inv_sq, tab, ts_out, status);
}
__global__ void __launch_bounds__(64) k_ipp_vs_s(uint32_t nthreads, ipp_shape sh, const uint32_t *tab, const uint32_t *status, uint32_t *s_out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < nthreads) ipp_vs_s_thread(tid, sh, tab, status, s_out);
}

// LinearProof front end (linear.h): same launch shape as k_ipp_prepare -- lane = proof, sponge state in LDS word-major
__global__ void __launch_bounds__(RP_BLOCK) k_lin_prepare(lin_shape sh, rp_strobe_init init, const uint8_t *proofs, const uint8_t *C,
                                                           const uint8_t *bvec, const uint8_t *G, const uint8_t *F, const uint8_t *B,
                                                           uint32_t *scalars, uint32_t *points, uint32_t *status, uint32_t *ts_out,
                                                           uint32_t *gen_sc) {
    __shared__ uint32_t lds[50 * RP_BLOCK];
    const uint32_t p = blockIdx.x * RP_BLOCK + threadIdx.x;
    kstate st;
    st.w = lds + threadIdx.x;
    st.stride = RP_BLOCK;
    if (p < sh.nproofs) lin_prepare_thread(p, sh, init, st, proofs, C, bvec, G, F, B, scalars, points, status, ts_out, gen_sc);
}

// ProofShare::audit_share front end (audit.h): lane = share
__global__ void __launch_bounds__(64) k_aud_prepare(aud_shape sh, const uint32_t *party, const uint8_t *shares, const uint8_t *bit_commitments,
                                                     const uint8_t *poly_commitments, const uint8_t *challenges, const uint32_t *gens, uint32_t *scalars,
                                                     uint32_t *points, uint32_t *status) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < sh.nshares) aud_prepare_thread(s, sh, party, shares, bit_commitments, poly_commitments, challenges, gens, scalars, points, status);
}

__global__ void __launch_bounds__(64) k_aud_verdict(uint32_t n, const uint32_t *status, const uint8_t *msm_status, const uint32_t *msm_out, uint8_t *verdict) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) aud_verdict_thread(s, status, msm_status, msm_out, verdict);
}
