// Prototypes of every kernel of libbpgpu.so.  The kernels are defined in k_*.hip (one translation unit per group so
// the library builds in parallel); the host runtime (bpgpu.hip and friends) launches them through these declarations.
#ifndef BPGPU_KERNELS_H
#define BPGPU_KERNELS_H
#include <hip/hip_runtime.h>
#include "msm_fixed.h"
#include "msm_vb.h"
#include "horner_wave.h"
#include "horner_quad.h"
#include "rlc.h"
#include "rangeproof.h"
#include "ipp.h"
#include "linear.h"
#include "audit.h"
#include "bucket.h"
#include "bucket2.h"
#include "ipp_prover.h"
#include "rp_prover.h"
#include "linear_prover.h"

#define BP_BLOCK 64   // one wavefront per workgroup: under contention a CU rarely has room for four waves of one group at once (256: -8% at 48 streams)
#define FB_BLOCK 64
#define RP_BLOCK 64

using namespace bp;

__global__ void k_vb_prepare(uint32_t total, const vb_chunk *chunks, const uint32_t *term_chunk, const uint32_t *scalars, const uint32_t *points, ge_cached *tab, uint32_t *recoded, uint32_t *status);
__global__ void k_vb_window(uint32_t nthreads, const vb_chunk *chunks, const ge_cached *tab, const uint32_t *recoded, ge_ext *part);
__global__ void k_vb_prepare_hi(uint32_t total, uint32_t n_lane_blocks, const vb_chunk *chunks, const uint32_t *term_chunk, const uint32_t *scalars, const uint32_t *points, ge_cached *tab, uint32_t *recoded, uint32_t *status, ge_cached *tab_hi, uint32_t levels);
__global__ void k_vb_window_hi(uint32_t nthreads, const vb_chunk *chunks, const ge_cached *tab, const uint32_t *recoded, ge_ext *part, const ge_cached *tab_hi, uint32_t levels, uint32_t total);
__global__ void k_vb_tail_narrow(uint32_t nbatch, const uint32_t *chunk_first, const ge_ext *part, const uint32_t *status, uint32_t *out_words, uint8_t *status_bytes, uint32_t levels);
__global__ void k_shared_tail_narrow(uint32_t nmsm, const uint32_t *chunk_first, const ge_ext *part, uint32_t npart, const ge_ext *partial, const uint32_t *status, uint32_t *out_words, uint8_t *verdict, uint8_t *status_bytes, uint32_t levels);
__global__ void k_vb_window_colc(uint32_t nthreads, const vb_chunk *chunks, const ge_cached *tab, const uint32_t *recoded, ge_ext *part, ge_cached *colc);
__global__ void k_vb_colsum(uint32_t nthreads, const uint32_t *chunk_first, const ge_ext *part, uint32_t *colq16, ge_cached *colc);
__global__ void k_horner_wave(const uint32_t *colq16, ge_ext *hq);
__global__ void k_vb_horner(uint32_t nbatch, const ge_ext *hq, const uint32_t *status, uint32_t *out);
__global__ void k_status_bytes(uint32_t n, const uint32_t *status, uint8_t *out);
__global__ void k_fb_base(fb_params prm, const uint32_t *gens, ge_ext *base, uint32_t *bad);
__global__ void k_fb_fill(fb_params prm, const ge_ext *base, fb_entry *table);
__global__ void k_fb_norm(uint64_t n_groups, uint64_t n_entries, fb_entry *table);
__global__ void k_fb_recode(uint32_t nthreads, fb_params prm, uint32_t nproofs, uint32_t n_gen_terms, const uint32_t *gen_scalars, fb_digit *digits, uint32_t *status);
__global__ void k_fb_accum(fb_params prm, uint32_t nproofs, uint32_t nblk_p, uint32_t nsplit, uint32_t npairs, const uint32_t *gen_ids, const fb_digit *digits, const fb_entry *table, ge_ext *partial);
__global__ void k_fb_recode_ct(uint32_t nthreads, fb_params prm, uint32_t nproofs, uint32_t n_gen_terms, const uint32_t *gen_scalars, fb_digit *digits);
__global__ void k_fb_accum_ct(fb_params prm, uint32_t nproofs, uint32_t nblk_p, uint32_t nsplit, uint32_t npairs, const uint32_t *gen_ids, const fb_digit *digits, const fb_entry *table, ge_ext *partial);
__global__ void k_fb_reduce(uint32_t nthreads, uint32_t nproofs, uint32_t nsplit, uint32_t group, const ge_ext *partial, ge_ext *out);
__global__ void k_shared_finish(uint32_t nproofs, uint32_t nsplit, const ge_ext *hq, int have_unique, const ge_ext *partial, const uint32_t *status, uint32_t *out_words, uint8_t *verdict,
                                uint8_t *status_bytes);
template <bool WITH_OUT> __global__ void k_finish1(uint32_t nproofs, uint32_t nparts, const ge_ext *hq, const ge_ext *partial, uint32_t *status, uint32_t *out_words, uint8_t *verdict, int reset_status, rp_seg_tab segs);
template <bool WITH_OUT>
__global__ void k_finish8(uint32_t nproofs, uint32_t nsplit, const ge_ext *hq, const ge_ext *partial, uint32_t *status, uint32_t *out_words, uint8_t *verdict, int reset_status, rp_seg_tab segs);
__global__ void k_rp_stage1_coop(rp_shape sh, rp_strobe_init init, uint32_t n_tr, const uint8_t *proofs, const uint8_t *commitments, const uint8_t *rng64, uint32_t *fields, ge_cached *tab, uint32_t *status, fb_params prm, uint32_t lg_m, uint32_t *recoded, fb_digit *digits, const uint8_t *rho64, const uint32_t *ts_in, uint32_t *ts_out, fb_entry *bk_pts, uint32_t bk_c, rp_seg_tab segs, const rp_script_hdr *script, uint32_t n_pt, ge_cached *tab_hi);
template <bool SCRIPTED> __global__ void k_rp_stage1(rp_shape sh, rp_strobe_init init, uint32_t n_tr, const uint8_t *proofs, const uint8_t *commitments, const uint8_t *rng64, uint32_t *fields, ge_cached *tab, uint32_t *status, fb_params prm, uint32_t lg_m, uint32_t *recoded, fb_digit *digits, const uint8_t *rho64, uint32_t ts_flags, const uint32_t *ts_in, uint32_t *ts_out, fb_entry *bk_pts, uint32_t bk_c, rp_seg_tab segs, const rp_script_hdr *script);
template <int FORM>
__global__ void k_rp_stage3(uint32_t n_win, uint32_t nthreads_win, const vb_chunk *chunks, const ge_cached *tab, const uint32_t *recoded, ge_ext *part, ge_cached *colc, uint32_t nthreads_exp, rp_shape sh, fb_params prm, const uint32_t *fields, fb_digit *digits, const uint32_t *status, uint32_t n_exp, uint32_t nthreads_rows, uint32_t lg_m, const ge_cached *tab_hi);
template <bool PAIRS>
__global__ void k_rp_exponents(uint32_t nthreads_exp, rp_shape sh, fb_params prm, const uint32_t *fields, fb_digit *digits, const uint32_t *status);
__global__ void k_rp_exponents_w3(uint32_t nthreads_exp, rp_shape sh, fb_params prm, const uint32_t *fields, fb_digit *digits, const uint32_t *status);
__global__ void k_rp_horner1(uint32_t nproofs, const ge_cached *colc, ge_ext *hq);
template <bool R5>
__global__ void k_rp_horner_wide(uint32_t nproofs, const ge_cached *colc, const ge_cached *extra, uint32_t extra_stride, ge_ext *hq);
template <bool R5>
__global__ void k_vb_window_wide(uint32_t nthreads, uint32_t U, uint32_t k0, const ge_cached *tab, const uint32_t *recoded, ge_cached *colc);
template <int HL>
__global__ void k_rp_stage4(uint32_t n_hw, const uint32_t *chunk_first, const ge_ext *part, const ge_cached *colc, ge_ext *hq, fb_params prm, uint32_t nproofs, uint32_t nblk_p, uint32_t nsplit, uint32_t npairs, const uint32_t *gen_ids, const fb_digit *digits, const fb_entry *table, ge_ext *partial, uint32_t walk_form, uint32_t *fin_cnt, uint32_t *status, uint8_t *verdict, rp_seg_tab segs);
__global__ void k_rlc_stage3(uint32_t n_win, uint32_t nthreads_win, const vb_chunk *chunks, const ge_cached *tab, const uint32_t *recoded, ge_ext *part, uint32_t nthreads_exp, rp_shape sh, fb_params prm, const uint32_t *fields, const uint32_t *status, unsigned long long *acc, int uniform);
__global__ void k_rlc_colsum_scalars(uint32_t n_red, uint32_t nthreads, uint32_t rows_in, uint32_t group, const ge_ext *in, ge_ext *out, uint32_t n_rows, const unsigned long long *acc, fb_digit *digits, fb_params prm, uint32_t *ctl, uint32_t rows_out);
template <bool WITH_OUT>
__global__ void k_rlc_finish(uint32_t nsplit, const ge_ext *hq, const ge_ext *partial, uint32_t nproofs, uint32_t *status, uint8_t *verdict, uint8_t *batch_out, rp_seg_tab segs);
__global__ void k_rp_verdict(uint32_t n, uint32_t *status, const uint8_t *msm_verdict, uint8_t *out);
__global__ void k_ipp_prepare(ipp_shape sh, rp_strobe_init init, const uint8_t *proofs, const uint8_t *Gf, const uint8_t *Hf, const uint8_t *P, const uint8_t *Q, const uint8_t *G, const uint8_t *H, uint32_t *scalars, uint32_t *points, uint32_t *status);
__global__ void k_ipp_vs_front(ipp_shape sh, rp_strobe_init init, const uint8_t *proofs, const uint32_t *ts_in, uint32_t *u_sq, uint32_t *u_inv_sq, uint32_t *tab, uint32_t *ts_out, uint32_t *status);
__global__ void k_ipp_vs_s(uint32_t nthreads, ipp_shape sh, const uint32_t *tab, const uint32_t *status, uint32_t *s_out);
__global__ void k_ipp_verdict(uint32_t n, const uint32_t *status, const uint8_t *msm_status, const uint32_t *msm_out, uint8_t *verdict);
__global__ void k_aud_prepare(aud_shape sh, const uint32_t *party, const uint8_t *shares, const uint8_t *bit_commitments, const uint8_t *poly_commitments, const uint8_t *challenges, const uint32_t *gens, uint32_t *scalars, uint32_t *points, uint32_t *status);
__global__ void k_aud_verdict(uint32_t n, const uint32_t *status, const uint8_t *msm_status, const uint32_t *msm_out, uint8_t *verdict);
__global__ void k_lin_prepare(lin_shape sh, rp_strobe_init init, const uint8_t *proofs, const uint8_t *C, const uint8_t *bvec, const uint8_t *G, const uint8_t *F, const uint8_t *B, uint32_t *scalars, uint32_t *points, uint32_t *status, uint32_t *ts_out, uint32_t *gen_sc);
__global__ void k_from_uniform(uint32_t n, const uint32_t *uniform, uint32_t *out);

// k_rlc.hip, bucket variant of the batch combination
__global__ void k_rlc_accum_scalars(uint32_t n_acc, uint32_t nthreads, bk_params bk, uint32_t total, const bk_desc *desc, const uint32_t *idx, const fb_entry *pts, ge_ext *bsum, uint32_t n_rows, const unsigned long long *acc, fb_digit *digits, fb_params prm, uint32_t *ctl, uint32_t lim);
__global__ void k_rlc_stage4b(const uint32_t *colq16, ge_ext *hq, fb_params prm, uint32_t nsplit, uint32_t npairs, const uint32_t *gen_ids, const fb_digit *digits, const fb_entry *table, ge_ext *partial);
// k_rpp.hip
__global__ void k_rpp_commit1(uint32_t n_b, uint32_t nthreads, rpp_shape sh, const uint64_t *values, const uint8_t *blindings, const uint8_t *rng, uint32_t *gen_scalars, uint32_t *party, uint32_t *sL, uint32_t *sR);
__global__ void k_rpp_chal1(rpp_shape sh, const uint32_t *msm_out, uint32_t *ts, uint32_t *fields, uint8_t *proofs, uint8_t *commitments);
__global__ void k_rpp_poly(uint32_t nthreads, rpp_shape sh, const uint64_t *values, const uint32_t *fields, const uint32_t *sL, const uint32_t *sR, uint32_t *l0, uint32_t *l1, uint32_t *r0, uint32_t *r1, uint32_t *party);
__global__ void k_rpp_tcommit(rpp_shape sh, const uint8_t *rng, uint32_t *gen_scalars, uint32_t *party);
__global__ void k_rpp_chal2(rpp_shape sh, const uint32_t *msm_out, uint32_t *ts, uint32_t *fields, const uint32_t *party, uint32_t *gen_scalars, uint8_t *proofs);
__global__ void k_rpp_vectors(uint32_t nthreads, rpp_shape sh, const uint32_t *fields, const uint32_t *l0, const uint32_t *l1, const uint32_t *r0, const uint32_t *r1, uint32_t *a_vec, uint32_t *b_vec, uint32_t *Gf, uint32_t *Hf);
__global__ void k_gather32(uint32_t count, const uint32_t *ids, const uint32_t *src, uint32_t *dst);
// k_ippc.hip
__global__ void k_ippc_init(uint32_t nthreads, ippc_shape sh, const uint8_t *a_in, const uint8_t *b_in, const uint8_t *Gf, const uint8_t *Hf, uint32_t *a, uint32_t *b, uint32_t *wG, uint32_t *wH, uint32_t *status);
__global__ void k_ippc_terms(uint32_t n_q, uint32_t nthreads, ippc_shape sh, uint32_t j, const uint32_t *a, const uint32_t *b, const uint32_t *wG, const uint32_t *wH, const uint8_t *G, const uint8_t *H, const uint8_t *Q, uint32_t *msm_sc, uint32_t *msm_pt);
__global__ void k_ippc_terms_fixed(uint32_t n_q, uint32_t nthreads, ippc_shape sh, uint32_t j, const uint32_t *a, const uint32_t *b, const uint32_t *wG, const uint32_t *wH, const uint32_t *w_all, uint32_t *gen_scalars);
__global__ void k_ippc_challenge(ippc_shape sh, uint32_t j, const uint32_t *msm_out, const uint8_t *msm_status, uint32_t *ts, uint32_t *u, uint32_t *uinv, uint8_t *proofs, uint32_t proof_len, uint32_t *status);
__global__ void k_ippc_fold(uint32_t nthreads, ippc_shape sh, uint32_t j, const uint32_t *u, const uint32_t *uinv, uint32_t *a, uint32_t *b, uint32_t *wG, uint32_t *wH);
__global__ void k_ippc_final(ippc_shape sh, const uint32_t *a, const uint32_t *b, uint8_t *proofs, uint32_t proof_len, const uint32_t *status, uint8_t *status_out);
// k_bucket.hip
__global__ void k_bk_prepare(uint32_t total, uint32_t nbatch, const uint32_t *msm_first, const uint32_t *scalars, const uint32_t *points, fb_entry *pts, uint32_t *rwords, uint32_t *status, bk_params prm);
template <int LANES>
__global__ void k_bk_sort(bk_params prm, const uint32_t *msm_first, uint32_t total, int single, const uint32_t *rwords, uint32_t *idx, bk_desc *desc, const uint32_t *skip_status, uint32_t skip_div);
template <int LANES>
__global__ void k_bk_sort_big(int phase, uint32_t nsub, bk_params prm, const uint32_t *msm_first, uint32_t total, int single, const uint32_t *rwords, uint32_t *idx, bk_desc *desc, uint32_t *gcnt, uint32_t *gcur, const uint32_t *skip_status, uint32_t skip_div);
__global__ void k_bk_accum(uint32_t nthreads, bk_params prm, uint32_t total, const bk_desc *desc, const uint32_t *idx, const fb_entry *pts, ge_ext *bsum, uint32_t lim);
__global__ void k_bk_heavy(bk_params prm, uint32_t total, const bk_desc *desc, const uint32_t *idx, const fb_entry *pts, ge_ext *bsum, uint32_t lim, uint32_t G);
__global__ void k_bk_leaf(uint32_t nthreads, bk_params prm, const ge_ext *bsum, ge_ext *gS, ge_ext *gA);
template <int C>
__global__ void k_bk_tree(bk_params prm, uint32_t nbw, const ge_ext *gS, const ge_ext *gA, uint32_t *colq16);
// k_bucket2.hip
__global__ void k_bk2_prepare(uint32_t total, uint32_t nbatch, const uint32_t *msm_first, const uint32_t *scalars, const uint32_t *points, fb_entry *pts, uint8_t *dig, uint32_t *status);
template <int LANES>
__global__ void k_bk2_window(uint32_t nmsm, int xcd_map, const uint32_t *msm_first, uint32_t total, const uint8_t *dig, const fb_entry *pts, ge_ext *bsum);
template <int WAVES>
__global__ void k_fb_walk(fb_params prm, uint32_t nproofs, uint32_t nblk_p, uint32_t nwg, uint32_t n_gen_terms, const uint32_t *gen_scalars, const uint32_t *gen_ids, const fb_entry *table, ge_ext *partial, uint32_t *status);
__global__ void k_fb_walk1(fb_params prm, uint32_t nproofs, uint32_t nwg, uint32_t n_gen_terms, const uint32_t *gen_scalars, const uint32_t *gen_ids, const fb_entry *table, ge_ext *partial, uint32_t *status);
__global__ void k_bk2_leafv(uint32_t nthreads, const ge_ext *bsum, ge_ext *gV);
__global__ void k_msm_tail_fast(uint32_t nmsm, int have_bucket, const ge_ext *gV, uint32_t npart, const ge_ext *partial, const uint32_t *status, uint32_t *out_words, uint8_t *verdict, uint8_t *status_bytes);
__global__ void k_msm_tail(uint32_t nmsm, int have_bucket, const ge_ext *gS, const ge_ext *gA, uint32_t npart, const ge_ext *partial, const uint32_t *status, uint32_t *out_words, uint8_t *verdict, uint8_t *status_bytes);

// k_linc.hip
__global__ void k_linc_init(uint32_t nthreads, linc_shape sh, const uint8_t *a_in, const uint8_t *b_in, uint32_t *a, uint32_t *b, uint32_t *wG, uint32_t *status);
__global__ void k_linc_public(linc_shape sh, const uint8_t *C, const uint8_t *b_in, const uint8_t *G, const uint8_t *F, const uint8_t *B, const uint8_t *r_in, const uint8_t *rng, uint32_t *ts, uint32_t *r_out, uint32_t *draws, uint32_t *status);
__global__ void k_linc_terms(uint32_t n_q, uint32_t nthreads, linc_shape sh, uint32_t j, const uint32_t *a, const uint32_t *b, const uint32_t *wG, const uint32_t *draws, const uint8_t *G, const uint8_t *F, const uint8_t *B, uint32_t *msm_sc, uint32_t *msm_pt);
__global__ void k_linc_challenge(linc_shape sh, uint32_t j, const uint32_t *msm_out, const uint8_t *msm_status, uint32_t *ts, const uint32_t *draws, uint32_t *r_io, uint32_t *x, uint32_t *xinv, uint8_t *proofs, uint32_t proof_len, uint32_t *status);
__global__ void k_linc_fold(uint32_t nthreads, linc_shape sh, uint32_t j, const uint32_t *x, const uint32_t *xinv, uint32_t *a, uint32_t *b, uint32_t *wG);
__global__ void k_linc_sterms(uint32_t n_q, uint32_t nthreads, linc_shape sh, const uint32_t *b, const uint32_t *wG, const uint32_t *draws, const uint8_t *G, const uint8_t *F, const uint8_t *B, uint32_t *msm_sc, uint32_t *msm_pt);
__global__ void k_linc_terms_fixed(uint32_t n_q, uint32_t nthreads, linc_shape sh, uint32_t j, const uint32_t *a, const uint32_t *b, const uint32_t *wG, const uint32_t *draws, uint32_t *gen_scalars);
__global__ void k_linc_sterms_fixed(uint32_t n_q, uint32_t nthreads, linc_shape sh, const uint32_t *b, const uint32_t *wG, const uint32_t *draws, uint32_t *gen_scalars);
__global__ void k_linc_final(linc_shape sh, const uint32_t *msm_out, const uint8_t *msm_status, uint32_t *ts, const uint32_t *a, const uint32_t *draws, const uint32_t *r, uint8_t *proofs, uint32_t proof_len, uint32_t *status, uint8_t *status_out);

#endif
