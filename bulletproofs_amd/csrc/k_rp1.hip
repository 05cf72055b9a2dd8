// k_rp1.hip: HIP kernels of libbpgpu.so (gfx950); thin __global__ wrappers around the per-lane bodies in the headers.
#include <hip/hip_runtime.h>
#define BP_KECCAK_OUTOFLINE 1   // byte-wise STROBE framing reaches Keccak-f[1600] through ONE out-of-line copy in this translation unit (keccak.h)
#include "kernels.h"

using namespace bp;

// ---- range-proof front end ----------------------------------------------------
// The device overlaps at most a handful of kernels, so a chain of narrow launches leaves most CUs idle.
// Stages that do not depend on each other therefore share ONE launch: the leading blocks of the grid take
// one role, the rest the other ("role-fused" launches; the long-running role gets the low block indices so
// the dispatcher starts it first).
// launch 1: [0, n_tr) Fiat-Shamir transcript replay, then the per-proof scalars (one inversion by division
// steps, the U coefficient recodings, the Montgomery tables for launch 2), lane = proof  ||  [n_tr, ..) decode
// the proof's and the commitments' points straight from the input bytes and build their 8-entry tables,
// lane = point
// Two wavefronts per SIMD (256 registers): alone the launch takes 407 instead of 400 us, but a wavefront that needs a whole SIMD's
// register file waits for a completely free SIMD and, while it waits, holds up the dispatch of the queues behind it -- with 128
// streams in flight the cap is worth +2...+4 % (5.43 -> 5.67 and 5.22 -> 5.33 M/s on two boxes, interleaved A/B; three waves per
// SIMD: no gain, more spills).
// SCRIPTED: the kernel carries ONE transcript path -- the per-shape script (rp_script.h: every proof starts at the same STROBE
// position; one inlined Keccak-f[1600]) or the byte-wise replay (per-proof states at different positions; ~10 framing call sites
// around the permutation, which is therefore a function call there).  Round 3 had both in one kernel: 926 KB of machine code.
template <bool SCRIPTED>
__global__ void __attribute__((amdgpu_waves_per_eu(2, 2))) __launch_bounds__(RP_BLOCK) k_rp_stage1(rp_shape sh, rp_strobe_init init, uint32_t n_tr, const uint8_t *proofs,
                                                         const uint8_t *commitments, const uint8_t *rng64, uint32_t *fields,
                                                         ge_cached *tab, uint32_t *status, fb_params prm, uint32_t lg_m,
                                                         uint32_t *recoded, fb_digit *digits, const uint8_t *rho64, uint32_t ts_flags,
                                                         const uint32_t *ts_in, uint32_t *ts_out, fb_entry *bk_pts, uint32_t bk_c,
                                                         rp_seg_tab segs, const rp_script_hdr *script) {
    __shared__ uint32_t lds[50 * RP_BLOCK];   // sponge states, word-major: word w of lane t at w*RP_BLOCK + t
    if (blockIdx.x < n_tr) {
        const uint32_t p = blockIdx.x * RP_BLOCK + threadIdx.x;
        kstate st;
        st.w = lds + threadIdx.x;
        st.stride = RP_BLOCK;
        if (p < sh.nproofs) {
            // (BP_EXP_*: timing experiments only -- tools/archive/stage1_breakdown.py builds variants with one role compiled out)
#ifndef BP_EXP_NOTR
            if (SCRIPTED) rp_transcript_scripted(p, sh, init, st, rp_resolve(p, sh, proofs, commitments, rng64, segs), script, fields, status, ts_out, ts_in);
            else rp_transcript_thread(p, sh, init, st, rp_resolve(p, sh, proofs, commitments, rng64, segs), fields, status, ts_flags, ts_in, ts_out);
#endif
#ifndef BP_EXP_NOSC
            if (!sh.shape_verdict) rp_expand_a_thread(p, sh, prm, lg_m, fields, recoded, digits, status, rho64, bk_c);
#endif
        }
    } else {
        const uint32_t t = (blockIdx.x - n_tr) * RP_BLOCK + threadIdx.x;
#ifndef BP_EXP_NOPT
        if (t < sh.nproofs * sh.U) rp_points_thread(t, sh, rp_resolve(t / sh.U, sh, proofs, commitments, nullptr, segs), tab, status, bk_pts);
#endif
    }
}
template __global__ void k_rp_stage1<true>(rp_shape, rp_strobe_init, uint32_t, const uint8_t *, const uint8_t *, const uint8_t *, uint32_t *, ge_cached *, uint32_t *, fb_params,
                                           uint32_t, uint32_t *, fb_digit *, const uint8_t *, uint32_t, const uint32_t *, uint32_t *, fb_entry *, uint32_t, rp_seg_tab,
                                           const rp_script_hdr *);
template __global__ void k_rp_stage1<false>(rp_shape, rp_strobe_init, uint32_t, const uint8_t *, const uint8_t *, const uint8_t *, uint32_t *, ge_cached *, uint32_t *, fb_params,
                                            uint32_t, uint32_t *, fb_digit *, const uint8_t *, uint32_t, const uint32_t *, uint32_t *, fb_entry *, uint32_t, rp_seg_tab,
                                            const rp_script_hdr *);

// launch 1 of a NARROW chain (option "transcript_coop", chains of up to 256 proofs): blocks [0, n_tr) = two proofs each, 32 lanes per proof
// -- the group's leader replays the script, all lanes run the permutations together (keccak.h: keccak_f1600_masked_coop), then the
// leader derives the per-proof scalars  ||  [n_tr, ..) the decode role as in k_rp_stage1.  No register cap: a handful of wavefronts.
__global__ void __launch_bounds__(RP_BLOCK) k_rp_stage1_coop(rp_shape sh, rp_strobe_init init, uint32_t n_tr, const uint8_t *proofs, const uint8_t *commitments,
                                                             const uint8_t *rng64, uint32_t *fields, ge_cached *tab, uint32_t *status, fb_params prm, uint32_t lg_m,
                                                             uint32_t *recoded, fb_digit *digits, const uint8_t *rho64, const uint32_t *ts_in, uint32_t *ts_out,
                                                             fb_entry *bk_pts, uint32_t bk_c, rp_seg_tab segs, const rp_script_hdr *script, uint32_t n_pt, ge_cached *tab_hi) {
    __shared__ uint32_t lds[2 * 52];   // one 50-word sponge state per group
    __shared__ sc28 dslot[2][RP_DEFER_CAP + 1];   // (option coop_defer_emit) the scalar role's coefficients, parked for the group's lanes
    __shared__ uint32_t dmeta[2][2];
    __shared__ uint32_t spark[2][32 * 8];   // (option coop_split) the k + 1 values the group's lanes invert, one each
    __shared__ uint32_t sgo[2];
    __shared__ uint32_t schal[2][16 * 32];  // the raw challenges, parked by the leader for the lanes that reduce them
    __shared__ uint32_t sflag[2];
    __shared__ uint32_t sstage[2][RP_COOP_STAGE_WORDS];   // the proof's bytes and its commitments
    __shared__ uint32_t sops[RP_COOP_OPS_CAP * 4 + RP_COOP_MASKS_CAP * RS_MASK_WORDS];   // the script's operations and masks (the same for both groups)
    if (blockIdx.x < n_tr) {
        const uint32_t lane = threadIdx.x, g = lane >> 5, p = blockIdx.x * 2 + g;
        const bool valid = p < sh.nproofs;
        const uint32_t pp = valid ? p : sh.nproofs - 1;   // an idle group walks the script on the last proof's pointers and touches nothing
        kstate st;
        st.w = lds + 52 * g;
        st.stride = 1;
        const bool defer = sh.defer_emit && sh.U <= RP_DEFER_CAP;   // (wavefront-uniform)
        const bool split = sh.coop_split && !rho64 && !bk_c && !(sh.seeded & RP_SEED_WEIGHTS) && sh.k <= 26;   // (launch-uniform)
        if ((lane & 31) == 0) {
            dmeta[g][0] = 0;
            sgo[g] = 0;
            sflag[g] = 0;
        }
        rp_inputs in = rp_resolve(pp, sh, proofs, commitments, rng64, segs);
        rp_defer df;
        df.slot = dslot[g];
        df.meta = dmeta[g];
        if (split) {
            // everything the leader would fetch from global memory one dependent load after the other, staged by all lanes at once
            const rp_script_op *ops_l = nullptr;
            const uint32_t *masks_l = nullptr;
            if (script->n_ops <= RP_COOP_OPS_CAP && script->n_masks <= RP_COOP_MASKS_CAP) {
                const uint32_t *src = (const uint32_t *)rp_script_ops(script);
                const uint32_t nw = script->n_ops * 4 + script->n_masks * RS_MASK_WORDS;   // (the masks follow the operations)
                for (uint32_t i = lane; i < nw; i += RP_BLOCK) sops[i] = src[i];
                ops_l = (const rp_script_op *)sops;
                masks_l = sops + script->n_ops * 4;
            }
            const uint32_t pw = sh.proof_len / 4, cw = 8 * sh.m;
            if (pw + cw <= RP_COOP_STAGE_WORDS) {
                const uint32_t *s0 = (const uint32_t *)in.pr, *s1 = (const uint32_t *)in.cm;
                for (uint32_t i = lane & 31; i < pw; i += 32) sstage[g][i] = s0[i];
                for (uint32_t i = lane & 31; i < cw; i += 32) sstage[g][pw + i] = s1[i];
                in.pr = (const uint8_t *)sstage[g];
                in.cm = (const uint8_t *)(sstage[g] + pw);
            }
            __syncthreads();
            rp_chal_park cp;
            cp.raw = schal[g];
            cp.flag = sflag + g;
            rp_transcript_scripted_coop(pp, valid, lane, sh, init, st, in, script, fields, status, ts_out, ts_in, &cp, ops_l, masks_l);
            __syncthreads();
            // the 5 + k challenges reduced on 5 + k lanes; the k + 1 inversions on k + 1 lanes; the leader forms the rest; the basepoint
            // coefficients wait for launch 3
            rp_split sp;
            sp.park = spark[g];
            sp.go = sgo + g;
            if (valid) rp_coop_reduce_lane(lane & 31, p, sh, cp, fields, sp.park);
            if ((lane & 31) == 0) sgo[g] = (valid && (sflag[g] & 2u) && !sh.shape_verdict) ? 1u : 0u;
            __syncthreads();
            if (valid) rp_split_invert_lane(lane & 31, p, sh, fields, recoded, sp, defer ? &df : nullptr);
#ifndef BP_EXP_NOREST   // timing experiments only
            if (valid && (lane & 31) == 0 && !sh.shape_verdict)
                rp_expand_a_thread(p, sh, prm, lg_m, fields, recoded, digits, status, nullptr, 0, defer ? &df : nullptr, RP_SKIP_INV | RP_SKIP_ROWS);
#endif
        } else {
            rp_transcript_scripted_coop(pp, valid, lane, sh, init, st, in, script, fields, status, ts_out, ts_in);
            if (valid && (lane & 31) == 0 && !sh.shape_verdict) rp_expand_a_thread(p, sh, prm, lg_m, fields, recoded, digits, status, rho64, bk_c, defer ? &df : nullptr);
        }
        if (defer) {   // the leader parked the U coefficients: one lane each recodes them
            __syncthreads();
            if (valid && !sh.shape_verdict) rp_emit_deferred(lane & 31, p, sh, recoded, df, bk_c);
        }
    } else if (blockIdx.x < n_tr + n_pt) {
        const uint32_t t = (blockIdx.x - n_tr) * RP_BLOCK + threadIdx.x;
        if (t < sh.nproofs * sh.U) rp_points_thread(t, sh, rp_resolve(t / sh.U, sh, proofs, commitments, nullptr, segs), tab, status, bk_pts);
    } else {
        // Very narrow chains (sh.narrow_hi): one WAVEFRONT per per-proof point decodes it again (the inverse square root's 254 squarings one limb
        // per lane: hw_ristretto_decode) and doubles it 128 times in the wavefront-cooperative form (horner_wave.h: ~0.45 us per doubling
        // instead of ~2.5 in one lane), then builds the 8-entry table of Q = 2^128 P.
        // A coefficient s = s_lo + 2^128 s_hi contributes s_lo P + s_hi Q: 32 windows for the Horner chain instead of 64 -- 124 dependent
        // doublings on the chain's critical path instead of 252; the other 128 run here, beside the transcript.  (A, whose coefficient
        // is 1, has no upper digits: skipped.)
        // (sh.narrow_hi = 4: three such tables per point -- 2^64 P, 2^128 P, 2^192 P, a wavefront each -- and a 16-window chain)
        const uint32_t idx = blockIdx.x - n_tr - n_pt, npts = sh.nproofs * sh.U, lv = idx / npts + 1, t = idx - (lv - 1) * npts, p = t / sh.U, u = t - p * sh.U;
        if (u == 0) return;
        const rp_inputs in = rp_resolve(p, sh, proofs, commitments, nullptr, segs);
        uint32_t w[8];
        load_words8(w, rp_unique_point_ptr(sh, in, u));
        ge_ext pt;
        hw_ristretto_decode(pt, w);   // (an undecodable point is reported by the decode role; its tables are never used)
        hw_shift_table8(pt, (int)(lv * (256u / sh.narrow_hi)), tab_hi + 8 * ((uint64_t)(lv - 1) * npts + t));   // (the eight multiples of Q in the wavefront's layout too: ~10 us instead of ~25 in one lane)
    }
}

// verdict[p] = status (Format / shape / Verification) if set, else the identity test of the mega-check
__global__ void __launch_bounds__(64) k_rp_verdict(uint32_t n, uint32_t *status, const uint8_t *msm_verdict, uint8_t *out) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) {
        out[p] = status[p] ? (uint8_t)status[p] : msm_verdict[p];
        status[p] = 0;   // handed back clean (see bpgpu_ctx::rp_status)
    }
}
