// CPU harness: compiles the device headers (bulletproofs_amd/csrc/*.h) with g++
// and BP_FE_CHECK so the exact per-lane code of the HIP kernels is unit-tested
// without a GPU.  TEST-ONLY: never part of libbpgpu.so, never a fallback.
#define BP_FE_CHECK 1
#include "../../bulletproofs_amd/csrc/fe25519.h"
#include "../../bulletproofs_amd/csrc/ge25519.h"
#include "../../bulletproofs_amd/csrc/msm_vb.h"
#include "../../bulletproofs_amd/csrc/msm_fixed.h"
#include "../../bulletproofs_amd/csrc/rangeproof.h"
#include "../../bulletproofs_amd/csrc/horner_wave.h"
#include "../../bulletproofs_amd/csrc/horner_quad.h"
#include "../../bulletproofs_amd/csrc/ipp.h"
#include "../../bulletproofs_amd/csrc/linear.h"
#include "../../bulletproofs_amd/csrc/audit.h"
#include "../../bulletproofs_amd/csrc/scinv.h"
#include "../../bulletproofs_amd/csrc/rlc.h"
#include "../../bulletproofs_amd/csrc/bucket.h"
#include "../../bulletproofs_amd/csrc/bucket2.h"
#include "../../bulletproofs_amd/csrc/ipp_prover.h"
#include "../../bulletproofs_amd/csrc/rp_prover.h"
#include "../../bulletproofs_amd/csrc/linear_prover.h"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <deque>
#include <vector>
using namespace bp;

static void load(fe &f, const uint8_t *b) { uint32_t w[8]; memcpy(w, b, 32); fe_from_words(f, w); }
static void store(uint8_t *b, const fe &f) { uint32_t w[8]; fe_to_words(w, f); memcpy(b, w, 32); }

// column sums -> Horner results through the wavefront-cooperative chain (lockstep emulation of the 64 lanes),
// cross-checked against the one-lane chain on the ristretto encoding
static int horner_all(uint32_t nbatch, const std::vector<ge_ext> &col, const std::vector<uint32_t> &colq16, std::vector<ge_ext> &hq) {
    hq.resize(nbatch + 1);
    for (uint32_t b = 0; b < nbatch; b++) {
        hw_horner_msm((const uint16_t *)(colq16.data() + (size_t)b * 64 * 32), &hq[b]);
        ge_ext ref; vb_horner_point(ref, col.data() + (size_t)b * 64);
        uint32_t e1[8], e2[8]; ristretto_compress(e1, hq[b]); ristretto_compress(e2, ref);
        if (memcmp(e1, e2, 32) != 0) { std::fprintf(stderr, "horner_wave mismatch at msm %u\n", b); return -99; }
    }
    // and the four-lane chain (horner_quad.h) from the same column sums in cached form
    std::vector<ge_cached> colc((size_t)nbatch * 64 + 1);
    std::vector<ge_ext> hq4(nbatch + 1);
    for (size_t i = 0; i < (size_t)nbatch * 64; i++) ge_to_cached(colc[i], col[i]);
    for (uint32_t b = 0; b < nbatch; b++) {
        hq_horner_msm(b, nbatch, colc.data(), hq4.data());
        uint32_t e1[8], e2[8]; ristretto_compress(e1, hq[b]); ristretto_compress(e2, hq4[b]);
        if (memcmp(e1, e2, 32) != 0) { std::fprintf(stderr, "horner_quad mismatch at msm %u\n", b); return -98; }
    }
    return 0;
}

extern "C" {
// op: 0 mul 1 sq 2 add 3 sub 4 neg 5 invert 6 pow22523 7 carry-roundtrip 8 lazy-chain
void h_fe_op(int op, const uint8_t *a, const uint8_t *b, uint8_t *out) {
    fe x, y, r; load(x, a); load(y, b);
    switch (op) {
    case 0: fe_mul(r, x, y); break;
    case 1: fe_sq(r, x); break;
    case 2: fe_add(r, x, y); break;
    case 3: fe_sub(r, x, y); break;
    case 4: fe_neg(r, x); break;
    case 5: fe_invert(r, x); break;
    case 6: fe_pow22523(r, x); break;
    case 7: r = x; fe_carry(r); break;
    case 8: { fe t, u; fe_add(t, x, y); fe_add(t, t, x); fe_add(u, y, y); fe_add(u, u, x); fe_mul(r, t, u); fe_sq(t, t); fe_sub(r, r, t); } break;  // (2x+y)(x+2y)-(2x+y)^2
    default: fe_0(r);
    }
    store(out, r);
}
int h_fe_flags(const uint8_t *a) { fe x; load(x, a); return (fe_isneg(x) ? 1 : 0) | (fe_iszero(x) ? 2 : 0); }

int h_decompress(const uint8_t *in, uint8_t *xyzt /*4x32*/) {
    uint32_t w[8]; memcpy(w, in, 32);
    ge_ext p; bool ok = ristretto_decompress(p, w);
    store(xyzt, p.X); store(xyzt + 32, p.Y); store(xyzt + 64, p.Z); store(xyzt + 96, p.T);
    return ok ? 1 : 0;
}
void h_compress_xyzt(const uint8_t *xyzt, uint8_t *out) {
    ge_ext p; load(p.X, xyzt); load(p.Y, xyzt + 32); load(p.Z, xyzt + 64); load(p.T, xyzt + 96);
    uint32_t w[8]; ristretto_compress(w, p); memcpy(out, w, 32);
}
// op: 0 add 1 sub 2 dbl 3 madd(+) 4 madd(-) 5 roundtrip 6 dbl x4 (3 without T) 7 neg ; inputs compressed
int h_point_op(int op, const uint8_t *a, const uint8_t *b, uint8_t *out) {
    uint32_t wa[8], wb[8], wo[8]; memcpy(wa, a, 32); memcpy(wb, b, 32);
    ge_ext p, q, r; if (!ristretto_decompress(p, wa)) return 0; if (!ristretto_decompress(q, wb)) return 0;
    // de-normalise p so Z != 1 paths are exercised
    ge_dbl(r, p); ge_cached pc; ge_to_cached(pc, p); ge_add_cached(p, r, pc, true);
    ge_cached qc; ge_to_cached(qc, q);
    ge_niels qn; qn.ypx = qc.YpX; qn.ymx = qc.YmX; qn.t2d = qc.T2d;   // q has Z = 1
    switch (op) {
    case 0: ge_add_cached(r, p, qc, false); break;
    case 1: ge_add_cached(r, p, qc, true); break;
    case 2: ge_dbl(r, p); break;
    case 3: ge_madd(r, p, qn, false); break;
    case 4: ge_madd(r, p, qn, true); break;
    case 5: r = p; break;
    case 6: r = p; ge_dbl(r, r, false); ge_dbl(r, r, false); ge_dbl(r, r, false); ge_dbl(r, r, true); ge_add_cached(r, r, qc, false); break;
    case 7: ge_neg(r, p); break;
    default: ge_identity(r);
    }
    ristretto_compress(wo, r); memcpy(out, wo, 32);
    return (ge_is_identity(r) ? 2 : 0) | 1;
}
void h_from_uniform(const uint8_t *in64, uint8_t *out) {
    uint32_t w[16], wo[8]; memcpy(w, in64, 64);
    ge_ext r; ristretto_from_uniform(r, w); ristretto_compress(wo, r); memcpy(out, wo, 32);
}

// Emulates the 4-stage variable-base pipeline lane by lane (same bodies as the HIP kernels).
void h_msm_vb(uint32_t nbatch, const uint32_t *n_terms, const uint8_t *scalars, const uint8_t *points,
              uint8_t *out, uint8_t *status_out) {
    std::vector<vb_chunk> chunks; std::vector<uint32_t> chunk_first(nbatch + 1), term_chunk;
    uint32_t t0 = 0;
    for (uint32_t b = 0; b < nbatch; b++) {
        chunk_first[b] = (uint32_t)chunks.size();
        for (uint32_t k = 0; k < n_terms[b]; k += BP_VB_CHUNK) {
            vb_chunk c; c.msm = b; c.first = t0 + k; c.count = n_terms[b] - k < BP_VB_CHUNK ? n_terms[b] - k : BP_VB_CHUNK; c.pad = 0;
            for (uint32_t i = 0; i < c.count; i++) term_chunk.push_back((uint32_t)chunks.size());
            chunks.push_back(c);
        }
        t0 += n_terms[b];
    }
    chunk_first[nbatch] = (uint32_t)chunks.size();
    uint32_t total = t0;
    std::vector<ge_cached> tab((size_t)total * 8 + 1);
    std::vector<uint32_t> rec((size_t)total * 8 + 1), status(nbatch + 1, 0), outw((size_t)nbatch * 8 + 1);
    std::vector<ge_ext> part(chunks.size() * 64 + 1), col((size_t)nbatch * 64 + 1), hq;
    std::vector<uint32_t> colq16((size_t)nbatch * 64 * 32 + 1);
    for (uint32_t t = 0; t < total; t++)
        vb_prepare_thread(t, chunks.data(), term_chunk.data(), (const uint32_t *)scalars, (const uint32_t *)points, tab.data(), rec.data(), status.data());
    for (uint32_t tid = 0; tid < chunks.size() * 64; tid++) vb_window_thread(tid, chunks.data(), tab.data(), rec.data(), part.data());
    for (uint32_t tid = 0; tid < nbatch * 64; tid++) vb_colsum_thread(tid, chunk_first.data(), part.data(), col.data(), colq16.data());
    if (horner_all(nbatch, col, colq16, hq)) { memset(out, 0xee, (size_t)nbatch * 32); return; }
    for (uint32_t b = 0; b < nbatch; b++) vb_horner_thread(b, nullptr, hq.data(), status.data(), outw.data(), nullptr);
    memcpy(out, outw.data(), (size_t)nbatch * 32);
    for (uint32_t b = 0; b < nbatch; b++) status_out[b] = (uint8_t)status[b];
}
// The narrow form of the same pipeline (k_vb_prepare_hi / k_vb_window_hi / k_vb_tail_narrow: a few small MSMs per call): chunks of `chunk` terms,
// second tables of the 2^128 multiples from the wavefront-cooperative decode + doublings, 32-window chain, encoding through the split
// inverse-square-root (lane 0 front / wavefront chain / lane 0 back).
void h_msm_vb_narrow(uint32_t nbatch, const uint32_t *n_terms, uint32_t chunk, uint32_t levels, const uint8_t *scalars, const uint8_t *points, uint8_t *out, uint8_t *status_out) {
    std::vector<vb_chunk> chunks; std::vector<uint32_t> chunk_first(nbatch + 1), term_chunk;
    uint32_t t0 = 0;
    for (uint32_t b = 0; b < nbatch; b++) {
        chunk_first[b] = (uint32_t)chunks.size();
        for (uint32_t k = 0; k < n_terms[b]; k += chunk) {
            vb_chunk c; c.msm = b; c.first = t0 + k; c.count = n_terms[b] - k < chunk ? n_terms[b] - k : chunk; c.pad = 0;
            for (uint32_t i = 0; i < c.count; i++) term_chunk.push_back((uint32_t)chunks.size());
            chunks.push_back(c);
        }
        t0 += n_terms[b];
    }
    chunk_first[nbatch] = (uint32_t)chunks.size();
    const uint32_t total = t0;
    std::vector<ge_cached> tab((size_t)total * 8 + 1), tab_hi((size_t)total * 8 * (levels - 1) + 1);
    std::vector<uint32_t> rec((size_t)total * 8 + 1), status(nbatch + 1, 0), outw((size_t)nbatch * 8 + 1);
    std::vector<ge_ext> part(chunks.size() * 64 + 1);
    for (uint32_t t = 0; t < total; t++)
        vb_prepare_thread(t, chunks.data(), term_chunk.data(), (const uint32_t *)scalars, (const uint32_t *)points, tab.data(), rec.data(), status.data());
    for (uint32_t lv = 1; lv < levels; lv++)
        for (uint32_t t = 0; t < total; t++) {
            ge_ext pt;
            hw_ristretto_decode(pt, (const uint32_t *)points + 8 * (size_t)t);
            hw_shift_table8(pt, (int)(lv * (256u / levels)), tab_hi.data() + 8 * ((size_t)(lv - 1) * total + t));
        }
    for (uint32_t tid = 0; tid < chunks.size() * 64; tid++)
        vb_window_thread(tid, chunks.data(), tab.data(), rec.data(), part.data(), nullptr, nullptr, tab_hi.data(), levels, (uint64_t)8 * total);
    std::vector<uint8_t> sb(nbatch + 1);
    for (uint32_t b = 0; b < nbatch; b++) {
        ge_ext fin; fe tin, raw; uint32_t tw[8];
        hw_colsum_horner_msm(b, chunk_first.data(), part.data(), &fin, (int)levels);
        bk2_tail_t4a(&fin, &tin, tw);
        hw_invsqrt_raw_fe((const uint16_t *)tw, nullptr, &raw);
        bk2_tail_t4b(b, &fin, &raw, &tin, status.data(), outw.data(), nullptr, sb.data());
    }
    memcpy(out, outw.data(), (size_t)nbatch * 32);
    for (uint32_t b = 0; b < nbatch; b++) status_out[b] = sb[b];
}
// Emulates the shared-generator pipeline (table build, recode, split accumulation, finish).
// gens: n_gens_loaded compressed points in table order; gen_ids: n_gen_terms ids (the (n,m) subset).
int h_msm_shared(uint32_t W, uint32_t nsplit, uint32_t n_gens_loaded, const uint8_t *gens, uint32_t n_gen_terms, const uint32_t *gen_ids,
                 uint32_t nbatch, uint32_t n_unique, const uint8_t *gen_scalars, const uint8_t *uniq_scalars, const uint8_t *uniq_points,
                 uint8_t *out, uint8_t *status_out, uint8_t *verdict_out) {
    fb_params prm; prm.W = W; prm.nwin = fb_nwin(W); prm.half = 1u << (W - 1); prm.n_gens = n_gens_loaded;
    std::vector<ge_ext> base((size_t)prm.n_gens * prm.nwin);
    std::vector<fb_entry> table((size_t)prm.n_gens * prm.nwin * prm.half);
    uint32_t bad = 0;
    for (uint32_t g = 0; g < prm.n_gens; g++) fb_base_thread(g, prm, (const uint32_t *)gens, base.data(), &bad);
    if (bad) return -5;
    for (uint32_t t = 0; t < prm.n_gens * prm.nwin; t++) fb_fill_thread(t, prm, base.data(), table.data());
    for (uint64_t gq = 0; gq < (table.size() + BP_FB_NORM_GROUP - 1) / BP_FB_NORM_GROUP; gq++) fb_norm_thread(gq, table.size(), table.data());
    // unique part
    std::vector<vb_chunk> chunks; std::vector<uint32_t> chunk_first(nbatch + 1), term_chunk;
    uint32_t t0 = 0;
    for (uint32_t b = 0; b < nbatch; b++) {
        chunk_first[b] = (uint32_t)chunks.size();
        for (uint32_t k = 0; k < n_unique; k += BP_VB_CHUNK) {
            vb_chunk c; c.msm = b; c.first = t0 + k; c.count = n_unique - k < BP_VB_CHUNK ? n_unique - k : BP_VB_CHUNK; c.pad = 0;
            for (uint32_t i = 0; i < c.count; i++) term_chunk.push_back((uint32_t)chunks.size());
            chunks.push_back(c);
        }
        t0 += n_unique;
    }
    chunk_first[nbatch] = (uint32_t)chunks.size();
    std::vector<ge_cached> tab((size_t)t0 * 8 + 1);
    std::vector<uint32_t> rec((size_t)t0 * 8 + 1), status(nbatch + 1, 0), outw((size_t)nbatch * 8 + 1);
    std::vector<ge_ext> part(chunks.size() * 64 + 1), col((size_t)nbatch * 64 + 1), hq;
    std::vector<uint32_t> colq16((size_t)nbatch * 64 * 32 + 1);
    for (uint32_t t = 0; t < t0; t++)
        vb_prepare_thread(t, chunks.data(), term_chunk.data(), (const uint32_t *)uniq_scalars, (const uint32_t *)uniq_points, tab.data(), rec.data(), status.data());
    for (uint32_t tid = 0; tid < chunks.size() * 64; tid++) vb_window_thread(tid, chunks.data(), tab.data(), rec.data(), part.data());
    for (uint32_t tid = 0; tid < nbatch * 64; tid++) vb_colsum_thread(tid, chunk_first.data(), part.data(), col.data(), colq16.data());
    if (horner_all(nbatch, col, colq16, hq)) return -99;
    // generator part
    const uint32_t npairs = n_gen_terms * prm.nwin;
    std::vector<fb_digit> digits((size_t)npairs * nbatch + 1);
    for (uint32_t tid = 0; tid < n_gen_terms * nbatch; tid++) fb_recode_thread(tid, prm, nbatch, n_gen_terms, (const uint32_t *)gen_scalars, digits.data(), status.data());
    std::vector<ge_ext> partial((size_t)nsplit * nbatch + 1);
    const uint32_t per = (npairs + nsplit - 1) / nsplit;
    for (uint32_t sp = 0; sp < nsplit; sp++) {
        uint32_t q0 = sp * per, q1 = q0 + per < npairs ? q0 + per : npairs; if (q0 > npairs) q0 = npairs;
        for (uint32_t p = 0; p < nbatch; p++) fb_accum_thread(p, sp, q0, q1, prm, nbatch, gen_ids, digits.data(), table.data(), partial.data());
    }
    std::vector<uint8_t> verdict(nbatch + 1);
    for (uint32_t p = 0; p < nbatch; p++) shared_finish_thread(p, nbatch, nsplit, nullptr, n_unique != 0, hq.data(), partial.data(), status.data(), outw.data(), verdict.data());
    memcpy(out, outw.data(), (size_t)nbatch * 32);
    for (uint32_t b = 0; b < nbatch; b++) { status_out[b] = (uint8_t)status[b]; verdict_out[b] = verdict[b]; }
    return 0;
}
// host-side Merlin prefix: Transcript::new(label) + rangeproof_domain_sep(n, m)  (same code the runtime uses)
static void make_strobe_init(rp_strobe_init &init, const uint8_t *label, uint32_t label_len, uint64_t n, uint64_t m) {
    kstate st; st.w = init.w; st.stride = 1;
    strobe t; merlin_strobe_init(t, st);
    const uint8_t dom[7] = {'d','o','m','-','s','e','p'}, rp[13] = {'r','a','n','g','e','p','r','o','o','f',' ','v','1'}, ln[1] = {'n'}, lm[1] = {'m'};
    merlin_append_message(t, dom, 7, label, label_len);
    merlin_append_message(t, dom, 7, rp, 13);
    merlin_append_u64(t, ln, 1, n); merlin_append_u64(t, lm, 1, m);
    init.pos = t.pos; init.pos_begin = t.pos_begin; init.cur_flags = t.cur_flags;
}

// The scripted transcript replay (rp_script.h + rp_transcript_scripted) against the byte-wise one (rp_transcript_thread), lane by
// lane: same per-proof scalars, same status, same advanced transcript.  state208: the start state (Transcript::new(label) with
// whatever the caller appended); domsep: rangeproof_domain_sep(n, m) still to be applied.  Returns 0 when everything is identical,
// and hands back the byte-wise path's fields / status / transcripts for comparison with the oracle.
int h_rp_transcript_compare(uint32_t n, uint32_t m, uint32_t nbatch, const uint8_t *proofs, uint32_t proof_len, const uint8_t *commitments,
                            const uint8_t *rng64, const uint8_t *state208, int domsep, uint32_t *n_perm_out, uint8_t *status_out, uint8_t *ts_out208) {
    uint32_t k = 0; while ((1u << k) < n * m) k++;
    rp_shape sh; sh.n = n; sh.m = m; sh.nm = n * m; sh.k = k; sh.U = 4 + 2 * k + m; sh.proof_len = proof_len; sh.nproofs = nbatch; sh.shape_verdict = 0;
    if (proof_len != 32 * (9 + 2 * k)) return -1;
    rp_strobe_init init;
    memcpy(init.w, state208, 200);
    init.pos = state208[200]; init.pos_begin = state208[201]; init.cur_flags = state208[202];
    const std::vector<uint32_t> img = rp_script_build(n, m, k, init.pos, init.pos_begin, init.cur_flags, domsep != 0);
    const rp_script_hdr *script = (const rp_script_hdr *)img.data();
    if (n_perm_out) *n_perm_out = script->n_masks;
    const rp_fields fl = rp_field_layout(k, m);
    const size_t nf = (size_t)fl.count * nbatch * BP_RP_REC + 8;
    std::vector<uint32_t> f1(nf, 0xabababab), f2(nf, 0xabababab), s1(nbatch + 1, 0), s2(nbatch + 1, 0), t1((size_t)nbatch * BP_TS_WORDS, 7), t2((size_t)nbatch * BP_TS_WORDS, 7);
    std::vector<uint32_t> f3(nf, 0xabababab), s3(nbatch + 1, 0), t3((size_t)nbatch * BP_TS_WORDS, 7);
    std::vector<uint32_t> f4(nf, 0xabababab), s4(nbatch + 1, 0), t4((size_t)nbatch * BP_TS_WORDS, 7);
    rp_seg_tab none; memset(&none, 0, sizeof none);
    for (uint32_t p = 0; p < nbatch; p++) {
        uint32_t w1[50], w2[50]; kstate st1, st2; st1.w = w1; st1.stride = 1; st2.w = w2; st2.stride = 1;
        const rp_inputs in = rp_resolve(p, sh, proofs, commitments, rng64, none);
        rp_transcript_thread(p, sh, init, st1, in, f1.data(), s1.data(), domsep ? BP_TS_DOMSEP : 0, nullptr, t1.data());
        rp_transcript_scripted(p, sh, init, st2, in, script, f2.data(), s2.data(), t2.data());
        // the narrow-chain variant (32 lanes per proof): the leader's path, the permutations through the 25-lane phase functions
        uint32_t w3[52]; kstate st3; st3.w = w3; st3.stride = 1;
        rp_transcript_scripted_coop(p, true, 0, sh, init, st3, in, script, f3.data(), s3.data(), t3.data());
        // ... and with the challenges parked raw by the leader and reduced by the group's lanes (option coop_split), operations and masks from a staged copy
        if (k <= 26) {
            uint32_t w4[52]; kstate st4; st4.w = w4; st4.stride = 1;
            std::vector<uint32_t> raw(16 * 32, 0xdeadbeef), park(32 * 8, 0xdeadbeef);
            uint32_t flag = 0;
            rp_chal_park cp; cp.raw = raw.data(); cp.flag = &flag;
            std::vector<uint32_t> staged((const uint32_t *)rp_script_ops(script), (const uint32_t *)rp_script_ops(script) + script->n_ops * 4 + script->n_masks * RS_MASK_WORDS);
            rp_transcript_scripted_coop(p, true, 0, sh, init, st4, in, script, f4.data(), s4.data(), t4.data(), nullptr, &cp, (const rp_script_op *)staged.data(),
                                        staged.data() + script->n_ops * 4);
            for (uint32_t id = 0; id < 32; id++) rp_coop_reduce_lane(id, p, sh, cp, f4.data(), park.data());
            if (flag & 1u) {   // what the lanes that invert read: y and the u_i in canonical form
                for (uint32_t i = 0; i <= k; i++) {
                    sc v; rp_load(v, f1.data(), nbatch, i < k ? fl.u + i : (uint32_t)RPF_Y, p);
                    for (int q = 0; q < 8; q++) if (park[8 * i + q] != v.v[q]) return 10;
                }
                if (((flag >> 1) & 1u) != (s1[p] == 0 ? 1u : 0u)) return 11;
            } else if (s1[p] == 0) return 12;
        }
    }
    if (k <= 26 && (s1 != s4 || t1 != t4 || f1 != f4)) return 13;
    if (s1 != s3) return 4;
    if (t1 != t3) return 5;
    if (f1 != f3) return 6;
    for (uint32_t p = 0; p < nbatch; p++) {
        status_out[p] = (uint8_t)s1[p];
        memcpy(ts_out208 + (size_t)p * 208, &t1[(size_t)p * BP_TS_WORDS], 200);
        const uint32_t meta = t1[(size_t)p * BP_TS_WORDS + 50];
        memset(ts_out208 + (size_t)p * 208 + 200, 0, 8);
        ts_out208[(size_t)p * 208 + 200] = meta & 0xff; ts_out208[(size_t)p * 208 + 201] = (meta >> 8) & 0xff; ts_out208[(size_t)p * 208 + 202] = (meta >> 16) & 0xff;
    }
    if (s1 != s2) return 1;
    if (t1 != t2) return 2;
    if (f1 != f2) return 3;
    return 0;
}

// The same comparison with ONE START STATE PER PROOF, all at the same STROBE position (what the pool's combining queue hands a
// chain: bpgpu_pool_rangeproof_verify_ts, CK_UNIFORM): scripted replay from ts_in[p] against the byte-wise replay from ts_in[p].
// states208: nbatch x 208 bytes.  Returns 0 when identical; -2 when the states do not share a position.
int h_rp_transcript_compare_per_proof(uint32_t n, uint32_t m, uint32_t nbatch, const uint8_t *proofs, uint32_t proof_len, const uint8_t *commitments,
                                      const uint8_t *rng64, const uint8_t *states208, uint8_t *status_out, uint8_t *ts_out208) {
    uint32_t k = 0; while ((1u << k) < n * m) k++;
    rp_shape sh; sh.n = n; sh.m = m; sh.nm = n * m; sh.k = k; sh.U = 4 + 2 * k + m; sh.proof_len = proof_len; sh.nproofs = nbatch; sh.shape_verdict = 0;
    if (proof_len != 32 * (9 + 2 * k)) return -1;
    std::vector<uint32_t> ts_in((size_t)nbatch * BP_TS_WORDS, 0);
    for (uint32_t p = 0; p < nbatch; p++) {
        const uint8_t *st = states208 + (size_t)p * 208;
        if (st[200] != states208[200] || st[201] != states208[201] || st[202] != states208[202]) return -2;
        memcpy(&ts_in[(size_t)p * BP_TS_WORDS], st, 200);
        ts_in[(size_t)p * BP_TS_WORDS + 50] = rp_ts_meta(st[200], st[201], st[202]);
    }
    rp_strobe_init init; memset(&init, 0, sizeof init);   // only the position is shared; the words come from ts_in
    init.pos = states208[200]; init.pos_begin = states208[201]; init.cur_flags = states208[202];
    const std::vector<uint32_t> img = rp_script_build(n, m, k, init.pos, init.pos_begin, init.cur_flags, true);
    const rp_script_hdr *script = (const rp_script_hdr *)img.data();
    const rp_fields fl = rp_field_layout(k, m);
    const size_t nf = (size_t)fl.count * nbatch * BP_RP_REC + 8;
    std::vector<uint32_t> f1(nf, 0xabababab), f2(nf, 0xabababab), s1(nbatch + 1, 0), s2(nbatch + 1, 0), t1((size_t)nbatch * BP_TS_WORDS, 7), t2((size_t)nbatch * BP_TS_WORDS, 7);
    std::vector<uint32_t> f3(nf, 0xabababab), s3(nbatch + 1, 0), t3((size_t)nbatch * BP_TS_WORDS, 7);
    std::vector<uint32_t> f4(nf, 0xabababab), s4(nbatch + 1, 0), t4((size_t)nbatch * BP_TS_WORDS, 7);
    rp_seg_tab none; memset(&none, 0, sizeof none);
    for (uint32_t p = 0; p < nbatch; p++) {
        uint32_t w1[50], w2[50]; kstate st1, st2; st1.w = w1; st1.stride = 1; st2.w = w2; st2.stride = 1;
        const rp_inputs in = rp_resolve(p, sh, proofs, commitments, rng64, none);
        rp_transcript_thread(p, sh, init, st1, in, f1.data(), s1.data(), BP_TS_DOMSEP, ts_in.data(), t1.data());
        rp_transcript_scripted(p, sh, init, st2, in, script, f2.data(), s2.data(), t2.data(), ts_in.data());
        uint32_t w3[52]; kstate st3; st3.w = w3; st3.stride = 1;
        rp_transcript_scripted_coop(p, true, 0, sh, init, st3, in, script, f3.data(), s3.data(), t3.data(), ts_in.data());
        if (k <= 26) {
            uint32_t w4[52]; kstate st4; st4.w = w4; st4.stride = 1;
            std::vector<uint32_t> raw(16 * 32, 0xdeadbeef), park(32 * 8, 0xdeadbeef);
            uint32_t flag = 0;
            rp_chal_park cp; cp.raw = raw.data(); cp.flag = &flag;
            rp_transcript_scripted_coop(p, true, 0, sh, init, st4, in, script, f4.data(), s4.data(), t4.data(), ts_in.data(), &cp);
            for (uint32_t id = 0; id < 32; id++) rp_coop_reduce_lane(id, p, sh, cp, f4.data(), park.data());
        }
    }
    if (k <= 26 && (s1 != s4 || t1 != t4 || f1 != f4)) return 13;
    if (s1 != s3) return 4;
    if (t1 != t3) return 5;
    if (f1 != f3) return 6;
    for (uint32_t p = 0; p < nbatch; p++) {
        status_out[p] = (uint8_t)s1[p];
        memcpy(ts_out208 + (size_t)p * 208, &t1[(size_t)p * BP_TS_WORDS], 200);
        const uint32_t meta = t1[(size_t)p * BP_TS_WORDS + 50];
        memset(ts_out208 + (size_t)p * 208 + 200, 0, 8);
        ts_out208[(size_t)p * 208 + 200] = meta & 0xff; ts_out208[(size_t)p * 208 + 201] = (meta >> 8) & 0xff; ts_out208[(size_t)p * 208 + 202] = (meta >> 16) & 0xff;
    }
    if (s1 != s2) return 1;
    if (t1 != t2) return 2;
    if (f1 != f2) return 3;
    return 0;
}

// n m != 2^k (or m == 0): verification_scalars fails AFTER the range proof's own transcript part (ipp.rs:203-211), so the caller's
// transcript comes back as of the `w` challenge -- or as of an identity A / S / T_1 / T_2 before it.  Byte-wise replay only (the
// library compiles no script for such a shape).  k comes from the proof's length.
int h_rp_transcript_shape_stop(uint32_t n, uint32_t m, const uint8_t *proof, uint32_t proof_len, const uint8_t *commitments, const uint8_t *rng64,
                               const uint8_t *state208, uint8_t *status_out, uint8_t *ts_out208) {
    if (proof_len % 32 || proof_len < 9 * 32 || ((proof_len / 32 - 9) & 1)) return -1;
    const uint32_t k = (proof_len / 32 - 9) / 2;
    rp_shape sh; sh.n = n; sh.m = m; sh.nm = n * m; sh.k = k; sh.U = 4 + 2 * k + m; sh.proof_len = proof_len; sh.nproofs = 1; sh.shape_verdict = BP_VERDICT_VERIFICATION;
    rp_strobe_init init;
    memcpy(init.w, state208, 200);
    init.pos = state208[200]; init.pos_begin = state208[201]; init.cur_flags = state208[202];
    const rp_fields fl = rp_field_layout(k, m);
    std::vector<uint32_t> f((size_t)fl.count * BP_RP_REC + 8, 0), s(2, 0), t(BP_TS_WORDS, 7);
    rp_seg_tab none; memset(&none, 0, sizeof none);
    uint32_t w[50]; kstate st; st.w = w; st.stride = 1;
    rp_transcript_thread(0, sh, init, st, rp_resolve(0, sh, proof, commitments, rng64, none), f.data(), s.data(), BP_TS_DOMSEP, nullptr, t.data());
    status_out[0] = (uint8_t)s[0];
    memset(ts_out208, 0, 208);
    memcpy(ts_out208, t.data(), 200);
    const uint32_t meta = t[50];
    ts_out208[200] = meta & 0xff; ts_out208[201] = (meta >> 8) & 0xff; ts_out208[202] = (meta >> 16) & 0xff;
    return 0;
}

// Keccak-f[1600] through the 25-lane phase functions (keccak.h: keccak_f1600_masked_coop's host twin) and through the serial form,
// with an optional XOR mask on the first nmask words: out_coop / out_serial = the permuted 200-byte states
void h_keccak_coop(const uint8_t *state200, const uint32_t *mask, uint32_t nmask, uint8_t *out_coop, uint8_t *out_serial) {
    uint32_t a[50], b[50];
    memcpy(a, state200, 200); memcpy(b, state200, 200);
    kstate sa, sb; sa.w = a; sa.stride = 1; sb.w = b; sb.stride = 1;
    uint32_t zero[RS_MASK_WORDS] = {0};
    keccak_f1600_masked_coop(sa, mask ? mask : zero, mask ? nmask : 0, 0);
    keccak_f1600_masked(sb, mask ? mask : zero, mask ? nmask : 0);
    memcpy(out_coop, a, 200); memcpy(out_serial, b, 200);
}
void h_merlin_kat(const uint8_t *label, uint32_t label_len, const uint8_t *mlabel, uint32_t mlabel_len, const uint8_t *msg, uint32_t msg_len,
                  const uint8_t *clabel, uint32_t clabel_len, uint8_t *out, uint32_t out_len) {
    uint32_t w[50]; kstate st; st.w = w; st.stride = 1;
    strobe t; merlin_strobe_init(t, st);
    const uint8_t dom[7] = {'d','o','m','-','s','e','p'};
    merlin_append_message(t, dom, 7, label, label_len);
    merlin_append_message(t, mlabel, mlabel_len, msg, msg_len);
    merlin_challenge_bytes(t, clabel, clabel_len, out, out_len);
}
void h_shake256(const uint8_t *in, uint32_t n, uint8_t *out, uint32_t out_len) {
    uint32_t w[50]; kstate st; st.w = w; st.stride = 1; sponge k; sponge_init(k, st, BP_SHAKE256_RATE);
    sponge_absorb(k, in, n); sponge_finish(k, 0x1f); sponge_squeeze(k, out, out_len);
}
void h_sha3_512(const uint8_t *in, uint32_t n, uint8_t *out) {
    uint32_t w[50]; kstate st; st.w = w; st.stride = 1; sponge k; sponge_init(k, st, BP_SHA3_512_RATE);
    sponge_absorb(k, in, n); sponge_finish(k, 0x06); sponge_squeeze(k, out, 64);
}
// op: 0 mul 1 add 2 sub 3 neg 4 invert 5 from_wide(a||b) 6 montmul 7 / 8 division-step inversion 9 its variable-time form
void h_sc_op(int op, const uint8_t *a, const uint8_t *b, uint8_t *out) {
    sc x, y, r; memcpy(x.v, a, 32); memcpy(y.v, b, 32);
    switch (op) {
    case 0: sc_mul(r, x, y); break;
    case 1: sc_add(r, x, y); break;
    case 2: sc_sub(r, x, y); break;
    case 3: sc_neg(r, x); break;
    case 4: sc_invert(r, x); break;
    case 5: { uint32_t w[16]; memcpy(w, a, 32); memcpy(w + 8, b, 32); sc_from_wide(r, w); } break;
    case 6: { sc28 a28, b28, t28; sc28_from_sc(a28, x); sc28_from_sc(b28, y); sc28_montmul(t28, a28, b28); sc28_to_mont(t28, t28); sc28_to_mont(t28, t28); sc_from_mont28(r, t28); } break;  // lazy chain: ((xy/R)*R*R)/R = xy
    case 7: sc_invert_safegcd(r, x); break;
    case 8: { sc28 m28, i28; sc_to_mont28(m28, x); sc28_invert_mont_safegcd(i28, m28); sc_from_mont28(r, i28); } break;
    case 9: sc_invert_safegcd_var(r, x); break;
    default: sc_0(r);
    }
    memcpy(out, r.v, 32);
}

// sum_{i < 2^lg} x^i as the device computes it (rp_sum_of_powers_pow2: src/util.rs:240-256 for powers of two)
void h_sum_of_powers_pow2(const uint8_t *x, uint32_t lg, uint8_t *out) {
    sc xs, r; memcpy(xs.v, x, 32);
    sc28 xm, rm; sc_to_mont28(xm, xs);
    rp_sum_of_powers_pow2(rm, xm, lg);
    sc_from_mont28(r, rm);
    memcpy(out, r.v, 32);
}

static int g_horner_lanes = 4;
void h_set_horner_lanes(int lanes) { g_horner_lanes = lanes; }
void h_recode32(const uint8_t *s32, int32_t *digits51) {
    uint32_t w[8], r[8];
    memcpy(w, s32, 32);
    sc_recode32(r, w);
    for (uint32_t i = 0; i < BP_VB5_WINDOWS; i++) digits51[i] = sc_digit32(r, i);
}
static int g_radix5 = 0, g_a_outside = 0;   // the wide chains' forms (per-proof verification only): radix-32 window sums + chain; A added after the chain
void h_set_radix5(int on) { g_radix5 = on; }
void h_set_a_outside(int on) { g_a_outside = on; }
// coalesced launches (rp_seg, bpgpu_pool_*): when set, h_rp_verify reads its inputs through a segment table whose
// items live in separate buffers (every second one without rng bytes of its own) and reports through it
static int g_defer_emit = 0;
void h_set_defer_emit(int on) { g_defer_emit = on; }
static int g_coop_split = 0;   // the narrow chain's split scalar role (rp_split_*, rp_rows_thread); per-proof verification only
void h_set_coop_split(int on) { g_coop_split = on; }
static int g_narrow_hi = 0;    // very narrow chains: second tables of the points' 2^128 multiples (hw_point_shift), 32-window chain; with horner_lanes 64 only
void h_set_narrow_hi(int on) { g_narrow_hi = on; }
static std::vector<uint32_t> g_seg_sizes;
void h_set_segments(uint32_t count, const uint32_t *sizes) { g_seg_sizes.assign(sizes, sizes + count); }
// ... and item i verifies under label i mod count (labels of ONE length: they share every transcript position, rp_seg::init_w)
static std::vector<std::vector<uint8_t>> g_seg_labels;
void h_set_segment_labels(uint32_t count, const uint8_t *labels, uint32_t label_len) {
    g_seg_labels.clear();
    for (uint32_t i = 0; i < count; i++) g_seg_labels.emplace_back(labels + (size_t)i * label_len, labels + (size_t)(i + 1) * label_len);
}

// Whole verification pipeline, lane by lane, in the launch structure of the HIP runtime:
//   launch 1: rp_transcript + rp_expand_a  ||  rp_points
//   launch 2: rp_expand_b  ||  vb_window
//   launch 3: fb_accum  ||  (column sums + wavefront Horner)
//   launch 4: finish8
// weights64 != nullptr: the batch-combination pipeline (rlc.h) -- verdict_out as bpgpu_rangeproof_verify_rlc_dev,
// msm_out = 33 bytes (batch verdict, encoding of the combined point)
// device-expanded randomness (rp_shape::seed / seeded, rangeproof.h): when set, launch 1 derives the rng bytes and / or the combination
// weights from this key instead of reading buffers
static uint32_t g_seed_flags = 0;
static uint8_t g_seed[32];
void h_set_chain_seed(const uint8_t *key32, uint32_t flags) {
    g_seed_flags = key32 ? flags : 0;
    if (key32) memcpy(g_seed, key32, 32);
}
// block p of ChaCha20(key, nonce = domain): what the expansion yields for proof p (for the tests' explicit twin)
void h_chain_seed_block(const uint8_t *key32, uint32_t p, uint32_t dom, uint8_t *out64) {
    uint32_t k[8], w[16];
    memcpy(k, key32, 32);
    chacha20_block(k, (uint64_t)p, dom, 0u, w);
    memcpy(out64, w, 64);
}
static int rp_verify_impl(uint32_t W, uint32_t nsplit, uint32_t gens_capacity, uint32_t party_capacity, const uint8_t *gens /*Bb,B,G..,H..*/,
                uint32_t n, uint32_t m, uint32_t nbatch, const uint8_t *proofs, uint32_t proof_len, const uint8_t *commitments,
                const uint8_t *label, uint32_t label_len, const uint8_t *rng64, uint8_t *verdict_out, uint8_t *msm_out,
                const uint8_t *weights64, bool rlc) {
    uint32_t k = 0; while ((1u << k) < n * m) k++;
    uint32_t lg_m = 0; while ((1u << lg_m) < m) lg_m++;
    rp_shape sh; sh.n = n; sh.m = m; sh.nm = n * m; sh.k = k; sh.U = 4 + 2 * k + m; sh.proof_len = proof_len; sh.nproofs = nbatch; sh.shape_verdict = 0;
    const bool r5 = g_radix5 && !rlc, a_out = g_a_outside && !rlc;
    if (g_seed_flags) {   // device-expanded randomness (rp_shape::seed): the buffers it replaces are withheld from launch 1
        sh.seeded = g_seed_flags;
        memcpy(sh.seed, g_seed, 32);
        if (g_seed_flags & RP_SEED_RNG) rng64 = nullptr;
        if (g_seed_flags & RP_SEED_WEIGHTS) weights64 = nullptr;
    }
    sh.radix5 = r5 ? 1u : 0u;
    sh.a_outside = a_out ? 1u : 0u;
    if (proof_len != 32 * (9 + 2 * k)) return -1;
    fb_params prm; prm.W = W; prm.nwin = fb_nwin(W); prm.half = 1u << (W - 1); prm.n_gens = 2 + 2 * gens_capacity * party_capacity;
    std::vector<ge_ext> base((size_t)prm.n_gens * prm.nwin);
    std::vector<fb_entry> table((size_t)prm.n_gens * prm.nwin * prm.half);
    uint32_t bad = 0;
    for (uint32_t g = 0; g < prm.n_gens; g++) fb_base_thread(g, prm, (const uint32_t *)gens, base.data(), &bad);
    if (bad) return -5;
    for (uint32_t t = 0; t < prm.n_gens * prm.nwin; t++) fb_fill_thread(t, prm, base.data(), table.data());
    for (uint64_t gq = 0; gq < (table.size() + BP_FB_NORM_GROUP - 1) / BP_FB_NORM_GROUP; gq++) fb_norm_thread(gq, table.size(), table.data());
    std::vector<uint32_t> ids; ids.push_back(0); ids.push_back(1);
    const uint32_t tot = gens_capacity * party_capacity;
    for (uint32_t j = 0; j < m; j++) for (uint32_t i = 0; i < n; i++) ids.push_back(2 + j * gens_capacity + i);
    for (uint32_t j = 0; j < m; j++) for (uint32_t i = 0; i < n; i++) ids.push_back(2 + tot + j * gens_capacity + i);
    const uint32_t n_gen_terms = 2 * n * m + 2, npairs = n_gen_terms * prm.nwin;

    rp_strobe_init init; make_strobe_init(init, label, label_len, n, m);
    const rp_fields fl = rp_field_layout(k, m);
    const uint32_t t0 = nbatch * sh.U;
    std::vector<uint32_t> fields((size_t)fl.count * nbatch * BP_RP_REC + 8), rec((size_t)t0 * 8 + 8, 0xdeadbeefu), status(nbatch + 1, 0), outw((size_t)nbatch * 8 + 1);
    std::vector<fb_digit> digits((size_t)npairs * nbatch + 1, 0xffff);   // poison: unwritten rows must be masked
    std::vector<ge_cached> tab((size_t)t0 * (r5 ? 16 : 8) + 1);
    // coalesced launch: scatter the inputs into per-item buffers and hide the contiguous ones
    std::vector<rp_seg> segs;
    std::vector<std::vector<uint8_t>> seg_bufs;
    std::vector<std::vector<uint8_t>> seg_verdict;
    std::vector<std::vector<uint32_t>> seg_msm;
    std::deque<rp_strobe_init> seg_inits;
    if (!g_seg_sizes.empty() && !rlc) {
        uint32_t first = 0;
        for (size_t i = 0; i < g_seg_sizes.size() && first < nbatch; i++) {
            const uint32_t cnt = (i + 1 == g_seg_sizes.size() || first + g_seg_sizes[i] > nbatch) ? nbatch - first : g_seg_sizes[i];
            if (cnt == 0) continue;
            rp_seg sg;
            seg_bufs.emplace_back(proofs + (size_t)first * proof_len, proofs + (size_t)(first + cnt) * proof_len);
            sg.proofs = seg_bufs.back().data();
            seg_bufs.emplace_back(commitments + (size_t)first * m * 32, commitments + (size_t)(first + cnt) * m * 32);
            sg.commitments = seg_bufs.back().data();
            if (i & 1) sg.rng64 = nullptr;
            else {
                if (rng64) {
                    seg_bufs.emplace_back(rng64 + (size_t)first * 64, rng64 + (size_t)(first + cnt) * 64);
                    sg.rng64 = seg_bufs.back().data();
                } else {
                    sg.rng64 = nullptr;
                }
            }
            seg_verdict.emplace_back(cnt, 0xee);
            seg_msm.emplace_back((size_t)cnt * 8, 0xeeeeeeeeu);
            sg.init_w = nullptr;
            if (!g_seg_labels.empty()) {   // items that differ in their label (all of one length): own start state per item
                const std::vector<uint8_t> &lb = g_seg_labels[segs.size() % g_seg_labels.size()];
                seg_inits.emplace_back();
                make_strobe_init(seg_inits.back(), lb.data(), (uint32_t)lb.size(), n, m);
                if (seg_inits.back().pos != init.pos || seg_inits.back().pos_begin != init.pos_begin || seg_inits.back().cur_flags != init.cur_flags) return -6;
            }
            sg.first = first;
            sg.count = cnt;
            segs.push_back(sg);
            first += cnt;
        }
        for (size_t i = 0; i < segs.size(); i++) {
            segs[i].verdict = seg_verdict[i].data();
            segs[i].msm_out = seg_msm[i].data();
            if (!seg_inits.empty()) segs[i].init_w = seg_inits[i].w;
        }
    }
    rp_seg_tab segtab;
    memset(&segtab, 0, sizeof segtab);
    segtab.n = (uint32_t)segs.size();
    if (segs.size() <= RP_SEG_INLINE) memcpy(segtab.in, segs.data(), segs.size() * sizeof(rp_seg));   // the inline form of the argument block
    else segtab.ext = segs.data();
    const rp_seg_tab *sgp = segs.empty() ? nullptr : &segtab;   // (segtab.n == 0: contiguous launch)
    const uint8_t *proofs_l1 = sgp ? nullptr : proofs, *coms_l1 = sgp ? nullptr : commitments;
    // launch 1
    for (uint32_t p = 0; p < nbatch; p++) {
        uint32_t stw[50]; kstate st; st.w = stw; st.stride = 1;
        rp_transcript_thread(p, sh, init, st, rp_resolve(p, sh, proofs_l1, coms_l1, rng64, segtab), fields.data(), status.data());
        if (g_coop_split && !rlc && sh.k < 32) {   // the narrow chain's split form, phase by phase as k_rp_stage1_coop runs it (launch 2 below adds rp_rows_thread)
            std::vector<sc28> slots(RP_DEFER_CAP + 1);
            uint32_t meta[2] = {0, 0}, park[32 * 8], go = 0;
            const bool defer = g_defer_emit && sh.U <= RP_DEFER_CAP;
            rp_defer df; df.slot = slots.data(); df.meta = meta;
            rp_split sp; sp.park = park; sp.go = &go;
            go = status[p] == 0 ? 1u : 0u;   // (on the device the lanes that reduce the challenges park y and the u_i: rp_coop_reduce_lane, h_rp_transcript_compare)
            if (go) {
                const rp_fields fl = rp_field_layout(sh.k, sh.m);
                for (uint32_t i = 0; i <= sh.k; i++) {
                    sc v; rp_load(v, fields.data(), nbatch, i < sh.k ? fl.u + i : (uint32_t)RPF_Y, p);
                    for (int q = 0; q < 8; q++) park[8 * i + q] = v.v[q];
                }
            }
            for (uint32_t lane = 0; lane < 32; lane++) rp_split_invert_lane(lane, p, sh, fields.data(), rec.data(), sp, defer ? &df : nullptr);
            rp_expand_a_thread(p, sh, prm, lg_m, fields.data(), rec.data(), digits.data(), status.data(), nullptr, 0, defer ? &df : nullptr, RP_SKIP_INV | RP_SKIP_ROWS);
            if (defer) for (uint32_t lane = 0; lane < 32; lane++) rp_emit_deferred(lane, p, sh, rec.data(), df, 0);
        } else
        if (g_defer_emit && sh.U <= RP_DEFER_CAP) {   // the narrow chain's form: the leader parks the coefficients, 32 lanes recode them (rp_defer)
            std::vector<sc28> slots(RP_DEFER_CAP + 1);
            uint32_t meta[2] = {0, 0};
            rp_defer df; df.slot = slots.data(); df.meta = meta;
            rp_expand_a_thread(p, sh, prm, lg_m, fields.data(), rec.data(), digits.data(), status.data(), weights64, 0, &df);
            for (uint32_t lane = 0; lane < 32; lane++) rp_emit_deferred(lane, p, sh, rec.data(), df, 0);
        } else
        rp_expand_a_thread(p, sh, prm, lg_m, fields.data(), rec.data(), digits.data(), status.data(), weights64);
    }
    for (uint32_t t = 0; t < t0; t++) rp_points_thread(t, sh, rp_resolve(t / sh.U, sh, proofs_l1, coms_l1, nullptr, segtab), tab.data(), status.data());
    const bool hi = g_narrow_hi && !rlc && !r5 && !a_out && g_horner_lanes == 64;
    const uint32_t hi_lev = hi ? (uint32_t)g_narrow_hi : 1u;   // 2: tables of 2^128 P; 4: of 2^64 P, 2^128 P, 2^192 P
    std::vector<ge_cached> tab_hi(hi ? (size_t)t0 * 8 * (hi_lev - 1) + 1 : 1);
    if (hi) {   // k_rp_stage1_coop's third role: one wavefront per (point, level), Q = 2^(256 lv / levels) P by wavefront-cooperative doublings, then Q's table
      for (uint32_t lv = 1; lv < hi_lev; lv++)
        for (uint32_t t = 0; t < t0; t++) {
            const int nshift = (int)(lv * (256u / hi_lev));
            ge_cached *dst = tab_hi.data() + 8 * ((size_t)(lv - 1) * t0 + t);
            const uint32_t p = t / sh.U, u = t - p * sh.U;
            if (u == 0) continue;
            uint32_t w[8];
            load_words8(w, rp_unique_point_ptr(sh, rp_resolve(p, sh, proofs_l1, coms_l1, nullptr, segtab), u));
            ge_ext pt, q, pt_ref;
            hw_ristretto_decode(pt, w);
            if (ristretto_decompress(pt_ref, w)) {   // (a valid encoding: the wavefront's decode is the lane's, coordinate for coordinate)
                uint32_t a[8], b[8];
                fe_to_words(a, pt.X); fe_to_words(b, pt_ref.X); if (memcmp(a, b, 32)) return -91;
                fe_to_words(a, pt.Y); fe_to_words(b, pt_ref.Y); if (memcmp(a, b, 32)) return -92;
                fe_to_words(a, pt.T); fe_to_words(b, pt_ref.T); if (memcmp(a, b, 32)) return -93;
            }
            hw_point_shift(pt, nshift, &q);
            hw_shift_table8(pt, nshift, dst);
            {   // the wavefront's table holds the same eight points as the lane's (vb_build_table): e Q + entry == (e + 2) Q ... compared as encodings
                ge_cached ref[8];
                vb_build_table(ref, q);
                for (int e = 0; e < 8; e++) {
                    ge_ext a, b, idn; ge_identity(idn);
                    ge_add_cached(a, idn, dst[e], false);
                    ge_add_cached(b, idn, ref[e], false);
                    uint32_t ea[8], eb[8];
                    ristretto_compress(ea, a); ristretto_compress(eb, b);
                    if (memcmp(ea, eb, 32)) return -94;
                }
            }
            // cross-check: the lane-serial doublings give the same point (projectively)
            ge_ext r = pt;
            for (int i = 0; i < nshift; i++) ge_dbl(r, r, true);
            uint32_t e1[8], e2[8];
            ristretto_compress(e1, q); ristretto_compress(e2, r);
            if (memcmp(e1, e2, 32)) return -90;
        }
    }
    // launch 2
    std::vector<uint64_t> acc((size_t)n_gen_terms * 10, 0);
    if (rlc) {
        for (uint32_t tid = 0; tid < (sh.nm / 4) * nbatch; tid++) {
            const uint32_t t4 = tid / nbatch, p = tid - t4 * nbatch;
            sc g[4], h[4]; uint64_t l[10];
            rp_expand_b4_thread(tid, sh, prm, fields.data(), nullptr, status.data(), g, h);
            for (uint32_t j = 0; j < 4; j++) {
                rlc_limbs(l, g[j]); for (int q = 0; q < 10; q++) acc[(size_t)(2 + 4 * t4 + j) * 10 + q] += l[q];
                rlc_limbs(l, h[j]); for (int q = 0; q < 10; q++) acc[(size_t)(2 + sh.nm + 4 * t4 + j) * 10 + q] += l[q];
            }
            if (t4 == 0 && status[p] == 0) {
                for (uint32_t row = 0; row < 2; row++) {
                    sc r; rp_load(r, fields.data(), nbatch, RPF_ROW0 + row, p);
                    rlc_limbs(l, r); for (int q = 0; q < 10; q++) acc[(size_t)row * 10 + q] += l[q];
                }
            }
        }
    } else {
        for (uint32_t tid = 0; tid < (sh.nm / 4) * nbatch; tid++) rp_expand_b4_thread(tid, sh, prm, fields.data(), digits.data(), status.data());
        if (g_coop_split && sh.k < 32)
            for (uint32_t p = 0; p < nbatch; p++) rp_rows_thread(p, sh, prm, lg_m, fields.data(), digits.data(), status.data());
        // the mirrored-pair form of the role (rp_expand_b8_thread, what the device runs since round 6): the same digits, every row
        if (sh.nm >= 8) {
            std::vector<fb_digit> d4(digits), d8(digits);
            const size_t rows0 = (size_t)2 * prm.nwin * nbatch, rows1 = (size_t)(2 + 2 * sh.nm) * prm.nwin * nbatch;
            for (size_t q = rows0; q < rows1 && q < d8.size(); q++) d8[q] = (fb_digit)0x5a5a5a5a;
            for (uint32_t tid = 0; tid < (sh.nm / 8) * nbatch; tid++) rp_expand_b8_thread(tid, sh, prm, fields.data(), d8.data(), status.data());
            for (uint32_t p = 0; p < nbatch; p++) {
                if (status[p] != 0) continue;   // (rows of rejected proofs are never written by either form)
                for (size_t row = 2; row < 2 + 2 * (size_t)sh.nm; row++)
                    for (uint32_t w = 0; w < prm.nwin; w++)
                        if (d8[(row * prm.nwin + w) * nbatch + p] != d4[(row * prm.nwin + w) * nbatch + p]) return -77;
            }
        }
        // ... and the one-index-per-lane form of narrow chains (rp_expand_b1_thread)
        if (sh.nm >= 4) {
            std::vector<fb_digit> d1(digits);
            const size_t rows0 = (size_t)2 * prm.nwin * nbatch, rows1 = (size_t)(2 + 2 * sh.nm) * prm.nwin * nbatch;
            for (size_t q = rows0; q < rows1 && q < d1.size(); q++) d1[q] = (fb_digit)0x5a5a5a5a;
            for (uint32_t tid = 0; tid < sh.nm * nbatch; tid++) rp_expand_b1_thread(tid, sh, prm, fields.data(), d1.data(), status.data());
            for (uint32_t p = 0; p < nbatch; p++) {
                if (status[p] != 0) continue;
                for (size_t row = 2; row < 2 + 2 * (size_t)sh.nm; row++)
                    for (uint32_t w = 0; w < prm.nwin; w++)
                        if (d1[(row * prm.nwin + w) * nbatch + p] != digits[(row * prm.nwin + w) * nbatch + p]) return -78;
            }
        }
    }
    std::vector<vb_chunk> chunks; std::vector<uint32_t> chunk_first(nbatch + 1); uint32_t tt = 0;
    for (uint32_t b = 0; b < nbatch; b++) {
        chunk_first[b] = (uint32_t)chunks.size();
        for (uint32_t kk = 0; kk < sh.U; kk += BP_VB_CHUNK) {
            vb_chunk c; c.msm = b; c.first = tt + kk; c.count = sh.U - kk < BP_VB_CHUNK ? sh.U - kk : BP_VB_CHUNK; c.pad = 0;
            chunks.push_back(c);
        }
        tt += sh.U;
    }
    chunk_first[nbatch] = (uint32_t)chunks.size();
    std::vector<ge_ext> part(chunks.size() * 64 + 1), hq(nbatch + 1);
    const bool quad = g_horner_lanes != 64;
    const bool one_chunk = chunks.size() == nbatch;
    std::vector<ge_cached> colc((size_t)nbatch * 64 + 1);
    if (rlc) {
        // window sums with rejected proofs skipped, then one column sum over all chunks of all proofs (tree as on
        // the device: 16-way while > 64 rows, then 8-way, the last <= 8 rows are added by the Horner wavefront)
        for (uint32_t tid = 0; tid < chunks.size() * 64; tid++)
            vb_window_thread(tid, chunks.data(), tab.data(), rec.data(), part.data(), nullptr, status.data());
        std::vector<ge_ext> cur(part.begin(), part.begin() + chunks.size() * 64), nxt;
        uint32_t rows = (uint32_t)chunks.size();
        while (rows > 8) {
            const uint32_t group = rows > 64 ? 16 : 8, ng = (rows + group - 1) / group;
            nxt.assign((size_t)ng * 64, ge_ext());
            for (uint32_t tid = 0; tid < ng * 64; tid++) fb_reduce_thread(tid, 64, rows, group, cur.data(), nxt.data());
            cur.swap(nxt);
            rows = ng;
        }
        // batch-of-one tail: coefficients mod l -> digits -> table walk -> Horner over the combined column sums -> finish
        std::vector<fb_digit> dig1((size_t)npairs + 1);
        for (uint32_t g = 0; g < n_gen_terms; g++) {
            sc v; rlc_acc_to_sc(v, &acc[(size_t)g * 10]);
            fb_recode(dig1.data() + (size_t)g * prm.nwin, 1, v.v, prm);
        }
        std::vector<ge_ext> partial1((size_t)nsplit + 1), hq1(2);
        const uint32_t per1 = (npairs + nsplit - 1) / nsplit;
        for (uint32_t sp = 0; sp < nsplit; sp++) {
            uint32_t q0 = sp * per1, q1 = q0 + per1 < npairs ? q0 + per1 : npairs; if (q0 > npairs) q0 = npairs;
            fb_accum_thread(0, sp, q0, q1, prm, 1, ids.data(), dig1.data(), table.data(), partial1.data());
        }
        const uint32_t cf[2] = {0, rows};
        hw_colsum_horner_msm(0, cf, cur.data(), &hq1[0]);
        ge_ext x[8];
        for (uint32_t j = 0; j < 8; j++) shared_finish8_gather(x[j], 0, j, 1, nsplit, hq1.data(), partial1.data());
        for (uint32_t step = 4; step >= 1; step >>= 1)
            for (uint32_t j = 0; j < step; j++) { const ge_ext q = x[j + step]; ge_add(x[j], x[j], q); }
        uint32_t zero_status = 0, rw[8]; uint8_t bv = 0;
        shared_finish_tail(0, x[0], &zero_status, rw, &bv);
        for (uint32_t p = 0; p < nbatch; p++) verdict_out[p] = status[p] ? (uint8_t)status[p] : (bv ? (uint8_t)BP_VERDICT_UNDECIDED : (uint8_t)0);
        if (msm_out) { msm_out[0] = bv; memcpy(msm_out + 1, rw, 32); }
        return 0;
    }
    if (r5)
        for (uint32_t tid = 0; tid < nbatch * BP_VB5_WINDOWS; tid++) vb_window_wide_thread<true>(tid, sh.U, a_out ? 1 : 0, tab.data(), rec.data(), colc.data());
    else if (a_out)
        for (uint32_t tid = 0; tid < nbatch * BP_VB_WINDOWS; tid++) vb_window_wide_thread<false>(tid, sh.U, 1, tab.data(), rec.data(), colc.data());
    else
    for (uint32_t tid = 0; tid < chunks.size() * 64; tid++)
        vb_window_thread(tid, chunks.data(), tab.data(), rec.data(), part.data(), (quad && one_chunk) ? colc.data() : nullptr, nullptr, hi ? tab_hi.data() : nullptr, hi_lev, (uint64_t)8 * t0);
    if (quad && !one_chunk && !r5 && !a_out)
        for (uint32_t tid = 0; tid < nbatch * 64; tid++) vb_colsum_thread(tid, chunk_first.data(), part.data(), nullptr, nullptr, colc.data());
    // launch 3
    std::vector<ge_ext> partial((size_t)nsplit * nbatch + 1);
    const uint32_t per = (npairs + nsplit - 1) / nsplit;
    for (uint32_t sp = 0; sp < nsplit; sp++) {
        uint32_t q0 = sp * per, q1 = q0 + per < npairs ? q0 + per : npairs; if (q0 > npairs) q0 = npairs;
        for (uint32_t p = 0; p < nbatch; p++) fb_accum_thread(p, sp, q0, q1, prm, nbatch, ids.data(), digits.data(), table.data(), partial.data());
    }
    for (uint32_t b = 0; b < nbatch; b++) {
        if (r5) vb_horner_wide_thread<true>(b, nbatch, colc.data(), a_out ? tab.data() : nullptr, 16ull * sh.U, hq.data());
        else if (a_out) vb_horner_wide_thread<false>(b, nbatch, colc.data(), tab.data(), 8ull * sh.U, hq.data());
        else if (g_horner_lanes == 1) vb_horner_cached_thread(b, nbatch, colc.data(), hq.data());
        else if (quad) hq_horner_msm(b, nbatch, colc.data(), hq.data());
        else hw_colsum_horner_msm(b, chunk_first.data(), part.data(), &hq[b], (int)hi_lev);
    }
    std::vector<uint8_t> verdict(nbatch + 1);
    // launch 4 (k_finish8): 8 lanes per proof gather, 3-level fold, lane 0 finishes
    for (uint32_t p = 0; p < nbatch; p++) {
        ge_ext x[8];
        for (uint32_t j = 0; j < 8; j++) shared_finish8_gather(x[j], p, j, nbatch, nsplit, hq.data(), partial.data());
        for (uint32_t step = 4; step >= 1; step >>= 1)
            for (uint32_t j = 0; j < step; j++) { const ge_ext q = x[j + step]; ge_add(x[j], x[j], q); }
        if (sgp) {
            const rp_seg sg = rp_seg_lookup(*sgp, p);
            shared_finish_tail(p, p - sg.first, x[0], status.data(), sg.msm_out, sg.verdict);
        } else {
            shared_finish_tail(p, x[0], status.data(), outw.data(), verdict.data());
        }
    }
    if (sgp) {   // gather the per-item outputs
        for (size_t i = 0; i < segs.size(); i++) {
            memcpy(verdict.data() + segs[i].first, seg_verdict[i].data(), segs[i].count);
            memcpy(outw.data() + (size_t)segs[i].first * 8, seg_msm[i].data(), (size_t)segs[i].count * 32);
        }
    }
    for (uint32_t p = 0; p < nbatch; p++) verdict_out[p] = verdict[p];
    if (msm_out) memcpy(msm_out, outw.data(), (size_t)nbatch * 32);
    return 0;
}
int h_rp_verify(uint32_t W, uint32_t nsplit, uint32_t gens_capacity, uint32_t party_capacity, const uint8_t *gens, uint32_t n, uint32_t m,
                uint32_t nbatch, const uint8_t *proofs, uint32_t proof_len, const uint8_t *commitments, const uint8_t *label,
                uint32_t label_len, const uint8_t *rng64, uint8_t *verdict_out, uint8_t *msm_out) {
    return rp_verify_impl(W, nsplit, gens_capacity, party_capacity, gens, n, m, nbatch, proofs, proof_len, commitments, label, label_len, rng64,
                          verdict_out, msm_out, nullptr, false);
}
int h_rp_verify_rlc(uint32_t W, uint32_t nsplit, uint32_t gens_capacity, uint32_t party_capacity, const uint8_t *gens, uint32_t n, uint32_t m,
                    uint32_t nbatch, const uint8_t *proofs, uint32_t proof_len, const uint8_t *commitments, const uint8_t *label,
                    uint32_t label_len, const uint8_t *rng64, const uint8_t *weights64, uint8_t *verdict_out, uint8_t *batch_out) {
    return rp_verify_impl(W, nsplit, gens_capacity, party_capacity, gens, n, m, nbatch, proofs, proof_len, commitments, label, label_len, rng64,
                          verdict_out, batch_out, weights64, true);
}
// window recoding of one scalar: returns nwin, digits[win] = unsigned W-bit value (real digit + half)
uint32_t h_fb_recode(uint32_t W, const uint8_t *scalar, uint32_t *digits_out) {
    fb_params prm; prm.W = W; prm.nwin = fb_nwin(W); prm.half = 1u << (W - 1); prm.n_gens = 0;
    uint32_t s[8]; memcpy(s, scalar, 32);
    std::vector<fb_digit> d(prm.nwin);
    fb_recode(d.data(), 1, s, prm);
    for (uint32_t i = 0; i < prm.nwin; i++) digits_out[i] = d[i];
    return prm.nwin;
}
// sum of `count` canonical scalars through the limb accumulator (rlc.h)
void h_rlc_sum(uint32_t count, const uint8_t *scalars, uint8_t *out) {
    uint64_t acc[10] = {0};
    for (uint32_t i = 0; i < count; i++) {
        sc v; memcpy(v.v, scalars + 32 * (size_t)i, 32);
        uint64_t l[10]; rlc_limbs(l, v);
        for (int q = 0; q < 10; q++) acc[q] += l[q];
    }
    sc r; rlc_acc_to_sc(r, acc);
    memcpy(out, r.v, 32);
}

void h_msm_vb(uint32_t nbatch, const uint32_t *n_terms, const uint8_t *scalars, const uint8_t *points, uint8_t *out, uint8_t *status_out);

// Stand-alone IPP verification, lane by lane: ipp_prepare -> variable-base MSM pipeline -> verdict
int h_ipp_verify(uint32_t n, uint32_t nbatch, const uint8_t *proofs, uint32_t proof_len, const uint8_t *label, uint32_t label_len,
                 const uint8_t *Gf, const uint8_t *Hf, const uint8_t *P, const uint8_t *Q, const uint8_t *G, const uint8_t *H,
                 uint8_t *verdict_out, uint8_t *msm_out) {
    if (proof_len % 32 || proof_len / 32 < 2 || (proof_len / 32 - 2) % 2) return -1;
    uint32_t k = (proof_len / 32 - 2) / 2;
    ipp_shape sh; sh.bases_shared = 0; sh.n = n; sh.k = k; sh.proof_len = proof_len; sh.nproofs = nbatch;
    sh.shape_verdict = (n == (1u << k)) ? 0 : BP_VERDICT_VERIFICATION;
    if (sh.shape_verdict) sh.n = 0;
    sh.N = 2 * sh.n + 2 * (sh.shape_verdict ? 0 : k) + 2;
    rp_strobe_init init;
    {
        kstate st; st.w = init.w; st.stride = 1; strobe t; merlin_strobe_init(t, st);
        const uint8_t dom[7] = {'d','o','m','-','s','e','p'}, ipp[6] = {'i','p','p',' ','v','1'}, ln[1] = {'n'};
        merlin_append_message(t, dom, 7, label, label_len); merlin_append_message(t, dom, 7, ipp, 6); merlin_append_u64(t, ln, 1, n);
        init.pos = t.pos; init.pos_begin = t.pos_begin; init.cur_flags = t.cur_flags;
    }
    std::vector<uint32_t> scal((size_t)nbatch * sh.N * 8 + 8, 0), pts((size_t)nbatch * sh.N * 8 + 8, 0), status(nbatch + 1, 0), nt(nbatch, sh.N);
    for (uint32_t p = 0; p < nbatch; p++) {
        uint32_t stw[50]; kstate st; st.w = stw; st.stride = 1;
        ipp_prepare_thread(p, sh, init, st, proofs, Gf, Hf, P, Q, G, H, scal.data(), pts.data(), status.data());
    }
    std::vector<uint8_t> out((size_t)nbatch * 32 + 32), mst(nbatch + 1);
    h_msm_vb(nbatch, nt.data(), (const uint8_t *)scal.data(), (const uint8_t *)pts.data(), out.data(), mst.data());
    for (uint32_t p = 0; p < nbatch; p++) ipp_verdict_thread(p, status.data(), mst.data(), (const uint32_t *)out.data(), verdict_out);
    if (msm_out) memcpy(msm_out, out.data(), (size_t)nbatch * 32);
    return 0;
}

// LinearProof verification, lane by lane: lin_prepare -> variable-base MSM pipeline -> verdict (linear.h)
int h_lin_verify(uint32_t n, uint32_t nbatch, const uint8_t *proofs, uint32_t proof_len, const uint8_t *label, uint32_t label_len,
                 const uint8_t *Cc, const uint8_t *G, const uint8_t *F, const uint8_t *B, const uint8_t *b, uint32_t b_shared,
                 uint8_t *verdict_out, uint8_t *msm_out) {
    if (proof_len % 32 || proof_len / 32 < 3 || (proof_len / 32 - 3) % 2) return -1;
    uint32_t k = (proof_len / 32 - 3) / 2;
    lin_shape sh; sh.b_shared = b_shared; sh.n = n; sh.k = k; sh.proof_len = proof_len; sh.nproofs = nbatch; sh.fixed = 0;
    sh.shape_verdict = (k < 32 && n == (1u << k)) ? 0 : BP_VERDICT_VERIFICATION;
    if (sh.shape_verdict) sh.n = 0;
    sh.N = sh.shape_verdict ? 4 : n + 2 * k + 4;
    rp_strobe_init init;
    {
        kstate st; st.w = init.w; st.stride = 1; strobe t; merlin_strobe_init(t, st);
        const uint8_t dom[7] = {'d','o','m','-','s','e','p'}, ipp[6] = {'i','p','p',' ','v','1'}, ln[1] = {'n'};
        merlin_append_message(t, dom, 7, label, label_len); merlin_append_message(t, dom, 7, ipp, 6); merlin_append_u64(t, ln, 1, n);
        init.pos = t.pos; init.pos_begin = t.pos_begin; init.cur_flags = t.cur_flags;
    }
    std::vector<uint32_t> scal((size_t)nbatch * sh.N * 8 + 8, 0), pts((size_t)nbatch * sh.N * 8 + 8, 0), status(nbatch + 1, 0), nt(nbatch, sh.N);
    for (uint32_t p = 0; p < nbatch; p++) {
        uint32_t stw[50]; kstate st; st.w = stw; st.stride = 1;
        lin_prepare_thread(p, sh, init, st, proofs, Cc, b, G, F, B, scal.data(), pts.data(), status.data());
    }
    std::vector<uint8_t> out((size_t)nbatch * 32 + 32), mst(nbatch + 1);
    h_msm_vb(nbatch, nt.data(), (const uint8_t *)scal.data(), (const uint8_t *)pts.data(), out.data(), mst.data());
    for (uint32_t p = 0; p < nbatch; p++) ipp_verdict_thread(p, status.data(), mst.data(), (const uint32_t *)out.data(), verdict_out);
    if (msm_out) memcpy(msm_out, out.data(), (size_t)nbatch * 32);
    return 0;
}

// The same with G, F, B = the loaded generators (generator-table mode of lin_prepare_thread): gens = [B_blinding, B, G_0..G_{n-1}]
// encodings; the n + 2 generator coefficients go through the fixed-base table emulation (h_msm_shared), C, L_j, R_j, S through the lists.
int h_lin_verify_fixed(uint32_t W, uint32_t n, uint32_t nbatch, const uint8_t *proofs, uint32_t proof_len, const uint8_t *label, uint32_t label_len,
                       const uint8_t *Cc, const uint8_t *gens, const uint8_t *b, uint32_t b_shared, uint8_t *verdict_out, uint8_t *msm_out) {
    if (proof_len % 32 || proof_len / 32 < 3 || (proof_len / 32 - 3) % 2) return -1;
    uint32_t k = (proof_len / 32 - 3) / 2;
    if (n != (1u << k)) return -2;
    lin_shape sh; sh.b_shared = b_shared; sh.n = n; sh.k = k; sh.proof_len = proof_len; sh.nproofs = nbatch; sh.shape_verdict = 0; sh.fixed = 1;
    sh.N = 2 * k + 2;
    rp_strobe_init init;
    {
        kstate st; st.w = init.w; st.stride = 1; strobe t; merlin_strobe_init(t, st);
        const uint8_t dom[7] = {'d','o','m','-','s','e','p'}, ipp[6] = {'i','p','p',' ','v','1'}, ln[1] = {'n'};
        merlin_append_message(t, dom, 7, label, label_len); merlin_append_message(t, dom, 7, ipp, 6); merlin_append_u64(t, ln, 1, n);
        init.pos = t.pos; init.pos_begin = t.pos_begin; init.cur_flags = t.cur_flags;
    }
    std::vector<uint32_t> scal((size_t)nbatch * sh.N * 8 + 8, 0), pts(scal.size(), 0), status(nbatch + 1, 0), grows((size_t)nbatch * (n + 2) * 8 + 8, 0);
    for (uint32_t p = 0; p < nbatch; p++) {
        uint32_t stw[50]; kstate st; st.w = stw; st.stride = 1;
        lin_prepare_thread(p, sh, init, st, proofs, Cc, b, gens + 64, gens + 32, gens, scal.data(), pts.data(), status.data(), nullptr, grows.data());
    }
    std::vector<uint32_t> ids(n + 2);
    for (uint32_t i = 0; i < n + 2; i++) ids[i] = i;
    std::vector<uint8_t> out((size_t)nbatch * 32 + 32), mst(nbatch + 1), vd(nbatch + 1);
    int rc = h_msm_shared(W, 2, n + 2, gens, n + 2, ids.data(), nbatch, sh.N, (const uint8_t *)grows.data(), (const uint8_t *)scal.data(), (const uint8_t *)pts.data(),
                          out.data(), mst.data(), vd.data());
    if (rc) return rc;
    for (uint32_t p = 0; p < nbatch; p++) ipp_verdict_thread(p, status.data(), mst.data(), (const uint32_t *)out.data(), verdict_out);
    if (msm_out) memcpy(msm_out, out.data(), (size_t)nbatch * 32);
    return 0;
}

// ProofShare::audit_share, lane by lane (audit.h): front end -> the two MSMs per share through the variable-base pipeline -> verdict.
// gens: [B_blinding, B, G (party-major, capacity each), H (...)] encodings.
int h_audit_shares(uint32_t n, uint32_t nshares, uint32_t gens_capacity, uint32_t party_capacity, const uint8_t *gens, const uint32_t *party,
                   const uint8_t *shares, const uint8_t *bitc, const uint8_t *polyc, const uint8_t *chal, uint32_t chal_shared, uint8_t *verdict_out,
                   uint8_t *checks_out) {
    aud_shape sh; sh.n = n; sh.lg_n = 0; while ((1u << sh.lg_n) < n) sh.lg_n++;
    sh.nshares = nshares; sh.gens_capacity = gens_capacity; sh.party_capacity = party_capacity; sh.chal_shared = chal_shared;
    const uint32_t per = 2 * n + 8;
    std::vector<uint32_t> scal((size_t)nshares * per * 8 + 8, 0), pts(scal.size(), 0), status(nshares + 1, 0), nt(2 * nshares);
    for (uint32_t s = 0; s < nshares; s++) {
        aud_prepare_thread(s, sh, party, shares, bitc, polyc, chal, (const uint32_t *)gens, scal.data(), pts.data(), status.data());
        nt[2 * s] = 2 * n + 3; nt[2 * s + 1] = 5;
    }
    std::vector<uint8_t> out((size_t)nshares * 64 + 64), mst(2 * nshares + 1);
    h_msm_vb(2 * nshares, nt.data(), (const uint8_t *)scal.data(), (const uint8_t *)pts.data(), out.data(), mst.data());
    for (uint32_t s = 0; s < nshares; s++) aud_verdict_thread(s, status.data(), mst.data(), (const uint32_t *)out.data(), verdict_out);
    if (checks_out) memcpy(checks_out, out.data(), (size_t)nshares * 64);
    return 0;
}

// (tests) limit of a lane's bucket chain (>= 33): 0 = bk_chain_lim; how many buckets the last h_msm_bucket call sent through the heavy pass
static uint32_t g_bk_cap = 0, g_bk_heavy_buckets = 0, g_bk_groups = 0;
void h_set_bucket_cap(uint32_t cap) { g_bk_cap = cap; }
void h_set_bucket_groups(uint32_t g) { g_bk_groups = g; }
uint32_t h_bucket_heavy_count() { return g_bk_heavy_buckets; }
// The bucket (Pippenger) MSM pipeline, lane by lane and phase by phase (bucket.h; the same bodies as k_bucket.hip).
// c = 8 or 12.  skip_div != 0: ONE MSM over all terms, terms of "proof" t / skip_div are left out when skip[proof] != 0.
int h_msm_bucket(uint32_t nbatch, const uint32_t *n_terms, const uint8_t *scalars, const uint8_t *points, uint32_t c,
                 const uint32_t *skip, uint32_t skip_div, uint32_t nsub, uint8_t *out, uint8_t *status_out) {
    const bk_params prm = bk_make(c);
    std::vector<uint32_t> msm_first(nbatch + 1, 0);
    for (uint32_t b = 0; b < nbatch; b++) msm_first[b + 1] = msm_first[b] + n_terms[b];
    const uint32_t total = msm_first[nbatch];
    const bool single = skip_div != 0;
    const uint32_t nmsm = single ? 1 : nbatch, nbw = nmsm * prm.nwin;
    std::vector<fb_entry> pts(total + 1);
    std::vector<uint32_t> rwords((size_t)total * BK_RWORDS + BK_RWORDS), status(nbatch + 1, 0), idx((size_t)prm.nwin * total + 1, 0xdeadbeefu);
    for (uint32_t t = 0; t < total; t++)
        bk_prepare_thread(t, nbatch, msm_first.data(), (const uint32_t *)scalars, (const uint32_t *)points, pts.data(), rwords.data(), status.data(), prm);
    std::vector<bk_desc> desc((size_t)nbw * prm.half);
    std::vector<uint32_t> l_cnt(prm.half), l_off(prm.half), l_part(prm.lanes), l_hist2(256);
    bk_lds l; l.cnt = l_cnt.data(); l.off = l_off.data(); l.part = l_part.data(); l.hist2 = l_hist2.data();
    for (uint32_t bw = 0; bw < nbw; bw++) {
        const uint32_t b = bw / prm.nwin;
        bk_seg sg; sg.w = bw - b * prm.nwin; sg.first = single ? 0 : msm_first[b]; sg.count = single ? total : msm_first[b + 1] - sg.first;
        sg.skip_status = single ? skip : nullptr; sg.skip_div = skip_div ? skip_div : 1;
        uint32_t *idx_w = idx.data() + (size_t)sg.w * total;
        sg.sub = 0; sg.nsub = 1;
        if (nsub <= 1) {
            for (uint32_t lane = 0; lane < prm.lanes; lane++) bk_sort_p0(lane, prm, l);
            for (uint32_t lane = 0; lane < prm.lanes; lane++) bk_sort_p1(lane, prm, sg, rwords.data(), l);
            for (uint32_t lane = 0; lane < prm.lanes; lane++) bk_sort_p2(lane, prm, l);
            for (uint32_t lane = 0; lane < prm.lanes; lane++) bk_sort_p3(lane, prm, l);
            for (uint32_t lane = 0; lane < prm.lanes; lane++) bk_sort_p4(lane, prm, l);
            for (uint32_t lane = 0; lane < prm.lanes; lane++) bk_sort_p5(lane, l);
            for (uint32_t lane = 0; lane < prm.lanes; lane++) bk_sort_p6(lane, prm, sg, l, desc.data() + (size_t)bw * prm.half);
            for (uint32_t lane = 0; lane < prm.lanes; lane++) bk_sort_p7(lane, prm, sg, rwords.data(), l, idx_w);
        } else {   // the split sort of large MSMs: nsub workgroups histogram / scatter, one scans (k_bk_sort_big)
            std::vector<uint32_t> gcnt(prm.half, 0), gcur(prm.half, 0);
            for (uint32_t sub = 0; sub < nsub; sub++) {
                sg.sub = sub; sg.nsub = nsub;
                for (uint32_t lane = 0; lane < prm.lanes; lane++) bk_sort_p0(lane, prm, l);
                for (uint32_t lane = 0; lane < prm.lanes; lane++) bk_sort_p1(lane, prm, sg, rwords.data(), l);
                for (uint32_t lane = 0; lane < prm.lanes; lane++) bk_sort_merge(lane, prm, l, gcnt.data());
            }
            sg.sub = 0; sg.nsub = 1;
            for (uint32_t lane = 0; lane < prm.lanes; lane++) bk_sort_load(lane, prm, l, gcnt.data());
            for (uint32_t lane = 0; lane < prm.lanes; lane++) bk_sort_p2(lane, prm, l);
            for (uint32_t lane = 0; lane < prm.lanes; lane++) bk_sort_p3(lane, prm, l);
            for (uint32_t lane = 0; lane < prm.lanes; lane++) bk_sort_p4(lane, prm, l);
            for (uint32_t lane = 0; lane < prm.lanes; lane++) bk_sort_p5(lane, l);
            for (uint32_t lane = 0; lane < prm.lanes; lane++) bk_sort_p6(lane, prm, sg, l, desc.data() + (size_t)bw * prm.half);
            for (uint32_t lane = 0; lane < prm.lanes; lane++) bk_sort_publish(lane, prm, l, gcur.data());
            bk_lds lg = l; lg.cnt = gcur.data();
            for (uint32_t sub = 0; sub < nsub; sub++) {
                sg.sub = sub; sg.nsub = nsub;
                for (uint32_t lane = 0; lane < prm.lanes; lane++) bk_sort_p7(lane, prm, sg, rwords.data(), lg, idx_w);
            }
        }
        // descriptors are sorted by population (clamped at 255), descending, and partition the window's listed terms
        uint32_t listed = 0;
        for (uint32_t r = 0; r < prm.half; r++) {
            const bk_desc &d = desc[(size_t)bw * prm.half + r];
            if (r && std::min(d.cnt, 255u) > std::min(desc[(size_t)bw * prm.half + r - 1].cnt, 255u)) return -3;
            listed += d.cnt;
        }
        if (listed > sg.count) return -4;
    }
    std::vector<ge_ext> bsum((size_t)nbw * prm.half);
    uint32_t per_max = 0;
    for (uint32_t b = 0; b < nbatch; b++) per_max = std::max(per_max, n_terms[b]);
    const uint32_t lim = g_bk_cap ? g_bk_cap : bk_chain_lim(single ? total : per_max, prm);
    for (uint32_t tid = 0; tid < nbw * prm.half; tid++) {
        const uint32_t bw = tid / prm.half, r = tid - bw * prm.half, w = bw % prm.nwin;
        bk_accum_thread(bw, r, prm, desc.data(), idx.data() + (size_t)w * total, pts.data(), bsum.data(), lim);
    }
    // the heavy pass (k_bk_heavy), wavefront by wavefront and phase by phase
    g_bk_heavy_buckets = 0;
    const uint32_t G = g_bk_groups ? g_bk_groups : bk_heavy_groups(prm, nmsm);
    for (uint32_t blk = 0; blk < nbw * G; blk++) {
        const uint32_t bw = blk / G, g = blk - bw * G;
        uint32_t hn[1]; std::vector<uint32_t> hlist(BK_HEAVY_MAX); std::vector<ge_ext> xch(64);
        bk_heavy_lds hl; hl.n = hn; hl.list = hlist.data(); hl.xch = xch.data();
        const uint32_t *idx_w = idx.data() + (size_t)(bw % prm.nwin) * total;
        for (uint32_t lane = 0; lane < 64; lane++) bk_heavy_h0(lane, hl);
        for (uint32_t lane = 0; lane < 64; lane++) bk_heavy_h1(lane, bw, g, G, prm, desc.data(), lim, hl);
        const uint32_t hc = hn[0];
        if (hc > BK_HEAVY_MAX) return -6;
        g_bk_heavy_buckets += hc;
        for (uint32_t i = 0; i < hc; i++) {
            for (uint32_t lane = 0; lane < 64; lane++) bk_heavy_h2(lane, bw, i, prm, desc.data(), lim, idx_w, pts.data(), hl);
            for (uint32_t step = 32; step >= 1; step >>= 1)
                for (uint32_t lane = 0; lane < 64; lane++) bk_heavy_h3(lane, step, hl);
            for (uint32_t lane = 0; lane < 64; lane++) bk_heavy_h4(lane, bw, i, prm, desc.data(), hl, bsum.data());
        }
    }
    std::vector<uint32_t> colq16((size_t)nmsm * 64 * 32 + 32, 0);
    // running-sum tree: leaf level lane by lane, then the packed upper levels exactly as k_bk_tree walks them
    const uint32_t nl = bk_leaves(prm);
    std::vector<ge_ext> gS((size_t)nbw * nl), gA((size_t)nbw * nl);
    for (uint32_t tid = 0; tid < nbw * nl; tid++) bk_leaf_thread(tid, prm, bsum.data(), gS.data(), gA.data());
    for (uint32_t bw = 0; bw < nbw; bw++) {
        ge_ext S, A;
        if (prm.c == 8) {
            bk_combine(S, A, gS.data(), gA.data(), bw * 8, 8, 1, 16);
        } else {
            std::vector<ge_ext> lS(32), lA(32);
            for (uint32_t g = 0; g < 32; g++) bk_combine(lS[g], lA[g], gS.data(), gA.data(), bw * 256 + g * 8, 8, 1, 8);
            ge_ext s4[4], a4[4];
            for (uint32_t g = 0; g < 4; g++) bk_combine(s4[g], a4[g], lS.data(), lA.data(), g * 8, 8, 1, 64);
            for (uint32_t g = 0; g < 4; g++) { lS[g * 8] = s4[g]; lA[g * 8] = a4[g]; }
            bk_combine(S, A, lS.data(), lA.data(), 0, 4, 8, 512);
        }
        const uint32_t b = bw / prm.nwin, w = bw - b * prm.nwin;
        bk_emit_columns(w, prm, A, colq16.data() + (size_t)b * 64 * 32);
    }
    std::vector<ge_ext> hq(nmsm + 1);
    std::vector<uint32_t> outw((size_t)nmsm * 8 + 8), st1(nmsm + 1, 0);
    for (uint32_t b = 0; b < nmsm; b++) hw_horner_msm((const uint16_t *)(colq16.data() + (size_t)b * 64 * 32), &hq[b]);
    for (uint32_t b = 0; b < nmsm; b++) vb_horner_thread(b, nullptr, hq.data(), single ? st1.data() : status.data(), outw.data(), nullptr);
    memcpy(out, outw.data(), (size_t)nmsm * 32);
    for (uint32_t b = 0; b < nbatch; b++) status_out[b] = (uint8_t)status[b];
    return 0;
}

// The fused bucket chain (bucket2.h; the same bodies as k_bucket2.hip), lane by lane and phase by phase: `lanes` lanes per (MSM, window)
// workgroup (64, 128, 256 on the device; any value >= 1 here, so that runs of one entry per lane and buckets spread over many lanes are
// reachable with small inputs).  stats (optional, 4 words): head pieces written, buckets owned across lanes, the longest chain of head
// pieces one owner added, entries listed in all.
int h_msm_bucket2(uint32_t nbatch, const uint32_t *n_terms, const uint8_t *scalars, const uint8_t *points, uint32_t lanes, uint8_t *out, uint8_t *status_out,
                  uint32_t *stats) {
    const bk_params prm = bk_make(BK2_C);
    std::vector<uint32_t> msm_first(nbatch + 1, 0);
    for (uint32_t b = 0; b < nbatch; b++) {
        if (n_terms[b] > BK2_MAX_TERMS) return -1;
        msm_first[b + 1] = msm_first[b] + n_terms[b];
    }
    const uint32_t total = msm_first[nbatch], nbw = nbatch * prm.nwin;
    std::vector<fb_entry> pts(total + 1);
    std::vector<uint8_t> dig((size_t)BK2_NWIN * total + 1, 0xee);
    std::vector<uint32_t> status(nbatch + 1, 0);
    for (uint32_t t = 0; t < total; t++)
        bk2_prepare_thread(t, total, nbatch, msm_first.data(), (const uint32_t *)scalars, (const uint32_t *)points, pts.data(), dig.data(), status.data());
    // the short-register decode against the plain one, point by point
    for (uint32_t t = 0; t < total; t++) {
        ge_ext a, b2;
        uint32_t pw[8];
        memcpy(pw, points + 32 * (size_t)t, 32);
        const bool oa = ristretto_decompress(a, pw), ob = ristretto_decompress_lp(b2, (const uint32_t *)(points + 32 * (size_t)t));
        if (oa != ob || memcmp(&a, &b2, sizeof a)) return -2;
    }
    std::vector<ge_ext> bsum((size_t)nbw * prm.half);
    memset((void *)bsum.data(), 0xee, bsum.size() * sizeof(ge_ext));
    uint32_t st[4] = {0, 0, 0, 0};
    std::vector<uint32_t> l_cnt(BK2_HALF), l_off(BK2_HALF + 1), l_tmp(BK2_HALF);
    std::vector<uint16_t> l_list(BK2_MAX_TERMS);
    std::vector<ge_ext> l_head(lanes);
    std::vector<bk2_tail> tails(lanes);
    for (uint32_t bw = 0; bw < nbw; bw++) {
        const uint32_t b = bw / prm.nwin, w = bw - b * prm.nwin;
        bk2_seg sg; sg.first = msm_first[b]; sg.count = msm_first[b + 1] - sg.first; sg.lanes = lanes; sg.dig_w = dig.data() + (size_t)w * total;
        bk2_lds l; l.cnt = l_cnt.data(); l.off = l_off.data(); l.tmp = l_tmp.data(); l.list = l_list.data(); l.head = l_head.data();
        std::fill(l_list.begin(), l_list.end(), (uint16_t)0xffff);
        for (uint32_t lane = 0; lane < lanes; lane++) bk2_w0(lane, sg, l);
        for (uint32_t lane = 0; lane < lanes; lane++) bk2_w1(lane, sg, l);
        {
            uint32_t *src = l.cnt, *dst = l.tmp;
            for (uint32_t s = 1; s < BK2_HALF; s <<= 1) {
                for (uint32_t lane = 0; lane < lanes; lane++) bk2_w2_step(lane, s, sg, src, dst);
                std::swap(src, dst);
            }
        }
        for (uint32_t lane = 0; lane < lanes; lane++) bk2_w2_fin(lane, sg, l);
        for (uint32_t lane = 0; lane < lanes; lane++) bk2_w3(lane, sg, l);
        // the list is a permutation of the window's non-zero digits, sorted by |digit|
        const uint32_t n = l.off[BK2_HALF];
        if (n > sg.count) return -3;
        for (uint32_t j = 0; j < BK2_HALF; j++) {
            if (l.off[j] > l.off[j + 1] || l.cnt[j] != l.off[j + 1]) return -4;
            for (uint32_t pos = l.off[j]; pos < l.off[j + 1]; pos++) {
                const uint32_t e = l.list[pos], i = e & 0x7fffu;
                if (i >= sg.count) return -5;
                const int d = (int)sg.dig_w[sg.first + i] - 128;
                if ((uint32_t)(d < 0 ? -d : d) != j + 1 || (d < 0) != ((e >> 15) != 0)) return -6;
            }
        }
        st[3] += n;
        ge_ext *bsum_w = bsum.data() + (size_t)bw * prm.half;
        for (uint32_t lane = 0; lane < lanes; lane++) memset((void *)&l_head[lane], 0xdd, sizeof(ge_ext));
        for (uint32_t lane = 0; lane < lanes; lane++) bk2_w4(lane, sg, l, pts.data() + sg.first, bsum_w, tails[lane]);
        const uint32_t q = bk2_q(n, lanes);
        for (uint32_t lane = 0; lane < lanes; lane++) {
            if (!tails[lane].owner) continue;
            st[1]++;
            uint32_t chain = 0;
            for (uint32_t k = lane + 1; k < lanes && k * q < l.off[tails[lane].bucket + 1]; k++) chain++;
            st[0] += chain;
            st[2] = std::max(st[2], chain);
        }
        for (uint32_t lane = 0; lane < lanes; lane++) bk2_w5(lane, sg, l, bsum_w, tails[lane]);
    }
    if (stats) memcpy(stats, st, sizeof st);
    // the chain's end, both forms (k_bucket2.hip: k_bk_leaf + k_msm_tail for wide batches, k_bk2_leafv + k_msm_tail_fast for narrow ones),
    // phase by phase; the wavefront shuffles of the narrow form are plain sums here.  Both must give the same encodings.
    std::vector<uint32_t> outw[2];
    for (int fast = 0; fast < 2; fast++) {
        outw[fast].assign((size_t)nbatch * 8 + 8, 0);
        const uint32_t nl = fast ? BK2_FAST_LEAVES : bk_leaves(prm);
        std::vector<ge_ext> gS((size_t)nbw * nl), gA((size_t)nbw * nl);
        for (uint32_t tid = 0; tid < nbw * nl; tid++) {
            if (fast) bk2_leafv_thread(tid, bsum.data(), gA.data());
            else bk_leaf_thread(tid, prm, bsum.data(), gS.data(), gA.data());
        }
        for (uint32_t b = 0; b < nbatch; b++) {
            std::vector<uint32_t> colq8(BK2_NWIN * 32, 0), hwlds(128, 0), tw(8, 0);
            if (fast) {
                for (uint32_t w = 0; w < BK2_NWIN; w++) {
                    ge_ext V[8];
                    for (uint32_t g = 0; g < 8; g++) bk2_fast_v4(w, g, b, gA.data(), V[g]);
                    for (uint32_t step = 4; step >= 1; step >>= 1)
                        for (uint32_t g = 0; g < step; g++) ge_add(V[g], V[g], V[g + step]);
                    vb_encode_colq16(colq8.data() + w * 32, V[0]);
                }
            } else {
                for (uint32_t lane = 0; lane < 64; lane++) bk2_tail_t1(lane, b, gS.data(), gA.data(), colq8.data());
            }
            ge_ext fin[2];
            hw_horner8_msm((const uint16_t *)colq8.data(), hwlds.data(), &fin[0]);
            fe tin, raw, raw_lane;
            bk2_tail_t4a(&fin[0], &tin, tw.data());
            hw_invsqrt_raw_fe((const uint16_t *)tw.data(), hwlds.data(), &raw);
            fe_invsqrt_raw(raw_lane, tin);   // the one-lane chain: same field element
            if (!fe_eq(raw, raw_lane)) return -8;
            bk2_tail_t4b(b, &fin[0], &raw, &tin, status.data(), outw[fast].data(), nullptr, nullptr);
        }
    }
    if (memcmp(outw[0].data(), outw[1].data(), (size_t)nbatch * 32)) return -7;
    memcpy(out, outw[0].data(), (size_t)nbatch * 32);
    for (uint32_t b = 0; b < nbatch; b++) status_out[b] = (uint8_t)status[b];
    return 0;
}

// The batched inner-product-proof prover (ipp_prover.h), lane by lane; the MSMs go through the variable-base pipeline
// emulation above.  ts0: the 208-byte transcript state BEFORE innerproduct_domain_sep(n).
// InnerProductProof::verification_scalars alone (ipp_vs_front_thread + ipp_vs_s_thread, bpgpu_ipp_verification_scalars): per-proof start
// states (per_proof = 1: states208 holds nbatch of them, the domain separator is applied by the lane) or one for the batch (the host applies
// innerproduct_domain_sep(n) first, as the runtime does).  Hands back status, u_i^2, u_i^-2, s_i and the transcripts as the DEVICE leaves them
// (the runtime replaces those of FormatError / wrong-n proofs by the caller's own state afterwards in the one-state mode).
int h_ipp_vs(uint32_t n, uint32_t nbatch, const uint8_t *proofs, uint32_t proof_len, const uint8_t *states208, int per_proof, uint8_t *status_out, uint8_t *u_sq,
             uint8_t *u_inv_sq, uint8_t *s_out, uint8_t *ts_out208) {
    if (proof_len % 32 || proof_len < 64 || ((proof_len / 32 - 2) & 1)) return -1;
    const uint32_t k = (proof_len / 32 - 2) / 2;
    ipp_shape sh; sh.n = n; sh.k = k; sh.N = 0; sh.proof_len = proof_len; sh.nproofs = nbatch; sh.shape_verdict = (n != (1u << k)) ? BP_VERDICT_VERIFICATION : 0; sh.bases_shared = 0;
    rp_strobe_init init; memset(&init, 0, sizeof init);
    std::vector<uint32_t> ts_in;
    if (per_proof) {
        ts_in.assign((size_t)nbatch * BP_TS_WORDS, 0);
        for (uint32_t p = 0; p < nbatch; p++) {
            const uint8_t *st = states208 + (size_t)p * 208;
            memcpy(&ts_in[(size_t)p * BP_TS_WORDS], st, 200);
            ts_in[(size_t)p * BP_TS_WORDS + 50] = rp_ts_meta(st[200], st[201], st[202]);
        }
    } else {
        memcpy(init.w, states208, 200);
        kstate st0; st0.w = init.w; st0.stride = 1;
        strobe t; t.st = st0; t.pos = states208[200]; t.pos_begin = states208[201]; t.cur_flags = states208[202];
        const uint8_t dom[7] = {'d','o','m','-','s','e','p'}, ipp[6] = {'i','p','p',' ','v','1'}, ln[1] = {'n'};
        merlin_append_message(t, dom, 7, ipp, 6);
        merlin_append_u64(t, ln, 1, n);
        init.pos = t.pos; init.pos_begin = t.pos_begin; init.cur_flags = t.cur_flags;
    }
    std::vector<uint32_t> us((size_t)nbatch * (k ? k : 1) * 8 + 8, 0), ui((size_t)nbatch * (k ? k : 1) * 8 + 8, 0), tab((size_t)nbatch * (k ? k : 1) * 20 + 8, 0),
        tso((size_t)nbatch * BP_TS_WORDS, 7), status(nbatch + 1, 0), sv((size_t)nbatch * (n ? n : 1) * 8 + 8, 0);
    for (uint32_t p = 0; p < nbatch; p++) {
        uint32_t w[50]; kstate st; st.w = w; st.stride = 1;
        ipp_vs_front_thread(p, sh, init, st, proofs, per_proof ? ts_in.data() : nullptr, us.data(), ui.data(), tab.data(), tso.data(), status.data());
    }
    if (!sh.shape_verdict)
        for (uint32_t tid = 0; tid < n * nbatch; tid++) ipp_vs_s_thread(tid, sh, tab.data(), status.data(), sv.data());
    for (uint32_t p = 0; p < nbatch; p++) {
        status_out[p] = (uint8_t)status[p];
        memset(ts_out208 + (size_t)p * 208, 0, 208);
        memcpy(ts_out208 + (size_t)p * 208, &tso[(size_t)p * BP_TS_WORDS], 200);
        const uint32_t meta = tso[(size_t)p * BP_TS_WORDS + 50];
        ts_out208[(size_t)p * 208 + 200] = meta & 0xff; ts_out208[(size_t)p * 208 + 201] = (meta >> 8) & 0xff; ts_out208[(size_t)p * 208 + 202] = (meta >> 16) & 0xff;
    }
    memcpy(u_sq, us.data(), (size_t)nbatch * k * 32);
    memcpy(u_inv_sq, ui.data(), (size_t)nbatch * k * 32);
    if (!sh.shape_verdict) memcpy(s_out, sv.data(), (size_t)nbatch * n * 32);
    return 0;
}

int h_ipp_create(uint32_t n, uint32_t nbatch, const uint8_t *ts0, const uint8_t *Q, const uint8_t *Gf, const uint8_t *Hf, const uint8_t *G,
                 const uint8_t *H, int bases_shared, const uint8_t *a_in, const uint8_t *b_in, uint8_t *proofs, uint8_t *status_out) {
    uint32_t k = 0; while ((1u << k) < n) k++;
    ippc_shape sh; sh.n = n; sh.k = k; sh.nproofs = nbatch; sh.bases_shared = bases_shared ? 1 : 0;
    const uint32_t proof_len = 32 * (2 * k + 2), N = n + 1;
    std::vector<uint32_t> a((size_t)nbatch * n * 8), b(a.size()), wG(a.size()), wH(a.size()), status(nbatch + 1, 0), u((size_t)nbatch * 8), ui(u.size());
    std::vector<uint32_t> ts((size_t)nbatch * BP_TS_WORDS);
    {
        uint32_t w[50]; memcpy(w, ts0, 200);
        strobe t; t.st.w = w; t.st.stride = 1; t.pos = ts0[200]; t.pos_begin = ts0[201]; t.cur_flags = ts0[202];
        const uint8_t dom[7] = {'d','o','m','-','s','e','p'}, ipp[6] = {'i','p','p',' ','v','1'}, ln[1] = {'n'};
        merlin_append_message(t, dom, 7, ipp, 6); merlin_append_u64(t, ln, 1, n);
        for (uint32_t p = 0; p < nbatch; p++) {
            memcpy(&ts[(size_t)p * BP_TS_WORDS], w, 200);
            ts[(size_t)p * BP_TS_WORDS + 50] = rp_ts_meta(t.pos, t.pos_begin, t.cur_flags);
            ts[(size_t)p * BP_TS_WORDS + 51] = 0;
        }
    }
    for (uint32_t tid = 0; tid < nbatch * n; tid++) ippc_init_thread(tid, sh, a_in, b_in, Gf, Hf, a.data(), b.data(), wG.data(), wH.data(), status.data());
    std::vector<uint32_t> msc((size_t)2 * nbatch * N * 8 + 8), mpt(msc.size()), mout((size_t)2 * nbatch * 8 + 8), nt(2 * nbatch, N);
    std::vector<uint8_t> mst(2 * nbatch + 1);
    memset(proofs, 0, (size_t)nbatch * proof_len);
    for (uint32_t j = 0; j < k; j++) {
        for (uint32_t p = 0; p < nbatch; p++) ippc_q_thread(p, sh, j, a.data(), b.data(), Q, msc.data(), mpt.data());
        for (uint32_t tid = 0; tid < nbatch * n; tid++) ippc_terms_thread(tid, sh, j, a.data(), b.data(), wG.data(), wH.data(), G, H, msc.data(), mpt.data());
        h_msm_vb(2 * nbatch, nt.data(), (const uint8_t *)msc.data(), (const uint8_t *)mpt.data(), (uint8_t *)mout.data(), mst.data());
        for (uint32_t p = 0; p < nbatch; p++) {
            uint32_t stw[50]; kstate st; st.w = stw; st.stride = 1;
            ippc_challenge_thread(p, sh, j, st, mout.data(), mst.data(), ts.data(), u.data(), ui.data(), proofs, proof_len, status.data());
        }
        for (uint32_t tid = 0; tid < nbatch * n; tid++) ippc_fold_thread(tid, sh, j, u.data(), ui.data(), a.data(), b.data(), wG.data(), wH.data());
    }
    for (uint32_t p = 0; p < nbatch; p++) { ippc_final_thread(p, sh, a.data(), b.data(), proofs, proof_len); status_out[p] = (uint8_t)status[p]; }
    return 0;
}

// The batched LinearProof prover (linear_prover.h), lane by lane; MSMs through the variable-base pipeline emulation.
// ts0: 208-byte transcript state before innerproduct_domain_sep; rng: nbatch x 64 (2k + 2) bytes.
int h_lin_create(uint32_t n, uint32_t nbatch, const uint8_t *ts0, const uint8_t *rng, const uint8_t *Cc, const uint8_t *r_in, const uint8_t *a_in,
                 const uint8_t *b_in, int b_shared, const uint8_t *G, const uint8_t *F, const uint8_t *B, uint8_t *proofs, uint8_t *status_out,
                 uint8_t *ts_out) {
    uint32_t k = 0; while ((1u << k) < n) k++;
    linc_shape sh; sh.n = n; sh.k = k; sh.nproofs = nbatch; sh.b_shared = b_shared ? 1 : 0;
    const uint32_t proof_len = 32 * (2 * k + 3), N = n / 2 + 2, NS = n + 2, nd = 2 * k + 2;
    std::vector<uint32_t> a((size_t)nbatch * n * 8), b(a.size()), wG(a.size()), status(nbatch + 1, 0), x((size_t)nbatch * 8), xi(x.size()), r(x.size()),
        draws((size_t)nbatch * nd * 8), ts((size_t)nbatch * BP_TS_WORDS);
    {
        uint32_t w[50]; memcpy(w, ts0, 200);
        strobe t; t.st.w = w; t.st.stride = 1; t.pos = ts0[200]; t.pos_begin = ts0[201]; t.cur_flags = ts0[202];
        const uint8_t dom[7] = {'d','o','m','-','s','e','p'}, ipp[6] = {'i','p','p',' ','v','1'}, ln[1] = {'n'};
        merlin_append_message(t, dom, 7, ipp, 6); merlin_append_u64(t, ln, 1, n);
        for (uint32_t p = 0; p < nbatch; p++) {
            memcpy(&ts[(size_t)p * BP_TS_WORDS], w, 200);
            ts[(size_t)p * BP_TS_WORDS + 50] = rp_ts_meta(t.pos, t.pos_begin, t.cur_flags);
            ts[(size_t)p * BP_TS_WORDS + 51] = 0;
        }
    }
    for (uint32_t tid = 0; tid < nbatch * n; tid++) linc_init_thread(tid, sh, a_in, b_in, a.data(), b.data(), wG.data(), status.data());
    for (uint32_t p = 0; p < nbatch; p++) {
        uint32_t stw[50]; kstate st; st.w = stw; st.stride = 1;
        linc_public_thread(p, sh, st, Cc, b_in, G, F, B, r_in, rng, ts.data(), r.data(), draws.data(), status.data());
    }
    const size_t terms = std::max<size_t>((size_t)2 * nbatch * N, (size_t)nbatch * NS);
    std::vector<uint32_t> msc(terms * 8 + 8), mpt(msc.size()), mout((size_t)2 * nbatch * 8 + 8), nt(2 * nbatch, N), nts(nbatch, NS);
    std::vector<uint8_t> mst(2 * nbatch + 1);
    memset(proofs, 0, (size_t)nbatch * proof_len);
    for (uint32_t j = 0; j < k; j++) {
        for (uint32_t p = 0; p < nbatch; p++) linc_q_thread(p, sh, j, a.data(), b.data(), draws.data(), F, B, msc.data(), mpt.data());
        for (uint32_t tid = 0; tid < nbatch * n; tid++) linc_terms_thread(tid, sh, j, a.data(), wG.data(), G, msc.data(), mpt.data());
        h_msm_vb(2 * nbatch, nt.data(), (const uint8_t *)msc.data(), (const uint8_t *)mpt.data(), (uint8_t *)mout.data(), mst.data());
        for (uint32_t p = 0; p < nbatch; p++) {
            uint32_t stw[50]; kstate st; st.w = stw; st.stride = 1;
            linc_challenge_thread(p, sh, j, st, mout.data(), mst.data(), ts.data(), draws.data(), r.data(), x.data(), xi.data(), proofs, proof_len, status.data());
        }
        for (uint32_t tid = 0; tid < nbatch * n; tid++) linc_fold_thread(tid, sh, j, x.data(), xi.data(), a.data(), b.data(), wG.data());
    }
    for (uint32_t p = 0; p < nbatch; p++) linc_sq_thread(p, sh, b.data(), draws.data(), F, B, msc.data(), mpt.data());
    for (uint32_t tid = 0; tid < nbatch * n; tid++) linc_sterms_thread(tid, sh, wG.data(), draws.data(), G, msc.data(), mpt.data());
    h_msm_vb(nbatch, nts.data(), (const uint8_t *)msc.data(), (const uint8_t *)mpt.data(), (uint8_t *)mout.data(), mst.data());
    for (uint32_t p = 0; p < nbatch; p++) {
        uint32_t stw[50]; kstate st; st.w = stw; st.stride = 1;
        linc_final_thread(p, sh, st, mout.data(), mst.data(), ts.data(), a.data(), draws.data(), r.data(), proofs, proof_len, status.data());
        status_out[p] = (uint8_t)status[p];
        if (ts_out) {
            memset(ts_out + (size_t)p * 208, 0, 208);
            memcpy(ts_out + (size_t)p * 208, &ts[(size_t)p * BP_TS_WORDS], 200);
            const uint32_t meta = ts[(size_t)p * BP_TS_WORDS + 50];
            ts_out[(size_t)p * 208 + 200] = meta & 0xff; ts_out[(size_t)p * 208 + 201] = (meta >> 8) & 0xff; ts_out[(size_t)p * 208 + 202] = (meta >> 16) & 0xff;
        }
    }
    return 0;
}

// The batched range-proof prover (rp_prover.h), lane by lane: commitments through the shared-generator pipeline emulation
// (h_msm_shared, tables of window W), the inner-product argument through h_ipp_create's rounds.
// gens: [B_blinding, B, G(party-major, capacity), H(...)] encodings; ts0: 208-byte transcript state before rangeproof_domain_sep.
int h_rp_prove(uint32_t W, uint32_t gens_capacity, uint32_t party_capacity, const uint8_t *gens, uint32_t n, uint32_t m, uint32_t nbatch,
               const uint64_t *values, const uint8_t *blindings, const uint8_t *ts0, const uint8_t *rng, uint8_t *proofs, uint8_t *commitments,
               uint8_t *ts_out) {
    rpp_shape sh; sh.n = n; sh.m = m; sh.nm = n * m; sh.k = 0; while ((1u << sh.k) < sh.nm) sh.k++;
    sh.nproofs = nbatch; sh.proof_len = 32 * (9 + 2 * sh.k); sh.n_gen_terms = 2 * sh.nm + 2; sh.rng_per_proof = 64 * (m * (2 * n + 2) + 2 * m);
    const uint32_t nm = sh.nm, tot = gens_capacity * party_capacity, n_loaded = 2 + 2 * tot;
    std::vector<uint32_t> ids; ids.push_back(0); ids.push_back(1);
    for (uint32_t j = 0; j < m; j++) for (uint32_t i = 0; i < n; i++) ids.push_back(2 + j * gens_capacity + i);
    for (uint32_t j = 0; j < m; j++) for (uint32_t i = 0; i < n; i++) ids.push_back(2 + tot + j * gens_capacity + i);
    std::vector<uint32_t> ts((size_t)nbatch * BP_TS_WORDS);
    {
        uint32_t w[50]; memcpy(w, ts0, 200);
        strobe t; t.st.w = w; t.st.stride = 1; t.pos = ts0[200]; t.pos_begin = ts0[201]; t.cur_flags = ts0[202];
        const uint8_t dom[7] = {'d','o','m','-','s','e','p'}, rp[13] = {'r','a','n','g','e','p','r','o','o','f',' ','v','1'}, ln[1] = {'n'}, lm[1] = {'m'};
        merlin_append_message(t, dom, 7, rp, 13); merlin_append_u64(t, ln, 1, n); merlin_append_u64(t, lm, 1, m);
        for (uint32_t p = 0; p < nbatch; p++) {
            memcpy(&ts[(size_t)p * BP_TS_WORDS], w, 200);
            ts[(size_t)p * BP_TS_WORDS + 50] = rp_ts_meta(t.pos, t.pos_begin, t.cur_flags); ts[(size_t)p * BP_TS_WORDS + 51] = 0;
        }
    }
    const uint32_t nmsm1 = nbatch * (2 + m);
    std::vector<uint32_t> gsc((size_t)nmsm1 * sh.n_gen_terms * 8, 0), mo((size_t)nmsm1 * 8 + 8), fields((size_t)RPP_FIXED * nbatch * 8),
        party((size_t)RPP_PARTY_FIELDS * nbatch * m * 8);
    std::vector<uint8_t> mst(nmsm1 + 1), mver(nmsm1 + 1);
    std::vector<uint32_t> sL((size_t)nbatch * nm * 8), sR(sL.size()), l0(sL.size()), l1(sL.size()), r0(sL.size()), r1(sL.size()), av(sL.size()), bv(sL.size()),
        Gf(sL.size()), Hf(sL.size());
    memset(proofs, 0, (size_t)nbatch * sh.proof_len);
    for (uint32_t p = 0; p < nbatch; p++) rpp_blind_thread(p, sh, values, blindings, rng, gsc.data(), party.data());
    for (uint32_t tid = 0; tid < nbatch * nm; tid++) rpp_bits_thread(tid, sh, values, rng, gsc.data(), sL.data(), sR.data());
    if (h_msm_shared(W, 3, n_loaded, gens, sh.n_gen_terms, ids.data(), nmsm1, 0, (const uint8_t *)gsc.data(), nullptr, nullptr, (uint8_t *)mo.data(), mst.data(), mver.data())) return -1;
    for (uint32_t p = 0; p < nbatch; p++) { uint32_t stw[50]; kstate st; st.w = stw; st.stride = 1; rpp_chal1_thread(p, sh, st, mo.data(), ts.data(), fields.data(), proofs, commitments); }
    for (uint32_t tid = 0; tid < nbatch * m; tid++) rpp_poly_thread(tid, sh, values, fields.data(), sL.data(), sR.data(), l0.data(), l1.data(), r0.data(), r1.data(), party.data());
    std::fill(gsc.begin(), gsc.end(), 0u);
    for (uint32_t p = 0; p < nbatch; p++) rpp_tcommit_thread(p, sh, rng, gsc.data(), party.data());
    if (h_msm_shared(W, 2, n_loaded, gens, sh.n_gen_terms, ids.data(), 2 * nbatch, 0, (const uint8_t *)gsc.data(), nullptr, nullptr, (uint8_t *)mo.data(), mst.data(), mver.data())) return -2;
    std::fill(gsc.begin(), gsc.end(), 0u);
    for (uint32_t p = 0; p < nbatch; p++) { uint32_t stw[50]; kstate st; st.w = stw; st.stride = 1; rpp_chal2_thread(p, sh, st, mo.data(), ts.data(), fields.data(), party.data(), gsc.data(), proofs); }
    for (uint32_t tid = 0; tid < nbatch * nm; tid++) rpp_vectors_thread(tid, sh, fields.data(), l0.data(), l1.data(), r0.data(), r1.data(), av.data(), bv.data(), Gf.data(), Hf.data());
    // inner-product rounds over the generator tables (ippc_*_fixed_thread: Q = w B folded into the B coefficient)
    ippc_shape ish; ish.n = nm; ish.k = sh.k; ish.nproofs = nbatch; ish.bases_shared = 1;
    std::vector<uint32_t> a((size_t)nbatch * nm * 8), b(a.size()), wG(a.size()), wH(a.size()), status(nbatch + 1, 0), u((size_t)nbatch * 8), ui(u.size());
    for (uint32_t tid = 0; tid < nbatch * nm; tid++)
        ippc_init_thread(tid, ish, (const uint8_t *)av.data(), (const uint8_t *)bv.data(), (const uint8_t *)Gf.data(), (const uint8_t *)Hf.data(), a.data(), b.data(), wG.data(), wH.data(), status.data());
    std::vector<uint32_t> mout((size_t)2 * nbatch * 8 + 8);
    std::vector<uint8_t> mst2(2 * nbatch + 1), mver2(2 * nbatch + 1);
    const uint32_t *w_all = fields.data() + (size_t)RPP_W * nbatch * 8;
    for (uint32_t j = 0; j < sh.k; j++) {
        for (uint32_t p = 0; p < nbatch; p++) ippc_q_fixed_thread(p, ish, j, a.data(), b.data(), w_all, gsc.data());
        for (uint32_t tid = 0; tid < nbatch * nm; tid++) ippc_terms_fixed_thread(tid, ish, j, a.data(), b.data(), wG.data(), wH.data(), gsc.data());
        if (h_msm_shared(W, 2, n_loaded, gens, sh.n_gen_terms, ids.data(), 2 * nbatch, 0, (const uint8_t *)gsc.data(), nullptr, nullptr, (uint8_t *)mout.data(), mst2.data(), mver2.data())) return -4;
        for (uint32_t p = 0; p < nbatch; p++) {
            uint32_t stw[50]; kstate st; st.w = stw; st.stride = 1;
            ippc_challenge_thread(p, ish, j, st, mout.data(), mst2.data(), ts.data(), u.data(), ui.data(), proofs + 224, sh.proof_len, status.data());
        }
        for (uint32_t tid = 0; tid < nbatch * nm; tid++) ippc_fold_thread(tid, ish, j, u.data(), ui.data(), a.data(), b.data(), wG.data(), wH.data());
    }
    for (uint32_t p = 0; p < nbatch; p++) ippc_final_thread(p, ish, a.data(), b.data(), proofs + 224, sh.proof_len);
    if (ts_out)
        for (uint32_t p = 0; p < nbatch; p++) {
            memset(ts_out + (size_t)p * 208, 0, 208); memcpy(ts_out + (size_t)p * 208, &ts[(size_t)p * BP_TS_WORDS], 200);
            const uint32_t meta = ts[(size_t)p * BP_TS_WORDS + 50];
            ts_out[(size_t)p * 208 + 200] = meta & 0xff; ts_out[(size_t)p * 208 + 201] = (meta >> 8) & 0xff; ts_out[(size_t)p * 208 + 202] = (meta >> 16) & 0xff;
        }
    for (uint32_t p = 0; p < nbatch; p++) if (status[p]) return -10;
    return 0;
}
}
