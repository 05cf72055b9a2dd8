// Range-proof verification front end on the device: everything
// RangeProof::verify_multiple_with_rng does before its multiscalar
// multiplication (src/range_proof/mod.rs:345-420 of the reference):
//   rp_transcript : lane = proof   parse (mod.rs:504-538, ipp.rs:373-407), replay the Merlin
//                                  transcript (mod.rs:368-393, ipp.rs:213-222), derive y,z,x,w,u_i
//   rp_expand_a   : lane = proof   batch inversion, u_i^2, u_i^-2, delta(y,z) (mod.rs:587-593),
//                                  the 4+2k+m "unique" coefficients and the B / B_blinding ones
//   rp_expand_b   : lane = (i, proof)  s_i, g_i = -z - a s_i, h_i = z + y^-i (z^2 z^j 2^i' - b s_i^-1)
//                                  (ipp.rs:241-250, mod.rs:406-419), written straight as
//                                  fixed-window digits for the generator tables
// Per-proof intermediate scalars live in HBM, field-major ([field][proof][10 words]) so that the
// 64 proofs of a wavefront read consecutive 40-byte records (8 words: canonical scalar; 10 words:
// lazy 28-bit-limb Montgomery form, sc25519.h).
#ifndef BPGPU_RANGEPROOF_H
#define BPGPU_RANGEPROOF_H
#include "keccak.h"
#include "rp_script.h"
#include "msm_fixed.h"
#include "sc25519.h"
#include "scinv.h"
#include "bucket.h"
#include "chacha20.h"

namespace bp {

#define BP_RP_MAX_K 16   // lg(n*m) supported by the device front end

#define BP_VERDICT_OK 0
#define BP_VERDICT_VERIFICATION 1
#define BP_VERDICT_FORMAT 2

struct rp_shape {
    uint32_t n, m, nm, k;          // k = lg(nm)
    uint32_t U;                    // unique points per proof: 4 + 2k + m
    uint32_t proof_len;            // bytes, = 32*(9+2k)
    uint32_t nproofs;
    uint32_t shape_verdict;        // != 0: only parse, then report this verdict (InvalidBitsize, ...)
    uint32_t radix5 = 0;           // 1: the per-proof points in signed radix 32 (16-entry tables, 51 windows, msm_vb.h); 0: radix 16
    uint32_t a_outside = 0;        // 1: A -- coefficient 1 -- is added after the Horner chain: no recoding, only 1A in its table (wide chains)
    // per-proof randomness the caller did not bring is expanded ON THE DEVICE from one 32-byte key per launch chain (drawn on the host
    // by the library's generator, hostrng.h): proof p's 64 bytes are block p of ChaCha20(key, nonce = domain) -- nothing per proof is
    // drawn, staged or copied on the host (64 bytes per proof at 6 ... 10 M proofs/s would be a core's worth of ChaCha)
    uint32_t defer_emit = 0;       // narrow chains (32 lanes per proof): the U coefficient recodings of the scalar role are done by U lanes at once instead of by
                                   // the leader one after the other (rp_defer; option coop_defer_emit)
    uint32_t narrow_hi = 0;        // very narrow chains: 2 -- every per-proof point also has the table of its 2^128 multiple (k_rp_stage1_coop's third role), the Horner chain 32
                                   // windows; 4 -- of its 2^64, 2^128 and 2^192 multiples, 16 windows; 0 -- one table, 64 windows
    uint32_t coop_split = 0;       // narrow chains, per-proof check: the k + 1 inversions run on k + 1 lanes of the group at once (rp_split_invert_lane) and the two
                                   // basepoint coefficients are formed in launch 3 beside the exponents (rp_rows_thread); option coop_split
    uint32_t seeded = 0;           // RP_SEED_RNG | RP_SEED_WEIGHTS
    uint32_t seed[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
enum { RP_SEED_RNG = 1,        // the batching challenge's rng bytes (mod.rs:396: thread_rng() in verify_multiple), where no buffer was given
       RP_SEED_WEIGHTS = 2 };  // the combination weights of the batch-combined check, where the caller supplied none
// proof p's 64 pseudo-random bytes of domain `dom` (16 words, wide-reduced by the callers)
BP_HD void rp_seed_words(uint32_t w[16], const rp_shape &sh, uint32_t p, uint32_t dom) { chacha20_block(sh.seed, (uint64_t)p, dom, 0u, w); }

// Merlin state after Transcript::new(label) + rangeproof_domain_sep(n, m), computed once on the host
struct rp_strobe_init {
    uint32_t w[50];
    uint32_t pos, pos_begin, cur_flags;
};

// ---- coalesced launches (bpgpu_pool_*, include/bpgpu.h) ---------------------------------------------------------
// Several submitted batches of one shape verified by ONE launch chain: proofs [first, first + count) of the launch are
// item `i`'s, read from and reported to that item's own buffers.  Only the first launch (inputs) and the last one
// (verdicts) look at the table; everything in between indexes the launch-wide scratch by the global proof number.
struct rp_seg {
    const uint8_t *proofs, *commitments;
    const uint8_t *rng64;     // may be null: the launch-wide rng buffer at the proof's global index
    uint8_t *verdict;
    uint32_t *msm_out;        // may be null
    const uint32_t *init_w;   // may be null: the chain's common start state (the kernels' `init` argument); else the 50 sponge words this item's proofs
                              // start from -- Transcript::new(ITS label) + rangeproof_domain_sep -- at the position the chain's script was compiled for
                              // (labels of one length share every position: items that differ only in their label share a chain)
    uint32_t first, count;
};
// index of the segment holding global proof p (segs sorted by first, segs[0].first == 0)
BP_HD uint32_t rp_seg_find(const rp_seg *segs, uint32_t nseg, uint32_t p) {
    uint32_t lo = 0, hi = nseg;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (segs[mid].first <= p) lo = mid;
        else hi = mid;
    }
    return lo;
}
// The table as the kernels get it, BY VALUE in their argument block: up to RP_SEG_INLINE items inline (the usual case: a few
// batches of ~1024 proofs per chain -- no upload, no extra copy command in the stream: 16 x 48 bytes of the argument block), more
// through `ext` (device memory).
// n == 0: not a coalesced launch.
#define RP_SEG_INLINE 16
struct rp_seg_tab {
    const rp_seg *ext;
    uint32_t n, pad;
    rp_seg in[RP_SEG_INLINE];
};
BP_HD rp_seg rp_seg_lookup(const rp_seg_tab &t, uint32_t p) {
    if (t.ext) return t.ext[rp_seg_find(t.ext, t.n, p)];
    rp_seg r = t.in[0];
#pragma unroll
    for (uint32_t i = 1; i < RP_SEG_INLINE; i++)   // static indices only: the argument block is read with scalar loads
        if (i < t.n && t.in[i].first <= p) r = t.in[i];
    return r;
}
// one proof's inputs: its bytes, its m commitments, its 64 rng bytes
struct rp_inputs {
    const uint8_t *pr, *cm, *rs;
    const uint32_t *init_w;   // rp_seg::init_w of the proof's item (null: the launch-wide start state)
};
BP_HD rp_inputs rp_resolve(uint32_t p, const rp_shape &sh, const uint8_t *proofs, const uint8_t *commitments, const uint8_t *rng64,
                           const rp_seg_tab &segs) {
    rp_inputs in;
    if (segs.n) {
        const rp_seg sg = rp_seg_lookup(segs, p);
        const uint32_t q = p - sg.first;
        in.pr = sg.proofs + (uint64_t)q * sh.proof_len;
        in.cm = sg.commitments + (uint64_t)q * sh.m * 32;
        in.rs = sg.rng64 ? sg.rng64 + (uint64_t)q * 64 : (rng64 ? rng64 + (uint64_t)p * 64 : nullptr);
        in.init_w = sg.init_w;
    } else {
        in.init_w = nullptr;
        in.pr = proofs + (uint64_t)p * sh.proof_len;
        in.cm = commitments + (uint64_t)p * sh.m * 32;
        in.rs = rng64 ? rng64 + (uint64_t)p * 64 : nullptr;
    }
    return in;
}

// field-major scalar store
enum {
    RPF_Y = 0, RPF_Z, RPF_X, RPF_W, RPF_C, RPF_TX, RPF_TXB, RPF_EB, RPF_A, RPF_B,
    RPF_ZZ,            // z^2
    RPF_MINUS_Z,
    RPF_A_M, RPF_B_M, RPF_Z_M, RPF_ZZ_M,   // Montgomery forms used by expand_b
    RPF_ROW0, RPF_ROW1,                    // weighted B_blinding / B coefficients (batch-combination mode only)
    RPF_FIXED_COUNT
};
struct rp_fields {
    uint32_t u;         // k challenges u_i (plain)
    uint32_t u_m;       // k: u_i      (Montgomery)
    uint32_t uinv_m;    // k: u_i^-1   (Montgomery)
    uint32_t yinvp_m;   // k: y^-(2^b) (Montgomery)
    uint32_t zzzj_m;    // m: z^2 * z^j (Montgomery)
    uint32_t count;
};
BP_HD rp_fields rp_field_layout(uint32_t k, uint32_t m) {
    rp_fields f;
    f.u = RPF_FIXED_COUNT;
    f.u_m = f.u + k;
    f.uinv_m = f.u_m + k;
    f.yinvp_m = f.uinv_m + k;
    f.zzzj_m = f.yinvp_m + k;
    f.count = f.zzzj_m + m;
    return f;
}
#define BP_RP_REC 10   // words per record
BP_HD void rp_store(uint32_t *buf, uint32_t nproofs, uint32_t field, uint32_t p, const sc &s) {
    uint32_t *d = buf + ((uint64_t)field * nproofs + p) * BP_RP_REC;
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = s.v[i];
}
BP_HD void rp_load(sc &s, const uint32_t *buf, uint32_t nproofs, uint32_t field, uint32_t p) {
    const uint32_t *d = buf + ((uint64_t)field * nproofs + p) * BP_RP_REC;
#pragma unroll
    for (int i = 0; i < 8; i++) s.v[i] = d[i];
}
BP_HD void rp_store28(uint32_t *buf, uint32_t nproofs, uint32_t field, uint32_t p, const sc28 &s) {
    uint32_t *d = buf + ((uint64_t)field * nproofs + p) * BP_RP_REC;
#pragma unroll
    for (int i = 0; i < 10; i++) d[i] = s.v[i];
}
BP_HD void rp_load28(sc28 &s, const uint32_t *buf, uint32_t nproofs, uint32_t field, uint32_t p) {
    const uint32_t *d = buf + ((uint64_t)field * nproofs + p) * BP_RP_REC;
#pragma unroll
    for (int i = 0; i < 10; i++) s.v[i] = d[i];
}

// 32-byte record -> 8 LE words; all proof / commitment / rng buffers are 4-byte aligned
// (bpgpu.h requires it of device pointers; the host entry points stage through hipMalloc memory)
BP_HD void load_words8(uint32_t w[8], const uint8_t *src) {
    const uint32_t *s4 = (const uint32_t *)src;
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = s4[i];
}
BP_HD bool words8_zero(const uint32_t w[8]) {
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r |= w[i];
    return r == 0;
}

BP_HD void rp_challenge_scalar(strobe &t, const uint8_t *label, uint32_t label_len, sc &out) {
    uint32_t w[16];
    merlin_challenge_words16(t, label, label_len, w);
    sc_from_wide(out, w);
}

// ---- transcripts across the ABI (include/bpgpu.h BPGPU_TRANSCRIPT_BYTES) ---------------------------------
// one state = 52 words: the 50 sponge words, then pos | pos_begin << 8 | cur_flags << 16, then 0
#define BP_TS_WORDS 52
#define BP_TS_DOMSEP 1u   // the start state is the caller's transcript: rangeproof_domain_sep(n, m) is applied on the device
BP_HD uint32_t rp_ts_meta(uint32_t pos, uint32_t pos_begin, uint32_t cur_flags) { return (pos & 0xffu) | ((pos_begin & 0xffu) << 8) | ((cur_flags & 0xffu) << 16); }
// the untouched input state of proof p -> ts_out (proofs the parser / the parameter checks reject)
BP_HD void rp_ts_passthrough(uint32_t p, const rp_strobe_init &init, const uint32_t *ts_in, uint32_t *ts_out) {
    if (!ts_out) return;
    uint32_t *o = ts_out + (uint64_t)p * BP_TS_WORDS;
    if (ts_in) {
        for (uint32_t i = 0; i < BP_TS_WORDS; i++) o[i] = ts_in[(uint64_t)p * BP_TS_WORDS + i];
    } else {
        for (uint32_t i = 0; i < 50; i++) o[i] = init.w[i];
        o[50] = rp_ts_meta(init.pos, init.pos_begin, init.cur_flags);
        o[51] = 0;
    }
}

// the state a caller's `&mut Transcript` is left in when the replay stops early (an identity point: transcript.rs:75-87; n m != 2^k:
// ipp.rs:203-211): the sponge words as they are now (+ framing bytes the scripted replay has not applied yet), STROBE's bookkeeping `meta`
BP_HD void rp_ts_emit(uint32_t p, const kstate &st, uint32_t meta, uint32_t *ts_out, const uint32_t *pending_mask = nullptr) {
    uint32_t *o = ts_out + (uint64_t)p * BP_TS_WORDS;
    for (uint32_t i = 0; i < 50; i++) o[i] = ks_get32(st, i) ^ ((pending_mask && i < RS_MASK_WORDS) ? pending_mask[i] : 0u);
    o[50] = meta;
    o[51] = 0;
}
#define RP_NO_STOP 0xffffffffu

// ---- stage 1: parse + transcript ------------------------------------------------------
// thread p.  `st` = this lane's 50-word sponge state (LDS on the device).
// Outputs: fields (plain scalars) and status[p] (0, VerificationError, FormatError).  The per-proof points
// (A,S,T1,T2,L_*,R_*,V_*) are decoded straight from the proof bytes by rp_points_thread, which does not
// depend on the transcript and therefore shares a launch with it.
// ts_flags / ts_in / ts_out: caller-supplied transcripts (bpgpu_rangeproof_verify_batch_ts): the start state is
// ts_in[p] if given, else `init`; with BP_TS_DOMSEP it does not contain rangeproof_domain_sep(n, m) yet
// (transcript.rs:44-48), which is then applied here; ts_out[p] (optional) receives the advanced state.
// `in`: where this proof's bytes, commitments and rng bytes are (rp_resolve)
BP_HD void rp_transcript_thread(uint32_t p, rp_shape sh, const rp_strobe_init &init, kstate st, const rp_inputs &in, uint32_t *fields, uint32_t *status,
                                uint32_t ts_flags = 0, const uint32_t *ts_in = nullptr, uint32_t *ts_out = nullptr) {
    const uint32_t B = sh.nproofs, k = sh.k;
    const uint8_t *pr = in.pr;
    const rp_fields fl = rp_field_layout(k, sh.m);
    uint32_t w[8];
    // --- from_bytes: the five scalars must be canonical (mod.rs:519-524, ipp.rs:401-404)
    sc tx, txb, eb, a, b;
    bool fmt_ok = true;
    load_words8(tx.v, pr + 128);   fmt_ok = fmt_ok && sc_is_canonical_sc(tx);
    load_words8(txb.v, pr + 160);  fmt_ok = fmt_ok && sc_is_canonical_sc(txb);
    load_words8(eb.v, pr + 192);   fmt_ok = fmt_ok && sc_is_canonical_sc(eb);
    load_words8(a.v, pr + 224 + 64 * k);       fmt_ok = fmt_ok && sc_is_canonical_sc(a);
    load_words8(b.v, pr + 224 + 64 * k + 32);  fmt_ok = fmt_ok && sc_is_canonical_sc(b);
    if (!fmt_ok) {
        status_raise(status + p, BP_VERDICT_FORMAT);   // FormatError outranks whatever the point decoder reports
        rp_ts_passthrough(p, init, ts_in, ts_out);
        return;
    }
    // n m != 2^k (or m == 0) is found by verification_scalars (ipp.rs:203-211), i.e. AFTER the range proof's own part of the transcript:
    // a caller that wants its transcript back gets it as of that moment; InvalidBitsize / InvalidGeneratorsLength come before any
    // transcript operation (mod.rs:358-366)
    const bool shape_stop = sh.shape_verdict == BP_VERDICT_VERIFICATION && ts_out;
    if (sh.shape_verdict && !shape_stop) {
        status_raise(status + p, sh.shape_verdict);
        rp_ts_passthrough(p, init, ts_in, ts_out);
        return;
    }
    rp_store(fields, B, RPF_TX, p, tx);
    rp_store(fields, B, RPF_TXB, p, txb);
    rp_store(fields, B, RPF_EB, p, eb);
    rp_store(fields, B, RPF_A, p, a);
    rp_store(fields, B, RPF_B, p, b);

    strobe t;
    t.st = st;
    if (ts_in) {
        const uint32_t *src = ts_in + (uint64_t)p * BP_TS_WORDS;
        for (uint32_t i = 0; i < 50; i++) ks_set32(st, i, src[i]);
        const uint32_t meta = src[50];
        t.pos = meta & 0xffu;
        t.pos_begin = (meta >> 8) & 0xffu;
        t.cur_flags = (meta >> 16) & 0xffu;
    } else {
        const uint32_t *iw = in.init_w ? in.init_w : init.w;
        for (uint32_t i = 0; i < 50; i++) ks_set32(st, i, iw[i]);
        t.pos = init.pos;
        t.pos_begin = init.pos_begin;
        t.cur_flags = init.cur_flags;
    }
    if (ts_flags & BP_TS_DOMSEP) {   // rangeproof_domain_sep(n, m), transcript.rs:44-48
        const uint8_t dsep[7] = {'d', 'o', 'm', '-', 's', 'e', 'p'}, l_n[1] = {'n'}, l_m[1] = {'m'};
        const uint8_t rpv1[13] = {'r', 'a', 'n', 'g', 'e', 'p', 'r', 'o', 'o', 'f', ' ', 'v', '1'};
        merlin_append_message(t, dsep, 7, rpv1, 13);
        merlin_append_u64(t, l_n, 1, sh.n);
        merlin_append_u64(t, l_m, 1, sh.m);
    }

    bool verr = false;
    const uint8_t lV[1] = {'V'}, lA[1] = {'A'}, lS[1] = {'S'}, ly[1] = {'y'}, lz[1] = {'z'}, lx[1] = {'x'}, lw[1] = {'w'},
                  lL[1] = {'L'}, lR[1] = {'R'}, lu[1] = {'u'}, ln[1] = {'n'};
    const uint8_t lT1[3] = {'T', '_', '1'}, lT2[3] = {'T', '_', '2'}, ltx[3] = {'t', '_', 'x'};
    const uint8_t ltxb[12] = {'t', '_', 'x', '_', 'b', 'l', 'i', 'n', 'd', 'i', 'n', 'g'};
    const uint8_t leb[10] = {'e', '_', 'b', 'l', 'i', 'n', 'd', 'i', 'n', 'g'};
    const uint8_t ldom[7] = {'d', 'o', 'm', '-', 's', 'e', 'p'};
    const uint8_t lipp[6] = {'i', 'p', 'p', ' ', 'v', '1'};

    // V_j: append_point, no identity check (mod.rs:370-374)
    for (uint32_t j = 0; j < sh.m; j++) {
        load_words8(w, in.cm + (uint64_t)j * 32);
        merlin_append_words8(t, lV, 1, w);
    }
    // A, S: validate_and_append_point (transcript.rs:75-87) -- the first identity encoding ends the reference's replay BEFORE the message:
    // the state handed back is the one of that moment (the replay itself goes on: the later launches read every field)
    bool ts_stopped = false;
#define RP_VALIDATE(wrec)                                                                              \
    if (words8_zero(wrec)) {                                                                           \
        if (ts_out && !ts_stopped) rp_ts_emit(p, st, rp_ts_meta(t.pos, t.pos_begin, t.cur_flags), ts_out); \
        ts_stopped = true;                                                                             \
        verr = true;                                                                                   \
    }
    load_words8(w, pr + 0);
    RP_VALIDATE(w)
    merlin_append_words8(t, lA, 1, w);
    load_words8(w, pr + 32);
    RP_VALIDATE(w)
    merlin_append_words8(t, lS, 1, w);
    sc y, z, x, wch, c;
    rp_challenge_scalar(t, ly, 1, y);
    rp_challenge_scalar(t, lz, 1, z);
    load_words8(w, pr + 64);
    RP_VALIDATE(w)
    merlin_append_words8(t, lT1, 3, w);
    load_words8(w, pr + 96);
    RP_VALIDATE(w)
    merlin_append_words8(t, lT2, 3, w);
    rp_challenge_scalar(t, lx, 1, x);
    merlin_append_words8(t, ltx, 3, tx.v);
    merlin_append_words8(t, ltxb, 12, txb.v);
    merlin_append_words8(t, leb, 10, eb.v);
    rp_challenge_scalar(t, lw, 1, wch);
    // batching challenge c = Scalar::random(rng) (mod.rs:396): 64 rng bytes, wide-reduced
    {
        uint32_t cw[16];
        const uint8_t *rs = in.rs;
        if (rs) {
            load_words8(cw, rs);
            load_words8(cw + 8, rs + 32);
        } else {
            rp_seed_words(cw, sh, p, RP_SEED_RNG);
        }
        sc_from_wide(c, cw);
    }
    rp_store(fields, B, RPF_Y, p, y);
    rp_store(fields, B, RPF_Z, p, z);
    rp_store(fields, B, RPF_X, p, x);
    rp_store(fields, B, RPF_W, p, wch);
    rp_store(fields, B, RPF_C, p, c);
    if (shape_stop) {   // verification_scalars returns Err before innerproduct_domain_sep (ipp.rs:203-213)
        if (!ts_stopped) rp_ts_emit(p, st, rp_ts_meta(t.pos, t.pos_begin, t.cur_flags), ts_out);
        status_raise(status + p, sh.shape_verdict);
        return;
    }
    // inner-product part (ipp.rs:213-222)
    merlin_append_message(t, ldom, 7, lipp, 6);
    merlin_append_u64(t, ln, 1, sh.nm);
    for (uint32_t i = 0; i < k; i++) {
        load_words8(w, pr + 224 + 64 * i);
        RP_VALIDATE(w)
        merlin_append_words8(t, lL, 1, w);
        load_words8(w, pr + 224 + 64 * i + 32);
        RP_VALIDATE(w)
        merlin_append_words8(t, lR, 1, w);
        sc u;
        rp_challenge_scalar(t, lu, 1, u);
        rp_store(fields, B, fl.u + i, p, u);
    }
#undef RP_VALIDATE
    if (verr) status_raise(status + p, BP_VERDICT_VERIFICATION);
    if (ts_out && !ts_stopped) rp_ts_emit(p, st, rp_ts_meta(t.pos, t.pos_begin, t.cur_flags), ts_out);
}

// ---- stage 1, scripted: the same replay driven by the per-shape script of rp_script.h ------------------------------------
// Bit-identical to rp_transcript_thread when every proof of the launch starts from the same state `init` (label mode, or one
// transcript shared by the batch).  All positions come from the script, i.e. are uniform across the wavefront: framing bytes
// arrive as one XOR mask per permutation, 32-byte records are XORed in as words, a challenge is state words 0..15.
BP_HD void rp_script_xor_record(const kstate &st, uint32_t pos, const uint32_t w[8]) {
    const uint32_t wi = pos >> 2, sh = (pos & 3) * 8;
    if (sh == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) st.w[(wi + i) * st.stride] ^= w[i];
    } else {
        st.w[wi * st.stride] ^= w[0] << sh;
#pragma unroll
        for (int i = 1; i < 8; i++) st.w[(wi + i) * st.stride] ^= (w[i] << sh) | (w[i - 1] >> (32 - sh));
        st.w[(wi + 8) * st.stride] ^= w[7] >> (32 - sh);
    }
}
// ts_in (optional): one caller-supplied start state per proof -- allowed here when all of them sit at the SAME STROBE position
// (pos, pos_begin, cur_flags: what the script was compiled for; the pool's combining queue groups requests that way); only the
// 50 sponge words differ from proof to proof.
BP_HD void rp_transcript_scripted(uint32_t p, rp_shape sh, const rp_strobe_init &init, kstate st, const rp_inputs &in, const rp_script_hdr *script,
                                  uint32_t *fields, uint32_t *status, uint32_t *ts_out = nullptr, const uint32_t *ts_in = nullptr) {
    const uint32_t B = sh.nproofs, k = sh.k;
    const uint8_t *pr = in.pr;
    const rp_fields fl = rp_field_layout(k, sh.m);
    uint32_t w[8];
    // --- from_bytes: the five scalars must be canonical (mod.rs:519-524, ipp.rs:401-404)
    sc tx, txb, eb, a, b;
    bool fmt_ok = true;
    load_words8(tx.v, pr + 128);   fmt_ok = fmt_ok && sc_is_canonical_sc(tx);
    load_words8(txb.v, pr + 160);  fmt_ok = fmt_ok && sc_is_canonical_sc(txb);
    load_words8(eb.v, pr + 192);   fmt_ok = fmt_ok && sc_is_canonical_sc(eb);
    load_words8(a.v, pr + 224 + 64 * k);       fmt_ok = fmt_ok && sc_is_canonical_sc(a);
    load_words8(b.v, pr + 224 + 64 * k + 32);  fmt_ok = fmt_ok && sc_is_canonical_sc(b);
    if (!fmt_ok) {
        status_raise(status + p, BP_VERDICT_FORMAT);
        rp_ts_passthrough(p, init, ts_in, ts_out);
        return;
    }
    if (sh.shape_verdict) {
        status_raise(status + p, sh.shape_verdict);
        rp_ts_passthrough(p, init, ts_in, ts_out);
        return;
    }
    rp_store(fields, B, RPF_TX, p, tx);
    rp_store(fields, B, RPF_TXB, p, txb);
    rp_store(fields, B, RPF_EB, p, eb);
    rp_store(fields, B, RPF_A, p, a);
    rp_store(fields, B, RPF_B, p, b);
    // validate_and_append_point (transcript.rs:75-87): A, S, T_1, T_2, L_i, R_i must not be the identity encoding.  The bytes are read
    // in transcript order (L_i, R_i interleaved as they lie in the proof): the first identity is where the reference's replay ends
    bool verr = false;
    uint32_t stop_u = RP_NO_STOP;
    for (uint32_t u = 0; u < 4 + 2 * k; u++) {
        load_words8(w, pr + (u < 4 ? 32 * u : 224 + 32 * (u - 4)));
        if (words8_zero(w)) {
            verr = true;
            if (stop_u == RP_NO_STOP) stop_u = u;
        }
    }
    if (ts_in) {
        const uint32_t *src = ts_in + (uint64_t)p * BP_TS_WORDS;
        for (uint32_t i = 0; i < 50; i++) ks_set32(st, i, src[i]);
    } else {
        const uint32_t *iw = in.init_w ? in.init_w : init.w;
        for (uint32_t i = 0; i < 50; i++) ks_set32(st, i, iw[i]);
    }
    const rp_script_op *ops = rp_script_ops(script);
    const uint32_t *masks = rp_script_masks(script);
    const uint32_t n_ops = script->n_ops;
    // the operation before which this proof's transcript is handed back (a rare, per-lane matter; the replay itself goes on)
    const uint32_t stop_op = (ts_out && stop_u != RP_NO_STOP) ? rp_script_stops(script)[stop_u].op : RP_NO_STOP;
    for (uint32_t oi = 0; oi < n_ops; oi++) {
        const rp_script_op op = ops[oi];
        if (oi == stop_op) {
            const rp_script_stop *sp = rp_script_stops(script) + stop_u;
            rp_ts_emit(p, st, sp->meta, ts_out, sp->mask);
        }
        const uint8_t *src = (op.src == RS_SRC_PROOF ? pr : in.cm) + op.off;
        if (op.kind == RS_MSG) {
            load_words8(w, src);
            rp_script_xor_record(st, op.pos, w);
        } else if (op.kind == RS_MSGB) {      // the part of a record before / after a rate boundary: bytes
            for (uint32_t q = 0; q < op.nbytes; q++) ks_xor8(st, op.pos + q, src[q]);
        } else if (op.kind == RS_PERM) {
            keccak_f1600_masked(st, masks + (uint64_t)op.arg * RS_MASK_WORDS, RS_MASK_WORDS);
        } else {                              // RS_CHAL: 64 squeezed bytes = words 0..15, zeroed behind the read (STROBE's PRF)
            uint32_t cw[16];
#pragma unroll
            for (int q = 0; q < 16; q++) {
                cw[q] = ks_get32(st, q);
                ks_set32(st, q, 0);
            }
            sc ch;
            sc_from_wide(ch, cw);
            const uint32_t id = op.arg;
            rp_store(fields, B, id == 0 ? (uint32_t)RPF_Y : (id == 1 ? (uint32_t)RPF_Z : (id == 2 ? (uint32_t)RPF_X : (id == 3 ? (uint32_t)RPF_W : fl.u + (id - 4)))), p, ch);
        }
    }
    // batching challenge c = Scalar::random(rng) (mod.rs:396): 64 rng bytes, wide-reduced
    {
        uint32_t cw[16];
        sc c;
        if (in.rs) {
            load_words8(cw, in.rs);
            load_words8(cw + 8, in.rs + 32);
        } else {
            rp_seed_words(cw, sh, p, RP_SEED_RNG);
        }
        sc_from_wide(c, cw);
        rp_store(fields, B, RPF_C, p, c);
    }
    if (verr) status_raise(status + p, BP_VERDICT_VERIFICATION);
    if (ts_out && stop_op == RP_NO_STOP) rp_ts_emit(p, st, rp_ts_meta(script->end_pos, script->end_pos_begin, script->end_flags), ts_out);
}

// The same replay for NARROW chains, 32 lanes per proof (keccak.h: keccak_f1600_masked_coop): the group's LEADER (lane 0 of the
// half-wavefront) does everything a lane of rp_transcript_scripted does except the permutation, which all lanes of the group run
// together on the leader's sponge state (`st`: the same LDS words in all lanes of the group, stride 1).  Every lane of the workgroup
// walks the whole script -- the barriers and the exchanges need them all -- whatever happened to its proof: a rejected proof's
// group permutes a state nobody reads.  valid: the group has a proof at all.
#if defined(__HIP_DEVICE_COMPILE__)
#define BP_GROUP_SYNC() __syncthreads()
#else
#define BP_GROUP_SYNC() ((void)0)
#endif
// cp (narrow chains, option coop_split): the leader does not reduce the 64-byte challenges mod l between the permutations (~2 us each on a
// lone lane, 5 + k of them: nothing in a VERIFIER's transcript depends on a challenge's value) -- it parks the raw words, and the group's
// lanes reduce one each afterwards (rp_coop_reduce_lane).  ops_l / masks_l: the script's operations and masks staged in LDS by the caller
// (every operation fetched from global memory is a dependent ~0.5 us on the leader's path; ~50 operations per proof).
#define RP_COOP_STAGE_WORDS 640    // proof + commitments staged per group (2560 bytes: up to (64, 32), 992 + 1024)
#define RP_COOP_OPS_CAP 160        // operations of a script staged per workgroup (16 bytes each; (64, 1): 53, (64, 32): 92)
#define RP_COOP_MASKS_CAP 24       // ... and its permutation masks (168 bytes each; (64, 1): 13)
struct rp_chal_park {
    uint32_t *raw;    // [5 + k][16]: y, z, x, w, u_0 .. u_{k-1}, then the 64 rng bytes of the batching challenge c
    uint32_t *flag;   // [1]: bit 0 -- the leader replayed the whole script (the parked words are valid); bit 1 -- and the proof is still undecided
};
BP_HD void rp_transcript_scripted_coop(uint32_t p, bool valid, uint32_t lane, rp_shape sh, const rp_strobe_init &init, kstate st, const rp_inputs &in,
                                       const rp_script_hdr *script, uint32_t *fields, uint32_t *status, uint32_t *ts_out = nullptr,
                                       const uint32_t *ts_in = nullptr, const rp_chal_park *cp = nullptr, const rp_script_op *ops_l = nullptr,
                                       const uint32_t *masks_l = nullptr) {
    const uint32_t B = sh.nproofs, k = sh.k;
    const uint8_t *pr = in.pr;
    const rp_fields fl = rp_field_layout(k, sh.m);
    uint32_t w[8];
    bool run = valid && (lane & 31) == 0;   // this lane does the per-proof work
    bool verr = false;
    uint32_t stop_u = RP_NO_STOP;
    if (run) {
        sc tx, txb, eb, a, b;
        bool fmt_ok = true;
        load_words8(tx.v, pr + 128);   fmt_ok = fmt_ok && sc_is_canonical_sc(tx);
        load_words8(txb.v, pr + 160);  fmt_ok = fmt_ok && sc_is_canonical_sc(txb);
        load_words8(eb.v, pr + 192);   fmt_ok = fmt_ok && sc_is_canonical_sc(eb);
        load_words8(a.v, pr + 224 + 64 * k);       fmt_ok = fmt_ok && sc_is_canonical_sc(a);
        load_words8(b.v, pr + 224 + 64 * k + 32);  fmt_ok = fmt_ok && sc_is_canonical_sc(b);
        if (!fmt_ok || sh.shape_verdict) {
            status_raise(status + p, fmt_ok ? sh.shape_verdict : (uint32_t)BP_VERDICT_FORMAT);
            rp_ts_passthrough(p, init, ts_in, ts_out);
            run = false;
        } else {
            rp_store(fields, B, RPF_TX, p, tx);
            rp_store(fields, B, RPF_TXB, p, txb);
            rp_store(fields, B, RPF_EB, p, eb);
            rp_store(fields, B, RPF_A, p, a);
            rp_store(fields, B, RPF_B, p, b);
            for (uint32_t u = 0; u < 4 + 2 * k; u++) {
                load_words8(w, pr + (u < 4 ? 32 * u : 224 + 32 * (u - 4)));
                if (words8_zero(w)) {
                    verr = true;
                    if (stop_u == RP_NO_STOP) stop_u = u;
                }
            }
            if (ts_in) {
                const uint32_t *src = ts_in + (uint64_t)p * BP_TS_WORDS;
                for (uint32_t i = 0; i < 50; i++) ks_set32(st, i, src[i]);
            } else {
                const uint32_t *iw = in.init_w ? in.init_w : init.w;
                for (uint32_t i = 0; i < 50; i++) ks_set32(st, i, iw[i]);
            }
        }
    }
    const rp_script_op *ops = ops_l ? ops_l : rp_script_ops(script);
    const uint32_t *masks = masks_l ? masks_l : rp_script_masks(script);
    const uint32_t n_ops = script->n_ops;
    const uint32_t stop_op = (run && ts_out && stop_u != RP_NO_STOP) ? rp_script_stops(script)[stop_u].op : RP_NO_STOP;   // (leader only)
    for (uint32_t oi = 0; oi < n_ops; oi++) {
        const rp_script_op op = ops[oi];
        if (oi == stop_op) {   // (between permutations: the leader's sponge words are at rest)
            const rp_script_stop *sp = rp_script_stops(script) + stop_u;
            rp_ts_emit(p, st, sp->meta, ts_out, sp->mask);
        }
        if (op.kind == RS_PERM) {
            BP_GROUP_SYNC();
            keccak_f1600_masked_coop(st, masks + (uint64_t)op.arg * RS_MASK_WORDS, RS_MASK_WORDS, lane);
            BP_GROUP_SYNC();
        } else if (run) {
            const uint8_t *src = (op.src == RS_SRC_PROOF ? pr : in.cm) + op.off;
            if (op.kind == RS_MSG) {
                load_words8(w, src);
                rp_script_xor_record(st, op.pos, w);
            } else if (op.kind == RS_MSGB) {
                for (uint32_t q = 0; q < op.nbytes; q++) ks_xor8(st, op.pos + q, src[q]);
            } else if (cp) {
                uint32_t *dst = cp->raw + 16 * op.arg;
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    dst[q] = ks_get32(st, q);
                    ks_set32(st, q, 0);
                }
            } else {
                uint32_t cw[16];
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    cw[q] = ks_get32(st, q);
                    ks_set32(st, q, 0);
                }
                sc ch;
                sc_from_wide(ch, cw);
                const uint32_t id = op.arg;
                rp_store(fields, B, id == 0 ? (uint32_t)RPF_Y : (id == 1 ? (uint32_t)RPF_Z : (id == 2 ? (uint32_t)RPF_X : (id == 3 ? (uint32_t)RPF_W : fl.u + (id - 4)))), p, ch);
            }
        }
    }
    if (!run) return;
    {
        uint32_t cw[16];
        sc c;
        if (in.rs) {
            load_words8(cw, in.rs);
            load_words8(cw + 8, in.rs + 32);
        } else {
            rp_seed_words(cw, sh, p, RP_SEED_RNG);
        }
        if (cp) {
#pragma unroll
            for (int q = 0; q < 16; q++) cp->raw[16 * (4 + k) + q] = cw[q];
            cp->flag[0] = verr ? 1u : 3u;
        } else {
            sc_from_wide(c, cw);
            rp_store(fields, B, RPF_C, p, c);
        }
    }
    if (verr) status_raise(status + p, BP_VERDICT_VERIFICATION);
    if (ts_out && stop_op == RP_NO_STOP) rp_ts_emit(p, st, rp_ts_meta(script->end_pos, script->end_pos_begin, script->end_flags), ts_out);
}
// lane id = 0 .. 4 + k of the group, after the leader's replay: reduce parked challenge `id` mod l (Scalar::from_bytes_mod_order_wide,
// transcript.rs:93) into its field; y and the u_i are also parked in canonical form for the lanes that invert them (rp_split_invert_lane)
BP_HD void rp_coop_reduce_lane(uint32_t id, uint32_t p, const rp_shape &sh, const rp_chal_park &cp, uint32_t *fields, uint32_t *park) {
    const uint32_t B = sh.nproofs, k = sh.k;
    if (!(cp.flag[0] & 1u) || id > 4 + k) return;
    const rp_fields fl = rp_field_layout(k, sh.m);
    uint32_t cw[16];
#pragma unroll
    for (int q = 0; q < 16; q++) cw[q] = cp.raw[16 * id + q];
    sc ch;
    sc_from_wide(ch, cw);
    rp_store(fields, B, id == 0 ? (uint32_t)RPF_Y : (id == 1 ? (uint32_t)RPF_Z : (id == 2 ? (uint32_t)RPF_X : (id == 3 ? (uint32_t)RPF_W : (id == 4 + k ? (uint32_t)RPF_C : fl.u + (id - 4))))), p, ch);
    if (park && (id == 0 || (id >= 4 && id < 4 + k))) {
        uint32_t *dst = park + 8 * (id == 0 ? k : id - 4);
#pragma unroll
        for (int q = 0; q < 8; q++) dst[q] = ch.v[q];
    }
}

// ---- stage 1b: per-proof points -----------------------------------------------------------
// unique term u of proof p, in the order A, S, T_1, T_2, L_0..L_{k-1}, R_0..R_{k-1}, V_0..V_{m-1}
// (the non-generator part of mod.rs:433-443)
BP_HD const uint8_t *rp_unique_point_ptr(const rp_shape &sh, const rp_inputs &in, uint32_t u) {
    const uint32_t k = sh.k;
    if (u < 4) return in.pr + 32 * u;
    if (u < 4 + k) return in.pr + 224 + 64 * (u - 4);
    if (u < 4 + 2 * k) return in.pr + 224 + 64 * (u - 4 - k) + 32;
    return in.cm + (uint64_t)(u - 4 - 2 * k) * 32;
}
// thread t = p * U + u: decode the point (mod.rs:433-443 .decompress()) and build its {1..8}P table.
// An undecodable point is the Option::None of optional_multiscalar_mul -> VerificationError (mod.rs:445).
// pts (optional, instead of tab): bucket path (bucket.h) -- store the point as one affine Niels record instead.
// (p = t / U; `in` = rp_resolve(p, ...))
BP_HD void rp_points_thread(uint32_t t, rp_shape sh, const rp_inputs &in, ge_cached *tab, uint32_t *status, fb_entry *pts = nullptr) {
    const uint32_t p = t / sh.U, u = t - p * sh.U;
    uint32_t w[8];
    load_words8(w, rp_unique_point_ptr(sh, in, u));
    ge_ext pt;
    // (the short-register decode of bucket2.h here: 117 -> 81 spilled registers in k_rp_stage1<true>, no difference in any bench form --
    // profiles/r06/stage1_short_register_decode_ab.txt; not kept)
    if (!ristretto_decompress(pt, w)) status_raise(status + p, BP_VERDICT_VERIFICATION);
    if (pts) bk_store_point(pts + t, pt);
    else if (sh.radix5) vb_build_table16(tab + 16 * (uint64_t)t, pt, sh.a_outside && u == 0);
    else if (sh.a_outside && u == 0) ge_to_cached(tab[8 * (uint64_t)t], pt);
    else vb_build_table(tab + 8 * (uint64_t)t, pt);
}

// sum_{i<2^lg} x^i by repeated doubling (src/util.rs:240-256); Montgomery form in and out
BP_HD void rp_sum_of_powers_pow2(sc28 &r, const sc28 &xm, uint32_t lg) {
    sc28 one_m;
    sc28_one_mont(one_m);
    if (lg == 0) {
        r = one_m;
        return;
    }
    // result = 1 + x (lazy limb-wise add: both < 2^254 -> renormalise through a multiplication by 1)
    sc a, b, sum;
    sc_from_sc28(a, one_m);
    sc_from_sc28(b, xm);
    sc_add(sum, a, b);
    sc28 result, factor = xm, t;
    sc28_from_sc(result, sum);
    for (uint32_t i = 1; i < lg; i++) {
        sc28_montmul(factor, factor, factor);
        sc28_montmul(t, factor, result);
        sc_from_sc28(a, t);
        sc_from_sc28(b, result);
        sc_add(sum, a, b);
        sc28_from_sc(result, sum);
    }
    r = result;
}

BP_HD void store_words8(uint32_t *dst, const sc &s) {
#pragma unroll
    for (int q = 0; q < 8; q++) dst[q] = s.v[q];
}
// a canonical coefficient of a per-proof point, stored as its signed radix-16 recoding (msm_vb.h)
BP_HD void store_recoded(uint32_t *dst, const sc &s) {
    uint32_t r[8];
    sc_recode16(r, s.v);
#pragma unroll
    for (int q = 0; q < 8; q++) dst[q] = r[q];
}

// a coefficient held in Montgomery form -> (times the proof's batch weight, if any) -> coefficient u of the proof's
// list `us`: radix-16 recoding (8 words per coefficient), or -- bk_c != 0, bucket path -- the c-bit window recoding
// of bucket.h (BK_RWORDS words per coefficient)
BP_HD void rp_emit_coeff(uint32_t *us, uint32_t u, const sc28 &vm, const sc28 *rho_m, uint32_t bk_c, uint32_t salt = 0, bool radix5 = false) {
    sc28 t = vm;
    if (rho_m) sc28_montmul(t, vm, *rho_m);
    sc s;
    sc_from_mont28(s, t);
    if (bk_c) {
        uint32_t r[BK_RWORDS];
        bk_recode(r, s.v, bk_make(bk_c), salt + u);
#pragma unroll
        for (int q = 0; q < BK_RWORDS; q++) us[u * BK_RWORDS + q] = r[q];
    } else if (radix5) {
        uint32_t r[8];
        sc_recode32(r, s.v);
#pragma unroll
        for (int q = 0; q < 8; q++) us[u * 8 + q] = r[q];
    } else {
        store_recoded(us + u * 8, s);
    }
}

// The scalar role of a NARROW chain has 31 idle lanes beside its leader: instead of recoding its U coefficients one after the other
// (from Montgomery form, times the weight in batch-combination mode, into signed radix-16 digits: ~1.1 us each, U = 17 at (64, 1)) the leader
// parks them in LDS and the lanes of its group recode one each.  slot[u] = coefficient u in Montgomery form, slot[RP_DEFER_CAP] = the
// weight (meta[1] != 0), meta[0] = the leader got as far as parking them (a rejected proof parks nothing).
#define RP_DEFER_CAP 96
struct rp_defer {
    sc28 *slot;       // [RP_DEFER_CAP + 1]
    uint32_t *meta;   // [2]
};
BP_HD void rp_emit_deferred(uint32_t lane32, uint32_t p, const rp_shape &sh, uint32_t *recoded, const rp_defer &df, uint32_t bk_c) {
    if (!df.meta[0]) return;
    uint32_t *us = recoded + (uint64_t)p * sh.U * (bk_c ? BK_RWORDS : 8);
    const sc28 *rho = df.meta[1] ? &df.slot[RP_DEFER_CAP] : nullptr;
    for (uint32_t u = lane32; u < sh.U; u += 32) {
        if (sh.a_outside && u == 0) continue;   // (wide chains add A after the Horner chain: nothing was parked for it)
        const sc28 v = df.slot[u];
        rp_emit_coeff(us, u, v, rho, bk_c, p * sh.U, sh.radix5 != 0);
    }
}

// ---- stage 2: per-proof scalars -----------------------------------------------------------
// thread p.  Writes the radix-16 recodings of the U per-proof coefficients (recoded[p][U][8], order of
// rp_unique_point_ptr), the Montgomery-form tables for stage 3, and the digits of the B_blinding (row 0) and
// B (row 1) coefficients.
// rho64 (optional): batch-combination mode (bpgpu_rangeproof_verify_rlc): every coefficient of proof p is
// multiplied by its weight rho_p = from_bytes_mod_order_wide(rho64[p]); the B_blinding / B coefficients go to
// the ROW0 / ROW1 fields instead of `digits` (they are summed over the batch in the next launch).
// skip (narrow chains, option coop_split; never with weights): RP_SKIP_INV -- the inversions, u_i^2, u_i^-2 and the y^-(2^b) table were
// written by the group's lanes (rp_split_invert_lane); RP_SKIP_ROWS -- the B_blinding / B coefficients are formed in launch 3 (rp_rows_thread).
enum { RP_SKIP_INV = 1, RP_SKIP_ROWS = 2 };
BP_HD void rp_expand_a_thread(uint32_t p, rp_shape sh, fb_params prm, uint32_t lg_m, uint32_t *fields, uint32_t *recoded,
                              fb_digit *digits, const uint32_t *status, const uint8_t *rho64 = nullptr, uint32_t bk_c = 0, const rp_defer *df = nullptr,
                              uint32_t skip = 0) {
    if (status[p] != 0) return;   // digits/scalars of rejected proofs are never consumed (finish masks them)
    const uint32_t B = sh.nproofs, k = sh.k;
    const rp_fields fl = rp_field_layout(k, sh.m);
    sc t0, t1;
    sc28 rho_m;
    const sc28 *rho = nullptr;
    if (rho64 || (sh.seeded & RP_SEED_WEIGHTS)) {
        uint32_t w16[16];
        if (rho64) {
            const uint32_t *src = (const uint32_t *)(rho64 + 64 * (uint64_t)p);
#pragma unroll
            for (int i = 0; i < 16; i++) w16[i] = src[i];
        } else {
            rp_seed_words(w16, sh, p, RP_SEED_WEIGHTS);
        }
        sc r;
        sc_from_wide(r, w16);
        sc_to_mont28(rho_m, r);
        rho = &rho_m;
    }
    sc28 ym;
    if (!(skip & RP_SKIP_INV) || !(skip & RP_SKIP_ROWS)) {
        sc y;
        rp_load(y, fields, B, RPF_Y, p);
        sc_to_mont28(ym, y);
    }
#define RP_EMIT(idx, val)                                                                 \
    do {                                                                                  \
        if (df) df->slot[(idx)] = (val);                                                  \
        else rp_emit_coeff(recoded + (uint64_t)p * sh.U * (bk_c ? BK_RWORDS : 8), (idx), (val), rho, bk_c, p * sh.U, sh.radix5 != 0);        \
    } while (0)

    // batch inversion of (y, u_0, .., u_{k-1}) (Scalar::batch_invert, ipp.rs:226-227; y.invert(), mod.rs:414),
    // everything in Montgomery form; prefix products are parked in the uinv_m slots (overwritten below)
    if (!(skip & RP_SKIP_INV)) {
    sc28 acc = ym, um;
    sc u;
    for (uint32_t i = 0; i < k; i++) {
        rp_store28(fields, B, fl.uinv_m + i, p, acc);     // prefix before u_i
        rp_load(u, fields, B, fl.u + i, p);
        sc_to_mont28(um, u);
        rp_store28(fields, B, fl.u_m + i, p, um);
        sc28_montmul(acc, acc, um);
    }
    sc28 inv;
    sc28_invert_mont_safegcd(inv, acc);                   // (y * prod u_i)^-1
    for (uint32_t ii = k; ii-- > 0;) {
        sc28 pre, uim, sq;
        rp_load28(pre, fields, B, fl.uinv_m + ii, p);
        rp_load28(um, fields, B, fl.u_m + ii, p);
        sc28_montmul(uim, inv, pre);                      // u_ii^-1
        sc28_montmul(inv, inv, um);                       // drop u_ii from the running inverse
        rp_store28(fields, B, fl.uinv_m + ii, p, uim);
        sc28_montmul(sq, um, um);                         // u_i^2   -> L_i coefficient
        RP_EMIT(4 + ii, sq);
        sc28_montmul(sq, uim, uim);                       // u_i^-2  -> R_i coefficient
        RP_EMIT(4 + k + ii, sq);
    }
    // what is left in inv is y^-1: table of y^-(2^b)
    {
        sc28 pw = inv;
        for (uint32_t bb = 0; bb < k; bb++) {
            rp_store28(fields, B, fl.yinvp_m + bb, p, pw);
            sc28_montmul(pw, pw, pw);
        }
    }
    }   // (!RP_SKIP_INV)
    if (df) {   // the coefficients are parked for the group's lanes (rp_emit_deferred)
        df->meta[1] = rho ? 1u : 0u;
        if (rho) df->slot[RP_DEFER_CAP] = rho_m;
    }
    // (the remaining inputs are loaded only now: nothing but y and the u_i is live across the inversion)
    sc z, x, w, c, tx, txb, eb, a, b;
    rp_load(z, fields, B, RPF_Z, p);
    rp_load(x, fields, B, RPF_X, p);
    rp_load(w, fields, B, RPF_W, p);
    rp_load(c, fields, B, RPF_C, p);
    rp_load(tx, fields, B, RPF_TX, p);
    rp_load(txb, fields, B, RPF_TXB, p);
    rp_load(eb, fields, B, RPF_EB, p);
    rp_load(a, fields, B, RPF_A, p);
    rp_load(b, fields, B, RPF_B, p);
    sc28 zm, xm, wm, cm, txbm, am, bm;
    sc_to_mont28(zm, z);
    sc_to_mont28(xm, x);
    sc_to_mont28(wm, w);
    sc_to_mont28(cm, c);
    sc_to_mont28(txbm, txb);
    sc_to_mont28(am, a);
    sc_to_mont28(bm, b);
    sc28 zzm;
    sc28_montmul(zzm, zm, zm);
    sc zz, minus_z;
    sc_from_mont28(zz, zzm);
    sc_neg(minus_z, z);
    rp_store(fields, B, RPF_ZZ, p, zz);
    rp_store28(fields, B, RPF_Z_M, p, zm);
    rp_store28(fields, B, RPF_ZZ_M, p, zzm);
    if (rho) {   // what stage 3 reads (z, -z, a, b, z^2 z^j) carries the weight; g_i and h_i are linear in them
        sc28 t;
        sc zr;
        sc28_montmul(t, zm, rho_m);
        sc_from_mont28(zr, t);
        sc_neg(minus_z, zr);
        rp_store(fields, B, RPF_Z, p, zr);
        rp_store(fields, B, RPF_MINUS_Z, p, minus_z);
        sc28_montmul(t, am, rho_m);
        rp_store28(fields, B, RPF_A_M, p, t);
        sc28_montmul(t, bm, rho_m);
        rp_store28(fields, B, RPF_B_M, p, t);
    } else {
        rp_store(fields, B, RPF_MINUS_Z, p, minus_z);
        rp_store28(fields, B, RPF_A_M, p, am);
        rp_store28(fields, B, RPF_B_M, p, bm);
    }
    // unique coefficients: 1, x, c x, c x^2 (A, S, T_1, T_2)
    sc28 cxm, cxxm;
    sc28_montmul(cxm, cm, xm);
    sc28_montmul(cxxm, cxm, xm);
    if (!sh.a_outside) {   // (wide chains add A, coefficient 1, after the Horner chain)
        sc28 one_m;
        sc28_one_mont(one_m);
        RP_EMIT(0, one_m);
    }
    RP_EMIT(1, xm);
    RP_EMIT(2, cxm);
    RP_EMIT(3, cxxm);
    // V_j coefficients c z^2 z^j, and the z^2 z^j table
    {
        sc28 czzj, zzj = zzm;
        sc28_montmul(czzj, cm, zzm);
        for (uint32_t j = 0; j < sh.m; j++) {
            RP_EMIT(4 + 2 * k + j, czzj);
            if (rho) {
                sc28 t;
                sc28_montmul(t, zzj, rho_m);
                rp_store28(fields, B, fl.zzzj_m + j, p, t);
            } else {
                rp_store28(fields, B, fl.zzzj_m + j, p, zzj);
            }
            sc28_montmul(czzj, czzj, zm);
            sc28_montmul(zzj, zzj, zm);
        }
    }
    // B_blinding coefficient: -e_blinding - c t_x_blinding  (row 0)
    if (!(skip & RP_SKIP_ROWS)) {
        sc28 pm;
        sc28_montmul(pm, cm, txbm);
        sc_from_mont28(t0, pm);
        sc_add(t0, t0, eb);
        sc_neg(t0, t0);
        if (rho) {
            sc28 t;
            sc_to_mont28(t, t0);
            sc28_montmul(t, t, rho_m);
            sc_from_mont28(t0, t);
            rp_store(fields, B, RPF_ROW0, p, t0);
        } else {
            fb_recode(digits + ((uint64_t)0 * prm.nwin) * B + p, B, t0.v, prm);
        }
    }
    // B coefficient: w (t_x - a b) + c (delta(y,z) - t_x)  (row 1)
    if (!(skip & RP_SKIP_ROWS)) {
        sc28 abm, pm, sum_ym, sum_zm, dm;
        sc ab, dl, sum_2;
        sc28_montmul(abm, am, bm);
        sc_from_mont28(ab, abm);
        sc_sub(t0, tx, ab);
        sc_to_mont28(pm, t0);
        sc28_montmul(pm, wm, pm);
        sc_from_mont28(t0, pm);                               // w (t_x - a b)
        rp_sum_of_powers_pow2(sum_ym, ym, k);
        rp_sum_of_powers_pow2(sum_zm, zm, lg_m);
        {   // sum_{i<n} 2^i = 2^n - 1, n <= 64
            sc_0(sum_2);
            if (sh.n >= 64) { sum_2.v[0] = 0xffffffffu; sum_2.v[1] = 0xffffffffu; }
            else if (sh.n >= 32) { sum_2.v[0] = 0xffffffffu; sum_2.v[1] = (sh.n == 32) ? 0u : ((1u << (sh.n - 32)) - 1u); }
            else sum_2.v[0] = (1u << sh.n) - 1u;
        }
        // delta = (z - z^2) <1, y^nm> - z^3 <1, 2^n> sum_j z^j        (mod.rs:587-593)
        sc_sub(t1, z, zz);
        sc_to_mont28(dm, t1);
        sc28_montmul(dm, dm, sum_ym);
        sc_from_mont28(dl, dm);
        sc28 z3m, s2m;
        sc28_montmul(z3m, zzm, zm);
        sc_to_mont28(s2m, sum_2);
        sc28_montmul(z3m, z3m, s2m);
        sc28_montmul(z3m, z3m, sum_zm);
        sc_from_mont28(t1, z3m);
        sc_sub(dl, dl, t1);
        sc_sub(t1, dl, tx);
        sc_to_mont28(pm, t1);
        sc28_montmul(pm, cm, pm);
        sc_from_mont28(t1, pm);                               // c (delta - t_x)
        sc_add(t0, t0, t1);
        if (rho) {
            sc28 t;
            sc_to_mont28(t, t0);
            sc28_montmul(t, t, rho_m);
            sc_from_mont28(t0, t);
            rp_store(fields, B, RPF_ROW1, p, t0);
        } else {
            fb_recode(digits + ((uint64_t)1 * prm.nwin) * B + p, B, t0.v, prm);
        }
    }
#undef RP_EMIT
    if (df) df->meta[0] = 1;   // every coefficient is parked
}

// ---- the scalar role of a NARROW chain, split (option coop_split; per-proof check only, never with weights) -----------------------
// After its transcript the leader of a 32-lane group used to spend ~125 us alone: 6 prefix products, one inversion by division steps
// (~60 us), 4 products per round to peel the u_i^-1 off again, the y^-(2^b) table, and ~35 products for the two basepoint coefficients.
// Split three ways:
//   * lanes 0 .. k of the group invert ONE value each -- u_0 .. u_{k-1} and y -- in lockstep (the division steps have a fixed structure,
//     scinv.h), then square their value and its inverse: no prefix / suffix products at all, and lane k runs the y^-(2^b) squarings;
//   * the leader forms what is left of rp_expand_a_thread (the x, z, c coefficients, the z tables);
//   * the B_blinding / B coefficients -- the longest chain of products, needed by launch 4 only -- are a role of launch 3 (rp_rows_thread).
// Same values mod l everywhere (the Montgomery-form fields are lazy representatives and may differ in their limbs; every digit that
// leaves the stage is recoded from a canonical scalar).
struct rp_split {
    uint32_t *park;   // [32][8]: canonical u_0 .. u_{k-1}, y -- parked by the lanes that reduced them (rp_coop_reduce_lane)
    uint32_t *go;     // [1]: the proof is still undecided
};
// lane i = 0 .. k of the group (the others return at once)
BP_HD void rp_split_invert_lane(uint32_t i, uint32_t p, const rp_shape &sh, uint32_t *fields, uint32_t *recoded, const rp_split &sp, const rp_defer *df) {
    const uint32_t B = sh.nproofs, k = sh.k;
    if (!sp.go[0] || i > k) return;
    const rp_fields fl = rp_field_layout(k, sh.m);
    sc v, vi;
#pragma unroll
    for (int q = 0; q < 8; q++) v.v[q] = sp.park[8 * i + q];
#if defined(BP_EXP_NOINV)   // timing experiments only (tools/r06/call32.sh builds variants with one phase compiled out; results are then wrong)
    vi = v;
#elif defined(BP_EXP_CTINV)
    sc_invert_safegcd(vi, v);
#else
    sc_invert_safegcd_var(vi, v);
#endif
    sc28 vm, im, sq, isq;
    sc_to_mont28(vm, v);
    sc_to_mont28(im, vi);
    sc28_montmul(sq, vm, vm);
    sc28_montmul(isq, im, im);
    if (i < k) {
        rp_store28(fields, B, fl.u_m + i, p, vm);
        rp_store28(fields, B, fl.uinv_m + i, p, im);
        if (df) {
            df->slot[4 + i] = sq;           // u_i^2   -> L_i coefficient
            df->slot[4 + k + i] = isq;      // u_i^-2  -> R_i coefficient
        } else {
            uint32_t *us = recoded + (uint64_t)p * sh.U * 8;
            rp_emit_coeff(us, 4 + i, sq, nullptr, 0, p * sh.U, sh.radix5 != 0);
            rp_emit_coeff(us, 4 + k + i, isq, nullptr, 0, p * sh.U, sh.radix5 != 0);
        }
    } else {   // y: the table of y^-(2^b)
        rp_store28(fields, B, fl.yinvp_m + 0, p, im);
        sc28 pw = isq;
        for (uint32_t bb = 1; bb < k; bb++) {
            rp_store28(fields, B, fl.yinvp_m + bb, p, pw);
            sc28_montmul(pw, pw, pw);
        }
    }
}
// launch 3, lane = proof: the digits of the B_blinding (row 0) and B (row 1) coefficients from the fields launch 1 left
// (mod.rs:422-432's last two scalars; delta: mod.rs:587-593)
BP_HD void rp_rows_thread(uint32_t p, rp_shape sh, fb_params prm, uint32_t lg_m, const uint32_t *fields, fb_digit *digits, const uint32_t *status) {
    if (status[p] != 0) return;
    const uint32_t B = sh.nproofs, k = sh.k;
    sc y, z, zz, w, c, tx, txb, eb, t0, t1;
    rp_load(y, fields, B, RPF_Y, p);
    rp_load(z, fields, B, RPF_Z, p);
    rp_load(zz, fields, B, RPF_ZZ, p);
    rp_load(w, fields, B, RPF_W, p);
    rp_load(c, fields, B, RPF_C, p);
    rp_load(tx, fields, B, RPF_TX, p);
    rp_load(txb, fields, B, RPF_TXB, p);
    rp_load(eb, fields, B, RPF_EB, p);
    sc28 ym, zm, zzm, wm, cm, txbm, am, bm;
    sc_to_mont28(ym, y);
    sc_to_mont28(wm, w);
    sc_to_mont28(cm, c);
    sc_to_mont28(txbm, txb);
    rp_load28(zm, fields, B, RPF_Z_M, p);
    rp_load28(zzm, fields, B, RPF_ZZ_M, p);
    rp_load28(am, fields, B, RPF_A_M, p);
    rp_load28(bm, fields, B, RPF_B_M, p);
    {   // -e_blinding - c t_x_blinding
        sc28 pm;
        sc28_montmul(pm, cm, txbm);
        sc_from_mont28(t0, pm);
        sc_add(t0, t0, eb);
        sc_neg(t0, t0);
        fb_recode(digits + ((uint64_t)0 * prm.nwin) * B + p, B, t0.v, prm);
    }
    {   // w (t_x - a b) + c (delta(y,z) - t_x)
        sc28 abm, pm, sum_ym, sum_zm, dm;
        sc ab, dl, sum_2;
        sc28_montmul(abm, am, bm);
        sc_from_mont28(ab, abm);
        sc_sub(t0, tx, ab);
        sc_to_mont28(pm, t0);
        sc28_montmul(pm, wm, pm);
        sc_from_mont28(t0, pm);
        rp_sum_of_powers_pow2(sum_ym, ym, k);
        rp_sum_of_powers_pow2(sum_zm, zm, lg_m);
        sc_0(sum_2);
        if (sh.n >= 64) { sum_2.v[0] = 0xffffffffu; sum_2.v[1] = 0xffffffffu; }
        else if (sh.n >= 32) { sum_2.v[0] = 0xffffffffu; sum_2.v[1] = (sh.n == 32) ? 0u : ((1u << (sh.n - 32)) - 1u); }
        else sum_2.v[0] = (1u << sh.n) - 1u;
        sc_sub(t1, z, zz);
        sc_to_mont28(dm, t1);
        sc28_montmul(dm, dm, sum_ym);
        sc_from_mont28(dl, dm);
        sc28 z3m, s2m;
        sc28_montmul(z3m, zzm, zm);
        sc_to_mont28(s2m, sum_2);
        sc28_montmul(z3m, z3m, s2m);
        sc28_montmul(z3m, z3m, sum_zm);
        sc_from_mont28(t1, z3m);
        sc_sub(dl, dl, t1);
        sc_sub(t1, dl, tx);
        sc_to_mont28(pm, t1);
        sc28_montmul(pm, cm, pm);
        sc_from_mont28(t1, pm);
        sc_add(t0, t0, t1);
        fb_recode(digits + ((uint64_t)1 * prm.nwin) * B + p, B, t0.v, prm);
    }
}

// ---- stage 3: per-(generator, proof) scalars -------------------------------------------------
// 2^i * R mod l (Montgomery form of 2^i), i < 64: the factor 2^(i mod n) of h_i without a multiplication by R^2
BP_HD void rp_two_pow_mont(sc28 &r, uint32_t i) {
    const sc28 T[64] = {
    {{0xcf5d3edu, 0x4305db8u, 0x676a4b2u, 0x80113d7u, 0x0622aafu, 0xfffeb21u, 0xfffffffu, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x73a89cbu, 0x550730cu, 0x0643d7fu, 0x0c44080u, 0xfffd642u, 0xfffffffu, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0xd4ee1f1u, 0x3040fc0u, 0x12a90cfu, 0x1886c21u, 0xfffac84u, 0xfffffffu, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x977923du, 0xe6b4929u, 0x2b7376eu, 0x310c363u, 0xfff5908u, 0xfffffffu, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x1c8f2d5u, 0x539bbfbu, 0x5d084aeu, 0x62171e7u, 0xffeb210u, 0xfffffffu, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x26bb405u, 0x2d6a19eu, 0xc031f2du, 0xc42ceefu, 0xffd6420u, 0xfffffffu, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x3b13665u, 0xe106ce4u, 0x868542au, 0x8858900u, 0xffac841u, 0xfffffffu, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x63c3b25u, 0x4840370u, 0x132be26u, 0x10afd22u, 0xff59083u, 0xfffffffu, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0xb5244a5u, 0x16b3088u, 0x2c7921du, 0x215e565u, 0xfeb2106u, 0xfffffffu, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x57e57a5u, 0xb398ab9u, 0x5f13a0au, 0x42bb5ebu, 0xfd6420cu, 0xfffffffu, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x9d67da5u, 0xed63f1au, 0xc4489e5u, 0x85756f7u, 0xfac8418u, 0xfffffffu, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x286c9a5u, 0x60fa7ddu, 0x8eb299cu, 0x0ae9910u, 0xf590831u, 0xfffffffu, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x3e761a5u, 0x4827962u, 0x2386909u, 0x15d1d42u, 0xeb21062u, 0xfffffffu, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x6a891a5u, 0x1681c6cu, 0x4d2e7e3u, 0x2ba25a5u, 0xd6420c4u, 0xfffffffu, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0xc2af1a5u, 0xb336280u, 0xa07e596u, 0x574366bu, 0xac84188u, 0xfffffffu, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x72fb1a5u, 0xec9eea9u, 0x471e0fdu, 0xae857f8u, 0x5908310u, 0xfffffffu, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0xd3931a5u, 0x5f706fau, 0x945d7ccu, 0x5d09b11u, 0xb210621u, 0xffffffeu, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x94c31a5u, 0x451379du, 0x2edc569u, 0xba12144u, 0x6420c42u, 0xffffffdu, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x17231a5u, 0x10598e3u, 0x63da0a3u, 0x7422da9u, 0xc841885u, 0xffffffau, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x1be31a5u, 0xa6e5b6eu, 0xcdd5716u, 0xe844673u, 0x908310au, 0xffffff5u, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x25631a5u, 0xd3fe084u, 0xa1cc3fdu, 0xd087808u, 0x2106215u, 0xfffffebu, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x38631a5u, 0x2e2eab0u, 0x49b9dccu, 0xa10db32u, 0x420c42bu, 0xfffffd6u, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x5e631a5u, 0xe28ff08u, 0x9995168u, 0x421a185u, 0x8418857u, 0xfffffacu, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0xaa631a5u, 0x4b527b8u, 0x394b8a2u, 0x8432e2cu, 0x08310aeu, 0xfffff59u, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x42631a5u, 0x1cd7919u, 0x78b8715u, 0x0864779u, 0x106215du, 0xffffeb2u, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x72631a5u, 0xbfe1bdau, 0xf7923fau, 0x10c7a13u, 0x20c42bau, 0xffffd64u, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0xd2631a5u, 0x05f615cu, 0xf545dc6u, 0x218df48u, 0x4188574u, 0xffffac8u, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x92631a5u, 0x921ec61u, 0xf0ad15cu, 0x431a9b2u, 0x8310ae8u, 0xffff590u, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0xaa7026bu, 0xe77b889u, 0x8633e86u, 0x06215d0u, 0xfffeb21u, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0xdb12e7eu, 0xd5186e3u, 0x0c6682eu, 0x0c42ba1u, 0xfffd642u, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0x3c586a4u, 0xb052398u, 0x18cbb7eu, 0x1885742u, 0xfffac84u, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0xfee36f0u, 0x66c5d00u, 0x319621eu, 0x310ae84u, 0xfff5908u, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0x83f9788u, 0xd3acfd2u, 0x632af5du, 0x6215d08u, 0xffeb210u, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0x8e258b8u, 0xad7b575u, 0xc6549dcu, 0xc42ba10u, 0xffd6420u, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0xa27db18u, 0x61180bbu, 0x8ca7edau, 0x8857421u, 0xffac841u, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0xcb2dfd8u, 0xc851747u, 0x194e8d5u, 0x10ae843u, 0xff59083u, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0x1c8e958u, 0x96c4460u, 0x329bcccu, 0x215d086u, 0xfeb2106u, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0xbf4fc58u, 0x33a9e90u, 0x65364bau, 0x42ba10cu, 0xfd6420cu, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0x04d2258u, 0x6d752f2u, 0xca6b495u, 0x8574218u, 0xfac8418u, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0x8fd6e58u, 0xe10bbb4u, 0x94d544bu, 0x0ae8431u, 0xf590831u, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0xa5e0658u, 0xc838d39u, 0x29a93b8u, 0x15d0863u, 0xeb21062u, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0xd1f3658u, 0x9693043u, 0x5351292u, 0x2ba10c6u, 0xd6420c4u, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0x2a19658u, 0x3347658u, 0xa6a1046u, 0x574218cu, 0xac84188u, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0xda65658u, 0x6cb0280u, 0x4d40badu, 0xae84319u, 0x5908310u, 0xfffffffu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0x3afd658u, 0xdf81ad2u, 0x9a8027bu, 0x5d08632u, 0xb210621u, 0xffffffeu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0xfc2d658u, 0xc524b74u, 0x34ff018u, 0xba10c65u, 0x6420c42u, 0xffffffdu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0x7e8d658u, 0x906acbau, 0x69fcb52u, 0x74218cau, 0xc841885u, 0xffffffau, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0x834d658u, 0x26f6f45u, 0xd3f81c6u, 0xe843194u, 0x908310au, 0xffffff5u, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0x8ccd658u, 0x540f45bu, 0xa7eeeadu, 0xd086329u, 0x2106215u, 0xfffffebu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0x9fcd658u, 0xae3fe87u, 0x4fdc87bu, 0xa10c653u, 0x420c42bu, 0xfffffd6u, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0xc5cd658u, 0x62a12dfu, 0x9fb7c18u, 0x4218ca6u, 0x8418857u, 0xfffffacu, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0x11cd658u, 0xcb63b90u, 0x3f6e351u, 0x843194du, 0x08310aeu, 0xfffff59u, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0xa9cd658u, 0x9ce8cf0u, 0x7edb1c4u, 0x086329au, 0x106215du, 0xffffeb2u, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0xd9cd658u, 0x3ff2fb1u, 0xfdb4eaau, 0x10c6534u, 0x20c42bau, 0xffffd64u, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0x39cd658u, 0x8607534u, 0xfb68875u, 0x218ca69u, 0x4188574u, 0xffffac8u, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0xf9cd658u, 0x1230038u, 0xf6cfc0cu, 0x43194d3u, 0x8310ae8u, 0xffff590u, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0x79cd658u, 0x2a81642u, 0xed9e339u, 0x86329a7u, 0x06215d0u, 0xfffeb21u, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0x79cd658u, 0x5b24255u, 0xdb3b193u, 0x0c6534fu, 0x0c42ba1u, 0xfffd642u, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0x79cd658u, 0xbc69a7bu, 0xb674e47u, 0x18ca69fu, 0x1885742u, 0xfffac84u, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0x79cd658u, 0x7ef4ac7u, 0x6ce87b0u, 0x3194d3fu, 0x310ae84u, 0xfff5908u, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0x79cd658u, 0x040ab5fu, 0xd9cfa82u, 0x6329a7eu, 0x6215d08u, 0xffeb210u, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0x79cd658u, 0x0e36c8fu, 0xb39e025u, 0xc6534fdu, 0xc42ba10u, 0xffd6420u, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0x79cd658u, 0x228eeefu, 0x673ab6bu, 0x8ca69fbu, 0x8857421u, 0xffac841u, 0xfffffffu, 0x0000000u}},
    {{0xcf5d3edu, 0x12631a5u, 0x79cd658u, 0x4b3f3afu, 0xce741f7u, 0x194d3f6u, 0x10ae843u, 0xff59083u, 0xfffffffu, 0x0000000u}}};
    r = T[i & 63];
}
// r = a - b + 4l with normalised limbs (a, b lazy, < 2^254): stays in Montgomery form, costs ~40 instructions
// instead of two conversions out of it, a canonical subtraction and a conversion back
BP_HD void sc28_sub_lazy(sc28 &r, const sc28 &a, const sc28 &b) {
    const uint32_t L4[10] = {0x3d74fb4u, 0x498c697u, 0xe735960u, 0xe77a8bdu, 0x000537bu, 0u, 0u, 0u, 0u, 0x4u};   // 4l
    int32_t cy = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const int32_t t = (int32_t)a.v[i] - (int32_t)b.v[i] + (int32_t)L4[i] + cy;   // |t| < 2^30
        if (i < 9) {
            r.v[i] = (uint32_t)t & BP_M28;
            cy = t >> 28;
        } else {
            r.v[i] = (uint32_t)t;   // a - b + 4l > 0: the top limb is non-negative
        }
    }
}

// FOUR consecutive generator indices per thread: tid = t * nproofs + p, t < nm/4, i = 4t .. 4t+3.
// s_i = prod_b (bit_b(i) ? u : u^-1)[k-1-b]  (ipp.rs:241-250) and s_i^-1 = s_{nm-1-i}: the factors of bits >= 2
// are shared by the four indices, the four products over bits 0, 1 serve s (index j) and s^-1 (index 3-j);
// y^-i likewise.  g_i = -z - a s_i (mod.rs:415), h_i = z + y^-i (z^2 z^j 2^i' - b s_i^-1), j = i / n,
// i' = i % n (mod.rs:416-419).  Writes the digits of rows 2+i and 2+nm+i, or -- g_out / h_out given (batch-
// combination mode) -- returns the eight coefficients (zero for a rejected proof).
// Round 3 kept the four products P[j] = u_{k-1}^{+-1} u_{k-2}^{+-1} in a register array picked by the run-time index j: 176 bytes of
// scratch memory and 254 registers (two wavefronts per SIMD).  Now nothing is indexed at run time: the two factors of index j are
// re-read from the field-major scalar store by ADDRESS (u or u^-1, chosen by the bits of j), as are a, b, z, -z, y^-1, y^-2 where
// they are used -- three running products stay live across the loop, 167 registers (three wavefronts per SIMD), no scratch, no LDS
// (a 10 KB scratchpad per wavefront was measured first: same kernel time, but its LDS footprint kept the transcript wavefronts of
// the NEXT chain off the CUs: -12 % on 20 x 1024 bursts).  Price: one more Montgomery product per index (8 + instead of 7 +).
// j_lo .. j_hi: which of the four indices this lane handles (all four: the throughput form; one: rp_expand_b1_thread)
BP_HD void rp_expand_b4_thread(uint32_t tid, rp_shape sh, fb_params prm, const uint32_t *fields, fb_digit *digits,
                               const uint32_t *status, sc *g_out = nullptr, sc *h_out = nullptr, uint32_t j_lo = 0, uint32_t j_hi = 4) {
    const uint32_t B = sh.nproofs, k = sh.k;
    const uint32_t t4 = tid / B, p = tid - t4 * B, i0 = 4 * t4;
    if (g_out) {
        for (int j = 0; j < 4; j++) {
            sc_0(g_out[j]);
            sc_0(h_out[j]);
        }
    }
    if (status[p] != 0) return;
    const fb_bias bias = fb_make_bias(prm);   // wavefront-uniform: once for the eight recodings below
    const rp_fields fl = rp_field_layout(k, sh.m);
    sc28 s_hi, sinv_hi, y_hi, t;
    {
        sc28 um, uim;
        sc28_one_mont(s_hi);
        sinv_hi = s_hi;
        y_hi = s_hi;
        for (uint32_t bb = 2; bb < k; bb++) {
            rp_load28(um, fields, B, fl.u_m + (k - 1 - bb), p);
            rp_load28(uim, fields, B, fl.uinv_m + (k - 1 - bb), p);
            const bool bit = (i0 >> bb) & 1;
            sc28 f1, f2;
#pragma unroll
            for (int q = 0; q < 10; q++) {
                f1.v[q] = bit ? um.v[q] : uim.v[q];
                f2.v[q] = bit ? uim.v[q] : um.v[q];
            }
            sc28_montmul(s_hi, s_hi, f1);
            sc28_montmul(sinv_hi, sinv_hi, f2);
            if (bit) {
                rp_load28(t, fields, B, fl.yinvp_m + bb, p);
                sc28_montmul(y_hi, y_hi, t);
            }
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (uint32_t j = j_lo; j < j_hi; j++) {
        const uint32_t i = i0 + j;
        // bits 0 (challenge k-1) and 1 (challenge k-2) of the index: s_i takes u or u^-1 by the bit, s_i^-1 the other one
        const uint32_t fa = (j & 1) ? fl.u_m + (k - 1) : fl.uinv_m + (k - 1), fa_c = (j & 1) ? fl.uinv_m + (k - 1) : fl.u_m + (k - 1);
        const uint32_t fb = (j & 2) ? fl.u_m + (k - 2) : fl.uinv_m + (k - 2), fb_c = (j & 2) ? fl.uinv_m + (k - 2) : fl.u_m + (k - 2);
        sc28 s, yp, r;
        sc g, h, v;
        // g_i = -z - a s_i
        {
            sc28 a_m;
            rp_load28(t, fields, B, fa, p);
            sc28_montmul(s, s_hi, t);
            rp_load28(t, fields, B, fb, p);
            sc28_montmul(s, s, t);
            rp_load28(a_m, fields, B, RPF_A_M, p);
            sc28_montmul(t, a_m, s);
            sc_from_mont28(v, t);
            sc minus_z;
            rp_load(minus_z, fields, B, RPF_MINUS_Z, p);
            sc_sub(g, minus_z, v);
            // (recoded at once: g is not carried through the computation of h)
            if (!g_out) fb_recode(digits + ((uint64_t)(2 + i) * prm.nwin) * B + p, B, g.v, prm, bias);
        }
        // y^-i: the high bits' product times y^-1 (bit 0 of j) and y^-2 (bit 1)
        yp = y_hi;
        if (j & 1) {
            rp_load28(t, fields, B, fl.yinvp_m + 0, p);
            sc28_montmul(yp, yp, t);
        }
        if (j & 2) {
            rp_load28(t, fields, B, fl.yinvp_m + 1, p);
            sc28_montmul(yp, yp, t);
        }
        // h_i = z + y^-i (z^2 z^j 2^i' - b s_i^-1)
        {
            sc28 sinv, b_m, two_m, zzzj;
            rp_load28(t, fields, B, fa_c, p);
            sc28_montmul(sinv, sinv_hi, t);
            rp_load28(t, fields, B, fb_c, p);
            sc28_montmul(sinv, sinv, t);
            const uint32_t jj = i / sh.n, ib = i - jj * sh.n;
            rp_load28(zzzj, fields, B, fl.zzzj_m + jj, p);
            rp_two_pow_mont(two_m, ib);
            sc28_montmul(r, zzzj, two_m);    // z^2 z^j 2^i'
            rp_load28(b_m, fields, B, RPF_B_M, p);
            sc28_montmul(t, b_m, sinv);      // b / s_i
            sc28_sub_lazy(r, r, t);
            sc28_montmul(r, r, yp);
            sc_from_mont28(v, r);
            sc z;
            rp_load(z, fields, B, RPF_Z, p);
            sc_add(h, z, v);
        }
        if (g_out) {
            g_out[j] = g;
            h_out[j] = h;
        } else {
            fb_recode(digits + ((uint64_t)(2 + sh.nm + i) * prm.nwin) * B + p, B, h.v, prm, bias);
        }
    }
}
// ONE generator index per thread (narrow chains: launch 3 is latency, not work): tid = i * nproofs + p, i < nm.  The lane forms the shared
// products of bits >= 2 itself (k - 2 rounds) and then the one index: ~21 Montgomery products in sequence instead of ~55 (four indices)
// or ~80 (the mirrored pairs); 1.6 x the role's instructions, which a chain of <= 256 proofs does not notice.  Same code, same digits.
BP_HD void rp_expand_b1_thread(uint32_t tid, rp_shape sh, fb_params prm, const uint32_t *fields, fb_digit *digits, const uint32_t *status) {
    const uint32_t B = sh.nproofs, i = tid / B, p = tid - i * B;
    rp_expand_b4_thread((i >> 2) * B + p, sh, prm, fields, digits, status, nullptr, nullptr, i & 3, (i & 3) + 1);
}

// EIGHT generator indices per thread, in mirrored pairs (round 6): tid = t * nproofs + p, t < nm/8; the four indices i = 4t .. 4t+3 of the
// lower half and their mirror images i' = nm - 1 - i.  nm is a power of two, so the bits of i' are the complements of the bits of i:
//     s_i' = s_i^-1,   s_i'^-1 = s_i      (ipp.rs:241-250: s_i takes u or u^-1 by the bit)
// -- the two products a lane forms for index i (s_i for g_i, s_i^-1 for h_i) are exactly the two that index i' needs, in the other
// roles: g_i' = -z - a s_i^-1, h_i' = z + y^-i' (z^2 z^j' 2^i'' - b s_i).  A pair of indices costs 2 (k - 2) / 4 + 4 products for s and
// s^-1 once instead of twice: 25.5 Montgomery products per pair instead of 34.5 at k = 12 (m = 32) -- -24 % of the role.  y^-i' is
// the product over the bits that i does NOT have: the high-bit loop multiplies one of two running products per bit (as many products as
// before per pair of lanes).  j' = m - 1 - j, i'' = n - 1 - (i mod n).
BP_HD void rp_expand_b8_thread(uint32_t tid, rp_shape sh, fb_params prm, const uint32_t *fields, fb_digit *digits, const uint32_t *status) {
    const uint32_t B = sh.nproofs, k = sh.k;
    const uint32_t t8 = tid / B, p = tid - t8 * B, i0 = 4 * t8;
    if (status[p] != 0) return;
    const fb_bias bias = fb_make_bias(prm);
    const rp_fields fl = rp_field_layout(k, sh.m);
    sc28 s_hi, sinv_hi, y_hi, yc_hi, t;   // yc_hi: y^-(the high bits i0 does not have)
    {
        sc28 um, uim;
        sc28_one_mont(s_hi);
        sinv_hi = s_hi;
        y_hi = s_hi;
        yc_hi = s_hi;
        for (uint32_t bb = 2; bb < k; bb++) {
            rp_load28(um, fields, B, fl.u_m + (k - 1 - bb), p);
            rp_load28(uim, fields, B, fl.uinv_m + (k - 1 - bb), p);
            const bool bit = (i0 >> bb) & 1;
            sc28 f1, f2;
#pragma unroll
            for (int q = 0; q < 10; q++) {
                f1.v[q] = bit ? um.v[q] : uim.v[q];
                f2.v[q] = bit ? uim.v[q] : um.v[q];
            }
            sc28_montmul(s_hi, s_hi, f1);
            sc28_montmul(sinv_hi, sinv_hi, f2);
            rp_load28(t, fields, B, fl.yinvp_m + bb, p);
            if (bit) sc28_montmul(y_hi, y_hi, t);
            else sc28_montmul(yc_hi, yc_hi, t);
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (uint32_t j = 0; j < 4; j++) {
        const uint32_t i = i0 + j, im = sh.nm - 1 - i;
        const uint32_t fa = (j & 1) ? fl.u_m + (k - 1) : fl.uinv_m + (k - 1), fa_c = (j & 1) ? fl.uinv_m + (k - 1) : fl.u_m + (k - 1);
        const uint32_t fb = (j & 2) ? fl.u_m + (k - 2) : fl.uinv_m + (k - 2), fb_c = (j & 2) ? fl.uinv_m + (k - 2) : fl.u_m + (k - 2);
        sc28 s, sinv;
        rp_load28(t, fields, B, fa, p);
        sc28_montmul(s, s_hi, t);
        rp_load28(t, fields, B, fb, p);
        sc28_montmul(s, s, t);
        rp_load28(t, fields, B, fa_c, p);
        sc28_montmul(sinv, sinv_hi, t);
        rp_load28(t, fields, B, fb_c, p);
        sc28_montmul(sinv, sinv, t);
        // the pair: (index, its s, its s^-1, y^-index from the running product `yh` and the low bits `lo`)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
        for (uint32_t side = 0; side < 2; side++) {
            const uint32_t idx = side ? im : i, lo = side ? (3u - j) : j;
            sc28 sg, sh_, yp, r;
#pragma unroll
            for (int q = 0; q < 10; q++) {
                sg.v[q] = side ? sinv.v[q] : s.v[q];      // s_idx
                sh_.v[q] = side ? s.v[q] : sinv.v[q];     // s_idx^-1
                yp.v[q] = side ? yc_hi.v[q] : y_hi.v[q];
            }
            sc g, h, v;
            {   // g_idx = -z - a s_idx
                sc28 a_m;
                rp_load28(a_m, fields, B, RPF_A_M, p);
                sc28_montmul(t, a_m, sg);
                sc_from_mont28(v, t);
                sc minus_z;
                rp_load(minus_z, fields, B, RPF_MINUS_Z, p);
                sc_sub(g, minus_z, v);
                fb_recode(digits + ((uint64_t)(2 + idx) * prm.nwin) * B + p, B, g.v, prm, bias);
            }
            if (lo & 1) {
                rp_load28(t, fields, B, fl.yinvp_m + 0, p);
                sc28_montmul(yp, yp, t);
            }
            if (lo & 2) {
                rp_load28(t, fields, B, fl.yinvp_m + 1, p);
                sc28_montmul(yp, yp, t);
            }
            {   // h_idx = z + y^-idx (z^2 z^j 2^i' - b s_idx^-1)
                sc28 b_m, two_m, zzzj;
                const uint32_t jj = idx / sh.n, ib = idx - jj * sh.n;
                rp_load28(zzzj, fields, B, fl.zzzj_m + jj, p);
                rp_two_pow_mont(two_m, ib);
                sc28_montmul(r, zzzj, two_m);
                rp_load28(b_m, fields, B, RPF_B_M, p);
                sc28_montmul(t, b_m, sh_);
                sc28_sub_lazy(r, r, t);
                sc28_montmul(r, r, yp);
                sc_from_mont28(v, r);
                sc z;
                rp_load(z, fields, B, RPF_Z, p);
                sc_add(h, z, v);
                fb_recode(digits + ((uint64_t)(2 + sh.nm + idx) * prm.nwin) * B + p, B, h.v, prm, bias);
            }
        }
    }
}

}  // namespace bp
#endif
