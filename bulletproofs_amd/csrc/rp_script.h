// The verifier's Fiat-Shamir transcript (src/range_proof/mod.rs:368-393, src/inner_product_proof.rs:213-222 over
// merlin's STROBE-128 framing, src/transcript.rs:43-95) compiled into a per-shape SCRIPT.
//
// For a given shape (n, m, k) and start position, every byte position of the replay is the same for every proof: which
// rate byte each label / length / flag byte is XORed into, where each 32-byte message lands, when the permutation runs.
// Only the message bytes and the squeezed challenges depend on the proof.  So the host runs the STROBE position machine ONCE
// per shape (rp_script_build) and emits
//   * one 42-word XOR mask per permutation: all framing bytes of the span before it (labels, little-endian lengths, the
//     begin-op bytes, STROBE's own padding at run_f) -- applied by the device while it loads the state for the permutation;
//   * an operation list: MSG (a 32-byte record of the proof / the commitments at a rate position), MSGB (the part of such a
//     record before / after a rate boundary), PERM (mask index), CHAL (squeeze the 64 challenge bytes = state words 0..15,
//     which STROBE then zeroes).
// The device interpreter (rp_transcript_scripted, rangeproof.h) does word-wide XORs at wavefront-uniform positions instead
// of ~20 byte-wide read-modify-writes of the sponge state per framed message: launch 1's transcript role loses the ~45 us
// (of 280) it spent on framing.  Results are bit-identical to the byte-wise replay (rp_transcript_thread), which stays for
// caller-supplied per-proof transcripts (their positions may differ from proof to proof) and as the reference of the CPU tests.
#ifndef BPGPU_RP_SCRIPT_H
#define BPGPU_RP_SCRIPT_H
#include "keccak.h"

namespace bp {

enum { RS_MSG = 0, RS_MSGB = 1, RS_PERM = 2, RS_CHAL = 3 };
enum { RS_SRC_PROOF = 0, RS_SRC_COMMITMENTS = 1 };
#define RS_MASK_WORDS 42   // rate (166) + the two STROBE padding bytes = 168 bytes
struct rp_script_op {
    uint16_t kind, src;    // RS_*, RS_SRC_*
    uint16_t pos, nbytes;  // rate position of the first byte; bytes (MSGB only)
    uint32_t off;          // byte offset of the first byte inside the proof / the proof's commitments
    uint32_t arg;          // PERM: mask index; CHAL: 0 y, 1 z, 2 x, 3 w, 4 + i: u_i
};
struct rp_script_hdr {
    uint32_t n_ops, n_masks;
    uint32_t end_pos, end_pos_begin, end_flags;   // STROBE bookkeeping after the last challenge (for transcripts handed back)
    uint32_t n_stops;                             // 4 + 2k: one record per validated point, in transcript order A, S, T_1, T_2, L_0, R_0, L_1, ...
    uint32_t pad[2];
};
// Where the replay STOPS when a validated point is the identity encoding: validate_and_append_point returns Err before it absorbs
// anything of that message (src/transcript.rs:75-87; call sites mod.rs:376-393, ipp.rs:217-222), so the caller's `&mut Transcript`
// stays as it was just before the message's framing.  Per validated point: the index of the first operation that belongs to the
// message (the state handed back is the one BEFORE that operation), the framing bytes of the current span that the interpreter has
// not applied yet (they normally arrive with the span's permutation mask), and STROBE's bookkeeping at that moment.
struct rp_script_stop {
    uint32_t op;      // stop before ops[op]
    uint32_t meta;    // pos | pos_begin << 8 | cur_flags << 16 (the layout of word 50 of a transcript state)
    uint32_t mask[RS_MASK_WORDS];
};
// device layout: hdr | ops[n_ops] | masks[n_masks][RS_MASK_WORDS] | stops[n_stops]
BP_HD const rp_script_op *rp_script_ops(const rp_script_hdr *h) { return (const rp_script_op *)(h + 1); }
BP_HD const uint32_t *rp_script_masks(const rp_script_hdr *h) { return (const uint32_t *)(rp_script_ops(h) + h->n_ops); }
BP_HD const rp_script_stop *rp_script_stops(const rp_script_hdr *h) { return (const rp_script_stop *)(rp_script_masks(h) + (uint64_t)h->n_masks * RS_MASK_WORDS); }

}  // namespace bp
#include <cstring>
#include <vector>
namespace bp {
// ---- host: the compiler ----------------------------------------------------------------------------------------------
struct rp_script_builder {
    std::vector<rp_script_op> ops;
    std::vector<uint32_t> masks;
    std::vector<rp_script_stop> stops;
    uint8_t cur[RS_MASK_WORDS * 4];
    uint32_t pos, pos_begin, cur_flags;
    void init(uint32_t p, uint32_t pb, uint32_t fl) {
        ops.clear();
        masks.clear();
        stops.clear();
        for (auto &b : cur) b = 0;
        pos = p;
        pos_begin = pb;
        cur_flags = fl;
    }
    void mark_stop() {   // the next message is a validated point
        rp_script_stop st{};
        st.op = (uint32_t)ops.size();
        st.meta = (pos & 0xffu) | ((pos_begin & 0xffu) << 8) | ((cur_flags & 0xffu) << 16);
        for (int w = 0; w < RS_MASK_WORDS; w++)
            st.mask[w] = (uint32_t)cur[4 * w] | ((uint32_t)cur[4 * w + 1] << 8) | ((uint32_t)cur[4 * w + 2] << 16) | ((uint32_t)cur[4 * w + 3] << 24);
        stops.push_back(st);
    }
    // validate_and_append_point(label, a 32-byte record of the proof)
    void append_point(const char *label, uint32_t label_len, uint32_t off) {
        mark_stop();
        append_record(label, label_len, RS_SRC_PROOF, off);
    }
    void run_f() {   // strobe_run_f (keccak.h)
        cur[pos] ^= (uint8_t)pos_begin;
        cur[pos + 1] ^= 0x04;
        cur[BP_STROBE_R + 1] ^= 0x80;
        rp_script_op o{};
        o.kind = RS_PERM;
        o.arg = (uint32_t)(masks.size() / RS_MASK_WORDS);
        for (int w = 0; w < RS_MASK_WORDS; w++)
            masks.push_back((uint32_t)cur[4 * w] | ((uint32_t)cur[4 * w + 1] << 8) | ((uint32_t)cur[4 * w + 2] << 16) | ((uint32_t)cur[4 * w + 3] << 24));
        ops.push_back(o);
        for (auto &b : cur) b = 0;
        pos = 0;
        pos_begin = 0;
    }
    void absorb_const(uint32_t b) {
        cur[pos++] ^= (uint8_t)b;
        if (pos == BP_STROBE_R) run_f();
    }
    void begin_op(uint32_t flags, bool more) {   // strobe_begin_op
        if (more) return;
        const uint32_t old_begin = pos_begin;
        pos_begin = pos + 1;
        cur_flags = flags;
        absorb_const(old_begin);
        absorb_const(flags);
        if ((flags & (BP_FLAG_C | BP_FLAG_K)) && pos != 0) run_f();
    }
    void meta_ad(const char *label, uint32_t n) {
        begin_op(BP_FLAG_M | BP_FLAG_A, false);
        for (uint32_t i = 0; i < n; i++) absorb_const((uint8_t)label[i]);
    }
    void meta_len(uint32_t n) {
        for (int i = 0; i < 4; i++) absorb_const((n >> (8 * i)) & 0xff);
    }
    // Transcript::append_message(label, constant bytes)
    void append_const(const char *label, uint32_t label_len, const uint8_t *msg, uint32_t n) {
        meta_ad(label, label_len);
        meta_len(n);
        begin_op(BP_FLAG_A, false);
        for (uint32_t i = 0; i < n; i++) absorb_const(msg[i]);
    }
    void append_u64(const char *label, uint32_t label_len, uint64_t x) {
        uint8_t b[8];
        for (int i = 0; i < 8; i++) b[i] = (uint8_t)(x >> (8 * i));
        append_const(label, label_len, b, 8);
    }
    // Transcript::append_message(label, a 32-byte record of the proof / the commitments)
    void append_record(const char *label, uint32_t label_len, uint32_t src, uint32_t off) {
        meta_ad(label, label_len);
        meta_len(32);
        begin_op(BP_FLAG_A, false);
        uint32_t done = 0;
        while (done < 32) {
            const uint32_t room = BP_STROBE_R - pos, take = (32 - done) < room ? (32 - done) : room;
            rp_script_op o{};
            o.kind = (uint16_t)((take == 32) ? RS_MSG : RS_MSGB);
            o.src = (uint16_t)src;
            o.pos = (uint16_t)pos;
            o.nbytes = (uint16_t)take;
            o.off = off + done;
            ops.push_back(o);
            pos += take;
            done += take;
            if (pos == BP_STROBE_R) run_f();
        }
    }
    // Transcript::challenge_bytes(label, 64)
    void challenge(const char *label, uint32_t label_len, uint32_t id) {
        meta_ad(label, label_len);
        meta_len(64);
        begin_op(BP_FLAG_I | BP_FLAG_A | BP_FLAG_C, false);   // leaves pos == 0
        rp_script_op o{};
        o.kind = RS_CHAL;
        o.arg = id;
        ops.push_back(o);
        pos = 64;   // 64 squeezed bytes; 64 < R: no permutation inside
    }
};

// The whole verifier transcript of an (n, m) range proof with k = lg(n m) inner-product rounds, starting at STROBE position
// (pos, pos_begin, cur_flags); domsep: the start state does not contain rangeproof_domain_sep(n, m) yet (transcript.rs:44-48).
// Returns the device image (hdr | ops | masks) as 32-bit words.
inline std::vector<uint32_t> rp_script_build(uint32_t n, uint32_t m, uint32_t k, uint32_t pos, uint32_t pos_begin, uint32_t cur_flags, bool domsep) {
    rp_script_builder b;
    b.init(pos, pos_begin, cur_flags);
    if (domsep) {
        b.append_const("dom-sep", 7, (const uint8_t *)"rangeproof v1", 13);
        b.append_u64("n", 1, n);
        b.append_u64("m", 1, m);
    }
    for (uint32_t j = 0; j < m; j++) b.append_record("V", 1, RS_SRC_COMMITMENTS, 32 * j);   // mod.rs:370-374
    b.append_point("A", 1, 0);
    b.append_point("S", 1, 32);
    b.challenge("y", 1, 0);
    b.challenge("z", 1, 1);
    b.append_point("T_1", 3, 64);
    b.append_point("T_2", 3, 96);
    b.challenge("x", 1, 2);
    b.append_record("t_x", 3, RS_SRC_PROOF, 128);
    b.append_record("t_x_blinding", 12, RS_SRC_PROOF, 160);
    b.append_record("e_blinding", 10, RS_SRC_PROOF, 192);
    b.challenge("w", 1, 3);
    b.append_const("dom-sep", 7, (const uint8_t *)"ipp v1", 6);                              // ipp.rs:213
    b.append_u64("n", 1, (uint64_t)n * m);
    for (uint32_t i = 0; i < k; i++) {
        b.append_point("L", 1, 224 + 64 * i);
        b.append_point("R", 1, 224 + 64 * i + 32);
        b.challenge("u", 1, 4 + i);
    }
    // bytes still pending in `cur` would be framing absorbed after the last permutation: there are none (the replay ends with a squeeze)
    rp_script_hdr h{};
    h.n_ops = (uint32_t)b.ops.size();
    h.n_masks = (uint32_t)(b.masks.size() / RS_MASK_WORDS);
    h.end_pos = b.pos;
    h.end_pos_begin = b.pos_begin;
    h.end_flags = b.cur_flags;
    h.n_stops = (uint32_t)b.stops.size();
    const size_t o_ops = sizeof(rp_script_hdr) / 4, o_masks = o_ops + b.ops.size() * (sizeof(rp_script_op) / 4), o_stops = o_masks + b.masks.size();
    std::vector<uint32_t> img(o_stops + b.stops.size() * (sizeof(rp_script_stop) / 4));
    memcpy(img.data(), &h, sizeof h);
    if (!b.ops.empty()) memcpy(img.data() + o_ops, b.ops.data(), b.ops.size() * sizeof(rp_script_op));
    if (!b.masks.empty()) memcpy(img.data() + o_masks, b.masks.data(), b.masks.size() * 4);
    if (!b.stops.empty()) memcpy(img.data() + o_stops, b.stops.data(), b.stops.size() * sizeof(rp_script_stop));
    return img;
}

}  // namespace bp
#endif
