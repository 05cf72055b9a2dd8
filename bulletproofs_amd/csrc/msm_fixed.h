// Fixed-base part of the verification MSM: the 2nm+2 generator terms
// (B_blinding, B, G[0..nm), H[0..nm) -- src/range_proof/mod.rs:439-442 and
// src/generators.rs:207-259 of the reference) are the same points for every
// proof of a batch and for every batch, so they get precomputed window tables
//     T[g][win][k] = (k+1) * 2^(W*win) * P_g ,  k < 2^(W-1),  affine Niels form
// that live in HBM.  With them a generator term costs ceil(255/W) mixed additions
// and no doublings at all (the reference's Straus/Pippenger spend ~256
// doublings + 256/5..256/8 additions per term on them).
//
// Lane mapping of the accumulation kernel: lane = proof.  All 64 lanes of a
// wavefront walk the same (generator, window) sequence, so the scalar digits
// are read coalesced from a [generator][window][proof] array and all table
// gathers of one step fall into one 2^(W-1)*128-byte sub-table (16 KiB at
// W = 8) that stays in L1/L2.  No cross-lane reduction is needed; the pair
// range is split across SPLIT workgroups to fill the chip and the SPLIT
// partial points per proof are added in the finishing kernel.
#ifndef BPGPU_MSM_FIXED_H
#define BPGPU_MSM_FIXED_H
#include "msm_vb.h"

namespace bp {

// one table entry: affine Niels point, 10-limb form, padded to a 128-byte line
struct __attribute__((aligned(16))) fb_entry {
    fe ypx, ymx, t2d;   // 120 bytes
    uint32_t pad[2];
};

struct fb_params {
    uint32_t W;        // window bits (1..16)
    uint32_t nwin;     // fb_nwin(W) = ceil(255 / W)
    uint32_t half;     // 2^(W-1) = entries per (generator, window)
    uint32_t n_gens;   // generators in the table
};

// one recoded window: the unsigned W-bit value digit + half (W <= 20)
typedef uint32_t fb_digit;
// Windows needed so that the recoded value s + sum_win half*2^(W*win) (s < 2^253, W >= 2) stays below
// 2^(W*nwin): the added constant is < 2^(W*nwin-1) (1 + 1/(2^W-1)), so W*nwin >= 255 suffices
// (2^253 <= 2^254 (2^W-2)/(2^W-1) for W >= 2).  W = 17 is the first window that saves one (15 instead of 16).
BP_HD uint32_t fb_nwin(uint32_t W) { return (255 + W - 1) / W; }

// ---- table construction ------------------------------------------------------
// thread g < n_gens : decode generator, emit base[g][win] = 2^(W*win) * P_g (extended)
BP_HD void fb_base_thread(uint32_t g, fb_params prm, const uint32_t *gens_compressed, ge_ext *base, uint32_t *bad) {
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = gens_compressed[8 * (uint64_t)g + i];
    ge_ext p;
    if (!ristretto_decompress(p, w)) *bad = 1;
    for (uint32_t win = 0; win < prm.nwin; win++) {
        base[(uint64_t)g * prm.nwin + win] = p;
        for (uint32_t k = 0; k < prm.W; k++) ge_dbl(p, p, k + 1 == prm.W);
    }
}

// Table construction runs in two passes so that the field inversions of the projective -> affine
// normalisation are shared (Montgomery's trick, 8 entries per inversion):
//   fb_fill : thread = g * nwin + win   writes the 2^(W-1) multiples of base[g][win] as raw (X, Y, Z)
//   fb_norm : thread = group of 8 consecutive entries   (X,Y,Z) -> (y+x, y-x, 2d*x*y)
#define BP_FB_NORM_GROUP 8

BP_HD void fb_fill_thread(uint32_t tid, fb_params prm, const ge_ext *base, fb_entry *table) {
    const ge_ext p = base[tid];
    ge_cached pc;
    ge_to_cached(pc, p);
    ge_ext cur = p;
    fb_entry *out = table + (uint64_t)tid * prm.half;
    for (uint32_t k = 0; k < prm.half; k++) {
        fb_entry e;
        e.ypx = cur.X;   // raw projective coordinates, normalised by fb_norm
        e.ymx = cur.Y;
        e.t2d = cur.Z;
        e.pad[0] = 0;
        e.pad[1] = 0;
        out[k] = e;
        ge_add_cached(cur, cur, pc, false);
    }
}

BP_HD void fb_norm_thread(uint64_t gid, uint64_t n_entries, fb_entry *table) {
    const fe d2 = BP_FE_D2;
    fb_entry *e = table + gid * BP_FB_NORM_GROUP;
    const uint32_t cnt = (gid * BP_FB_NORM_GROUP + BP_FB_NORM_GROUP <= n_entries) ? BP_FB_NORM_GROUP
                                                                                   : (uint32_t)(n_entries - gid * BP_FB_NORM_GROUP);
    fe pre[BP_FB_NORM_GROUP];   // pre[i] = Z_0 * ... * Z_i
    pre[0] = e[0].t2d;
#pragma unroll
    for (uint32_t i = 1; i < BP_FB_NORM_GROUP; i++) {
        if (i < cnt) fe_mul(pre[i], pre[i - 1], e[i].t2d);
        else pre[i] = pre[i - 1];
    }
    fe inv;
    fe_invert(inv, pre[BP_FB_NORM_GROUP - 1]);
#pragma unroll
    for (uint32_t ii = BP_FB_NORM_GROUP; ii-- > 0;) {
        if (ii >= cnt) continue;
        fe zinv, x, y, xy;
        const fe Z = e[ii].t2d;
        if (ii > 0) {
            fe_mul(zinv, inv, pre[ii - 1]);
            fe_mul(inv, inv, Z);
        } else {
            zinv = inv;
        }
        fe_mul(x, e[ii].ypx, zinv);
        fe_mul(y, e[ii].ymx, zinv);
        fe_mul(xy, x, y);
        fb_entry o;
        fe_add(o.ypx, y, x);
        fe_carry(o.ypx);
        fe_sub(o.ymx, y, x);
        fe_mul(o.t2d, xy, d2);
        o.pad[0] = 0;
        o.pad[1] = 0;
        e[ii] = o;
    }
}

// ---- scalar recoding -----------------------------------------------------------
// Signed fixed-window recoding of a canonical scalar: add half at every window
// position, then window value v in [0, 2^W) encodes digit d = v - half.
// The added constant K = sum_win half << (W win) depends on W only: formed once per thread from wavefront-uniform values (static
// indices, scalar registers) and reused by every recoding of the thread.
struct fb_bias {
    uint32_t k[10];
};
BP_HD fb_bias fb_make_bias(fb_params prm) {
    fb_bias b;
#pragma unroll
    for (int i = 0; i < 10; i++) b.k[i] = 0;
    for (uint32_t win = 0; win < prm.nwin; win++) {
        const uint32_t bit = win * prm.W + (prm.W - 1), word = bit >> 5, m = 1u << (bit & 31);
#pragma unroll
        for (int i = 0; i < 10; i++) b.k[i] |= (word == (uint32_t)i) ? m : 0u;
    }
    return b;
}
// No array is indexed by a run-time value (such arrays live in scratch memory: the first version's r[idx] read-modify-writes made the
// generator-exponent launch run at 0.44 of the device's instruction rate): one 288-bit addition, then per window "low W bits, shift right by W".
BP_HD void fb_recode(fb_digit *digits /*stride*/, uint64_t stride, const uint32_t s[8], fb_params prm, const fb_bias &bias) {
    uint32_t r[10], carry = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const uint64_t t = (uint64_t)(i < 8 ? s[i] : 0u) + bias.k[i] + carry;
        r[i] = (uint32_t)t;
        carry = (uint32_t)(t >> 32);
    }
    const uint32_t W = prm.W, mask = (1u << W) - 1u;   // 2 <= W <= 20
    for (uint32_t win = 0; win < prm.nwin; win++) {
        digits[(uint64_t)win * stride] = (fb_digit)(r[0] & mask);
#pragma unroll
        for (int i = 0; i < 9; i++) r[i] = (r[i] >> W) | (r[i + 1] << (32 - W));
        r[9] >>= W;
    }
}
BP_HD void fb_recode(fb_digit *digits /*stride*/, uint64_t stride, const uint32_t s[8], fb_params prm) {
    fb_recode(digits, stride, s, prm, fb_make_bias(prm));
}

// thread = g_local * nproofs + p : recode scalar of generator term g_local of proof p into
// digits[(g_local*nwin + win) * nproofs + p]
BP_HD void fb_recode_thread(uint32_t tid, fb_params prm, uint32_t nproofs, uint32_t n_gen_terms,
                            const uint32_t *gen_scalars /*[p][g_local][8]*/, fb_digit *digits, uint32_t *status) {
    const uint32_t g = tid / nproofs, p = tid % nproofs;
    uint32_t s[8];
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = gen_scalars[((uint64_t)p * n_gen_terms + g) * 8 + i];
    if (!sc_is_canonical(s)) status_raise(status + p, BP_STATUS_BAD_SCALAR);
    fb_recode(digits + ((uint64_t)g * prm.nwin) * nproofs + p, nproofs, s, prm);
}

// ---- accumulation -----------------------------------------------------------------
// thread (split, p): partial[split*nproofs + p] = sum over pairs q in [q0, q1) of d(q,p) * 2^(W*win) * P_{gen_ids[g]}
// Software-pipelined: the table line of pair q+1 and the digit of pair q+2 are in flight while the
// seven field multiplications of pair q execute, so the random 128-byte gathers never stall the lane.
struct fb_line {
    uint32_t w[30];   // ypx, ymx, t2d as 3 x 10 limbs (the payload of one fb_entry)
};
BP_HD void fb_load_line(fb_line &l, const fb_entry *e) {
    const uint32_t *src = (const uint32_t *)e;
#pragma unroll
    for (int i = 0; i < 30; i++) l.w[i] = src[i];
}
// the pair index as (generator term, window), advanced without a division per pair
struct fb_cursor {
    uint32_t g, win;
};
BP_HD const fb_entry *fb_entry_at(const fb_entry *table, const uint32_t *gen_ids, fb_params prm, fb_cursor &c, uint32_t v) {
    const int d = (int)(v & (2u * prm.half - 1u)) - (int)prm.half;   // masked: rows of rejected proofs are never written
    const uint32_t a = (uint32_t)(d < 0 ? -d : d);
    const fb_entry *e = table + ((uint64_t)gen_ids[c.g] * prm.nwin + c.win) * prm.half + (a ? a - 1 : 0);
    if (++c.win == prm.nwin) {
        c.win = 0;
        c.g++;
    }
    return e;
}
BP_HD void fb_accum_step(ge_ext &acc, const fb_line &line, uint32_t v, fb_params prm, bool first) {
    const int d = (int)(v & (2u * prm.half - 1u)) - (int)prm.half;
    if (d != 0) {
        ge_niels n;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            n.ypx.v[i] = line.w[i];
            n.ymx.v[i] = line.w[10 + i];
            n.t2d.v[i] = line.w[20 + i];
        }
        if (first) ge_from_niels(acc, n, d < 0);   // the accumulator is still the identity: 1 multiplication, not 7
        else ge_madd(acc, acc, n, d < 0);
    }
}
// Two pairs per trip, two line buffers used alternately: the line of the next pair is loaded straight into the buffer the
// previous pair has just vacated (no register copies between trips).
BP_HD void fb_accum_point(ge_ext &acc, uint32_t p, uint32_t q0, uint32_t q1, fb_params prm, uint32_t nproofs,
                          const uint32_t *gen_ids, const fb_digit *digits, const fb_entry *table) {
    ge_identity(acc);
    if (q0 < q1) {
        const fb_digit *dg = digits + p;
        fb_cursor cur;   // the next line to request
        cur.g = q0 / prm.nwin;
        cur.win = q0 - cur.g * prm.nwin;
        uint32_t va = dg[(uint64_t)q0 * nproofs];
        uint32_t vb = (q0 + 1 < q1) ? dg[(uint64_t)(q0 + 1) * nproofs] : prm.half;
        fb_line la, lb;
        fb_load_line(la, fb_entry_at(table, gen_ids, prm, cur, va));
        // first pair on its own (the accumulator is the identity), so that the loop below is uniform
        if (q0 + 1 < q1) fb_load_line(lb, fb_entry_at(table, gen_ids, prm, cur, vb));
        uint32_t vc = (q0 + 2 < q1) ? dg[(uint64_t)(q0 + 2) * nproofs] : prm.half;
        fb_accum_step(acc, la, va, prm, true);
        // invariant at the top of a trip: pair q's line is in lb (digit vb), the digit of pair q+1 is vc, no line beyond q requested
        uint32_t q = q0 + 1;
        for (; q + 1 < q1; q += 2) {
            fb_load_line(la, fb_entry_at(table, gen_ids, prm, cur, vc));                              // pair q+1
            const uint32_t vd = (q + 2 < q1) ? dg[(uint64_t)(q + 2) * nproofs] : prm.half;            // digit of pair q+2
            fb_accum_step(acc, lb, vb, prm, false);                                                   // pair q
            if (q + 2 < q1) fb_load_line(lb, fb_entry_at(table, gen_ids, prm, cur, vd));              // pair q+2
            const uint32_t ve = (q + 3 < q1) ? dg[(uint64_t)(q + 3) * nproofs] : prm.half;            // digit of pair q+3
            fb_accum_step(acc, la, vc, prm, false);                                                   // pair q+1
            vb = vd;
            vc = ve;
        }
        if (q < q1) fb_accum_step(acc, lb, vb, prm, false);   // odd tail: pair q1-1 is in lb
    }
}
BP_HD void fb_accum_thread(uint32_t p, uint32_t split, uint32_t q0, uint32_t q1, fb_params prm, uint32_t nproofs,
                           const uint32_t *gen_ids, const fb_digit *digits, const fb_entry *table, ge_ext *partial) {
    ge_ext acc;
    fb_accum_point(acc, p, q0, q1, prm, nproofs, gen_ids, digits, table);
    partial[(uint64_t)split * nproofs + p] = acc;
}

// ---- the walk with the recoding folded in (round 6: bpgpu_msm_batch_shared's generator half as ONE launch) ----------------------------
// fb_recode + fb_accum + three fb_reduce launches were five kernels for what is one pass over the scalars: the device runs only about four
// kernels at a time whatever their width (profiles/r06/timeline_cfg5_16_before.txt: 4.05 in flight on average with 32 streams busy), so a
// chain's throughput is set by the SUM of its kernels' durations, and the five cost 0.78 ms of a chain's 2.5.  Here a lane reads the
// scalar of a generator term itself, recodes it in registers (no digit array: 295 kB written and read per 4 098-term MSM before) and
// walks that generator's windows; a wavefront is one slice of the generator terms for 64 MSMs (lane = MSM, as in fb_accum_thread); the
// wavefronts of a workgroup add their sums through LDS, so that a workgroup leaves ONE partial sum per MSM.
//   lane (p, slice): generator terms [g0, g1) of MSM p;  gen_scalars [p][n_gen_terms][8 words]
// The table line of window w+1 is requested while window w is added (the scalar of the next generator term is not requested ahead: the
// eight registers it would hold through the walk cost the third wavefront per SIMD, whose work hides that load better).  Two line
// buffers used alternately (no copy between windows: -4.5 % instructions per addition) spill 58 registers under the 168-register cap and were
// measured slower, 109-110 against 115-117 k MSMs/s (profiles/r06/cfg5_walk_two_buffers_ab.txt) -- as in the bucket stage's loop.
BP_HD void fb_walk_thread(ge_ext &acc, uint32_t p, uint32_t g0, uint32_t g1, fb_params prm, uint32_t n_gen_terms, const uint32_t *gen_scalars,
                          const uint32_t *gen_ids, const fb_entry *table, uint32_t *status) {
    ge_identity(acc);
    if (g0 >= g1) return;
    const fb_bias bias = fb_make_bias(prm);
    const uint32_t W = prm.W, mask = (1u << W) - 1u;
    const uint32_t *sp = gen_scalars + ((uint64_t)p * n_gen_terms + g0) * 8;
    bool bad = false;
    for (uint32_t g = g0; g < g1; g++) {
        uint32_t r[10];
        {
            uint32_t s[8];
#pragma unroll
            for (int i = 0; i < 8; i++) s[i] = sp[8 * (uint64_t)(g - g0) + i];
            bad = bad || !sc_is_canonical(s);
            uint32_t carry = 0;
#pragma unroll
            for (int i = 0; i < 10; i++) {
                const uint64_t t = (uint64_t)(i < 8 ? s[i] : 0u) + bias.k[i] + carry;
                r[i] = (uint32_t)t;
                carry = (uint32_t)(t >> 32);
            }
        }
        const fb_entry *base = table + (uint64_t)gen_ids[g] * prm.nwin * prm.half;
        uint32_t v = r[0] & mask;
        fb_line line;
        {
            const int d = (int)v - (int)prm.half;
            const uint32_t a = (uint32_t)(d < 0 ? -d : d);
            fb_load_line(line, base + (a ? a - 1 : 0));
        }
        for (uint32_t win = 0; win < prm.nwin; win++) {
#pragma unroll
            for (int i = 0; i < 9; i++) r[i] = (r[i] >> W) | (r[i + 1] << (32 - W));
            r[9] >>= W;
            const uint32_t vn = r[0] & mask;
            fb_line next = line;
            if (win + 1 < prm.nwin) {
                const int d = (int)vn - (int)prm.half;
                const uint32_t a = (uint32_t)(d < 0 ? -d : d);
                fb_load_line(next, base + (uint64_t)(win + 1) * prm.half + (a ? a - 1 : 0));
            }
            fb_accum_step(acc, line, v, prm, false);
            line = next;
            v = vn;
        }
    }
    if (bad) status_raise(status + p, BP_STATUS_BAD_SCALAR);
}
// slice s of nslice: generator terms [g0, g1)
BP_HD void fb_walk_slice(uint32_t &g0, uint32_t &g1, uint32_t s, uint32_t nslice, uint32_t n_gen_terms) {
    const uint32_t per = (n_gen_terms + nslice - 1) / nslice;
    g0 = s * per < n_gen_terms ? s * per : n_gen_terms;
    g1 = g0 + per < n_gen_terms ? g0 + per : n_gen_terms;
}
// the workgroup's wavefronts fold their sums: step `half` = waves/2 .. 1: wavefronts [half, 2 half) store, [0, half) add
BP_HD void fb_walk_fold_store(uint32_t wave, uint32_t lane, uint32_t half, const ge_ext &acc, ge_ext *xch /*[waves/2][64]*/) {
    if (wave >= half && wave < 2 * half) xch[(uint64_t)(wave - half) * 64 + lane] = acc;
}
BP_HD void fb_walk_fold_add(uint32_t wave, uint32_t lane, uint32_t half, ge_ext &acc, const ge_ext *xch) {
    if (wave < half) {
        const ge_ext q = xch[(uint64_t)wave * 64 + lane];
        ge_add(acc, acc, q);
    }
}

// ONE MSM (or a few): lane = slice of the generator terms, a wavefront's 64 sums are folded in six steps through LDS, the workgroup
// leaves one partial sum.  (With lane = MSM a lone MSM used one lane of every wavefront: 148 us for its table walk.)
BP_HD void fb_walk1_fold(uint32_t lane, uint32_t step, ge_ext &acc, ge_ext *xch /*[64]*/, bool store_phase) {
    if (store_phase) {
        if (lane >= step && lane < 2 * step) xch[lane - step] = acc;
    } else if (lane < step) {
        const ge_ext q = xch[lane];
        ge_add(acc, acc, q);
    }
}

// ---- constant-time variant (the prover's secret-dependent commitments) -----------------------------------------------
// The reference computes V, A, S, T_1, T_2 with curve25519-dalek's constant-time multiscalar_mul (party.rs:99-124, 179-187,
// generators.rs:39-41).  The variable-time walk above leaks a secret scalar's digits through table addresses and through the
// skipped additions of zero digits.  This path, selected by the context option "prover_constant_time", uses a small-window
// table (W = 4: 8 entries per (generator, window)) and for every (generator, window) pair
//   * reads ALL 8 entries -- the addresses depend on (generator, window) only, never on the digit;
//   * picks the wanted one with arithmetic masks, the neutral Niels element (1, 1, 0) for digit 0, sign applied by selects;
//   * always performs the same complete mixed addition.
// Instruction stream and memory requests are therefore the same for every scalar (checked with SQ counters for two different
// secret sets: tools/ct_counters.sh, tests/test_gpu_prover_ct.py); results are bit-identical to the variable-time path.
#define BP_FB_CT_W 4
// branch-free fb_recode: adds the (public) constant sum_win half << (W win) with a full carry chain
BP_HD void fb_recode_ct(fb_digit *digits /*stride*/, uint64_t stride, const uint32_t s[8], fb_params prm) {
    uint32_t k[10];
#pragma unroll
    for (int i = 0; i < 10; i++) k[i] = 0;
    for (uint32_t win = 0; win < prm.nwin; win++) {   // public: depends on W only
        const uint32_t bit = win * prm.W + (prm.W - 1);
        if ((bit >> 5) < 10) k[bit >> 5] |= 1u << (bit & 31);
    }
    uint32_t r[10], carry = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const uint64_t t = (uint64_t)(i < 8 ? s[i] : 0u) + k[i] + carry;
        r[i] = (uint32_t)t;
        carry = (uint32_t)(t >> 32);
    }
    for (uint32_t win = 0; win < prm.nwin; win++) {
        const uint32_t bit = win * prm.W, idx = bit >> 5, sh = bit & 31;
        const uint64_t two = (uint64_t)r[idx] | ((uint64_t)(idx + 1 < 10 ? r[idx + 1] : 0u) << 32);
        digits[(uint64_t)win * stride] = (fb_digit)((two >> sh) & ((1u << prm.W) - 1u));
    }
}
BP_HD void fb_recode_ct_thread(uint32_t tid, fb_params prm, uint32_t nproofs, uint32_t n_gen_terms, const uint32_t *gen_scalars, fb_digit *digits) {
    const uint32_t g = tid / nproofs, p = tid % nproofs;
    uint32_t s[8];
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = gen_scalars[((uint64_t)p * n_gen_terms + g) * 8 + i];
    fb_recode_ct(digits + ((uint64_t)g * prm.nwin) * nproofs + p, nproofs, s, prm);   // (the prover's scalars are canonical by construction)
}
// thread (split, p), prm.half == 8
BP_HD void fb_accum_ct_thread(uint32_t p, uint32_t split, uint32_t q0, uint32_t q1, fb_params prm, uint32_t nproofs, const uint32_t *gen_ids,
                              const fb_digit *digits, const fb_entry *table, ge_ext *partial) {
    ge_ext acc;
    ge_identity(acc);
    for (uint32_t q = q0; q < q1; q++) {
        const uint32_t g = q / prm.nwin, win = q - g * prm.nwin;
        const uint32_t *sub = (const uint32_t *)(table + ((uint64_t)gen_ids[g] * prm.nwin + win) * 8);   // 8 entries of 32 words, same address in every lane
        const int d = (int)(digits[(uint64_t)q * nproofs + p] & 15u) - 8;
        const int sgn = d >> 31;                          // all ones for a negative digit
        const uint32_t a = (uint32_t)((d ^ sgn) - sgn);   // |d| in 0..8
        uint32_t sel[30];
#pragma unroll
        for (int i = 0; i < 30; i++) sel[i] = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t m = 0u - (uint32_t)(a == k + 1);
#pragma unroll
            for (int i = 0; i < 30; i++) sel[i] |= sub[32 * k + i] & m;
        }
        const uint32_t m0 = 0u - (uint32_t)(a == 0);      // digit 0: the neutral element (y + x, y - x, 2dxy) = (1, 1, 0)
        sel[0] |= 1u & m0;
        sel[10] |= 1u & m0;
        ge_niels n;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            n.ypx.v[i] = sel[i];
            n.ymx.v[i] = sel[10 + i];
            n.t2d.v[i] = sel[20 + i];
        }
        ge_madd(acc, acc, n, sgn != 0);                   // sign by per-limb selects inside (ge25519.h)
    }
    partial[(uint64_t)split * nproofs + p] = acc;
}

// ---- partial reduction -------------------------------------------------------------------
// thread tid = g * nproofs + p: out[g][p] = sum_{r < group} partial[g*group + r][p]  (r bounded by nsplit)
BP_HD void fb_reduce_thread(uint32_t tid, uint32_t nproofs, uint32_t nsplit, uint32_t group, const ge_ext *partial, ge_ext *out) {
    const uint32_t g = tid / nproofs, p = tid - g * nproofs;
    const uint32_t s0 = g * group, s1 = (s0 + group < nsplit) ? s0 + group : nsplit;
    ge_ext acc = partial[(uint64_t)s0 * nproofs + p];
    for (uint32_t s = s0 + 1; s < s1; s++) {
        const ge_ext q = partial[(uint64_t)s * nproofs + p];
        ge_add(acc, acc, q);
    }
    out[tid] = acc;
}

// ---- finish --------------------------------------------------------------------------
// last step for proof p: compress / identity test of the accumulated point, masked by the front end's status
// (po: the proof's index in the output arrays -- differs from p when the launch is a coalesced one, rangeproof.h rp_seg)
BP_HD void shared_finish_tail(uint32_t p, uint32_t po, const ge_ext &acc, const uint32_t *status, uint32_t *out_words, uint8_t *verdict) {
    const bool bad = status[p] != 0;
    if (out_words) {
        uint32_t w[8];
        ristretto_compress(w, acc);
#pragma unroll
        for (int i = 0; i < 8; i++) out_words[8 * (uint64_t)po + i] = bad ? 0u : w[i];
    }
    if (verdict) verdict[po] = bad ? (uint8_t)status[p] : (ge_is_identity(acc) ? 0 : 1);
}
BP_HD void shared_finish_tail(uint32_t p, const ge_ext &acc, const uint32_t *status, uint32_t *out_words, uint8_t *verdict) {
    shared_finish_tail(p, p, acc, status, out_words, verdict);
}
// lane j (0..7) of proof p: its share of the partial sums, s = j, j+8, ...; lane 7 also takes the Horner result.
// Returns false when the lane has nothing to add (acc is then the identity).
BP_HD bool shared_finish8_gather(ge_ext &acc, uint32_t p, uint32_t j, uint32_t nproofs, uint32_t nsplit, const ge_ext *horner_pre,
                                 const ge_ext *partial) {
    bool have = false;
    for (uint32_t s = j; s < nsplit; s += 8) {
        const ge_ext q = partial[(uint64_t)s * nproofs + p];
        if (have) ge_add(acc, acc, q);
        else acc = q;
        have = true;
    }
    if (j == 7 && horner_pre) {
        const ge_ext q = horner_pre[p];
        if (have) ge_add(acc, acc, q);
        else acc = q;
        have = true;
    }
    if (!have) ge_identity(acc);
    return have;
}
// thread p: result = Horner(col[p]) (unique, variable-base terms) + sum_split partial[split][p]
// out_words (optional): compressed result; verdict (optional): status[p] if set, else 0 identity / 1 not
// horner_pre (optional): Horner results already computed by the wavefront-cooperative kernel (horner_wave.h)
BP_HD void shared_finish_thread(uint32_t p, uint32_t nproofs, uint32_t nsplit, const ge_ext *col, bool have_unique,
                                const ge_ext *horner_pre, const ge_ext *partial, const uint32_t *status, uint32_t *out_words,
                                uint8_t *verdict) {
    ge_ext acc;
    if (have_unique && horner_pre) acc = horner_pre[p];
    else if (have_unique) vb_horner_point(acc, col + (uint64_t)p * 64);
    else ge_identity(acc);
    for (uint32_t s = 0; s < nsplit; s++) {
        const ge_ext q = partial[(uint64_t)s * nproofs + p];
        ge_add(acc, acc, q);
    }
    shared_finish_tail(p, acc, status, out_words, verdict);
}

}  // namespace bp
#endif
