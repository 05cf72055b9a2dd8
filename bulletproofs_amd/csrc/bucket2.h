// Bucket multiscalar multiplication, second form (round 6): the chain north_star describes -- "LDS-staged bucket accumulation" --
// for MSMs of up to a few thousand variable-base terms (c = 8: 32 windows x 128 buckets), i.e. the R1CS verifier's shape
// (src/r1cs/verifier.rs:459-491: 2 081 per-proof points) and the stand-alone inner-product check (src/inner_product_proof.rs:308-319).
// bucket.h's chain stays for what this one does not serve: c = 12 (>= 6 000 terms per MSM), the split sort of very large MSMs and
// the batch-combination mode of the range-proof check.
//
//   bk2_prepare : lane = term          decode (short-register form, ge25519.h: ristretto_decompress_lp) into a 128-byte affine Niels
//                                      record; recode r = s + k l + sum_w 128 * 256^w (bucket.h: bk_recode); the 32 bytes of r ARE the
//                                      32 window digits (digit = byte - 128): written window-major, dig[w][t], so that a window's
//                                      workgroup reads its digits as one contiguous run (the old chain read 36-byte records with a
//                                      stride: 1.9 MB of sort traffic per 6 179-term MSM)
//   bk2_window  : workgroup = (MSM, window)   ONE launch for what were bk_sort + bk_accum + bk_heavy:
//        w0-w2  histogram of |digit| in LDS (atomics), scan, cursors
//        w3     scatter of (term, sign) as 16-bit entries into the LDS list, sorted by bucket -- the list never leaves LDS
//        w4     every lane takes the SAME number of consecutive list entries (q = ceil(n / LANES)) whatever the bucket populations
//               are: one mixed addition per entry, the point record of entry i+1 in flight while entry i is added.  A lane's run
//               of entries crosses bucket boundaries; a finished bucket's sum goes straight to bsum (the common case: the bucket
//               began and ended inside the lane's run), the piece of a bucket that began in an earlier lane goes to that lane's
//               "head" slot in LDS
//        w5     the lane in which a bucket began adds the head pieces of the lanes after it (nearly always one) and stores the sum
//      Bucket populations no longer matter: the old chain gave one lane one bucket (population-sorted: 27 additions in the
//      longest lane of a wavefront against 16 on average -- 0.76 of the lanes' time used) and needed a second pass (bk_heavy) so that
//      crowded buckets (equal or structured scalars) would not become one lane's serial chain.  Here every lane adds q entries, a
//      crowded bucket is simply spread over many lanes, and its pieces are combined by the lane that owns it.
//   then bucket.h's running-sum tree (bk_leaf / bk_tree) and the Horner chain, unchanged.
//
// Bit-exact by construction: the order of additions differs from bucket.h (and between runs: the scatter's order inside a bucket
// is set by atomics), the group element does not, and only its canonical encoding is ever compared.
//
// As in bucket.h the workgroup stages are per-lane PHASE functions separated by barriers, so that tests/cpu_harness runs the
// identical code with a loop over lanes per phase.
#ifndef BPGPU_BUCKET2_H
#define BPGPU_BUCKET2_H
#include "bucket.h"
#include "horner_wave.h"

namespace bp {

#define BK2_C 8u
#define BK2_NWIN 32u
#define BK2_HALF 128u
#define BK2_MAX_TERMS 6144u   // per MSM: the 16-bit list entries hold a term index < 32 768 and a sign; LDS: 2 bytes per term

// ---- stage 1: lane = term --------------------------------------------------------------------------------------------------
// dig: [32][total] bytes, window-major
BP_HD void bk2_prepare_thread(uint32_t t, uint32_t total, uint32_t nbatch, const uint32_t *msm_first, const uint32_t *scalars, const uint32_t *points,
                              fb_entry *pts, uint8_t *dig, uint32_t *status) {
    uint32_t sw[8], r[BK_RWORDS];
#pragma unroll
    for (int i = 0; i < 8; i++) sw[i] = scalars[8 * (uint64_t)t + i];
    const bool canon = sc_is_canonical(sw);
    bk_recode(r, sw, bk_make(BK2_C), t);   // r < 2^256 (k < 7): r[8] == 0
#pragma unroll
    for (int i = 0; i < 8; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) dig[(uint64_t)(4 * i + j) * total + t] = (uint8_t)(r[i] >> (8 * j));
    }
    ge_ext p;
    const bool ok = ristretto_decompress_lp(p, points + 8 * (uint64_t)t);
    if (!canon || !ok) {
        const uint32_t msm = bk_find_msm(t, msm_first, nbatch);
        status_raise(status + msm, canon ? BP_STATUS_BAD_POINT : BP_STATUS_BAD_SCALAR);
    }
    bk_store_point(pts + t, p);
}

// ---- stage 2: workgroup = (MSM, window) -----------------------------------------------------------------------------------
struct bk2_seg {
    uint32_t first, count;   // the MSM's terms: [first, first + count), count <= BK2_MAX_TERMS
    uint32_t lanes;          // lanes of the workgroup: 64, 128 or 256
    const uint8_t *dig_w;    // digits of this window: dig + w * total
};
struct bk2_lds {
    uint32_t *cnt;     // [128] populations, later the scatter cursors
    uint32_t *off;     // [129] exclusive prefix: entries of bucket j are list[off[j] .. off[j+1])
    uint32_t *tmp;     // [128] scan ping-pong
    uint16_t *list;    // [count] (term - first) | sign << 15, sorted by bucket
    ge_ext *head;      // [lanes] the piece of a bucket that began in an earlier lane
};
BP_HD void bk2_w0(uint32_t lane, const bk2_seg &sg, const bk2_lds &l) {
    for (uint32_t j = lane; j < BK2_HALF; j += sg.lanes) l.cnt[j] = 0;
}
BP_HD void bk2_w1(uint32_t lane, const bk2_seg &sg, const bk2_lds &l) {   // histogram
    for (uint32_t i = lane; i < sg.count; i += sg.lanes) {
        const int d = (int)sg.dig_w[sg.first + i] - (int)BK2_HALF;
        if (d != 0) BK_ATOMIC_ADD(&l.cnt[(uint32_t)(d < 0 ? -d : d) - 1], 1u);
    }
}
// inclusive scan of cnt[0 .. 128) in seven steps (a <-> b ping-pong, one barrier per step): step s reads `src`, writes `dst`
BP_HD void bk2_w2_step(uint32_t lane, uint32_t s, const bk2_seg &sg, const uint32_t *src, uint32_t *dst) {
    for (uint32_t j = lane; j < BK2_HALF; j += sg.lanes) dst[j] = src[j] + (j >= s ? src[j - s] : 0u);
}
// after the seven steps the inclusive sums are in tmp (7 is odd: cnt -> tmp -> cnt ... -> tmp): offsets and cursors
BP_HD void bk2_w2_fin(uint32_t lane, const bk2_seg &sg, const bk2_lds &l) {
    for (uint32_t j = lane; j < BK2_HALF; j += sg.lanes) {
        const uint32_t ex = j ? l.tmp[j - 1] : 0u;
        l.off[j] = ex;
        l.cnt[j] = ex;
    }
    if (lane == 0) l.off[BK2_HALF] = l.tmp[BK2_HALF - 1];
}
BP_HD void bk2_w3(uint32_t lane, const bk2_seg &sg, const bk2_lds &l) {   // scatter
    for (uint32_t i = lane; i < sg.count; i += sg.lanes) {
        const int d = (int)sg.dig_w[sg.first + i] - (int)BK2_HALF;
        if (d != 0) {
            const uint32_t pos = BK_ATOMIC_ADD(&l.cnt[(uint32_t)(d < 0 ? -d : d) - 1], 1u);
            l.list[pos] = (uint16_t)(i | (d < 0 ? 0x8000u : 0u));
        }
    }
}
// what a lane knows after its run of entries (registers between w4 and w5)
struct bk2_tail {
    ge_ext acc;        // sum of the lane's last piece
    uint32_t bucket;   // its bucket
    uint32_t owner;    // 1: the bucket began in this lane and goes on in the next ones -- w5 adds their head pieces and stores it
};
BP_HD uint32_t bk2_q(uint32_t n, uint32_t lanes) { return (n + lanes - 1) / lanes; }
// bucket that holds list position pos (pos < off[128]): the last j with off[j] <= pos
BP_HD uint32_t bk2_bucket_of(uint32_t pos, const uint32_t *off) {
    uint32_t lo = 0, hi = BK2_HALF;   // invariant: off[lo] <= pos < off[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= pos) lo = mid;
        else hi = mid;
    }
    return lo;
}
BP_HD void bk2_load_niels(ge_niels &nn, const fb_line &line) {
#pragma unroll
    for (int q = 0; q < 10; q++) {
        nn.ypx.v[q] = line.w[q];
        nn.ymx.v[q] = line.w[10 + q];
        nn.t2d.v[q] = line.w[20 + q];
    }
}
// one entry of a lane's run: close the bucket that ended before `pos` (its sum, or head piece, is complete), then add the entry
BP_HD void bk2_w4_step(uint32_t lane, uint32_t pos, uint32_t &j, uint32_t &bend, bool &head_piece, ge_ext &acc, const fb_line &line, uint32_t e,
                       const bk2_lds &l, ge_ext *bsum_w) {
    if (pos >= bend) {
        if (head_piece) l.head[lane] = acc;
        else bsum_w[j] = acc;
        head_piece = false;
        ge_identity(acc);
        do j++;
        while (l.off[j + 1] <= pos);
        bend = l.off[j + 1];
    }
    ge_niels nn;
    bk2_load_niels(nn, line);
    ge_madd(acc, acc, nn, (e >> 15) != 0);   // (also for the first entry of a piece: the formulas are complete)
}
// w4: the lane's run of q entries.  bsum_w: this (MSM, window)'s 128 bucket sums in global memory; pts_m: the MSM's point records
BP_HD void bk2_w4(uint32_t lane, const bk2_seg &sg, const bk2_lds &l, const fb_entry *pts_m, ge_ext *bsum_w, bk2_tail &tl) {
    // empty buckets: the identity
    for (uint32_t j = lane; j < BK2_HALF; j += sg.lanes) {
        if (l.off[j + 1] == l.off[j]) {
            ge_ext id;
            ge_identity(id);
            bsum_w[j] = id;
        }
    }
    const uint32_t n = l.off[BK2_HALF], q = bk2_q(n, sg.lanes);
    const uint32_t lo = lane * q, hi = (lo + q < n) ? lo + q : n;
    tl.owner = 0;
    tl.bucket = 0;
    if (lo >= hi) return;
    uint32_t j = bk2_bucket_of(lo, l.off);
    uint32_t bend = l.off[j + 1];
    bool head_piece = lo > l.off[j];   // the lane's first piece continues a bucket that began in an earlier lane
    ge_ext acc;
    ge_identity(acc);
    // one record buffer and a copy per trip: alternating two buffers (no copies) needs 25 registers more, which under the
    // three-wavefront cap spill -- measured slower (238 against 202 us for 64 MSMs, profiles/r06/cfg5_two_buffers_ab.txt); uncapped
    // (191 registers, two wavefronts per SIMD) it is slower too: 108.5-109.4 against 110.9-111.4 k MSMs/s (cfg5_two_buffers_uncapped_and_walk_waves_ab.txt)
    uint32_t e_cur = l.list[lo];
    fb_line line_cur;
    fb_load_line(line_cur, pts_m + (e_cur & 0x7fffu));
    for (uint32_t pos = lo; pos < hi; pos++) {
        fb_line line_next = line_cur;
        uint32_t e_next = 0;
        if (pos + 1 < hi) {
            e_next = l.list[pos + 1];
            fb_load_line(line_next, pts_m + (e_next & 0x7fffu));
        }
        bk2_w4_step(lane, pos, j, bend, head_piece, acc, line_cur, e_cur, l, bsum_w);
        line_cur = line_next;
        e_cur = e_next;
    }
    // the last piece: bucket j, entries [max(lo, off[j]), hi)
    if (head_piece) {   // the whole run lies inside a bucket that began earlier
        l.head[lane] = acc;
    } else if (bend == hi) {   // began here, ends here
        bsum_w[j] = acc;
    } else {   // began here, goes on in the following lanes
        tl.acc = acc;
        tl.bucket = j;
        tl.owner = 1;
    }
}
// w5: owners add the head pieces of the lanes that continue their bucket.  Those are the lanes lane+1 .. whose run begins before the
// bucket's end; nearly always one (a bucket of the average population is as long as a run), up to lanes-1 when one bucket holds
// everything (equal scalars: the owner's chain is then lanes-1 additions after runs of n / lanes -- bounded by the workgroup's
// width, where bucket.h's one-lane-per-bucket form had a chain of n and needed the heavy pass).
BP_HD void bk2_w5(uint32_t lane, const bk2_seg &sg, const bk2_lds &l, ge_ext *bsum_w, bk2_tail &tl) {
    if (!tl.owner) return;
    const uint32_t n = l.off[BK2_HALF], q = bk2_q(n, sg.lanes), bend = l.off[tl.bucket + 1];
    ge_ext acc = tl.acc;
    for (uint32_t k = lane + 1; k < sg.lanes && k * q < bend; k++) {
        const ge_ext h = l.head[k];
        ge_add(acc, acc, h);
    }
    bsum_w[tl.bucket] = acc;
}

// ---- the chain's tail: ONE launch, workgroup (one wavefront) = MSM -----------------------------------------------------------
// What were bk_tree + horner_wave + (fb_reduce x 3) + shared_finish.  The device runs about four kernels at a time whatever their
// width, so four narrow launches in a row cost a chain four slots' worth of serial latency plus the gaps between them.
//   t1  lane w < 32: window sum of window w from the eight leaf nodes (bucket.h: bk_combine) -> colq8[w] in LDS
//   t2  lane l: the generator half's partial sums l, l + 64, ...; six folding steps through LDS leave their sum in lane 0
//   t3  the wavefront-cooperative Horner chain over the 32 window sums (horner_wave.h: hw_horner8), straight from LDS
//   t4  lane 0: Horner result + generator half, status, canonical encoding (short-register form) / identity test
BP_HD void bk2_tail_t1(uint32_t lane, uint32_t b, const ge_ext *gS, const ge_ext *gA, uint32_t *colq8 /*[32][32 words]*/) {
    if (lane >= BK2_NWIN) return;
    ge_ext S, A;
    bk_combine(S, A, gS, gA, (b * BK2_NWIN + lane) * 8, 8, 1, 16);
    vb_encode_colq16(colq8 + lane * 32, A);
}
BP_HD void bk2_tail_t2(uint32_t lane, uint32_t b, uint32_t nmsm, uint32_t npart, const ge_ext *partial, ge_ext &acc) {
    bool have = false;
    for (uint32_t s = lane; s < npart; s += 64) {
        const ge_ext q = partial[(uint64_t)s * nmsm + b];
        if (have) ge_add(acc, acc, q);
        else acc = q;
        have = true;
    }
    if (!have) ge_identity(acc);
}
// fin: the MSM's sum, parked in LDS (the encoding reads it twice).  In three steps so that the 252-squaring chain in the middle can be run
// by the whole wavefront (horner_wave.h: hw_invsqrt_raw_fe):  t4a (lane 0)  t = u1 u2^2 and its canonical limbs;  chain: raw = t^3 (t^7)^((p-5)/8);
// t4b (lane 0)  the encoding from raw, the verdict, the status
BP_HD void bk2_tail_t4a(const ge_ext *fin, fe *tin, uint32_t *tw /*[8]*/) {
    fe t;
    ristretto_compress_front(t, fin);
    *tin = t;
    fe_to_words(tw, t);
}
BP_HD void bk2_tail_t4b(uint32_t b, const ge_ext *fin, const fe *raw, const fe *tin, const uint32_t *status, uint32_t *out_words, uint8_t *verdict, uint8_t *status_bytes) {
    const uint32_t st = status[b];
    if (out_words) {
        uint32_t w[8];
        fe I;
        fe_invsqrt_fix(I, *raw, *tin);
        ristretto_compress_back(w, fin, I);
#pragma unroll
        for (int i = 0; i < 8; i++) out_words[8 * (uint64_t)b + i] = st ? 0u : w[i];
    }
    if (verdict) verdict[b] = st ? (uint8_t)st : (ge_is_identity(*fin) ? 0 : 1);
    if (status_bytes) status_bytes[b] = (uint8_t)st;
}

// ---- the tail for narrow chains (fewer than BK2_FAST_MAX_MSMS MSMs): leaves of 4 buckets, workgroup = 256 lanes = MSM -----------------
// One MSM's chain ended with 30 + 27 additions in sequence (a leaf's running sum over 16 buckets, then the eight leaves of a window
// combined by one lane) before the Horner chain could start: 167 us of the 0.65 ms a lone 6 179-term MSM took.  With lanes to spare the
// same sums are formed with short chains:
//   leaf l (of 32 per window) = 4 buckets: the running sums S, A (6 additions), then  V = A + 4 l S  by double-and-add in the same lane
//   (l < 32: 6 doublings + 6 additions, the same code in every lane) -- the window sum is the sum of its 32 V;
//   lane (w, g) of the tail adds four V, the 8 lanes of a window add theirs with three wavefront shuffles (__shfl_down of the 40 words:
//   no LDS, no barrier);  then wavefront 0 runs the Horner chain while wavefront 1 adds the generator half's partial sums (six shuffle steps).
// (The leaf launch spreads its 1 024 lanes per MSM over many CUs; the first version formed V in the tail's workgroup, two wavefronts per
// SIMD of ONE CU: 77 us for what is 50 here.)  A wide batch keeps bk_leaf_thread / bk2_tail_t1: a third of the instructions.
#define BK2_FAST_LEAVES 32u
#define BK2_FAST_MAX_MSMS 48u
// tid = bw * 32 + l
BP_HD void bk2_leafv_thread(uint32_t tid, const ge_ext *bsum, ge_ext *gV) {
    const uint32_t bw = tid / BK2_FAST_LEAVES, l = tid % BK2_FAST_LEAVES, m = BK2_HALF / BK2_FAST_LEAVES;
    const ge_ext *b = bsum + (uint64_t)bw * BK2_HALF + (uint64_t)l * m;
    ge_ext run = b[m - 1], acc = run;
    for (uint32_t i = m - 1; i-- > 0;) {
        const ge_ext q = b[i];
        ge_add(run, run, q);
        ge_add(acc, acc, run);
    }
    ge_ext Y, T;
    ge_identity(Y);
#pragma unroll 1
    for (uint32_t bit = 0; bit < 5; bit++) {
        ge_add(T, Y, run);
        ge_select(Y, Y, T, ((l >> bit) & 1u) != 0);
        if (bit < 4) ge_dbl(run, run);
    }
    ge_dbl(Y, Y, false);
    ge_dbl(Y, Y);
    ge_add(acc, acc, Y);
    gV[tid] = acc;
}
// lane (w, g), g < 8: the sum of V[4 g .. 4 g + 4) of window w of MSM b
BP_HD void bk2_fast_v4(uint32_t w, uint32_t g, uint32_t b, const ge_ext *gV, ge_ext &V) {
    const ge_ext *v = gV + ((uint64_t)b * BK2_NWIN + w) * BK2_FAST_LEAVES + 4 * g;
    V = v[0];
    for (uint32_t i = 1; i < 4; i++) {
        const ge_ext q = v[i];
        ge_add(V, V, q);
    }
}
}  // namespace bp
#endif
