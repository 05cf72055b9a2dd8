// Bucket ("Pippenger") multiscalar multiplication over variable-base points -- the algorithm the reference
// reaches through RistrettoPoint::optional_multiscalar_mul for large inputs (upstream curve25519-dalek switches to
// its Pippenger implementation at 190 terms; call sites src/r1cs/verifier.rs:459-491 with 2081 per-proof points,
// src/range_proof/mod.rs:421-445 for aggregated proofs), here laid out for a GPU:
//
//   bk_prepare : lane = term            decode the point into affine Niels form (one 128-byte record), check and
//                                       recode the scalar: r = s + sum_w half * 2^(c w), so that window w's signed
//                                       digit is ((r >> c w) & (2^c - 1)) - half
//   bk_sort    : workgroup = (MSM, window)   counting sort of the MSM's terms by |digit| in LDS (histogram, scan,
//                                       scatter of term indices), then a second counting sort of the BUCKETS by
//                                       their population, so that the 64 lanes of a wavefront own buckets of
//                                       (nearly) equal length
//   bk_accum   : lane = (MSM, window, bucket rank)   bucket sum: one mixed addition (7 multiplications) per listed
//                                       term, software-pipelined gathers of the 128-byte point records
//   bk_leaf / bk_tree : window sum = sum_j j * B_j by a tree of running sums: every node keeps
//                                       (S, A) = (sum of its buckets, sum of (j - j0 + 1) B_j) and a parent of k children
//                                       of width w forms S = sum S_i, A = sum A_i + w * sum i S_i with one running sum;
//                                       lane = leaf for the bottom level, upper levels packed many windows per wavefront
//   then the window sums enter the existing Horner chain (horner_wave.h) as radix-16 column sums: window w of c
//   bits is column (c/4) w, the columns in between are the identity.
//
// c = 8 (32 windows x 128 buckets) for MSMs of up to a few thousand terms, c = 12 (22 windows x 2048
// buckets) beyond: the per-window cost is N + 2^(c-1) * ~2.5 additions.  Results are bit-identical to the
// table-lookup path (msm_vb.h) and to the oracle because only the canonical encoding of the sum is ever compared:
// the order of additions inside a bucket (set by atomics) does not change the group element.
//
// The workgroup-cooperative stages are written as per-lane PHASE functions separated by barriers, so that the CPU
// harness (tests/cpu_harness) runs the identical code with a loop over lanes per phase.
#ifndef BPGPU_BUCKET_H
#define BPGPU_BUCKET_H
#include "msm_fixed.h"

namespace bp {

struct bk_params {
    uint32_t c;      // window bits: 8 or 12
    uint32_t nwin;   // windows: 32 (c = 8), 22 (c = 12; the last one holds bit 252 and the recoding carry)
    uint32_t half;   // 2^(c-1) buckets per window (magnitudes 1 .. half)
    uint32_t lanes;  // lanes of a (MSM, window) workgroup in bk_sort: 64 (c = 8), 256 (c = 12)
};
BP_HD bk_params bk_make(uint32_t c) {
    bk_params p;
    p.c = c;
    p.nwin = c == 8 ? 32u : 22u;
    p.half = 1u << (c - 1);
    p.lanes = c == 8 ? 64u : 256u;
    return p;
}
#define BK_RWORDS 9          // recoded scalar: < 2^264
// Measured crossovers (MI355X, tools/archive/msm_crossover.py, profiles/r02/msm_crossover.txt): a single MSM gains from the
// buckets from ~2000 terms (1.8x at 8192, 3.1x at 20 000, 6.6x at 65 536), a batch of 64 MSMs from ~1500 terms per MSM
// (1.25x at 2081, 1.43x at 4096); below that the per-point 8-entry tables of msm_vb.h win (fewer, wider launches).
#define BK_MIN_TERMS 1536     // terms per MSM from which the bucket path is taken (option "bucket_min_terms")
// The batch-combined range-proof check is ONE MSM over batch * (4 + 2k + m) terms, but it sits in a chain of narrow
// launches where the extra sort / tree levels cost latency: measured (cfg2 proofs, 17 terms each) the bucket variant
// does 7.5 M proofs/s against 8.1 M/s at batch 1024 (17 408 terms), 10.7 against 9.8 at batch 2048, 13.5 - 14.4
// against 10.7 at batch 4096, 12.6 against 10.8 at batch 16 384: it is taken from 32 768 terms.
#define BK_RLC_MIN_TERMS 32768

// bucket descriptor, sorted by population (descending) inside each (MSM, window)
struct bk_desc {
    uint32_t off;      // first entry of the bucket's list in idx[window][msm_first + ...]
    uint32_t cnt;      // entries
    uint32_t bucket;   // magnitude - 1
    uint32_t pad;
};

#if defined(__HIP_DEVICE_COMPILE__)
#define BK_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#else
static inline uint32_t bk_host_atomic_add(uint32_t *p, uint32_t v) {
    const uint32_t o = *p;
    *p = o + v;
    return o;
}
#define BK_ATOMIC_ADD(p, v) bk_host_atomic_add((p), (v))
#endif

// r = s + k l + sum_{w < nwin} half << (c w),  k = a pseudo-random multiple count derived from the term index `salt`.
// Adding multiples of the group order l does not change the result: l P lies in the 4-torsion for every decoded
// ristretto point (they are even points of the curve), and the ristretto encoding is invariant under it -- the same
// fact that makes P + (l - 1) P encode as the identity.  What it buys: the scalar the windows see is uniform over
// [0, K l) whatever the caller's scalars are, so (i) the TOP window, which for canonical scalars (< 2^253) would only
// ever hold the digits 0, 1, 2 -- i.e. one bucket with half of all terms, a serial chain for one lane -- is as evenly
// populated as the others, and (ii) equal or structured scalars (all ones, small values) no longer pile up in one
// bucket per window.  K = 7 (c = 8: 7 l < 2^255 - 2^247) / 1024 (c = 12: 1024 l < 2^263), so r < 2^(c nwin).
// LIMIT of (ii): l = 2^252 + delta with delta < 2^125, so k l only stirs bits 0 .. ~134 (k delta) and 252 .. (k): scalars that agree
// in bits 135 .. 251 still share those windows' buckets.  Measured in round 4: the batch-combined check with 128-BIT weights (A's
// coefficient is the bare weight) put 1/9 of all proofs into ONE bucket of window 11 -- the bucket-sum launch of a 4096-proof
// combination went from 0.21 to 2.2 ms (profiles/r04/ab_rlc_r03_vs_r04.txt).  Two consequences: the library's own weights are
// full-width (expanded on the device from a per-chain key, rangeproof.h: rp_shape::seed), and crowded buckets no longer cost a serial
// chain whatever the scalars are (stage 3b below: a lane's share is capped, the rest is summed 64 lanes at a time).
BP_HD void bk_recode(uint32_t r[BK_RWORDS], const uint32_t s[8], bk_params prm, uint32_t salt) {
    const uint32_t l[8] = BP_L_WORDS;
    const uint32_t h = (salt * 0x9E3779B1u) >> 12;
    const uint32_t k = prm.c == 8 ? h % 7u : (h & 1023u);
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t t = (uint64_t)s[i] + (uint64_t)k * l[i] + carry;
        r[i] = (uint32_t)t;
        carry = t >> 32;
    }
    r[8] = (uint32_t)carry;
    for (uint32_t w = 0; w < prm.nwin; w++) {
        const uint32_t bit = w * prm.c + (prm.c - 1);
        uint32_t idx = bit >> 5;
        uint64_t t = (uint64_t)r[idx] + (1u << (bit & 31));
        r[idx] = (uint32_t)t;
        uint32_t cy = (uint32_t)(t >> 32);
        while (cy && ++idx < BK_RWORDS) {
            t = (uint64_t)r[idx] + cy;
            r[idx] = (uint32_t)t;
            cy = (uint32_t)(t >> 32);
        }
    }
}
// signed digit of window w: in [-half, half)
BP_HD int bk_digit(const uint32_t *r /*BK_RWORDS words*/, uint32_t w, bk_params prm) {
    const uint32_t bit = w * prm.c, idx = bit >> 5, sh = bit & 31;
    uint64_t two = (uint64_t)r[idx];
    if (idx + 1 < BK_RWORDS) two |= (uint64_t)r[idx + 1] << 32;
    return (int)((uint32_t)(two >> sh) & ((1u << prm.c) - 1u)) - (int)prm.half;
}

// decoded point -> affine Niels record (the decoder returns Z = 1)
BP_HD void bk_store_point(fb_entry *dst, const ge_ext &p) {
    const fe d2 = BP_FE_D2;
    fb_entry e;
    fe_add(e.ypx, p.Y, p.X);
    fe_carry(e.ypx);
    fe_sub(e.ymx, p.Y, p.X);
    fe_mul(e.t2d, p.T, d2);
    e.pad[0] = 0;
    e.pad[1] = 0;
    *dst = e;
}

// which MSM term t belongs to: msm_first[b] <= t < msm_first[b + 1]
BP_HD uint32_t bk_find_msm(uint32_t t, const uint32_t *msm_first, uint32_t nbatch) {
    uint32_t lo = 0, hi = nbatch;   // invariant: msm_first[lo] <= t < msm_first[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (msm_first[mid] <= t) lo = mid;
        else hi = mid;
    }
    return lo;
}

// ---- stage 1: lane = term ----------------------------------------------------------------------------------
BP_HD void bk_prepare_thread(uint32_t t, uint32_t nbatch, const uint32_t *msm_first, const uint32_t *scalars, const uint32_t *points,
                             fb_entry *pts, uint32_t *rwords, uint32_t *status, bk_params prm) {
    uint32_t sw[8], pw[8], r[BK_RWORDS];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        sw[i] = scalars[8 * (uint64_t)t + i];
        pw[i] = points[8 * (uint64_t)t + i];
    }
    const uint32_t msm = bk_find_msm(t, msm_first, nbatch);
    ge_ext p;
    const bool ok = ristretto_decompress(p, pw);
    const bool canon = sc_is_canonical(sw);
    if (!canon) status_raise(status + msm, BP_STATUS_BAD_SCALAR);
    else if (!ok) status_raise(status + msm, BP_STATUS_BAD_POINT);
    bk_recode(r, sw, prm, t);
#pragma unroll
    for (int i = 0; i < BK_RWORDS; i++) rwords[BK_RWORDS * (uint64_t)t + i] = r[i];
    bk_store_point(pts + t, p);
}

// ---- stage 2: workgroup = (MSM b, window w); LDS: cnt[half], off[half], part[lanes], hist2[256] ------------------
// All phases take (lane, seg) where seg describes the workgroup's MSM segment.
struct bk_seg {
    uint32_t first, count;   // the MSM's terms: [first, first + count)
    uint32_t w;              // window
    uint32_t sub, nsub;      // this workgroup's share of the terms: [count sub / nsub, count (sub + 1) / nsub)  (1 of 1: all)
    // batch-combination mode: term t belongs to proof t / skip_div; terms of proofs whose status word is set stay out
    const uint32_t *skip_status;
    uint32_t skip_div;
};
BP_HD bool bk_term_skipped(const bk_seg &sg, uint32_t t) { return sg.skip_status && sg.skip_status[t / sg.skip_div] != 0; }
BP_HD uint32_t bk_sub_lo(const bk_seg &sg) { return (uint32_t)((uint64_t)sg.count * sg.sub / sg.nsub); }
BP_HD uint32_t bk_sub_hi(const bk_seg &sg) { return (uint32_t)((uint64_t)sg.count * (sg.sub + 1) / sg.nsub); }
struct bk_lds {
    uint32_t *cnt;    // [half] bucket populations, later the scatter cursors
    uint32_t *off;    // [half] exclusive prefix of cnt
    uint32_t *part;   // [lanes] scan partials
    uint32_t *hist2;  // [256] histogram of min(cnt, 255), later the descriptor cursors
};
BP_HD void bk_sort_p0(uint32_t lane, bk_params prm, const bk_lds &l) {   // clear
    for (uint32_t j = lane; j < prm.half; j += prm.lanes) l.cnt[j] = 0;
    for (uint32_t j = lane; j < 256; j += prm.lanes) l.hist2[j] = 0;
}
BP_HD void bk_sort_p1(uint32_t lane, bk_params prm, const bk_seg &sg, const uint32_t *rwords, const bk_lds &l) {   // histogram
    for (uint32_t i = bk_sub_lo(sg) + lane; i < bk_sub_hi(sg); i += prm.lanes) {
        if (bk_term_skipped(sg, sg.first + i)) continue;
        const int d = bk_digit(rwords + BK_RWORDS * (uint64_t)(sg.first + i), sg.w, prm);
        if (d != 0) BK_ATOMIC_ADD(&l.cnt[(uint32_t)(d < 0 ? -d : d) - 1], 1u);
    }
}
BP_HD void bk_sort_p2(uint32_t lane, bk_params prm, const bk_lds &l) {   // per-lane partial sums of a contiguous chunk
    const uint32_t per = prm.half / prm.lanes;
    uint32_t s = 0;
    for (uint32_t j = lane * per; j < (lane + 1) * per; j++) s += l.cnt[j];
    l.part[lane] = s;
}
BP_HD void bk_sort_p3(uint32_t lane, bk_params prm, const bk_lds &l) {   // lane 0: scan of the partials (<= 256 entries)
    if (lane != 0) return;
    uint32_t run = 0;
    for (uint32_t i = 0; i < prm.lanes; i++) {
        const uint32_t v = l.part[i];
        l.part[i] = run;
        run += v;
    }
}
BP_HD void bk_sort_p4(uint32_t lane, bk_params prm, const bk_lds &l) {   // offsets of the chunk + histogram of populations
    const uint32_t per = prm.half / prm.lanes;
    uint32_t run = l.part[lane];
    for (uint32_t j = lane * per; j < (lane + 1) * per; j++) {
        const uint32_t cn = l.cnt[j];
        l.off[j] = run;
        run += cn;
        BK_ATOMIC_ADD(&l.hist2[cn < 255 ? cn : 255], 1u);
    }
}
BP_HD void bk_sort_p5(uint32_t lane, const bk_lds &l) {   // lane 0: descending scan: first descriptor slot of each population class
    if (lane != 0) return;
    uint32_t run = 0;
    for (int k = 255; k >= 0; k--) {
        const uint32_t v = l.hist2[k];
        l.hist2[k] = run;
        run += v;
    }
}
BP_HD void bk_sort_p6(uint32_t lane, bk_params prm, const bk_seg &sg, const bk_lds &l, bk_desc *desc /*this workgroup's [half]*/) {
    // descriptors sorted by population (descending); cnt[] becomes the scatter cursor
    const uint32_t per = prm.half / prm.lanes;
    for (uint32_t j = lane * per; j < (lane + 1) * per; j++) {
        const uint32_t cn = l.cnt[j];
        const uint32_t pos = BK_ATOMIC_ADD(&l.hist2[cn < 255 ? cn : 255], 1u);
        bk_desc d;
        d.off = sg.first + l.off[j];
        d.cnt = cn;
        d.bucket = j;
        d.pad = 0;
        desc[pos] = d;
        l.cnt[j] = l.off[j];
    }
}
BP_HD void bk_sort_p7(uint32_t lane, bk_params prm, const bk_seg &sg, const uint32_t *rwords, const bk_lds &l, uint32_t *idx_w /*idx of window w: [total]*/) {
    for (uint32_t i = bk_sub_lo(sg) + lane; i < bk_sub_hi(sg); i += prm.lanes) {
        const uint32_t t = sg.first + i;
        if (bk_term_skipped(sg, t)) continue;
        const int d = bk_digit(rwords + BK_RWORDS * (uint64_t)t, sg.w, prm);
        if (d != 0) {
            const uint32_t pos = BK_ATOMIC_ADD(&l.cnt[(uint32_t)(d < 0 ? -d : d) - 1], 1u);
            idx_w[sg.first + pos] = t | (d < 0 ? 0x80000000u : 0u);
        }
    }
}

// Large MSMs (tens of thousands of terms and more): the terms of one (MSM, window) are shared by `nsub` workgroups.
// Each histograms its share in LDS and merges into a global histogram (bk_sort_merge); ONE workgroup per (MSM, window)
// then scans / emits the descriptors (p2 .. p6 on a copy of the global histogram) and publishes the cursors; the
// scatter (p7) runs again on all `nsub` workgroups with the cursors in global memory.
BP_HD void bk_sort_merge(uint32_t lane, bk_params prm, const bk_lds &l, uint32_t *gcnt /*[half] of this (MSM, window)*/) {
    for (uint32_t j = lane; j < prm.half; j += prm.lanes) {
        const uint32_t v = l.cnt[j];
        if (v) BK_ATOMIC_ADD(&gcnt[j], v);
    }
}
BP_HD void bk_sort_load(uint32_t lane, bk_params prm, const bk_lds &l, const uint32_t *gcnt) {
    for (uint32_t j = lane; j < prm.half; j += prm.lanes) l.cnt[j] = gcnt[j];
    for (uint32_t j = lane; j < 256; j += prm.lanes) l.hist2[j] = 0;
}
BP_HD void bk_sort_publish(uint32_t lane, bk_params prm, const bk_lds &l, uint32_t *gcur) {
    for (uint32_t j = lane; j < prm.half; j += prm.lanes) gcur[j] = l.cnt[j];
}

// ---- stage 3: lane = (workgroup id bw, rank r) ------------------------------------------------------------------
// A lane adds its whole bucket as long as that is at most `lim` = 2 x the average population + 32 terms (bk_chain_lim; with uniform
// digits no bucket comes near it: average 34 -> largest ~60).  Of a CROWDED bucket it adds the first lim - 32; the rest (>= 33 terms)
// is added by the heavy pass below, 64 lanes at a time.  Without this, one crowded bucket is one lane's serial chain: equal or
// structured scalars (bits 135 .. 251 are out of bk_recode's reach) turned a launch of 0.2 ms into one of 2.2 ms with 1/9 of the
// terms in one bucket, and made an MSM of N equal scalars N serial additions.
#define BK_HEAVY_SLACK 32u
BP_HD uint32_t bk_chain_lim(size_t terms_per_msm, bk_params prm) {
    const size_t l = 2 * ((terms_per_msm + prm.half - 1) / prm.half) + BK_HEAVY_SLACK;
    return l > 0x7fffffffu ? 0x7fffffffu : (uint32_t)l;
}
BP_HD uint32_t bk_lane_share(uint32_t cnt, uint32_t lim) { return cnt <= lim ? cnt : lim - BK_HEAVY_SLACK; }
// entries [i0, i0 + n * stride) step stride of descriptor d's list -> acc (identity when n == 0): software pipeline as before -- the
// index of term i+2 and the point record of term i+1 are in flight while term i is added
BP_HD void bk_sum_entries(ge_ext &acc, const bk_desc &d, uint32_t i0, uint32_t stride, uint32_t n, const uint32_t *idx_w, const fb_entry *pts) {
    ge_identity(acc);
    if (n == 0) return;
    const uint32_t *e = idx_w + d.off + i0;
    uint32_t e_cur = e[0];
    uint32_t e_next = n > 1 ? e[stride] : 0u;
    fb_line line_cur;
    fb_load_line(line_cur, pts + (e_cur & 0x7fffffffu));
    for (uint32_t i = 0; i < n; i++) {
        fb_line line_next = line_cur;
        uint32_t e_next2 = 0;
        if (i + 1 < n) fb_load_line(line_next, pts + (e_next & 0x7fffffffu));
        if (i + 2 < n) e_next2 = e[(uint64_t)(i + 2) * stride];
        ge_niels nn;
#pragma unroll
        for (int q = 0; q < 10; q++) {
            nn.ypx.v[q] = line_cur.w[q];
            nn.ymx.v[q] = line_cur.w[10 + q];
            nn.t2d.v[q] = line_cur.w[20 + q];
        }
        const bool neg = (e_cur >> 31) != 0;
        if (i == 0) ge_from_niels(acc, nn, neg);
        else ge_madd(acc, acc, nn, neg);
        line_cur = line_next;
        e_cur = e_next;
        e_next = e_next2;
    }
}
BP_HD void bk_accum_thread(uint32_t bw, uint32_t r, bk_params prm, const bk_desc *desc, const uint32_t *idx_w, const fb_entry *pts, ge_ext *bsum,
                           uint32_t lim = 0x7fffffffu) {
    const bk_desc d = desc[(uint64_t)bw * prm.half + r];
    ge_ext acc;
    bk_sum_entries(acc, d, 0, 1, bk_lane_share(d.cnt, lim), idx_w, pts);
    bsum[(uint64_t)bw * prm.half + d.bucket] = acc;
}

// ---- stage 3b: the heavy pass, G wavefronts per (MSM, window): wavefront g owns the descriptor ranks r = g (mod G) ---------------
// h0: clear the list counter.  h1: the 64 lanes scan the wavefront's ranks and list those with more than `lim` terms (usually
// none: the launch is one read of the descriptors).  Then per listed bucket: h2: lane l sums entries share + l, share + l + 64,
// ... into xch[l]; h3(step = 32 .. 1): xch[l] += xch[l + step]; h4: lane 0 adds xch[0] to the bucket's sum.  A listed bucket costs
// rest / 64 + 7 additions in sequence, and fewer than half / (2 G) buckets can be listed per wavefront.
// G: 16 wavefronts per window for a lone large MSM (c = 12), 2 when the batch already brings many (MSM, window) pairs
BP_HD uint32_t bk_heavy_groups(bk_params prm, size_t nmsm) { return (prm.c == 8 || nmsm >= 8) ? 2u : 16u; }
#define BK_HEAVY_MAX 1024   // >= half / G ranks per wavefront
struct bk_heavy_lds {
    uint32_t *n;      // [1]
    uint32_t *list;   // [BK_HEAVY_MAX] ranks
    ge_ext *xch;      // [64]
};
BP_HD void bk_heavy_h0(uint32_t lane, const bk_heavy_lds &l) {
    if (lane == 0) l.n[0] = 0;
}
BP_HD void bk_heavy_h1(uint32_t lane, uint32_t bw, uint32_t g, uint32_t G, bk_params prm, const bk_desc *desc, uint32_t lim, const bk_heavy_lds &l) {
    for (uint32_t r = g + G * lane; r < prm.half; r += 64 * G) {
        if (desc[(uint64_t)bw * prm.half + r].cnt > lim) {
            const uint32_t pos = BK_ATOMIC_ADD(l.n, 1u);
            if (pos < BK_HEAVY_MAX) l.list[pos] = r;   // (never more: a wavefront owns half / G <= BK_HEAVY_MAX ranks)
        }
    }
}
BP_HD void bk_heavy_h2(uint32_t lane, uint32_t bw, uint32_t i, bk_params prm, const bk_desc *desc, uint32_t lim, const uint32_t *idx_w, const fb_entry *pts,
                       const bk_heavy_lds &l) {
    const bk_desc d = desc[(uint64_t)bw * prm.half + l.list[i]];
    const uint32_t share = bk_lane_share(d.cnt, lim), rest = d.cnt - share;
    ge_ext acc;
    bk_sum_entries(acc, d, share + lane, 64, lane < rest ? (rest - lane + 63) / 64 : 0u, idx_w, pts);
    l.xch[lane] = acc;
}
BP_HD void bk_heavy_h3(uint32_t lane, uint32_t step, const bk_heavy_lds &l) {
    if (lane < step) {
        ge_ext a = l.xch[lane];
        const ge_ext q = l.xch[lane + step];
        ge_add(a, a, q);
        l.xch[lane] = a;
    }
}
BP_HD void bk_heavy_h4(uint32_t lane, uint32_t bw, uint32_t i, bk_params prm, const bk_desc *desc, const bk_heavy_lds &l, ge_ext *bsum) {
    if (lane != 0) return;
    const bk_desc d = desc[(uint64_t)bw * prm.half + l.list[i]];
    ge_ext a = bsum[(uint64_t)bw * prm.half + d.bucket];
    const ge_ext q = l.xch[0];
    ge_add(a, a, q);
    bsum[(uint64_t)bw * prm.half + d.bucket] = a;
}

// ---- stage 4: window sum = sum_j (j + 1) * B_j, j = bucket index, by a tree of running sums ------------------------
// Every node of the tree covers a contiguous range of buckets [j0, j0 + width) and keeps
//     S = sum of its buckets,   A = sum_j (j - j0 + 1) B_j .
// A leaf forms both with one running sum over its m buckets (2 m - 2 additions); a parent of k children of `width`
// buckets each forms S = sum S_i, A = sum A_i + width * sum_i i S_i with one running sum over the children
// (3 k - 2 additions + lg(width) doublings).  The root's A is the window sum.
// Leaves are wide (one lane per leaf of every window of every MSM: bk_leaf_thread); the upper levels are packed so
// that a wavefront serves many windows at once (k_bk_tree).
BP_HD uint32_t bk_leaves(bk_params prm) { return prm.c == 8 ? 8u : 256u; }   // leaves per window: 16 / 8 buckets each
// tid = bw * leaves + l
BP_HD void bk_leaf_thread(uint32_t tid, bk_params prm, const ge_ext *bsum, ge_ext *gS, ge_ext *gA) {
    const uint32_t nl = bk_leaves(prm), bw = tid / nl, l = tid - bw * nl, m = prm.half / nl;
    const ge_ext *b = bsum + (uint64_t)bw * prm.half + (uint64_t)l * m;
    ge_ext run = b[m - 1], acc = run;
    for (uint32_t i = m - 1; i-- > 0;) {
        const ge_ext q = b[i];
        ge_add(run, run, q);
        ge_add(acc, acc, run);
    }
    gS[tid] = run;
    gA[tid] = acc;
}
// parent of the k children at S[first + i * stride], A[first + i * stride]; a child spans `width` buckets (a power of two)
BP_HD void bk_combine(ge_ext &S_out, ge_ext &A_out, const ge_ext *S, const ge_ext *A, uint32_t first, uint32_t k, uint32_t stride, uint32_t width) {
    ge_ext run = S[first + (k - 1) * stride], acc = run, asum = A[first + (k - 1) * stride];
    for (uint32_t i = k - 1; i-- > 1;) {
        const ge_ext s = S[first + i * stride], a = A[first + i * stride];
        ge_add(run, run, s);
        ge_add(acc, acc, run);     // after the loop: acc = sum_{i >= 1} i S_i
        ge_add(asum, asum, a);
    }
    {
        const ge_ext s0 = S[first], a0 = A[first];
        ge_add(run, run, s0);
        ge_add(asum, asum, a0);
    }
    for (uint32_t wd = width; wd > 1; wd >>= 1) ge_dbl(acc, acc, wd == 2);
    ge_add(asum, asum, acc);
    S_out = run;
    A_out = asum;
}

// window sum -> the MSM's radix-16 column sums for the wavefront Horner chain: column (c/4) w carries the window
// sum, the columns up to the next window (if they exist) the identity
BP_HD void bk_emit_columns(uint32_t w, bk_params prm, const ge_ext &sum, uint32_t *colq16_msm /*[64][32 words]*/) {
    const uint32_t per = prm.c / 4, c0 = w * per;
    ge_ext id;
    ge_identity(id);
    for (uint32_t i = 0; i < per && c0 + i < BP_VB_WINDOWS; i++) vb_encode_colq16(colq16_msm + (uint64_t)(c0 + i) * 32, i == 0 ? sum : id);
}

}  // namespace bp
#endif
