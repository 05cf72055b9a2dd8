// Wave-vector abstraction: code written against wu32 / wu64 / wbool and the wv_* cross-lane
// primitives compiles (a) for the GPU, where the types are plain per-lane scalars and the primitives
// are DPP / ds_bpermute instructions of a 64-lane wavefront, and (b) for the host, where the types are
// arrays of 64 lanes executed in lockstep.  This keeps wavefront-cooperative kernels (horner_wave.h)
// testable on the CPU, lane for lane.
#ifndef BPGPU_WAVEVEC_H
#define BPGPU_WAVEVEC_H
#include <stdint.h>

#if defined(__HIP_DEVICE_COMPILE__)
namespace bp {
typedef uint32_t wu32;
typedef uint64_t wu64;
typedef bool wbool;
#define WV_FN __device__ __forceinline__
WV_FN wu32 wv_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
WV_FN wu32 wv_splat(uint32_t x) { return x; }
// lane i of each 16-lane row receives the value of lane (i - S) mod 16 of that row
template <int S>
WV_FN wu32 wv_row_ror(wu32 x) {
    if (S == 0) return x;
    return (wu32)__builtin_amdgcn_update_dpp(0, (int)x, 0x120 + (S & 15), 0xf, 0xf, false);
}
// lane i of each row receives lane i - S of that row, or 0 when i < S      (DPP row_shr, bound_ctrl: zero fill)
template <int S>
WV_FN wu32 wv_row_shr0(wu32 x) {
    if (S == 0) return x;
    if (S >= 16) return 0;
    return (wu32)__builtin_amdgcn_update_dpp(0, (int)x, 0x110 + (S & 15), 0xf, 0xf, true);
}
// lane i of each row receives lane i + S of that row, or 0 when i + S > 15  (DPP row_shl, zero fill)
template <int S>
WV_FN wu32 wv_row_shl0(wu32 x) {
    if (S == 0) return x;
    if (S >= 16) return 0;
    return (wu32)__builtin_amdgcn_update_dpp(0, (int)x, 0x100 + (S & 15), 0xf, 0xf, true);
}
// lane i receives x from lane src[i] (any lane of the wavefront)
WV_FN wu32 wv_bperm(wu32 x, wu32 src) { return (wu32)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)x); }
WV_FN wu32 wv_select(wbool c, wu32 a, wu32 b) { return c ? a : b; }
WV_FN wu64 wv_mad64(wu64 acc, wu32 a, wu32 b) { return acc + (uint64_t)a * b; }
WV_FN wu64 wv_widen(wu32 a) { return (uint64_t)a; }
WV_FN wu32 wv_lo32(wu64 a) { return (uint32_t)a; }
WV_FN wu32 wv_hi32(wu64 a) { return (uint32_t)(a >> 32); }
WV_FN wu32 wv_load_u16(const uint16_t *base, wu32 idx) { return base[idx]; }
// Scratch for the two wavefront-wide exchanges below: 128 words of LDS owned by ONE wavefront
// (workgroup = 64 lanes).  A wavefront issues its DS instructions in order and the LDS executes them in
// order, so a ds_read that follows a ds_write of the same wavefront observes it -- no s_barrier needed;
// WV_LDS_ORDER only stops the compiler from moving the accesses across each other.
struct wv_ctx {
    uint32_t *lds;   // 128 x u32, 16-byte aligned
};
#define WV_LDS_ORDER()                       \
    do {                                     \
        __builtin_amdgcn_wave_barrier();     \
        asm volatile("" ::: "memory");       \
    } while (0)
// out[i] = value of lane (row base + i) of this lane's 16-lane row, i = 0..15  (1 store + 4 ds_read_b128)
WV_FN void wv_row_gather16(const wv_ctx &cx, wu32 x, wu32 out[16]) {
    const uint32_t lane = wv_lane();
    WV_LDS_ORDER();
    cx.lds[lane] = x;
    WV_LDS_ORDER();
    const uint4 *p = (const uint4 *)(cx.lds + (lane & 48u));
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint4 q = p[j];
        out[4 * j] = q.x;
        out[4 * j + 1] = q.y;
        out[4 * j + 2] = q.z;
        out[4 * j + 3] = q.w;
    }
}
// out[r] = value of lane (16 r + k) where k = this lane's position in its row, r = 0..3  (1 store + 1 ds_read_b128)
WV_FN void wv_rows4(const wv_ctx &cx, wu32 x, wu32 out[4]) {
    const uint32_t lane = wv_lane();
    WV_LDS_ORDER();
    cx.lds[64 + ((lane & 15u) << 2) + (lane >> 4)] = x;
    WV_LDS_ORDER();
    const uint4 q = *(const uint4 *)(cx.lds + 64 + ((lane & 15u) << 2));
    out[0] = q.x;
    out[1] = q.y;
    out[2] = q.z;
    out[3] = q.w;
}
}  // namespace bp
#else
namespace bp {
#define WV_FN inline
#define WV_N 64
struct wbool {
    bool l[WV_N];
};
struct wu32 {
    uint32_t l[WV_N];
};
struct wu64 {
    uint64_t l[WV_N];
};
#define WV_BINOP(T, op)                                                                       \
    inline T operator op(const T &a, const T &b) { T r; for (int i = 0; i < WV_N; i++) r.l[i] = a.l[i] op b.l[i]; return r; } \
    inline T operator op(const T &a, uint32_t b) { T r; for (int i = 0; i < WV_N; i++) r.l[i] = a.l[i] op b; return r; }
WV_BINOP(wu32, +) WV_BINOP(wu32, -) WV_BINOP(wu32, *) WV_BINOP(wu32, &) WV_BINOP(wu32, |) WV_BINOP(wu32, ^) WV_BINOP(wu32, >>) WV_BINOP(wu32, <<)
WV_BINOP(wu64, +) WV_BINOP(wu64, >>) WV_BINOP(wu64, &)
#define WV_CMP(op)                                                                            \
    inline wbool operator op(const wu32 &a, const wu32 &b) { wbool r; for (int i = 0; i < WV_N; i++) r.l[i] = a.l[i] op b.l[i]; return r; } \
    inline wbool operator op(const wu32 &a, uint32_t b) { wbool r; for (int i = 0; i < WV_N; i++) r.l[i] = a.l[i] op b; return r; }
WV_CMP(<) WV_CMP(>=) WV_CMP(==) WV_CMP(!=)
inline wbool operator||(const wbool &a, const wbool &b) { wbool r; for (int i = 0; i < WV_N; i++) r.l[i] = a.l[i] || b.l[i]; return r; }
inline wbool operator&&(const wbool &a, const wbool &b) { wbool r; for (int i = 0; i < WV_N; i++) r.l[i] = a.l[i] && b.l[i]; return r; }
WV_FN wu32 wv_lane() { wu32 r; for (int i = 0; i < WV_N; i++) r.l[i] = (uint32_t)i; return r; }
WV_FN wu32 wv_splat(uint32_t x) { wu32 r; for (int i = 0; i < WV_N; i++) r.l[i] = x; return r; }
template <int S>
WV_FN wu32 wv_row_ror(const wu32 &x) {
    wu32 r;
    for (int i = 0; i < WV_N; i++) r.l[i] = x.l[(i & ~15) | ((i - S) & 15)];
    return r;
}
template <int S>
WV_FN wu32 wv_row_shr0(const wu32 &x) {
    wu32 r;
    for (int i = 0; i < WV_N; i++) r.l[i] = ((i & 15) - S >= 0) ? x.l[i - S] : 0u;
    return r;
}
template <int S>
WV_FN wu32 wv_row_shl0(const wu32 &x) {
    wu32 r;
    for (int i = 0; i < WV_N; i++) r.l[i] = ((i & 15) + S <= 15) ? x.l[i + S] : 0u;
    return r;
}
WV_FN wu32 wv_bperm(const wu32 &x, const wu32 &src) { wu32 r; for (int i = 0; i < WV_N; i++) r.l[i] = x.l[src.l[i] & 63]; return r; }
WV_FN wu32 wv_select(const wbool &c, const wu32 &a, const wu32 &b) { wu32 r; for (int i = 0; i < WV_N; i++) r.l[i] = c.l[i] ? a.l[i] : b.l[i]; return r; }
WV_FN wu64 wv_mad64(const wu64 &acc, const wu32 &a, const wu32 &b) { wu64 r; for (int i = 0; i < WV_N; i++) r.l[i] = acc.l[i] + (uint64_t)a.l[i] * b.l[i]; return r; }
WV_FN wu64 wv_widen(const wu32 &a) { wu64 r; for (int i = 0; i < WV_N; i++) r.l[i] = a.l[i]; return r; }
WV_FN wu32 wv_lo32(const wu64 &a) { wu32 r; for (int i = 0; i < WV_N; i++) r.l[i] = (uint32_t)a.l[i]; return r; }
WV_FN wu32 wv_hi32(const wu64 &a) { wu32 r; for (int i = 0; i < WV_N; i++) r.l[i] = (uint32_t)(a.l[i] >> 32); return r; }
WV_FN wu32 wv_load_u16(const uint16_t *base, const wu32 &idx) { wu32 r; for (int i = 0; i < WV_N; i++) r.l[i] = base[idx.l[i]]; return r; }
struct wv_ctx {
    int unused;
};
WV_FN void wv_row_gather16(const wv_ctx &, const wu32 &x, wu32 out[16]) {
    for (int j = 0; j < 16; j++)
        for (int i = 0; i < WV_N; i++) out[j].l[i] = x.l[(i & 48) + j];
}
WV_FN void wv_rows4(const wv_ctx &, const wu32 &x, wu32 out[4]) {
    for (int r = 0; r < 4; r++)
        for (int i = 0; i < WV_N; i++) out[r].l[i] = x.l[16 * r + (i & 15)];
}
}  // namespace bp
#endif
#endif
