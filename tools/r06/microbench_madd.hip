// ge_madd bake-off for gfx950 (round 6).  Not part of libbpgpu.so.
//   M0  ge_madd as shipped up to round 5 (ge25519.h: three carried subtractions)
//   M1  Y-X and B-A formed without a carry chain (fe_sub_rr: f + 2p - g for REDUCED f, g -> a lazy value, <= 3 x reduced)
//   M2  M1 + multiplication with the column carry folded into the next column's multiply-accumulate chain (no 64-bit additions)
// at 1 / 2 / 3 / 4 wavefronts per SIMD; M1 / M2 are compared with M0 on the device (canonical encodings).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/microbench_madd tools/r06/microbench_madd.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../../bulletproofs_amd/csrc/ge25519.h"
using namespace bp;

__device__ __forceinline__ void sub_rr(fe &h, const fe &f, const fe &g) {
    h.v[0] = f.v[0] + 0x7ffffdau - g.v[0];
#pragma unroll
    for (int i = 1; i < 10; i++) h.v[i] = f.v[i] + ((i & 1) ? 0x3fffffeu : 0x7fffffeu) - g.v[i];
}
__device__ __forceinline__ void mad(uint64_t &acc, uint32_t a, uint32_t b) {
    uint64_t sc;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(sc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mul_sc(fe &h, const fe &f, const fe &g) {
    uint32_t g19[10], f2[10];
#pragma unroll
    for (int i = 0; i < 10; i++) { g19[i] = 19u * g.v[i]; f2[i] = 2u * f.v[i]; }
    uint64_t acc = 0;
    uint32_t out[10];
#pragma unroll
    for (int k = 0; k < 10; k++) {
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const int j = (k - i + 10) % 10;
            const uint32_t a = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
            const uint32_t b = (i + j >= 10) ? g19[j] : g.v[j];
            if (k == 0 && i == 0) acc = (uint64_t)a * b;
            else mad(acc, a, b);
        }
        if (k & 1) { out[k] = (uint32_t)acc & BP_M25; acc >>= 25; }
        else { out[k] = (uint32_t)acc & BP_M26; acc >>= 26; }
    }
    const uint64_t t = (uint64_t)out[0] + 19 * acc;
    h.v[0] = (uint32_t)t & BP_M26;
    h.v[1] = out[1] + (uint32_t)(t >> 26);
#pragma unroll
    for (int k = 2; k < 10; k++) h.v[k] = out[k];
}
template <int V> __device__ __forceinline__ void M(fe &h, const fe &a, const fe &b) { if (V == 2) mul_sc(h, a, b); else fe_mul(h, a, b); }
template <int V> __device__ __forceinline__ void madd_v(ge_ext &r, const ge_ext &p, const ge_niels &q, bool neg) {
    fe ypx, ymx, a, b, c, d, qa, qb;
    fe_select(qa, q.ymx, q.ypx, neg);
    fe_select(qb, q.ypx, q.ymx, neg);
    fe_add(ypx, p.Y, p.X);
    if (V == 0) fe_sub(ymx, p.Y, p.X); else sub_rr(ymx, p.Y, p.X);
    M<V>(a, ymx, qa);
    M<V>(b, ypx, qb);
    M<V>(c, p.T, q.t2d);
    fe_add(d, p.Z, p.Z);
    fe e, f, g, h, dmc, dpc;
    if (V == 0) fe_sub(e, b, a); else sub_rr(e, b, a);
    fe_add(h, b, a);
    fe_sub(dmc, d, c);
    fe_add(dpc, d, c);
    fe_select(f, dmc, dpc, neg);
    fe_select(g, dpc, dmc, neg);
    M<V>(r.X, f, e);
    M<V>(r.Y, h, g);
    M<V>(r.Z, f, g);
    M<V>(r.T, h, e);
}
template <int V> __global__ void __launch_bounds__(64) k_bench(const uint32_t *in, uint32_t *out, int iters) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    ge_niels q[2];
    ge_ext acc;
    ge_identity(acc);
    for (int i = 0; i < 10; i++) {
        q[0].ypx.v[i] = in[(t * 60 + i) % 4096] & ((i & 1) ? BP_M25 : BP_M26); q[0].ymx.v[i] = in[(t * 60 + 10 + i) % 4096] & ((i & 1) ? BP_M25 : BP_M26);
        q[0].t2d.v[i] = in[(t * 60 + 20 + i) % 4096] & ((i & 1) ? BP_M25 : BP_M26); q[1].ypx.v[i] = in[(t * 60 + 30 + i) % 4096] & ((i & 1) ? BP_M25 : BP_M26);
        q[1].ymx.v[i] = in[(t * 60 + 40 + i) % 4096] & ((i & 1) ? BP_M25 : BP_M26); q[1].t2d.v[i] = in[(t * 60 + 50 + i) % 4096] & ((i & 1) ? BP_M25 : BP_M26);
    }
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        madd_v<V>(acc, acc, q[0], (it & 4) != 0);
        madd_v<V>(acc, acc, q[1], (it & 2) != 0);
    }
    uint32_t w[8];
    fe_to_words(w, acc.X); for (int i = 0; i < 8; i++) out[t * 32 + i] = w[i];
    fe_to_words(w, acc.Y); for (int i = 0; i < 8; i++) out[t * 32 + 8 + i] = w[i];
    fe_to_words(w, acc.Z); for (int i = 0; i < 8; i++) out[t * 32 + 16 + i] = w[i];
    fe_to_words(w, acc.T); for (int i = 0; i < 8; i++) out[t * 32 + 24 + i] = w[i];
}
template <typename Fn> static double time_ms(Fn f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    uint32_t h_in[4096];
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < 4096; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h_in[i] = (uint32_t)(s >> 11); }
    uint32_t *d_in, *d_o[3];
    const int max_threads = 256 * 4 * 4 * 64;
    hipMalloc(&d_in, sizeof h_in); hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice);
    for (int v = 0; v < 3; v++) hipMalloc(&d_o[v], (size_t)max_threads * 128);
    std::vector<uint32_t> r0(4096 * 32), r1(4096 * 32);
    bool all_ok = true;
#define LAUNCH(V, blocks, iters) hipLaunchKernelGGL((k_bench<V>), dim3(blocks), dim3(64), 0, 0, d_in, d_o[V], iters)
    LAUNCH(0, 64, 7); LAUNCH(1, 64, 7); LAUNCH(2, 64, 7); hipDeviceSynchronize();
    hipMemcpy(r0.data(), d_o[0], 4096 * 128, hipMemcpyDeviceToHost);
    for (int v = 1; v < 3; v++) { hipMemcpy(r1.data(), d_o[v], 4096 * 128, hipMemcpyDeviceToHost); int bad = 0; for (size_t i = 0; i < r0.size(); i++) bad += r0[i] != r1[i];
        printf("check M%d vs M0: %s\n", v, bad ? "MISMATCH" : "identical"); all_ok = all_ok && !bad; }
    const int iters = 200;
    for (int wps = 1; wps <= 4; wps++) {
        const int blocks = 256 * 4 * wps;
#define RUN(V) { double ms = time_ms([&] { LAUNCH(V, blocks, iters); }); double ops = (double)blocks * 64 * iters * 2; \
        printf("madd M%d  waves/SIMD=%d  %8.3f ms  %10.3e madds/s  %7.1f cycles per wave-madd per SIMD\n", V, wps, ms, ops / (ms * 1e-3), 2.4e9 * 1024.0 * 64 / (ops / (ms * 1e-3))); }
        RUN(0) RUN(1) RUN(2)
    }
    printf("%s\n", all_ok ? "all variants agree with M0" : "SOME VARIANT DISAGREES");
    return all_ok ? 0 : 1;
}
