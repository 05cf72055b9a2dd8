// Field-multiplication bake-off for gfx950.  Not part of libbpgpu.so.
//   V0  the engine's form: 10 limbs of 25.5 bits, 100 v_mad_u64_u32 into ten 64-bit columns (fe25519.h)
//   V1  same limbs, even/odd-limb Karatsuba: f = E + 2^26 O in y = 2^51, three 5x5 products mod (y^5 - 19) = 75 + 1 v_mad_u64_u32,
//       f g = EE + 2 y OO + 2^26 ((E+O)(E+O) - EE - OO).  Needs inputs <= 2^26 / 2^25 (+ carries): 19 (g_e + g_o) must fit 32 bits,
//       so the engine's lazy additions would have to be carried first (the madd-shaped kernel below pays for that)
//   V2  8 limbs of 32 bits, saturated: 64 multiply-accumulates into a 96-bit column accumulator (v_mad_u64_u32 with its carry-out
//       into VCC + v_addc_co_u32), product scanning, then 2^256 = 38: 8 more multiply-accumulates and a carry chain
// For each: fe_mul, fe_sq and a ge_madd-shaped body (7 multiplications, 4 additions, 4 subtractions wired like ge_madd) at 1 / 2 / 3 / 4
// wavefronts per SIMD; every variant's results are compared with V0's on the device (canonical encodings).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o microbench_fe tools/microbench_fe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../bulletproofs_amd/csrc/fe25519.h"
using namespace bp;

// ---------------------------------------------------------------- V1: Karatsuba on the 25.5-bit limbs
__device__ __forceinline__ void k5(uint64_t P[5], const uint32_t A[5], const uint32_t B[5]) {   // A B mod (y^5 - 19)
    uint32_t B19[5];
#pragma unroll
    for (int j = 1; j < 5; j++) B19[j] = 19u * B[j];
#pragma unroll
    for (int k = 0; k < 5; k++) P[k] = 0;
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const int k = i + j;
            P[k % 5] += (uint64_t)A[i] * (k >= 5 ? B19[j] : B[j]);
        }
}
__device__ __forceinline__ void fe_mul_k75(fe &h, const fe &f, const fe &g) {
    uint32_t Ef[5], Of[5], Eg[5], Og[5], Sf[5], Sg[5];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        Ef[i] = f.v[2 * i]; Of[i] = f.v[2 * i + 1]; Eg[i] = g.v[2 * i]; Og[i] = g.v[2 * i + 1];
        Sf[i] = Ef[i] + Of[i]; Sg[i] = Eg[i] + Og[i];
    }
    uint64_t EE[5], OO[5], SS[5], c[10];
    k5(EE, Ef, Eg);
    k5(OO, Of, Og);
    k5(SS, Sf, Sg);
    // even limbs: EE[k] + 2 (y OO)[k], (y OO)[0] = 19 OO[4]
    c[0] = EE[0] + 38ull * OO[4];
#pragma unroll
    for (int k = 1; k < 5; k++) c[2 * k] = EE[k] + 2 * OO[k - 1];
#pragma unroll
    for (int k = 0; k < 5; k++) c[2 * k + 1] = SS[k] - EE[k] - OO[k];
    fe_reduce_columns(h, c);
}

// ---------------------------------------------------------------- V2: 8 x 32 saturated
struct fs { uint32_t w[8]; };   // value < 2^256, congruent mod p
__device__ __forceinline__ void mac(uint64_t &lo, uint32_t &hi, uint32_t a, uint32_t b) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void fs_reduce16(fs &r, const uint32_t t[16]) {
    // lo + 38 hi: column k = t[k] + 38 t[8 + k] + carry
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        acc += (uint64_t)t[k] + (uint64_t)t[8 + k] * 38u;   // < 2^32 + 38 2^32 + carry: fits
        r.w[k] = (uint32_t)acc;
        acc >>= 32;
    }
    // acc < 39: fold once more, 2^256 = 38
    uint64_t c = (uint64_t)r.w[0] + acc * 38u;
    r.w[0] = (uint32_t)c;
    c >>= 32;
#pragma unroll
    for (int k = 1; k < 8; k++) {
        c += r.w[k];
        r.w[k] = (uint32_t)c;
        c >>= 32;
    }
    r.w[0] += 38u * (uint32_t)c;   // (a second wrap needs the value within 2^11 of 2^256; the sum then cannot carry again)
}
__device__ __forceinline__ void fs_mul(fs &r, const fs &a, const fs &b) {
    uint32_t t[16];
    uint64_t lo = 0;
    uint32_t hi = 0;
#pragma unroll
    for (int k = 0; k < 15; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int j = k - i;
            if (j >= 0 && j < 8) mac(lo, hi, a.w[i], b.w[j]);
        }
        t[k] = (uint32_t)lo;
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
    t[15] = (uint32_t)lo;
    fs_reduce16(r, t);
}
__device__ __forceinline__ void fs_sq(fs &r, const fs &a) {
    // off-diagonal products once (28), doubled by a 1-bit shift of the 16-word sum, then the 8 squares
    uint32_t t[16];
    uint64_t lo = 0;
    uint32_t hi = 0;
    t[0] = 0;
#pragma unroll
    for (int k = 1; k < 14; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int j = k - i;
            if (j > i && j < 8) mac(lo, hi, a.w[i], a.w[j]);
        }
        t[k] = (uint32_t)lo;
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
    t[14] = (uint32_t)lo;
    t[15] = (uint32_t)(lo >> 32);
#pragma unroll
    for (int k = 15; k > 0; k--) t[k] = (t[k] << 1) | (t[k - 1] >> 31);
    t[0] <<= 1;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t s = (uint64_t)a.w[i] * a.w[i];
        c += (uint64_t)t[2 * i] + (uint32_t)s;
        t[2 * i] = (uint32_t)c;
        c >>= 32;
        c += (uint64_t)t[2 * i + 1] + (uint32_t)(s >> 32);
        t[2 * i + 1] = (uint32_t)c;
        c >>= 32;
    }
    fs_reduce16(r, t);
}
__device__ __forceinline__ void fs_add(fs &r, const fs &a, const fs &b) {
    uint64_t c = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        c += (uint64_t)a.w[k] + b.w[k];
        r.w[k] = (uint32_t)c;
        c >>= 32;
    }
    uint64_t d = (uint64_t)r.w[0] + 38u * (uint32_t)c;
    r.w[0] = (uint32_t)d;
    d >>= 32;
#pragma unroll
    for (int k = 1; k < 8; k++) {
        d += r.w[k];
        r.w[k] = (uint32_t)d;
        d >>= 32;
    }
    r.w[0] += 38u * (uint32_t)d;
}
__device__ __forceinline__ void fs_sub(fs &r, const fs &a, const fs &b) {
    int64_t c = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        c += (int64_t)a.w[k] - (int64_t)b.w[k];
        r.w[k] = (uint32_t)c;
        c >>= 32;
    }
    // borrow: the true value is r - 2^256 = r - 38 (mod p)
    int64_t d = (int64_t)r.w[0] - (int64_t)(38u & (uint32_t)-(int32_t)(c != 0));
    r.w[0] = (uint32_t)d;
    d >>= 32;
#pragma unroll
    for (int k = 1; k < 8; k++) {
        d += r.w[k];
        r.w[k] = (uint32_t)d;
        d >>= 32;
    }
    r.w[0] -= 38u & (uint32_t)-(int32_t)(d != 0);
}
__device__ __forceinline__ void fs_to_fe(fe &h, const fs &a) {
    uint32_t w[8];
    for (int i = 0; i < 8; i++) w[i] = a.w[i];
    const uint32_t top = w[7] >> 31;
    w[7] &= 0x7fffffffu;
    fe_from_words(h, w);
    h.v[0] += 19u * top;
}
__device__ __forceinline__ void fe_to_fs(fs &a, const fe &h) { fe_to_words(a.w, h); }

// ---------------------------------------------------------------- kernels
// the shape of ge_madd (ge25519.h): 7 multiplications, 4 additions, 4 subtractions; V selects the arithmetic
template <int V> struct F;
template <> struct F<0> { typedef fe T; static __device__ __forceinline__ void mul(T &h, const T &a, const T &b) { fe_mul(h, a, b); }
    static __device__ __forceinline__ void sq(T &h, const T &a) { fe_sq(h, a); } static __device__ __forceinline__ void add(T &h, const T &a, const T &b) { fe_add(h, a, b); }
    static __device__ __forceinline__ void sub(T &h, const T &a, const T &b) { fe_sub(h, a, b); } };
template <> struct F<1> { typedef fe T; static __device__ __forceinline__ void mul(T &h, const T &a, const T &b) { fe_mul_k75(h, a, b); }
    static __device__ __forceinline__ void sq(T &h, const T &a) { fe_mul_k75(h, a, a); }
    static __device__ __forceinline__ void add(T &h, const T &a, const T &b) { fe_add(h, a, b); fe_carry(h); }   // K75 cannot take lazy sums
    static __device__ __forceinline__ void sub(T &h, const T &a, const T &b) { fe_sub(h, a, b); } };
// V3 = V1 with the discipline the engine could adopt: the FIRST operand may be lazy (3x reduced), the SECOND at most one lazy sum of two
// reduced values (19 (g_e + g_o) < 2^32); fe_sq stays the engine's own 55-product squaring.  In the madd shape only Y3 = (D + C)(B + A) has
// two lazy operands: ONE carry.
template <> struct F<3> { typedef fe T; static __device__ __forceinline__ void mul(T &h, const T &a, const T &b) { fe_mul_k75(h, a, b); }
    static __device__ __forceinline__ void sq(T &h, const T &a) { fe_sq(h, a); }
    static __device__ __forceinline__ void add(T &h, const T &a, const T &b) { fe_add(h, a, b); }
    static __device__ __forceinline__ void sub(T &h, const T &a, const T &b) { fe_sub(h, a, b); } };
template <> struct F<2> { typedef fs T; static __device__ __forceinline__ void mul(T &h, const T &a, const T &b) { fs_mul(h, a, b); }
    static __device__ __forceinline__ void sq(T &h, const T &a) { fs_sq(h, a); } static __device__ __forceinline__ void add(T &h, const T &a, const T &b) { fs_add(h, a, b); }
    static __device__ __forceinline__ void sub(T &h, const T &a, const T &b) { fs_sub(h, a, b); } };

template <int V> __device__ __forceinline__ void madd_shape(typename F<V>::T &X, typename F<V>::T &Y, typename F<V>::T &Z, typename F<V>::T &Tt,
                                                            const typename F<V>::T &ypx, const typename F<V>::T &ymx, const typename F<V>::T &t2d) {
    typedef F<V> A;
    typename A::T a, b, c, d, e, f, g, h;
    A::sub(a, Y, X); A::mul(a, a, ymx);
    A::add(b, Y, X); A::mul(b, b, ypx);
    A::mul(c, Tt, t2d);
    A::add(d, Z, Z);
    A::sub(e, b, a); A::sub(f, d, c); A::add(g, d, c); A::add(h, b, a);
    if (V == 3) {   // second operands: f, e (carried by the subtraction), g (one lazy sum); h x g would have two lazy sums: carry h once
        A::mul(X, e, f); A::mul(Z, g, f); A::mul(Tt, h, e);
        fe_carry(*(fe *)&h);
        A::mul(Y, g, h);
    } else {
        A::mul(X, e, f); A::mul(Y, g, h); A::mul(Z, f, g); A::mul(Tt, e, h);
    }
}

template <int V, int OP> __global__ void __launch_bounds__(64) k_bench(const uint32_t *in, uint32_t *out, int iters) {
    typedef F<V> A;
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    fe x0, y0, z0;
    for (int i = 0; i < 10; i++) { x0.v[i] = in[(t * 30 + i) % 4096] & ((i & 1) ? BP_M25 : BP_M26); y0.v[i] = in[(t * 30 + 10 + i) % 4096] & ((i & 1) ? BP_M25 : BP_M26); z0.v[i] = in[(t * 30 + 20 + i) % 4096] & ((i & 1) ? BP_M25 : BP_M26); }
    typename A::T x, y, z, w;
    if constexpr (V == 2) { fe_to_fs(x, x0); fe_to_fs(y, y0); fe_to_fs(z, z0); }
    else { x = x0; y = y0; z = z0; }
    w = x;
    for (int it = 0; it < iters; it++) {
        if (OP == 0) A::mul(x, x, y);
        if (OP == 1) A::sq(x, x);
        if (OP == 2) madd_shape<V>(x, w, z, y, y, z, x);   // (operands reused as table entry: only the shape matters)
    }
    fe r;
    if constexpr (V == 2) fs_to_fe(r, x); else r = x;
    uint32_t wds[8];
    fe_to_words(wds, r);
    for (int i = 0; i < 8; i++) out[t * 8 + i] = wds[i];
}

template <typename Fn> static double time_ms(Fn f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("device: %s CUs=%d\n", prop.name, prop.multiProcessorCount);
    uint32_t h_in[4096];
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < 4096; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h_in[i] = (uint32_t)(s >> 11); }
    uint32_t *d_in, *d_o[4];
    const int max_threads = 256 * 4 * 4 * 64;
    hipMalloc(&d_in, sizeof h_in); hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice);
    for (int v = 0; v < 4; v++) hipMalloc(&d_o[v], (size_t)max_threads * 32);
    // ---- correctness: each variant against V0, few iterations of every op
    const char *vn[4] = {"V0 10x25.5 schoolbook (100 mad)", "V1 10x25.5 Karatsuba (76 mad)", "V2 8x32 saturated (64+8 mad+addc)", "V3 Karatsuba, engine discipline"};
    const char *on[3] = {"fe_mul", "fe_sq", "madd-shaped"};
    std::vector<uint32_t> r0(4096 * 8), r1(4096 * 8);
    bool all_ok = true;
#define LAUNCH(V, OP, blocks, iters) hipLaunchKernelGGL((k_bench<V, OP>), dim3(blocks), dim3(64), 0, 0, d_in, d_o[V & 3], iters)
#define CHECK(OP) { LAUNCH(0, OP, 64, 5); LAUNCH(1, OP, 64, 5); LAUNCH(2, OP, 64, 5); LAUNCH(3, OP, 64, 5); hipDeviceSynchronize(); \
        hipMemcpy(r0.data(), d_o[0], 4096 * 32, hipMemcpyDeviceToHost); \
        for (int v = 1; v < 4; v++) { hipMemcpy(r1.data(), d_o[v], 4096 * 32, hipMemcpyDeviceToHost); int bad = 0; for (size_t i = 0; i < r0.size(); i++) bad += r0[i] != r1[i]; \
            printf("check %-12s %s vs V0: %s\n", on[OP], vn[v], bad ? "MISMATCH" : "identical"); all_ok = all_ok && !bad; } }
    CHECK(0) CHECK(1) CHECK(2)
    // ---- throughput
    const int iters[3] = {2000, 2000, 300};
    for (int wps = 1; wps <= 4; wps++) {
        const int blocks = 256 * 4 * wps;   // one wavefront per block: wps blocks per SIMD
#define RUN(V, OP) { double ms = time_ms([&] { LAUNCH(V, OP, blocks, iters[OP]); }); double ops = (double)blocks * 64 * iters[OP]; \
        printf("%-12s waves/SIMD=%d  %-36s %8.3f ms  %10.3e ops/s  %7.1f cycles per wave-op per SIMD\n", on[OP], wps, vn[V], ms, ops / (ms * 1e-3), 2.4e9 * 1024.0 * 64 / (ops / (ms * 1e-3))); }
        RUN(0, 0) RUN(1, 0) RUN(2, 0) RUN(0, 1) RUN(1, 1) RUN(2, 1) RUN(0, 2) RUN(1, 2) RUN(2, 2) RUN(3, 2)
    }
    printf("%s\n", all_ok ? "all variants agree with V0" : "SOME VARIANT DISAGREES");
    return all_ok ? 0 : 1;
}
