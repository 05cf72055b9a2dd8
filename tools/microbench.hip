// Instruction-issue microbenchmarks for gfx950: how many lanes*ops/s the VALU sustains for the
// integer instructions a 256-bit field multiplication can be built from.  Not part of libbpgpu.so.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench tools/microbench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../bulletproofs_amd/csrc/ge25519.h"
typedef unsigned short __attribute__((ext_vector_type(2))) us2;

#define ITERS 4096
#define CHAINS 8
template <int OP> __global__ void __launch_bounds__(256) k_op(uint32_t *out, uint32_t seed) {
    uint32_t a[CHAINS], b = seed | 1, c2 = threadIdx.x * 2654435761u + seed;
    uint64_t w[CHAINS];
    for (int i = 0; i < CHAINS; i++) { a[i] = threadIdx.x + i * 77 + seed; w[i] = a[i]; }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            if (OP == 0) w[i] = (uint64_t)(uint32_t)w[i] * b + w[i];                      // v_mad_u64_u32
            if (OP == 1) a[i] = a[i] * b + c2;                                             // v_mul_lo_u32 (+add)
            if (OP == 2) a[i] = __umulhi(a[i], b) + c2;                                    // v_mul_hi_u32
            if (OP == 3) a[i] = __builtin_amdgcn_udot2(*(us2 *)&a[i], *(us2 *)&b, a[i], false);  // v_dot2_u32_u16
            if (OP == 4) a[i] = __builtin_amdgcn_udot4(a[i], b, a[i], false);              // v_dot4_u32_u8
            if (OP == 5) a[i] = (a[i] & 0xffffff) * (b & 0xffffff) + c2;                   // v_mad_u32_u24
            if (OP == 6) a[i] = (a[i] << 3) + c2;                                          // v_lshl_add_u32
            if (OP == 7) a[i] = a[i] + c2;                                                 // v_add_u32
            if (OP == 8) w[i] = (w[i] >> 7) + c2;                                          // 64-bit shift + add
            if (OP == 9) { double d = __longlong_as_double(w[i] | 0x3ff0000000000000ull); d = fma(d, 1.0000001, 0.5); w[i] = __double_as_longlong(d); } // v_fma_f64
            if (OP == 10) { float f = __uint_as_float(a[i] | 0x3f800000u); f = fmaf(f, 1.0001f, 0.5f); a[i] = __float_as_uint(f); }  // v_fma_f32
            if (OP == 11) a[i] = __builtin_amdgcn_alignbit(a[i], b, 7) ^ c2;               // v_alignbit
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < CHAINS; i++) r ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// field-op throughput: FEOP 0 = fe_mul, 1 = fe_sq, 2 = ge_madd, 3 = ge_dbl, 4 = ge_add_cached
template <int FEOP> __global__ void __launch_bounds__(256) k_fe(uint32_t *out, uint32_t seed, int iters) {
    using namespace bp;
    fe x, y; 
    for (int i = 0; i < 10; i++) { x.v[i] = (threadIdx.x * 2654435761u + i * 40503u + seed) & 0x1ffffff; y.v[i] = (threadIdx.x * 97u + i * 7919u + seed) & 0x1ffffff; }
    ge_ext p; p.X = x; p.Y = y; fe_1(p.Z); fe_mul(p.T, x, y);
    ge_niels n; n.ypx = y; n.ymx = x; n.t2d = p.T;
    ge_cached c; c.YpX = y; c.YmX = x; c.Z = y; c.T2d = p.T;
    for (int it = 0; it < iters; it++) {
        if (FEOP == 0) { fe_mul(x, x, y); }
        if (FEOP == 1) { fe_sq(x, x); }
        if (FEOP == 2) { ge_madd(p, p, n, (it & 1)); }
        if (FEOP == 3) { ge_dbl(p, p, true); }
        if (FEOP == 4) { ge_add_cached(p, p, c, (it & 1)); }
    }
    uint32_t r = 0;
    for (int i = 0; i < 10; i++) r ^= x.v[i] ^ p.X.v[i] ^ p.Y.v[i] ^ p.Z.v[i] ^ p.T.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <typename F> static double time_ms(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("device: %s CUs=%d clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
    const char *names[] = {"v_mad_u64_u32", "v_mul_lo_u32+add", "v_mul_hi_u32+add", "v_dot2_u32_u16", "v_dot4_u32_u8", "v_mad_u32_u24(+and)", "v_lshl_add_u32", "v_add_u32", "shr64+add64", "v_fma_f64", "v_fma_f32", "v_alignbit+xor"};
    const int blocks = 256 * 8;   // 8 blocks of 4 waves per CU -> 8 waves/SIMD
#define RUN(OP) { double ms = time_ms([&] { hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u); }); \
    double ops = (double)blocks * 256 * ITERS * CHAINS; printf("%-22s %8.3f ms  %10.3e lane-ops/s  (%.2f cyc/wave-instr/SIMD @2.4GHz)\n", names[OP], ms, ops / (ms * 1e-3), 2.4e9 * 1024.0 * 64 / (ops / (ms * 1e-3))); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11)
    const char *fn[] = {"fe_mul", "fe_sq", "ge_madd", "ge_dbl", "ge_add_cached"};
    for (int wpb = 1; wpb <= 4; wpb *= 2) {
        const int fblocks = 256 * 4 * wpb;   // wpb*4 blocks of 64 -> waves per CU = 4*wpb... (1,2,4 waves/SIMD)
        const int iters = 2000;
#define RUNF(OP) { double ms = time_ms([&] { hipLaunchKernelGGL(k_fe<OP>, dim3(fblocks), dim3(64), 0, 0, d, 777u, iters); }); \
        double ops = (double)fblocks * 64 * iters; printf("%-14s waves/SIMD=%d %8.3f ms  %10.3e ops/s\n", fn[OP], wpb, ms, ops / (ms * 1e-3)); }
        RUNF(0) RUNF(1) RUNF(2) RUNF(3) RUNF(4)
    }
    hipFree(d);
    return 0;
}
