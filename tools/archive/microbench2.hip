// Dev microbenchmark: why does fb_accum's per-iteration cost exceed the bare ge_madd loop?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../bulletproofs_amd/csrc/msm_fixed.h"
using namespace bp;

template <int V> __global__ void __launch_bounds__(64) k_var(uint32_t *out, const fb_entry *table, const uint16_t *digits, uint32_t iters, uint32_t nwin, uint32_t half) {
    const uint32_t lane = blockIdx.x * 64 + threadIdx.x;
    ge_ext acc; ge_identity(acc);
    fe x; for (int i = 0; i < 10; i++) x.v[i] = (lane * 2654435761u + i * 40503u) & 0x1ffffff;
    acc.X = x; acc.T = x;
    ge_niels n0; n0.ypx = x; n0.ymx = x; n0.t2d = x;
    for (uint32_t it = 0; it < iters; it++) {
        if (V == 0) {                       // constant operand
            ge_madd(acc, acc, n0, it & 1);
        } else if (V == 1) {                // operand loaded from memory, same line for all lanes/iterations
            const fb_entry *e = table + (it & 7);
            ge_niels n; n.ypx = e->ypx; n.ymx = e->ymx; n.t2d = e->t2d;
            ge_madd(acc, acc, n, it & 1);
        } else if (V == 2) {                // + digit load, sign from digit
            const uint32_t v = digits[(uint64_t)it * gridDim.x * 64 + lane];
            const int d = (int)(v & (2 * half - 1)) - (int)half;
            const fb_entry *e = table + (it & 7);
            ge_niels n; n.ypx = e->ypx; n.ymx = e->ymx; n.t2d = e->t2d;
            if (d != 0) ge_madd(acc, acc, n, d < 0);
        } else if (V == 3) {                // + real address computation (division), real gather
            const uint32_t v = digits[(uint64_t)it * gridDim.x * 64 + lane];
            const int d = (int)(v & (2 * half - 1)) - (int)half;
            const uint32_t g = it / nwin, win = it - g * nwin;
            const uint32_t a = (uint32_t)(d < 0 ? -d : d);
            const fb_entry *e = table + ((uint64_t)g * nwin + win) * half + (a ? a - 1 : 0);
            ge_niels n; n.ypx = e->ypx; n.ymx = e->ymx; n.t2d = e->t2d;
            if (d != 0) ge_madd(acc, acc, n, d < 0);
        }
    }
    uint32_t r = 0; for (int i = 0; i < 10; i++) r ^= acc.X.v[i] ^ acc.Y.v[i] ^ acc.Z.v[i] ^ acc.T.v[i];
    out[lane] = r;
}

int main() {
    const uint32_t blocks = 4096, iters = 520, nwin = 32, half = 128;
    uint32_t *d_out; fb_entry *d_tab; uint16_t *d_dig;
    const size_t ntab = (size_t)(iters / nwin + 1) * nwin * half;
    hipMalloc(&d_out, blocks * 64 * 4); hipMalloc(&d_tab, ntab * sizeof(fb_entry)); hipMalloc(&d_dig, (size_t)iters * blocks * 64 * 2);
    hipMemset(d_tab, 1, ntab * sizeof(fb_entry));
    uint16_t *h = (uint16_t *)malloc((size_t)iters * blocks * 64 * 2); uint32_t s = 12345;
    for (size_t i = 0; i < (size_t)iters * blocks * 64; i++) { s = s * 1664525u + 1013904223u; h[i] = (s >> 16) & 255; }
    hipMemcpy(d_dig, h, (size_t)iters * blocks * 64 * 2, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
#define RUN(V, B) { hipLaunchKernelGGL(k_var<V>, dim3(B), dim3(64), 0, 0, d_out, d_tab, d_dig, iters, nwin, half); hipDeviceSynchronize(); \
    hipEventRecord(a); hipLaunchKernelGGL(k_var<V>, dim3(B), dim3(64), 0, 0, d_out, d_tab, d_dig, iters, nwin, half); hipEventRecord(b); hipEventSynchronize(b); \
    float ms; hipEventElapsedTime(&ms, a, b); printf("variant %d blocks %4d: %.3f ms  -> %.3e madd/s\n", V, B, ms, (double)B * 64 * iters / (ms * 1e-3)); }
    RUN(0, 1024) RUN(0, 2048) RUN(0, 4096) RUN(1, 2048) RUN(2, 2048) RUN(3, 2048) RUN(3, 1024) RUN(3, 3072)
    return 0;
}
