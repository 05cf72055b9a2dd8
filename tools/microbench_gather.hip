// Random-gather ceiling of the HBM: every lane reads LINE consecutive bytes at a pseudo-random line of a large buffer (the access
// pattern of the window-table walk: one 128-byte table line per mixed addition, no reuse).  Prints GB/s for 64- and 128-byte
// lines, buffers of 1 / 16 / 64 GiB, with `inflight` independent loads per lane in flight.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench_gather tools/microbench_gather.hip && tools/microbench_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int WORDS16, int INFLIGHT>
__global__ void k_gather(const uint4 *buf, uint64_t nlines, uint32_t iters, uint32_t *sink) {
    uint64_t x = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint4 v[INFLIGHT][WORDS16];
#pragma unroll
        for (int j = 0; j < INFLIGHT; j++) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            const uint64_t line = (x >> 20) % nlines;
#pragma unroll
            for (int w = 0; w < WORDS16; w++) v[j][w] = buf[line * WORDS16 + w];
        }
#pragma unroll
        for (int j = 0; j < INFLIGHT; j++)
#pragma unroll
            for (int w = 0; w < WORDS16; w++) acc ^= v[j][w].x ^ v[j][w].y ^ v[j][w].z ^ v[j][w].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// the table walk's pattern: in one step all lanes of a wavefront gather random 128-byte lines of ONE sub-table (sub_bytes, e.g. the
// 2 MiB of a W = 15 (generator, window) pair); the sub-table changes every step
template <int INFLIGHT>
__global__ void k_gather_sub(const uint4 *buf, uint64_t nsub, uint32_t lines_per_sub, uint32_t iters, uint32_t *sink) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint64_t x = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345, y = wave * 0xD1342543DE82EF95ull + 99;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint4 v[INFLIGHT][8];
#pragma unroll
        for (int j = 0; j < INFLIGHT; j++) {
            y = y * 6364136223846793005ull + 1442695040888963407ull;   // same sequence in every lane of the wavefront
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            const uint64_t line = ((y >> 20) % nsub) * lines_per_sub + (uint32_t)(x >> 33) % lines_per_sub;
#pragma unroll
            for (int w = 0; w < 8; w++) v[j][w] = buf[line * 8 + w];
        }
#pragma unroll
        for (int j = 0; j < INFLIGHT; j++)
#pragma unroll
            for (int w = 0; w < 8; w++) acc ^= v[j][w].x ^ v[j][w].y ^ v[j][w].z ^ v[j][w].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <int INFLIGHT>
static void run_sub(const uint4 *buf, uint64_t bytes, uint64_t sub_bytes, uint32_t *sink) {
    const uint64_t nsub = bytes / sub_bytes;
    const uint32_t lines_per_sub = (uint32_t)(sub_bytes / 128), blocks = 256 * 16, threads = 256, iters = 64;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k_gather_sub<INFLIGHT>), dim3(blocks), dim3(threads), 0, 0, buf, nsub, lines_per_sub, 4u, sink);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k_gather_sub<INFLIGHT>), dim3(blocks), dim3(threads), 0, 0, buf, nsub, lines_per_sub, iters, sink);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double gb = (double)blocks * threads * iters * INFLIGHT * 128.0 / 1e9;
    printf("wavefront-local sub-table %8.0f KiB of a %3.0f GiB buffer, 128-B lines, in flight %d : %8.1f GB/s (%.2e lines/s)\n", sub_bytes / 1024.0, bytes / 1073741824.0,
           INFLIGHT, gb / (ms * 1e-3), gb * 1e9 / 128.0 / (ms * 1e-3));
}
template <int WORDS16, int INFLIGHT>
static void run(const uint4 *buf, uint64_t bytes, uint32_t *sink, const char *name) {
    const uint64_t nlines = bytes / (16 * WORDS16);
    const uint32_t blocks = 256 * 16, threads = 256, iters = 64;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((k_gather<WORDS16, INFLIGHT>), dim3(blocks), dim3(threads), 0, 0, buf, nlines, 4u, sink);
    hipEventRecord(a);
    hipLaunchKernelGGL((k_gather<WORDS16, INFLIGHT>), dim3(blocks), dim3(threads), 0, 0, buf, nlines, iters, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double gb = (double)blocks * threads * iters * INFLIGHT * 16.0 * WORDS16 / 1e9;
    printf("%-28s buffer %5.0f GiB  line %3d B  in flight %d : %8.1f GB/s (%.2e lines/s)\n", name, bytes / 1073741824.0, 16 * WORDS16, INFLIGHT, gb / (ms * 1e-3),
           gb * 1e9 / (16.0 * WORDS16) / (ms * 1e-3));
}
int main() {
    uint32_t *sink;
    hipMalloc((void **)&sink, 4);
    for (uint64_t gib : {1ull, 16ull, 64ull}) {
        uint4 *buf;
        if (hipMalloc((void **)&buf, gib << 30) != hipSuccess) { printf("alloc %llu GiB failed\n", (unsigned long long)gib); continue; }
        hipMemset(buf, 1, gib << 30);
        run<8, 1>(buf, gib << 30, sink, "random 128-B lines");
        run<8, 2>(buf, gib << 30, sink, "random 128-B lines");
        run<8, 4>(buf, gib << 30, sink, "random 128-B lines");
        run<4, 2>(buf, gib << 30, sink, "random 64-B lines");
        run<4, 4>(buf, gib << 30, sink, "random 64-B lines");
        run<2, 4>(buf, gib << 30, sink, "random 32-B lines");
        if (gib == 64)
            for (uint64_t sub : {64ull << 10, 1ull << 20, 2ull << 20, 4ull << 20, 32ull << 20}) {   // W = 10, 14, 15, 16, 19
                run_sub<1>(buf, gib << 30, sub, sink);
                run_sub<2>(buf, gib << 30, sub, sink);
            }
        hipFree(buf);
    }
    return 0;
}
