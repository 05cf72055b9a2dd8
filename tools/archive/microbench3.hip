// Lone-wavefront issue/latency microbenchmark (gfx950): one 64-lane workgroup, dependent vs independent
// instruction chains, to price the wavefront-cooperative kernels (horner_wave.h) which run one wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench3.hip -o /tmp/mb3 && /tmp/mb3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP16(x) x x x x x x x x x x x x x x x x
#define KERNEL(name, body)                                                     \
    __global__ void __launch_bounds__(64) name(uint32_t *out, int iters) {    \
        uint32_t a = threadIdx.x, b = a * 3 + 1, c = a ^ 5, d = a + 7, e = 9; \
        uint64_t q = a, r = b;                                                 \
        for (int i = 0; i < iters; i++) { REP16(body) }                        \
        out[threadIdx.x] = a + b + c + d + e + (uint32_t)q + (uint32_t)r;      \
    }
KERNEL(k_add_dep, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));)
KERNEL(k_add_ind4, asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_mad64_dep, asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(q) : "v"(b), "v"(c) : "s20", "s21");)
KERNEL(k_mad64_ind2, asm volatile("v_mad_u64_u32 %0, s[20:21], %2, %3, %0\n v_mad_u64_u32 %1, s[20:21], %2, %3, %1" : "+v"(q), "+v"(r) : "v"(b), "v"(c) : "s20", "s21");)
KERNEL(k_dpp_dep, asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a));)
KERNEL(k_dpp_ind4, asm volatile("v_mov_b32_dpp %0, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
KERNEL(k_add_then_dpp, asm volatile("v_add_u32 %0, %0, %1\n s_nop 1\n v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b));)
KERNEL(k_swap32_dep, asm volatile("s_nop 1\n v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));)
KERNEL(k_swap16_dep, asm volatile("s_nop 1\n v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));)
KERNEL(k_mad64_dpp_mix, asm volatile("v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_or_b32_dpp %1, %3, %1 row_shl:15 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mad_u64_u32 %0, s[20:21], %1, %3, %0" : "+v"(q), "+v"(a) : "v"(b), "v"(c) : "s20", "s21");)
KERNEL(k_mullo_dep, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));)
KERNEL(k_alignbit_dep, asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a) : "v"(b));)
KERNEL(k_bfi_dep, asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
KERNEL(k_lds_rt, { __shared__ uint32_t l[64]; l[threadIdx.x] = a; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); a = l[(threadIdx.x + 1) & 63] + 1; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); })
KERNEL(k_salu_dep, { uint32_t s; asm volatile("s_add_u32 %0, %1, 1\n s_add_u32 %0, %0, 1" : "=s"(s) : "s"(iters)); e += s; })

template <typename K>
static void run(const char *name, K kern, int per_iter, uint32_t *d, int blocks) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 16 * per_iter;
    printf("%-18s blocks=%5d  %8.3f ms  %6.2f ns/instr  (%5.2f cycles @2.4GHz)\n", name, blocks, ms, ms * 1e6 / n, ms * 1e6 / n * 2.4);
}
int main() {
    uint32_t *d;
    hipMalloc(&d, 4096);
    for (int blocks : {1, 1024}) {
        run("add_dep", k_add_dep, 1, d, blocks);
        run("add_ind4", k_add_ind4, 4, d, blocks);
        run("mad64_dep", k_mad64_dep, 1, d, blocks);
        run("mad64_ind2", k_mad64_ind2, 2, d, blocks);
        run("mullo_dep", k_mullo_dep, 1, d, blocks);
        run("dpp_dep(+nop)", k_dpp_dep, 1, d, blocks);
        run("dpp_ind4", k_dpp_ind4, 4, d, blocks);
        run("add+dpp(+nop)", k_add_then_dpp, 2, d, blocks);
        run("swap32_dep(+nop)", k_swap32_dep, 1, d, blocks);
        run("swap16_dep(+nop)", k_swap16_dep, 1, d, blocks);
        run("dpp,dpp,mad64", k_mad64_dpp_mix, 3, d, blocks);
        run("alignbit_dep", k_alignbit_dep, 1, d, blocks);
        run("bfi_dep", k_bfi_dep, 1, d, blocks);
        run("lds_roundtrip", k_lds_rt, 1, d, blocks);
        run("salu_dep2", k_salu_dep, 2, d, blocks);
    }
    return 0;
}
