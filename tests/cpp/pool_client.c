/* A plain C client of the pool (include/bpgpu.h, "pool"): what the Rust drop-in of INTEGRATION.md section 3 does through
 * `extern "C"` -- ONE call for a whole file of proofs, from one thread, on one or several devices.
 *   pool_client <fixture.bin> <ndev> <nproofs> <tamper_stride>
 * Reads a bench_data fixture (header "BPBENCH1", u32 n, m, count, proof_len, label_len, label, records), tiles it to
 * nproofs, flips one bit in t_x of every tamper_stride-th proof, verifies with bpgpu_pool_rangeproof_verify on `ndev` shards
 * of device 0 and prints "rejected <k> of <nproofs>: <indices...>".  The test compares that line with the planted pattern. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bpgpu.h"

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    char magic[8];
    uint32_t hdr[5];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "BPBENCH1", 8) || fread(hdr, 4, 5, f) != 5) return 2;
    const uint32_t n = hdr[0], m = hdr[1], count = hdr[2], proof_len = hdr[3], label_len = hdr[4];
    uint8_t label[256];
    if (label_len > sizeof label || fread(label, 1, label_len, f) != label_len) return 2;
    const size_t rec = proof_len + 32 * (size_t)m;
    uint8_t *recs = malloc(rec * count);
    if (fread(recs, rec, count, f) != count) return 2;
    fclose(f);
    const int ndev = atoi(argv[2]);
    const size_t nproofs = (size_t)atol(argv[3]), stride = (size_t)atol(argv[4]);
    uint8_t *proofs = malloc(nproofs * proof_len), *coms = malloc(nproofs * 32 * m), *verdict = malloc(nproofs);
    for (size_t i = 0; i < nproofs; i++) {
        memcpy(proofs + i * proof_len, recs + (i % count) * rec, proof_len);
        memcpy(coms + i * 32 * m, recs + (i % count) * rec + proof_len, 32 * (size_t)m);
        if (stride && i % stride == stride - 1) proofs[i * proof_len + 128] ^= 1;
    }
    int devices[16];
    for (int i = 0; i < ndev && i < 16; i++) devices[i] = 0;
    bpgpu_pool *pool = NULL;
    int rc = bpgpu_pool_create(devices, ndev, 8, &pool);
    if (rc) {
        fprintf(stderr, "bpgpu_pool_create: %d\n", rc);
        return 1;
    }
    if ((rc = bpgpu_pool_set_option(pool, "fixed_window_bits", 16)) || (rc = bpgpu_pool_gens_create(pool, n, m))) {
        fprintf(stderr, "gens: %d %s\n", rc, bpgpu_pool_last_error(pool));
        return 1;
    }
    memset(verdict, 0xff, nproofs);
    /* rng64 = NULL: the library draws the batching challenges itself, as RangeProof::verify_multiple does with thread_rng() */
    rc = bpgpu_pool_rangeproof_verify(pool, n, m, nproofs, proofs, proof_len, coms, label, label_len, NULL, verdict, NULL);
    if (rc) {
        fprintf(stderr, "verify: %d %s\n", rc, bpgpu_pool_last_error(pool));
        return 1;
    }
    size_t k = 0;
    for (size_t i = 0; i < nproofs; i++) k += verdict[i] != BPGPU_VERDICT_OK;
    printf("rejected %zu of %zu:", k, nproofs);
    for (size_t i = 0; i < nproofs; i++)
        if (verdict[i] != BPGPU_VERDICT_OK) printf(" %zu=%d", i, verdict[i]);
    printf("\n");
    bpgpu_pool_destroy(pool);
    return 0;
}
