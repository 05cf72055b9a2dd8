// hostrng.h: the library's host-side randomness (host code only; included by pool.hip and bpgpu.hip).
#ifndef BPGPU_HOSTRNG_H
#define BPGPU_HOSTRNG_H
#include <stdint.h>
#include <string.h>
#include <sys/random.h>
#include <sys/types.h>

namespace bp {

// ---- the batching challenge's randomness when the caller brings none ------------------------------------------------------
// verify_multiple draws `c` from thread_rng() (src/range_proof/mod.rs:396, 455-470).  Here: one ChaCha20 generator per calling
// thread, keyed from the OS CSPRNG, re-keyed from its own output after every request (fast key erasure) and from the OS every
// 16 MiB -- a getrandom() system call per proof would cost more than staging the proof.
struct chacha_rng {
    uint32_t key[8];
    uint64_t counter = 0, since_seed = ~0ull;
};
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static inline void chacha20_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) {
    uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                       (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
    uint32_t x[16];
    for (int i = 0; i < 16; i++) x[i] = in[i];
#define BP_QR(a, b, c, d)                    \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16); \
    x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12); \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);  \
    x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
    for (int r = 0; r < 10; r++) {
        BP_QR(0, 4, 8, 12) BP_QR(1, 5, 9, 13) BP_QR(2, 6, 10, 14) BP_QR(3, 7, 11, 15)
        BP_QR(0, 5, 10, 15) BP_QR(1, 6, 11, 12) BP_QR(2, 7, 8, 13) BP_QR(3, 4, 9, 14)
    }
#undef BP_QR
    for (int i = 0; i < 16; i++) out[i] = x[i] + in[i];
}
static inline bool fast_random(uint8_t *dst, size_t bytes) {
    static thread_local chacha_rng g;
    if (g.since_seed > (16ull << 20)) {
        size_t got = 0;
        while (got < 32) {
            const ssize_t r = getrandom((char *)g.key + got, 32 - got, 0);
            if (r <= 0) return false;
            got += (size_t)r;
        }
        g.since_seed = 0;
        g.counter = 0;
    }
    uint32_t blk[16];
    while (bytes) {
        chacha20_block(g.key, g.counter++, blk);
        const size_t take = bytes < 64 ? bytes : 64;
        memcpy(dst, blk, take);
        dst += take;
        bytes -= take;
        g.since_seed += 64;
    }
    chacha20_block(g.key, g.counter++, blk);   // the next request runs under a key this one's output does not reveal
    memcpy(g.key, blk, 32);
    return true;
}


}  // namespace bp
#endif
