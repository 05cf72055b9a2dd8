// hostrng.h: the library's host-side randomness (host code only; included by pool.hip and bpgpu.hip).
#ifndef BPGPU_HOSTRNG_H
#define BPGPU_HOSTRNG_H
#include <stdint.h>
#include <string.h>
#include <sys/random.h>
#include <sys/types.h>
#include <unistd.h>
#include "chacha20.h"

namespace bp {

// ---- the batching challenge's randomness when the caller brings none ------------------------------------------------------
// verify_multiple draws `c` from thread_rng() (src/range_proof/mod.rs:396, 455-470).  Here: one ChaCha20 generator per calling
// thread, keyed from the OS CSPRNG, re-keyed from its own output after every request (fast key erasure) and from the OS every
// 16 MiB -- a getrandom() system call per proof would cost more than staging the proof.  fork() copies the state: a child that went
// on from it would repeat its parent's prover nonces and chain keys (ThreadRng reseeds on fork), so the state remembers
// the process it was keyed in and is keyed afresh in any other.
struct chacha_rng {
    uint32_t key[8];
    uint64_t counter = 0, since_seed = ~0ull;
    pid_t owner = 0;
};
static inline void wipe(void *p, size_t n) {
    memset(p, 0, n);
    __asm__ __volatile__("" : : "r"(p) : "memory");   // (the stores stay: explicit_bzero without the libc dependency)
}
static inline bool fast_random(uint8_t *dst, size_t bytes) {
    static thread_local chacha_rng g;
    const pid_t me = getpid();
    if (g.since_seed > (16ull << 20) || g.owner != me) {
        size_t got = 0;
        while (got < 32) {
            const ssize_t r = getrandom((char *)g.key + got, 32 - got, 0);
            if (r <= 0) return false;
            got += (size_t)r;
        }
        g.since_seed = 0;
        g.counter = 0;
        g.owner = me;
    }
    uint32_t blk[16];
    while (bytes) {
        chacha20_block(g.key, g.counter++, 0u, 0u, blk);
        const size_t take = bytes < 64 ? bytes : 64;
        memcpy(dst, blk, take);
        dst += take;
        bytes -= take;
        g.since_seed += 64;
    }
    chacha20_block(g.key, g.counter++, 0u, 0u, blk);   // the next request runs under a key this one's output does not reveal
    memcpy(g.key, blk, 32);
    wipe(blk, sizeof blk);   // the next key does not linger on the stack
    return true;
}


}  // namespace bp
#endif
