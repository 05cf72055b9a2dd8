/* ORACLE -- TEST INFRASTRUCTURE ONLY.
 * Keccak-f[1600], SHAKE256, SHA3-512 (FIPS 202) and the Merlin transcript
 * (STROBE-128) as used through
 * /root/reference/src/transcript.rs:43-95 and src/generators.rs:48,64-72.
 * merlin ^2 / sha3 0.8 are external crates (Cargo.toml:23,31), not vendored.
 */
#ifndef ORACLE_MERLIN_H
#define ORACLE_MERLIN_H
#include <stdint.h>
#include <stddef.h>

void keccak_f1600(uint64_t st[25]);

typedef struct { uint64_t st[25]; size_t pos; size_t rate; int squeezing; } keccak_sponge;
void shake256_init(keccak_sponge *k);
void sponge_absorb(keccak_sponge *k, const uint8_t *in, size_t n);
void shake256_squeeze(keccak_sponge *k, uint8_t *out, size_t n);
void sha3_512(uint8_t out[64], const uint8_t *in, size_t n);

typedef struct {
    uint8_t st[200];
    uint8_t pos, pos_begin, cur_flags;
} merlin_transcript;

void merlin_init(merlin_transcript *t, const uint8_t *label, size_t label_len);
void merlin_append_message(merlin_transcript *t, const char *label, const uint8_t *msg, size_t n);
void merlin_append_u64(merlin_transcript *t, const char *label, uint64_t x);
void merlin_challenge_bytes(merlin_transcript *t, const char *label, uint8_t *out, size_t n);
#endif
