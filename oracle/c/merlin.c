/* ORACLE -- TEST INFRASTRUCTURE ONLY. See merlin.h. */
#include "merlin.h"
#include <string.h>
#include <assert.h>

static const uint64_t KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KECCAK_ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
static const int KECCAK_PI[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
#define ROTL64(x, n) (((x) << (n)) | ((x) >> (64 - (n))))

void keccak_f1600(uint64_t st[25]) {
    uint64_t bc[5], t;
    for (int r = 0; r < 24; r++) {
        for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; i++) {
            t = bc[(i + 4) % 5] ^ ROTL64(bc[(i + 1) % 5], 1);
            for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
        }
        t = st[1];
        for (int i = 0; i < 24; i++) {
            int j = KECCAK_PI[i];
            bc[0] = st[j];
            st[j] = ROTL64(t, KECCAK_ROT[i]);
            t = bc[0];
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = st[j + i];
            for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= KECCAK_RC[r];
    }
}

/* ---- sponge (little-endian host) ---- */
static void sponge_init(keccak_sponge *k, size_t rate) {
    memset(k, 0, sizeof *k); k->rate = rate;
}
void shake256_init(keccak_sponge *k) { sponge_init(k, 136); }
void sponge_absorb(keccak_sponge *k, const uint8_t *in, size_t n) {
    uint8_t *s = (uint8_t *)k->st;
    for (size_t i = 0; i < n; i++) {
        s[k->pos++] ^= in[i];
        if (k->pos == k->rate) { keccak_f1600(k->st); k->pos = 0; }
    }
}
static void sponge_finish(keccak_sponge *k, uint8_t suffix) {
    uint8_t *s = (uint8_t *)k->st;
    s[k->pos] ^= suffix;
    s[k->rate - 1] ^= 0x80;
    keccak_f1600(k->st);
    k->pos = 0; k->squeezing = 1;
}
void shake256_squeeze(keccak_sponge *k, uint8_t *out, size_t n) {
    if (!k->squeezing) sponge_finish(k, 0x1f);
    uint8_t *s = (uint8_t *)k->st;
    for (size_t i = 0; i < n; i++) {
        if (k->pos == k->rate) { keccak_f1600(k->st); k->pos = 0; }
        out[i] = s[k->pos++];
    }
}
void sha3_512(uint8_t out[64], const uint8_t *in, size_t n) {
    keccak_sponge k; sponge_init(&k, 72);
    sponge_absorb(&k, in, n);
    sponge_finish(&k, 0x06);
    memcpy(out, k.st, 64);
}

/* ---- STROBE-128 subset used by Merlin ---- */
#define STROBE_R 166
#define FLAG_I 1
#define FLAG_A 2
#define FLAG_C 4
#define FLAG_T 8
#define FLAG_M 16
#define FLAG_K 32

static void strobe_run_f(merlin_transcript *t) {
    t->st[t->pos] ^= t->pos_begin;
    t->st[t->pos + 1] ^= 0x04;
    t->st[STROBE_R + 1] ^= 0x80;
    uint64_t w[25]; memcpy(w, t->st, 200);
    keccak_f1600(w);
    memcpy(t->st, w, 200);
    t->pos = 0; t->pos_begin = 0;
}
static void strobe_absorb(merlin_transcript *t, const uint8_t *d, size_t n) {
    for (size_t i = 0; i < n; i++) {
        t->st[t->pos++] ^= d[i];
        if (t->pos == STROBE_R) strobe_run_f(t);
    }
}
static void strobe_squeeze(merlin_transcript *t, uint8_t *d, size_t n) {
    for (size_t i = 0; i < n; i++) {
        d[i] = t->st[t->pos];
        t->st[t->pos++] = 0;
        if (t->pos == STROBE_R) strobe_run_f(t);
    }
}
static void strobe_begin_op(merlin_transcript *t, uint8_t flags, int more) {
    if (more) { assert(flags == t->cur_flags); return; }
    uint8_t old_begin = t->pos_begin;
    t->pos_begin = t->pos + 1;
    t->cur_flags = flags;
    uint8_t hdr[2] = {old_begin, flags};
    strobe_absorb(t, hdr, 2);
    if ((flags & (FLAG_C | FLAG_K)) && t->pos != 0) strobe_run_f(t);
}
static void strobe_meta_ad(merlin_transcript *t, const uint8_t *d, size_t n, int more) {
    strobe_begin_op(t, FLAG_M | FLAG_A, more); strobe_absorb(t, d, n);
}
static void strobe_ad(merlin_transcript *t, const uint8_t *d, size_t n, int more) {
    strobe_begin_op(t, FLAG_A, more); strobe_absorb(t, d, n);
}
static void strobe_prf(merlin_transcript *t, uint8_t *d, size_t n) {
    strobe_begin_op(t, FLAG_I | FLAG_A | FLAG_C, 0); strobe_squeeze(t, d, n);
}

void merlin_init(merlin_transcript *t, const uint8_t *label, size_t label_len) {
    memset(t, 0, sizeof *t);
    const uint8_t hdr[6] = {1, STROBE_R + 2, 1, 0, 1, 96};
    memcpy(t->st, hdr, 6);
    memcpy(t->st + 6, "STROBEv1.0.2", 12);
    uint64_t w[25]; memcpy(w, t->st, 200); keccak_f1600(w); memcpy(t->st, w, 200);
    strobe_meta_ad(t, (const uint8_t *)"Merlin v1.0", 11, 0);
    merlin_append_message(t, "dom-sep", label, label_len);
}
void merlin_append_message(merlin_transcript *t, const char *label, const uint8_t *msg, size_t n) {
    uint8_t len4[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    strobe_meta_ad(t, (const uint8_t *)label, strlen(label), 0);
    strobe_meta_ad(t, len4, 4, 1);
    strobe_ad(t, msg, n, 0);
}
void merlin_append_u64(merlin_transcript *t, const char *label, uint64_t x) {
    uint8_t b[8]; for (int i = 0; i < 8; i++) b[i] = (uint8_t)(x >> (8 * i));
    merlin_append_message(t, label, b, 8);
}
void merlin_challenge_bytes(merlin_transcript *t, const char *label, uint8_t *out, size_t n) {
    uint8_t len4[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    strobe_meta_ad(t, (const uint8_t *)label, strlen(label), 0);
    strobe_meta_ad(t, len4, 4, 1);
    strobe_prf(t, out, n);
}
