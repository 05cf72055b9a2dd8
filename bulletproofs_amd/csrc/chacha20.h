// chacha20.h: the ChaCha20 block function (RFC 8439 layout: 4 constants, 8 key words, a 64-bit block counter, two nonce words), one
// body for the host (hostrng.h: the library's own generator) and the device (rangeproof.h: per-proof randomness expanded from a
// per-chain key inside launch 1).
#ifndef BPGPU_CHACHA20_H
#define BPGPU_CHACHA20_H
#include <stdint.h>
#include "fe25519.h"   // BP_HD

namespace bp {

BP_HD uint32_t cc_rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
#define BP_CC_QR(a, b, c, d)              \
    a += b; d = cc_rotl(d ^ a, 16);       \
    c += d; b = cc_rotl(b ^ c, 12);       \
    a += b; d = cc_rotl(d ^ a, 8);        \
    c += d; b = cc_rotl(b ^ c, 7);
// (named words instead of an array: on the device nothing is indexed at run time, the state lives in 16 registers)
BP_HD void chacha20_block(const uint32_t key[8], uint64_t counter, uint32_t nonce0, uint32_t nonce1, uint32_t out[16]) {
    const uint32_t i0 = 0x61707865u, i1 = 0x3320646eu, i2 = 0x79622d32u, i3 = 0x6b206574u;
    const uint32_t i12 = (uint32_t)counter, i13 = (uint32_t)(counter >> 32);
    uint32_t x0 = i0, x1 = i1, x2 = i2, x3 = i3, x4 = key[0], x5 = key[1], x6 = key[2], x7 = key[3], x8 = key[4], x9 = key[5], x10 = key[6],
             x11 = key[7], x12 = i12, x13 = i13, x14 = nonce0, x15 = nonce1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int r = 0; r < 10; r++) {
        BP_CC_QR(x0, x4, x8, x12) BP_CC_QR(x1, x5, x9, x13) BP_CC_QR(x2, x6, x10, x14) BP_CC_QR(x3, x7, x11, x15)
        BP_CC_QR(x0, x5, x10, x15) BP_CC_QR(x1, x6, x11, x12) BP_CC_QR(x2, x7, x8, x13) BP_CC_QR(x3, x4, x9, x14)
    }
    out[0] = x0 + i0, out[1] = x1 + i1, out[2] = x2 + i2, out[3] = x3 + i3;
    out[4] = x4 + key[0], out[5] = x5 + key[1], out[6] = x6 + key[2], out[7] = x7 + key[3];
    out[8] = x8 + key[4], out[9] = x9 + key[5], out[10] = x10 + key[6], out[11] = x11 + key[7];
    out[12] = x12 + i12, out[13] = x13 + i13, out[14] = x14 + nonce0, out[15] = x15 + nonce1;
}
#undef BP_CC_QR

}  // namespace bp
#endif
