// Host-side Keccak-256 (EVM transcript hash, /root/reference/crates/halo2-verifier/templates/
// Halo2Verifier.sol:105-124) and the ChaCha20 block function that expands the caller's 32-byte
// blinding seed (Fr draw i = from_u512 of ChaCha20 block i; draw order: include/shielder_hip.h at
// shp_prove_with_randomness); the ChaCha block is also compiled for the device.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>

#include "field.hip.hpp"

namespace shp {

// ---------------------------------------------------------------- ChaCha20 (host + device)
SHP_HD uint32_t cc_rotl(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
#define SHP_CC_QR(a, b, c, d)                                     \
    a += b; d ^= a; d = cc_rotl(d, 16); c += d; b ^= c; b = cc_rotl(b, 12); \
    a += b; d ^= a; d = cc_rotl(d, 8);  c += d; b ^= c; b = cc_rotl(b, 7);

// key = 8 words, 64-bit block counter in words 12..13, nonce 0; out = 16 words
SHP_HD void chacha20_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) {
    uint32_t s[16], x[16];
    s[0] = 0x61707865u; s[1] = 0x3320646eu; s[2] = 0x79622d32u; s[3] = 0x6b206574u;
#pragma unroll
    for (int i = 0; i < 8; i++) s[4 + i] = key[i];
    s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32); s[14] = 0; s[15] = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = s[i];
    for (int r = 0; r < 10; r++) {
        SHP_CC_QR(x[0], x[4], x[8], x[12]) SHP_CC_QR(x[1], x[5], x[9], x[13])
        SHP_CC_QR(x[2], x[6], x[10], x[14]) SHP_CC_QR(x[3], x[7], x[11], x[15])
        SHP_CC_QR(x[0], x[5], x[10], x[15]) SHP_CC_QR(x[1], x[6], x[11], x[12])
        SHP_CC_QR(x[2], x[7], x[8], x[13]) SHP_CC_QR(x[3], x[4], x[9], x[14])
    }
#pragma unroll
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}

// Fr draw #counter of the stream keyed by `key`: 64 bytes, wide reduction (halo2curves from_u512)
SHP_HD Fr fr_draw(const uint32_t key[8], uint64_t counter) {
    uint32_t blk[16];
    chacha20_block(key, counter, blk);
    Fr d0, d1;
#pragma unroll
    for (int i = 0; i < 8; i++) { d0.l[i] = blk[i]; d1.l[i] = blk[8 + i]; }
    return Fr::from_u512(d0, d1);
}

// ---------------------------------------------------------------- Keccak-256 (host)
inline uint64_t kk_rol(uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; }

inline void keccak_f1600(uint64_t a[25]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    // rho offsets indexed by lane x + 5 y, pi destination computed on the fly
    static const int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    for (int rnd = 0; rnd < 24; rnd++) {
        uint64_t c[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) {
            uint64_t d = c[(x + 4) % 5] ^ kk_rol(c[(x + 1) % 5], 1);
            for (int y = 0; y < 5; y++) a[x + 5 * y] ^= d;
        }
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = kk_rol(a[x + 5 * y], RHO[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= RC[rnd];
    }
}

inline void keccak256(const uint8_t* in, size_t len, uint8_t out[32]) {
    uint64_t st[25];
    memset(st, 0, sizeof(st));
    const size_t rate = 136;
    while (len >= rate) {
        for (size_t i = 0; i < rate / 8; i++) { uint64_t w; memcpy(&w, in + 8 * i, 8); st[i] ^= w; }
        keccak_f1600(st);
        in += rate;
        len -= rate;
    }
    uint8_t blk[136];
    memset(blk, 0, rate);
    memcpy(blk, in, len);
    blk[len] ^= 0x01;
    blk[rate - 1] ^= 0x80;
    for (size_t i = 0; i < rate / 8; i++) { uint64_t w; memcpy(&w, blk + 8 * i, 8); st[i] ^= w; }
    keccak_f1600(st);
    memcpy(out, st, 32);
}

// 32-byte big-endian canonical encoding of an Fr / Fq (transcript + proof bytes)
template <class F>
inline void to_be32(const F& mont, uint8_t out[32]) {
    F raw = mont.to_raw();
    for (int i = 0; i < 8; i++) {
        uint32_t w = raw.l[7 - i];
        out[4 * i] = (uint8_t)(w >> 24); out[4 * i + 1] = (uint8_t)(w >> 16);
        out[4 * i + 2] = (uint8_t)(w >> 8); out[4 * i + 3] = (uint8_t)w;
    }
}

// the EVM Keccak transcript (Halo2Verifier.sol:105-124, 247-307)
struct Transcript {
    std::vector<uint8_t> buf, proof;
    explicit Transcript(const uint8_t vk_digest[32]) : buf(vk_digest, vk_digest + 32) {}
    void common_scalar(const Fr& v) {
        uint8_t b[32];
        to_be32(v, b);
        buf.insert(buf.end(), b, b + 32);
    }
    void write_scalar(const Fr& v) {
        uint8_t b[32];
        to_be32(v, b);
        buf.insert(buf.end(), b, b + 32);
        proof.insert(proof.end(), b, b + 32);
    }
    void write_point(const Fq& x, const Fq& y) {
        uint8_t b[64];
        to_be32(x, b);
        to_be32(y, b + 32);
        buf.insert(buf.end(), b, b + 64);
        proof.insert(proof.end(), b, b + 64);
    }
    Fr squeeze() {
        if (buf.size() == 32) buf.push_back(0x01);
        uint8_t h[32];
        keccak256(buf.data(), buf.size(), h);
        buf.assign(h, h + 32);
        // hash mod r: the 256-bit big-endian value as raw limbs, reduced through Montgomery form
        Fr raw;
        for (int i = 0; i < 8; i++)
            raw.l[i] = ((uint32_t)h[31 - 4 * i]) | ((uint32_t)h[30 - 4 * i] << 8) | ((uint32_t)h[29 - 4 * i] << 16) |
                       ((uint32_t)h[28 - 4 * i] << 24);
        return Fr::from_raw(raw);      // mul by R^2 accepts any 256-bit input and returns it reduced
    }
};

}  // namespace shp
