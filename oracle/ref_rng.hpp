// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// The reference's seed -> wall (riichienv-core/src/state/wall.rs:36-67, state_3p/wall.rs:75-110) runs through
// third-party crates that are NOT in /root/reference; Cargo.lock pins them (Cargo.lock:667-681, 84-92, 862-868):
//     rand 0.10.0 (StdRng, SliceRandom::shuffle, random_range), rand_core 0.10.0 (SeedableRng::seed_from_u64),
//     chacha20 0.10.0 (ChaCha12 block function behind StdRng), sha2 0.10.9 (Sha256).
// This header restates their PUBLISHED algorithms.  What pins what (tests/test_oracle_ref_rng.py):
//   * ChaCha block function            — the Strombergson draft test vectors (TC1, 8 / 12 / 20 rounds), RFC 7539 §2.3.2
//   * StdRng = ChaCha12, key = seed, 64-bit block counter from 0, stream 0, u32 words consumed in order,
//     next_u64 = lo | hi << 32, fill_bytes = whole words   — rand's own value-stability vector (rand src/rngs/std.rs,
//     `test_stdrng_construction`: seed [1,0,0,0, 23,0,0,0, 200,1,0,0, 210,30,0,0, 0...] -> 10719222850664546238, and
//     StdRng::from_rng of it -> 14064965282130556830)
//   * SHA-256                          — FIPS 180-4 examples and Python's hashlib on random messages
//   * UNPINNED (no published vector, no reference test): the PCG32 seed expansion of `seed_from_u64`, the index draws of
//     `shuffle` (IncreasingUniform chunks + Canon's method on u32 samples).  They are restated from the crates'
//     sources as published for rand 0.9 / rand_core 0.9, whose output rand 0.10 documents as unchanged for StdRng.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace orc {
namespace refrng {

// (n may be 0 - PCG32's rotation is data dependent -: the complementary shift is taken mod 32; round 6, found by UBSan: "shift exponent 32")
inline uint32_t rotl32(uint32_t x, int n) { return (x << (n & 31)) | (x >> ((32 - n) & 31)); }
inline uint32_t rotr32(uint32_t x, int n) { return (x >> (n & 31)) | (x << ((32 - n) & 31)); }

// ChaCha block (D. J. Bernstein; state layout of RFC 7539 with a 64-bit counter in words 12-13 and a 64-bit stream id in
// 14-15, as rand_chacha / chacha20's `rng` module use it).  `rounds` = 8, 12 or 20.
inline void chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds, uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                      (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    uint32_t w[16];
    std::memcpy(w, s, sizeof w);
    auto qr = [&](int a, int b, int c, int d) {
        w[a] += w[b]; w[d] = rotl32(w[d] ^ w[a], 16);
        w[c] += w[d]; w[b] = rotl32(w[b] ^ w[c], 12);
        w[a] += w[b]; w[d] = rotl32(w[d] ^ w[a], 8);
        w[c] += w[d]; w[b] = rotl32(w[b] ^ w[c], 7);
    };
    for (int r = 0; r < rounds; r += 2) {
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
    }
    for (int i = 0; i < 16; i++) out[i] = w[i] + s[i];
}

// rand_core `SeedableRng::seed_from_u64` (default method): a PCG32 (XSH-RR) stream fills the seed four bytes at a time.
inline void seed_from_u64(uint64_t state, uint8_t seed[32]) {
    const uint64_t MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
    for (int i = 0; i < 8; i++) {
        state = state * MUL + INC;  // advance first (the input may have low Hamming weight)
        uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
        uint32_t rot = (uint32_t)(state >> 59);
        uint32_t x = rotr32(xorshifted, (int)rot & 31);
        seed[4 * i] = (uint8_t)x; seed[4 * i + 1] = (uint8_t)(x >> 8); seed[4 * i + 2] = (uint8_t)(x >> 16); seed[4 * i + 3] = (uint8_t)(x >> 24);  // to_le_bytes
    }
}

// rand::rngs::StdRng = ChaCha12Rng: a block RNG over the ChaCha12 key stream (the crates buffer four blocks; the word
// sequence a caller sees is the key stream in order, whatever the buffering - BlockRng::next_u64 at the buffer edge takes
// the last word of one buffer as the low half and the first of the next as the high half).
struct StdRng {
    uint32_t key[8];
    uint64_t block = 0;  // next block to generate
    uint32_t buf[16];
    int idx = 16;
    uint64_t words_drawn = 0;
    static StdRng from_seed(const uint8_t seed[32]) {
        StdRng r;
        for (int i = 0; i < 8; i++)
            r.key[i] = (uint32_t)seed[4 * i] | (uint32_t)seed[4 * i + 1] << 8 | (uint32_t)seed[4 * i + 2] << 16 | (uint32_t)seed[4 * i + 3] << 24;
        return r;
    }
    static StdRng seed_from_u64(uint64_t s) {
        uint8_t seed[32];
        refrng::seed_from_u64(s, seed);
        return from_seed(seed);
    }
    uint32_t next_u32() {
        if (idx >= 16) { chacha_block(key, block++, 0, 12, buf); idx = 0; }
        words_drawn++;
        return buf[idx++];
    }
    uint64_t next_u64() { uint64_t lo = next_u32(); uint64_t hi = next_u32(); return hi << 32 | lo; }
    void fill_bytes(uint8_t* dst, size_t n) {  // BlockRng::fill_bytes: whole words, little endian, a partial last word is dropped
        while (n) { uint32_t w = next_u32(); for (int b = 0; b < 4 && n; b++, n--) *dst++ = (uint8_t)(w >> (8 * b)); }
    }
};

// rand `UniformInt<u32>::sample_single(0, bound)` = sample_single_inclusive(0, bound - 1): Canon's method, one u32 sample,
// a second one only when the low product could carry ("biased" variant, the crate's default feature set).
inline uint32_t random_range_u32(StdRng& rng, uint32_t bound) {
    const uint32_t range = bound;  // (high - low + 1); `bound` is never 0 or 2^32 here
    uint64_t m = (uint64_t)rng.next_u32() * range;
    uint32_t result = (uint32_t)(m >> 32), lo_order = (uint32_t)m;
    if (lo_order > (uint32_t)(0u - range)) {
        uint32_t new_hi = (uint32_t)(((uint64_t)rng.next_u32() * range) >> 32);
        bool overflow = (uint64_t)lo_order + new_hi > 0xFFFFFFFFull;
        result += overflow ? 1u : 0u;
    }
    return result;
}

// rand `seq::increasing_uniform::calculate_bound_u32`: the largest product m (m+1) ... (m+count-1) that fits a u32.
inline void calculate_bound_u32(uint32_t m, uint32_t& product_out, uint8_t& count_out) {
    uint32_t product = m, current = m + 1;
    for (;;) {
        uint64_t p = (uint64_t)product * current;
        if (p > 0xFFFFFFFFull) break;
        product = (uint32_t)p;
        current++;
    }
    product_out = product;
    count_out = (uint8_t)(current - m);
}

// rand `SliceRandom::shuffle` for len < 2^32 (seq/slice.rs: partial_shuffle(len) -> IncreasingUniform): for i in 0..len
// swap(i, index_i) with index_i uniform in [0, i]; several indices are cut from one u32 sample whose range is the product
// of their bounds (value % n, value /= n; the last index of a chunk is what remains).
template <class T>
inline void shuffle(std::vector<T>& v, StdRng& rng) {
    if (v.size() <= 1) return;
    uint32_t n = 0, chunk = 0;
    uint8_t chunk_remaining = 1;  // n == 0: the first index is 0 without a sample
    for (size_t i = 0; i < v.size(); i++) {
        uint32_t next_n = n + 1;
        uint8_t next_remaining;
        if (chunk_remaining == 0) {
            uint32_t bound; uint8_t remaining;
            calculate_bound_u32(next_n, bound, remaining);
            chunk = random_range_u32(rng, bound);
            next_remaining = (uint8_t)(remaining - 1);
        } else
            next_remaining = (uint8_t)(chunk_remaining - 1);
        uint32_t index;
        if (next_remaining == 0)
            index = chunk;
        else { index = chunk % next_n; chunk /= next_n; }
        chunk_remaining = next_remaining;
        n = next_n;
        std::swap(v[i], v[index]);
    }
}

// FIPS 180-4 SHA-256
struct Sha256 {
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    uint8_t blk[64];
    size_t fill = 0;
    uint64_t total = 0;
    static void compress(uint32_t h[8], const uint8_t b[64]) {
        static const uint32_t K[64] = {
            0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu,
            0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau,
            0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u,
            0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u,
            0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu,
            0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = (uint32_t)b[4 * i] << 24 | (uint32_t)b[4 * i + 1] << 16 | (uint32_t)b[4 * i + 2] << 8 | b[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3);
            uint32_t s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25), ch = (e & f) ^ (~e & g);
            uint32_t t1 = hh + S1 + ch + K[i] + w[i];
            uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22), maj = (a & bb) ^ (a & c) ^ (bb & c);
            uint32_t t2 = S0 + maj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
        }
        h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void update(const uint8_t* p, size_t n) {
        total += n;
        while (n) {
            size_t k = 64 - fill < n ? 64 - fill : n;
            std::memcpy(blk + fill, p, k);
            fill += k; p += k; n -= k;
            if (fill == 64) { compress(h, blk); fill = 0; }
        }
    }
    void finalize(uint8_t out[32]) {
        uint64_t bits = total * 8;
        uint8_t pad = 0x80;
        update(&pad, 1);
        uint8_t z = 0;
        while (fill != 56) update(&z, 1);
        uint8_t len[8];
        for (int i = 0; i < 8; i++) len[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(len, 8);
        for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
    }
};

inline std::string hex(const uint8_t* p, size_t n) {
    static const char* d = "0123456789abcdef";
    std::string s;
    for (size_t i = 0; i < n; i++) { s += d[p[i] >> 4]; s += d[p[i] & 15]; }
    return s;
}

// state/wall.rs:36-56 up to (not including) the reversal: w = ids shuffled by StdRng::seed_from_u64(hand_seed),
// salt = format!("{:016x}", rng.next_u64()), wall_digest = hex(SHA-256(salt bytes || w)).
struct RefWall {
    std::vector<uint8_t> w;
    uint64_t salt_u64;
    std::string salt, digest;
    uint64_t words_drawn;
};
inline std::string wall_digest(const std::string& salt, const std::vector<uint8_t>& w) {
    Sha256 hs;
    hs.update(reinterpret_cast<const uint8_t*>(salt.data()), salt.size());
    for (uint8_t t : w) hs.update(&t, 1);
    uint8_t dg[32];
    hs.finalize(dg);
    return hex(dg, 32);
}
inline RefWall reference_wall(uint64_t hand_seed, std::vector<uint8_t> ids) {
    StdRng rng = StdRng::seed_from_u64(hand_seed);
    shuffle(ids, rng);
    RefWall r;
    r.salt_u64 = rng.next_u64();
    uint8_t sb[8];
    for (int i = 0; i < 8; i++) sb[i] = (uint8_t)(r.salt_u64 >> (56 - 8 * i));
    r.salt = hex(sb, 8);  // {:016x}
    r.digest = wall_digest(r.salt, ids);
    r.w = std::move(ids);
    r.words_drawn = rng.words_drawn;
    return r;
}

}  // namespace refrng
}  // namespace orc
