// RMJ_RULE_REFERENCE_RNG: the reference's seed -> wall (state/wall.rs:36-56, state_3p/wall.rs:75-100) on one wave.
//
// The reference draws its wall through third-party crates (Cargo.lock: rand 0.10.0, rand_core 0.10.0, chacha20 0.10.0):
//     rng  = StdRng::seed_from_u64(hand_seed)      PCG32 (XSH-RR) expands the u64 to a 32-byte key; StdRng = ChaCha12,
//                                                  64-bit block counter from 0, stream 0, u32 words in key stream order
//     w.shuffle(&mut rng)                          for i in 0..len: swap(i, index_i), index_i uniform in [0, i]; the
//                                                  indices are cut from u32 chunks whose range is the product of their bounds
//                                                  (IncreasingUniform), a chunk = Canon's method on one or two u32 samples
//     salt = rng.next_u64()                        the next two words
// The published algorithms are restated in oracle/ref_rng.hpp (what is pinned on published vectors and what is not is
// listed there and in DESIGN.md §6); this file is the same definition laid out for a wave:
//   * the four ChaCha12 blocks a wall can consume (28 chunks x at most two words + the salt = 58 words) are computed by
//     16 lanes, four per block, one state column each; the diagonal rounds are DPP quad permutes;
//   * the chunk walk is the only data dependent chain (does chunk c take one word or two?): 28 scalar steps on v_readlane;
//   * the indices leave their chunks lane-parallel (lane = position: chunk / divisor % (i + 1), divisors precomputed);
//   * the swaps are a serial inside-out pass by one lane over bytes in LDS (w[i] = w[j]; w[j] = id_i) - latency, not issue
//     slots, and the step kernels are issue bound.
// SHA-256(salt || wall) is not computed here: the digest is a function of (salt, wall), both kept in the wall slab, and is
// evaluated on demand by k_wall_digest (rmj_get_wall_digest) - the reference only ever reads it back.
#pragma once
#include <stdint.h>

namespace rmj {

#define RMJ_RR_CHUNKS 28   /* chunks of a 136-element shuffle (a 108-element one uses the first 21) */

struct RefRngTab {
    uint32_t prod[32];      // range of chunk c
    uint32_t div[136];      // position i: product of the bounds of the earlier indices of its chunk
    uint8_t chunk_of[136];  // position i: its chunk
};
constexpr RefRngTab make_refrng_tab() {
    RefRngTab t{};
    for (int i = 0; i < 32; i++) t.prod[i] = 1;
    t.div[0] = 1; t.chunk_of[0] = 31;   // index 0 is 0 without a sample: chunk 31 holds 0, 0 / 1 % 1 = 0
    int i = 1, c = 0;
    while (i < 136) {
        // rand `calculate_bound_u32(m)`: the largest m (m+1) ... (m+count-1) below 2^32
        uint32_t m = (uint32_t)(i + 1), product = m, current = m + 1;
        while ((uint64_t)product * current <= 0xFFFFFFFFull) { product *= current; current++; }
        const int count = (int)(current - m);
        t.prod[c] = product;
        uint32_t d = 1;
        for (int k = 0; k < count && i + k < 136; k++) {
            t.chunk_of[i + k] = (uint8_t)c;
            t.div[i + k] = d;
            d *= (uint32_t)(i + k + 1);   // (the last one overflows nothing: d <= product)
        }
        i += count;
        c++;
    }
    return t;
}
__constant__ const RefRngTab g_refrng = make_refrng_tab();

__device__ __forceinline__ uint32_t rr_rotl(uint32_t x, int n) { return __builtin_rotateleft32(x, (uint32_t)n); }
#define RR_QR(a, b, c, d)                          \
    do {                                           \
        a += b; d = rr_rotl(d ^ a, 16);            \
        c += d; b = rr_rotl(b ^ c, 12);            \
        a += b; d = rr_rotl(d ^ a, 8);             \
        c += d; b = rr_rotl(b ^ c, 7);             \
    } while (0)
#define RR_QP(v, ctrl) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, 0xf, 0xf, false))

// The wall of `hand_seed` BEFORE the reversal into w[0..N) (LDS bytes), its salt returned (uniform).  `scr`: 64 dwords of LDS
// (the key stream words, then reused: idx[N] bytes).  `w` may overlap nothing of `scr`.  All 64 lanes must call.
template <int N, bool SANMA>
__device__ inline uint64_t refrng_wall(uint64_t hand_seed, int lane, uint32_t* scr, uint8_t* w) {
    // ---- rand_core `seed_from_u64`: eight PCG32 outputs = the ChaCha key (uniform arithmetic)
    uint64_t st = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(hand_seed >> 32)) << 32) |
                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)hand_seed);
    uint32_t key[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        st = st * 6364136223846793005ull + 11634580027462260723ull;
        const uint32_t xs = (uint32_t)(((st >> 18) ^ st) >> 27), rot = (uint32_t)(st >> 59);
        key[i] = __builtin_rotateright32(xs, rot);
    }
    // ---- ChaCha12, blocks 0..3: lane = 4 * block + column, rows a (constants), b, c (key), d (counter | stream)
    {
        const int col = lane & 3, blk = (lane >> 2) & 15;
        const uint32_t a0 = col == 0 ? 0x61707865u : col == 1 ? 0x3320646eu : col == 2 ? 0x79622d32u : 0x6b206574u;
        const uint32_t b0 = col == 0 ? key[0] : col == 1 ? key[1] : col == 2 ? key[2] : key[3];
        const uint32_t c0 = col == 0 ? key[4] : col == 1 ? key[5] : col == 2 ? key[6] : key[7];
        const uint32_t d0 = col == 0 ? (uint32_t)blk : 0u;   // words 12-13: block counter, 14-15: stream id 0
        uint32_t a = a0, b = b0, c = c0, d = d0;
#pragma unroll
        for (int r = 0; r < 6; r++) {
            RR_QR(a, b, c, d);                                         // columns
            b = RR_QP(b, 0x39); c = RR_QP(c, 0x4E); d = RR_QP(d, 0x93);   // lane gets row k from column + k: diagonals
            RR_QR(a, b, c, d);
            b = RR_QP(b, 0x93); c = RR_QP(c, 0x4E); d = RR_QP(d, 0x39);
        }
        if (lane < 16) {
            uint32_t* o = scr + 16 * blk + col;
            o[0] = a + a0; o[4] = b + b0; o[8] = c + c0; o[12] = d + d0;
        }
    }
    wave_sync();
    const uint32_t wp = scr[lane];   // lane p: word p of the key stream
    // ---- IncreasingUniform: chunk c = random_range(..prod[c]) (Canon's method: a second sample only if the low half could carry)
    constexpr int NCH = N == 136 ? 28 : 21;
    const uint32_t prodv = g_refrng.prod[lane & 31];
    uint32_t mychunk = 0;
    int p = 0;
    for (int c = 0; c < NCH; c++) {
        const uint32_t prod = (uint32_t)__builtin_amdgcn_readlane((int)prodv, c);
        const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)wp, p);
        const uint64_t m = (uint64_t)x * prod;
        uint32_t res = (uint32_t)(m >> 32);
        const uint32_t lo = (uint32_t)m;
        p += 1;
        if (lo > 0u - prod) {
            const uint32_t y = (uint32_t)__builtin_amdgcn_readlane((int)wp, p);
            p += 1;
            res += (uint32_t)(((uint64_t)lo + (uint32_t)(((uint64_t)y * prod) >> 32)) >> 32);
        }
        if (lane == c) mychunk = res;
    }
    const uint64_t salt = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)wp, p) |
                          ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)wp, p + 1) << 32);
    wave_sync();   // everybody has its word: the scratch becomes idx[]
    uint8_t* idx = reinterpret_cast<uint8_t*>(scr);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int i = lane + 64 * k;
        const int ii = i < N ? i : 0;
        const uint32_t cv = (uint32_t)__shfl((int)mychunk, (int)g_refrng.chunk_of[ii], 64);   // (lane 31 holds 0 for i = 0)
        const uint32_t v = (cv / g_refrng.div[ii]) % (uint32_t)(ii + 1);
        if (i < N) idx[i] = (uint8_t)v;
    }
    wave_sync();
    // ---- for i in 0..N: swap(i, idx[i]) on w = the tile universe in order.  Position i is untouched before step i, so
    // the swap is w[i] = w[j]; w[j] = id_i (j == i: the second store wins).
    if (lane == 0) {
#pragma unroll 4
        for (int i = 0; i < N; i++) {
            const int j = idx[i];
            const uint8_t b = w[j];
            w[i] = b;
            w[j] = (uint8_t)((SANMA && i >= 4) ? i + 28 : i);   // i-th id of the universe (3P: no 2m-8m, types.rs:378-382)
        }
    }
    wave_sync();
    return salt;
}

// SHA-256 (FIPS 180-4) of  salt as 16 lower-case hex digits || wall before the reversal  by ONE lane: wall_digest of
// state/wall.rs:50-55.  `W` = the game's wall row (reversed orientation: w[k] = W[n - 1 - k]).  n + 16 <= 152: three blocks.
__device__ inline void sha256_wall(const uint8_t* W, int n, uint64_t salt, uint32_t out[8]) {
    const uint32_t K[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu,
        0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau,
        0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u,
        0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u,
        0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu,
        0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    const int len = n + 16;
    auto byte_at = [&](int k) -> uint32_t {   // message byte k (with the padding of FIPS 180-4 §5.1.1)
        if (k < 16) { const uint32_t nib = (uint32_t)(salt >> (60 - 4 * k)) & 15u; return nib < 10 ? 48u + nib : 87u + nib; }
        if (k < len) return W[n - 1 - (k - 16)];
        if (k == len) return 0x80u;
        if (k >= 192 - 4) return ((uint32_t)(len * 8) >> (8 * (191 - k))) & 0xFFu;   // 64-bit big endian bit count, high half 0
        return 0u;
    };
    for (int blk = 0; blk < 3; blk++) {
        uint32_t wv[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int k = 64 * blk + 4 * i;
            wv[i] = byte_at(k) << 24 | byte_at(k + 1) << 16 | byte_at(k + 2) << 8 | byte_at(k + 3);
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
        for (int i = 0; i < 64; i++) {
            if (i >= 16) {
                const uint32_t w15 = wv[(i + 1) & 15], w2 = wv[(i + 14) & 15];
                const uint32_t s0 = __builtin_rotateright32(w15, 7) ^ __builtin_rotateright32(w15, 18) ^ (w15 >> 3);
                const uint32_t s1 = __builtin_rotateright32(w2, 17) ^ __builtin_rotateright32(w2, 19) ^ (w2 >> 10);
                wv[i & 15] = wv[i & 15] + s0 + wv[(i + 9) & 15] + s1;
            }
            const uint32_t S1 = __builtin_rotateright32(e, 6) ^ __builtin_rotateright32(e, 11) ^ __builtin_rotateright32(e, 25);
            const uint32_t t1 = hh + S1 + ((e & f) ^ (~e & g)) + K[i] + wv[i & 15];
            const uint32_t S0 = __builtin_rotateright32(a, 2) ^ __builtin_rotateright32(a, 13) ^ __builtin_rotateright32(a, 22);
            const uint32_t t2 = S0 + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = h[i];
}

}  // namespace rmj
