// Shanten (row A7): replacement-number tables generated FROM FIRST PRINCIPLES at library init
// (no data copied from the reference's nyanten_*.bin blobs) + device lookup.
//
// Reference semantics: shanten.rs:163-261 (calc_normal / calc_chitoi / calc_kokushi /
// calc_shanten_from_counts), :407-461 (3P relocation of 1m/9m into empty honor slots).
// The reference looks the "replacement number" up in Cryolite/nyanten tables; the quantity itself is
//   r(c, m) = min over winning shapes t (m mentsu + 1 pair, every tile count <= 4) of sum_i max(0, t_i - c_i)
// and shanten = r - 1.  Here the same quantity is tabulated per suit for (k mentsu, p pair) by a DP over tile
// ranks (state = sequences started at the two previous ranks) and suits are merged by a (min,+) convolution.
#pragma once
#include <stdint.h>

#include <vector>

#include "rmj_hand.hip.h"

namespace rmj {

#define SH_SUIT_ENTRIES 405350
#define SH_HONOR_ENTRIES 43130

struct ShantenTables {             // device pointers
    const uint64_t* suit;          // [405350] 10 x 4-bit costs: idx = p*5 + k
    const uint64_t* honor;         // [43130]
    const uint32_t* rank9;         // [9][15][5]
    const uint32_t* rank7;         // [7][15][5]
    // Round 5, for the pair-dense ukeire walk (rmj_ukeire.hip.h): the same perfect hash as two dependent reads instead of nine,
    // and the same cost vectors with 6-bit fields (sums of two costs do not carry into the next field)
    const uint32_t* r2;            // [SH_R2_WORDS]: partial ranks keyed by the packed 3-bit fields themselves, see sh_gen_r2
    const uint64_t* v6;            // [405350 + 43130]: low dword = pair 0 (k = 0..4, 6 bits each), high dword = pair 1; honors behind the suits
};
#define SH_R2_HI9 0u                       /* [32768] fields 0..4 of a suit word -> (partial rank | running sum << 20) */
#define SH_R2_HI7 32768u                   /* [32768] the same for the honor word */
#define SH_R2_LO9 65536u                   /* [15][4096] (running sum, fields 5..8) -> partial rank */
#define SH_R2_LO7 (65536u + 15u * 4096u)   /* [15][64]   (running sum, fields 5..6) */
#define SH_R2_WORDS (SH_R2_LO7 + 15u * 64u)

// ---------------------------------------------------------------- host-side generator
struct ShantenHostTables {
    std::vector<uint64_t> suit, honor;
    std::vector<uint32_t> rank9, rank7;
    std::vector<uint32_t> r2;
    std::vector<uint64_t> v6;
};

inline void sh_rank_table(int n, std::vector<uint32_t>& T) {
    // N[i][s] = number of ways to fill ranks i..n-1 with digits 0..4 and total (incl. s) <= 14
    std::vector<std::vector<uint64_t>> N(n + 1, std::vector<uint64_t>(16, 0));
    for (int s = 0; s <= 14; s++) N[n][s] = 1;
    for (int i = n - 1; i >= 0; i--)
        for (int s = 0; s <= 14; s++) {
            uint64_t v = 0;
            for (int d = 0; d <= 4 && s + d <= 14; d++) v += N[i + 1][s + d];
            N[i][s] = v;
        }
    T.assign((size_t)n * 15 * 5, 0);
    for (int i = 0; i < n; i++)
        for (int s = 0; s <= 14; s++) {
            uint64_t acc = 0;
            for (int c = 0; c <= 4; c++) {
                T[((size_t)i * 15 + s) * 5 + c] = (uint32_t)acc;
                if (s + c <= 14) acc += N[i + 1][s + c];
            }
        }
}

struct SuitDP {  // f[k][p][y1][y2], y* in 0..2 (three identical sequences == three triplets, so y <= 2 WLOG)
    uint8_t f[5][2][3][3];
};

inline void sh_gen_suit(ShantenHostTables& H) {
    H.suit.assign(SH_SUIT_ENTRIES, 0);
    size_t leaf = 0;
    SuitDP stack[10];
    for (auto& k : stack[0].f)
        for (auto& p : k)
            for (auto& a : p)
                for (auto& b : a) b = 99;
    stack[0].f[0][0][0][0] = 0;
    int digits[9] = {0};
    // iterative DFS in lexicographic order (digit 0..4 per rank, running sum <= 14)
    struct Rec {
        int i, sum;
    };
    std::vector<Rec> st;
    // recursive lambda
    struct Gen {
        ShantenHostTables& H;
        size_t& leaf;
        SuitDP* stack;
        void go(int i, int sum) {
            if (i == 9) {
                uint64_t v = 0;
                for (int p = 0; p < 2; p++)
                    for (int k = 0; k < 5; k++) {
                        uint8_t c = stack[9].f[k][p][0][0];
                        if (c > 14) c = 15;
                        v |= (uint64_t)c << (4 * (p * 5 + k));
                    }
                H.suit[leaf++] = v;
                return;
            }
            for (int c = 0; c <= 4 && sum + c <= 14; c++) {
                SuitDP& in = stack[i];
                SuitDP& out = stack[i + 1];
                for (auto& k : out.f)
                    for (auto& p : k)
                        for (auto& a : p)
                            for (auto& b : a) b = 99;
                for (int k = 0; k < 5; k++)
                    for (int p = 0; p < 2; p++)
                        for (int y1 = 0; y1 < 3; y1++)
                            for (int y2 = 0; y2 < 3; y2++) {
                                int base = in.f[k][p][y1][y2];
                                if (base >= 99) continue;
                                for (int x = 0; x <= 1; x++)
                                    for (int z = 0; z <= 1 - p; z++)
                                        for (int y = 0; y <= (i <= 6 ? 2 : 0); y++) {
                                            int t = 3 * x + 2 * z + y + y1 + y2;
                                            if (t > 4) continue;
                                            int k2 = k + x + y;
                                            if (k2 > 4) continue;
                                            int cost = base + (t > c ? t - c : 0);
                                            uint8_t& dst = out.f[k2][p + z][y][y1];
                                            if (cost < dst) dst = (uint8_t)cost;
                                        }
                            }
                go(i + 1, sum + c);
            }
        }
    } gen{H, leaf, stack};
    (void)digits;
    (void)st;
    gen.go(0, 0);
}

inline void sh_gen_honor(ShantenHostTables& H) {
    H.honor.assign(SH_HONOR_ENTRIES, 0);
    size_t leaf = 0;
    uint8_t stack[8][5][2];
    for (auto& k : stack[0])
        for (auto& p : k) p = 99;
    stack[0][0][0] = 0;
    struct Gen {
        ShantenHostTables& H;
        size_t& leaf;
        uint8_t (*stack)[5][2];
        void go(int i, int sum) {
            if (i == 7) {
                uint64_t v = 0;
                for (int p = 0; p < 2; p++)
                    for (int k = 0; k < 5; k++) {
                        uint8_t c = stack[7][k][p];
                        if (c > 14) c = 15;
                        v |= (uint64_t)c << (4 * (p * 5 + k));
                    }
                H.honor[leaf++] = v;
                return;
            }
            for (int c = 0; c <= 4 && sum + c <= 14; c++) {
                for (int k = 0; k < 5; k++)
                    for (int p = 0; p < 2; p++) stack[i + 1][k][p] = 99;
                for (int k = 0; k < 5; k++)
                    for (int p = 0; p < 2; p++) {
                        int base = stack[i][k][p];
                        if (base >= 99) continue;
                        for (int x = 0; x <= 1; x++)
                            for (int z = 0; z <= 1 - p; z++) {
                                int t = 3 * x + 2 * z;
                                if (t > 4 || k + x > 4) continue;
                                int cost = base + (t > c ? t - c : 0);
                                uint8_t& dst = stack[i + 1][k + x][p + z];
                                if (cost < dst) dst = (uint8_t)cost;
                            }
                    }
                go(i + 1, sum + c);
            }
        }
    } gen{H, leaf, stack};
    gen.go(0, 0);
}

// sh_rank's sum over the ranks [i0, i1) of a word, with its clamps, from running sum s0: (partial rank, running sum behind i1)
inline void sh_rank_part(const std::vector<uint32_t>& T, uint32_t word, int i0, int i1, uint32_t s0, uint32_t& part, uint32_t& s_out) {
    uint32_t h = 0, s = s0;
    for (int i = i0; i < i1; i++) {
        uint32_t c = (word >> (3 * (i - i0))) & 7u;
        if (c > 4u) c = 4u;
        if (s + c > 14u) c = 14u - s;
        h += T[((size_t)i * 15 + s) * 5 + c];
        s += c;
    }
    part = h;
    s_out = s;
}
inline void sh_gen_r2(ShantenHostTables& H) {
    H.r2.assign(SH_R2_WORDS, 0);
    for (uint32_t w = 0; w < 32768u; w++) {
        uint32_t part, s;
        sh_rank_part(H.rank9, w, 0, 5, 0, part, s);
        H.r2[SH_R2_HI9 + w] = part | (s << 20);
        sh_rank_part(H.rank7, w, 0, 5, 0, part, s);
        H.r2[SH_R2_HI7 + w] = part | (s << 20);
    }
    for (uint32_t s0 = 0; s0 < 15u; s0++) {
        uint32_t part, s;
        for (uint32_t w = 0; w < 4096u; w++) {
            sh_rank_part(H.rank9, w, 5, 9, s0, part, s);
            H.r2[SH_R2_LO9 + s0 * 4096u + w] = part;
        }
        for (uint32_t w = 0; w < 64u; w++) {
            sh_rank_part(H.rank7, w, 5, 7, s0, part, s);
            H.r2[SH_R2_LO7 + s0 * 64u + w] = part;
        }
    }
    H.v6.assign(H.suit.size() + H.honor.size(), 0);
    for (size_t i = 0; i < H.v6.size(); i++) {
        const uint64_t v = i < H.suit.size() ? H.suit[i] : H.honor[i - H.suit.size()];
        uint64_t o = 0;
        for (int p = 0; p < 2; p++)
            for (int k = 0; k < 5; k++) o |= ((v >> (4 * (p * 5 + k))) & 15ull) << (32 * p + 6 * k);
        H.v6[i] = o;
    }
}

inline const ShantenHostTables& shanten_host_tables() {
    static ShantenHostTables H;
    static bool done = false;
    if (!done) {
        sh_rank_table(9, H.rank9);
        sh_rank_table(7, H.rank7);
        sh_gen_suit(H);
        sh_gen_honor(H);
        sh_gen_r2(H);
        done = true;
    }
    return H;
}

// ---------------------------------------------------------------- device lookup
__device__ __forceinline__ uint32_t sh_rank(uint32_t x, int n, const uint32_t* T) {  // x: n fields of 3 bits
    uint32_t h = 0, s = 0;
    for (int i = 0; i < n; i++) {
        uint32_t c = (x >> (3 * i)) & 7u;
        if (c > 4u) c = 4u;
        if (s + c > 14u) c = 14u - s;
        h += T[(i * 15 + s) * 5 + c];
        s += c;
    }
    return h;
}
// (min,+) merge of two packed cost vectors (idx = p*5 + k, 4 bits each, 15 = infeasible)
__device__ __forceinline__ uint64_t sh_merge(uint64_t a, uint64_t b) {
    uint64_t out = 0;
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int k = 0; k < 5; k++) {
            uint32_t best = 15;
#pragma unroll
            for (int p1 = 0; p1 <= p; p1++)
#pragma unroll
                for (int k1 = 0; k1 <= k; k1++) {
                    uint32_t va = (uint32_t)(a >> (4 * (p1 * 5 + k1))) & 15u;
                    uint32_t vb = (uint32_t)(b >> (4 * ((p - p1) * 5 + (k - k1)))) & 15u;
                    uint32_t v = va + vb;
                    best = v < best ? v : best;
                }
            out |= (uint64_t)best << (4 * (p * 5 + k));
        }
    return out;
}
// shanten.rs:186-196
__device__ __forceinline__ int sh_normal(const PH& h, int m, const ShantenTables& T) {
    uint64_t a = T.suit[sh_rank(h.a, 9, T.rank9)];
    uint64_t b = T.suit[sh_rank(h.b, 9, T.rank9)];
    uint64_t c = T.suit[sh_rank(h.c, 9, T.rank9)];
    uint64_t d = T.honor[sh_rank(h.d, 7, T.rank7)];
    uint64_t r = sh_merge(sh_merge(a, b), sh_merge(c, d));
    if (m > 4) m = 4;
    int rep = (int)((r >> (4 * (5 + m))) & 15u);
    return rep - 1;
}
// shanten.rs:198-211 ; sanma: 2m-8m skipped (shanten.rs:437-452) — they are absent from a sanma hand anyway
__device__ __forceinline__ int sh_chiitoi(PH h, bool sanma) {
    if (sanma) h.a &= (7u | (7u << 24));
    // kinds = number of non-empty fields (bit0|bit1|bit2 of each 3-bit field)
    int kinds = __popc((h.a | (h.a >> 1) | (h.a >> 2)) & O9_1) + __popc((h.b | (h.b >> 1) | (h.b >> 2)) & O9_1) +
                __popc((h.c | (h.c >> 1) | (h.c >> 2)) & O9_1) + __popc((h.d | (h.d >> 1) | (h.d >> 2)) & O7_1);
    // count >= 2  <=> bit1 or bit2 of the field
    int pairs = __popc(((h.a >> 1) | (h.a >> 2)) & O9_1) + __popc(((h.b >> 1) | (h.b >> 2)) & O9_1) +
                __popc(((h.c >> 1) | (h.c >> 2)) & O9_1) + __popc(((h.d >> 1) | (h.d >> 2)) & O7_1);
    int red = kinds < 7 ? 7 - kinds : 0;
    return 7 - pairs + red - 1;
}
// shanten.rs:213-226
__device__ __forceinline__ int sh_kokushi(const PH& h) {
    const uint32_t T9 = 1u | (1u << 24);  // bit0 of the 1 and 9 fields
    const uint32_t any_a = h.a | (h.a >> 1) | (h.a >> 2), any_b = h.b | (h.b >> 1) | (h.b >> 2);
    const uint32_t any_c = h.c | (h.c >> 1) | (h.c >> 2), any_d = h.d | (h.d >> 1) | (h.d >> 2);
    const int kinds = __popc(any_a & T9) + __popc(any_b & T9) + __popc(any_c & T9) + __popc(any_d & O7_1);
    const uint32_t two = (((h.a >> 1) | (h.a >> 2)) & T9) | (((h.b >> 1) | (h.b >> 2)) & T9) | (((h.c >> 1) | (h.c >> 2)) & T9) |
                         (((h.d >> 1) | (h.d >> 2)) & O7_1);  // a terminal kind held at least twice
    return 14 - kinds - (two ? 1 : 0) - 1;
}
// shanten.rs:407-435: relocate 1m / 9m counts into empty honor slots (3P)
__device__ __forceinline__ PH sh_relocate_3p(const PH& h) {
    // branch-free form of the reference's scan: 1m takes the first empty honor slot, 9m the next one; without a slot the count stays
    PH t = h;
    uint32_t E = ~(h.d | (h.d >> 1) | (h.d >> 2)) & O7_1;   // bit 3 * slot of every empty honor slot
    const uint32_t m0 = h.a & 7u, m1 = (h.a >> 24) & 7u;
    t.a &= ~(7u | (7u << 24));
    if (m0 != 0u) {
        if (E != 0u) { t.d |= m0 << (__ffs((int)E) - 1); E &= E - 1u; }
        else t.a |= m0;
    }
    if (m1 != 0u) {
        if (E != 0u) t.d |= m1 << (__ffs((int)E) - 1);
        else t.a |= m1 << 24;
    }
    return t;
}
// shanten.rs:228-241 (4P) / :454-468 (3P)
__device__ __forceinline__ int sh_shanten(const PH& h, int len_div3, bool sanma, const ShantenTables& T) {
    int s = sh_normal(sanma ? sh_relocate_3p(h) : h, len_div3, T);
    if (s <= 0 || len_div3 < 4) return s;
    int c = sh_chiitoi(h, sanma);
    s = c < s ? c : s;
    if (s > 0) {
        int k = sh_kokushi(h);
        s = k < s ? k : s;
    }
    return s;
}

// entry (p,k) of the (min,+) merge of two packed cost vectors, for a lane-local target (p,k); x/y are given as their
// p=0 / p=1 halves (five nibbles each).  Same capping at 15 as sh_merge.
__device__ __forceinline__ uint32_t sh_merge_entry(uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1, int p, int k) {
    uint32_t best = 15u;
#pragma unroll
    for (int p1 = 0; p1 < 2; p1++)
#pragma unroll
        for (int k1 = 0; k1 < 5; k1++) {
            uint32_t vx = ((p1 ? x1 : x0) >> (4 * k1)) & 15u;
            int pp = p - p1, kk = k - k1;
            uint32_t vy = ((pp ? y1 : y0) >> (4 * (kk & 7))) & 15u;
            uint32_t v = vx + vy;
            best = (pp >= 0 && kk >= 0 && v < best) ? v : best;
        }
    return best;
}
__device__ __forceinline__ uint32_t row_min16(uint32_t v) {  // minimum over a 16-lane row, in lane 15 of the row
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(99, (int)v, 0x111, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(99, (int)v, 0x112, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(99, (int)v, 0x114, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(99, (int)v, 0x118, 0xf, 0xf, false));
    return v;
}

// Wave-cooperative 4P shanten of ONE wave-uniform hand (used as a sound prefilter for the riichi probe).
// lane = 16*suit + rank computes its term of the perfect hash, one DPP row sum per suit, four table loads; the
// (min,+) merges are lane-parallel too: lanes 0..9 hold merge(a,b)[idx], lanes 16..25 the entry of merge(c,d) that
// pairs with it in the final entry (pair, m), and a row minimum finishes.
__device__ __forceinline__ int sh_shanten_wave(const PH& h, int len_div3, const ShantenTables& T, int lane) {
    const int q = lane >> 4, i = lane & 15;
    uint32_t val = 0;
    if (i < (q < 3 ? 9 : 7)) {
        uint32_t x = ph_get(h, q);
        uint32_t c = (x >> (3 * i)) & 7u;
        uint32_t s = (uint32_t)field_sum(x & ((1u << (3 * i)) - 1u));
        if (c > 4u) c = 4u;
        if (s > 14u) s = 14u;
        if (s + c > 14u) c = 14u - s;
        val = (q < 3 ? T.rank9 : T.rank7)[(i * 15 + s) * 5 + c];
    }
    val = row_sum16(val);
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)val, 15), r1 = (uint32_t)__builtin_amdgcn_readlane((int)val, 31);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)val, 47), r3 = (uint32_t)__builtin_amdgcn_readlane((int)val, 63);
    const uint64_t a = T.suit[r0], b = T.suit[r1], c = T.suit[r2], d = T.honor[r3];
    const int m = len_div3 > 4 ? 4 : len_div3;
    const bool upper = lane >= 16;
    const int p = i >= 5 ? 1 : 0, k = i - 5 * p;
    const bool valid = lane < 32 && i < 10 && k <= m;
    const uint64_t x = upper ? c : a, y = upper ? d : b;
    uint32_t e = 15u;
    if (valid)
        e = sh_merge_entry((uint32_t)x & 0xFFFFFu, (uint32_t)(x >> 20) & 0xFFFFFu, (uint32_t)y & 0xFFFFFu, (uint32_t)(y >> 20) & 0xFFFFFu,
                           upper ? 1 - p : p, upper ? m - k : k);
    const uint32_t partner = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane + 16) & 63) << 2, (int)e);  // value of lane + 16
    uint32_t t = (valid && !upper) ? e + partner : 99u;
    t = row_min16(t);
    uint32_t rep = (uint32_t)__builtin_amdgcn_readlane((int)t, 15);
    if (rep > 15u) rep = 15u;
    int sres = (int)rep - 1;
    if (sres <= 0 || len_div3 < 4) return sres;
    int ch = sh_chiitoi(h, false);
    sres = ch < sres ? ch : sres;
    if (sres > 0) {
        int kk = sh_kokushi(h);
        sres = kk < sres ? kk : sres;
    }
    return sres;
}

// ---- the suit vectors of a hand and their pair merges (the ukeire walk judges hands that differ from it in one or two suits)
struct ShBase {
    uint64_t v[4];   // packed cost vectors of man / pin / sou / honors
    uint64_t ab, cd; // merge(v0, v1), merge(v2, v3)
};
__device__ __forceinline__ uint64_t sh_vec(uint32_t word, int q, const ShantenTables& T) {
    return q < 3 ? T.suit[sh_rank(word, 9, T.rank9)] : T.honor[sh_rank(word, 7, T.rank7)];
}
// Up to four (min,+) merges at once, one per 16-lane row: lane i < 10 of a row computes entry i of merge(a, b) for the row's
// (row-uniform) inputs, the nibbles are OR-reduced over the row and handed to every lane of it.  ~100 VALU for four merges where
// sh_merge takes ~210 for one (every lane of a wave computing the same 45 terms).
__device__ __forceinline__ uint32_t sh_row_or16(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    return v;   // lane 15 of the row holds the OR
}
__device__ __forceinline__ uint64_t sh_merge_rows(uint64_t a, uint64_t b, int lane) {
    const int i = lane & 15;
    const int p = i >= 5 ? 1 : 0, k = i - 5 * p;
    uint32_t e = 0u;
    if (i < 10)
        e = sh_merge_entry((uint32_t)a & 0xFFFFFu, (uint32_t)(a >> 20) & 0xFFFFFu, (uint32_t)b & 0xFFFFFu, (uint32_t)(b >> 20) & 0xFFFFFu, p, k);
    uint32_t lo = i < 8 ? e << (4 * i) : 0u, hi = (i >= 8 && i < 10) ? e << (4 * (i - 8)) : 0u;
    lo = sh_row_or16(lo);
    hi = sh_row_or16(hi);
    return (uint64_t)lo | ((uint64_t)hi << 32);   // valid in lane 15 of every row
}
__device__ __forceinline__ uint64_t sh_rl64(uint64_t v, int src) {
    return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src) << 32);
}
// entry (pair, m) of merge(a, b) - the replacement number of a hand whose suits are split between a and b
__device__ __forceinline__ int sh_entry_pm(uint64_t a, uint64_t b, int m) {
    return (int)sh_merge_entry((uint32_t)a & 0xFFFFFu, (uint32_t)(a >> 20) & 0xFFFFFu, (uint32_t)b & 0xFFFFFu, (uint32_t)(b >> 20) & 0xFFFFFu, 1, m);
}
// calculate_effective_tiles(_3p)_with_discard ("eff") and calculate_best_ukeire(_3p) ("uke"), shanten.rs:265-405 /
// :488-626, for ONE wave-uniform hand: lane = drawn tile type (my_cnt / my_vis = this lane's held / visible count), the
// loop over discard candidates runs over held types.  eff on a 3n hand yields 0xFFFFFFFF (the reference asserts).
// Both quantities judge the same (discard d, draw t) pairs - "does t lower the shanten of the hand without d" - so one
// pass serves both, and the shanten after each discard is evaluated for all d at once (lane = d): 2 + #held-types
// per-lane table evaluations instead of 2 + 4 x #held-types for the two separate walks.
#ifndef RMJ_UKEIRE_DENSE
#define RMJ_UKEIRE_DENSE 1   // 1: the fast case runs the pair-dense walk of rmj_ukeire.hip.h (round 5); 0: the round-4 walk below (A/B)
#endif
__device__ inline void sh_ukeire_dense(const ShantenTables& T, const PH& h, uint32_t my_vis, bool sm, int lane, bool want_eff, bool want_uke,
                                       uint32_t& eff, uint32_t& uke, int cur_in, int* nsh_out, int* cur_out);
__device__ inline void sh_ukeire_both(const ShantenTables& T, const PH& h, uint32_t my_cnt, uint32_t my_vis, bool sm, int lane,
                                      bool want_eff, bool want_uke, uint32_t& eff, uint32_t& uke, int cur_in = -99, int* nsh_out = nullptr,
                                      int* cur_out = nullptr) {   // cur_out: the shanten of h itself (computed here unless the caller passed it as cur_in)
    const int t = lane;
    const bool t_ok = t < 34 && (!sm || t == 0 || t >= 8);  // SANMA_VALID_TILE_TYPES (shanten.rs:244-247)
    const int total = ph_total(h);
    // 3P (round 4): the same walk on the RELOCATED hand (shanten.rs:407-452: the counts of 1m / 9m move into empty honor slots, the manzu
    // suit is left empty) - exact while every hand of the walk relocates both terminals, i.e. while h leaves three honor slots free
    // (h - d + t then has two) and holds no 2m..8m; 1m / 9m then belong to the honor "suit" and a hand's honor word is that of its own
    // relocation.  Anything else takes the general walk below.
    const bool fast = !sm || (__popc(~(h.d | (h.d >> 1) | (h.d >> 2)) & O7_1) >= 3 && (h.a & ~(7u | (7u << 24))) == 0u);
    if (RMJ_UKEIRE_DENSE && fast && total <= 14) {
        sh_ukeire_dense(T, h, my_vis, sm, lane, want_eff, want_uke, eff, uke, cur_in, nsh_out, cur_out);
        return;
    }
    if (fast) {
        // every hand judged here is the hand h with one tile removed and / or one added, i.e. it differs from h in at most two suits
        const int q = t < 34 ? ((sm && (t == 0 || t == 8)) ? 3 : t_suit(t)) : 0;   // the suit whose vector the tile type changes
        auto word = [&](const PH& y, int qq) -> uint32_t {                            // suit qq of hand y as the tables see it
            if (!sm) return ph_get(y, qq);
            return qq == 0 ? 0u : (qq == 3 ? sh_relocate_3p(y).d : ph_get(y, qq));
        };
        auto finish = [&](int rep, const PH& x, int len_div3) -> int {                // shanten.rs:228-241 / :454-468
            int s0 = rep - 1;
            if (s0 <= 0 || len_div3 < 4) return s0;
            const int c = sh_chiitoi(x, sm);
            s0 = c < s0 ? c : s0;
            if (s0 > 0) { const int k = sh_kokushi(x); s0 = k < s0 ? k : s0; }
            return s0;
        };
        ShBase Bh;
        {   // sh_base_wave with the two pair merges lane-parallel (rows 0 and 1)
            const uint64_t mine = sh_vec(word(h, lane & 3), lane & 3, T);
#pragma unroll
            for (int k = 0; k < 4; k++) Bh.v[k] = sh_rl64(mine, k);
            const bool r0 = (lane >> 4) == 0;
            const uint64_t M = sh_merge_rows(r0 ? Bh.v[0] : Bh.v[2], r0 ? Bh.v[1] : Bh.v[3], lane);
            Bh.ab = sh_rl64(M, 15);
            Bh.cd = sh_rl64(M, 31);
        }
        const int m_h = total / 3 > 4 ? 4 : total / 3;
        const int cur = cur_in != -99 ? cur_in : finish(sh_entry_pm(Bh.ab, Bh.cd, m_h), h, total / 3);
        if (cur_out) *cur_out = cur;
        // O_q: the three other suits of h merged (row q computes O_q)
        uint64_t O_mine;
        {
            const int rw = lane >> 4;
            const uint64_t pa = rw == 0 ? Bh.v[1] : (rw == 1 ? Bh.v[0] : (rw == 2 ? Bh.v[3] : Bh.v[2]));
            const uint64_t ot = rw < 2 ? Bh.cd : Bh.ab;
            const uint64_t Om = sh_merge_rows(pa, ot, lane);
            const uint64_t O0 = sh_rl64(Om, 15), O1 = sh_rl64(Om, 31), O2 = sh_rl64(Om, 47), O3 = sh_rl64(Om, 63);
            O_mine = q == 0 ? O0 : (q == 1 ? O1 : (q == 2 ? O2 : O3));
        }
        const bool drawable = t_ok && ph_cnt(h, t) < 4;
        uint64_t vt = 0;          // vector of this lane's suit with the lane's type drawn
        PH hp = h;
        if (drawable) ph_add(hp, t);
        eff = 0;
        uke = 0;
        const bool eff13 = want_eff && total % 3 == 1;
        if (eff13) {
            if (drawable) vt = sh_vec(word(hp, q), q, T);
            const bool f = drawable && finish(sh_entry_pm(vt, O_mine, (total + 1) / 3 > 4 ? 4 : (total + 1) / 3), hp, (total + 1) / 3) < cur;
            eff = (uint32_t)__popcll(__ballot(f));
        }
        if (want_eff && total % 3 == 0) eff = 0xFFFFFFFFu;
        const bool eff_loop = want_eff && total % 3 == 2;
        if (!eff_loop && !want_uke) return;
        // Round 4: every hand of the walk is h - d + t.  (min,+) is associative, so its replacement number is ONE entry of a merge of two
        // vectors that are known before the pair is looked at:
        //   t in the suit of d            : entry(vec(suit of h - d + t), O_q)   O_q = the three other suits of h merged (per hand)
        //   t in the partner suit of d's  : entry(vec(suit of h + t),     W_d)   W_d = vec(suit of h - d) (+) the other half of h (per d)
        //   t in the other half           : entry(P_t,                    H_d)   P_t = vec(suit of h + t) (+) partner suit of h (per lane), H_d = vec(suit of h - d) (+) its partner suit (per d)
        // - one table lookup per same-suit pair and no merge per pair; the merges per hand / per d run lane-parallel (sh_merge_rows).
        const int m_loop = total / 3 > 4 ? 4 : total / 3;            // hands of the loop hold total tiles again (h - d + t)
        const int m_sub = (total - 1) / 3 > 4 ? 4 : (total - 1) / 3;
        int nsh_l = 127;
        uint64_t nv_d = 0;
        if (t < 34 && ph_cnt(h, t) > 0) {
            PH sub = h;
            ph_sub(sub, t);
            nv_d = sh_vec(word(sub, q), q, T);
            nsh_l = finish(sh_entry_pm(nv_d, O_mine, m_sub), sub, (total - 1) / 3);
        }
        if (nsh_out) *nsh_out = nsh_l;
        uint64_t cand = __ballot(nsh_l <= cur);
        if (!cand) return;
        uint64_t Pt = 0;
        if (drawable && !eff13) vt = sh_vec(word(hp, q), q, T);   // (only hands with a discard that keeps the shanten get here)
        if (drawable) Pt = sh_merge(vt, q == 0 ? Bh.v[1] : (q == 1 ? Bh.v[0] : (q == 2 ? Bh.v[3] : Bh.v[2])));
        while (cand) {
            const int d = __ffsll((long long)cand) - 1;
            cand &= cand - 1ull;
            const int nsh = __builtin_amdgcn_readlane(nsh_l, d);
            const int qd = (sm && (d == 0 || d == 8)) ? 3 : t_suit(d);
            PH sub = h;
            ph_sub(sub, d);
            const uint64_t V = sh_rl64(nv_d, d);
            // row 0: H_d, row 1: W_d
            const uint64_t pd = qd == 0 ? Bh.v[1] : (qd == 1 ? Bh.v[0] : (qd == 2 ? Bh.v[3] : Bh.v[2]));
            const uint64_t xd = qd < 2 ? Bh.cd : Bh.ab;
            const uint64_t HW = sh_merge_rows(V, (lane >> 4) == 0 ? pd : xd, lane);
            const uint64_t Hd = sh_rl64(HW, 15), Wd = sh_rl64(HW, 31);
            bool f = false;
            if (t_ok && ph_cnt(sub, t) < 4) {
                PH x = sub;
                ph_add(x, t);
                uint64_t a, b;
                if (q == qd) { a = sh_vec(word(x, q), q, T); b = O_mine; }
                else if ((q ^ 1) == qd) { a = vt; b = Wd; }
                else { a = Pt; b = Hd; }
                f = finish(sh_entry_pm(a, b, m_loop), x, total / 3) < nsh;
            }
            const uint64_t fb = __ballot(f);
            if (eff_loop) {
                const uint32_t e = (uint32_t)__popcll(fb);
                eff = e > eff ? e : eff;
            }
            if (want_uke) {
                int held = (int)my_cnt - (t == d ? 1 : 0);
                int rem = 4 - (int)my_vis;
                rem = rem < 0 ? 0 : rem;
                rem -= held;
                const uint32_t w = f ? (uint32_t)(rem < 0 ? 0 : rem) : 0u;
                const uint32_t u = (uint32_t)__popcll(__ballot(w & 1u)) + 2u * (uint32_t)__popcll(__ballot(w & 2u)) +
                                   4u * (uint32_t)__popcll(__ballot(w & 4u));
                uke = u > uke ? u : uke;
            }
        }
        return;
    }
    const int cur = cur_in != -99 ? cur_in : sh_shanten(h, total / 3, sm, T);  // the caller may know it already
    if (cur_out) *cur_out = cur;
    // does drawing this lane's type lower the shanten of `base` (a 3n+1 hand)?
    auto improves = [&](const PH& base, int base_total, int base_sh) -> bool {
        if (!(t_ok && ph_cnt(base, t) < 4)) return false;
        PH x = base;
        ph_add(x, t);
        return sh_shanten(x, (base_total + 1) / 3, sm, T) < base_sh;
    };
    eff = 0;
    uke = 0;
    if (want_eff && total % 3 == 1) eff = (uint32_t)__popcll(__ballot(improves(h, total, cur)));
    if (want_eff && total % 3 == 0) eff = 0xFFFFFFFFu;
    const bool eff_loop = want_eff && total % 3 == 2;
    if (!eff_loop && !want_uke) return;
    // shanten after discarding this lane's type (127: not held)
    int nsh_l = 127;
    if (lane < 34 && ph_cnt(h, lane) > 0) {
        PH sub = h;
        ph_sub(sub, lane);
        nsh_l = sh_shanten(sub, (total - 1) / 3, sm, T);
    }
    if (nsh_out) *nsh_out = nsh_l;  // lane = type: shanten after discarding one tile of it (127: not held)
    uint64_t cand = __ballot(nsh_l <= cur);
    while (cand) {
        const int d = __ffsll((long long)cand) - 1;
        cand &= cand - 1ull;
        const int nsh = __builtin_amdgcn_readlane(nsh_l, d);
        PH sub = h;
        ph_sub(sub, d);
        const bool f = improves(sub, total - 1, nsh);
        const uint64_t fb = __ballot(f);
        if (eff_loop) {
            const uint32_t e = (uint32_t)__popcll(fb);
            eff = e > eff ? e : eff;
        }
        if (want_uke) {  // remaining = 4 - visible - held (both saturating), held counted after the discard
            int held = (int)my_cnt - (t == d ? 1 : 0);
            int rem = 4 - (int)my_vis;
            rem = rem < 0 ? 0 : rem;
            rem -= held;
            const uint32_t w = f ? (uint32_t)(rem < 0 ? 0 : rem) : 0u;   // 0..4
            const uint32_t u = (uint32_t)__popcll(__ballot(w & 1u)) + 2u * (uint32_t)__popcll(__ballot(w & 2u)) +
                               4u * (uint32_t)__popcll(__ballot(w & 4u));
            uke = u > uke ? u : uke;
        }
    }
}
// one of the two (rmj_effective_tiles: mode 0, rmj_best_ukeire: mode 1)
__device__ inline uint32_t sh_ukeire_wave(const ShantenTables& T, const PH& h, uint32_t my_cnt, uint32_t my_vis, bool sm, int mode, int lane) {
    uint32_t eff, uke;
    sh_ukeire_both(T, h, my_cnt, my_vis, sm, lane, mode == 0, mode == 1, eff, uke);
    return mode == 0 ? eff : uke;
}

}  // namespace rmj

#include "rmj_ukeire.hip.h"
