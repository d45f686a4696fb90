// Feature encoder (row A14): Observation.encode() -> 74 x 34 f32, channel-major
// (reference: observation/python.rs:457-806, spec docs/FEATURE_ENCODING.md:8-82).
// One wavefront per (game, seat): the tensor is assembled in LDS (lane = tile type for the per-tile channels,
// wave-uniform scalars broadcast over the 34 columns) and streamed out as 629 coalesced 16-byte stores.
#pragma once
#include "rmj_common.hip.h"

namespace rmj {

#define ENC_CH 74
#define ENC_W4 34
#define ENC_W3 27 /* observation_3p/helpers.rs:3 */
#define ENC_FLOATS (ENC_CH * ENC_W4)
#define ENC_FLOATS3 (ENC_CH * ENC_W3)
// observation_3p/helpers.rs:7-15: tile34 -> compact column (2m-8m have none)
template <bool SANMA>
__device__ __forceinline__ int enc_col(int t34) {
    if (!SANMA) return t34 < 34 ? t34 : -1;
    return t34 == 0 ? 0 : ((t34 >= 8 && t34 < 34) ? t34 - 7 : -1);
}

// observation/helpers.rs:24-50 — takes a 136-id, returns a 136-id (copy 0 of the next type)
__device__ __forceinline__ int enc_next_tile136(int tile) {
    int tt = tile / 4;
    if (tt < 27) {
        int num = tt % 9;
        return ((tt - num) + (num == 8 ? 0 : num + 1)) * 4;
    }
    if (tt < 31) return (27 + (tt - 27 + 1) % 4) * 4;
    if (tt < 34) return (31 + (tt - 31 + 1) % 3) * 4;
    return tile & 0xFF;
}

// waits of the observation (state/mod.rs:220-225): 13-tile hands only; uses the cache when valid, never writes it
__device__ __forceinline__ uint64_t enc_waits(const PState& P, int lane) {
    if (P.hand_len + 3 * P.n_melds != 13) return 0ull;
    if (P.flags & PF_WAITS_VALID) return P.waits13;
    PH h = build_ph(P);
    return wave_waits(h, lane);
}

// observation_3p/helpers.rs:38-47
__device__ __forceinline__ int enc_next_tile136_sanma(int tile) {
    int tt = tile / 4;
    if (tt == 0) return 32;
    if (tt == 8) return 0;
    if (tt < 8) return tile & 0xFF;
    return enc_next_tile136(tile);
}

template <bool SANMA>
__device__ inline void encode_seat(const GState& S, int pid, float* buf, int lane) {
    constexpr int ENC_W = SANMA ? ENC_W3 : ENC_W4;
    constexpr int NPP = SANMA ? 3 : 4;
    auto enc_bcast = [&](float* b, int ch, float v, int l) {
        if (l < ENC_W) b[ch * ENC_W + l] = v;
    };
    auto put = [&](int ch, int t34) {  // scatter one cell (skips tiles without a column)
        int col = enc_col<SANMA>(t34);
        if (col >= 0) buf[ch * ENC_W + col] = 1.0f;
    };
    const int my34 = SANMA ? (lane == 0 ? 0 : lane + 7) : lane;  // tile type of this lane's column
    for (int i = lane; i < ENC_CH * ENC_W; i += 64) buf[i] = 0.0f;
    wave_sync();
    const PState& P = S.p[pid];
    // 1-2. hand counts + red (ch 0-4)
    if (lane < ENC_W) {
        int c = 0;
        bool red = false;
        for (int j = 0; j < P.hand_len; j++) {
            int t = P.hand[j];
            if ((t >> 2) == my34) {
                c++;
                red = red || is_aka(t);
            }
        }
        if (c >= 1) buf[0 * ENC_W + lane] = 1.0f;
        if (c >= 2) buf[1 * ENC_W + lane] = 1.0f;
        if (c >= 3) buf[2 * ENC_W + lane] = 1.0f;
        if (c >= 4) buf[3 * ENC_W + lane] = 1.0f;
        if (red) buf[4 * ENC_W + lane] = 1.0f;
    }
    // 3. own melds (ch 5-8), 4. dora indicators (ch 9)
    if (lane < 16) {
        int m = lane >> 2, k = lane & 3;
        if (m < P.n_melds && k < ((P.meld_type[m] >= RMJ_MELD_DAIMINKAN) ? 4 : 3)) put(5 + m, P.meld_tiles[m][k] >> 2);
    }
    if (lane < S.n_dora) put(9, S.dora[lane] >> 2);
    // 5-6. recent discards: self ch10-13 (+64-67), opponents ch14-25 (+68-69 for the first)
    if (lane < 16) {
        int rel = lane >> 2, j = lane & 3;
        if (rel < NPP) {
            const PState& Q = S.p[(pid + rel) % NPP];
            int n = Q.n_discards;
            if (j < n) put(10 + rel * 4 + j, Q.discards[n - 1 - j] >> 2);
        }
    }
    if (lane < 4) {
        int n = P.n_discards;
        if (4 + lane < n) put(64 + lane, P.discards[n - 1 - (4 + lane)] >> 2);
    }
    if (lane < 2) {
        const PState& Q = S.p[(pid + 1) % NPP];
        int n = Q.n_discards;
        if (4 + lane < n) put(68 + lane, Q.discards[n - 1 - (4 + lane)] >> 2);
    }
    // wave-uniform scalars
    int tiles_used = P.hand_len + S.n_dora;
    int32_t my_score = P.score;
    int rank = 0;
    for (int q = 0; q < NPP; q++) {
        const PState& Q = S.p[q];
        tiles_used += Q.n_discards;
        for (int m = 0; m < Q.n_melds; m++) tiles_used += (Q.meld_type[m] >= RMJ_MELD_DAIMINKAN) ? 4 : 3;
        rank += (Q.score > my_score);
    }
    int tiles_left = (SANMA ? 108 : 136) - tiles_used;
    if (tiles_left < 0) tiles_left = 0;
    enc_bcast(buf, 30, (float)tiles_left / 70.0f, lane);
    for (int rel = 0; rel < NPP; rel++) {
        const PState& Q = S.p[(pid + rel) % NPP];
        enc_bcast(buf, 26 + rel, (float)Q.n_discards / 24.0f, lane);
        if (Q.flags & PF_RIICHI_DECLARED) enc_bcast(buf, 31 + rel, 1.0f, lane);
        int32_t sc = Q.score;
        int32_t s1 = sc < 0 ? 0 : (sc > 100000 ? 100000 : sc);
        int32_t s2 = sc < 0 ? 0 : (sc > 30000 ? 30000 : sc);
        enc_bcast(buf, 39 + rel, (float)s1 / 100000.0f, lane);
        enc_bcast(buf, 43 + rel, (float)s2 / 30000.0f, lane);
        enc_bcast(buf, 59 + rel, (float)Q.n_melds / 4.0f, lane);
    }
    wave_sync();
    // 10. winds (ch 35-36)
    {
        int rw = S.round_wind;
        if (lane == 0 && 27 + rw < 34) put(35, 27 + rw);
        int seat = (pid + NPP - S.oya) % NPP;
        if (lane == 0) put(36, 27 + seat);
    }
    enc_bcast(buf, 37, (float)S.honba / 10.0f, lane);
    enc_bcast(buf, 38, (float)S.riichi_sticks / 5.0f, lane);
    // 14-15. waits / tenpai (ch 47-48)
    uint64_t W = enc_waits(P, lane);
    if (lane < ENC_W && ((W >> my34) & 1ull)) buf[47 * ENC_W + lane] = 1.0f;
    enc_bcast(buf, 48, W != 0ull ? 1.0f : 0.0f, lane);
    if (rank < NPP) enc_bcast(buf, 49 + rank, 1.0f, lane);
    enc_bcast(buf, 53, (float)S.kyoku_idx / 8.0f, lane);
    enc_bcast(buf, 54, ((float)S.round_wind * 4.0f + (float)S.kyoku_idx) / 7.0f, lane);
    // 19. dora counts (ch 55-58) and 21. tiles seen (ch 63): lane = tile type accumulates, then reduce for dora
    {
        int seen = 0;
        uint32_t dmask_lo = 0, dmask_hi = 0;  // multiset of dora types as counts per type would need 34 counters:
        // dora counting is per indicator (duplicates count twice), so loop over indicators explicitly below.
        (void)dmask_lo; (void)dmask_hi;
        int dcount[4] = {0, 0, 0, 0};
        // NOTE: the dora count is taken over tile TYPES (34-wide, also types without a column in 3P); lanes >= ENC_W
        // cover nothing in 4P, and in 3P the types 1..7 (2m-8m) cannot occur in a sanma game.
        for (int q = 0; q < NPP; q++) {
            const PState& Q = S.p[q];
            int mine = 0;  // tiles of this lane's type visible for player q (melds + discards [+ own hand])
            for (int m = 0; m < Q.n_melds; m++) {
                int nt = (Q.meld_type[m] >= RMJ_MELD_DAIMINKAN) ? 4 : 3;
                for (int k = 0; k < nt; k++) mine += ((Q.meld_tiles[m][k] >> 2) == my34);
            }
            for (int j = 0; j < Q.n_discards; j++) mine += ((Q.discards[j] >> 2) == my34);
            seen += mine;
            if (q == pid) {
                int hc = 0;
                for (int j = 0; j < P.hand_len; j++) hc += ((P.hand[j] >> 2) == my34);
                seen += hc;
                mine += hc;
            }
            // contribution of this tile type to q's dora count = mine * (#indicators whose dora type is this type)
            int mult = 0;
            for (int k = 0; k < S.n_dora; k++)
                mult += (((SANMA ? enc_next_tile136_sanma(S.dora[k]) : enc_next_tile136(S.dora[k])) >> 2) == my34);
            int contrib = (lane < ENC_W) ? mine * mult : 0;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) contrib += __shfl_xor(contrib, off, 64);
            dcount[q] = contrib & 0xFF;  // u8 accumulator in the reference
        }
        for (int k = 0; k < S.n_dora; k++) seen += ((S.dora[k] >> 2) == my34);
        if (lane < ENC_W) buf[63 * ENC_W + lane] = (float)(seen & 0xFF) / 4.0f;
        for (int rel = 0; rel < NPP; rel++) {
            int q = (pid + rel) % NPP;
            int d = q == 0 ? dcount[0] : (q == 1 ? dcount[1] : (q == 2 ? dcount[2] : dcount[3]));
            enc_bcast(buf, 55 + rel, (float)d / 12.0f, lane);
        }
    }
    // ch 70-73 stay 0: tsumogiri_flags is never filled (observation/mod.rs:105)
    wave_sync();
}

}  // namespace rmj
