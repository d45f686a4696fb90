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

// `hist`: 5 x 36 u32 of LDS scratch (type histograms of the four seats' visible tiles and of the own hand)
#define ENC_HIST_WORDS (5 * 36)
// x / C for a non-negative integer x <= XMAX, bit-equal to the IEEE division the reference performs ((x as f32) / C): the
// product with the rounded reciprocal, corrected once with the exact remainder (two FMAs instead of the ~12 instructions of
// a division, and the encoder divides ~35 times per observation).  The identity holds for every integer of the stated range:
// checked exhaustively with exact rational arithmetic by tests/test_exact_division.py; larger x take the division.
template <int C, int XMAX>
__device__ __forceinline__ float enc_div(int x) {
    constexpr float c = (float)C, rc = 1.0f / (float)C;
    const float xf = (float)x;
    if (x > XMAX) return xf / c;   // (wave-uniform operands: a poked state, never a played one)
    const float q = __fmul_rn(xf, rc);
    const float r = __builtin_fmaf(-q, c, xf);
    return __builtin_fmaf(r, rc, q);
}
// ---- output sinks of encode_seat: ONE body produces the 74 channels, the sink decides how they are staged.
// EncFloatSink: the channels [ch_lo, ch_lo + ch_n) as floats in buf[0 .. ch_n * W) (the extended encoder stages the whole
// tensor; windows exist for experiments).  Every channel of Observation.encode() is either a 0/1 pattern over the tile
// columns (hand counts, melds, discards, dora, waits, winds), one value broadcast over all columns (counts, scores, flags),
// or - channel 63 only - a per-column count / 4: EncByteSink stages a one-byte code per cell.
template <int W>
struct EncFloatSink {
    float* buf;
    int ch_lo, ch_n, lane;
    __device__ __forceinline__ int slot(int ch) const { const int k = ch - ch_lo; return (k >= 0 && k < ch_n) ? k : -1; }
    __device__ __forceinline__ bool wants(int ch) const { return slot(ch) >= 0; }
    __device__ __forceinline__ void zero() const {   // 16-byte LDS stores (buf is 16-byte aligned), then the odd floats
        const int n4 = (ch_n * W) >> 2;
        for (int i = lane; i < n4; i += 64) reinterpret_cast<float4*>(buf)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < ((ch_n * W) & 3)) buf[4 * n4 + lane] = 0.0f;
    }
    __device__ __forceinline__ void put(int ch, int col) const {   // one cell = 1
        const int k = slot(ch);
        if (k >= 0 && col >= 0) buf[k * W + col] = 1.0f;
    }
    __device__ __forceinline__ void bcast(int ch, float v) const {   // every column = v
        const int k = slot(ch);
        if (k >= 0 && lane < W) buf[k * W + lane] = v;
    }
    __device__ __forceinline__ void bcast_always(int ch, float v) const { bcast(ch, v); }
    __device__ __forceinline__ void cell(int ch, float v) const {   // this lane's column = v (1.0 except for channel 63)
        const int k = slot(ch);
        if (k >= 0 && lane < W) buf[k * W + lane] = v;
    }
    __device__ __forceinline__ void cell_quarters(int ch, int c) const { cell(ch, (float)c / 4.0f); }
    __device__ __forceinline__ void flush() const {}
};
// EncByteSink: one byte per cell of the 74 x W tensor, a code into a table of floats: 0 -> 0.0, 1 -> 1.0, 2 + k -> the value
// broadcast over channel 26 + k (k < 37), 40 + c -> c / 4 (channel 63, c <= 203 even in a poked state).  2.5 KB instead of
// 10 KB per observation, and the way out stays a plain stream of 16-byte stores (enc_emit_bytes).  The table holds the codes below 64
// (c <= 23; a played state has c <= 4); an observation with a larger count (`big`) leaves through the emit loop that decodes such codes
// arithmetically (the same float: c x 0.25 is exact).
#define ENC_LUT 64        /* round 5: 64 entries (256 B; rounds 3-4: 256 entries) - codes from 64 on (a per-column count above 23: poked states only) are decoded arithmetically */
#define ENC_Q_BASE 40     /* code of a per-column count c: ENC_Q_BASE + c */
// Round 5 (an alternative, off by default - RMJ_ENC_IMAGE): 23 (4P) / 18 (3P) of the broadcast channels are written for EVERY observation - counts, scores, round numbers; only the riichi
// flags and the rank one-hot depend on the state - and their cells hold the channel's code whatever the value is (the value sits in the
// table).  The staging area therefore starts from a constant image with those rows filled in instead of zeros, and the sink only notes
// the value (bcast_always): one LDS byte store and its addressing less per channel.  One image per byte offset of `cells` in its 16-byte
// aligned buffer (0..3: the row's distance to the next 16-byte boundary of the output).
template <int W>
struct EncCellImage {
    static constexpr int NV = (ENC_CH * W + 4 + 15) / 16;
    uint32_t w[4][NV * 4];
};
template <int W>
constexpr bool enc_always_bcast(int ch) {
    constexpr int NPP = W == ENC_W3 ? 3 : 4;
    if (ch == 30 || ch == 37 || ch == 38 || ch == 48 || ch == 53 || ch == 54) return true;
    constexpr int per_seat[5] = {26, 39, 43, 55, 59};
    for (int k = 0; k < 5; k++)
        if (ch >= per_seat[k] && ch < per_seat[k] + NPP) return true;
    return false;
}
template <int W>
constexpr EncCellImage<W> make_enc_cell_image() {
    EncCellImage<W> t{};
    for (int s = 0; s < 4; s++)
        for (int ch = 26; ch <= 62; ch++)
            if (enc_always_bcast<W>(ch))
                for (int col = 0; col < W; col++) {
                    const int e = s + ch * W + col;
                    t.w[s][e >> 2] |= (uint32_t)(2 + ch - 26) << (8 * (e & 3));
                }
    return t;
}
__constant__ const EncCellImage<ENC_W4> g_enc_image4 = make_enc_cell_image<ENC_W4>();
__constant__ const EncCellImage<ENC_W3> g_enc_image3 = make_enc_cell_image<ENC_W3>();
#ifndef RMJ_ENC_IMAGE
#define RMJ_ENC_IMAGE 0   /* 1: the image (measured: the trainer loop's step + encode launch +1.5 %, the stand-alone encoders -2...-4 % - the image's loads sit in front of every observation; left off) */
#endif
#ifndef RMJ_ENC_EMIT_UNROLL
#define RMJ_ENC_EMIT_UNROLL 1
#endif
template <int W>
struct EncByteSink {
    uint8_t* cells;   // cell e of the tensor at cells[e]; cells + head is 4-byte aligned (head: see enc_emit_bytes)
    float* lut;       // [ENC_LUT]; entries 0, 1 and 40.. are filled once per block (enc_lut_init)
    int lane;
    float acc;        // lane k < 37 collects the value broadcast over channel 26 + k (-1 = none)
    bool big = false; // this lane staged a code outside the table
    __device__ __forceinline__ bool wants(int) const { return true; }
    __device__ __forceinline__ void zero() {
        uint4* z = reinterpret_cast<uint4*>(reinterpret_cast<uintptr_t>(cells) & ~(uintptr_t)15);   // the 16-byte aligned raw buffer
#if RMJ_ENC_IMAGE
        const int shift = (int)(reinterpret_cast<uintptr_t>(cells) & 3u);
        const uint4* img;
        if constexpr (W == ENC_W3) img = reinterpret_cast<const uint4*>(g_enc_image3.w[shift]);
        else img = reinterpret_cast<const uint4*>(g_enc_image4.w[shift]);
        for (int i = lane; i < (ENC_CH * W + 4 + 15) / 16; i += 64) z[i] = img[i];
#else
        for (int i = lane; i < (ENC_CH * W + 4 + 15) / 16; i += 64) z[i] = make_uint4(0u, 0u, 0u, 0u);
#endif
        acc = -1.0f;
        big = false;
    }
    __device__ __forceinline__ void bcast_always(int ch, float v) {   // a channel of enc_always_bcast: its cells are in the image already
#if RMJ_ENC_IMAGE
        if (lane == ch - 26) acc = v;
#else
        bcast(ch, v);
#endif
    }
    __device__ __forceinline__ void put(int ch, int col) const {
        if (col >= 0) cells[ch * W + col] = 1;
    }
    __device__ __forceinline__ void bcast(int ch, float v) {   // wave-uniform calls; every broadcast channel lies in 26..62
        if (lane == ch - 26) acc = v;
        if (lane < W) cells[ch * W + lane] = (uint8_t)(2 + ch - 26);
    }
    __device__ __forceinline__ void cell(int ch, float) const {   // this lane's column = 1
        if (lane < W) cells[ch * W + lane] = 1;
    }
    __device__ __forceinline__ void cell_quarters(int ch, int c) {   // this lane's column = c / 4
        if (lane < W) {
            cells[ch * W + lane] = (uint8_t)(ENC_Q_BASE + c);
            big = big || ENC_Q_BASE + c >= ENC_LUT;
        }
    }
    __device__ __forceinline__ void flush() const {
        if (lane < 37 && acc >= 0.0f) lut[2 + lane] = acc;
    }
};
__device__ __forceinline__ void enc_lut_init(float* lut, int lane) {
    for (int i = lane; i < ENC_LUT; i += 64) lut[i] = i < 2 ? (float)i : (i >= ENC_Q_BASE ? (float)(i - ENC_Q_BASE) * 0.25f : 0.0f);
}
// NCH x W floats from the byte-staged form, in 16-byte stores: `head` = 0..3 floats precede the first 16-byte boundary of the
// (4-byte aligned) `dst`; the cells are laid out so that cells + head is dword aligned
// big (any lane): the staging area holds codes outside the table - every code is decoded with the range check (rare: poked states)
template <int W, int NCH = ENC_CH>
__device__ __forceinline__ void enc_emit_bytes(float* dst, const uint8_t* cells, const float* lut, int lane, int head, bool big = false) {
    constexpr int N = NCH * W;
    const int body = (N - head) >> 2, tail0 = head + 4 * body;
    auto dec1 = [&](uint32_t code) -> float { return code < (uint32_t)ENC_LUT ? lut[code] : (float)((int)code - ENC_Q_BASE) * 0.25f; };
    if (lane < head) dst[lane] = dec1(cells[lane]);
    float4* d4 = reinterpret_cast<float4*>(dst + head);
    const uint32_t* c4 = reinterpret_cast<const uint32_t*>(cells + head);
    typedef float enc_v4f __attribute__((ext_vector_type(4)));
    auto st16 = [&](int i, const enc_v4f& v4) {
        // streaming (non-temporal) 16-byte stores: the tensor is written once and read by another kernel much later
#if defined(RMJ_ENC_NOSTORE)   /* experiment: the encoder's work without its stores */
        asm volatile("" :: "v"(v4), "v"(&d4[i]));
#elif defined(RMJ_ENC_NO_NT)
        *reinterpret_cast<enc_v4f*>(&d4[i]) = v4;
#else
        __builtin_nontemporal_store(v4, reinterpret_cast<enc_v4f*>(&d4[i]));
#endif
    };
    auto dec = [&](uint32_t w) -> enc_v4f { return enc_v4f{lut[w & 255u], lut[(w >> 8) & 255u], lut[(w >> 16) & 255u], lut[w >> 24]}; };
    int i = lane;
    const bool slow = __ballot(big) != 0ull;
#if RMJ_ENC_EMIT_UNROLL > 1
    // RMJ_ENC_EMIT_UNROLL stores per trip (A/B switch, measured slower in round 5: journal r05 section 11): the code words of all of them are
    // read first, then their table entries, then the stores leave back to back.  Only when every code is inside the table (`slow`: codes up to
    // 255 would index past the 64-entry table).
    constexpr int U = RMJ_ENC_EMIT_UNROLL;
    if (!slow) {
        for (; i + 64 * (U - 1) < body; i += 64 * U) {
            uint32_t w[U];
#pragma unroll
            for (int u = 0; u < U; u++) w[u] = c4[i + 64 * u];
            enc_v4f v[U];
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = dec(w[u]);
#pragma unroll
            for (int u = 0; u < U; u++) st16(i + 64 * u, v[u]);
        }
    }
#endif
    if (slow) {
        for (; i < body; i += 64) {
            const uint32_t w = c4[i];
            st16(i, enc_v4f{dec1(w & 255u), dec1((w >> 8) & 255u), dec1((w >> 16) & 255u), dec1(w >> 24)});
        }
    }
    for (; i < body; i += 64) st16(i, dec(c4[i]));
    if (lane < N - tail0) dst[tail0 + lane] = dec1(cells[tail0 + lane]);
}
// ext_base: the base block as encode_extended writes it (encode_base_into, observation/encode.rs:94-111, observation_3p/encode.rs:106-122):
// its "tiles left" does not count a meld's called tile twice (it is in the discards already); Observation.encode() itself
// (observation/python.rs:568-587) counts every meld tile.  The two differ in channel 30 only.
template <bool SANMA, class SINK>
__device__ inline void encode_seat_to(const GState& S, int pid, int lane, uint32_t* hist, SINK& o, bool first, bool ext_base = false) {
    constexpr int ENC_W = SANMA ? ENC_W3 : ENC_W4;
    constexpr int NPP = SANMA ? 3 : 4;
    auto enc_bcast = [&](float*, int ch, float v, int) { o.bcast(ch, v); };
    auto enc_always = [&](int ch, float v) { o.bcast_always(ch, v); };   // the channels of enc_always_bcast
    auto put = [&](int ch, int t34) { o.put(ch, enc_col<SANMA>(t34)); };   // scatter one cell (skips tiles without a column)
    auto cell = [&](int ch, float v) { o.cell(ch, v); };
    auto slot = [&](int ch) { return o.wants(ch) ? 0 : -1; };
    float* const buf = nullptr;
    (void)buf;
    const int my34 = SANMA ? (lane == 0 ? 0 : lane + 7) : lane;  // tile type of this lane's column
    o.zero();
    if (first)
        for (int i = lane; i < ENC_HIST_WORDS; i += 64) hist[i] = 0u;
    wave_sync();
    const PState& P = S.p[pid];
    // type histograms by LDS atomics, lane = tile slot: hist[q] = melds + discards of seat q, hist[4] = own hand
    if (first) {
        for (int q = 0; q < NPP; q++) {
            const PState& Q = S.p[q];
            if (lane < Q.n_discards) atomicAdd(&hist[q * 36 + (Q.discards[lane] >> 2)], 1u);
            if (lane < 16) {
                const int m = lane >> 2, k = lane & 3;
                if (m < Q.n_melds && k < ((Q.meld_type[m] >= RMJ_MELD_DAIMINKAN) ? 4 : 3)) atomicAdd(&hist[q * 36 + (Q.meld_tiles[m][k] >> 2)], 1u);
            }
        }
        if (lane < P.hand_len) atomicAdd(&hist[4 * 36 + (P.hand[lane] >> 2)], 1u);
    }
    if (lane < P.hand_len) {
        const int t = P.hand[lane];
        if (is_aka(t)) put(4, t >> 2);   // red five in hand (ch 4)
    }
    wave_sync();
    // 1-2. hand counts (ch 0-3)
    if (lane < ENC_W) {
        const int c = (int)hist[4 * 36 + my34];
        if (c >= 1) cell(0, 1.0f);
        if (c >= 2) cell(1, 1.0f);
        if (c >= 3) cell(2, 1.0f);
        if (c >= 4) cell(3, 1.0f);
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
        for (int m = 0; m < Q.n_melds; m++)
            tiles_used += ((Q.meld_type[m] >= RMJ_MELD_DAIMINKAN) ? 4 : 3) - ((ext_base && Q.meld_called[m] != 0xFF) ? 1 : 0);
        rank += (Q.score > my_score);
    }
    int tiles_left = (SANMA ? 108 : 136) - tiles_used;
    if (tiles_left < 0) tiles_left = 0;
    enc_always(30, enc_div<70, 255>(tiles_left));
    for (int rel = 0; rel < NPP; rel++) {
        const PState& Q = S.p[(pid + rel) % NPP];
        enc_always(26 + rel, enc_div<24, 255>((int)Q.n_discards));
        if (Q.flags & PF_RIICHI_DECLARED) enc_bcast(buf, 31 + rel, 1.0f, lane);
        int32_t sc = Q.score;
        int32_t s1 = sc < 0 ? 0 : (sc > 100000 ? 100000 : sc);
        int32_t s2 = sc < 0 ? 0 : (sc > 30000 ? 30000 : sc);
        enc_always(39 + rel, enc_div<100000, 100000>(s1));
        enc_always(43 + rel, enc_div<30000, 30000>(s2));
        enc_always(59 + rel, (float)Q.n_melds / 4.0f);
    }
    wave_sync();
    // 10. winds (ch 35-36)
    {
        int rw = S.round_wind;
        if (lane == 0 && 27 + rw < 34) put(35, 27 + rw);
        int seat = (pid + NPP - S.oya) % NPP;
        if (lane == 0) put(36, 27 + seat);
    }
    enc_always(37, enc_div<10, 255>((int)S.honba));
    enc_always(38, enc_div<5, 4096>((int)(S.riichi_sticks > 0x7FFFFFFFu ? 0x7FFFFFFFu : S.riichi_sticks)));
    // 14-15. waits / tenpai (ch 47-48)
    if (slot(47) >= 0 || slot(48) >= 0) {   // (wave-uniform) only the window that holds the wait channels pays for the probe
        uint64_t W = enc_waits(P, lane);
        if (lane < ENC_W && ((W >> my34) & 1ull)) cell(47, 1.0f);
        enc_always(48, W != 0ull ? 1.0f : 0.0f);
    }
    if (rank < NPP) enc_bcast(buf, 49 + rank, 1.0f, lane);
    enc_always(53, (float)S.kyoku_idx / 8.0f);
    enc_always(54, enc_div<7, 1275>((int)S.round_wind * 4 + (int)S.kyoku_idx));
    // 19. dora counts (ch 55-58): per seat, the number of visible tiles (own hand included for the observer) whose type is
    //     the dora of an indicator, counted per indicator (u8 accumulator in the reference); 21. tiles seen (ch 63)
    {
        int dcount[4] = {0, 0, 0, 0};
        for (int k = 0; k < S.n_dora; k++) {
            const int dt = (SANMA ? enc_next_tile136_sanma(S.dora[k]) : enc_next_tile136(S.dora[k])) >> 2;
            if (dt < 34) {
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (q < NPP) dcount[q] += (int)hist[q * 36 + dt] + (q == pid ? (int)hist[4 * 36 + dt] : 0);
            }
        }
        if (lane < ENC_W) {
            int seen = (int)hist[4 * 36 + my34];
            for (int q = 0; q < NPP; q++) seen += (int)hist[q * 36 + my34];
            for (int k = 0; k < S.n_dora; k++) seen += ((S.dora[k] >> 2) == my34);
            o.cell_quarters(63, seen & 0xFF);
        }
        for (int rel = 0; rel < NPP; rel++) {
            int q = (pid + rel) % NPP;
            int d = q == 0 ? dcount[0] : (q == 1 ? dcount[1] : (q == 2 ? dcount[2] : dcount[3]));
            enc_always(55 + rel, enc_div<12, 255>(d & 0xFF));
        }
    }
    // ch 70-73 stay 0: tsumogiri_flags is never filled (observation/mod.rs:105)
    o.flush();
    wave_sync();
}

// the float-staged form (extended encoder: whole tensor; ch_lo / ch_n select a window)
template <bool SANMA>
__device__ inline void encode_seat(const GState& S, int pid, float* buf, int lane, uint32_t* hist, int ch_lo = 0, int ch_n = ENC_CH,
                                   bool first = true) {
    EncFloatSink<SANMA ? ENC_W3 : ENC_W4> o{buf, ch_lo, ch_n, lane};
    encode_seat_to<SANMA>(S, pid, lane, hist, o, first);
}

// ---------------------------------------------------------------- encode_extended
// Channels 74..214 of Observation.encode_extended (python.rs:1271-1296): observation/encode.rs:293-585 (4P) /
// observation_3p/encode.rs:315-615 (3P).  `buf` holds all 215 x W floats; channels 0..73 were written by encode_seat.
// `decay[a]` = expf(-0.2f * a) computed on the HOST (the reference's f32::exp is the platform expf; a device exp could
// differ in the last bit), added in turn order like the reference's accumulation.
// Snapshot quirks kept: Observation.last_discard is the DISCARDER'S SEAT (state/mod.rs:252), dora membership compares the
// full 136-id with the copy-0 id of the next tile, riichi_sutehais is never set on a reachable path.
#define ENC_EXT_CH 215
// The 141 extended channels are produced in two LDS-sized groups (one 84-channel staging buffer per block):
//   group B ("scalars"): 74..93 (decay, shanten) -> slots 0..19 and 178..214 (availability, candidates, contexts) -> slots 20..56
//   group C ("melds")  : 94..177 (ankan, fuuro overview) -> slots 0..83
#define ENC_EXT_B_SLOTS 57
#define ENC_EXT_C_SLOTS 84
__device__ __forceinline__ int enc_ext_b_slot(int ch) { return ch < 94 ? ch - 74 : ch - 178 + 20; }
template <bool SANMA>
__device__ inline void encode_ext_scalars(const GState& S, int pid, float* tab, float* col4, int lane, const ShantenTables& T, const float* decay,
                                          const uint64_t* legal, int n_legal) {
    constexpr int ENC_W = SANMA ? ENC_W3 : ENC_W4;
    constexpr int NPP = SANMA ? 3 : 4;
    const PState& P = S.p[pid];
    const int my34 = SANMA ? (lane == 0 ? 0 : lane + 7) : lane;  // tile type of this lane's column
    // group B is four per-column channels (74..77 -> col4[4][W]) and 53 channels that hold one value in every column
    // (-> tab[slot], wave-uniform calls): 0.8 KB instead of 57 x W floats
    auto bc = [&](int ch, float v) {
        if (lane == 0) tab[enc_ext_b_slot(ch)] = v;
    };
    if (lane < ENC_EXT_B_SLOTS) tab[lane] = 0.0f;
    for (int i = lane; i < 4 * ENC_W; i += 64) col4[i] = 0.0f;
    wave_sync();
    // 74..77 discard history decay: lane = column, discards visited in turn order
    for (int c = 0; c < NPP; c++) {
        const PState& Q = S.p[(pid + c) % NPP];
        const int n = Q.n_discards;
        if (lane < ENC_W) {
            float acc = 0.0f;
            for (int turn = 0; turn < n; turn++)
                if ((Q.discards[turn] >> 2) == my34) acc += decay[n - 1 - turn];
            col4[c * ENC_W + lane] = acc;
        }
    }
    // hand / visible histograms: lane = tile type (34 lanes, also in 3P)
    uint32_t my_cnt = 0, my_vis = 0;
    if (lane < 34) {
        for (int j = 0; j < P.hand_len; j++) my_cnt += (P.hand[j] >> 2) == lane;
        for (int q = 0; q < NPP; q++) {
            const PState& Q = S.p[q];
            for (int j = 0; j < Q.n_discards; j++) my_vis += (Q.discards[j] >> 2) == lane;
            for (int m = 0; m < Q.n_melds; m++) {
                int nt = (Q.meld_type[m] >= RMJ_MELD_DAIMINKAN) ? 4 : 3;
                for (int k = 0; k < nt; k++) my_vis += (Q.meld_tiles[m][k] >> 2) == lane;
            }
        }
        for (int k = 0; k < S.n_dora; k++) my_vis += (S.dora[k] >> 2) == lane;
    }
    PH h = {0, 0, 0, 0};
    {
        const int s = lane < 34 ? t_suit(lane) : 0;
        uint32_t f = lane < 34 ? (my_cnt & 7u) << (3 * (lane - 9 * s)) : 0u;
        uint32_t w[4] = {s == 0 ? f : 0u, s == 1 ? f : 0u, s == 2 ? f : 0u, s == 3 ? f : 0u};
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) w[k] |= (uint32_t)__shfl_xor((int)w[k], off, 64);
        }
        h.a = w[0]; h.b = w[1]; h.c = w[2]; h.d = w[3];
    }
    const int total = ph_total(h);
    int cur_sh = 0;      // shanten of the hand (the ukeire walk computes it from the suit vectors it needs anyway)
    int nsh_type = 127;  // lane = tile type: shanten after discarding one tile of the type (filled by the ukeire walk)
    // 78..93 shanten efficiency
    {
        uint32_t eff, uke;
        sh_ukeire_both(T, h, my_cnt, my_vis, SANMA, lane, true, true, eff, uke, -99, &nsh_type, &cur_sh);
        for (int c = 0; c < NPP; c++) {
            const int base = 78 + c * 4;
            if (c == 0) {
                bc(base, fmaxf((float)cur_sh, 0.0f) / 8.0f);
                bc(base + 1, (float)eff / (SANMA ? 27.0f : 34.0f));
                bc(base + 2, (float)uke / 80.0f);
            } else {
                bc(base, 0.5f); bc(base + 1, 0.5f); bc(base + 2, 0.5f);
            }
            bc(base + 3, fminf((float)S.p[(pid + c) % NPP].n_discards / 18.0f, 1.0f));
        }
    }
    wave_sync();
    // 178..188 action availability over the seat's legal list
    {
        uint32_t kind = 0xFFu;
        if (lane < n_legal) {
            const uint64_t a = legal[lane];
            switch (a_type(a)) {
                case RMJ_RIICHI: kind = 0; break;
                case RMJ_CHI:
                    if (a_n(a) == 2) {
                        int t0 = (int)(a_c(a, 0) >> 2), t1 = (int)(a_c(a, 1) >> 2);
                        int diff = t1 > t0 ? t1 - t0 : t0 - t1;
                        if (diff == 1) kind = t0 < t1 ? 1 : 3;
                        else if (diff == 2) kind = 2;
                    }
                    break;
                case RMJ_PON: kind = 4; break;
                case RMJ_DAIMINKAN: kind = 5; break;
                case RMJ_ANKAN: kind = 6; break;
                case RMJ_KAKAN: kind = 7; break;
                case RMJ_TSUMO:
                case RMJ_RON: kind = 8; break;
                case RMJ_KYUSHU: kind = 9; break;
                case RMJ_PASS: kind = 10; break;
                default: break;
            }
        }
#pragma unroll
        for (int k = 0; k < 11; k++)
            if (__ballot(kind == (uint32_t)k)) bc(178 + k, 1.0f);
    }
    // 189..193 discard candidates: lane = hand slot
    {
        const int n = P.hand_len;
        // the shanten after discarding slot `lane` = the per-type number the ukeire walk has computed already
        const int ty = lane < n ? (P.hand[lane] >> 2) : 0;
        const int got = __shfl(nsh_type, ty, 64);
        const int ns = lane < n ? got : 99;
        const int keep = __popcll(__ballot(lane < n && ns == cur_sh));
        const int inc = __popcll(__ballot(lane < n && ns > cur_sh));
        bc(189, (float)n / 34.0f);
        if (n) {
            bc(190, (float)keep / (float)n);
            bc(191, (float)inc / (float)n);
        }
        bc(192, cur_sh == -1 ? 1.0f : 0.0f);
        bc(193, (P.flags & PF_RIICHI_DECLARED) ? 1.0f : 0.0f);
    }
    // 194..196 pass context, 197..205 last tedashis, 206..214 riichi sutehais
    {
        auto tile_feats = [&](int ch, int tile) {
            const int t34 = tile >> 2;
            if (SANMA) {
                const int k = enc_col<true>(t34);
                if (k >= 0) bc(ch, (float)k / 26.0f);
            } else {
                bc(ch, (float)t34 / 33.0f);
            }
            bc(ch + 1, is_aka(tile) ? 1.0f : 0.0f);
            bool dora = false;
            for (int k = 0; k < S.n_dora; k++)
                dora = dora || (SANMA ? enc_next_tile136_sanma(S.dora[k]) : enc_next_tile136(S.dora[k])) == tile;
            bc(ch + 2, dora ? 1.0f : 0.0f);
        };
        if (S.last_discard_pid != 0xFF) tile_feats(194, S.last_discard_pid);
        int opp = 0;
        for (int q = 0; q < NPP; q++) {
            if (q == pid) continue;
            if (S.p[q].last_tedashi != 0xFF) tile_feats(197 + opp * 3, S.p[q].last_tedashi);
            if (S.p[q].riichi_sutehai != 0xFF) tile_feats(206 + opp * 3, S.p[q].riichi_sutehai);
            opp++;
        }
    }
    wave_sync();
}

template <bool SANMA>
__device__ inline void encode_ext_melds(const GState& S, int pid, uint8_t* cells, int lane) {
    // group C (94..177) is a 0/1 pattern: one byte per cell (cell e of the group at cells[e]; the caller zeroed them)
    constexpr int ENC_W = SANMA ? ENC_W3 : ENC_W4;
    constexpr int NPP = SANMA ? 3 : 4;
    // 94..97 ankan overview, 98..177 fuuro overview: lane = 16*rel + 4*meld + slot
    {
        const int c = lane >> 4, mi = (lane >> 2) & 3, sl = lane & 3;
        if (c < NPP) {
            const PState& Q = S.p[(pid + c) % NPP];
            if (mi < Q.n_melds) {
                const int nt = (Q.meld_type[mi] >= RMJ_MELD_DAIMINKAN) ? 4 : 3;
                if (sl < nt) {
                    const int tile = Q.meld_tiles[mi][sl];
                    const int col = enc_col<SANMA>(tile >> 2);
                    if (col >= 0) {
                        cells[(4 + c * 20 + mi * 5 + sl) * ENC_W + col] = 1;
                        if (is_aka(tile)) cells[(4 + c * 20 + mi * 5 + 4) * ENC_W + col] = 1;
                        if (sl == 0 && Q.meld_type[mi] == RMJ_MELD_ANKAN) cells[c * ENC_W + col] = 1;
                    }
                }
            }
        }
    }
    wave_sync();
}

}  // namespace rmj
