// Host-only pieces of the C-ABI that touch no HIP API: state record <-> RmjStateView (rmj_peek_state / rmj_poke_state), the MJAI
// formatter of binary event records (rmj_format_event / rmj_format_events).  Plain C++17 on purpose: rmj_api.hip includes it, and so
// does the sanitizer translation unit (tests/host_san/host_san.cpp, built with -fsanitize=address,undefined by scripts/run_sanitizers.sh -
// GPU-side sanitizers do not exist on this pool, the host side of the boundary can be checked on the CPU).
//
// Reference: GameState / PlayerState (state/mod.rs:31-91, state/player.rs:6-39), MJAI strings with alphabetical keys
// (state/mod.rs:2094-2148), tile names (parser.rs:301-334).
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "../../include/riichi_mi355x.h"
#include "rmj_state.h"

namespace rmjh {

// ---------------------------------------------------------------- state peek / poke
inline void to_view(const GState& S, const uint8_t* W, RmjStateView* v) {
    memset(v, 0, sizeof(*v));
    int len = (int)S.live_end - (int)S.rinshan_count;
    if (len < 0) len = 0;
    v->wall_len = (uint8_t)len;
    for (int i = 0; i < len && i < 136 && S.rinshan_count + i < RMJ_WALL_STRIDE; i++) v->wall[i] = W[S.rinshan_count + i];
    v->n_dora = S.n_dora;
    for (int i = 0; i < 5; i++) v->dora[i] = i < S.n_dora ? S.dora[i] : 0;
    v->rinshan_draw_count = S.rinshan_count;
    v->pending_kan_dora_count = S.pending_kan_dora;
    v->drawable_count = S.drawable_count;
    v->wall_seed = S.wall_seed;
    v->hand_index = S.hand_index;
    for (int p = 0; p < 4; p++) {
        const PState& P = S.p[p];
        RmjPlayerView& q = v->players[p];
        q.hand_len = P.hand_len;
        for (int i = 0; i < P.hand_len && i < 14; i++) q.hand[i] = P.hand[i];
        q.n_melds = P.n_melds;
        for (int m = 0; m < P.n_melds && m < 4; m++) {
            RmjMeldView& mv = q.melds[m];
            mv.meld_type = P.meld_type[m];
            mv.n_tiles = P.meld_type[m] >= RMJ_MELD_DAIMINKAN ? 4 : 3;
            for (int k = 0; k < mv.n_tiles; k++) mv.tiles[k] = P.meld_tiles[m][k];
            mv.opened = P.meld_type[m] != RMJ_MELD_ANKAN;
            mv.from_who = P.meld_from[m] == 0xFF ? -1 : (int8_t)P.meld_from[m];
            mv.called_tile = P.meld_called[m] == 0xFF ? -1 : (int16_t)P.meld_called[m];
        }
        q.n_discards = P.n_discards;
        for (int i = 0; i < P.n_discards && i < RMJ_MAX_DISCARDS; i++) q.discards[i] = P.discards[i];
        q.discard_from_hand_bits = P.discard_from_hand_bits;
        q.discard_is_riichi_bits = P.discard_is_riichi_bits;
        q.riichi_declaration_index = P.riichi_decl_idx == 0xFF ? -1 : (int8_t)P.riichi_decl_idx;
        q.score = P.score;
        q.score_delta = P.score_delta;
        q.riichi_declared = (P.flags & PF_RIICHI_DECLARED) != 0;
        q.riichi_stage = (P.flags & PF_RIICHI_STAGE) != 0;
        q.double_riichi_declared = (P.flags & PF_DOUBLE_RIICHI) != 0;
        q.missed_agari_riichi = (P.flags & PF_MISSED_RIICHI) != 0;
        q.missed_agari_doujun = (P.flags & PF_MISSED_DOUJUN) != 0;
        q.nagashi_eligible = (P.flags & PF_NAGASHI) != 0;
        q.ippatsu_cycle = (P.flags & PF_IPPATSU) != 0;
        q.pao_daisangen = P.pao37 == 0xFF ? -1 : (int8_t)P.pao37;
        q.pao_daisuushi = P.pao50 == 0xFF ? -1 : (int8_t)P.pao50;
        q.n_forbidden = P.n_forbidden;
        for (int i = 0; i < P.n_forbidden && i < 2; i++) q.forbidden[i] = P.forbidden[i];
        q.riichi_sutehai = P.riichi_sutehai == 0xFF ? -1 : (int16_t)P.riichi_sutehai;
        q.last_tedashi = P.last_tedashi == 0xFF ? -1 : (int16_t)P.last_tedashi;
        q.n_kita = P.n_kita;
        for (int i = 0; i < P.n_kita && i < 4; i++) q.kita[i] = P.kita[i];
    }
    v->current_player = S.current_player;
    v->is_done = S.is_done;
    v->needs_tsumo = S.needs_tsumo;
    v->phase = S.phase;
    v->active_mask = S.active_mask;
    v->turn_count = S.turn_count;
    v->riichi_sticks = S.riichi_sticks;
    v->last_discard_pid = S.last_discard_pid == 0xFF ? -1 : (int16_t)S.last_discard_pid;
    v->last_discard_tile = S.last_discard_pid == 0xFF ? -1 : (int16_t)S.last_discard_tile;
    v->pending_kan_pid = S.pending_kan_pid == 0xFF ? -1 : (int16_t)S.pending_kan_pid;
    v->pending_kan_action = S.pending_kan_pid == 0xFF ? 0 : S.pending_kan_action;
    v->oya = S.oya;
    v->honba = S.honba;
    v->kyoku_idx = S.kyoku_idx;
    v->round_wind = S.round_wind;
    v->is_rinshan_flag = S.is_rinshan;
    v->is_first_turn = S.is_first_turn;
    v->riichi_pending_acceptance = S.riichi_pending == 0xFF ? -1 : (int16_t)S.riichi_pending;
    v->drawn_tile = S.drawn_tile == 0xFF ? -1 : (int16_t)S.drawn_tile;
    v->last_error_pid = S.last_error_pid == 0xFF ? -1 : (int16_t)S.last_error_pid;
}

// The view written over a record (S, W as fetched from the device): what the reference's Python setters do (env.rs:134-622).
// Returns 0, or a message for RMJ_ERR_ARG.  Derived caches (wait cache, stale claim counts) are invalidated.
inline const char* from_view(GState& S, uint8_t* W, const RmjStateView* v) {
    S.rinshan_count = v->rinshan_draw_count;
    const int len = v->wall_len;
    if (S.rinshan_count + len > 136) return "wall too long";
    for (int p = 0; p < 4; p++) S.stale_n[p] = 0;
    for (int i = 0; i < len; i++) W[S.rinshan_count + i] = v->wall[i];
    S.live_end = (uint8_t)(S.rinshan_count + len);
    S.n_dora = v->n_dora > 5 ? 5 : v->n_dora;
    for (int i = 0; i < S.n_dora; i++) S.dora[i] = v->dora[i];
    S.pending_kan_dora = v->pending_kan_dora_count;
    S.drawable_count = v->drawable_count;
    S.hand_index = (uint32_t)v->hand_index;
    for (int p = 0; p < 4; p++) {
        const RmjPlayerView& q = v->players[p];
        if (q.hand_len > 14 || q.n_melds > 4 || q.n_discards > RMJ_MAX_DISCARDS) return "player view out of range";
    }
    for (int p = 0; p < 4; p++) {
        PState& P = S.p[p];
        const RmjPlayerView& q = v->players[p];
        P.hand_len = q.hand_len;
        for (int i = 0; i < q.hand_len; i++) P.hand[i] = q.hand[i];
        P.n_melds = q.n_melds;
        for (int m = 0; m < q.n_melds; m++) {
            const RmjMeldView& mv = q.melds[m];
            P.meld_type[m] = mv.meld_type;
            uint8_t t[4] = {0, 0, 0, 0};
            const int nt = mv.n_tiles > 4 ? 4 : mv.n_tiles;
            for (int k = 0; k < nt; k++) t[k] = mv.tiles[k];
            std::sort(t, t + nt);
            for (int k = 0; k < 4; k++) P.meld_tiles[m][k] = t[k];
            P.meld_from[m] = mv.from_who < 0 ? 0xFF : (uint8_t)mv.from_who;
            P.meld_called[m] = mv.called_tile < 0 ? 0xFF : (uint8_t)mv.called_tile;
        }
        P.n_discards = q.n_discards;
        P.discard_type_mask = 0;
        for (int i = 0; i < q.n_discards; i++) {
            P.discards[i] = q.discards[i];
            P.discard_type_mask |= 1ull << ((q.discards[i] >> 2) & 63);
        }
        P.discard_from_hand_bits = q.discard_from_hand_bits;
        P.discard_is_riichi_bits = q.discard_is_riichi_bits;
        P.riichi_decl_idx = q.riichi_declaration_index < 0 ? 0xFF : (uint8_t)q.riichi_declaration_index;
        P.score = q.score;
        P.score_delta = q.score_delta;
        P.flags = (uint8_t)((q.riichi_declared ? PF_RIICHI_DECLARED : 0) | (q.riichi_stage ? PF_RIICHI_STAGE : 0) |
                            (q.double_riichi_declared ? PF_DOUBLE_RIICHI : 0) | (q.missed_agari_riichi ? PF_MISSED_RIICHI : 0) |
                            (q.missed_agari_doujun ? PF_MISSED_DOUJUN : 0) | (q.nagashi_eligible ? PF_NAGASHI : 0) |
                            (q.ippatsu_cycle ? PF_IPPATSU : 0));
        P.pao37 = q.pao_daisangen < 0 ? 0xFF : (uint8_t)q.pao_daisangen;
        P.pao50 = q.pao_daisuushi < 0 ? 0xFF : (uint8_t)q.pao_daisuushi;
        P.n_forbidden = q.n_forbidden > 2 ? 2 : q.n_forbidden;
        for (int i = 0; i < P.n_forbidden; i++) P.forbidden[i] = q.forbidden[i];
        P.riichi_sutehai = q.riichi_sutehai < 0 ? 0xFF : (uint8_t)q.riichi_sutehai;
        P.last_tedashi = q.last_tedashi < 0 ? 0xFF : (uint8_t)q.last_tedashi;
        P.n_kita = q.n_kita > 4 ? 4 : q.n_kita;
        for (int i = 0; i < P.n_kita; i++) P.kita[i] = q.kita[i];
    }
    S.current_player = v->current_player;
    S.is_done = v->is_done;
    S.needs_tsumo = v->needs_tsumo;
    S.phase = v->phase;
    S.active_mask = v->active_mask;
    S.turn_count = v->turn_count;
    S.riichi_sticks = v->riichi_sticks;
    S.last_discard_pid = v->last_discard_pid < 0 ? 0xFF : (uint8_t)v->last_discard_pid;
    S.last_discard_tile = v->last_discard_pid < 0 ? 0 : (uint8_t)v->last_discard_tile;
    S.pending_kan_pid = v->pending_kan_pid < 0 ? 0xFF : (uint8_t)v->pending_kan_pid;
    S.pending_kan_action = v->pending_kan_pid < 0 ? 0 : v->pending_kan_action;
    S.oya = v->oya;
    S.honba = v->honba;
    S.kyoku_idx = v->kyoku_idx;
    S.round_wind = v->round_wind;
    S.is_rinshan = v->is_rinshan_flag;
    S.is_first_turn = v->is_first_turn;
    S.riichi_pending = v->riichi_pending_acceptance < 0 ? 0xFF : (uint8_t)v->riichi_pending_acceptance;
    S.drawn_tile = v->drawn_tile < 0 ? 0xFF : (uint8_t)v->drawn_tile;
    S.last_error_pid = v->last_error_pid < 0 ? 0xFF : (uint8_t)v->last_error_pid;
    return nullptr;
}

// ---------------------------------------------------------------- MJAI formatting
// Bounded appender: never writes past `end`, remembers how many bytes the text needs.
struct Out {
    char* p;
    char* end;
    uint64_t need;
    void put(char c) {
        if (p < end) *p++ = c;
        need++;
    }
    void str(const char* s) {
        while (*s) put(*s++);
    }
    void num(int64_t v) {
        char t[24];
        int n = 0;
        uint64_t u = v < 0 ? (uint64_t)(-(v + 1)) + 1u : (uint64_t)v;
        do { t[n++] = (char)('0' + u % 10u); u /= 10u; } while (u);
        if (v < 0) put('-');
        while (n) put(t[--n]);
    }
};
inline void put_tile(Out& o, uint8_t tid) {   // parser.rs:301-334
    if (tid == 16) { o.str("5mr"); return; }
    if (tid == 52) { o.str("5pr"); return; }
    if (tid == 88) { o.str("5sr"); return; }
    if (tid < 108) {
        static const char sc[3] = {'m', 'p', 's'};
        o.put((char)('1' + (tid % 36) / 4));
        o.put(sc[tid / 36]);
        return;
    }
    static const char* hon[7] = {"E", "S", "W", "N", "P", "F", "C"};
    const int num = (tid - 108) / 4;
    if (num < 7) { o.str(hon[num]); return; }
    o.num(num + 1);
    o.put('z');
}
inline void put_tiles(Out& o, const uint8_t* t, int n) {
    o.put('[');
    for (int i = 0; i < n; i++) {
        if (i) o.put(',');
        o.put('"');
        put_tile(o, t[i]);
        o.put('"');
    }
    o.put(']');
}
inline void put_ints(Out& o, const int32_t* v, int n) {
    o.put('[');
    for (int i = 0; i < n; i++) {
        if (i) o.put(',');
        o.num(v[i]);
    }
    o.put(']');
}
// One event (START_KYOKU consumes the two TEHAI records behind it) appended to `o` without a terminator.  Returns the records
// consumed, or a negative RMJ_ERR_*.  seat < 0: the full log string; 0..3: the seat's masked view (_push_mjai_event).
inline int format_event(Out& o, const RmjEvent* ev, uint32_t n_avail, int seat) {
    if (!ev || n_avail == 0) return RMJ_ERR_ARG;
    int used = 1;
    const RmjEvent& e = ev[0];
    int ncons = e.flags >> 4;
    if (ncons > 4) ncons = 4;
    const int np = (e.pad == 3) ? 3 : 4;  // seats, written into the pad byte by the device
    switch (e.type) {
        case RMJ_EV_START_GAME: o.str("{\"type\":\"start_game\"}"); break;
        case RMJ_EV_END_KYOKU: o.str("{\"type\":\"end_kyoku\"}"); break;
        case RMJ_EV_END_GAME: o.str("{\"type\":\"end_game\"}"); break;
        case RMJ_EV_START_KYOKU: {
            if (n_avail < 3 || ev[1].type != RMJ_EV_TEHAI || ev[2].type != RMJ_EV_TEHAI) return RMJ_ERR_ARG;
            used = 3;
            static const char* winds[4] = {"E", "S", "W", "N"};
            const uint32_t kyotaku = e.consumed[2] | ((uint32_t)e.consumed[3] << 8);
            o.str("{\"bakaze\":\""); o.str(winds[e.consumed[0] & 3]); o.str("\",\"dora_marker\":\""); put_tile(o, e.tile);
            o.str("\",\"honba\":"); o.num(e.consumed[1]); o.str(",\"kyoku\":"); o.num(e.target); o.str(",\"kyotaku\":"); o.num(kyotaku);
            o.str(",\"oya\":"); o.num(e.actor); o.str(",\"scores\":"); put_ints(o, e.deltas, np); o.str(",\"tehais\":[");
            for (int p = 0; p < np; p++) {
                const uint8_t* pl = reinterpret_cast<const uint8_t*>(&ev[1 + p / 2]) + 4 + 13 * (p & 1);
                if (p) o.put(',');
                if (seat < 0 || seat == p) put_tiles(o, pl, 13);
                else {
                    o.put('[');
                    for (int k = 0; k < 13; k++) o.str(k ? ",\"?\"" : "\"?\"");
                    o.put(']');
                }
            }
            o.str("],\"type\":\"start_kyoku\"}");
            break;
        }
        case RMJ_EV_TSUMO:
            o.str("{\"actor\":"); o.num(e.actor); o.str(",\"pai\":\"");
            if (seat < 0 || seat == e.actor) put_tile(o, e.tile); else o.put('?');
            o.str("\",\"type\":\"tsumo\"}");
            break;
        case RMJ_EV_DAHAI:
            o.str("{\"actor\":"); o.num(e.actor); o.str(",\"pai\":\""); put_tile(o, e.tile); o.str("\",\"tsumogiri\":");
            o.str((e.flags & 1) ? "true" : "false"); o.str(",\"type\":\"dahai\"}");
            break;
        case RMJ_EV_REACH: o.str("{\"actor\":"); o.num(e.actor); o.str(",\"type\":\"reach\"}"); break;
        case RMJ_EV_REACH_ACCEPTED: o.str("{\"actor\":"); o.num(e.actor); o.str(",\"type\":\"reach_accepted\"}"); break;
        case RMJ_EV_CHI:
        case RMJ_EV_PON:
        case RMJ_EV_DAIMINKAN:
            o.str("{\"actor\":"); o.num(e.actor); o.str(",\"consumed\":"); put_tiles(o, e.consumed, ncons); o.str(",\"pai\":\""); put_tile(o, e.tile);
            o.str("\",\"target\":"); o.num(e.target); o.str(",\"type\":\"");
            o.str(e.type == RMJ_EV_CHI ? "chi" : (e.type == RMJ_EV_PON ? "pon" : "daiminkan")); o.str("\"}");
            break;
        case RMJ_EV_ANKAN:
        case RMJ_EV_KAKAN:
            o.str("{\"actor\":"); o.num(e.actor); o.str(",\"consumed\":"); put_tiles(o, e.consumed, ncons); o.str(",\"pai\":\""); put_tile(o, e.tile);
            o.str("\",\"type\":\""); o.str(e.type == RMJ_EV_ANKAN ? "ankan" : "kakan"); o.str("\"}");
            break;
        case RMJ_EV_KITA:
            o.str("{\"actor\":"); o.num(e.actor); o.str(",\"pai\":\""); put_tile(o, e.tile); o.str("\",\"type\":\"kita\"}");
            break;
        case RMJ_EV_DORA: o.str("{\"dora_marker\":\""); put_tile(o, e.tile); o.str("\",\"type\":\"dora\"}"); break;
        case RMJ_EV_HORA:
            o.str("{\"actor\":"); o.num(e.actor); o.str(",\"deltas\":"); put_ints(o, e.deltas, np); o.str(",\"target\":"); o.num(e.target);
            if (e.flags & 1) o.str(",\"tsumo\":true");
            o.str(",\"type\":\"hora\",\"ura_markers\":"); put_tiles(o, e.ura, e.n_ura > 5 ? 5 : e.n_ura); o.put('}');
            break;
        case RMJ_EV_RYUKYOKU: {
            static const char* reasons[7] = {"exhaustive_draw", "nagashimangan", "kyushu_kyuhai", "sufuurenta", "suukansansen", "suucha_riichi",
                                             "sanchaho"};
            o.str("{\"deltas\":"); put_ints(o, e.deltas, np); o.str(",\"reason\":\"");
            if (e.flags < 7) o.str(reasons[e.flags]);
            else { o.str("Error: Illegal Action by Player "); o.num(e.actor); }
            o.str("\",\"type\":\"ryukyoku\"}");
            break;
        }
        default: return RMJ_ERR_ARG;
    }
    return used;
}

// The records of game g are ev[offsets[g] .. offsets[g + 1]); its log = the events' strings, one per line ('\n' after each), written
// at buf + text_offsets[g].  Two passes over the games by a pool of threads: sizes, then text (a record run that starts inside a
// start_kyoku triple or ends inside one is formatted up to the last complete event).  Returns the bytes needed in all;
// the text is written only when it fits `cap`.
inline uint64_t format_events(const RmjEvent* ev, const uint32_t* offsets, uint32_t n_games, int seat, char* buf, uint64_t cap, uint64_t* text_offsets,
                              int threads) {
    if (threads <= 0) {
        threads = (int)std::thread::hardware_concurrency();
        if (threads <= 0) threads = 1;
        if (threads > 32) threads = 32;
    }
    if ((uint32_t)threads > n_games) threads = n_games ? (int)n_games : 1;
    auto one = [&](uint32_t g, char* dst, uint64_t room) -> uint64_t {
        Out o{dst, dst ? dst + room : dst, 0};
        const uint32_t lo = offsets[g], hi = offsets[g + 1];
        for (uint32_t i = lo; i < hi;) {
            if (ev[i].type == RMJ_EV_TEHAI) { i++; continue; }   // the tail of a triple whose head was lost: skipped
            const uint64_t mark = o.need;
            char* const pm = o.p;
            const int used = format_event(o, ev + i, hi - i, seat);
            if (used <= 0) { o.need = mark; o.p = pm; break; }
            o.put('\n');
            i += (uint32_t)used;
        }
        return o.need;
    };
    auto run = [&](auto&& body) {
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; t++) pool.emplace_back(body, t);
        body(0);
        for (auto& th : pool) th.join();
    };
    std::vector<uint64_t> sz(n_games);
    run([&](int t) {
        for (uint32_t g = (uint32_t)t; g < n_games; g += (uint32_t)threads) sz[g] = one(g, nullptr, 0);
    });
    uint64_t total = 0;
    for (uint32_t g = 0; g < n_games; g++) {
        text_offsets[g] = total;
        total += sz[g];
    }
    text_offsets[n_games] = total;
    if (buf && total <= cap)
        run([&](int t) {
            for (uint32_t g = (uint32_t)t; g < n_games; g += (uint32_t)threads) one(g, buf + text_offsets[g], sz[g]);
        });
    return total;
}

}  // namespace rmjh
