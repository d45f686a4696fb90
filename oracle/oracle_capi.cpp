// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
// C wrapper (ctypes-friendly) around the CPU restatement in riichi_core.hpp /
// riichi_state.hpp.  Uses the POD views of include/riichi_mi355x.h so tests can
// compare the oracle and the HIP path byte for byte.
#include <chrono>
#include <cmath>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <thread>

#include "../include/riichi_mi355x.h"
#include "riichi_state.hpp"
#include "riichi_shanten.hpp"

using namespace orc;

static rmj_action_t pack_action(const Action& a) {
    uint64_t v = (uint64_t)a.type;
    v |= (uint64_t)(a.tile < 0 ? 0xFF : a.tile) << 8;
    v |= (uint64_t)a.consume.size() << 16;
    for (size_t i = 0; i < a.consume.size() && i < 4; i++) v |= (uint64_t)a.consume[i] << (24 + 8 * i);
    return v;
}
static Action unpack_action(rmj_action_t v, int actor) {
    Action a;
    a.type = (ActionType)(v & 0xFF);
    int t = (v >> 8) & 0xFF;
    a.tile = t == 0xFF ? -1 : t;
    int n = (v >> 16) & 0xFF;
    for (int i = 0; i < n && i < 4; i++) a.consume.push_back((uint8_t)((v >> (24 + 8 * i)) & 0xFF));
    std::sort(a.consume.begin(), a.consume.end());
    a.actor = actor;
    return a;
}

static Meld meld_from_view(const RmjMeldView& v) {
    Meld m;
    m.meld_type = (MeldType)v.meld_type;
    for (int i = 0; i < v.n_tiles; i++) m.tiles.push_back(v.tiles[i]);
    m.opened = v.opened;
    m.from_who = v.from_who;
    m.called_tile = v.called_tile;
    return m;
}
static void meld_to_view(const Meld& m, RmjMeldView& v) {
    std::memset(&v, 0, sizeof(v));
    v.meld_type = m.meld_type;
    v.n_tiles = (uint8_t)m.tiles.size();
    for (size_t i = 0; i < m.tiles.size() && i < 4; i++) v.tiles[i] = m.tiles[i];
    v.opened = m.opened;
    v.from_who = m.from_who;
    v.called_tile = (int16_t)m.called_tile;
}

// policy choice shared (by definition, not by code) with the HIP path; see include/riichi_mi355x.h rmj_step_random
// (round 6: a 32-bit finaliser over gs = splitmix64(seed + game) and the (step, seat) counter, a multiply-high instead of a modulo - the device
// spends 13 vector instructions on a pick instead of ~55; csrc/rmj_common.hip.h policy_key32 / policy_pick / policy_calls / policy_tie)
static inline uint32_t policy_key32(uint64_t gs, uint32_t step, uint32_t seat) {
    uint32_t x = ((uint32_t)gs ^ ((step * 4u + seat) * 0x9E3779B1u)) + (uint32_t)(gs >> 32);
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
static inline uint32_t policy_pick(uint32_t key, uint32_t n) { return (uint32_t)(((uint64_t)key * (uint64_t)n) >> 32); }
static inline uint64_t policy_choice(uint64_t seed, uint64_t game, uint64_t step, uint32_t seat, uint32_t n) {
    return policy_pick(policy_key32(splitmix64(seed + game), (uint32_t)step, seat), n);
}

extern "C" {

// ---------------------------------------------------------------- hand math
int orc_eval_hands(const RmjHandCase* cases, uint32_t n, RmjHandResult* out) {
    for (uint32_t k = 0; k < n; k++) {
        const RmjHandCase& c = cases[k];
        RmjHandResult& r = out[k];
        std::memset(&r, 0, sizeof(r));
        std::vector<uint8_t> tiles(c.tiles, c.tiles + c.n_tiles);
        std::vector<Meld> melds;
        for (int i = 0; i < c.n_melds; i++) melds.push_back(meld_from_view(c.melds[i]));
        HandEvaluator he(tiles, melds, c.is_sanma != 0);
        Conditions cd;
        cd.tsumo = c.tsumo; cd.riichi = c.riichi; cd.double_riichi = c.double_riichi; cd.ippatsu = c.ippatsu;
        cd.haitei = c.haitei; cd.houtei = c.houtei; cd.rinshan = c.rinshan; cd.chankan = c.chankan;
        cd.tsumo_first_turn = c.tsumo_first_turn;
        cd.player_wind = c.player_wind % 4; cd.round_wind = c.round_wind % 4;
        cd.honba = c.honba; cd.kita_count = c.kita_count; cd.is_sanma = c.is_sanma;
        std::vector<uint8_t> dora(c.dora, c.dora + c.n_dora), ura(c.ura, c.ura + c.n_ura);
        WinResult w = he.calc(c.win_tile, dora, ura, cd);
        r.is_win = w.is_win; r.yakuman = w.yakuman; r.has_win_shape = w.has_win_shape;
        r.n_yaku = (uint8_t)std::min<size_t>(w.yaku.size(), 20);
        for (int i = 0; i < r.n_yaku; i++) r.yaku[i] = (uint8_t)w.yaku[i];
        r.han = w.han; r.fu = w.fu; r.ron_agari = w.ron_agari; r.tsumo_agari_oya = w.tsumo_agari_oya;
        r.tsumo_agari_ko = w.tsumo_agari_ko;
        for (uint8_t t : he.get_waits_u8()) r.waits |= 1ull << t;
        r.is_tenpai = he.is_tenpai();
        Hand h14 = he.hand;
        if (he.current_total() == 13) h14.add(c.win_tile / 4);
        r.is_agari = is_agari(h14);
    }
    return 0;
}

int orc_agari_counts(const uint8_t* counts, uint32_t n, uint8_t* agari, uint8_t* tenpai, uint64_t* waits) {
    for (uint32_t k = 0; k < n; k++) {
        Hand h;
        std::memcpy(h.counts, counts + 34 * k, 34);
        agari[k] = is_agari(h);
        // hand_evaluator.rs:178-213 semantics on a raw histogram (no melds): only 13-tile hands have waits
        uint64_t w = 0;
        bool tp = false;
        if (h.total() == 13) {
            for (int i = 0; i < 34; i++)
                if (h.counts[i] < 4) {
                    h.add((uint8_t)i);
                    if (is_agari(h)) { w |= 1ull << i; tp = true; }
                    h.remove((uint8_t)i);
                }
        }
        tenpai[k] = tp;
        waits[k] = w;
    }
    return 0;
}
// agari.rs:15-63 free function (with neighbour pruning) — used by hands_negative.json check
int orc_is_tenpai_free(const uint8_t* counts34) {
    Hand h;
    std::memcpy(h.counts, counts34, 34);
    return is_tenpai_counts(h);
}

int orc_calculate_score(const uint8_t* han, const uint8_t* fu, const uint8_t* is_oya, const uint8_t* is_tsumo,
                        const uint32_t* honba, const uint8_t* np, uint32_t n, uint32_t* out) {
    for (uint32_t k = 0; k < n; k++) {
        Score s = calculate_score(han[k], fu[k], is_oya[k], is_tsumo[k], honba[k], np[k]);
        out[4 * k] = s.total; out[4 * k + 1] = s.pay_ron; out[4 * k + 2] = s.pay_tsumo_oya; out[4 * k + 3] = s.pay_tsumo_ko;
    }
    return 0;
}

int orc_find_divisions(const uint8_t* counts34, uint8_t* out /*[max][9]: head,n,(k,t)x4... */, int max_div) {
    Hand h;
    std::memcpy(h.counts, counts34, 34);
    auto d = find_divisions(h);
    int n = 0;
    for (auto& dv : d) {
        if (n >= max_div) break;
        uint8_t* o = out + 10 * n;
        o[0] = dv.head;
        o[1] = (uint8_t)dv.body.size();
        for (size_t i = 0; i < dv.body.size() && i < 4; i++) {
            o[2 + 2 * i] = dv.body[i].koutsu;
            o[3 + 2 * i] = dv.body[i].t;
        }
        n++;
    }
    return (int)d.size();
}

// shanten.rs:244-261 / :470-484
int orc_shanten(const uint8_t* counts, uint32_t n, int sanma, int8_t* out) {
    for (uint32_t k = 0; k < n; k++) {
        const uint8_t* c = counts + 34 * (size_t)k;
        int total = 0;
        for (int i = 0; i < 34; i++) total += c[i];
        out[k] = (int8_t)calc_shanten_from_counts(c, total / 3, sanma != 0);
    }
    return 0;
}

// shanten.rs:304-327 / 525-548 and :331-405 / 552-626 on type histograms
int orc_effective_tiles(const uint8_t* counts, uint32_t n, int sanma, uint32_t* out) {
    for (uint32_t k = 0; k < n; k++) out[k] = effective_tiles_with_discard(counts + 34 * (size_t)k, sanma != 0);
    return 0;
}
int orc_best_ukeire(const uint8_t* counts, const uint8_t* visible, uint32_t n, int sanma, uint32_t* out) {
    for (uint32_t k = 0; k < n; k++) out[k] = best_ukeire(counts + 34 * (size_t)k, visible + 34 * (size_t)k, sanma != 0);
    return 0;
}

// ---------------------------------------------------------------- MJAI event ingestion (row N1)
// apply_mjai_event: state/event_handler.rs:18-330 (4P), state_3p/event_handler.rs:18-362 (3P), on the binary records
// of include/riichi_mi355x.h (a start_kyoku is START_KYOKU + two TEHAI records, like the emitted log).  Tile strings
// were mapped by mjai_to_tid (parser.rs:336-385) on the host.  The caller-side logging of env.rs:56-72 is not restated.
static void erase_first(std::vector<uint8_t>& v, uint8_t t) {
    int i = vec_position(v, t);
    if (i >= 0) v.erase(v.begin() + i);
}
static void apply_claims(GameState* g, int actor, uint8_t tile, bool ron_only) {
    g->current_claims.clear();
    g->active_players.clear();
    std::vector<uint8_t> claim_active;
    for (int i = 0; i < g->NP; i++) {
        if (i == actor) continue;
        auto r = g->_get_claim_actions_for_player((uint8_t)i, (uint8_t)actor, tile);
        std::vector<Action> legals;
        for (auto& a : r.first)
            if (!ron_only || a.type == AT_RON) legals.push_back(a);
        if (!legals.empty()) {
            claim_active.push_back((uint8_t)i);
            g->current_claims[(uint8_t)i] = legals;
        }
    }
    if (!claim_active.empty()) {
        g->phase = WAIT_RESPONSE;
        g->active_players = claim_active;
    } else {
        g->phase = WAIT_ACT;
        g->active_players.clear();
        g->current_player = 0xFF;
    }
}
void orc_game_apply_event(void* gp, const RmjEvent* ev, int nrec) {
    GameState* g = (GameState*)gp;
    if (nrec < 1) return;
    const RmjEvent& e = ev[0];
    const int actor = e.actor;
    // Replay semantics (pad bit 0; KyokuStepIterator::_collect_pass_observations, replay/mod.rs:129-177): a seat that was offered
    // Ron on the last discard and does not win with this event has passed - same-turn furiten, permanent in riichi.
    if ((e.pad & 1) && g->phase == WAIT_RESPONSE && !g->pending_kan.has_value()) {
        for (auto& kv : g->current_claims) {
            if (std::find(g->active_players.begin(), g->active_players.end(), kv.first) == g->active_players.end()) continue;
            if (e.type == RMJ_EV_HORA && kv.first == (uint8_t)actor) continue;
            bool ron = false;
            for (auto& a : kv.second) ron = ron || a.type == AT_RON;
            if (!ron) continue;
            PlayerState& Q = g->players[kv.first];
            Q.missed_agari_doujun = true;
            if (Q.riichi_declared) Q.missed_agari_riichi = true;
        }
    }
    // ... and the walker's discard (apply_log_action, state/event_handler.rs:391-392) ends the discarder's same-turn furiten
    if ((e.pad & 1) && e.type == RMJ_EV_DAHAI && actor < g->NP) g->players[actor].missed_agari_doujun = false;
    // ... and the walker knows a replacement draw (apply_log_action: is_after_kan, state/event_handler.rs:415, :428-430, :565, :658,
    // state_3p/event_handler.rs:606 - kita too): the tile dealt after a kan is a rinshan draw, any other deal is not
    if (e.type == RMJ_EV_START_KYOKU) g->replay_after_kan = false;
    if (e.pad & 1) {
        switch (e.type) {
            case RMJ_EV_DAHAI: case RMJ_EV_PON: case RMJ_EV_CHI: g->replay_after_kan = false; break;
            case RMJ_EV_DAIMINKAN: case RMJ_EV_ANKAN: case RMJ_EV_KAKAN: g->replay_after_kan = true; break;
            case RMJ_EV_KITA: if (g->sanma) g->replay_after_kan = true; break;
            case RMJ_EV_TSUMO: g->is_rinshan_flag = g->replay_after_kan; g->replay_after_kan = false; break;
            default: break;
        }
    }
    switch (e.type) {
        case RMJ_EV_START_GAME:  // env.rs:56-72 (reset) + event_handler.rs:20-25
            g->reset();
            g->mjai_log.clear();
            for (auto& l : g->mjai_log_per_player) l.clear();
            g->current_player = 0xFF;
            g->active_players.clear();
            break;
        case RMJ_EV_START_KYOKU: {  // event_handler.rs:26-103
            g->honba = e.consumed[1];
            g->riichi_sticks = (uint32_t)e.consumed[2] | ((uint32_t)e.consumed[3] << 8);
            g->round_wind = e.consumed[0] & 3;
            g->oya = e.actor;
            g->kyoku_idx = e.target ? (uint8_t)(e.target - 1) : 0;  // kyoku.saturating_sub(1)
            g->current_player = 0xFF;
            g->turn_count = 0;
            g->is_done = false;
            g->needs_tsumo = true;
            g->phase = WAIT_ACT;
            g->active_players.clear();
            g->last_discard.reset();
            g->current_claims.clear();
            g->pending_kan.reset();
            g->is_rinshan_flag = false;
            g->is_first_turn = true;
            g->riichi_pending_acceptance = -1;
            g->drawn_tile = -1;
            g->win_results.clear();
            g->last_error.reset();
            for (int i = 0; i < 4; i++) { g->riichi_sutehais[i] = -1; g->last_tedashis[i] = -1; }
            g->wall.tiles.assign((size_t)((g->sanma ? 108 : 136) - 13 * g->NP), 0);
            g->wall.dora_indicators.assign(1, e.tile);
            g->wall.rinshan_draw_count = 0;
            g->wall.pending_kan_dora_count = 0;
            g->wall.wall_digest.clear();  // event_handler.rs:81-82
            g->wall.salt.clear();
            g->wall.drawable_count = (uint8_t)(g->wall.tiles.size() - 14);
            for (int i = 0; i < g->NP; i++) g->players[i].reset_round();
            for (int i = 0; i < g->NP; i++) g->players[i].score = e.deltas[i];
            for (int half = 0; half < 2 && 1 + half < nrec; half++) {
                const uint8_t* pl = reinterpret_cast<const uint8_t*>(&ev[1 + half]) + 4;
                for (int q = 0; q < 2; q++) {
                    int seat = 2 * half + q;
                    if (seat >= g->NP) continue;
                    std::vector<uint8_t> hand(pl + 13 * q, pl + 13 * q + 13);
                    std::sort(hand.begin(), hand.end());
                    g->players[seat].hand = hand;
                }
            }
            break;
        }
        case RMJ_EV_TSUMO: {  // :104-118
            PlayerState& P = g->players[actor];
            g->current_player = (uint8_t)actor;
            g->drawn_tile = e.tile;
            P.hand.push_back(e.tile);
            std::sort(P.hand.begin(), P.hand.end());
            P.forbidden_discards.clear();
            if (!g->wall.tiles.empty()) {
                g->wall.tiles.pop_back();
                g->wall.drawable_count = g->wall.drawable_count ? (uint8_t)(g->wall.drawable_count - 1) : 0;
            }
            g->phase = WAIT_ACT;
            g->active_players.assign(1, (uint8_t)actor);
            g->needs_tsumo = false;
            break;
        }
        case RMJ_EV_DAHAI: {  // :119-156
            PlayerState& P = g->players[actor];
            g->current_player = (uint8_t)actor;
            erase_first(P.hand, e.tile);
            P.discards.push_back(e.tile);
            g->last_discard = std::make_pair((uint8_t)actor, e.tile);
            g->drawn_tile = -1;
            if (P.riichi_stage) { P.riichi_declared = true; P.riichi_stage = false; }
            apply_claims(g, actor, e.tile, false);
            g->needs_tsumo = true;
            break;
        }
        case RMJ_EV_PON:
        case RMJ_EV_CHI: {  // :157-238 (3P: no kuikae suji for chi, state_3p/event_handler.rs:195-222)
            PlayerState& P = g->players[actor];
            g->current_player = (uint8_t)actor;
            uint8_t c1 = e.consumed[0], c2 = e.consumed[1];
            erase_first(P.hand, c1);
            erase_first(P.hand, c2);
            Meld m;
            m.meld_type = e.type == RMJ_EV_PON ? MT_PON : MT_CHI;
            m.tiles = {e.tile, c1, c2};
            m.opened = true;
            m.from_who = -1;
            m.called_tile = e.tile;
            P.melds.push_back(m);
            g->drawn_tile = -1;
            g->phase = WAIT_ACT;
            g->active_players.assign(1, (uint8_t)actor);
            g->needs_tsumo = false;
            if (e.type == RMJ_EV_PON || !g->sanma) {
                P.forbidden_discards.clear();
                if (g->rule.kuikae_forbidden) {
                    P.forbidden_discards.push_back(e.tile);
                    if (e.type == RMJ_EV_CHI) {
                        int t34 = e.tile / 4;
                        int a = c1 / 4, b = c2 / 4;
                        if (a > b) std::swap(a, b);
                        if (a == t34 + 1 && b == t34 + 2) {
                            if (t34 % 9 <= 5) P.forbidden_discards.push_back((uint8_t)((t34 + 3) * 4));
                        } else if (t34 >= 2 && b == t34 - 1 && a == t34 - 2 && t34 % 9 >= 3) {
                            P.forbidden_discards.push_back((uint8_t)((t34 - 3) * 4));
                        }
                    }
                }
            }
            break;
        }
        case RMJ_EV_DAIMINKAN: {  // :239-268
            PlayerState& P = g->players[actor];
            g->current_player = (uint8_t)actor;
            int n = (e.flags >> 4) & 15;
            Meld m;
            m.meld_type = MT_DAIMINKAN;
            m.tiles.push_back(e.tile);
            for (int k = 0; k < n && k < 4; k++) m.tiles.push_back(e.consumed[k]);
            for (int k = 0; k < n && k < 4; k++) erase_first(P.hand, e.consumed[k]);
            m.opened = true;
            m.from_who = -1;
            m.called_tile = e.tile;
            P.melds.push_back(m);
            g->phase = WAIT_ACT;
            g->active_players.assign(1, (uint8_t)actor);
            g->needs_tsumo = true;
            break;
        }
        case RMJ_EV_ANKAN: {  // :269-289
            PlayerState& P = g->players[actor];
            int n = (e.flags >> 4) & 15;
            Meld m;
            m.meld_type = MT_ANKAN;
            for (int k = 0; k < n && k < 4; k++) {
                m.tiles.push_back(e.consumed[k]);
                erase_first(P.hand, e.consumed[k]);
            }
            m.opened = false;
            m.from_who = -1;
            m.called_tile = -1;
            P.melds.push_back(m);
            g->current_player = (uint8_t)actor;
            g->phase = WAIT_ACT;
            g->active_players.assign(1, (uint8_t)actor);
            g->needs_tsumo = true;
            break;
        }
        case RMJ_EV_KAKAN: {  // :290-305
            PlayerState& P = g->players[actor];
            erase_first(P.hand, e.tile);
            for (auto& m : P.melds)
                if (m.meld_type == MT_PON && m.tiles[0] / 4 == e.tile / 4) {
                    m.meld_type = MT_KAKAN;
                    m.tiles.push_back(e.tile);
                    break;
                }
            g->current_player = (uint8_t)actor;
            g->phase = WAIT_ACT;
            g->active_players.assign(1, (uint8_t)actor);
            g->needs_tsumo = true;
            break;
        }
        case RMJ_EV_REACH: g->players[actor].riichi_stage = true; break;  // :306-310
        case RMJ_EV_REACH_ACCEPTED:                                        // :311-315
            g->players[actor].riichi_declared = true;
            g->riichi_sticks += 1;
            g->players[actor].score -= 1000;
            break;
        case RMJ_EV_DORA: g->wall.dora_indicators.push_back(e.tile); break;  // :316-319
        case RMJ_EV_KITA: {  // 4P: ignored (:320-322); 3P: state_3p/event_handler.rs:308-357
            if (!g->sanma) break;
            PlayerState& P = g->players[actor];
            int idx = -1;
            for (size_t i = 0; i < P.hand.size(); i++)
                if (P.hand[i] / 4 == 30) { idx = (int)i; break; }
            g->current_player = (uint8_t)actor;
            g->current_claims.clear();
            g->active_players.clear();
            if (idx >= 0) {
                uint8_t tile = P.hand[idx];
                P.hand.erase(P.hand.begin() + idx);
                P.kita_tiles.push_back(tile);
                apply_claims(g, actor, tile, true);
            } else {
                g->phase = WAIT_ACT;
                g->current_player = 0xFF;
            }
            g->needs_tsumo = true;
            break;
        }
        case RMJ_EV_HORA:
        case RMJ_EV_RYUKYOKU:
        case RMJ_EV_END_KYOKU: g->is_done = true; break;  // :323-325
        default: break;
    }
}

void orc_tid_to_mjai(uint8_t tid, char* buf) { std::strcpy(buf, tid_to_mjai(tid).c_str()); }

// ---------------------------------------------------------------- game
void* orc_game_new(int game_mode, int skip_log, uint64_t seed, int has_seed, int round_wind, uint32_t rule_bits) {
    std::optional<uint64_t> s;
    if (has_seed) s = seed;
    return new GameState((uint8_t)game_mode, skip_log != 0, s, (uint8_t)round_wind, GameRule::from_bits(rule_bits),
                         (rule_bits & RMJ_RULE_REFERENCE_RNG) != 0);
}
// WallState.salt / wall_digest (state/wall.rs:15-16): NUL-terminated, salt[17], digest[65]; both empty without RMJ_RULE_REFERENCE_RNG
void orc_game_wall_meta(void* gp, char* salt, char* digest) {
    GameState* g = (GameState*)gp;
    std::strcpy(salt, g->wall.salt.c_str());
    std::strcpy(digest, g->wall.wall_digest.c_str());
}
// ---- the pieces of ref_rng.hpp, exported so that tests can pin them on published vectors
void orc_chacha_block(const uint32_t* key8, uint64_t counter, uint64_t stream, int rounds, uint32_t* out16) {
    refrng::chacha_block(key8, counter, stream, rounds, out16);
}
void orc_seed_from_u64(uint64_t s, uint8_t* seed32) { refrng::seed_from_u64(s, seed32); }
void orc_stdrng_words(const uint8_t* seed32, uint32_t n, uint32_t* out) {
    refrng::StdRng r = refrng::StdRng::from_seed(seed32);
    for (uint32_t i = 0; i < n; i++) out[i] = r.next_u32();
}
void orc_sha256(const uint8_t* msg, uint64_t n, uint8_t* out32) {
    refrng::Sha256 h;
    h.update(msg, (size_t)n);
    h.finalize(out32);
}
uint32_t orc_random_range_u32(const uint8_t* seed32, uint32_t skip_words, uint32_t bound, uint32_t* words_used) {
    refrng::StdRng r = refrng::StdRng::from_seed(seed32);
    for (uint32_t i = 0; i < skip_words; i++) r.next_u32();
    uint64_t before = r.words_drawn;
    uint32_t v = refrng::random_range_u32(r, bound);
    *words_used = (uint32_t)(r.words_drawn - before);
    return v;
}
// w[n] (before the reversal), salt[17], digest[65]; returns the u32 words the generator handed out
uint32_t orc_reference_wall(uint64_t hand_seed, int sanma, uint8_t* w, char* salt, char* digest) {
    std::vector<uint8_t> ids;
    for (int i = 0; i < 136; i++)
        if (!(sanma && i / 4 >= 1 && i / 4 <= 7)) ids.push_back((uint8_t)i);
    refrng::RefWall r = refrng::reference_wall(hand_seed, ids);
    std::memcpy(w, r.w.data(), r.w.size());
    std::strcpy(salt, r.salt.c_str());
    std::strcpy(digest, r.digest.c_str());
    return (uint32_t)r.words_drawn;
}
void orc_game_free(void* g) { delete (GameState*)g; }

// wall: NULL or 136 tiles; oya/round_wind/honba/kyotaku <0 = default; scores NULL = default
void orc_game_reset(void* gp, const uint8_t* wall, int oya, int round_wind, const int32_t* scores, int honba, int kyotaku) {
    GameState* g = (GameState*)gp;
    std::vector<uint8_t> w;
    if (wall) w.assign(wall, wall + (g->sanma ? 108 : 136));
    std::vector<int32_t> sc;
    if (scores) sc.assign(scores, scores + g->NP);
    g->env_reset(oya, wall ? &w : nullptr, round_wind, scores ? &sc : nullptr, honba, kyotaku);
}

void orc_game_step(void* gp, const rmj_action_t* actions /*[4]*/) {
    GameState* g = (GameState*)gp;
    std::map<uint8_t, Action> acts;
    for (int p = 0; p < 4; p++)
        if (actions[p] != RMJ_NO_ACTION && (actions[p] & 0xFF) != 0xFF) acts[(uint8_t)p] = unpack_action(actions[p], p);
    g->step(acts);
}

int orc_game_legal(void* gp, int pid, rmj_action_t* out /*[64]*/) {
    GameState* g = (GameState*)gp;
    auto l = g->_get_legal_actions_internal((uint8_t)pid);
    int n = 0;
    for (auto& a : l) {
        if (n < RMJ_MAX_LEGAL) out[n] = pack_action(a);
        n++;
    }
    return n;
}
// observation/python.rs:98-111
int orc_game_mask(void* gp, int pid, uint8_t* mask82) {
    GameState* g = (GameState*)gp;
    std::memset(mask82, 0, 82);
    for (auto& a : g->_get_legal_actions_internal((uint8_t)pid)) {
        int id = g->sanma ? a.encode_3p() : a.encode();  // observation_3p/python.rs:100-112: 60 ids in 3P
        if (id >= 0 && id < (g->sanma ? 60 : 82)) mask82[id] = 1;
    }
    return 0;
}
int orc_action_encode(rmj_action_t a) { return unpack_action(a, -1).encode(); }
int orc_action_encode_3p(rmj_action_t a) { return unpack_action(a, -1).encode_3p(); }

uint64_t orc_game_waits(void* gp, int pid) {
    GameState* g = (GameState*)gp;
    uint64_t w = 0;
    for (uint8_t t : g->observation_waits(pid)) w |= 1ull << t;
    return w;
}

void orc_game_status(void* gp, uint8_t* active_mask, uint8_t* phase, uint8_t* done) {
    GameState* g = (GameState*)gp;
    uint8_t m = 0;
    // env.rs:870-871: observations are produced for active_players (after a finished game: legal list empty)
    for (uint8_t p : g->active_players) m |= 1u << p;
    *active_mask = m;
    *phase = g->phase;
    *done = g->is_done;
}
uint64_t orc_game_step_count(void* gp) { return ((GameState*)gp)->step_count; }

void orc_game_peek(void* gp, RmjStateView* v) {
    GameState* g = (GameState*)gp;
    std::memset(v, 0, sizeof(*v));
    v->wall_len = (uint8_t)g->wall.tiles.size();
    for (size_t i = 0; i < g->wall.tiles.size() && i < 136; i++) v->wall[i] = g->wall.tiles[i];
    v->n_dora = (uint8_t)g->wall.dora_indicators.size();
    for (size_t i = 0; i < g->wall.dora_indicators.size() && i < 5; i++) v->dora[i] = g->wall.dora_indicators[i];
    v->rinshan_draw_count = g->wall.rinshan_draw_count;
    v->pending_kan_dora_count = g->wall.pending_kan_dora_count;
    v->drawable_count = g->wall.drawable_count;
    v->wall_seed = g->wall.seed ? *g->wall.seed : 0;
    v->hand_index = g->wall.hand_index;
    for (int p = 0; p < 4; p++) {
        const PlayerState& P = g->players[p];
        RmjPlayerView& q = v->players[p];
        q.hand_len = (uint8_t)P.hand.size();
        for (size_t i = 0; i < P.hand.size() && i < 14; i++) q.hand[i] = P.hand[i];
        q.n_melds = (uint8_t)P.melds.size();
        for (size_t i = 0; i < P.melds.size() && i < 4; i++) meld_to_view(P.melds[i], q.melds[i]);
        q.n_discards = (uint8_t)P.discards.size();
        for (size_t i = 0; i < P.discards.size() && i < RMJ_MAX_DISCARDS; i++) {
            q.discards[i] = P.discards[i];
            // (apply_mjai_event pushes to `discards` only: the flag vectors may be shorter)
            if (i < P.discard_from_hand.size() && P.discard_from_hand[i]) q.discard_from_hand_bits |= 1u << i;
            if (i < P.discard_is_riichi.size() && P.discard_is_riichi[i]) q.discard_is_riichi_bits |= 1u << i;
        }
        q.riichi_declaration_index = (int8_t)P.riichi_declaration_index;
        q.score = P.score;
        q.score_delta = P.score_delta;
        q.riichi_declared = P.riichi_declared;
        q.riichi_stage = P.riichi_stage;
        q.double_riichi_declared = P.double_riichi_declared;
        q.missed_agari_riichi = P.missed_agari_riichi;
        q.missed_agari_doujun = P.missed_agari_doujun;
        q.nagashi_eligible = P.nagashi_eligible;
        q.ippatsu_cycle = P.ippatsu_cycle;
        q.pao_daisangen = q.pao_daisuushi = -1;
        auto f = P.pao.find(37);
        if (f != P.pao.end()) q.pao_daisangen = (int8_t)f->second;
        f = P.pao.find(50);
        if (f != P.pao.end()) q.pao_daisuushi = (int8_t)f->second;
        q.n_forbidden = (uint8_t)P.forbidden_discards.size();
        for (size_t i = 0; i < P.forbidden_discards.size() && i < 2; i++) q.forbidden[i] = P.forbidden_discards[i];
        q.riichi_sutehai = (int16_t)g->riichi_sutehais[p];
        q.last_tedashi = (int16_t)g->last_tedashis[p];
        q.n_kita = (uint8_t)P.kita_tiles.size();
        for (size_t i = 0; i < P.kita_tiles.size() && i < 4; i++) q.kita[i] = P.kita_tiles[i];
    }
    v->current_player = g->current_player;
    v->is_done = g->is_done;
    v->needs_tsumo = g->needs_tsumo;
    v->phase = g->phase;
    for (uint8_t p : g->active_players) v->active_mask |= 1u << p;
    v->turn_count = g->turn_count;
    v->riichi_sticks = g->riichi_sticks;
    v->last_discard_pid = g->last_discard ? g->last_discard->first : -1;
    v->last_discard_tile = g->last_discard ? g->last_discard->second : -1;
    v->pending_kan_pid = g->pending_kan ? g->pending_kan->first : -1;
    v->pending_kan_action = g->pending_kan ? pack_action(g->pending_kan->second) : 0;
    v->oya = g->oya;
    v->honba = g->honba;
    v->kyoku_idx = g->kyoku_idx;
    v->round_wind = g->round_wind;
    v->is_rinshan_flag = g->is_rinshan_flag;
    v->is_first_turn = g->is_first_turn;
    v->riichi_pending_acceptance = (int16_t)g->riichi_pending_acceptance;
    v->drawn_tile = (int16_t)g->drawn_tile;
    v->last_error_pid = -1;
    if (g->last_error) v->last_error_pid = (int16_t)std::atoi(g->last_error->c_str() + 32);
}

// test hook mirroring the reference's Python setters (env.rs:134-622)
void orc_game_poke(void* gp, const RmjStateView* v) {
    GameState* g = (GameState*)gp;
    g->wall.tiles.assign(v->wall, v->wall + v->wall_len);
    g->wall.dora_indicators.assign(v->dora, v->dora + v->n_dora);
    g->wall.rinshan_draw_count = v->rinshan_draw_count;
    g->wall.pending_kan_dora_count = v->pending_kan_dora_count;
    g->wall.drawable_count = v->drawable_count;
    g->wall.hand_index = v->hand_index;
    for (int p = 0; p < 4; p++) {
        PlayerState& P = g->players[p];
        const RmjPlayerView& q = v->players[p];
        P.hand.assign(q.hand, q.hand + q.hand_len);
        P.melds.clear();
        for (int i = 0; i < q.n_melds; i++) P.melds.push_back(meld_from_view(q.melds[i]));
        P.discards.assign(q.discards, q.discards + q.n_discards);
        P.discard_from_hand.clear();
        P.discard_is_riichi.clear();
        for (int i = 0; i < q.n_discards; i++) {
            P.discard_from_hand.push_back((q.discard_from_hand_bits >> i) & 1);
            P.discard_is_riichi.push_back((q.discard_is_riichi_bits >> i) & 1);
        }
        P.riichi_declaration_index = q.riichi_declaration_index;
        P.score = q.score;
        P.score_delta = q.score_delta;
        P.riichi_declared = q.riichi_declared;
        P.riichi_stage = q.riichi_stage;
        P.double_riichi_declared = q.double_riichi_declared;
        P.missed_agari_riichi = q.missed_agari_riichi;
        P.missed_agari_doujun = q.missed_agari_doujun;
        P.nagashi_eligible = q.nagashi_eligible;
        P.ippatsu_cycle = q.ippatsu_cycle;
        P.pao.clear();
        if (q.pao_daisangen >= 0) P.pao[37] = (uint8_t)q.pao_daisangen;
        if (q.pao_daisuushi >= 0) P.pao[50] = (uint8_t)q.pao_daisuushi;
        P.forbidden_discards.assign(q.forbidden, q.forbidden + q.n_forbidden);
        g->riichi_sutehais[p] = q.riichi_sutehai;
        g->last_tedashis[p] = q.last_tedashi;
        P.kita_tiles.assign(q.kita, q.kita + q.n_kita);
    }
    g->current_player = v->current_player;
    g->is_done = v->is_done;
    g->needs_tsumo = v->needs_tsumo;
    g->phase = (Phase)v->phase;
    g->active_players.clear();
    for (int p = 0; p < 4; p++)
        if (v->active_mask & (1u << p)) g->active_players.push_back((uint8_t)p);
    g->turn_count = v->turn_count;
    g->riichi_sticks = v->riichi_sticks;
    if (v->last_discard_pid >= 0)
        g->last_discard = std::make_pair((uint8_t)v->last_discard_pid, (uint8_t)v->last_discard_tile);
    else
        g->last_discard.reset();
    if (v->pending_kan_pid >= 0)
        g->pending_kan = std::make_pair((uint8_t)v->pending_kan_pid, unpack_action(v->pending_kan_action, v->pending_kan_pid));
    else
        g->pending_kan.reset();
    g->oya = v->oya;
    g->honba = v->honba;
    g->kyoku_idx = v->kyoku_idx;
    g->round_wind = v->round_wind;
    g->is_rinshan_flag = v->is_rinshan_flag;
    g->is_first_turn = v->is_first_turn;
    g->riichi_pending_acceptance = v->riichi_pending_acceptance;
    g->drawn_tile = v->drawn_tile;
    // claims are recomputed for a WaitResponse poke from the last discard (what the reference tests do by hand)
    g->current_claims.clear();
    if (g->phase == WAIT_RESPONSE && g->last_discard && !g->pending_kan) {
        for (uint8_t i = 0; i < 4; i++) {
            if (i == g->last_discard->first) continue;
            if (!(v->active_mask & (1u << i))) continue;
            auto r = g->_get_claim_actions_for_player(i, g->last_discard->first, g->last_discard->second);
            if (!r.first.empty()) g->current_claims[i] = r.first;
        }
    }
}

// observation/helpers.rs:24-50 (136-id in, 136-id out)
static uint8_t obs_get_next_tile(uint32_t tile) {
    uint32_t tile_type = (tile / 4) / 9, tile_num = (tile / 4) % 9;
    if (tile_type < 3) {
        uint32_t next_num = tile_num == 8 ? 0 : tile_num + 1;
        return (uint8_t)((tile_type * 9 + next_num) * 4);
    }
    uint32_t base = tile / 4;
    if (base >= 27 && base < 31) return (uint8_t)((27 + (base - 27 + 1) % 4) * 4);
    if (base >= 31 && base < 34) return (uint8_t)((31 + (base - 31 + 1) % 3) * 4);
    return (uint8_t)tile;
}

// observation_3p/helpers.rs:38-47
static uint8_t obs_get_next_tile_sanma(uint32_t tile) {
    uint32_t t34 = tile / 4;
    if (t34 == 0) return 32;
    if (t34 == 8) return 0;
    if (t34 < 8) return (uint8_t)tile;
    return obs_get_next_tile(tile);
}

// Observation.encode(): observation/python.rs:457-806 (4P, 74x34) and observation_3p/python.rs:395-... (3P, 74x27,
// compact column index observation_3p/helpers.rs:7-15) over the snapshot of state/mod.rs:189-263
void orc_game_encode(void* gp, int pid, float* arr) {
    GameState* g = (GameState*)gp;
    const bool sanma = g->sanma;
    const int W = sanma ? 27 : 34, NP = g->NP;
    std::memset(arr, 0, sizeof(float) * 74 * W);
    auto col = [&](int t34) -> int { return sanma ? (t34 == 0 ? 0 : (t34 >= 8 && t34 < 34 ? t34 - 7 : -1)) : (t34 < 34 ? t34 : -1); };
    auto set1 = [&](int ch, int t34) {
        int c = col(t34);
        if (c >= 0) arr[ch * W + c] = 1.0f;
    };
    auto bc = [&](int ch, float v) {
        for (int k = 0; k < W; k++) arr[ch * W + k] = v;
    };
    auto next = [&](uint8_t di) { return sanma ? obs_get_next_tile_sanma(di) : obs_get_next_tile(di); };
    const PlayerState& P = g->players[pid];
    std::vector<int> rel;
    for (int i = 0; i < NP; i++) rel.push_back((pid + i) % NP);
    std::vector<uint8_t> counts(W, 0);
    for (uint8_t t : P.hand) {
        int c = col(t / 4);
        if (c >= 0) {
            counts[c]++;
            if (t == 16 || t == 52 || t == 88) arr[4 * W + c] = 1.0f;
        }
    }
    for (int i = 0; i < W; i++)
        for (int k = 0; k < 4; k++)
            if (counts[i] >= k + 1) arr[k * W + i] = 1.0f;
    for (size_t m = 0; m < P.melds.size() && m < 4; m++)
        for (uint8_t t : P.melds[m].tiles) set1(5 + (int)m, t / 4);
    for (uint8_t t : g->wall.dora_indicators) set1(9, t / 4);
    {
        const auto& d = P.discards;
        for (size_t i = 0; i < 4 && i < d.size(); i++) set1(10 + (int)i, d[d.size() - 1 - i] / 4);
    }
    for (int i = 1; i < NP; i++) {
        const auto& d = g->players[(pid + i) % NP].discards;
        for (size_t j = 0; j < 4 && j < d.size(); j++) set1(14 + (i - 1) * 4 + (int)j, d[d.size() - 1 - j] / 4);
    }
    for (int c = 0; c < NP; c++) bc(26 + c, (float)g->players[rel[c]].discards.size() / 24.0f);
    int tiles_used = 0;
    for (int q = 0; q < NP; q++) {
        tiles_used += (int)g->players[q].discards.size();
        for (auto& m : g->players[q].melds) tiles_used += (int)m.tiles.size();
    }
    tiles_used += (int)P.hand.size() + (int)g->wall.dora_indicators.size();
    bc(30, (float)std::max((sanma ? 108 : 136) - tiles_used, 0) / 70.0f);
    if (P.riichi_declared) bc(31, 1.0f);
    for (int i = 1; i < NP; i++)
        if (g->players[(pid + i) % NP].riichi_declared) bc(32 + (i - 1), 1.0f);
    if (27 + g->round_wind < 34) set1(35, 27 + g->round_wind);
    set1(36, 27 + (pid + NP - g->oya) % NP);
    bc(37, (float)g->honba / 10.0f);
    bc(38, (float)g->riichi_sticks / 5.0f);
    for (int c = 0; c < NP; c++) {
        int32_t sc = g->players[rel[c]].score;
        bc(39 + c, (float)std::min(std::max(sc, 0), 100000) / 100000.0f);
        bc(43 + c, (float)std::min(std::max(sc, 0), 30000) / 30000.0f);
    }
    std::vector<uint8_t> waits = g->observation_waits(pid);
    for (uint8_t t : waits) set1(47, t);
    bc(48, waits.empty() ? 0.0f : 1.0f);
    int rank = 0;
    for (int q = 0; q < NP; q++)
        if (g->players[q].score > P.score) rank++;
    if (rank < NP) bc(49 + rank, 1.0f);
    bc(53, (float)g->kyoku_idx / 8.0f);
    bc(54, ((float)g->round_wind * 4.0f + (float)g->kyoku_idx) / 7.0f);
    uint8_t dora_counts[4] = {0, 0, 0, 0};
    for (int q = 0; q < NP; q++) {
        for (auto& m : g->players[q].melds)
            for (uint8_t tile : m.tiles)
                for (uint8_t di : g->wall.dora_indicators)
                    if ((tile / 4) == (next(di) / 4)) dora_counts[q]++;
        for (uint8_t tile : g->players[q].discards)
            for (uint8_t di : g->wall.dora_indicators)
                if ((tile / 4) == (next(di) / 4)) dora_counts[q]++;
    }
    for (uint8_t tile : P.hand)
        for (uint8_t di : g->wall.dora_indicators)
            if ((tile / 4) == (next(di) / 4)) dora_counts[pid]++;
    for (int c = 0; c < NP; c++) {
        bc(55 + c, (float)dora_counts[rel[c]] / 12.0f);
        bc(59 + c, (float)g->players[rel[c]].melds.size() / 4.0f);
    }
    std::vector<uint8_t> seen(W, 0);
    auto see = [&](uint8_t t) {
        int c = col(t / 4);
        if (c >= 0) seen[c]++;
    };
    for (uint8_t t : P.hand) see(t);
    for (int q = 0; q < NP; q++) {
        for (auto& m : g->players[q].melds)
            for (uint8_t t : m.tiles) see(t);
        for (uint8_t t : g->players[q].discards) see(t);
    }
    for (uint8_t t : g->wall.dora_indicators) see(t);
    for (int i = 0; i < W; i++) arr[63 * W + i] = (float)seen[i] / 4.0f;
    {
        const auto& d = P.discards;
        for (size_t i = 0; i < 4 && 4 + i < d.size(); i++) set1(64 + (int)i, d[d.size() - 1 - (4 + i)] / 4);
        const auto& e = g->players[(pid + 1) % NP].discards;
        for (size_t i = 0; i < 2 && 4 + i < e.size(); i++) set1(68 + (int)i, e[e.size() - 1 - (4 + i)] / 4);
    }
    // ch 70-73: tsumogiri_flags is never filled (observation/mod.rs:105) -> zeros
}

// Observation.encode_extended(): 215 channels = encode() + the nine blocks of observation/encode.rs:293-585 (4P, 34
// columns) / observation_3p/encode.rs:315-615 (3P, 27 compact columns), python.rs:1271-1296.
// Snapshot quirks restated: Observation.last_discard receives the DISCARDER'S SEAT, not the tile
// (state/mod.rs:252 destructures (pid, tile) as (tile, _pid)); dora membership compares full 136-ids with the copy-0
// id of the next tile; riichi_sutehais is only written on an unreachable branch (state/mod.rs:466).
// ---- auxiliary encoders (absolute seat order, public information only: the same for every observing seat) ----
// Observation.encode_kawa_overview (observation/python.rs:881-925, observation_3p/python.rs:759-810): out [NP][7][W].
// Channels 0..3: "at least k+1 discards of this type"; channels 4..6: red-five flags.  Quirks reproduced: the reference
// tests the ids 20 / 24 / 28 (not 16 / 52 / 88) and, in 4P, marks column 5 + 9 i (6m / 6p / 6s); 3P marks channel 5
// column 6 and channel 6 column 15, and never channel 4.
void orc_game_encode_kawa_overview(void* gp, float* out) {
    GameState* g = (GameState*)gp;
    const bool sanma = g->sanma;
    const int W = sanma ? 27 : 34, NP = g->NP;
    std::memset(out, 0, sizeof(float) * NP * 7 * W);
    auto col = [&](int t34) -> int { return sanma ? (t34 == 0 ? 0 : (t34 >= 8 && t34 < 34 ? t34 - 7 : -1)) : (t34 < 34 ? t34 : -1); };
    for (int p = 0; p < NP; p++) {
        float* a = out + (size_t)p * 7 * W;
        int cnt[34] = {0};
        bool aka[3] = {false, false, false};
        for (uint8_t t : g->players[p].discards) {
            const int c = col(t / 4);
            if (c >= 0) {
                const int k = cnt[c] < 3 ? cnt[c] : 3;
                a[k * W + c] = 1.0f;
                if (cnt[c] < 255) cnt[c]++;
            }
            if (t == 20) aka[0] = true;
            if (t == 24) aka[1] = true;
            if (t == 28) aka[2] = true;
        }
        if (!sanma) {
            for (int i = 0; i < 3; i++)
                if (aka[i]) a[(4 + i) * W + 5 + i * 9] = 1.0f;
        } else {
            if (aka[1]) a[5 * W + col(13)] = 1.0f;
            if (aka[2]) a[6 * W + col(22)] = 1.0f;
        }
    }
}
// Observation.encode_yaku_possibility (observation/python.rs:327-455, observation_3p/python.rs:275-400) over
// yaku_checker.rs:27-412: out [NP][21][2], 0.0 = "impossible given melds / discards / dora indicators", else 1.0
// (Possible and Unknown both encode as 1.0, yaku_checker.rs:18-25); tsumo and ron columns are equal.
void orc_game_encode_yaku_possibility(void* gp, float* out) {
    GameState* g = (GameState*)gp;
    const int NP = g->NP;
    auto yaochu = [](int tt) { return tt % 9 == 0 || tt % 9 == 8 || tt >= 27; };
    for (int p = 0; p < NP; p++) {
        const PlayerState& P = g->players[p];
        const auto& melds = P.melds;
        bool imp[21];
        for (bool& b : imp) b = false;
        auto visible = [&](int tt) {  // yaku_checker.rs:42-58: own discards + dora indicators
            int n = 0;
            for (uint8_t t : P.discards) n += (t / 4 == tt);
            for (uint8_t t : g->wall.dora_indicators) n += (t / 4 == tt);
            return n;
        };
        auto has_set = [&](int tt) {  // yaku_checker.rs:68-75
            for (const Meld& m : melds)
                if (!m.tiles.empty() && m.tiles[0] / 4 == tt && m.tiles.size() >= 3) return true;
            return false;
        };
        auto yakuhai_imp = [&](int tt) { return !has_set(tt) && visible(tt) >= 3; };
        // 0 tanyao (:28-39)
        for (const Meld& m : melds)
            for (uint8_t t : m.tiles)
                if (yaochu(t / 4)) imp[0] = true;
        // 1..3 dragons, 4 round wind, 5 seat wind (:61-84; python.rs:347-372)
        for (int k = 0; k < 3; k++) imp[1 + k] = yakuhai_imp(31 + k);
        imp[4] = yakuhai_imp(27 + g->round_wind);
        imp[5] = yakuhai_imp(27 + (p + NP - g->oya) % NP);
        // 6 honitsu, 7 chinitsu (:87-138)
        if (!melds.empty()) {
            bool suit[3] = {false, false, false}, honor = false;
            for (const Meld& m : melds)
                for (uint8_t t : m.tiles) {
                    if (t / 4 < 27) suit[t / 4 / 9] = true;
                    else honor = true;
                }
            const int ns = (int)suit[0] + suit[1] + suit[2];
            imp[6] = ns >= 2;
            imp[7] = ns >= 2 || (ns == 1 && honor);
        }
        // 8 toitoi (:141-166): a meld of three consecutive types
        for (const Meld& m : melds)
            if (m.tiles.size() == 3) {
                const int t0 = m.tiles[0] / 4, t1 = m.tiles[1] / 4, t2 = m.tiles[2] / 4;
                if (t0 + 1 == t1 && t1 + 1 == t2 && t0 < 27) imp[8] = true;
            }
        // 9 chiitoitsu (:169-175), 19 iipeikou (:385-392): any meld
        imp[9] = imp[19] = !melds.empty();
        // 10 shousangen (:178-201): some dragon fully visible
        for (int k = 0; k < 3; k++)
            if (visible(31 + k) >= 4) imp[10] = true;
        // 11 daisangen (:204-235): a dragon without a set and two or more visible
        for (int k = 0; k < 3; k++)
            if (!has_set(31 + k) && visible(31 + k) >= 2) imp[11] = true;
        // 12 tsuuiisou (:238-249), 13 chinroutou (:252-263), 14 honroutou (:266-278)
        for (const Meld& m : melds)
            for (uint8_t t : m.tiles) {
                const int tt = t / 4;
                if (tt < 27) imp[12] = true;
                if (tt >= 27 || (tt % 9 != 0 && tt % 9 != 8)) imp[13] = true;
                if (!yaochu(tt)) imp[14] = true;
            }
        // 15 kokushi (:281-300)
        if (!melds.empty()) imp[15] = true;
        else {
            static const int req[13] = {0, 8, 9, 17, 18, 26, 27, 28, 29, 30, 31, 32, 33};
            for (int tt : req)
                if (visible(tt) >= 4) imp[15] = true;
        }
        // 16 chanta (:303-328), 17 junchan (:331-359)
        for (const Meld& m : melds) {
            if (m.tiles.empty()) continue;
            bool any_yaochu = false, any_terminal = false, any_honor = false;
            for (uint8_t t : m.tiles) {
                const int tt = t / 4;
                if (yaochu(tt)) any_yaochu = true;
                if (tt >= 27) any_honor = true;
                else if (tt % 9 == 0 || tt % 9 == 8) any_terminal = true;
            }
            if (!any_yaochu) imp[16] = true;
            if (any_honor || !any_terminal) imp[17] = true;
        }
        // 18 sanshoku, 20 ittsu: never impossible (:363-382, :395-412)
        for (int y = 0; y < 21; y++) out[((size_t)p * 21 + y) * 2] = out[((size_t)p * 21 + y) * 2 + 1] = imp[y] ? 0.0f : 1.0f;
    }
}
// Observation.encode_furiten_ron_possibility (observation/python.rs:251-293): out [NP][21].  It zeroes a seat's row after
// three consecutive tsumogiri in `tsumogiri_flags`, which the reference never fills (observation/mod.rs:105): all ones.
void orc_game_encode_furiten_ron(void* gp, float* out) {
    GameState* g = (GameState*)gp;
    for (int i = 0; i < g->NP * 21; i++) out[i] = 1.0f;
}

void orc_game_encode_extended(void* gp, int pid, float* arr) {
    GameState* g = (GameState*)gp;
    const bool sanma = g->sanma;
    const int W = sanma ? 27 : 34, NP = g->NP;
    std::memset(arr, 0, sizeof(float) * 215 * W);
    orc_game_encode(gp, pid, arr);
    {   // encode_base_into (observation/encode.rs:12-293, observation_3p/encode.rs) is Observation.encode() (python.rs:457-806) except
        // for channel 30: it does not count a meld's called tile twice - "already counted in discards" (encode.rs:94-111)
        int used = (int)g->players[pid].hand.size() + (int)g->wall.dora_indicators.size();
        for (int q = 0; q < NP; q++) {
            used += (int)g->players[q].discards.size();
            for (auto& m : g->players[q].melds) used += (int)m.tiles.size() - (m.called_tile >= 0 ? 1 : 0);
        }
        for (int k = 0; k < W; k++) arr[30 * W + k] = (float)std::max((sanma ? 108 : 136) - used, 0) / 70.0f;
    }
    auto col = [&](int t34) -> int { return sanma ? (t34 == 0 ? 0 : (t34 >= 8 && t34 < 34 ? t34 - 7 : -1)) : (t34 < 34 ? t34 : -1); };
    auto bc = [&](int ch, float v) {
        for (int k = 0; k < W; k++) arr[ch * W + k] = v;
    };
    auto next = [&](uint8_t di) { return sanma ? obs_get_next_tile_sanma(di) : obs_get_next_tile(di); };
    const PlayerState& P = g->players[pid];
    std::vector<int> rel;
    for (int i = 0; i < NP; i++) rel.push_back((pid + i) % NP);
    // 74..77 discard history decay (encode.rs:295-313)
    for (int c = 0; c < NP; c++) {
        const auto& d = g->players[rel[c]].discards;
        for (size_t turn = 0; turn < d.size(); turn++) {
            int k = col(d[turn] / 4);
            if (k < 0) continue;
            float age = (float)(d.size() - 1 - turn);
            arr[(74 + c) * W + k] += std::exp(-0.2f * age);
        }
    }
    // 78..93 shanten efficiency (encode.rs:317-350)
    uint8_t hc[34] = {0}, vis[34] = {0};
    for (uint8_t t : P.hand) hc[t / 4]++;
    auto see = [&](uint8_t t) { if (t / 4 < 34 && vis[t / 4] < 255) vis[t / 4]++; };
    for (int q = 0; q < NP; q++) {
        for (uint8_t t : g->players[q].discards) see(t);
        for (auto& m : g->players[q].melds)
            for (uint8_t t : m.tiles) see(t);
    }
    for (uint8_t t : g->wall.dora_indicators) see(t);
    const int cur_sh = shanten_of(hc, sanma);
    for (int c = 0; c < NP; c++) {
        int base = 78 + c * 4;
        if (rel[c] == pid) {
            uint32_t eff = effective_tiles_with_discard(hc, sanma);  // (a 3n hand would panic in the reference)
            uint32_t uke = best_ukeire(hc, vis, sanma);
            bc(base, std::max((float)cur_sh, 0.0f) / 8.0f);
            bc(base + 1, (float)eff / (sanma ? 27.0f : 34.0f));
            bc(base + 2, (float)uke / 80.0f);
        } else {
            bc(base, 0.5f); bc(base + 1, 0.5f); bc(base + 2, 0.5f);
        }
        bc(base + 3, std::min((float)g->players[rel[c]].discards.size() / 18.0f, 1.0f));
    }
    // 94..97 ankan overview (encode.rs:354-368), 98..177 fuuro overview (:372-396)
    for (int c = 0; c < NP; c++) {
        const auto& ms = g->players[rel[c]].melds;
        for (size_t mi = 0; mi < ms.size(); mi++) {
            const Meld& m = ms[mi];
            if (m.meld_type == MT_ANKAN && !m.tiles.empty()) {
                int k = col(m.tiles[0] / 4);
                if (k >= 0) arr[(94 + c) * W + k] = 1.0f;
            }
            if (mi >= 4) continue;
            for (size_t sl = 0; sl < m.tiles.size() && sl < 4; sl++) {
                uint8_t t = m.tiles[sl];
                int k = col(t / 4);
                if (k >= 0) arr[(98 + c * 20 + (int)mi * 5 + (int)sl) * W + k] = 1.0f;
                if ((t == 16 || t == 52 || t == 88) && k >= 0) arr[(98 + c * 20 + (int)mi * 5 + 4) * W + k] = 1.0f;
            }
        }
    }
    // 178..188 action availability (encode.rs:399-429) over the observation's legal actions (state/mod.rs:199-208)
    std::vector<Action> legal;
    if (!g->is_done) {
        bool in_act = false;
        for (uint8_t a : g->active_players) in_act = in_act || a == pid;
        if ((g->phase == WAIT_ACT && g->current_player == pid) || (g->phase == WAIT_RESPONSE && in_act))
            legal = g->_get_legal_actions_internal((uint8_t)pid);
    }
    for (const Action& a : legal) {
        switch (a.type) {
            case AT_RIICHI: bc(178, 1.0f); break;
            case AT_CHI:
                if (a.consume.size() == 2) {
                    int t0 = a.consume[0] / 4, t1 = a.consume[1] / 4, diff = std::abs(t1 - t0);
                    if (diff == 1) bc(t0 < t1 ? 179 : 181, 1.0f);
                    else if (diff == 2) bc(180, 1.0f);
                }
                break;
            case AT_PON: bc(182, 1.0f); break;
            case AT_DAIMINKAN: bc(183, 1.0f); break;
            case AT_ANKAN: bc(184, 1.0f); break;
            case AT_KAKAN: bc(185, 1.0f); break;
            case AT_TSUMO:
            case AT_RON: bc(186, 1.0f); break;
            case AT_KYUSHU: bc(187, 1.0f); break;
            case AT_PASS: bc(188, 1.0f); break;
            default: break;
        }
    }
    // 189..193 discard candidates (encode.rs:433-475)
    {
        const size_t n = P.hand.size();
        bc(189, (float)n / 34.0f);
        int keep = 0, inc = 0;
        uint8_t t[34];
        std::memcpy(t, hc, 34);
        for (uint8_t tile : P.hand) {
            t[tile / 4]--;
            int ns = shanten_of(t, sanma);
            t[tile / 4]++;
            if (ns == cur_sh) keep++;
            else if (ns > cur_sh) inc++;
        }
        if (n) {
            bc(190, (float)keep / (float)n);
            bc(191, (float)inc / (float)n);
        }
        bc(192, cur_sh == -1 ? 1.0f : 0.0f);
        bc(193, P.riichi_declared ? 1.0f : 0.0f);
    }
    // 194..196 pass context (:479-509), 197..205 last tedashis (:513-546), 206..214 riichi sutehais (:550-585)
    std::vector<uint8_t> dora_tiles;
    for (uint8_t di : g->wall.dora_indicators) dora_tiles.push_back(next(di));
    auto is_dora = [&](uint8_t tile) { return std::find(dora_tiles.begin(), dora_tiles.end(), tile) != dora_tiles.end(); };
    auto tile_feats = [&](int ch, int tile) {
        int t34 = tile / 4;
        if (sanma) {
            int k = col(t34);
            if (k >= 0) bc(ch, (float)k / 26.0f);
        } else {
            bc(ch, (float)t34 / 33.0f);
        }
        bc(ch + 1, (tile == 16 || tile == 52 || tile == 88) ? 1.0f : 0.0f);
        bc(ch + 2, is_dora((uint8_t)tile) ? 1.0f : 0.0f);
    };
    if (g->last_discard) tile_feats(194, g->last_discard->first);  // the seat number, see the header comment
    int opp = 0;
    for (int q = 0; q < NP; q++) {
        if (q == pid) continue;
        if (g->last_tedashis[q] >= 0) tile_feats(197 + opp * 3, g->last_tedashis[q]);
        if (g->riichi_sutehais[q] >= 0) tile_feats(206 + opp * 3, g->riichi_sutehais[q]);
        opp++;
    }
}

uint32_t orc_game_log_len(void* gp, int seat) {
    GameState* g = (GameState*)gp;
    return (uint32_t)(seat < 0 ? g->mjai_log.size() : g->mjai_log_per_player[seat].size());
}
int orc_game_log_get(void* gp, int seat, uint32_t idx, char* buf, uint32_t cap) {
    GameState* g = (GameState*)gp;
    const std::string& s = seat < 0 ? g->mjai_log[idx] : g->mjai_log_per_player[seat][idx];
    if (s.size() + 1 > cap) return -1;
    std::memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}

// Random-policy actions for the current state (same definition as rmj_random_actions)
void orc_game_random_actions(void* gp, uint64_t policy_seed, uint64_t global_game, rmj_action_t* out /*[4]*/) {
    GameState* g = (GameState*)gp;
    for (int p = 0; p < 4; p++) out[p] = RMJ_NO_ACTION;
    if (g->is_done) return;
    for (uint8_t p : g->active_players) {
        auto l = g->_get_legal_actions_internal(p);
        if (l.empty()) continue;
        uint64_t c = policy_choice(policy_seed, global_game, g->step_count, p, (uint32_t)l.size());
        out[p] = pack_action(l[c]);
    }
}

// The greedy policy of rmj_step_greedy, restated (definition: include/riichi_mi355x.h; device twin: r4_policy_greedy).  Test
// infrastructure like the rest of this file: the checker's own legal lists and its own shanten (riichi_shanten.hpp).
static int greedy_class(ActionType t, bool call) {
    switch (t) {
        case AT_TSUMO: case AT_RON: return 0;
        case AT_KITA: return 1;
        case AT_RIICHI: return 2;
        case AT_ANKAN: return 3;
        case AT_KAKAN: return 4;
        case AT_DAIMINKAN: return 5;
        case AT_PON: return call ? 6 : 12;
        case AT_CHI: return call ? 7 : 12;
        case AT_DISCARD: return 8;
        case AT_PASS: return 9;
        case AT_KYUSHU: return 10;
        default: return 15;
    }
}
static size_t greedy_pick(const GameState& g, uint8_t p, const std::vector<Action>& l, uint32_t key, uint32_t call_rate) {
    const bool call = (key >> 24) < call_rate;
    int best_c = 99;
    size_t best_i = 0;
    for (size_t i = 0; i < l.size(); i++) {
        const int c = greedy_class(l[i].type, call);
        if (c < best_c) { best_c = c; best_i = i; }
    }
    if (best_c != 8) return best_i;
    std::vector<size_t> cand;
    for (size_t i = 0; i < l.size(); i++)
        if (l[i].type == AT_DISCARD) cand.push_back(i);
    if (cand.size() < 2) return best_i;
    const auto& hand = g.players[p].hand;
    const int len3 = ((int)hand.size() - 1) / 3;
    int smin = 99;
    std::vector<int> sh(cand.size());
    for (size_t k = 0; k < cand.size(); k++) {
        uint8_t cnt[34] = {0};
        bool removed = false;
        for (uint8_t t : hand) {
            if (!removed && (int)t == l[cand[k]].tile) { removed = true; continue; }
            cnt[t / 4]++;
        }
        sh[k] = calc_shanten_from_counts(cnt, len3, g.sanma);
        smin = std::min(smin, sh[k]);
    }
    std::vector<size_t> tie;
    for (size_t k = 0; k < cand.size(); k++)
        if (sh[k] == smin) tie.push_back(cand[k]);
    return tie[policy_pick(key * 0x9E3779B1u, (uint32_t)tie.size())];
}
void orc_game_greedy_actions(void* gp, uint64_t policy_seed, uint64_t global_game, uint32_t call_rate_256, rmj_action_t* out /*[4]*/) {
    GameState* g = (GameState*)gp;
    for (int p = 0; p < 4; p++) out[p] = RMJ_NO_ACTION;
    if (g->is_done) return;
    for (uint8_t p : g->active_players) {
        auto l = g->_get_legal_actions_internal(p);
        if (l.empty()) continue;
        const uint32_t key = policy_key32(splitmix64(policy_seed + global_game), (uint32_t)g->step_count, p);
        out[p] = pack_action(l[greedy_pick(*g, p, l, key, call_rate_256)]);
    }
}

// RiichiEnv.win_results (env.rs:606-607): out[4] by seat, returns the seat mask
int orc_game_win_results(void* h, RmjWinResult* out) {
    auto* g = (GameState*)h;
    int mask = 0;
    std::memset(out, 0, 4 * sizeof(RmjWinResult));
    for (auto& kv : g->win_results) {
        const WinResult& w = kv.second;
        RmjWinResult& r = out[kv.first & 3];
        r.is_win = w.is_win; r.yakuman = w.yakuman; r.has_win_shape = w.has_win_shape;
        r.n_yaku = (uint8_t)std::min<size_t>(w.yaku.size(), 20);
        for (int i = 0; i < r.n_yaku; i++) r.yaku[i] = (uint8_t)w.yaku[i];
        r.han = w.han; r.fu = w.fu; r.ron_agari = w.ron_agari; r.tsumo_agari_oya = w.tsumo_agari_oya; r.tsumo_agari_ko = w.tsumo_agari_ko;
        r.pao_payer = (int8_t)w.pao_payer;
        mask |= 1 << (kv.first & 3);
    }
    return mask;
}

// ---------------------------------------------------------------- CPU baseline (bench.py cpu_baseline leg)
// Runs `n_games` independent random-agent games (sharded over `threads`) for at least `min_steps_per_game`
// env.step calls each with auto-reset, returns total env steps; seconds via *secs.
uint64_t orc_bench_rollout(int game_mode, uint32_t rule_bits, int skip_log, uint32_t n_games, uint64_t base_seed,
                           uint64_t policy_seed, uint32_t steps_per_game, int threads, double* secs) {
    std::vector<uint64_t> counts(threads, 0);
    auto t0 = std::chrono::steady_clock::now();
    auto work = [&](int tid) {
        uint64_t total = 0;
        for (uint32_t gi = tid; gi < n_games; gi += threads) {
            GameState g((uint8_t)game_mode, skip_log != 0, splitmix64(base_seed + gi), 0, GameRule::from_bits(rule_bits));  // shard.game_seed
            g.env_reset(-1, nullptr, -1, nullptr, -1, -1);
            for (uint32_t s = 0; s < steps_per_game; s++) {
                if (g.is_done) {
                    g.env_reset(-1, nullptr, -1, nullptr, -1, -1);
                    continue;
                }
                std::map<uint8_t, Action> acts;
                for (uint8_t p : g.active_players) {
                    // faithful to the reference loop: legal actions are generated for the observation
                    // (state/mod.rs:205) and again inside step() for validation (state/mod.rs:343)
                    auto l = g._get_legal_actions_internal(p);
                    if (l.empty()) continue;
                    (void)g.observation_waits(p);
                    uint64_t c = policy_choice(policy_seed, gi, g.step_count, p, (uint32_t)l.size());
                    acts[p] = l[c];
                }
                g.step(acts);
                total++;
            }
        }
        counts[tid] = total;
    };
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(work, t);
    for (auto& t : th) t.join();
    auto t1 = std::chrono::steady_clock::now();
    *secs = std::chrono::duration<double>(t1 - t0).count();
    uint64_t sum = 0;
    for (auto c : counts) sum += c;
    return sum;
}

}  // extern "C"
