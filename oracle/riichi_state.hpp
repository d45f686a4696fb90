// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement (C++17) of the riichienv-core 4-player game state machine,
// legal-action generator and MJAI event emission.  Follows, statement by
// statement (quirks included):
//   riichienv-core/src/state/mod.rs            (GameState::step & friends)
//   riichienv-core/src/state/legal_actions.rs  (_get_legal_actions_internal,
//                                               _get_claim_actions_for_player)
//   riichienv-core/src/state/wall.rs, player.rs, game_mode.rs
//   riichienv-core/src/action.rs               (Action, encode)
//   riichienv-python/src/env.rs:799-872        (reset/step binding semantics)
//
// seed -> wall (documented in DESIGN.md §6).  The reference shuffles with rand 0.10
// StdRng (ChaCha12) + SliceRandom::shuffle; the crates are not in /root/reference and
// no reference test pins seed -> wall.  Two definitions exist here and in the HIP path:
//   * default: the build's own counter-based permutation (build_wall below);
//   * WallState::reference_rng (RMJ_RULE_REFERENCE_RNG): the restatement of the crates'
//     published algorithms in ref_rng.hpp, with salt and wall_digest (state/wall.rs:36-56).
// Parity with the reference itself is defined on identical WALLS (load_wall).
#pragma once
#include <map>
#include <optional>
#include <string>
#include <vector>

#include "ref_rng.hpp"
#include "riichi_core.hpp"

namespace orc {

// action.rs:55-68
enum ActionType : uint8_t {
    AT_DISCARD = 0, AT_CHI = 1, AT_PON = 2, AT_DAIMINKAN = 3, AT_RON = 4, AT_RIICHI = 5, AT_TSUMO = 6,
    AT_PASS = 7, AT_ANKAN = 8, AT_KAKAN = 9, AT_KYUSHU = 10, AT_KITA = 11
};
enum Phase : uint8_t { WAIT_ACT = 0, WAIT_RESPONSE = 1 };

// action.rs:76-105
struct Action {
    ActionType type = AT_PASS;
    int tile = -1;  // -1 = None
    std::vector<uint8_t> consume;
    int actor = -1;
    Action() {}
    Action(ActionType t, int tile_, std::vector<uint8_t> c, int actor_) : type(t), tile(tile_), consume(std::move(c)), actor(actor_) {
        std::sort(consume.begin(), consume.end());
    }
    // action.rs:158-227 (4P) ; returns -1 on error
    int encode() const {
        switch (type) {
            case AT_DISCARD: return tile >= 0 ? tile / 4 : -1;
            case AT_RIICHI: return 37;
            case AT_CHI: {
                if (tile < 0) return -1;
                int target = tile / 4;
                std::vector<int> t34;
                for (uint8_t x : consume) t34.push_back(x / 4);
                t34.push_back(target);
                std::sort(t34.begin(), t34.end());
                t34.erase(std::unique(t34.begin(), t34.end()), t34.end());
                if (t34.size() != 3) return -1;
                if (target == t34[0]) return 38;
                if (target == t34[1]) return 39;
                return 40;
            }
            case AT_PON: return 41;
            case AT_DAIMINKAN: return tile >= 0 ? 42 + tile / 4 : -1;
            case AT_ANKAN:
            case AT_KAKAN: return consume.empty() ? -1 : 42 + consume[0] / 4;
            case AT_RON:
            case AT_TSUMO: return 79;
            case AT_KYUSHU: return 80;
            case AT_PASS: return 81;
            default: return -1;
        }
    }
    // action.rs:262-346 (3P compact action space, 60 ids); -1 on error
    int encode_3p() const {
        static const int compact[34] = {0, -1, -1, -1, -1, -1, -1, -1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18,
                                        19, 20, 21, 22, 23, 24, 25, 26};
        switch (type) {
            case AT_DISCARD: return tile >= 0 ? compact[tile / 4] : -1;
            case AT_RIICHI: return 27;
            case AT_CHI: return -1;
            case AT_PON: return 28;
            case AT_DAIMINKAN: return (tile >= 0 && compact[tile / 4] >= 0) ? 29 + compact[tile / 4] : -1;
            case AT_ANKAN:
            case AT_KAKAN: return (!consume.empty() && compact[consume[0] / 4] >= 0) ? 29 + compact[consume[0] / 4] : -1;
            case AT_RON:
            case AT_TSUMO: return 56;
            case AT_KYUSHU: return 57;
            case AT_PASS: return 58;
            case AT_KITA: return 59;
            default: return -1;
        }
    }
};

// rule.rs:10-57
struct GameRule {
    bool allows_ron_on_ankan_for_kokushi_musou = false;
    bool is_kokushi_musou_13machi_double = false;
    bool is_suuankou_tanki_double = false;
    bool is_junsei_chuurenpoutou_double = false;
    bool is_daisuushii_double = false;
    bool yakuman_pao_is_liability_only = false;
    bool sanchaho_is_draw = true;
    bool kuikae_forbidden = true;
    static GameRule tenhou() { return GameRule(); }
    static GameRule mjsoul() {
        GameRule r;
        r.allows_ron_on_ankan_for_kokushi_musou = true;
        r.is_kokushi_musou_13machi_double = true;
        r.is_suuankou_tanki_double = true;
        r.is_junsei_chuurenpoutou_double = true;
        r.is_daisuushii_double = true;
        r.yakuman_pao_is_liability_only = true;
        r.sanchaho_is_draw = false;
        r.kuikae_forbidden = true;
        return r;
    }
    // bit layout shared with include/riichi_mi355x.h (RMJ_RULE_*)
    static GameRule from_bits(uint32_t b) {
        GameRule r;
        r.allows_ron_on_ankan_for_kokushi_musou = b & 1;
        r.is_kokushi_musou_13machi_double = b & 2;
        r.is_suuankou_tanki_double = b & 4;
        r.is_junsei_chuurenpoutou_double = b & 8;
        r.is_daisuushii_double = b & 16;
        r.yakuman_pao_is_liability_only = b & 32;
        r.sanchaho_is_draw = b & 64;
        r.kuikae_forbidden = b & 128;
        return r;
    }
};

inline uint64_t splitmix64(uint64_t x) {  // state/wall.rs:83-88
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// The build's own seed->wall definition (replaces rand::StdRng + shuffle, see header):
//   hand_seed = splitmix64(episode_seed + hand_index)          (state/wall.rs:40, kept)
//   key_i     = splitmix64(hand_seed + i * 0x9E3779B97F4A7C15)  for tile slot i
//   w         = tile ids sorted ascending by (key, id)
// `ids` is the tile universe (0..135 in 4P; 108 ids without 2m-8m in 3P).
inline std::vector<uint8_t> build_wall(uint64_t episode_seed, uint64_t hand_index, const std::vector<uint8_t>& ids) {
    uint64_t hs = splitmix64(episode_seed + hand_index);
    size_t n = ids.size();
    std::vector<std::pair<uint64_t, uint8_t>> k(n);
    for (size_t i = 0; i < n; i++) k[i] = {splitmix64(hs + (uint64_t)i * 0x9E3779B97F4A7C15ull), ids[i]};
    std::sort(k.begin(), k.end());
    std::vector<uint8_t> w(n);
    for (size_t i = 0; i < n; i++) w[i] = k[i].second;
    return w;
}

// state/wall.rs:8-19
struct WallState {
    std::vector<uint8_t> tiles;
    std::vector<uint8_t> dora_indicators;
    uint8_t rinshan_draw_count = 0, pending_kan_dora_count = 0, drawable_count = 0;
    std::optional<uint64_t> seed;
    uint64_t hand_index = 0;
    bool sanma = false;  // state_3p/wall.rs: 108 tiles, dora/ura pre-extracted
    bool reference_rng = false;      // RMJ_RULE_REFERENCE_RNG: StdRng + shuffle + salt + digest as the reference (ref_rng.hpp)
    std::string wall_digest, salt;   // state/wall.rs:15-16 (empty without reference_rng and after load_wall-less event replays)
    uint8_t dora_tiles[5] = {0, 0, 0, 0, 0}, ura_tiles[5] = {0, 0, 0, 0, 0};

    void finish_load() {
        dora_indicators.clear();
        if (sanma) {
            // state_3p/wall.rs:104-109 (shuffle) and :137-147 (load_wall: t[99-2i] / t[98-2i] before reversal,
            // i.e. tiles[8+2i] / tiles[9+2i] after it; extracted only for a 108-tile wall)
            if (tiles.size() == 108)
                for (int i = 0; i < 5; i++) {
                    dora_tiles[i] = tiles[8 + 2 * i];
                    ura_tiles[i] = tiles[9 + 2 * i];
                }
            dora_indicators.push_back(dora_tiles[0]);
        } else if (tiles.size() > 5)
            dora_indicators.push_back(tiles[4]);
        rinshan_draw_count = 0;
        pending_kan_dora_count = 0;
        drawable_count = 0;
    }
    // state/wall.rs:36-67, state_3p/wall.rs:75-110
    void shuffle() {
        std::vector<uint8_t> ids;
        for (int i = 0; i < 136; i++)
            if (!(sanma && i / 4 >= 1 && i / 4 <= 7)) ids.push_back((uint8_t)i);  // types.rs:378-382
        uint64_t s = seed ? *seed : 0;
        std::vector<uint8_t> w;
        if (reference_rng) {
            refrng::RefWall r = refrng::reference_wall(splitmix64(s + hand_index), ids);  // wall.rs:40-42, 45-55
            w = r.w;
            salt = r.salt;
            wall_digest = r.digest;
        } else
            w = build_wall(s, hand_index, ids);  // the build's own definition: no salt, no digest
        hand_index += 1;
        std::reverse(w.begin(), w.end());
        tiles = w;
        finish_load();
    }
    // state/wall.rs:69-80
    void load_wall(std::vector<uint8_t> t) {
        std::reverse(t.begin(), t.end());
        tiles = t;
        finish_load();
    }
};

// state/player.rs:6-39
struct PlayerState {
    std::vector<uint8_t> hand;
    std::vector<Meld> melds;
    std::vector<uint8_t> discards;
    std::vector<bool> discard_from_hand, discard_is_riichi;
    int riichi_declaration_index = -1;
    int32_t score = 25000, score_delta = 0;
    bool riichi_declared = false, riichi_stage = false, double_riichi_declared = false;
    bool missed_agari_riichi = false, missed_agari_doujun = false, nagashi_eligible = true, ippatsu_cycle = false;
    std::map<uint8_t, uint8_t> pao;  // yaku id -> liable seat
    std::vector<uint8_t> forbidden_discards;
    std::vector<uint8_t> kita_tiles;  // state_3p/player.rs:38
    // state/player.rs:66-86
    void reset_round() {
        hand.clear();
        melds.clear();
        discards.clear();
        discard_from_hand.clear();
        discard_is_riichi.clear();
        riichi_declaration_index = -1;
        riichi_declared = riichi_stage = double_riichi_declared = false;
        missed_agari_riichi = missed_agari_doujun = false;
        nagashi_eligible = true;
        ippatsu_cycle = false;
        forbidden_discards.clear();
        kita_tiles.clear();
        score_delta = 0;
        pao.clear();
    }
};

template <typename T>
inline int vec_position(const std::vector<T>& v, T x) {
    for (size_t i = 0; i < v.size(); i++)
        if (v[i] == x) return (int)i;
    return -1;
}

inline std::string json_str_array(const std::vector<std::string>& v) {
    std::string s = "[";
    for (size_t i = 0; i < v.size(); i++) {
        if (i) s += ",";
        s += "\"" + v[i] + "\"";
    }
    return s + "]";
}
template <typename It>
inline std::string json_int_array(It b, It e) {
    std::string s = "[";
    bool first = true;
    for (It i = b; i != e; ++i) {
        if (!first) s += ",";
        first = false;
        s += std::to_string(*i);
    }
    return s + "]";
}

struct GameState {
    int NP = 4;          // 4 (state/) or 3 (state_3p/)
    bool sanma = false;  // game_mode >= 3 (game_variant.rs:12-37)
    int32_t start_score() const { return sanma ? 35000 : 25000; }  // state_3p/game_mode.rs:31-33
    WallState wall;
    PlayerState players[4];
    uint8_t current_player = 0;
    uint32_t turn_count = 0;
    bool is_done = false, needs_tsumo = false;
    uint32_t riichi_sticks = 0;
    Phase phase = WAIT_ACT;
    std::vector<uint8_t> active_players;
    std::optional<std::pair<uint8_t, uint8_t>> last_discard;  // (pid, tile)
    std::map<uint8_t, std::vector<Action>> current_claims;
    std::optional<std::pair<uint8_t, Action>> pending_kan;
    uint8_t oya = 0, honba = 0, kyoku_idx = 0, round_wind = 0;
    bool is_rinshan_flag = false, is_first_turn = true;
    bool replay_after_kan = false;   // is_after_kan of the log walker (apply_log_action, state/event_handler.rs:415-430, :565, :658)
    int riichi_pending_acceptance = -1;
    int drawn_tile = -1;
    std::map<uint8_t, WinResult> win_results;
    std::vector<std::string> mjai_log;
    std::vector<std::string> mjai_log_per_player[4];
    size_t player_event_counts[4] = {0, 0, 0, 0};
    uint8_t game_mode = 0;
    bool skip_mjai_logging = false;
    GameRule rule;
    std::optional<std::string> last_error;
    int riichi_sutehais[4] = {-1, -1, -1, -1};
    int last_tedashis[4] = {-1, -1, -1, -1};
    uint64_t step_count = 0;

    // state/mod.rs:98-167
    GameState(uint8_t game_mode_, bool skip_log, std::optional<uint64_t> seed, uint8_t round_wind_, GameRule rule_,
              bool reference_rng = false)
        : game_mode(game_mode_), skip_mjai_logging(skip_log), rule(rule_) {
        sanma = game_mode_ >= 3;
        NP = sanma ? 3 : 4;
        wall.sanma = sanma;
        wall.reference_rng = reference_rng;
        wall.seed = seed;
        round_wind = round_wind_;
        for (auto& p : players) p.score = start_score();
        if (!skip_mjai_logging) push_event("{\"type\":\"start_game\"}", "start_game", -1, nullptr);
        _initialize_round(0, round_wind, 0, 0, nullptr, nullptr);
    }

    // state/mod.rs:171-187
    void reset() {
        mjai_log.clear();
        for (int i = 0; i < NP; i++) {
            mjai_log_per_player[i].clear();
            player_event_counts[i] = 0;
        }
        if (!skip_mjai_logging) push_event("{\"type\":\"start_game\"}", "start_game", -1, nullptr);
    }

    // riichienv-python/src/env.rs:799-851 (seed handling = quirk Q1: wall.seed untouched)
    void env_reset(int oya_, const std::vector<uint8_t>* wall_, int round_wind_, const std::vector<int32_t>* scores,
                   int honba_, int kyotaku_) {
        reset();
        std::vector<int32_t> def(NP, start_score());
        _initialize_round((uint8_t)(oya_ < 0 ? 0 : oya_), (uint8_t)(round_wind_ < 0 ? 0 : round_wind_),
                          (uint8_t)(honba_ < 0 ? 0 : honba_), (uint32_t)(kyotaku_ < 0 ? 0 : kyotaku_), wall_,
                          scores ? scores : &def);
    }

    // ---- event emission: state/mod.rs:2094-2148 (alphabetical keys = serde_json BTreeMap) ----
    // masked_tehais: for start_kyoku only, the 4 per-seat strings.
    void push_event(const std::string& json, const char* type, int actor, const std::string* per_seat) {
        if (skip_mjai_logging) return;
        mjai_log.push_back(json);
        for (int pid = 0; pid < NP; pid++) {
            if (per_seat)
                mjai_log_per_player[pid].push_back(per_seat[pid]);
            else
                mjai_log_per_player[pid].push_back(json);
        }
        (void)type;
        (void)actor;
    }
    void ev_simple_actor(const char* type, int actor) {
        if (skip_mjai_logging) return;
        push_event("{\"actor\":" + std::to_string(actor) + ",\"type\":\"" + type + "\"}", type, actor, nullptr);
    }
    void ev_type_only(const char* type) {
        if (skip_mjai_logging) return;
        push_event(std::string("{\"type\":\"") + type + "\"}", type, -1, nullptr);
    }
    void ev_tsumo(int actor, uint8_t t) {
        if (skip_mjai_logging) return;
        std::string full = "{\"actor\":" + std::to_string(actor) + ",\"pai\":\"" + tid_to_mjai(t) + "\",\"type\":\"tsumo\"}";
        std::string masked = "{\"actor\":" + std::to_string(actor) + ",\"pai\":\"?\",\"type\":\"tsumo\"}";
        std::string per[4];
        for (int i = 0; i < NP; i++) per[i] = (i == actor) ? full : masked;
        push_event(full, "tsumo", actor, per);
    }
    void ev_dora(uint8_t marker) {
        if (skip_mjai_logging) return;
        push_event("{\"dora_marker\":\"" + tid_to_mjai(marker) + "\",\"type\":\"dora\"}", "dora", -1, nullptr);
    }
    static std::string tiles_json(const std::vector<uint8_t>& ts) {
        std::vector<std::string> v;
        for (uint8_t t : ts) v.push_back(tid_to_mjai(t));
        return json_str_array(v);
    }

    // state/mod.rs:189-263 : waits for the observation (13-tile hands only)
    std::vector<uint8_t> observation_waits(int pid) const {
        HandEvaluator he(players[pid].hand, players[pid].melds, sanma);
        return he.get_waits_u8();
    }

    // ---------------------------------------------------------------- legal_actions.rs:11-252
    std::vector<Action> _get_legal_actions_internal(uint8_t pid) const {
        std::vector<Action> legals;
        const PlayerState& P = players[pid];
        if (is_done) return legals;
        if (phase == WAIT_ACT) {
            if (pid != current_player) return legals;
            // 1. Tsumo
            if (drawn_tile >= 0 && !P.riichi_stage) {
                uint8_t tile = (uint8_t)drawn_tile;
                Conditions c;
                c.tsumo = true;
                c.riichi = P.riichi_declared;
                c.double_riichi = P.double_riichi_declared;
                c.ippatsu = P.ippatsu_cycle;
                c.player_wind = (uint8_t)((pid + NP - oya) % NP);
                c.round_wind = round_wind % 4;
                c.is_sanma = sanma;
                c.num_players = (uint8_t)NP;
                c.haitei = wall.drawable_count == 0 && !is_rinshan_flag;
                c.rinshan = is_rinshan_flag;
                c.tsumo_first_turn = is_first_turn && P.discards.empty();
                c.riichi_sticks = riichi_sticks;
                c.honba = honba;
                std::vector<uint8_t> hand = P.hand;
                for (int i = (int)hand.size() - 1; i >= 0; i--)
                    if (hand[i] == tile) {
                        hand.erase(hand.begin() + i);
                        break;
                    }
                HandEvaluator he(hand, P.melds, sanma);
                WinResult r = he.calc(tile, wall.dora_indicators, {}, c);
                if (r.is_win && (r.yakuman || r.han >= 1)) legals.push_back(Action(AT_TSUMO, tile, {}, pid));
            }
            // 2. Discard / Riichi
            if (P.riichi_declared) {
                if (drawn_tile >= 0) legals.push_back(Action(AT_DISCARD, drawn_tile, {}, pid));
            } else if (P.riichi_stage) {
                for (uint8_t t : P.hand) {
                    bool forb = false;
                    for (uint8_t f : P.forbidden_discards)
                        if (f / 4 == t / 4) forb = true;
                    if (forb) continue;
                    std::vector<uint8_t> th = P.hand;
                    int idx = vec_position(th, t);
                    if (idx >= 0) th.erase(th.begin() + idx);
                    HandEvaluator he(th, P.melds, sanma);
                    if (he.is_tenpai()) legals.push_back(Action(AT_DISCARD, t, {}, pid));
                }
            } else {
                for (uint8_t t : P.hand) {
                    bool forb = false;
                    for (uint8_t f : P.forbidden_discards)
                        if (f / 4 == t / 4) forb = true;
                    if (!forb) legals.push_back(Action(AT_DISCARD, t, {}, pid));
                }
                bool all_closed = true;
                for (auto& m : P.melds)
                    if (m.opened) all_closed = false;
                // quirk Q8: 4P needs drawable_count >= 4, 3P only > 0 (state_3p/legal_actions.rs:116)
                if (P.score >= 1000 && (sanma ? wall.drawable_count > 0 : wall.drawable_count >= 4) && all_closed) {
                    bool can = false;
                    for (size_t skip = 0; skip < P.hand.size(); skip++) {
                        std::vector<uint8_t> th = P.hand;
                        th.erase(th.begin() + skip);
                        HandEvaluator he(th, P.melds, sanma);
                        if (he.is_tenpai()) {
                            can = true;
                            break;
                        }
                    }
                    if (can) legals.push_back(Action(AT_RIICHI, -1, {}, pid));
                }
            }
            // 3. Kan
            if (wall.drawable_count > 0 && drawn_tile >= 0) {
                int counts[34] = {0};
                for (uint8_t t : P.hand) counts[t / 4]++;
                if (!P.riichi_declared && !P.riichi_stage) {
                    for (int tv = 0; tv < 34; tv++)
                        if (counts[tv] == 4) {
                            uint8_t lo = (uint8_t)(tv * 4);
                            legals.push_back(Action(AT_ANKAN, lo, {lo, (uint8_t)(lo + 1), (uint8_t)(lo + 2), (uint8_t)(lo + 3)}, pid));
                        }
                    for (auto& m : P.melds)
                        if (m.meld_type == MT_PON) {
                            uint8_t target = m.tiles[0] / 4;
                            for (uint8_t t : P.hand)
                                if (t / 4 == target) legals.push_back(Action(AT_KAKAN, t, m.tiles, pid));
                        }
                } else if (P.riichi_declared) {
                    uint8_t t = (uint8_t)drawn_tile;
                    uint8_t t34 = t / 4;
                    if (counts[t34] == 4) {
                        std::vector<uint8_t> pre = P.hand;
                        int pos = vec_position(pre, t);
                        if (pos >= 0) pre.erase(pre.begin() + pos);
                        HandEvaluator cpre(pre, P.melds, sanma);
                        std::vector<uint8_t> wpre = cpre.get_waits_u8();
                        std::sort(wpre.begin(), wpre.end());
                        std::vector<uint8_t> post;
                        for (uint8_t x : P.hand)
                            if (x / 4 != t34) post.push_back(x);
                        std::vector<Meld> mpost = P.melds;
                        uint8_t lo = t34 * 4;
                        Meld am;
                        am.meld_type = MT_ANKAN;
                        am.tiles = {lo, (uint8_t)(lo + 1), (uint8_t)(lo + 2), (uint8_t)(lo + 3)};
                        am.opened = false;
                        am.from_who = -1;
                        mpost.push_back(am);
                        HandEvaluator cpost(post, mpost, sanma);
                        std::vector<uint8_t> wpost = cpost.get_waits_u8();
                        std::sort(wpost.begin(), wpost.end());
                        if (wpre == wpost && !wpre.empty())
                            legals.push_back(Action(AT_ANKAN, lo, {lo, (uint8_t)(lo + 1), (uint8_t)(lo + 2), (uint8_t)(lo + 3)}, pid));
                    }
                }
            }
            // 4. Kyushu kyuhai
            bool no_calls = true;
            for (auto& p : players)
                if (!p.melds.empty()) no_calls = false;
            if (is_first_turn && no_calls && !P.riichi_stage) {
                bool seen[34] = {false};
                int distinct = 0;
                for (uint8_t t : P.hand)
                    if (is_terminal_tile(t) && !seen[t / 4]) {
                        seen[t / 4] = true;
                        distinct++;
                    }
                if (distinct >= 9) legals.push_back(Action(AT_KYUSHU, -1, {}, pid));
            }
            // 5. Kita (state_3p/legal_actions.rs:241-243, sanma.rs:146-169)
            if (sanma && drawn_tile >= 0 && wall.drawable_count > 0)
                for (uint8_t t : P.hand)
                    if (t / 4 == 30) legals.push_back(Action(AT_KITA, t, {}, pid));
        } else {
            auto it = current_claims.find(pid);
            if (it != current_claims.end())
                for (auto& a : it->second) legals.push_back(a);
            legals.push_back(Action(AT_PASS, -1, {}, pid));
        }
        return legals;
    }

    // ---------------------------------------------------------------- legal_actions.rs:254-508
    std::pair<std::vector<Action>, bool> _get_claim_actions_for_player(uint8_t i, uint8_t pid, uint8_t tile) const {
        std::vector<Action> legals;
        bool missed_agari = false;
        const PlayerState& P = players[i];
        const std::vector<uint8_t>& hand = P.hand;
        uint8_t tile_class = tile / 4;
        bool in_discards = false;
        for (uint8_t d : P.discards)
            if (d / 4 == tile_class) in_discards = true;
        bool in_missed = P.missed_agari_doujun || (P.riichi_declared && P.missed_agari_riichi);
        if (!in_discards && !in_missed) {
            HandEvaluator he(hand, P.melds, sanma);
            Conditions c;
            c.tsumo = false;
            c.riichi = P.riichi_declared;
            c.double_riichi = P.double_riichi_declared;
            c.ippatsu = P.ippatsu_cycle;
            c.player_wind = (uint8_t)((i + NP - oya) % NP);
            c.round_wind = round_wind % 4;
            c.is_sanma = sanma;
            c.num_players = (uint8_t)NP;
            c.houtei = wall.drawable_count == 0 && !is_rinshan_flag;
            c.riichi_sticks = riichi_sticks;
            c.honba = honba;
            bool furiten = false;
            for (uint8_t w : he.get_waits_u8()) {
                for (uint8_t d : P.discards)
                    if (d / 4 == w) furiten = true;
                if (furiten) break;
            }
            if (P.missed_agari_riichi || P.missed_agari_doujun) furiten = true;
            if (!furiten) {
                WinResult r = he.calc(tile, wall.dora_indicators, {}, c);
                if (r.is_win)
                    legals.push_back(Action(AT_RON, tile, {}, i));
                else if (r.has_win_shape)
                    missed_agari = true;
            }
        }
        // 2. Pon / Kan
        if (!P.riichi_declared && wall.drawable_count > 0) {
            int count = 0;
            for (uint8_t t : hand)
                if (t / 4 == tile / 4) count++;
            if (count >= 2 && hand.size() >= 3) {
                auto check_pon_kuikae = [&](const std::vector<uint8_t>& consumes) {
                    std::vector<uint8_t> forb;
                    if (rule.kuikae_forbidden) forb.push_back(tile / 4);
                    std::vector<bool> used(consumes.size(), false);
                    for (uint8_t t : hand) {
                        bool consumed = false;
                        for (size_t k = 0; k < consumes.size(); k++)
                            if (!used[k] && consumes[k] == t) {
                                used[k] = true;
                                consumed = true;
                                break;
                            }
                        if (consumed) continue;
                        bool f = false;
                        for (uint8_t x : forb)
                            if (x == t / 4) f = true;
                        if (!f) return true;
                    }
                    return false;
                };
                std::vector<uint8_t> matching;
                for (uint8_t t : hand)
                    if (t / 4 == tile / 4) matching.push_back(t);
                for (size_t a = 0; a < matching.size(); a++)
                    for (size_t b = a + 1; b < matching.size(); b++) {
                        std::vector<uint8_t> cons = {matching[a], matching[b]};
                        if (check_pon_kuikae(cons)) legals.push_back(Action(AT_PON, tile, cons, i));
                    }
            }
            if (count >= 3) {
                std::vector<uint8_t> cons;
                for (uint8_t t : hand)
                    if (t / 4 == tile / 4 && cons.size() < 3) cons.push_back(t);
                legals.push_back(Action(AT_DAIMINKAN, tile, cons, i));
            }
        }
        // 3. Chi
        bool is_shimocha = !sanma && i == (pid + 1) % 4;  // no Chi in 3P (state_3p/legal_actions.rs:386)
        if (!P.riichi_declared && wall.drawable_count > 0 && is_shimocha && hand.size() >= 3) {
            int t_val = tile / 4;
            if (t_val < 27) {
                auto check_chi_kuikae = [&](uint8_t c1, uint8_t c2) {
                    std::vector<int> forb;
                    if (rule.kuikae_forbidden) {
                        forb.push_back(t_val);
                        int a = c1 / 4, b = c2 / 4;
                        if (a > b) std::swap(a, b);
                        if (a == t_val + 1 && b == t_val + 2) {
                            if (t_val % 9 <= 5) forb.push_back(t_val + 3);
                        } else if (t_val >= 2 && b == t_val - 1 && a == t_val - 2 && t_val % 9 >= 3) {
                            forb.push_back(t_val - 3);
                        }
                    }
                    bool u1 = false, u2 = false;
                    for (uint8_t t : hand) {
                        if (!u1 && t == c1) {
                            u1 = true;
                            continue;
                        }
                        if (!u2 && t == c2) {
                            u2 = true;
                            continue;
                        }
                        bool f = false;
                        for (int x : forb)
                            if (x == t / 4) f = true;
                        if (!f) return true;
                    }
                    return false;
                };
                auto opts = [&](int ty) {
                    std::vector<uint8_t> v;
                    for (uint8_t t : hand)
                        if (t / 4 == ty) v.push_back(t);
                    return v;
                };
                auto emit = [&](int ta, int tb) {
                    for (uint8_t c1 : opts(ta))
                        for (uint8_t c2 : opts(tb))
                            if (check_chi_kuikae(c1, c2)) legals.push_back(Action(AT_CHI, tile, {c1, c2}, i));
                };
                if (t_val % 9 >= 2) emit(t_val - 2, t_val - 1);
                if (t_val % 9 >= 1 && t_val % 9 <= 7) emit(t_val - 1, t_val + 1);
                if (t_val % 9 <= 6) emit(t_val + 1, t_val + 2);
            }
        }
        return {legals, missed_agari};
    }

    // ---------------------------------------------------------------- state/mod.rs:330-1315
    void step(const std::map<uint8_t, Action>& actions) {
        if (is_done) return;
        step_count++;
        // Validation (mod.rs:339-402)
        for (int pid = 0; pid < NP; pid++) {
            auto it = actions.find((uint8_t)pid);
            if (it == actions.end()) continue;
            const Action& act = it->second;
            std::vector<Action> legals = _get_legal_actions_internal((uint8_t)pid);
            bool valid = false;
            for (const Action& l : legals) {
                if (l.type != act.type) continue;
                bool tiles_match = l.tile == act.tile;
                bool consumes_match = l.consume == act.consume;
                bool ok = false;
                if (tiles_match) {
                    if (consumes_match) ok = true;
                    if (act.consume.empty() && l.type == AT_KAKAN) ok = true;
                    if (act.consume.empty() &&
                        (l.type == AT_DISCARD || l.type == AT_RIICHI || l.type == AT_TSUMO || l.type == AT_RON || l.type == AT_PASS))
                        ok = true;
                }
                if (!ok && consumes_match && (l.type == AT_ANKAN || l.type == AT_KAKAN)) ok = true;
                if (!ok && act.tile < 0)
                    ok = (l.type == AT_TSUMO || l.type == AT_RON || l.type == AT_RIICHI || l.type == AT_KYUSHU || l.type == AT_KITA);
                if (ok) {
                    valid = true;
                    break;
                }
            }
            if (!valid) {
                std::string reason = "Error: Illegal Action by Player " + std::to_string(pid);
                last_error = reason;
                _trigger_ryukyoku(reason);
                return;
            }
        }

        if (phase == WAIT_ACT) {
            uint8_t pid = current_player;
            auto it = actions.find(pid);
            if (it == actions.end()) return;
            const Action act = it->second;
            PlayerState& P = players[pid];
            switch (act.type) {
                case AT_DISCARD: {
                    if (act.tile < 0) break;
                    uint8_t tile = (uint8_t)act.tile;
                    bool tsumogiri = false, valid = false;
                    if (drawn_tile >= 0 && drawn_tile == tile) {
                        tsumogiri = true;
                        valid = true;
                    }
                    int idx = vec_position(P.hand, tile);
                    if (idx >= 0) {
                        P.hand.erase(P.hand.begin() + idx);
                        std::sort(P.hand.begin(), P.hand.end());
                        valid = true;
                        if (drawn_tile >= 0 && drawn_tile == tile) tsumogiri = true;
                    }
                    if (valid) _resolve_discard(pid, tile, tsumogiri);
                    break;
                }
                case AT_KYUSHU: _trigger_ryukyoku("kyushu_kyuhai"); break;
                case AT_RIICHI: {
                    if (P.score >= 1000 && (sanma ? wall.drawable_count > 0 : wall.drawable_count >= 4) && !P.riichi_declared &&
                        !P.riichi_stage) {
                        P.riichi_stage = true;
                        ev_simple_actor("reach", pid);
                        if (act.tile >= 0) {
                            uint8_t t = (uint8_t)act.tile;
                            bool tsumogiri = drawn_tile >= 0 && drawn_tile == t;
                            riichi_sutehais[pid] = t;
                            if (!tsumogiri) last_tedashis[pid] = t;
                            int idx = vec_position(P.hand, t);
                            if (idx >= 0) {
                                P.hand.erase(P.hand.begin() + idx);
                                std::sort(P.hand.begin(), P.hand.end());
                            }
                            _resolve_discard(pid, t, tsumogiri);
                        }
                    }
                    break;
                }
                case AT_ANKAN: {
                    uint8_t tile = act.tile >= 0 ? (uint8_t)act.tile : (act.consume.empty() ? 0 : act.consume[0]);
                    std::vector<uint8_t> ronners;
                    if (rule.allows_ron_on_ankan_for_kokushi_musou) {
                        for (uint8_t i = 0; i < NP; i++) {
                            if (i == pid) continue;
                            const PlayerState& Q = players[i];
                            bool in_disc = false;
                            for (uint8_t d : Q.discards)
                                if (d / 4 == tile / 4) in_disc = true;
                            if (in_disc) continue;
                            Conditions c;
                            c.tsumo = false;
                            c.riichi = Q.riichi_declared;
                            c.chankan = true;
                            c.player_wind = (uint8_t)((i + NP - oya) % NP);
                            c.round_wind = round_wind % 4;
                            c.is_sanma = sanma;
                            c.num_players = (uint8_t)NP;
                            HandEvaluator he(Q.hand, Q.melds, sanma);
                            WinResult r = he.calc(tile, wall.dora_indicators, {}, c);
                            bool kok = false;
                            for (uint32_t y : r.yaku)
                                if (y == 42 || y == 49) kok = true;
                            if (r.is_win && kok) {
                                ronners.push_back(i);
                                current_claims[i].push_back(Action(AT_RON, tile, {}, i));
                            }
                        }
                    }
                    if (!ronners.empty()) {
                        pending_kan = std::make_pair(pid, act);
                        phase = WAIT_RESPONSE;
                        active_players = ronners;
                        last_discard = std::make_pair(pid, tile);
                    } else {
                        _resolve_kan(pid, act);
                    }
                    break;
                }
                case AT_KAKAN: {
                    uint8_t tile = act.tile >= 0 ? (uint8_t)act.tile : (act.consume.empty() ? 0 : act.consume[0]);
                    int idx = vec_position(P.hand, tile);
                    if (idx >= 0) P.hand.erase(P.hand.begin() + idx);
                    for (auto& m : P.melds)
                        if (m.meld_type == MT_PON && m.tiles[0] / 4 == tile / 4) {
                            m.meld_type = MT_KAKAN;
                            m.tiles.push_back(tile);
                            std::sort(m.tiles.begin(), m.tiles.end());
                            break;
                        }
                    if (!skip_mjai_logging)
                        push_event("{\"actor\":" + std::to_string(pid) + ",\"consumed\":" + tiles_json(act.consume) +
                                       ",\"pai\":\"" + tid_to_mjai(tile) + "\",\"type\":\"kakan\"}",
                                   "kakan", pid, nullptr);
                    while (wall.pending_kan_dora_count > 0) {
                        wall.pending_kan_dora_count--;
                        _reveal_kan_dora();
                    }
                    std::vector<uint8_t> ronners;
                    for (uint8_t i = 0; i < NP; i++) {
                        if (i == pid) continue;
                        const PlayerState& Q = players[i];
                        Conditions c;
                        c.tsumo = false;
                        c.riichi = Q.riichi_declared;
                        c.double_riichi = Q.double_riichi_declared;
                        c.ippatsu = Q.ippatsu_cycle;
                        c.player_wind = (uint8_t)((i + NP - oya) % NP);
                        c.round_wind = round_wind % 4;
                        c.chankan = true;
                        c.riichi_sticks = riichi_sticks;
                        c.honba = honba;
                        c.is_sanma = sanma;
                        c.num_players = (uint8_t)NP;
                        HandEvaluator he(Q.hand, Q.melds, sanma);
                        bool furiten = false;
                        for (uint8_t w : he.get_waits_u8()) {
                            for (uint8_t d : Q.discards)
                                if (d / 4 == w) furiten = true;
                            if (furiten) break;
                        }
                        if (Q.missed_agari_riichi || Q.missed_agari_doujun) furiten = true;
                        WinResult r;
                        if (!furiten) r = he.calc(tile, wall.dora_indicators, {}, c);
                        if (r.is_win && (r.yakuman || r.han >= 1)) {
                            ronners.push_back(i);
                            current_claims[i].push_back(Action(AT_RON, tile, {}, i));
                        }
                    }
                    if (!ronners.empty()) {
                        pending_kan = std::make_pair(pid, act);
                        phase = WAIT_RESPONSE;
                        active_players = ronners;
                        last_discard = std::make_pair(pid, tile);
                    } else {
                        _resolve_kan(pid, act);
                    }
                    break;
                }
                case AT_TSUMO: {
                    Conditions c;
                    c.tsumo = true;
                    c.riichi = P.riichi_declared;
                    c.double_riichi = P.double_riichi_declared;
                    c.ippatsu = P.ippatsu_cycle;
                    c.haitei = wall.drawable_count == 0 && !is_rinshan_flag;
                    c.rinshan = is_rinshan_flag;
                    bool no_melds = true;
                    for (auto& p : players)
                        if (!p.melds.empty()) no_melds = false;
                    c.tsumo_first_turn = is_first_turn && no_melds;
                    c.player_wind = (uint8_t)((pid + NP - oya) % NP);
                    c.round_wind = round_wind % 4;
                    c.riichi_sticks = riichi_sticks;
                    c.honba = honba;
                    c.is_sanma = sanma;
                    c.num_players = (uint8_t)NP;
                    if (sanma) c.kita_count = (uint8_t)P.kita_tiles.size();  // state_3p/mod.rs:635
                    HandEvaluator he(P.hand, P.melds, sanma);
                    uint8_t win_tile = drawn_tile >= 0 ? (uint8_t)drawn_tile : 0;
                    std::vector<uint8_t> ura;
                    if (P.riichi_declared) ura = _get_ura_indicators();
                    WinResult res = he.calc(win_tile, wall.dora_indicators, ura, c);
                    cap_double_yakuman(res, pid == oya, true, c.honba);
                    if (res.is_win) {
                        int32_t deltas[4] = {0, 0, 0, 0};
                        int32_t total_win = 0;
                        int pao_payer = -1;
                        int32_t pao_val = 0, total_val = 0;
                        if (res.yakuman) {
                            for (uint32_t yid : res.yaku) {
                                int32_t val = yakuman_val(yid);
                                total_val += val;
                                auto f = P.pao.find((uint8_t)yid);
                                if (f != P.pao.end()) {
                                    pao_val += val;
                                    pao_payer = f->second;
                                }
                            }
                        }
                        if (pao_val > 0) {
                            // state_3p/mod.rs:713-721: (np-1)*16000 for the dealer, 16000+(np-2)*8000 otherwise
                            int32_t unit = pid == oya ? (NP - 1) * 16000 : 16000 + (NP - 2) * 8000;
                            int32_t honba_total = (int32_t)honba * (NP - 1) * 100;
                            if (pao_payer >= 0) {
                                if (rule.yakuman_pao_is_liability_only) {
                                    int32_t pao_amt = pao_val * unit + honba_total;
                                    int32_t non = total_val - pao_val;
                                    deltas[pao_payer] -= pao_amt;
                                    total_win += pao_amt;
                                    if (non > 0) {
                                        if (pid == oya) {
                                            int32_t share = non * 16000;
                                            for (int i = 0; i < NP; i++)
                                                if (i != pid) {
                                                    deltas[i] -= share;
                                                    total_win += share;
                                                }
                                        } else {
                                            int32_t oya_pay = non * 16000, ko_pay = non * 8000;
                                            for (int i = 0; i < NP; i++)
                                                if (i != pid) {
                                                    int32_t p = (i == oya) ? oya_pay : ko_pay;
                                                    deltas[i] -= p;
                                                    total_win += p;
                                                }
                                        }
                                    }
                                } else {
                                    int32_t full = total_val * unit + honba_total;
                                    deltas[pao_payer] -= full;
                                    total_win += full;
                                }
                            }
                        } else {
                            for (int i = 0; i < NP; i++)
                                if (i != pid) {
                                    int32_t p;
                                    if (pid == oya)
                                        p = (int32_t)res.tsumo_agari_ko;
                                    else
                                        p = (i == oya) ? (int32_t)res.tsumo_agari_oya : (int32_t)res.tsumo_agari_ko;
                                    deltas[i] = -p;
                                    total_win += p;
                                }
                        }
                        total_win += (int32_t)(riichi_sticks * 1000);
                        riichi_sticks = 0;
                        deltas[pid] += total_win;
                        for (int i = 0; i < NP; i++) {
                            players[i].score += deltas[i];
                            players[i].score_delta = deltas[i];
                        }
                        WinResult val = res;
                        for (auto& kv : P.pao) {
                            bool has = false;
                            for (uint32_t y : val.yaku)
                                if (y == kv.first) has = true;
                            if (has) {
                                val.pao_payer = kv.second;
                                break;
                            }
                        }
                        win_results[pid] = val;
                        if (!skip_mjai_logging) {
                            std::vector<std::string> um;
                            if (P.riichi_declared)
                                for (uint8_t t : _get_ura_indicators()) um.push_back(tid_to_mjai(t));
                            push_event("{\"actor\":" + std::to_string(pid) + ",\"deltas\":" + json_int_array(deltas, deltas + NP) +
                                           ",\"target\":" + std::to_string(pid) + ",\"tsumo\":true,\"type\":\"hora\",\"ura_markers\":" +
                                           json_str_array(um) + "}",
                                       "hora", pid, nullptr);
                        }
                        _initialize_next_round(pid == oya, false);
                    } else {
                        current_player = (uint8_t)((current_player + 1) % NP);
                        _deal_next();
                    }
                    break;
                }
                case AT_KITA:
                    if (sanma) handle_kita(pid, act);  // state_3p/mod.rs:833
                    break;
                default: break;
            }
        } else {
            // WaitResponse (mod.rs:900-1314)
            for (auto& kv : current_claims) {
                bool has_ron = false;
                for (auto& a : kv.second)
                    if (a.type == AT_RON) has_ron = true;
                if (has_ron) {
                    auto it = actions.find(kv.first);
                    bool roned = it != actions.end() && it->second.type == AT_RON;
                    if (!roned) {
                        players[kv.first].missed_agari_doujun = true;
                        if (players[kv.first].riichi_declared) players[kv.first].missed_agari_riichi = true;
                    }
                }
            }
            std::vector<uint8_t> ron_claims;
            std::optional<std::pair<uint8_t, Action>> call_claim;
            for (uint8_t pid : active_players) {
                auto it = actions.find(pid);
                if (it == actions.end()) continue;
                const Action& act = it->second;
                if (act.type == AT_RON) {
                    ron_claims.push_back(pid);
                } else if (act.type == AT_PON || act.type == AT_DAIMINKAN || (!sanma && act.type == AT_CHI)) {
                    if (call_claim) {
                        bool old_pon = call_claim->second.type == AT_PON || call_claim->second.type == AT_DAIMINKAN;
                        bool new_pon = act.type == AT_PON || act.type == AT_DAIMINKAN;
                        if (!old_pon && new_pon) call_claim = std::make_pair(pid, act);
                    } else {
                        call_claim = std::make_pair(pid, act);
                    }
                }
            }
            if (!ron_claims.empty()) {
                if (!sanma && ron_claims.size() >= (size_t)(NP - 1) && rule.sanchaho_is_draw) {
                    _trigger_ryukyoku("sanchaho");
                    return;
                }
                uint8_t target_pid = last_discard ? last_discard->first : current_player;
                uint8_t win_tile = last_discard ? last_discard->second : 0;
                std::stable_sort(ron_claims.begin(), ron_claims.end(), [&](uint8_t a, uint8_t b) {
                    return (a + NP - target_pid) % NP < (b + NP - target_pid) % NP;
                });
                int32_t total_deltas[4] = {0, 0, 0, 0};
                bool oya_won = false, deposit_taken = false, honba_taken = false;
                for (uint8_t w : ron_claims) {
                    PlayerState& W = players[w];
                    // 3P: a pending kita is a chankan-style claim but awards no chankan yaku (state_3p/mod.rs:896-902)
                    bool is_chankan = pending_kan.has_value() && pending_kan->second.type != AT_KITA;
                    uint32_t ron_honba = 0;
                    if (!honba_taken) {
                        honba_taken = true;
                        ron_honba = honba;
                    }
                    Conditions c;
                    c.tsumo = false;
                    c.riichi = W.riichi_declared;
                    c.double_riichi = W.double_riichi_declared;
                    c.ippatsu = W.ippatsu_cycle;
                    c.houtei = wall.drawable_count == 0 && !is_rinshan_flag;
                    c.chankan = is_chankan;
                    c.player_wind = (uint8_t)((w + NP - oya) % NP);
                    c.round_wind = round_wind % 4;
                    c.riichi_sticks = riichi_sticks;
                    c.honba = ron_honba;
                    c.is_sanma = sanma;
                    c.num_players = (uint8_t)NP;
                    if (sanma) c.kita_count = (uint8_t)W.kita_tiles.size();
                    HandEvaluator he(W.hand, W.melds, sanma);
                    std::vector<uint8_t> ura;
                    if (W.riichi_declared) ura = _get_ura_indicators();
                    WinResult res = he.calc(win_tile, wall.dora_indicators, ura, c);
                    cap_double_yakuman(res, w == oya, false, ron_honba);
                    if (res.is_win) {
                        int32_t score = (int32_t)res.ron_agari;
                        uint8_t pao_payer = target_pid;
                        int32_t pao_amt = 0;
                        if (res.yakuman) {
                            bool has_pao = false;
                            int32_t total_val = 0, pao_val = 0;
                            for (uint32_t yid : res.yaku) {
                                int32_t val = yakuman_val(yid);
                                total_val += val;
                                auto f = W.pao.find((uint8_t)yid);
                                if (f != W.pao.end()) {
                                    has_pao = true;
                                    pao_payer = f->second;
                                    pao_val += val;
                                }
                            }
                            if (has_pao) {
                                int32_t unit = (w == oya) ? 48000 : 32000;
                                int32_t honba_ron = (int32_t)ron_honba * (NP - 1) * 100;
                                int32_t split_base = rule.yakuman_pao_is_liability_only ? pao_val * unit : total_val * unit;
                                pao_amt = split_base / 2 + honba_ron;
                            }
                        }
                        int32_t this_d[4] = {0, 0, 0, 0};
                        this_d[w] += score;
                        this_d[pao_payer] -= pao_amt;
                        this_d[target_pid] -= score - pao_amt;
                        total_deltas[w] += score;
                        total_deltas[pao_payer] -= pao_amt;
                        total_deltas[target_pid] -= score - pao_amt;
                        if (!deposit_taken) {
                            int32_t sp = (int32_t)(riichi_sticks * 1000);
                            total_deltas[w] += sp;
                            this_d[w] += sp;
                            riichi_sticks = 0;
                            deposit_taken = true;
                        }
                        WinResult val = res;
                        for (auto& kv : W.pao) {
                            bool has = false;
                            for (uint32_t y : val.yaku)
                                if (y == kv.first) has = true;
                            if (has) {
                                val.pao_payer = kv.second;
                                break;
                            }
                        }
                        win_results[w] = val;
                        if (w == oya) oya_won = true;
                        if (!skip_mjai_logging) {
                            std::vector<std::string> um;
                            if (W.riichi_declared)
                                for (uint8_t t : _get_ura_indicators()) um.push_back(tid_to_mjai(t));
                            push_event("{\"actor\":" + std::to_string(w) + ",\"deltas\":" + json_int_array(this_d, this_d + NP) +
                                           ",\"target\":" + std::to_string(target_pid) + ",\"type\":\"hora\",\"ura_markers\":" +
                                           json_str_array(um) + "}",
                                       "hora", w, nullptr);
                        }
                    }
                }
                for (int i = 0; i < NP; i++) {
                    players[i].score += total_deltas[i];
                    players[i].score_delta = total_deltas[i];
                }
                _initialize_next_round(oya_won, false);
            } else if (call_claim) {
                uint8_t claimer = call_claim->first;
                Action action = call_claim->second;
                PlayerState& C = players[claimer];
                _accept_riichi();
                is_rinshan_flag = false;
                is_first_turn = false;
                C.missed_agari_doujun = false;
                if (last_discard) players[last_discard->first].nagashi_eligible = false;
                for (auto& p : players) p.ippatsu_cycle = false;
                if (action.type == AT_DAIMINKAN) {
                    current_player = claimer;
                    active_players = {claimer};
                    C.forbidden_discards.clear();
                    _resolve_kan(claimer, action);
                    return;
                }
                for (uint8_t t : action.consume) {
                    int idx = vec_position(C.hand, t);
                    if (idx >= 0) C.hand.erase(C.hand.begin() + idx);
                }
                uint8_t discarder = last_discard->first, tile = last_discard->second;
                std::vector<uint8_t> tiles = action.consume;
                tiles.push_back(tile);
                std::sort(tiles.begin(), tiles.end());
                Meld m;
                m.meld_type = action.type == AT_PON ? MT_PON : MT_CHI;
                m.tiles = tiles;
                m.opened = true;
                m.from_who = (int8_t)discarder;
                m.called_tile = tile;
                C.melds.push_back(m);
                if (!skip_mjai_logging) {
                    const char* ts = action.type == AT_PON ? "pon" : "chi";
                    push_event("{\"actor\":" + std::to_string(claimer) + ",\"consumed\":" + tiles_json(action.consume) + ",\"pai\":\"" +
                                   tid_to_mjai(tile) + "\",\"target\":" + std::to_string(discarder) + ",\"type\":\"" + ts + "\"}",
                               ts, claimer, nullptr);
                }
                if (m.meld_type == MT_PON) pao_check(claimer, discarder, tile);
                current_player = claimer;
                phase = WAIT_ACT;
                active_players = {claimer};
                C.forbidden_discards.clear();
                if (action.type == AT_PON) {
                    C.forbidden_discards.push_back(tile);
                } else if (action.type == AT_CHI) {
                    C.forbidden_discards.push_back(tile);
                    int t34 = tile / 4;
                    std::vector<int> c34;
                    for (uint8_t x : action.consume) c34.push_back(x / 4);
                    std::sort(c34.begin(), c34.end());
                    if (c34[0] == t34 + 1 && c34[1] == t34 + 2) {
                        if (t34 % 9 <= 5) C.forbidden_discards.push_back((uint8_t)((t34 + 3) * 4));
                    } else if (t34 >= 2 && c34[1] == t34 - 1 && c34[0] == t34 - 2 && t34 % 9 >= 3) {
                        C.forbidden_discards.push_back((uint8_t)((t34 - 3) * 4));
                    }
                }
                needs_tsumo = false;
                drawn_tile = -1;
            } else {
                current_claims.clear();
                active_players.clear();
                if (pending_kan) {
                    auto pk = *pending_kan;
                    pending_kan.reset();
                    if (pk.second.type == AT_KITA) {  // state_3p/mod.rs:1201-1209
                        for (auto& p : players) p.ippatsu_cycle = false;
                        resolve_kita_rinshan(pk.first);
                    } else {
                        _resolve_kan(pk.first, pk.second);
                    }
                } else {
                    _accept_riichi();
                    turn_count += 1;
                    current_player = (uint8_t)((current_player + 1) % NP);
                    _deal_next();
                    if (turn_count >= (uint32_t)NP) is_first_turn = false;
                }
            }
        }
    }

    int32_t yakuman_val(uint32_t yid) const {
        if (yid == 47 && rule.is_junsei_chuurenpoutou_double) return 2;
        if (yid == 48 && rule.is_suuankou_tanki_double) return 2;
        if (yid == 49 && rule.is_kokushi_musou_13machi_double) return 2;
        if (yid == 50 && rule.is_daisuushii_double) return 2;
        return 1;
    }
    // mod.rs:720-745 / 1005-1030
    void cap_double_yakuman(WinResult& res, bool is_oya, bool tsumo, uint32_t honba_) const {
        if (res.yakuman && res.han > 13) {
            uint32_t cap = 0;
            for (uint32_t y : res.yaku) {
                if (y == 47 && !rule.is_junsei_chuurenpoutou_double) cap += 13;
                if (y == 48 && !rule.is_suuankou_tanki_double) cap += 13;
                if (y == 49 && !rule.is_kokushi_musou_13machi_double) cap += 13;
                if (y == 50 && !rule.is_daisuushii_double) cap += 13;
            }
            if (cap > 0) {
                uint32_t h = res.han > cap ? res.han - cap : 0;
                res.han = h < 13 ? 13 : h;
                Score s = calculate_score((uint8_t)res.han, 0, is_oya, tsumo, honba_, NP);
                res.ron_agari = s.pay_ron;
                res.tsumo_agari_oya = s.pay_tsumo_oya;
                res.tsumo_agari_ko = s.pay_tsumo_ko;
            }
        }
    }
    // mod.rs:1228-1259 / 1443-1472
    void pao_check(uint8_t claimer, uint8_t discarder, uint8_t tile) {
        uint8_t tv = tile / 4;
        PlayerState& C = players[claimer];
        if (tv >= 31 && tv <= 33) {
            int n = 0;
            for (auto& m : C.melds) {
                uint8_t t = m.tiles[0] / 4;
                if (t >= 31 && t <= 33 && m.meld_type != MT_CHI) n++;
            }
            if (n == 3) C.pao[37] = discarder;
        } else if (tv >= 27 && tv <= 30) {
            int n = 0;
            for (auto& m : C.melds) {
                uint8_t t = m.tiles[0] / 4;
                if (t >= 27 && t <= 30 && m.meld_type != MT_CHI) n++;
            }
            if (n == 4) C.pao[50] = discarder;
        }
    }

    // mod.rs:1317-1413
    void _resolve_discard(uint8_t pid, uint8_t tile, bool tsumogiri) {
        PlayerState& P = players[pid];
        if (sanma) pending_kan.reset();  // quirk Q11 (state_3p/mod.rs:1224-1227)
        is_rinshan_flag = false;
        P.ippatsu_cycle = false;
        P.discards.push_back(tile);
        last_discard = std::make_pair(pid, tile);
        drawn_tile = -1;
        P.discard_from_hand.push_back(!tsumogiri);
        P.discard_is_riichi.push_back(P.riichi_stage);
        if (!tsumogiri) last_tedashis[pid] = tile;
        needs_tsumo = true;
        if (P.riichi_stage) {
            P.riichi_declared = true;
            if (is_first_turn) P.double_riichi_declared = true;
            P.riichi_declaration_index = (int)P.discards.size() - 1;
            P.riichi_stage = false;
            riichi_pending_acceptance = pid;
        }
        while (wall.pending_kan_dora_count > 0) {
            wall.pending_kan_dora_count--;
            _reveal_kan_dora();
        }
        if (!skip_mjai_logging)
            push_event("{\"actor\":" + std::to_string(pid) + ",\"pai\":\"" + tid_to_mjai(tile) + "\",\"tsumogiri\":" +
                           (tsumogiri ? "true" : "false") + ",\"type\":\"dahai\"}",
                       "dahai", pid, nullptr);
        P.missed_agari_doujun = false;
        P.nagashi_eligible = P.nagashi_eligible && is_terminal_tile(tile);
        current_claims.clear();
        active_players.clear();
        bool has_claims = false;
        std::vector<uint8_t> claim_active;
        for (uint8_t i = 0; i < NP; i++) {
            if (i == pid) continue;
            auto r = _get_claim_actions_for_player(i, pid, tile);
            if (r.second) players[i].missed_agari_doujun = true;
            if (!r.first.empty()) {
                has_claims = true;
                claim_active.push_back(i);
                current_claims[i] = r.first;
            }
        }
        if (has_claims) {
            phase = WAIT_RESPONSE;
            active_players = claim_active;
        } else {
            if (riichi_pending_acceptance >= 0) _accept_riichi();
            if (!check_abortive_draw()) {
                turn_count += 1;
                current_player = (uint8_t)((pid + 1) % NP);
                _deal_next();
                if (turn_count >= (uint32_t)NP) is_first_turn = false;
            }
        }
    }

    // mod.rs:1415-1547
    void _resolve_kan(uint8_t pid, const Action& action) {
        PlayerState& P = players[pid];
        if (action.type != AT_KAKAN) {
            for (uint8_t t : action.consume) {
                int idx = vec_position(P.hand, t);
                if (idx >= 0) P.hand.erase(P.hand.begin() + idx);
            }
            Meld m;
            if (action.type == AT_ANKAN) {
                m.meld_type = MT_ANKAN;
                m.tiles = action.consume;
                m.from_who = -1;
                m.called_tile = -1;
                m.opened = false;
            } else {
                uint8_t discarder = last_discard->first, tile = last_discard->second;
                m.meld_type = MT_DAIMINKAN;
                m.tiles = action.consume;
                m.tiles.push_back(tile);
                std::sort(m.tiles.begin(), m.tiles.end());
                m.from_who = (int8_t)discarder;
                m.called_tile = tile;
                m.opened = true;
            }
            P.melds.push_back(m);
            if (action.type == AT_DAIMINKAN) pao_check(pid, last_discard->first, last_discard->second);
        }
        is_first_turn = false;
        for (auto& p : players) p.ippatsu_cycle = false;
        if (wall.drawable_count > 0) {
            uint8_t t = wall.tiles.front();
            wall.tiles.erase(wall.tiles.begin());
            wall.drawable_count -= 1;
            P.hand.push_back(t);
            drawn_tile = t;
            wall.rinshan_draw_count += 1;
            is_rinshan_flag = true;
            if (!skip_mjai_logging) {
                if (action.type == AT_ANKAN) {
                    uint8_t tile = action.tile >= 0 ? (uint8_t)action.tile : action.consume[0];
                    push_event("{\"actor\":" + std::to_string(pid) + ",\"consumed\":" + tiles_json(action.consume) + ",\"pai\":\"" +
                                   tid_to_mjai(tile) + "\",\"type\":\"ankan\"}",
                               "ankan", pid, nullptr);
                } else if (action.type == AT_DAIMINKAN) {
                    std::string s = "{\"actor\":" + std::to_string(pid) + ",\"consumed\":" + tiles_json(action.consume);
                    if (last_discard)
                        s += ",\"pai\":\"" + tid_to_mjai(last_discard->second) + "\",\"target\":" + std::to_string(last_discard->first);
                    s += ",\"type\":\"daiminkan\"}";
                    push_event(s, "daiminkan", pid, nullptr);
                }
            }
            while (wall.pending_kan_dora_count > 0) {
                wall.pending_kan_dora_count--;
                _reveal_kan_dora();
            }
            if (action.type == AT_ANKAN)
                _reveal_kan_dora();
            else
                wall.pending_kan_dora_count += 1;
            ev_tsumo(pid, t);
            phase = WAIT_ACT;
            active_players = {pid};
        }
    }

    // state_3p/sanma.rs:9-144
    void handle_kita(uint8_t pid, const Action& act) {
        PlayerState& P = players[pid];
        int tile;
        if (act.tile >= 0 && act.tile / 4 == 30)
            tile = act.tile;
        else {
            tile = -1;
            for (uint8_t t : P.hand)
                if (t / 4 == 30) {
                    tile = t;
                    break;
                }
            if (tile < 0) tile = act.tile >= 0 ? act.tile : (act.consume.empty() ? 0 : act.consume[0]);
        }
        int idx = vec_position(P.hand, (uint8_t)tile);
        if (idx >= 0) P.hand.erase(P.hand.begin() + idx);
        P.kita_tiles.push_back((uint8_t)tile);
        is_first_turn = false;
        if (!skip_mjai_logging)
            push_event("{\"actor\":" + std::to_string(pid) + ",\"pai\":\"" + tid_to_mjai((uint8_t)tile) + "\",\"type\":\"kita\"}", "kita",
                       pid, nullptr);
        while (wall.pending_kan_dora_count > 0) {
            wall.pending_kan_dora_count--;
            _reveal_kan_dora();
        }
        std::vector<uint8_t> ronners;
        for (uint8_t i = 0; i < NP; i++) {
            if (i == pid) continue;
            const PlayerState& Q = players[i];
            HandEvaluator he(Q.hand, Q.melds, true);
            bool furiten = false;
            for (uint8_t w : he.get_waits_u8()) {
                for (uint8_t d : Q.discards)
                    if (d / 4 == w) furiten = true;
                if (furiten) break;
            }
            if (Q.missed_agari_riichi || Q.missed_agari_doujun) furiten = true;
            if (furiten) continue;
            Conditions c;
            c.tsumo = false;
            c.riichi = Q.riichi_declared;
            c.double_riichi = Q.double_riichi_declared;
            c.ippatsu = Q.ippatsu_cycle;
            c.chankan = false;
            c.player_wind = (uint8_t)((i + NP - oya) % NP);
            c.round_wind = round_wind % 4;
            c.riichi_sticks = riichi_sticks;
            c.honba = honba;
            c.is_sanma = true;
            c.num_players = 3;
            c.kita_count = (uint8_t)Q.kita_tiles.size();
            WinResult r = he.calc((uint8_t)tile, wall.dora_indicators, {}, c);
            if (r.is_win && (r.yakuman || r.han >= 1)) {
                ronners.push_back(i);
                current_claims[i].push_back(Action(AT_RON, tile, {}, i));
            }
        }
        if (!ronners.empty()) {
            phase = WAIT_RESPONSE;
            active_players = ronners;
            last_discard = std::make_pair(pid, (uint8_t)tile);
            pending_kan = std::make_pair(pid, act);
        } else {
            for (auto& p : players) p.ippatsu_cycle = false;
            resolve_kita_rinshan(pid);
        }
    }
    // state_3p/sanma.rs:171-204
    void resolve_kita_rinshan(uint8_t pid) {
        if (wall.drawable_count > 0) {
            while (wall.pending_kan_dora_count > 0) {
                wall.pending_kan_dora_count--;
                _reveal_kan_dora();
            }
            if (wall.tiles.empty()) return;
            uint8_t t = wall.tiles.front();  // draw_rinshan_tile, state_3p/wall.rs:117-124
            wall.tiles.erase(wall.tiles.begin());
            wall.drawable_count = wall.drawable_count > 0 ? wall.drawable_count - 1 : 0;
            players[pid].hand.push_back(t);
            drawn_tile = t;
            wall.rinshan_draw_count += 1;
            is_rinshan_flag = true;
            ev_tsumo(pid, t);
            phase = WAIT_ACT;
            active_players = {pid};
        }
    }

    // mod.rs:1549-1567
    void _accept_riichi() {
        if (riichi_pending_acceptance >= 0) {
            int p = riichi_pending_acceptance;
            players[p].score -= 1000;
            players[p].score_delta -= 1000;
            riichi_sticks += 1;
            players[p].riichi_declared = true;
            players[p].ippatsu_cycle = true;
            ev_simple_actor("reach_accepted", p);
            riichi_pending_acceptance = -1;
        }
    }

    // mod.rs:1569-1593
    void _deal_next() {
        is_rinshan_flag = false;
        if (wall.drawable_count == 0) {
            _trigger_ryukyoku("exhaustive_draw");
            return;
        }
        if (!wall.tiles.empty()) {
            uint8_t t = wall.tiles.back();
            wall.tiles.pop_back();
            wall.drawable_count -= 1;
            uint8_t pid = current_player;
            players[pid].hand.push_back(t);
            drawn_tile = t;
            needs_tsumo = false;
            phase = WAIT_ACT;
            active_players = {pid};
            ev_tsumo(pid, t);
            players[pid].forbidden_discards.clear();
        }
    }

    // mod.rs:1595-1688
    void _initialize_next_round(bool oya_won, bool is_draw) {
        if (is_done) return;
        const uint8_t np = NP;
        for (auto& p : players)
            if (p.score < 0) {
                _process_end_game();
                return;
            }
        int32_t dealer_score = players[oya].score;
        bool dealer_is_top = true;
        for (int seat = 0; seat < NP; seat++) {
            bool ok = seat == oya || dealer_score > players[seat].score || (dealer_score == players[seat].score && oya <= seat);
            if (!ok) dealer_is_top = false;
        }
        bool is_last_regular = false;
        const int32_t goal = sanma ? 40000 : 30000;  // state_3p/mod.rs:1524,1553,1561
        if (game_mode == 1 || game_mode == 4) is_last_regular = round_wind == 0 && oya == np - 1;
        if (game_mode == 2 || game_mode == 5) is_last_regular = round_wind == 1 && oya == np - 1;
        if (oya_won && is_last_regular && dealer_is_top && dealer_score >= goal) {
            _process_end_game();
            return;
        }
        uint8_t next_honba = honba, next_oya = oya, next_rw = round_wind;
        if (oya_won) {
            next_honba = next_honba == 255 ? 255 : next_honba + 1;
        } else if (is_draw) {
            next_honba = next_honba == 255 ? 255 : next_honba + 1;
            next_oya = (next_oya + 1) % np;
            if (next_oya == 0) next_rw += 1;
        } else {
            next_honba = 0;
            next_oya = (next_oya + 1) % np;
            if (next_oya == 0) next_rw += 1;
        }
        int32_t max_score = players[0].score;
        for (int i = 0; i < NP; i++) max_score = std::max(max_score, players[i].score);
        switch (game_mode) {
            case 1:
            case 4:
                if (next_rw >= 1 && (max_score >= goal || next_rw > 1)) {
                    _process_end_game();
                    return;
                }
                break;
            case 2:
            case 5:
                if (next_rw >= 2 && (max_score >= goal || next_rw > 2)) {
                    _process_end_game();
                    return;
                }
                break;
            case 0:
            case 3: _process_end_game(); return;
            default:
                if (next_rw >= 1) {
                    _process_end_game();
                    return;
                }
        }
        ev_type_only("end_kyoku");
        std::vector<int32_t> sc;
        for (int i = 0; i < NP; i++) sc.push_back(players[i].score);
        _initialize_round(next_oya, next_rw, next_honba, riichi_sticks, nullptr, &sc);
    }

    // mod.rs:1695-1844
    void _initialize_round(uint8_t oya_, uint8_t round_wind_, uint8_t honba_, uint32_t kyotaku,
                           const std::vector<uint8_t>* wall_, const std::vector<int32_t>* scores) {
        oya = oya_;
        kyoku_idx = oya_;
        current_player = oya_;
        honba = honba_;
        riichi_sticks = kyotaku;
        round_wind = round_wind_;
        for (auto& p : players) p.reset_round();
        is_done = false;
        current_claims.clear();
        pending_kan.reset();
        is_rinshan_flag = false;
        wall.rinshan_draw_count = 0;
        wall.pending_kan_dora_count = 0;
        is_first_turn = true;
        riichi_pending_acceptance = -1;
        turn_count = 0;
        needs_tsumo = true;
        last_discard.reset();
        win_results.clear();
        for (int i = 0; i < NP; i++) riichi_sutehais[i] = last_tedashis[i] = -1;
        if (scores)
            for (size_t i = 0; i < scores->size() && i < (size_t)NP; i++) players[i].score = (*scores)[i];
        if (wall_)
            wall.load_wall(*wall_);
        else
            wall.shuffle();
        for (int r = 0; r < 3; r++)
            for (int idx = 0; idx < NP; idx++) {
                int p = (idx + oya) % NP;
                for (int k = 0; k < 4; k++)
                    if (!wall.tiles.empty()) {
                        players[p].hand.push_back(wall.tiles.back());
                        wall.tiles.pop_back();
                    }
            }
        for (int idx = 0; idx < NP; idx++) {
            int p = (idx + oya) % NP;
            if (!wall.tiles.empty()) {
                players[p].hand.push_back(wall.tiles.back());
                wall.tiles.pop_back();
            }
        }
        for (auto& p : players) std::sort(p.hand.begin(), p.hand.end());
        wall.drawable_count = (uint8_t)(wall.tiles.size() - 14);
        if (!skip_mjai_logging) {
            static const char* winds[4] = {"E", "S", "W", "N"};
            int32_t sc[4];
            for (int i = 0; i < NP; i++) sc[i] = players[i].score;
            std::string head = std::string("{\"bakaze\":\"") + winds[round_wind % 4] + "\",\"dora_marker\":\"" +
                               tid_to_mjai(wall.dora_indicators[0]) + "\",\"honba\":" + std::to_string(honba) +
                               ",\"kyoku\":" + std::to_string(oya + 1) + ",\"kyotaku\":" + std::to_string(kyotaku) +
                               ",\"oya\":" + std::to_string(oya) + ",\"scores\":" + json_int_array(sc, sc + NP) + ",\"tehais\":[";
            std::string tail = "],\"type\":\"start_kyoku\"}";
            std::string th[4], masked[4];
            for (int i = 0; i < NP; i++) {
                th[i] = tiles_json(players[i].hand);
                std::vector<std::string> q(players[i].hand.size(), "?");
                masked[i] = json_str_array(q);
            }
            std::string full = head;
            for (int i = 0; i < NP; i++) full += (i ? "," : "") + th[i];
            full += tail;
            std::string per[4];
            for (int pid = 0; pid < NP; pid++) {
                std::string s = head;
                for (int i = 0; i < NP; i++) {
                    if (i) s += ",";
                    s += (i == pid) ? th[i] : masked[i];
                }
                per[pid] = s + tail;
            }
            push_event(full, "start_kyoku", -1, per);
        }
        current_player = oya;
        phase = WAIT_ACT;
        active_players = {oya};
        if (!wall.tiles.empty()) {
            uint8_t t = wall.tiles.back();
            wall.tiles.pop_back();
            wall.drawable_count -= 1;
            players[oya].hand.push_back(t);
            drawn_tile = t;
            needs_tsumo = false;
            ev_tsumo(oya, t);
        } else {
            needs_tsumo = true;
            drawn_tile = -1;
        }
    }

    // mod.rs:1846-1968
    void _trigger_ryukyoku(const std::string& reason) {
        _accept_riichi();
        const int np = NP;
        bool tenpai[4] = {false, false, false, false};
        std::string final_reason = reason;
        std::vector<uint8_t> nagashi;
        static const std::string illegal_prefix = "Error: Illegal Action by Player ";
        if (reason == "exhaustive_draw") {
            for (int i = 0; i < np; i++) {
                HandEvaluator he(players[i].hand, players[i].melds, sanma);
                if (he.is_tenpai()) tenpai[i] = true;
            }
            for (int i = 0; i < np; i++)
                if (players[i].nagashi_eligible) nagashi.push_back((uint8_t)i);
            if (!nagashi.empty()) {
                final_reason = "nagashimangan";
                for (uint8_t w : nagashi) {
                    bool is_oya = w == oya;
                    Score s = calculate_score(5, 30, is_oya, true, 0, np);
                    for (int i = 0; i < np; i++) {
                        if (i == w) continue;
                        int32_t pay = is_oya ? (int32_t)s.pay_tsumo_ko : (i == oya ? (int32_t)s.pay_tsumo_oya : (int32_t)s.pay_tsumo_ko);
                        players[i].score -= pay;
                        players[i].score_delta -= pay;
                        players[w].score += pay;
                        players[w].score_delta += pay;
                    }
                }
            } else {
                int32_t pool = sanma ? 2000 : 3000;  // state_3p/game_mode.rs:39-41
                int num_tp = 0;
                for (int i = 0; i < np; i++) num_tp += tenpai[i];
                if (num_tp > 0 && num_tp < np) {
                    int32_t pk = pool / num_tp, pn = pool / (np - num_tp);
                    for (int i = 0; i < np; i++) {
                        int32_t d = tenpai[i] ? pk : -pn;
                        players[i].score += d;
                        players[i].score_delta = d;
                    }
                }
            }
        } else if (reason.compare(0, illegal_prefix.size(), illegal_prefix) == 0) {
            int pid = std::atoi(reason.c_str() + illegal_prefix.size());
            if (pid >= 0 && pid < np) {
                if (pid == oya) {
                    int32_t penalty = 4000 * (np - 1);
                    int32_t each = penalty / (np - 1);
                    for (int i = 0; i < np; i++) {
                        if (i == pid) {
                            players[i].score -= penalty;
                            players[i].score_delta = -penalty;
                        } else {
                            players[i].score += each;
                            players[i].score_delta = each;
                        }
                    }
                } else {
                    int32_t total = 4000 + 2000 * (np - 2);
                    for (int i = 0; i < np; i++) {
                        if (i == pid) {
                            players[i].score -= total;
                            players[i].score_delta = -total;
                        } else if (i == oya) {
                            players[i].score += 4000;
                            players[i].score_delta = 4000;
                        } else {
                            players[i].score += 2000;
                            players[i].score_delta = 2000;
                        }
                    }
                }
            }
        }
        bool is_renchan;
        if (final_reason == "exhaustive_draw")
            is_renchan = tenpai[oya];
        else if (final_reason == "nagashimangan") {
            is_renchan = false;
            for (uint8_t w : nagashi)
                if (w == oya) is_renchan = true;
        } else
            is_renchan = true;
        if (!skip_mjai_logging) {
            int32_t d[4];
            for (int i = 0; i < np; i++) d[i] = players[i].score_delta;
            push_event("{\"deltas\":" + json_int_array(d, d + np) + ",\"reason\":\"" + final_reason + "\",\"type\":\"ryukyoku\"}",
                       "ryukyoku", -1, nullptr);
        }
        _initialize_next_round(is_renchan, true);
    }

    // mod.rs:1970-2019
    bool check_abortive_draw() {
        bool turns_ok = true, melds_empty = true;
        for (auto& p : players) {
            if (p.discards.size() != 1) turns_ok = false;
            if (!p.melds.empty()) melds_empty = false;
        }
        if (!sanma && turns_ok && melds_empty && !players[0].discards.empty()) {
            uint8_t first = players[0].discards[0] / 4;
            if (first >= 27 && first <= 30) {
                bool all = true;
                for (auto& p : players)
                    if (p.discards.empty() || p.discards[0] / 4 != first) all = false;
                if (all) {
                    _trigger_ryukyoku("sufuurenta");
                    return true;
                }
            }
        }
        std::vector<int> owners;
        for (int pid = 0; pid < NP; pid++)
            for (auto& m : players[pid].melds)
                if (meld_is_kan(m)) owners.push_back(pid);
        if (owners.size() == 4) {
            bool same = true;
            for (int o : owners)
                if (o != owners[0]) same = false;
            if (!same) {
                _trigger_ryukyoku("suukansansen");
                return true;
            }
        }
        bool all_riichi = true;
        for (auto& p : players)
            if (!p.riichi_declared) all_riichi = false;
        if (!sanma && all_riichi) {
            _trigger_ryukyoku("suucha_riichi");
            return true;
        }
        return false;
    }

    // mod.rs:2021-2046
    void _reveal_kan_dora() {
        size_t count = wall.dora_indicators.size();
        if (sanma) {  // state_3p/mod.rs:1894-1913: pre-extracted indicators, no bound check
            if (count < 5) {
                wall.dora_indicators.push_back(wall.dora_tiles[count]);
                ev_dora(wall.dora_indicators.back());
            }
            return;
        }
        if (count < 5) {
            size_t raw = 4 + 2 * count;
            size_t base = raw > wall.rinshan_draw_count ? raw - wall.rinshan_draw_count : 0;
            if (base < wall.tiles.size()) {
                wall.dora_indicators.push_back(wall.tiles[base]);
                ev_dora(wall.dora_indicators.back());
            }
        }
    }
    // mod.rs:2048-2057
    std::vector<uint8_t> _get_ura_indicators() const {
        std::vector<uint8_t> v;
        if (sanma) {  // state_3p/mod.rs:1925-1931
            for (size_t i = 0; i < wall.dora_indicators.size() && i < 5; i++) v.push_back(wall.ura_tiles[i]);
            return v;
        }
        for (size_t i = 0; i < wall.dora_indicators.size(); i++) {
            size_t raw = 5 + 2 * i;
            size_t idx = raw > wall.rinshan_draw_count ? raw - wall.rinshan_draw_count : 0;
            if (idx < wall.tiles.size()) v.push_back(wall.tiles[idx]);
        }
        return v;
    }
    // mod.rs:2071-2082
    void _process_end_game() {
        is_done = true;
        ev_type_only("end_kyoku");
        ev_type_only("end_game");
    }
};

}  // namespace orc
