// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement (C++17) of the riichienv-core hand mathematics, written to
// follow the reference's Rust sources statement-by-statement (including quirks)
// so that it can stand in for the un-buildable reference (no Rust toolchain in
// this image).  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use anything under oracle/.
//
// Pinning: checked against the reference's own golden vectors
// (riichienv-core/benches/data/agari_{4p,3p}.json, hands_negative.json, the
// score table in riichienv-core/tests/agari_correctness.rs:288-326, and the
// KATs transcribed in tests/).  See tests/test_oracle_*.py.
//
// Reference files followed (paths relative to /root/reference/riichienv-core/src):
//   types.rs, agari.rs, hand_evaluator.rs, hand_evaluator_3p.rs, yaku.rs,
//   yaku_3p.rs, score.rs
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace orc {

constexpr int TILE_MAX = 34;

// ---------------------------------------------------------------- types.rs:11-52
struct Hand {
    uint8_t counts[TILE_MAX];
    Hand() { std::memset(counts, 0, sizeof(counts)); }
    void add(uint8_t t) {
        if (t < TILE_MAX) counts[t] += 1;
    }
    void remove(uint8_t t) {
        if (t < TILE_MAX && counts[t] > 0) counts[t] -= 1;
    }
    int total() const {
        int s = 0;
        for (int i = 0; i < TILE_MAX; i++) s += counts[i];
        return s;
    }
};

// types.rs:54-62
enum MeldType : uint8_t { MT_CHI = 0, MT_PON = 1, MT_DAIMINKAN = 2, MT_ANKAN = 3, MT_KAKAN = 4 };

// types.rs:98-106
struct Meld {
    MeldType meld_type = MT_CHI;
    std::vector<uint8_t> tiles;
    bool opened = true;
    int8_t from_who = -1;
    int called_tile = -1;  // -1 = None
};

// types.rs:193-210
struct Conditions {
    bool tsumo = false, riichi = false, double_riichi = false, ippatsu = false;
    bool haitei = false, houtei = false, rinshan = false;
    uint8_t player_wind = 0, round_wind = 0;  // Wind 0..3
    bool chankan = false, tsumo_first_turn = false;
    uint32_t riichi_sticks = 0, honba = 0;
    uint8_t kita_count = 0;
    bool is_sanma = false;
    uint8_t num_players = 4;
};

// types.rs:282-293
struct WinResult {
    bool is_win = false, yakuman = false;
    uint32_t ron_agari = 0, tsumo_agari_oya = 0, tsumo_agari_ko = 0;
    std::vector<uint32_t> yaku;
    uint32_t han = 0, fu = 0;
    int pao_payer = -1;
    bool has_win_shape = false;
};

// types.rs:362-367
inline bool is_terminal_tile(uint8_t t136) {
    uint8_t tt = t136 / 4;
    uint8_t rank = tt % 9, suit = tt / 9;
    return suit == 3 || rank == 0 || rank == 8;
}

// ---------------------------------------------------------------- agari.rs
struct Mentsu {
    bool koutsu;  // true = Koutsu(t), false = Shuntsu(t)
    uint8_t t;
};
struct Division {
    uint8_t head = 0;
    std::vector<Mentsu> body;
};

// agari.rs:143-168
inline bool is_kokushi(const Hand& h) {
    static const int idx[13] = {0, 8, 9, 17, 18, 26, 27, 28, 29, 30, 31, 32, 33};
    bool pair_found = false;
    for (int k = 0; k < 13; k++) {
        uint8_t c = h.counts[idx[k]];
        if (c == 0) return false;
        if (c == 2) {
            if (pair_found) return false;
            pair_found = true;
        } else if (c > 2) {
            return false;
        }
    }
    return pair_found;
}

// agari.rs:170-181
inline bool is_chiitoitsu(const Hand& h) {
    int pairs = 0;
    for (int i = 0; i < TILE_MAX; i++) {
        uint8_t c = h.counts[i];
        if (c == 2)
            pairs++;
        else if (c != 0)
            return false;
    }
    return pairs == 7;
}

inline bool valid_seq_start(int i) { return (i <= 6) || (i >= 9 && i <= 15) || (i >= 18 && i <= 24); }

// agari.rs:201-245
inline bool decompose(Hand& h, int start) {
    int i = start;
    while (i < TILE_MAX && h.counts[i] == 0) i++;
    if (i == TILE_MAX) return true;
    if (h.counts[i] >= 3) {
        h.counts[i] -= 3;
        bool ok = decompose(h, i);
        h.counts[i] += 3;
        if (ok) return true;
    }
    if (i < 27 && valid_seq_start(i) && h.counts[i + 1] > 0 && h.counts[i + 2] > 0) {
        h.counts[i] -= 1;
        h.counts[i + 1] -= 1;
        h.counts[i + 2] -= 1;
        bool ok = decompose(h, i);
        h.counts[i] += 1;
        h.counts[i + 1] += 1;
        h.counts[i + 2] += 1;
        if (ok) return true;
    }
    return false;
}

// agari.rs:183-199
inline bool is_standard_agari(Hand& h) {
    for (int i = 0; i < TILE_MAX; i++) {
        if (h.counts[i] >= 2) {
            h.counts[i] -= 2;
            bool ok = decompose(h, 0);
            h.counts[i] += 2;
            if (ok) return true;
        }
    }
    return false;
}

// agari.rs:65-73
inline bool is_agari(Hand& h) {
    if (is_kokushi(h)) return true;
    if (is_chiitoitsu(h)) return true;
    return is_standard_agari(h);
}

// agari.rs:15-63  (free-function tenpai test over a raw histogram, with the
// reference's neighbour pruning for the standard form)
inline bool is_tenpai_counts(Hand& h) {
    for (int i = 0; i < TILE_MAX; i++) {
        if (h.counts[i] < 4) {
            h.add((uint8_t)i);
            if (is_kokushi(h)) {
                h.remove((uint8_t)i);
                return true;
            }
            if (is_chiitoitsu(h)) {
                h.remove((uint8_t)i);
                return true;
            }
            uint8_t c = h.counts[i];
            bool check_standard;
            if (i >= 27) {
                check_standard = c >= 2;
            } else {
                bool has_p = (i % 9 > 0) ? h.counts[i - 1] > 0 : false;
                bool has_n = (i % 9 < 8) ? h.counts[i + 1] > 0 : false;
                check_standard = c >= 2 || has_p || has_n;
            }
            if (check_standard && is_standard_agari(h)) {
                h.remove((uint8_t)i);
                return true;
            }
            h.remove((uint8_t)i);
        }
    }
    return false;
}

// agari.rs:96-141
inline void decompose_all(Hand& h, int start, std::vector<Mentsu>& cur, std::vector<std::vector<Mentsu>>& out) {
    int i = start;
    while (i < TILE_MAX && h.counts[i] == 0) i++;
    if (i == TILE_MAX) {
        out.push_back(cur);
        return;
    }
    if (h.counts[i] >= 3) {
        h.counts[i] -= 3;
        cur.push_back(Mentsu{true, (uint8_t)i});
        decompose_all(h, i, cur, out);
        cur.pop_back();
        h.counts[i] += 3;
    }
    if (i < 27) {
        if (valid_seq_start(i) && h.counts[i + 1] > 0 && h.counts[i + 2] > 0) {
            h.counts[i] -= 1;
            h.counts[i + 1] -= 1;
            h.counts[i + 2] -= 1;
            cur.push_back(Mentsu{false, (uint8_t)i});
            decompose_all(h, i, cur, out);
            cur.pop_back();
            h.counts[i] += 1;
            h.counts[i + 1] += 1;
            h.counts[i + 2] += 1;
        }
    }
}

// agari.rs:75-94
inline std::vector<Division> find_divisions(const Hand& hand) {
    std::vector<Division> divs;
    for (int i = 0; i < TILE_MAX; i++) {
        if (hand.counts[i] >= 2) {
            Hand ph = hand;
            ph.counts[i] -= 2;
            std::vector<std::vector<Mentsu>> bodies;
            std::vector<Mentsu> cur;
            decompose_all(ph, 0, cur, bodies);
            for (auto& b : bodies) {
                Division d;
                d.head = (uint8_t)i;
                d.body = b;
                divs.push_back(std::move(d));
            }
        }
    }
    return divs;
}

// ---------------------------------------------------------------- score.rs
struct Score {
    uint32_t total = 0, pay_ron = 0, pay_tsumo_oya = 0, pay_tsumo_ko = 0;
};
inline uint32_t ceil_100(uint32_t v) { return (v + 99) / 100 * 100; }  // score.rs:97-99
inline uint8_t round_up_fu(uint8_t fu) {                                // score.rs:90-95
    if (fu == 25) return 25;
    return (uint8_t)((fu + 9) / 10 * 10);
}
// score.rs:54-88
inline Score make_score_result(uint32_t base, bool is_oya, bool is_tsumo, uint32_t np) {
    uint32_t total_ron = is_oya ? ceil_100(base * 6) : ceil_100(base * 4);
    uint32_t pay_oya, pay_ko;
    if (is_oya) {
        pay_oya = 0;
        pay_ko = ceil_100(base * 2);
    } else {
        pay_oya = ceil_100(base * 2);
        pay_ko = ceil_100(base);
    }
    uint32_t total_tsumo = is_oya ? pay_ko * (np - 1) : pay_oya + pay_ko * (np - 2);
    Score s;
    if (is_tsumo) {
        s.total = total_tsumo;
        s.pay_ron = 0;
        s.pay_tsumo_oya = pay_oya;
        s.pay_tsumo_ko = pay_ko;
    } else {
        s.total = total_ron;
        s.pay_ron = total_ron;
    }
    return s;
}
// score.rs:13-52
inline Score calculate_score(uint8_t han, uint8_t fu, bool is_oya, bool is_tsumo, uint32_t honba, uint8_t num_players) {
    uint32_t np = num_players;
    Score s;
    if (han >= 5) {
        uint32_t base;
        if (han == 5)
            base = 2000;
        else if (han <= 7)
            base = 3000;
        else if (han <= 10)
            base = 4000;
        else if (han <= 12)
            base = 6000;
        else
            base = 8000u * (han / 13);
        s = make_score_result(base, is_oya, is_tsumo, np);
    } else {
        uint8_t f = round_up_fu(fu);
        uint32_t bp = (uint32_t)f * (1u << (2 + han));
        s = make_score_result(bp > 2000 ? 2000 : bp, is_oya, is_tsumo, np);
    }
    if (is_tsumo) {
        s.pay_tsumo_oya += honba * 100;
        s.pay_tsumo_ko += honba * 100;
        s.total += honba * 100 * (np - 1);
    } else {
        uint32_t hr = honba * 100 * (np - 1);
        s.pay_ron += hr;
        s.total += hr;
    }
    return s;
}

// ---------------------------------------------------------------- yaku.rs
enum : uint32_t {
    ID_TSUMO = 1, ID_RIICHI = 2, ID_CHANKAN = 3, ID_RINSHAN = 4, ID_HAITEI = 5, ID_HOUTEI = 6,
    ID_HAKU = 7, ID_HATSU = 8, ID_CHUN = 9, ID_JIKAZE = 10, ID_BAKAZE = 11, ID_TANYAO = 12,
    ID_IPEIKO = 13, ID_PINFU = 14, ID_CHANTA = 15, ID_ITTSU = 16, ID_SANSHOKU = 17,
    ID_DOUBLE_RIICHI = 18, ID_SANSHOKU_DOKO = 19, ID_SANKANTSU = 20, ID_TOITOI = 21,
    ID_SANANKOU = 22, ID_SHOSANGEN = 23, ID_HONROUTO = 24, ID_CHITOITSU = 25, ID_JUNCHAN = 26,
    ID_HONITSU = 27, ID_RYANPEIKO = 28, ID_CHINITSU = 29, ID_IPPATSU = 30, ID_DORA = 31,
    ID_AKADORA = 32, ID_URADORA = 33, ID_NUKIDORA = 34, ID_TENHO = 35, ID_CHIHO = 36,
    ID_DAISANGEN = 37, ID_SUANKO = 38, ID_TSUISO = 39, ID_RYUISOU = 40, ID_CHINROUTO = 41,
    ID_KOKUSHI = 42, ID_SHOUSUUSHI = 43, ID_SUKANTSU = 44, ID_CHUUREN = 45,
    ID_JUNSEI_CHUUREN = 47, ID_SUANKO_TANKI = 48, ID_KOKUSHI_13 = 49, ID_DAISUUSHI = 50
};

// yaku.rs:183-190
struct YakuResult {
    uint8_t han = 0, fu = 0;
    std::vector<uint32_t> yaku_ids;
    uint8_t yakuman_count = 0;
};
// yaku.rs:192-209 (+ yaku_3p.rs nukidora_count)
struct YakuContext {
    bool is_menzen = true, is_reach = false, is_ippatsu = false, is_tsumo = false;
    bool is_haitei = false, is_houtei = false, is_rinshan = false, is_chankan = false;
    bool is_tsumo_first_turn = false, is_daburu_reach = false;
    uint8_t dora_count = 0, aka_dora = 0, ura_dora_count = 0;
    uint8_t round_wind = 27, seat_wind = 27;
    uint8_t nukidora_count = 0;  // 3P only (yaku_3p.rs)
    bool sanma = false;
};

inline bool y_is_terminal(uint8_t t) { return t >= 27 || t % 9 == 0 || t % 9 == 8; }  // yaku.rs:767-769
inline bool y_is_number_terminal(uint8_t t) { return t < 27 && (t % 9 == 0 || t % 9 == 8); }
inline bool y_is_honor(uint8_t t) { return t >= 27; }
inline bool meld_is_kan(const Meld& m) { return m.meld_type == MT_DAIMINKAN || m.meld_type == MT_ANKAN || m.meld_type == MT_KAKAN; }

// yaku.rs:1212-1228
inline bool is_tanyao(const Hand& h, const std::vector<Meld>& melds) {
    static const int term[13] = {0, 8, 9, 17, 18, 26, 27, 28, 29, 30, 31, 32, 33};
    for (int k = 0; k < 13; k++)
        if (h.counts[term[k]] > 0) return false;
    for (auto& m : melds)
        for (uint8_t t : m.tiles)
            for (int k = 0; k < 13; k++)
                if (term[k] == t) return false;
    return true;
}
// yaku.rs:777-809
inline bool is_honitsu(const Hand& h, const std::vector<Meld>& melds) {
    bool suits[3] = {false, false, false};
    bool has_honor = false;
    auto mark = [&](int i) {
        if (i < 9)
            suits[0] = true;
        else if (i < 18)
            suits[1] = true;
        else if (i < 27)
            suits[2] = true;
        else
            has_honor = true;
    };
    for (int i = 0; i < TILE_MAX; i++)
        if (h.counts[i] > 0) mark(i);
    for (auto& m : melds)
        for (uint8_t t : m.tiles) mark(t);
    return (int)suits[0] + suits[1] + suits[2] == 1 && has_honor;
}
// yaku.rs:811-841
inline bool is_chinitsu(const Hand& h, const std::vector<Meld>& melds) {
    bool suits[3] = {false, false, false};
    for (int i = 0; i < TILE_MAX; i++)
        if (h.counts[i] > 0) {
            if (i >= 27) return false;
            suits[i / 9] = true;
        }
    for (auto& m : melds)
        for (uint8_t t : m.tiles) {
            if (t >= 27) return false;
            suits[t / 9] = true;
        }
    return (int)suits[0] + suits[1] + suits[2] == 1;
}
// yaku.rs:692-705
inline bool is_honroutou(const Hand& h, const std::vector<Meld>& melds) {
    for (int i = 0; i < TILE_MAX; i++)
        if (h.counts[i] > 0 && !y_is_terminal((uint8_t)i)) return false;
    for (auto& m : melds)
        for (uint8_t t : m.tiles)
            if (!y_is_terminal(t)) return false;
    return true;
}
// yaku.rs:707-731
inline bool is_junchan(const Division& d, const std::vector<Meld>& melds) {
    if (!y_is_number_terminal(d.head)) return false;
    for (auto& m : d.body) {
        if (m.koutsu) {
            if (!y_is_number_terminal(m.t)) return false;
        } else {
            if (!y_is_number_terminal(m.t) && !y_is_number_terminal(m.t + 2)) return false;
        }
    }
    for (auto& m : melds) {
        bool all_non = true;
        for (uint8_t t : m.tiles)
            if (y_is_number_terminal(t)) all_non = false;
        if (all_non) return false;
    }
    return true;
}
// yaku.rs:733-765
inline bool is_chantai(const Division& d, const std::vector<Meld>& melds) {
    if (!y_is_terminal(d.head)) return false;
    bool has_honor = y_is_honor(d.head);
    for (auto& m : d.body) {
        if (m.koutsu) {
            if (!y_is_terminal(m.t)) return false;
            if (y_is_honor(m.t)) has_honor = true;
        } else {
            if (!y_is_terminal(m.t) && !y_is_terminal(m.t + 2)) return false;
        }
    }
    for (auto& m : melds) {
        bool all_non = true, any_honor = false;
        for (uint8_t t : m.tiles) {
            if (y_is_terminal(t)) all_non = false;
            if (y_is_honor(t)) any_honor = true;
        }
        if (all_non) return false;
        if (any_honor) has_honor = true;
    }
    return has_honor;
}
// yaku.rs:1057-1069
inline bool is_tsuu_iisou(const Hand& h, const std::vector<Meld>& melds) {
    for (int i = 0; i < 27; i++)
        if (h.counts[i] > 0) return false;
    for (auto& m : melds)
        for (uint8_t t : m.tiles)
            if (t < 27) return false;
    return true;
}
// yaku.rs:1071-1083
inline bool is_chinroutou(const Hand& h, const std::vector<Meld>& melds) {
    for (int i = 0; i < TILE_MAX; i++)
        if (h.counts[i] > 0 && !y_is_number_terminal((uint8_t)i)) return false;
    for (auto& m : melds)
        for (uint8_t t : m.tiles)
            if (!y_is_number_terminal(t)) return false;
    return true;
}
// yaku.rs:1085-1099
inline bool is_ryuu_iisou(const Hand& h, const std::vector<Meld>& melds) {
    auto green = [](int t) { return t == 19 || t == 20 || t == 21 || t == 23 || t == 25 || t == 32; };
    for (int i = 0; i < TILE_MAX; i++)
        if (h.counts[i] > 0 && !green(i)) return false;
    for (auto& m : melds)
        for (uint8_t t : m.tiles)
            if (!green(t)) return false;
    return true;
}
// yaku.rs:1101-1131
inline bool is_chuuren_poutou(const Hand& h) {
    uint8_t counts[9] = {0};
    int suit = -1;
    for (int i = 0; i < TILE_MAX; i++) {
        uint8_t c = h.counts[i];
        if (c > 0) {
            if (i >= 27) return false;
            int s = i / 9;
            if (suit >= 0) {
                if (suit != s) return false;
            } else
                suit = s;
            counts[i % 9] = c;
        }
    }
    if (counts[0] < 3 || counts[8] < 3) return false;
    for (int k = 1; k < 8; k++)
        if (counts[k] == 0) return false;
    return true;
}
// yaku.rs:1133-1146
inline bool is_chuuren_9_wait(const Hand& h, uint8_t win_tile) {
    if (win_tile >= 27) return false;
    int val = win_tile % 9;
    const uint8_t* c = &h.counts[win_tile / 9 * 9];
    if (val == 0 || val == 8) return c[val] == 4;
    return c[val] == 2;
}
// yaku.rs:1148-1182
inline bool check_ittsu(const Division& d, const std::vector<Meld>& melds) {
    for (int off : {0, 9, 18}) {
        bool a = false, b = false, c = false;
        for (auto& m : d.body)
            if (!m.koutsu) {
                if (m.t == off)
                    a = true;
                else if (m.t == off + 3)
                    b = true;
                else if (m.t == off + 6)
                    c = true;
            }
        for (auto& m : melds)
            if (m.meld_type == MT_CHI) {
                uint8_t t = m.tiles[0];
                if (t == off)
                    a = true;
                else if (t == off + 3)
                    b = true;
                else if (t == off + 6)
                    c = true;
            }
        if (a && b && c) return true;
    }
    return false;
}
// yaku.rs:1184-1210
inline bool is_sanshoku_doujun(const Division& d, const std::vector<Meld>& melds) {
    for (int i = 0; i < 7; i++) {
        bool a = false, b = false, c = false;
        for (auto& m : d.body)
            if (!m.koutsu) {
                if (m.t == i) a = true;
                if (m.t == i + 9) b = true;
                if (m.t == i + 18) c = true;
            }
        for (auto& m : melds)
            if (m.meld_type == MT_CHI) {
                uint8_t t = m.tiles[0];
                if (t == i) a = true;
                if (t == i + 9) b = true;
                if (t == i + 18) c = true;
            }
        if (a && b && c) return true;
    }
    return false;
}
// yaku.rs:1230-1280
inline bool is_sanshoku_doukou(const Division& d, const std::vector<Meld>& melds) {
    for (int i = 0; i < 9; i++) {
        bool a = false, b = false, c = false;
        for (auto& m : d.body)
            if (m.koutsu) {
                if (m.t == i) a = true;
                if (m.t == i + 9) b = true;
                if (m.t == i + 18) c = true;
            }
        for (auto& m : melds)
            if (m.meld_type != MT_CHI) {
                uint8_t t = m.tiles[0];
                if (t == i) a = true;
                if (t == i + 9) b = true;
                if (t == i + 18) c = true;
            }
        if (a && b && c) return true;
    }
    return false;
}

// yaku.rs:843-890 (+ yaku_3p.rs:693-696 nukidora after ura)
inline void apply_static_yaku(YakuResult& r, const YakuContext& c) {
    if (c.is_reach && !c.is_daburu_reach) { r.han += 1; r.yaku_ids.push_back(ID_RIICHI); }
    if (c.is_daburu_reach) { r.han += 2; r.yaku_ids.push_back(ID_DOUBLE_RIICHI); }
    if (c.is_ippatsu) { r.han += 1; r.yaku_ids.push_back(ID_IPPATSU); }
    if (c.is_menzen && c.is_tsumo) { r.han += 1; r.yaku_ids.push_back(ID_TSUMO); }
    if (c.is_haitei && c.is_tsumo) { r.han += 1; r.yaku_ids.push_back(ID_HAITEI); }
    if (c.is_houtei && !c.is_tsumo) { r.han += 1; r.yaku_ids.push_back(ID_HOUTEI); }
    if (c.is_rinshan && c.is_tsumo) { r.han += 1; r.yaku_ids.push_back(ID_RINSHAN); }
    if (c.is_chankan && !c.is_tsumo) { r.han += 1; r.yaku_ids.push_back(ID_CHANKAN); }
    if (c.dora_count > 0) { r.han += c.dora_count; r.yaku_ids.push_back(ID_DORA); }
    if (c.aka_dora > 0) { r.han += c.aka_dora; r.yaku_ids.push_back(ID_AKADORA); }
    if (c.ura_dora_count > 0) { r.han += c.ura_dora_count; r.yaku_ids.push_back(ID_URADORA); }
    if (c.sanma && c.nukidora_count > 0) { r.han += c.nukidora_count; r.yaku_ids.push_back(ID_NUKIDORA); }
}

// yaku.rs:892-1055.  wg_idx: -1 = None (pair wait), else index into div.body.
inline void apply_yakuman(YakuResult& res, const Hand& hand, const std::vector<Meld>& melds, const YakuContext& ctx,
                          const Division& div, int wg_idx, uint8_t win_tile) {
    uint8_t yakuman_count = 0;
    if (is_tsuu_iisou(hand, melds)) { yakuman_count += 1; res.yaku_ids.push_back(ID_TSUISO); }
    if (is_chinroutou(hand, melds)) { yakuman_count += 1; res.yaku_ids.push_back(ID_CHINROUTO); }
    if (is_ryuu_iisou(hand, melds)) { yakuman_count += 1; res.yaku_ids.push_back(ID_RYUISOU); }
    int kans = 0;
    for (auto& m : melds)
        if (meld_is_kan(m)) kans++;
    if (kans == 4) { yakuman_count += 1; res.yaku_ids.push_back(ID_SUKANTSU); }
    if (ctx.is_menzen && (div.body.size() + melds.size()) == 4) {
        if (is_chuuren_poutou(hand)) {
            if (is_chuuren_9_wait(hand, win_tile)) {
                yakuman_count += 2;
                res.yaku_ids.push_back(ID_JUNSEI_CHUUREN);
            } else {
                yakuman_count += 1;
                res.yaku_ids.push_back(ID_CHUUREN);
            }
        }
    }
    if (ctx.is_tsumo_first_turn && ctx.is_menzen && ctx.is_tsumo) {
        yakuman_count += 1;
        res.yaku_ids.push_back(ctx.seat_wind == 27 ? ID_TENHO : ID_CHIHO);
    }
    int closed_koutsu = 0;
    for (size_t idx = 0; idx < div.body.size(); idx++)
        if (div.body[idx].koutsu) {
            if (!ctx.is_tsumo && (int)idx == wg_idx) continue;
            closed_koutsu++;
        }
    for (auto& m : melds)
        if (m.meld_type == MT_ANKAN) closed_koutsu++;
    if (closed_koutsu == 4) {
        if (wg_idx < 0) {
            yakuman_count += 2;
            res.yaku_ids.push_back(ID_SUANKO_TANKI);
        } else {
            yakuman_count += 1;
            res.yaku_ids.push_back(ID_SUANKO);
        }
    }
    auto dragon = [&](uint8_t t) {
        for (auto& m : div.body)
            if (m.koutsu && m.t == t) return true;
        for (auto& m : melds)
            for (uint8_t x : m.tiles)
                if (x == t) return true;  // m.tiles.contains(&t)
        return false;
    };
    if (dragon(31) && dragon(32) && dragon(33)) { yakuman_count += 1; res.yaku_ids.push_back(ID_DAISANGEN); }
    int wind_koutsu = 0, wind_pair = 0;
    for (uint8_t w = 27; w <= 30; w++) {
        bool has = false;
        for (auto& m : div.body)
            if (m.koutsu && m.t == w) has = true;
        for (auto& m : melds)
            if (m.tiles[0] == w && m.meld_type != MT_CHI) has = true;
        if (has)
            wind_koutsu++;
        else if (div.head == w)
            wind_pair++;
    }
    if (wind_koutsu == 4) {
        yakuman_count += 2;
        res.yaku_ids.push_back(ID_DAISUUSHI);
    } else if (wind_koutsu == 3 && wind_pair == 1) {
        yakuman_count += 1;
        res.yaku_ids.push_back(ID_SHOUSUUSHI);
    }
    if (yakuman_count > 0) {
        res.han = (uint8_t)(13 * yakuman_count);
        res.yakuman_count = yakuman_count;
    }
}

// yaku.rs:561-642
inline uint8_t calculate_fu_with_waiting(const Division& div, const std::vector<Meld>& melds, const YakuContext& ctx,
                                         int wg_idx, uint8_t win_tile) {
    uint8_t fu = 20;
    if (ctx.is_tsumo)
        fu += 2;
    else if (ctx.is_menzen)
        fu += 10;
    if (div.head == ctx.round_wind) fu += 2;
    if (div.head == ctx.seat_wind) fu += 2;
    if (div.head >= 31) fu += 2;
    if (wg_idx < 0) {
        fu += 2;
    } else {
        const Mentsu& m = div.body[wg_idx];
        if (!m.koutsu) {
            uint8_t t = m.t;
            if (win_tile == t + 1 || (win_tile == t + 2 && (t % 9 == 0)) || (win_tile == t && (t % 9 == 6))) fu += 2;
        }
    }
    for (size_t idx = 0; idx < div.body.size(); idx++) {
        const Mentsu& m = div.body[idx];
        if (m.koutsu) {
            uint8_t f = 4;
            if (!ctx.is_tsumo && (int)idx == wg_idx) f = 2;
            if (y_is_terminal(m.t)) f *= 2;
            fu += f;
        }
    }
    for (auto& m : melds) {
        if (m.tiles.size() >= 3 && m.tiles[0] == m.tiles[1]) {
            uint8_t f = 2;
            if (!m.opened) f = 4;
            if (y_is_terminal(m.tiles[0])) f *= 2;
            if (meld_is_kan(m)) f *= 4;
            fu += f;
        }
    }
    if (fu == 20 && !ctx.is_tsumo) fu = 30;
    return (uint8_t)((fu + 9) / 10 * 10);
}

// yaku.rs:644-690
inline bool check_pinfu(const Division& div, const std::vector<Meld>& melds, const YakuContext& ctx, int wg_idx,
                        uint8_t win_tile) {
    if (!ctx.is_menzen) return false;
    if (!melds.empty()) return false;
    for (auto& m : div.body)
        if (m.koutsu) return false;
    if (div.head >= 31 || div.head == ctx.round_wind || div.head == ctx.seat_wind) return false;
    if (wg_idx >= 0 && !div.body[wg_idx].koutsu) {
        uint8_t t = div.body[wg_idx].t;
        if (win_tile == t) return !(t % 9 == 6);
        if (win_tile == t + 2) return !(t % 9 == 0);
    }
    return false;
}

// yaku.rs:232-559 / yaku_3p.rs:45-...
inline YakuResult calculate_yaku(const Hand& hand, const std::vector<Meld>& melds, const YakuContext& ctx,
                                 uint8_t win_tile) {
    std::vector<Division> divisions = find_divisions(hand);
    YakuResult best;
    if (divisions.empty()) {
        if (is_kokushi(hand)) {
            if (hand.counts[win_tile] == 2) {
                best.han = 26;
                best.yakuman_count = 2;
                best.yaku_ids.push_back(ID_KOKUSHI_13);
            } else {
                best.han = 13;
                best.yakuman_count = 1;
                best.yaku_ids.push_back(ID_KOKUSHI);
            }
            return best;
        }
        if (is_chiitoitsu(hand)) {
            best.han = 2;
            best.fu = 25;
            best.yaku_ids.push_back(ID_CHITOITSU);
            if (is_tanyao(hand, melds)) { best.han += 1; best.yaku_ids.push_back(ID_TANYAO); }
            if (is_chinitsu(hand, melds)) {
                best.han += 6;
                best.yaku_ids.push_back(ID_CHINITSU);
            } else if (is_honitsu(hand, melds)) {
                best.han += 3;
                best.yaku_ids.push_back(ID_HONITSU);
            }
            if (is_honroutou(hand, melds)) { best.han += 2; best.yaku_ids.push_back(ID_HONROUTO); }
            Division empty;
            empty.head = 0;
            apply_yakuman(best, hand, melds, ctx, empty, -1, win_tile);
            apply_static_yaku(best, ctx);
            return best;
        }
        return best;
    }

    for (const Division& div : divisions) {
        std::vector<int> wgs;  // -1 = None
        if (div.head == win_tile) wgs.push_back(-1);
        for (size_t idx = 0; idx < div.body.size(); idx++) {
            const Mentsu& m = div.body[idx];
            if (m.koutsu) {
                if (m.t == win_tile) wgs.push_back((int)idx);
            } else {
                if (win_tile >= m.t && win_tile <= m.t + 2) wgs.push_back((int)idx);
            }
        }
        if (wgs.empty()) continue;
        for (int wg : wgs) {
            YakuResult res;
            apply_yakuman(res, hand, melds, ctx, div, wg, win_tile);
            if (res.han >= 13) {
                if (res.han > best.han) best = res;
                continue;
            }
            apply_static_yaku(res, ctx);
            if (is_tanyao(hand, melds)) { res.han += 1; res.yaku_ids.push_back(ID_TANYAO); }
            if (check_pinfu(div, melds, ctx, wg, win_tile)) {
                res.han += 1;
                res.yaku_ids.push_back(ID_PINFU);
                res.fu = ctx.is_tsumo ? 20 : 30;
            } else {
                res.fu = calculate_fu_with_waiting(div, melds, ctx, wg, win_tile);
            }
            const uint8_t yakuhai_tiles[5] = {31, 32, 33, ctx.round_wind, ctx.seat_wind};
            for (int i = 0; i < 5; i++) {
                uint8_t t = yakuhai_tiles[i];
                int count = 0;
                for (auto& m : div.body)
                    if (m.koutsu && m.t == t) count++;
                for (auto& m : melds)
                    if (m.tiles[0] == t && m.meld_type != MT_CHI) count++;
                if (count > 0) {
                    res.han += (uint8_t)count;
                    uint32_t id;
                    if (t == 31)
                        id = ID_HAKU;
                    else if (t == 32)
                        id = ID_HATSU;
                    else if (t == 33)
                        id = ID_CHUN;
                    else
                        id = (i == 3) ? ID_BAKAZE : ID_JIKAZE;
                    res.yaku_ids.push_back(id);
                }
            }
            auto dragon_k = [&](uint8_t t) {
                for (auto& m : div.body)
                    if (m.koutsu && m.t == t) return true;
                for (auto& m : melds)
                    if (m.tiles[0] == t && m.meld_type != MT_CHI) return true;
                return false;
            };
            bool hk = dragon_k(31), ht = dragon_k(32), ck = dragon_k(33);
            if (hk && ht && ck) {
            } else {
                int dk = (int)hk + ht + ck;
                int dp = (div.head == 31) + (div.head == 32) + (div.head == 33);
                if (dk == 2 && dp == 1) { res.han += 2; res.yaku_ids.push_back(ID_SHOSANGEN); }
            }
            int koutsu_total = 0;
            for (auto& m : div.body)
                if (m.koutsu) koutsu_total++;
            for (auto& m : melds)
                if (m.meld_type != MT_CHI) koutsu_total++;
            if (koutsu_total == 4) { res.han += 2; res.yaku_ids.push_back(ID_TOITOI); }
            int closed_k = 0;
            for (size_t idx = 0; idx < div.body.size(); idx++)
                if (div.body[idx].koutsu) {
                    if (!ctx.is_tsumo && (int)idx == wg) continue;
                    closed_k++;
                }
            for (auto& m : melds)
                if (m.meld_type == MT_ANKAN) closed_k++;
            if (closed_k == 3) { res.han += 2; res.yaku_ids.push_back(ID_SANANKOU); }
            int kantsu = 0;
            for (auto& m : melds)
                if (meld_is_kan(m)) kantsu++;
            if (kantsu == 3) { res.han += 2; res.yaku_ids.push_back(ID_SANKANTSU); }
            if (ctx.is_menzen) {
                std::vector<uint8_t> st;
                for (auto& m : div.body)
                    if (!m.koutsu) st.push_back(m.t);
                std::sort(st.begin(), st.end());
                int pairs = 0;
                size_t i = 0;
                while (i + 1 < st.size()) {
                    if (st[i] == st[i + 1]) {
                        pairs++;
                        i += 2;
                    } else
                        i += 1;
                }
                if (pairs == 2) {
                    res.han += 3;
                    res.yaku_ids.push_back(ID_RYANPEIKO);
                } else if (pairs == 1) {
                    res.han += 1;
                    res.yaku_ids.push_back(ID_IPEIKO);
                }
            }
            if (check_ittsu(div, melds)) { res.han += ctx.is_menzen ? 2 : 1; res.yaku_ids.push_back(ID_ITTSU); }
            if (is_sanshoku_doujun(div, melds)) { res.han += ctx.is_menzen ? 2 : 1; res.yaku_ids.push_back(ID_SANSHOKU); }
            if (is_sanshoku_doukou(div, melds)) { res.han += 2; res.yaku_ids.push_back(ID_SANSHOKU_DOKO); }
            if (is_chinitsu(hand, melds)) {
                res.han += ctx.is_menzen ? 6 : 5;
                res.yaku_ids.push_back(ID_CHINITSU);
            } else if (is_honitsu(hand, melds)) {
                res.han += ctx.is_menzen ? 3 : 2;
                res.yaku_ids.push_back(ID_HONITSU);
            }
            if (is_honroutou(hand, melds)) {
                res.han += 2;
                res.yaku_ids.push_back(ID_HONROUTO);
            } else if (is_junchan(div, melds)) {
                res.han += ctx.is_menzen ? 3 : 2;
                res.yaku_ids.push_back(ID_JUNCHAN);
            } else if (is_chantai(div, melds)) {
                res.han += ctx.is_menzen ? 2 : 1;
                res.yaku_ids.push_back(ID_CHANTA);
            }
            if (res.han > best.han || (res.han == best.han && res.fu > best.fu)) best = res;
        }
    }
    return best;
}

// ---------------------------------------------------------------- hand_evaluator.rs / hand_evaluator_3p.rs
// hand_evaluator.rs:286-300
inline uint8_t get_next_tile(uint8_t t) {
    if (t < 9) return t == 8 ? 0 : t + 1;
    if (t < 18) return t == 17 ? 9 : t + 1;
    if (t < 27) return t == 26 ? 18 : t + 1;
    if (t < 31) return t == 30 ? 27 : t + 1;
    if (t == 33) return 31;
    return t + 1;
}
// hand_evaluator_3p.rs:300-311
inline uint8_t get_next_tile_sanma(uint8_t t) {
    if (t == 0) return 8;
    if (t == 8) return 0;
    if (t >= 1 && t <= 7) return t;
    if (t <= 17) return 9 + (t - 9 + 1) % 9;
    if (t <= 26) return 18 + (t - 18 + 1) % 9;
    if (t <= 30) return 27 + (t - 27 + 1) % 4;
    if (t <= 33) return 31 + (t - 31 + 1) % 3;
    return t;
}

struct HandEvaluator {
    Hand hand, full_hand;
    std::vector<Meld> melds;  // tiles converted to 34-ids
    uint8_t aka_dora_count = 0;
    bool sanma = false;

    // hand_evaluator.rs:24-75
    HandEvaluator(const std::vector<uint8_t>& tiles_136, const std::vector<Meld>& in_melds, bool sanma_ = false)
        : sanma(sanma_) {
        for (uint8_t t : tiles_136) {
            if (t == 16 || t == 52 || t == 88) aka_dora_count++;
            full_hand.add(t / 4);
        }
        hand = full_hand;
        for (const Meld& m : in_melds) {
            Meld nm = m;
            if (meld_is_kan(nm)) {
                uint8_t t34 = nm.tiles[0] / 4;
                if (hand.counts[t34] == 4) hand.counts[t34] = 3;
            }
            std::vector<uint8_t> t34s;
            for (uint8_t t : nm.tiles) {
                if (t == 16 || t == 52 || t == 88) aka_dora_count++;
                t34s.push_back(t / 4);
                full_hand.add(t / 4);
            }
            nm.tiles = t34s;
            if (nm.meld_type == MT_CHI) std::sort(nm.tiles.begin(), nm.tiles.end());
            melds.push_back(nm);
        }
    }

    int current_total() const { return hand.total() + (int)melds.size() * 3; }

    // hand_evaluator.rs:77-176
    WinResult calc(uint8_t win_tile_136, const std::vector<uint8_t>& dora_ind, const std::vector<uint8_t>& ura_ind,
                   const Conditions& cond) const {
        uint8_t win34 = win_tile_136 / 4;
        Hand hand14 = hand, full14 = full_hand;
        int total = current_total();
        if (total == 13) {
            hand14.add(win34);
            full14.add(win34);
        }
        WinResult wr;
        if (!is_agari(hand14)) return wr;
        uint8_t dora = 0, ura = 0;
        for (uint8_t ind : dora_ind) {
            uint8_t nt = sanma ? get_next_tile_sanma(ind / 4) : get_next_tile(ind / 4);
            dora += full14.counts[nt];
            if (sanma && nt == 30) dora += cond.kita_count;
        }
        for (uint8_t ind : ura_ind) {
            uint8_t nt = sanma ? get_next_tile_sanma(ind / 4) : get_next_tile(ind / 4);
            ura += full14.counts[nt];
            if (sanma && nt == 30) ura += cond.kita_count;
        }
        uint8_t aka = aka_dora_count;
        if (total == 13 && (win_tile_136 == 16 || win_tile_136 == 52 || win_tile_136 == 88)) aka++;
        YakuContext ctx;
        ctx.is_tsumo = cond.tsumo;
        ctx.is_reach = cond.riichi;
        ctx.is_daburu_reach = cond.double_riichi;
        ctx.is_ippatsu = cond.ippatsu;
        ctx.is_haitei = cond.haitei;
        ctx.is_houtei = cond.houtei;
        ctx.is_rinshan = cond.rinshan;
        ctx.is_chankan = cond.chankan;
        ctx.is_tsumo_first_turn = cond.tsumo_first_turn;
        ctx.dora_count = dora;
        ctx.aka_dora = aka;
        ctx.ura_dora_count = ura;
        ctx.round_wind = 27 + cond.round_wind;
        ctx.seat_wind = 27 + cond.player_wind;
        ctx.is_menzen = true;
        for (auto& m : melds)
            if (m.opened) ctx.is_menzen = false;
        ctx.sanma = sanma;
        ctx.nukidora_count = sanma ? cond.kita_count : 0;
        YakuResult yr = calculate_yaku(hand14, melds, ctx, win34);
        bool is_oya = cond.player_wind == 0;
        uint8_t scoring_han = (yr.yakuman_count == 0 && yr.han >= 13) ? 13 : yr.han;
        Score sc = calculate_score(scoring_han, yr.fu, is_oya, cond.tsumo, cond.honba, sanma ? 3 : 4);
        bool has_yaku = false;
        for (uint32_t id : yr.yaku_ids)
            if (id != ID_DORA && id != ID_AKADORA && id != ID_URADORA && !(sanma && id == ID_NUKIDORA)) has_yaku = true;
        wr.is_win = (has_yaku || yr.yakuman_count > 0) && yr.han >= 1;
        wr.yakuman = yr.yakuman_count > 0;
        wr.ron_agari = sc.pay_ron;
        wr.tsumo_agari_oya = sc.pay_tsumo_oya;
        wr.tsumo_agari_ko = sc.pay_tsumo_ko;
        wr.yaku = yr.yaku_ids;
        wr.han = yr.han;
        wr.fu = yr.fu;
        wr.pao_payer = -1;
        wr.has_win_shape = true;
        return wr;
    }

    // hand_evaluator.rs:178-194
    bool is_tenpai() const {
        if (current_total() != 13) return false;
        Hand h = hand;
        for (int i = 0; i < TILE_MAX; i++)
            if (h.counts[i] < 4) {
                h.add((uint8_t)i);
                if (is_agari(h)) return true;
                h.remove((uint8_t)i);
            }
        return false;
    }
    // hand_evaluator.rs:196-213
    std::vector<uint8_t> get_waits_u8() const {
        std::vector<uint8_t> w;
        if (current_total() != 13) return w;
        Hand h = hand;
        for (int i = 0; i < TILE_MAX; i++)
            if (h.counts[i] < 4) {
                h.add((uint8_t)i);
                if (is_agari(h)) w.push_back((uint8_t)i);
                h.remove((uint8_t)i);
            }
        return w;
    }
};

// parser.rs:301-334
inline std::string tid_to_mjai(uint8_t tid) {
    if (tid == 16) return "5mr";
    if (tid == 52) return "5pr";
    if (tid == 88) return "5sr";
    int kind = tid / 36;
    if (kind < 3) {
        static const char sc[3] = {'m', 'p', 's'};
        int num = (tid % 36) / 4 + 1;
        std::string s;
        s += (char)('0' + num);
        s += sc[kind];
        return s;
    }
    int num = (tid - 108) / 4 + 1;
    static const char* honors[7] = {"E", "S", "W", "N", "P", "F", "C"};
    if (num >= 1 && num <= 7) return honors[num - 1];
    return std::to_string(num) + "z";
}

}  // namespace orc
