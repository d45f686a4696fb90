// ORACLE — TEST INFRASTRUCTURE ONLY.
// Shanten restated from riichienv-core/src/shanten.rs:163-261, 407-468.  The reference reads the
// "replacement number" from Cryolite/nyanten v? lookup blobs (src/data/nyanten_*.bin, produced by
// scripts/convert_nyanten_tables.py — a third-party table, not restated byte-for-byte).  The quantity the
// tables hold is the published definition
//     r = min over winning shapes t (m mentsu + 1 pair, every tile <= 4 copies) of sum_i max(0, t_i - c_i),
// computed here by plain enumeration of the shapes per suit (independent of the DP generator the HIP
// library uses).  Pinned by the reference's KATs: tests/test_shanten.py:4-114, README.md:245-273.
#pragma once
#include <cstdint>
#include <algorithm>
#include <cstring>
#include <map>
#include <unordered_map>
#include <vector>

namespace orc {

struct SuitCost {
    uint8_t c[5][2];  // [k mentsu][p pair]
};

inline SuitCost suit_cost_enum(const uint8_t* cnt, int n /*9 or 7*/, bool sequences) {
    SuitCost out;
    for (auto& a : out.c)
        for (auto& b : a) b = 99;
    // mentsu kinds: koutsu at 0..n-1, shuntsu at 0..n-3
    std::vector<std::vector<uint8_t>> kinds;
    for (int i = 0; i < n; i++) {
        std::vector<uint8_t> t(n, 0);
        t[i] = 3;
        kinds.push_back(t);
    }
    if (sequences)
        for (int i = 0; i + 2 < n; i++) {
            std::vector<uint8_t> t(n, 0);
            t[i] = t[i + 1] = t[i + 2] = 1;
            kinds.push_back(t);
        }
    const int K = (int)kinds.size();
    std::vector<uint8_t> t(n, 0);
    auto eval = [&](int k) {
        for (int pair = -1; pair < n; pair++) {
            int cost = 0;
            bool ok = true;
            for (int i = 0; i < n; i++) {
                int ti = t[i] + (pair == i ? 2 : 0);
                if (ti > 4) { ok = false; break; }
                if (ti > cnt[i]) cost += ti - cnt[i];
            }
            if (!ok) continue;
            uint8_t& dst = out.c[k][pair >= 0 ? 1 : 0];
            if (cost < dst) dst = (uint8_t)cost;
        }
    };
    // multisets of size k from K kinds
    struct Rec {
        std::vector<uint8_t>& t;
        const std::vector<std::vector<uint8_t>>& kinds;
        int K, n;
        decltype(eval)& ev;
        void go(int start, int k) {
            ev(k);
            if (k == 4) return;
            for (int j = start; j < K; j++) {
                bool ok = true;
                for (int i = 0; i < n; i++)
                    if (t[i] + kinds[j][i] > 4) ok = false;
                if (!ok) continue;
                for (int i = 0; i < n; i++) t[i] += kinds[j][i];
                go(j, k + 1);
                for (int i = 0; i < n; i++) t[i] -= kinds[j][i];
            }
        }
    } rec{t, kinds, K, n, eval};
    rec.go(0, 0);
    return out;
}

// The enumeration of one suit vector is memoised per thread (the ukeire walks and the golden-vector tests evaluate
// hundreds of hands that differ in one suit): same numbers, computed once per distinct vector.
inline const SuitCost& suit_cost_memo(const uint8_t* cnt, int n, bool sequences) {
    thread_local std::unordered_map<uint32_t, SuitCost> memo;
    uint32_t key = sequences ? 0u : 1u << 31;
    for (int i = 0; i < n; i++) key |= (uint32_t)(cnt[i] & 7u) << (3 * i);
    auto it = memo.find(key);
    if (it != memo.end()) return it->second;
    if (memo.size() > 400000) memo.clear();
    return memo.emplace(key, suit_cost_enum(cnt, n, sequences)).first->second;
}

inline int calc_normal_enum(const uint8_t* tiles34, int m) {
    SuitCost s[4];
    for (int q = 0; q < 3; q++) s[q] = suit_cost_memo(tiles34 + 9 * q, 9, true);
    s[3] = suit_cost_memo(tiles34 + 27, 7, false);
    if (m > 4) m = 4;
    int best = 99;
    for (int k0 = 0; k0 <= m; k0++)
        for (int k1 = 0; k0 + k1 <= m; k1++)
            for (int k2 = 0; k0 + k1 + k2 <= m; k2++) {
                int k3 = m - k0 - k1 - k2;
                int ks[4] = {k0, k1, k2, k3};
                for (int ps = 0; ps < 4; ps++) {
                    int tot = 0;
                    for (int q = 0; q < 4; q++) tot += s[q].c[ks[q]][q == ps ? 1 : 0];
                    if (tot < best) best = tot;
                }
            }
    return best - 1;
}

// shanten.rs:198-211 / :437-452
inline int calc_chitoi(const uint8_t* t, bool sanma) {
    int pairs = 0, kinds = 0;
    for (int i = 0; i < 34; i++) {
        if (sanma && i >= 1 && i <= 7) continue;
        if (t[i] > 0) {
            kinds++;
            if (t[i] >= 2) pairs++;
        }
    }
    int red = kinds < 7 ? 7 - kinds : 0;
    return 7 - pairs + red - 1;
}
// shanten.rs:213-226
inline int calc_kokushi(const uint8_t* t) {
    static const int term[13] = {0, 8, 9, 17, 18, 26, 27, 28, 29, 30, 31, 32, 33};
    int kinds = 0;
    bool pair = false;
    for (int k = 0; k < 13; k++)
        if (t[term[k]] > 0) {
            kinds++;
            if (t[term[k]] >= 2) pair = true;
        }
    return 14 - kinds - (pair ? 1 : 0) - 1;
}
// shanten.rs:407-435
inline void relocate_3p(uint8_t* t) {
    uint8_t mc[2] = {t[0], t[8]};
    int mp[2] = {0, 8};
    t[0] = t[8] = 0;
    int slot = 27;
    for (int i = 0; i < 2; i++) {
        if (mc[i] == 0) continue;
        while (slot < 34 && t[slot] != 0) slot++;
        if (slot < 34) {
            t[slot] = mc[i];
            slot++;
        } else {
            t[mp[i]] = mc[i];
        }
    }
}
// shanten.rs:228-241 / :454-468
inline int calc_shanten_from_counts(const uint8_t* tehai, int len_div3, bool sanma) {
    uint8_t t[34];
    std::memcpy(t, tehai, 34);
    if (sanma) relocate_3p(t);
    int s = calc_normal_enum(t, len_div3);
    if (s <= 0 || len_div3 < 4) return s;
    int c = calc_chitoi(tehai, sanma);
    if (c < s) s = c;
    if (s > 0) {
        int k = calc_kokushi(tehai);
        if (k < s) s = k;
    }
    return s;
}

// calculate_shanten (shanten.rs:250-261 / :473-484) on a type histogram: len_div3 = number of tiles / 3
inline int shanten_of(const uint8_t* c, bool sanma) {
    int total = 0;
    for (int i = 0; i < 34; i++) total += c[i];
    return calc_shanten_from_counts(c, total / 3, sanma);
}
// SANMA_VALID_TILE_TYPES (shanten.rs:244-247)
inline bool ukeire_type_ok(int t, bool sanma) { return !sanma || t == 0 || t >= 8; }
// calculate_effective_tiles / _3p (shanten.rs:265-300, 488-521); hand must hold 3n+1 tiles
inline uint32_t effective_tiles(const uint8_t* c, bool sanma) {
    int cur = shanten_of(c, sanma);
    uint32_t n = 0;
    uint8_t t[34];
    std::memcpy(t, c, 34);
    for (int ty = 0; ty < 34; ty++) {
        if (!ukeire_type_ok(ty, sanma) || c[ty] >= 4) continue;
        t[ty]++;
        if (shanten_of(t, sanma) < cur) n++;
        t[ty]--;
    }
    return n;
}
// calculate_effective_tiles_with_discard / _3p_with_discard (shanten.rs:304-327, 525-548); 0xFFFFFFFF for a 3n hand
// (the reference asserts).  The loop over hand tiles collapses to a loop over held types.
inline uint32_t effective_tiles_with_discard(const uint8_t* c, bool sanma) {
    int total = 0;
    for (int i = 0; i < 34; i++) total += c[i];
    if (total % 3 == 1) return effective_tiles(c, sanma);
    if (total % 3 != 2) return 0xFFFFFFFFu;
    int sh = shanten_of(c, sanma);
    uint32_t best = 0;
    uint8_t t[34];
    std::memcpy(t, c, 34);
    for (int d = 0; d < 34; d++) {
        if (!c[d]) continue;
        t[d]--;
        if (shanten_of(t, sanma) <= sh) best = std::max(best, effective_tiles(t, sanma));
        t[d]++;
    }
    return best;
}
// calculate_best_ukeire / _3p (shanten.rs:331-405, 552-626)
inline uint32_t best_ukeire(const uint8_t* c, const uint8_t* visible, bool sanma) {
    int cur = shanten_of(c, sanma);
    uint32_t best = 0;
    uint8_t t[34];
    std::memcpy(t, c, 34);
    for (int d = 0; d < 34; d++) {
        if (!c[d]) continue;
        t[d]--;
        int nsh = shanten_of(t, sanma);
        if (nsh <= cur) {
            uint32_t uke = 0;
            for (int ty = 0; ty < 34; ty++) {
                if (!ukeire_type_ok(ty, sanma) || t[ty] >= 4) continue;
                t[ty]++;
                bool better = shanten_of(t, sanma) < nsh;
                t[ty]--;
                if (better) {
                    int rem = 4 - (int)visible[ty];
                    if (rem < 0) rem = 0;
                    rem -= (int)t[ty];
                    if (rem < 0) rem = 0;
                    uke += (uint32_t)rem;
                }
            }
            best = std::max(best, uke);
        }
        t[d]++;
    }
    return best;
}

}  // namespace orc
