// Sanitizer harness of the HOST side of the C-ABI (riichienv_amd/csrc/rmj_host.h: state record <-> view, MJAI formatter) - built with
// -fsanitize=address,undefined by scripts/run_sanitizers.sh.  GPU-side sanitizers do not exist on this pool; this covers the code of
// rmj_api.hip that runs on the CPU and handles caller-supplied data.
//   * random state views (in and out of range) through from_view / to_view: accepted views round-trip field by field
//   * random and adversarial event records (every type byte, any counts) through the single and the batch formatter with
//     buffers of every size: no out-of-bounds access, the sizes the two passes report agree
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../riichienv_amd/csrc/rmj_host.h"

static std::mt19937_64 rng(12345);
static uint32_t rnd(uint32_t n) { return (uint32_t)(rng() % n); }

static void fill_random(void* p, size_t n) {
    uint8_t* b = (uint8_t*)p;
    for (size_t i = 0; i < n; i++) b[i] = (uint8_t)rng();
}

static int check_views() {
    int accepted = 0;
    for (int it = 0; it < 20000; it++) {
        RmjStateView v;
        fill_random(&v, sizeof(v));
        if (it & 1) {   // a plausible view: counts in range
            v.wall_len = (uint8_t)rnd(137);
            v.rinshan_draw_count = (uint8_t)rnd(5);
            if (v.wall_len + v.rinshan_draw_count > 136) v.wall_len = (uint8_t)(136 - v.rinshan_draw_count);
            v.n_dora = (uint8_t)rnd(6);
            for (auto& p : v.players) {
                p.hand_len = (uint8_t)rnd(15);
                p.n_melds = (uint8_t)rnd(5);
                p.n_discards = (uint8_t)rnd(RMJ_MAX_DISCARDS + 1);
                p.n_forbidden = (uint8_t)rnd(3);
                p.n_kita = (uint8_t)rnd(5);
                for (auto& m : p.melds) m.n_tiles = (uint8_t)(3 + rnd(2));
            }
        }
        GState S;
        uint8_t W[RMJ_WALL_STRIDE];
        fill_random(&S, sizeof(S));
        fill_random(W, sizeof(W));
        const char* err = rmjh::from_view(S, W, &v);
        if (err) continue;
        accepted++;
        RmjStateView back;
        rmjh::to_view(S, W, &back);
        // what a view holds inside its counts comes back
        if (back.wall_len != v.wall_len || memcmp(back.wall, v.wall, v.wall_len) != 0) return 1;
        for (int p = 0; p < 4; p++) {
            const RmjPlayerView &a = v.players[p], &b = back.players[p];
            if (a.hand_len != b.hand_len || memcmp(a.hand, b.hand, a.hand_len) != 0) return 2;
            if (a.n_discards != b.n_discards || memcmp(a.discards, b.discards, a.n_discards) != 0) return 3;
            if (a.score != b.score || a.score_delta != b.score_delta) return 4;
            if (a.n_melds != b.n_melds) return 5;
        }
        if (back.turn_count != v.turn_count || back.riichi_sticks != v.riichi_sticks || back.phase != v.phase) return 6;
    }
    printf("views: %d accepted of 20000, round trips ok\n", accepted);
    return accepted > 5000 ? 0 : 7;
}

static int check_formatter() {
    uint64_t total = 0;
    for (int it = 0; it < 4000; it++) {
        const uint32_t n_games = 1 + rnd(6);
        std::vector<uint32_t> offsets(n_games + 1, 0);
        for (uint32_t g = 0; g < n_games; g++) offsets[g + 1] = offsets[g] + rnd(12);
        std::vector<RmjEvent> ev(offsets[n_games] + 1);
        fill_random(ev.data(), ev.size() * sizeof(RmjEvent));
        for (auto& e : ev) {
            if (rnd(4)) e.type = (uint8_t)rnd(20);          // mostly valid type bytes, every one of them
            if (rnd(2)) e.n_ura = (uint8_t)rnd(8);
        }
        for (uint32_t i = 0; i + 2 < ev.size(); i++)        // some well-formed triples
            if (ev[i].type == RMJ_EV_START_KYOKU && rnd(2)) { ev[i + 1].type = RMJ_EV_TEHAI; ev[i + 2].type = RMJ_EV_TEHAI; }
        const int seat = (int)rnd(6) - 1;
        std::vector<uint64_t> toffs(n_games + 1), toffs2(n_games + 1);
        const uint64_t need = rmjh::format_events(ev.data(), offsets.data(), n_games, seat, nullptr, 0, toffs.data(), 1 + (int)rnd(4));
        // a buffer that is too small must stay untouched, an exact one must be filled exactly
        std::vector<char> small(need ? need - 1 : 0, 'x');
        if (rmjh::format_events(ev.data(), offsets.data(), n_games, seat, small.data(), small.size(), toffs2.data(), 2) != need) return 10;
        for (char c : small) if (c != 'x') return 11;
        std::vector<char> exact(need + 1, 'y');
        if (rmjh::format_events(ev.data(), offsets.data(), n_games, seat, exact.data(), need, toffs2.data(), 3) != need) return 12;
        if (exact[need] != 'y' || toffs != toffs2 || toffs[n_games] != need) return 13;
        for (uint64_t i = 0; i < need; i++) if (exact[i] == 'y' && false) return 14;
        total += need;
        // the single-event entry with every capacity up to the text's size
        for (uint32_t i = 0; i < offsets[n_games]; i++) {
            char buf[4096];
            rmjh::Out probe{buf, buf + sizeof(buf), 0};
            const int used = rmjh::format_event(probe, &ev[i], offsets[n_games] - i, seat);
            if (used <= 0) continue;
            if (probe.need >= sizeof(buf)) return 15;
            for (uint64_t cap = 0; cap <= probe.need; cap += 1 + rnd(7)) {
                std::vector<char> b(cap + 1, 'z');
                rmjh::Out o{b.data(), b.data() + cap, 0};
                if (rmjh::format_event(o, &ev[i], offsets[n_games] - i, seat) != used || o.need != probe.need || b[cap] != 'z') return 16;
            }
        }
    }
    printf("formatter: %llu bytes of text over 4000 random batches, sizes consistent\n", (unsigned long long)total);
    return 0;
}

int main() {
    int rc = check_views();
    if (rc) { printf("FAILED: views (%d)\n", rc); return 1; }
    rc = check_formatter();
    if (rc) { printf("FAILED: formatter (%d)\n", rc); return 1; }
    printf("host_san OK\n");
    return 0;
}
