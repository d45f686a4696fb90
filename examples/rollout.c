/* The C-ABI boundary from plain C (no HIP, no Python): 4 096 4p-red-half games stepped by the device RandomAgent, then the
 * scores of game 0 and the tail of its MJAI log.  Only include/riichi_mi355x.h is needed; the reference-side counterpart of
 * these calls is RiichiEnv.reset / step / mjai_log (riichienv-python/src/env.rs:799-872).
 *
 *   gcc -O2 -Iinclude examples/rollout.c -o /tmp/rollout -Lriichienv_amd -l:libriichi_mi355x.so -Wl,-rpath,$PWD/riichienv_amd
 *   /tmp/rollout            (needs an MI355X; the library has no CPU fallback and rmj_create fails loudly without one) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "riichi_mi355x.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc__ = (call);                                                           \
        if (rc__ != RMJ_OK) {                                                        \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc__, rmj_last_error());        \
            return 1;                                                                \
        }                                                                            \
    } while (0)

int main(void) {
    RmjConfig cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.n_games = 4096;
    cfg.game_mode = 2; /* 4p-red-half */
    cfg.rule_bits = RMJ_RULE_TENHOU;
    cfg.base_seed = 42;
    cfg.event_ring = 1024;
    printf("%s, %d device(s)\n", rmj_version(), rmj_device_count());
    rmj_handle h = NULL;
    CHECK(rmj_create(&cfg, &h));
    CHECK(rmj_reset(h, NULL, NULL, NULL, NULL, NULL, NULL, NULL));
    CHECK(rmj_step_random(h, /*policy_seed=*/7, /*n_steps=*/2000, /*auto_reset=*/0));
    uint64_t total = 0;
    CHECK(rmj_total_steps(h, &total));
    int32_t* scores = (int32_t*)malloc(sizeof(int32_t) * 4 * cfg.n_games);
    CHECK(rmj_get_scores(h, scores));
    printf("%llu env.step calls advanced a game; scores of game 0: %d %d %d %d\n", (unsigned long long)total, scores[0], scores[1],
           scores[2], scores[3]);
    uint32_t* counts = (uint32_t*)malloc(sizeof(uint32_t) * cfg.n_games);
    CHECK(rmj_get_event_counts(h, counts));
    RmjEvent ev[16];
    uint32_t n = 0;
    const uint32_t first = counts[0] > 8 ? counts[0] - 8 : 0;
    CHECK(rmj_get_events(h, 0, first, 16, ev, &n));
    char line[2048];
    for (uint32_t i = 0; i < n;) {
        const int used = rmj_format_event(ev + i, n - i, /*seat=*/-1, line, sizeof(line));
        if (used <= 0) break;
        puts(line);
        i += (uint32_t)used;
    }
    free(counts);
    free(scores);
    CHECK(rmj_destroy(h));
    return 0;
}
