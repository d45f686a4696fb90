/*
 * riichi_mi355x.h — C-ABI of the MI355X-native batched Riichi Mahjong step path.
 *
 * This is the drop-in boundary for the hot path of smly/RiichiEnv:
 *   RiichiEnv.__new__/reset/step/get_observations/done/scores/ranks/mjai_log
 *       (reference: riichienv-python/src/env.rs:82-118, 799-851, 857-872, 741-765,
 *        353-356, 401-404, 673-689, 729-739)
 *   Observation.legal_actions()/mask()          (observation/python.rs:93-111)
 *   HandEvaluator.calc / get_waits / is_tenpai  (hand_evaluator.rs:77-213)
 *   calculate_score                             (score.rs:13-52)
 *
 * One handle = one shard of independent games on one GPU.  All entry points are
 * plain C: opaque handle, caller-allocated arrays, int return code (0 = OK,
 * negative = RMJ_ERR_*).  No torch / C++ types cross this boundary.  A handle is
 * thread-compatible (use one handle per host thread / per GPU).
 *
 * The library has NO CPU fallback: every compute entry point returns
 * RMJ_ERR_NO_DEVICE when no HIP device is usable.
 */
#ifndef RIICHI_MI355X_H
#define RIICHI_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ constants */
#define RMJ_OK 0
#define RMJ_ERR_ARG (-1)       /* bad argument (ValueError in the reference binding) */
#define RMJ_ERR_NO_DEVICE (-2) /* no usable HIP device / HIP runtime failure at init */
#define RMJ_ERR_HIP (-3)       /* HIP runtime error (see rmj_last_error) */
#define RMJ_ERR_RANGE (-4)     /* index out of range */

#define RMJ_NP 4                /* seats in the 4-player state layout */
#define RMJ_MAX_LEGAL 64        /* max legal actions per seat (reference worst case ~56) */
#define RMJ_ACTION_SPACE_4P 82  /* action.rs:11 */
#define RMJ_ACTION_SPACE_3P 60  /* action.rs:12 */
#define RMJ_MAX_DISCARDS 32
#define RMJ_WALL_4P 136

/* ActionType, action.rs:55-68 */
enum {
    RMJ_DISCARD = 0, RMJ_CHI = 1, RMJ_PON = 2, RMJ_DAIMINKAN = 3, RMJ_RON = 4, RMJ_RIICHI = 5,
    RMJ_TSUMO = 6, RMJ_PASS = 7, RMJ_ANKAN = 8, RMJ_KAKAN = 9, RMJ_KYUSHU = 10, RMJ_KITA = 11
};
/* Phase, action.rs:29-33 */
enum { RMJ_WAIT_ACT = 0, RMJ_WAIT_RESPONSE = 1 };
/* MeldType, types.rs:54-62 */
enum { RMJ_MELD_CHI = 0, RMJ_MELD_PON = 1, RMJ_MELD_DAIMINKAN = 2, RMJ_MELD_ANKAN = 3, RMJ_MELD_KAKAN = 4 };

/* GameRule bits, rule.rs:10-22 (bit i = i-th field) */
#define RMJ_RULE_RON_ON_ANKAN_KOKUSHI 1u
#define RMJ_RULE_KOKUSHI13_DOUBLE 2u
#define RMJ_RULE_SUUANKOU_TANKI_DOUBLE 4u
#define RMJ_RULE_JUNSEI_CHUUREN_DOUBLE 8u
#define RMJ_RULE_DAISUUSHII_DOUBLE 16u
#define RMJ_RULE_PAO_LIABILITY_ONLY 32u
#define RMJ_RULE_SANCHAHO_DRAW 64u
#define RMJ_RULE_KUIKAE_FORBIDDEN 128u
/* Not a GameRule field: seed -> wall as the reference's crates define it (state/wall.rs:36-56: StdRng::seed_from_u64 =
 * PCG32 seed expansion + ChaCha12, SliceRandom::shuffle, salt = next_u64, wall_digest = SHA-256(salt || wall); see
 * rmj_get_wall_digest).  Without the bit the wall of a seed is the build's own counter-based permutation,
 * which costs nothing per round; with it a round start pays one serial Fisher-Yates pass.
 * STATUS OF THE CLAIM: the chain is restated from the published algorithms (rand 0.9 / rand_core 0.9 / chacha20 / sha2); ChaCha, StdRng's
 * construction and SHA-256 are pinned on published vectors, seed_from_u64's PCG32 expansion and the index draws of shuffle on nothing outside
 * this repository - equality of walls and digests with the Rust crates (rand 0.10 in the reference's Cargo.lock) is UNVERIFIED until one real
 * (seed -> wall, salt, digest) vector printed by the reference is pinned in tests/golden/ref_rng_vectors.json (INTEGRATION.md has the Rust
 * program; tests/test_oracle_ref_rng.py and tests/test_gpu_ref_rng.py consume it, and report xfail while it is absent). */
#define RMJ_RULE_REFERENCE_RNG 256u
#define RMJ_RULE_TENHOU (RMJ_RULE_SANCHAHO_DRAW | RMJ_RULE_KUIKAE_FORBIDDEN)          /* rule.rs:31-44 */
#define RMJ_RULE_MJSOUL (1u | 2u | 4u | 8u | 16u | 32u | RMJ_RULE_KUIKAE_FORBIDDEN) /* rule.rs:46-57 */

/*
 * Packed action (u64), used for step() input and for the legal-action lists:
 *   bits  0..7   action type (RMJ_*), 0xFF = "no action from this seat"
 *   bits  8..15  tile (136-id), 0xFF = None
 *   bits 16..23  number of consume tiles (0..4)
 *   bits 24..55  consume tiles c0..c3 (ascending, as Action::new sorts them; action.rs:97-98)
 */
typedef uint64_t rmj_action_t;
#define RMJ_NO_ACTION 0xFFFFFFFFFFFFFFFFull
#define RMJ_TILE_NONE 0xFFu

/* MJAI event record emitted by the device; formatted to the reference's JSON
 * strings (alphabetical keys, state/mod.rs:2094-2148) by rmj_format_event. */
enum {
    RMJ_EV_NONE = 0, RMJ_EV_START_GAME = 1, RMJ_EV_START_KYOKU = 2, RMJ_EV_TSUMO = 3, RMJ_EV_DAHAI = 4,
    RMJ_EV_REACH = 5, RMJ_EV_REACH_ACCEPTED = 6, RMJ_EV_CHI = 7, RMJ_EV_PON = 8, RMJ_EV_DAIMINKAN = 9,
    RMJ_EV_ANKAN = 10, RMJ_EV_KAKAN = 11, RMJ_EV_DORA = 12, RMJ_EV_HORA = 13, RMJ_EV_RYUKYOKU = 14,
    RMJ_EV_END_KYOKU = 15, RMJ_EV_END_GAME = 16, RMJ_EV_KITA = 17,
    RMJ_EV_TEHAI = 18 /* continuation of START_KYOKU: payload[0..25] = 26 hand tiles (2 seats) */
};
/* ryukyoku reasons (state/mod.rs:1846-1968) */
enum {
    RMJ_RK_EXHAUSTIVE = 0, RMJ_RK_NAGASHI = 1, RMJ_RK_KYUSHU = 2, RMJ_RK_SUFUURENTA = 3, RMJ_RK_SUUKANSANSEN = 4,
    RMJ_RK_SUUCHA_RIICHI = 5, RMJ_RK_SANCHAHO = 6, RMJ_RK_ILLEGAL = 7 /* + actor = offender */
};
typedef struct RmjEvent { /* 32 bytes */
    uint8_t type;        /* RMJ_EV_* */
    uint8_t actor;       /* actor / oya (start_kyoku) / offender (illegal ryukyoku) */
    uint8_t target;      /* target / kyoku number (start_kyoku) */
    uint8_t tile;        /* pai / dora_marker */
    uint8_t consumed[4]; /* consumed tiles; start_kyoku: [bakaze, honba, kyotaku_lo, kyotaku_hi] */
    int32_t deltas[4];   /* hora/ryukyoku deltas; start_kyoku: scores */
    uint8_t flags;       /* dahai: tsumogiri; hora: is_tsumo; ryukyoku: reason; n_consumed for melds in bits 4..7 */
    uint8_t n_ura;
    uint8_t ura[5];
    uint8_t pad;
} RmjEvent;

/* ------------------------------------------------------------------ state peek/poke view
 * Mirrors GameState/PlayerState/WallState (state/mod.rs:31-91, state/player.rs:6-39,
 * state/wall.rs:8-19).  Used by tests (the reference's Python setters, env.rs:134-622)
 * and by parity checks: the oracle fills the same struct. */
typedef struct RmjMeldView {
    uint8_t meld_type, n_tiles, tiles[4], opened;
    int8_t from_who;
    int16_t called_tile; /* -1 = None */
} RmjMeldView;

typedef struct RmjPlayerView {
    uint8_t hand_len, hand[14];
    uint8_t n_melds;
    RmjMeldView melds[4];
    uint8_t n_discards, discards[RMJ_MAX_DISCARDS];
    uint32_t discard_from_hand_bits, discard_is_riichi_bits;
    int8_t riichi_declaration_index; /* -1 = None */
    int32_t score, score_delta;
    uint8_t riichi_declared, riichi_stage, double_riichi_declared, missed_agari_riichi, missed_agari_doujun,
        nagashi_eligible, ippatsu_cycle;
    int8_t pao_daisangen, pao_daisuushi; /* liable seat for yaku 37 / 50, -1 = none */
    uint8_t n_forbidden, forbidden[2];
    int16_t riichi_sutehai, last_tedashi; /* -1 = None */
    uint8_t n_kita, kita[4];              /* 3P: kita_tiles (state_3p/player.rs:38) */
} RmjPlayerView;

typedef struct RmjStateView {
    uint8_t wall_len, wall[RMJ_WALL_4P]; /* WallState.tiles (after reverse; draw = pop from end) */
    uint8_t n_dora, dora[5];
    uint8_t rinshan_draw_count, pending_kan_dora_count, drawable_count;
    uint64_t wall_seed, hand_index;
    RmjPlayerView players[RMJ_NP];
    uint8_t current_player, is_done, needs_tsumo, phase, active_mask;
    uint32_t turn_count, riichi_sticks;
    int16_t last_discard_pid, last_discard_tile; /* -1 = None */
    int16_t pending_kan_pid;                     /* -1 = None */
    rmj_action_t pending_kan_action;
    uint8_t oya, honba, kyoku_idx, round_wind, is_rinshan_flag, is_first_turn;
    int16_t riichi_pending_acceptance, drawn_tile; /* -1 = None */
    int16_t last_error_pid;                        /* -1 = no error (quirk Q9) */
} RmjStateView;

/* ------------------------------------------------------------------ configuration */
typedef struct RmjConfig {
    uint32_t n_games;     /* games in this shard */
    uint8_t game_mode;    /* 0..2 = 4p-red-{single,east,half}; 3..5 = 3p (env.rs:93-100) */
    uint8_t skip_mjai_logging;
    uint8_t round_wind;   /* constructor round_wind (env.rs:113) */
    uint8_t reserved0;
    uint32_t rule_bits;   /* RMJ_RULE_* */
    int32_t device;       /* HIP device ordinal */
    uint64_t base_seed;   /* episode seed of game g = splitmix64(base_seed + game_offset + g) unless `seeds` given (consecutive
                             seeds would make game g's k-th hand deal game g+k's first wall, state/wall.rs:38) */
    uint64_t game_offset; /* global index of this shard's first game (multi-GPU sharding by index) */
    const uint64_t* seeds;/* optional [n_games] explicit episode seeds (RiichiEnv(seed=...)) */
    uint32_t event_ring;  /* per-game MJAI event ring capacity (power of two, >= 64) */
    uint32_t reserved1;
} RmjConfig;

typedef struct rmj_env* rmj_handle;

/* ------------------------------------------------------------------ lifecycle */
const char* rmj_version(void);
const char* rmj_last_error(void);
int rmj_device_count(void);
/* RiichiEnv.__new__ (env.rs:82-118): allocates SoA state for n_games, runs the constructor's
 * own _initialize_round (state/mod.rs:165) for every game. */
int rmj_create(const RmjConfig* cfg, rmj_handle* out);
int rmj_destroy(rmj_handle h);

/* RiichiEnv.reset (env.rs:799-851) for the games selected by `select` (NULL = all).
 * Optional per-game arrays (NULL = reference defaults): walls [n][136] in the reference's
 * `wall=` orientation, oya [n], round_wind [n], scores [n][4], honba [n], kyotaku [n]. */
int rmj_reset(rmj_handle h, const uint8_t* select, const uint8_t* walls, const uint8_t* oya, const uint8_t* round_wind,
              const int32_t* scores, const uint8_t* honba, const uint32_t* kyotaku);

/* RiichiEnv.step (env.rs:857-872): actions[n][4] packed, RMJ_NO_ACTION for seats that do not act.
 * Games that are done are left untouched (state/mod.rs:331-333). */
/* RiichiEnv.clone / __copy__ / __deepcopy__ (riichienv-python/src/env.rs:358-372) for the whole batch: a new handle on the same
 * device with the same configuration whose games are in exactly the state of `h`'s - records, walls, published lists / masks /
 * waits / status, event rings, win results (device-to-device copies of the slabs; SURVEY section 5, checkpoint / resume). */
int rmj_clone(rmj_handle h, rmj_handle* out);
/* The complete state of game src_idx[i] of `src` copied into game dst_idx[i] of `dst` (same device, player count and event ring
 * size; dst may be src when the destination games are not among the source games; destination indices distinct): forks for a
 * tree search, a pool of saved positions, refilling slots.  Host index arrays. */
int rmj_copy_games(rmj_handle dst, const uint32_t* dst_idx, rmj_handle src, const uint32_t* src_idx, uint32_t n);
/* Same with the index arrays on the device (a tree search that lives on the GPU): asynchronous on dst's stream; pairs with an index
 * out of range are skipped. */
int rmj_copy_games_device(rmj_handle dst, const uint32_t* d_dst_idx, rmj_handle src, const uint32_t* d_src_idx, uint32_t n);
int rmj_step(rmj_handle h, const rmj_action_t* actions);
/* Same, `actions` is a device pointer (zero-copy from a GPU policy). */
int rmj_step_device(rmj_handle h, const rmj_action_t* d_actions);
/* Device-side uniform-random policy (RandomAgent, src/riichienv/agents/random_agent.py:6-15,
 * keyed per (game, step, seat) — SURVEY §8(c)): choice = mulhi(key32(policy_seed, global_game, step_no, seat), n_legal) (the key: see rmj_step_greedy below)
 * over the ordered legal list.  Runs n_steps batched steps; with auto_reset != 0 a finished game is
 * re-`reset()` (defaults) at the start of the next step instead of stepping.
 * Games are independent, so a rollout of >= 2 steps needs no synchronisation between the steps of different games: it
 * is issued as ONE launch in which every wavefront keeps its four games' records in LDS and steps them n_steps times
 * (kernel k_step4<true>, four games per wavefront); a game whose discard draws claims answers them in the same pass of the
 * wave, so its four games run a step or two ahead of each other inside the launch.  Every game is stepped exactly n_steps times
 * and what the launch leaves behind - records, legal lists up to their counts, masks, waits, status, events - is what n_steps
 * launches of one step leave (entries of the list slab behind a seat's count are unspecified leftovers).  A rollout of >= 32 steps of a batch between one and eight chip-fulls of wavefronts is handed out in pieces instead: a grid
 * that fits the chip once pulls (quad, chunk of up to 64 steps) tickets from per-XCD queues (kernel k_step4_queue; a quad's chunks stay on one XCD, whose L2
 * carries the record from one wavefront to the next), so a batch that is not a whole multiple of the chip's wave slots leaves no
 * half-empty tail (65 536 games: +10 %); RMJ_QUEUE_CHUNK at create sets the chunk length, 0 switches the tickets off.  rmj_set_rollout_streams(h, 1) (or RMJ_STEP_STREAMS=1) makes every step its own launch on the handle's stream - what
 * a policy that is a barrier between steps gets; RMJ_STEP4=1 / 0 in the environment at create selects the earlier
 * schedules (one launch per step and part on up to four streams; one game per wavefront). */
int rmj_step_random(rmj_handle h, uint64_t policy_seed, uint32_t n_steps, int auto_reset);
/* The device policy that PLAYS mahjong (what the consumer of this path runs is a learned policy that wins,
 * riichienv-ml/src/riichienv_ml/trainers/_ppo_worker.py:147-239; the uniform RandomAgent wins once in ~250 rounds): for every
 * seat that is to act, over its ordered legal list, the first entry of the best class
 *   Tsumo / Ron > Kita > Riichi > Ankan > Kakan > Daiminkan > [Pon > Chi, only when (key >> 24) < call_rate_256] >
 *   Discard > Pass > Kyushu kyuhai
 * (Kita before Riichi: the 3P reference offers Kita in the riichi stage and can leave the seat without a legal action),
 * and among the Discard entries (when there are two or more) the one whose removal leaves the concealed hand with the lowest
 * shanten (calculate_shanten / _3p, shanten.rs:228-241 / :454-468, of the remaining tiles with len_div3 = (hand_len - 1) / 3),
 * ties broken by mulhi(key * 0x9E3779B1, #ties) in list order; key = the RandomAgent's 32-bit key of (game, step_no, seat): fmix32 of
 * (lo(gs) ^ (4 * step_no + seat) * 0x9E3779B1) + hi(gs) with gs = splitmix64(policy_seed + global game) - round 6; the RandomAgent takes
 * list entry mulhi(key, n).  Scheduling, auto_reset and outputs exactly like rmj_step_random (same kernels, compiled with this
 * policy in place of the random pick).  Needs the four-games-per-wave kernels (the default; RMJ_STEP4=0 -> RMJ_ERR_ARG). */
int rmj_step_greedy(rmj_handle h, uint64_t policy_seed, uint32_t n_steps, int auto_reset, uint32_t call_rate_256);
/* Fill actions[n][4] with what the device policy would choose for the CURRENT state (no step). */
int rmj_random_actions(rmj_handle h, uint64_t policy_seed, rmj_action_t* actions);
/* Same into a device buffer [n][4], asynchronous on the handle's stream (feeds rmj_step_device without a host trip). */
int rmj_random_actions_device(rmj_handle h, uint64_t policy_seed, rmj_action_t* d_actions);

/* ------------------------------------------------------------------ observations */
int rmj_get_status(rmj_handle h, uint8_t* active_mask, uint8_t* phase, uint8_t* done); /* each [n] */
int rmj_get_legal(rmj_handle h, rmj_action_t* legal /*[n][4][64]*/, uint8_t* counts /*[n][4]*/);
/* The lists a host agent loop reads per step without the [n][4][64] slab (2 KB per game): one row per seat that is to act, in
 * (game, seat) order: index[row] = game * 4 + seat, list = entries[offsets[row] .. offsets[row + 1]) (what Observation.legal_actions()
 * returns for that seat, observation/mod.rs:93-111).  Gathered on the device, copied through pinned staging memory owned by the
 * handle (~110 B per game).  *n_rows / *n_entries: totals of this state; when they exceed cap_rows / cap_entries only the first
 * cap_rows rows / cap_entries entries were written.  offsets has cap_rows + 1 slots. */
int rmj_get_legal_compact(rmj_handle h, uint32_t* index, uint32_t* offsets, rmj_action_t* entries, uint32_t cap_rows, uint32_t cap_entries,
                          uint32_t* n_rows, uint32_t* n_entries);
int rmj_get_mask(rmj_handle h, uint8_t* mask /*[n][4][82]*/);
int rmj_get_waits(rmj_handle h, uint64_t* waits /*[n][4] bit t = tile type t; 0 for seats without an observation (not active, env.rs:870-871) */);
int rmj_get_scores(rmj_handle h, int32_t* scores /*[n][4]*/);
int rmj_get_ranks(rmj_handle h, uint8_t* ranks /*[n][4], 1-based, ties by seat (env.rs:673-689)*/);
int rmj_get_step_counts(rmj_handle h, uint64_t* steps /*[n]*/);
int rmj_total_steps(rmj_handle h, uint64_t* total);
/* state.wall.salt / state.wall.wall_digest (riichienv-core/src/state/wall.rs:15-16, 48-55; state_3p/wall.rs:52-53, 91-99): the salt as 16
 * hex digits and SHA-256(salt || wall before the reversal) as 64, NUL-terminated.  Set by a seeded shuffle under RMJ_RULE_REFERENCE_RNG, left
 * alone by an injected wall (load_wall), cleared by a start_kyoku event (event_handler.rs:81-82); empty strings otherwise.  The digest is
 * computed on the device when asked for (salt and wall are state; the hash is a function of them). */
int rmj_get_wall_digest(rmj_handle h, uint32_t game, char* salt /*[17]*/, char* digest /*[65]*/);
int rmj_get_wall_digests(rmj_handle h, uint32_t first, uint32_t n, char* salts /*[n][17]*/, char* digests /*[n][65]*/);
int rmj_peek_state(rmj_handle h, uint32_t game, RmjStateView* out);
/* The observation outputs of ONE game (get_observations of its acting seats, env.rs:741-765): legal [4][64] + counts [4],
 * mask [4][82], waits [4], status = active_mask | phase << 8 | is_done << 16.  For sampled checks of large batches. */
int rmj_peek_outputs(rmj_handle h, uint32_t game, rmj_action_t* legal, uint8_t* counts, uint8_t* mask, uint64_t* waits,
                     uint32_t* status);
int rmj_poke_state(rmj_handle h, uint32_t game, const RmjStateView* in); /* recomputes legal actions */

/* RiichiEnv.win_results (env.rs:606-607; state/mod.rs:60, 863, 1107, 1729): the WinResult of every seat that won the
 * current round, with its pao payer.  The reference clears the map whenever a round is initialised, so it is non-empty
 * only while the game is over (the round that ended it was won).  out[4] is indexed by seat; *seat_mask tells which
 * entries are set. */
typedef struct RmjWinResult { /* WinResult, types.rs:282-293 */
    uint8_t is_win, yakuman, has_win_shape, n_yaku;
    uint8_t yaku[20];
    uint32_t han, fu, ron_agari, tsumo_agari_oya, tsumo_agari_ko;
    int8_t pao_payer; /* -1 = None */
    uint8_t pad[3];
} RmjWinResult;
int rmj_get_win_results(rmj_handle h, uint32_t game, RmjWinResult* out /*[4]*/, uint8_t* seat_mask);

/* MJAI events: records of the CURRENT game's log per slot (len(mjai_log): a reset / auto-reset starts it again, state/mod.rs:171-187),
 * and a window of them (`first` counts from the current game's first record). */
int rmj_get_event_counts(rmj_handle h, uint32_t* counts /*[n]*/);
int rmj_get_events(rmj_handle h, uint32_t game, uint32_t first, uint32_t max_events, RmjEvent* out, uint32_t* n_out);
/* Formats one event (START_KYOKU consumes the 2 TEHAI continuation records that follow it; returns
 * the number of records consumed, or <0).  seat = -1 -> full log string, 0..3 -> per-seat masked view. */
int rmj_format_event(const RmjEvent* ev, uint32_t n_avail, int seat, char* buf, uint32_t cap);
/* The logs of MANY games at once (RiichiEnv.mjai_log / the per-seat logs of every env, riichienv-python/src/env.rs:729-739,
 * state/mod.rs:2094-2148).
 * Every game SLOT writes one record stream: its position (the number of records the slot has emitted since rmj_create) never goes
 * back - a restart by auto-reset / rmj_reset / a start_game event only moves the position at which the current game's log begins
 * (rmj_get_log_positions: base) - and record i sits in ring slot i & (ring - 1) whichever game wrote it.  Cursors are such positions,
 * so a cursor stays valid across restarts and a window may hold the end of one game and the start of the next.
 * rmj_drain_events: the records every slot wrote since cursor[g] (0 = from the start of the stream), gathered on the device into
 * one dense buffer and copied down once: slot g's records are out[offsets[g] .. offsets[g + 1]) (offsets has n + 1 slots),
 * cursor[g] becomes the slot's position.  A slot whose ring was lapped since its cursor has lost its oldest records: the window
 * then starts at the oldest one still there and the loss is added to RmjEventViews.lost[g] - by the call that hands the window
 * over, not by a failed call, and not under RMJ_DRAIN_PEEK (cursors are then input only: a look at the rings).  When more than
 * cap_events records are waiting nothing is drained, *n_events holds the number, the result is RMJ_ERR_RANGE.
 * rmj_format_events: the strings of such a buffer, formatted by a pool of host threads: game g's log - its events' strings, each
 * followed by '\n' - is buf[text_offsets[g] .. text_offsets[g + 1]); *needed = bytes of all logs; RMJ_ERR_RANGE (nothing written) when
 * cap is smaller.  seat as in rmj_format_event.
 * rmj_drain_format: both in one call through pinned staging owned by the handle (no intermediate copy); ms, when given, receives the
 * milliseconds of the device gather, the copy to the host and the formatting.  A size call (buf = NULL: RMJ_ERR_RANGE, *needed set)
 * keeps what it gathered; the call that follows with the same cursors and seat only formats (the drain is "as of the size call").
 * rmj_get_log_positions: per slot, where the current game's log begins (base) and the stream position (pos); either may be NULL. */
#define RMJ_DRAIN_PEEK 1u
int rmj_get_log_positions(rmj_handle h, uint32_t* base /*[n]*/, uint32_t* pos /*[n]*/);
int rmj_drain_events(rmj_handle h, uint32_t* cursor /*[n] in/out*/, RmjEvent* out, uint32_t cap_events, uint32_t* offsets /*[n + 1]*/, uint32_t* n_events,
                     uint32_t flags);
int rmj_format_events(const RmjEvent* ev, const uint32_t* offsets, uint32_t n_games, int seat, char* buf, uint64_t cap, uint64_t* text_offsets /*[n + 1]*/,
                      uint64_t* needed);
int rmj_drain_format(rmj_handle h, uint32_t* cursor /*[n] in/out*/, int seat, char* buf, uint64_t cap, uint64_t* text_offsets /*[n + 1]*/, uint64_t* needed,
                     uint32_t* n_events, double* ms /*[3] or NULL*/, uint32_t flags);
/* Device views of the event stream for a consumer on the same GPU: slot g's record i (a stream position, i < count) sits at
 * events[g * ring + (i & (ring - 1))] while count - i <= ring; count = *(const uint32_t*)((const char*)ev_count + g * ev_count_stride);
 * the current game's log begins at position *(const uint32_t*)((const char*)ev_base + g * ev_count_stride). */
typedef struct RmjEventViews {
    uint32_t n_games, ring;
    const RmjEvent* events;      /* [n][ring] */
    const uint32_t* ev_count;    /* first game's record count; the others follow at ev_count_stride bytes */
    uint32_t ev_count_stride, reserved;
    const uint32_t* lost;        /* [n] records lost to a late drain, cumulative */
    const uint32_t* ev_base;     /* first slot's log base; the others follow at ev_count_stride bytes */
} RmjEventViews;
int rmj_event_views(rmj_handle h, RmjEventViews* out);
int rmj_get_events_lost(rmj_handle h, uint32_t* lost /*[n]*/); /* host copy of RmjEventViews.lost */

/* ------------------------------------------------------------------ batched hand math (kernel gate) */
typedef struct RmjHandCase {
    uint8_t n_tiles, tiles[14];
    uint8_t n_melds;
    RmjMeldView melds[4];
    uint8_t win_tile;
    uint8_t n_dora, dora[5], n_ura, ura[5];
    /* Conditions, types.rs:193-210 */
    uint8_t tsumo, riichi, double_riichi, ippatsu, haitei, houtei, rinshan, chankan, tsumo_first_turn;
    uint8_t player_wind, round_wind, kita_count, is_sanma;
    uint32_t honba;
} RmjHandCase;
typedef struct RmjHandResult { /* WinResult, types.rs:282-293 */
    uint8_t is_win, yakuman, has_win_shape, n_yaku;
    uint8_t yaku[20];
    uint32_t han, fu, ron_agari, tsumo_agari_oya, tsumo_agari_ko;
    uint64_t waits; /* HandEvaluator.get_waits of the tiles/melds (13-tile hands), bit per type */
    uint8_t is_tenpai, is_agari, pad[6];
} RmjHandResult;
int rmj_eval_hands(int device, const RmjHandCase* cases, uint32_t n, RmjHandResult* out);
/* agari.rs:65-73 / hand_evaluator.rs:178-213 over raw 34-histograms */
int rmj_agari_counts(int device, const uint8_t* counts /*[n][34]*/, uint32_t n, uint8_t* is_agari, uint8_t* is_tenpai,
                     uint64_t* waits);
/* score.rs:13-52 */
int rmj_calculate_score(int device, const uint8_t* han, const uint8_t* fu, const uint8_t* is_oya, const uint8_t* is_tsumo,
                        const uint32_t* honba, const uint8_t* num_players, uint32_t n, uint32_t* out /*[n][4] total,ron,oya,ko*/);

/* Observation.encode() (observation/python.rs:457-806, docs/FEATURE_ENCODING.md): 74 x 34 f32, channel-major, for
 * every seat of every game: out[n][4][74][34].  only_active != 0 -> seats that are not to act get zeros. */
#define RMJ_ENC_CHANNELS 74
#define RMJ_ENC_WIDTH_4P 34
#define RMJ_ENC_WIDTH_3P 27 /* 3P: out[n][4][74][27], compact tile index (observation_3p/helpers.rs:3-15) */
/* only_active: 0 = every seat, 1 = acting seats (rows of the others are zeroed), 2 = acting seats, rows of the others
 * are left untouched (no HBM traffic for them; meant for resident device buffers) */
int rmj_encode(rmj_handle h, int only_active, float* out);
/* Row stride of the outputs of the BASE encoder (this function, rmj_encode_device, rmj_encode_compact_device, rmj_step_random_encode,
 * rmj_step_ids_encode_device): every (game, seat) row of 74 x W floats starts `floats` floats after the previous one (out[n][4][floats],
 * compact: out[capacity][floats]); 0 = dense (74 x W, the default).  Rows padded to a multiple of 256 B - 2 048 floats in 3P, 2 560 in
 * 4P - are written at 1.3-1.4 x the rate of the unaligned dense rows (the acting seats' rows are one row in four of the tensor; DESIGN.md
 * section 11.7); the pad floats are never written.  `floats` must be even and >= 74 x W. */
int rmj_set_encode_row_stride(rmj_handle h, uint32_t floats);
int rmj_encode_device(rmj_handle h, int only_active, float* d_out); /* device pointer, asynchronous on the handle's stream */
/* Device-policy rollout WITH feature output (BASELINE configs[4]): n_steps x (one step of every game, then encode() of the
 * seats that are to act into the resident tensor d_out).  Same results as calling rmj_step_random(h, seed, 1, auto_reset)
 * and rmj_encode_device(h, only_active, d_out) n_steps times; issued like rmj_step_random as up to four parts of the batch on
 * as many HIP streams, each part running step, encode, step, encode ... in order. */
int rmj_step_random_encode(rmj_handle h, uint64_t policy_seed, uint32_t n_steps, int auto_reset, int only_active, float* d_out);
/* The batch a trainer stacks from the reference's `{pid: obs.encode() for pid, obs in env.step(...).items()}`
 * (riichienv-python/src/env.rs:857-872, observation/python.rs encode): the Observation.encode() tensors of the ACTING seats
 * only, one after the other in (game, seat) order.  d_out: [capacity][74][34 | 27] f32, d_index: [capacity] i32 = game * 4 + seat
 * of every row, *d_count (device u32): the number of observations of this state (finished games have none); when it exceeds
 * `capacity` only the first `capacity` rows were written.  Dense rows instead of 1 row in 4 of the [n_games][4] tensor:
 * the same bytes leave at 1.6x the rate.  Asynchronous on the handle's stream. */
int rmj_encode_compact_device(rmj_handle h, float* d_out, int32_t* d_index, uint32_t capacity, uint32_t* d_count);
/* rmj_step_random(h, seed, 1, auto_reset) + rmj_encode_compact_device n_steps times (one stream): BASELINE configs[4]. */
int rmj_step_random_encode_compact(rmj_handle h, uint64_t policy_seed, uint32_t n_steps, int auto_reset, float* d_out, int32_t* d_index,
                                   uint32_t capacity, uint32_t* d_count);

/* Observation.encode_extended (observation/python.rs:1271-1296): 215 channels = encode() + discard decay (4), shanten
 * efficiency (16), ankan (4), fuuro (80), action availability (11), discard candidates (5), pass context (3), last
 * tedashis (9), riichi sutehais (9); observation/encode.rs:293-585, observation_3p/encode.rs:315-615.
 * out[n][4][215][34] (3P: [n][4][215][27], seat 3 zero). */
#define RMJ_ENC_EXT_CHANNELS 215
int rmj_encode_extended(rmj_handle h, int only_active, float* out);
int rmj_encode_extended_device(rmj_handle h, int only_active, float* d_out); /* device pointer, asynchronous on the handle's stream */

/* shanten.rs:244-261 calculate_shanten / :470-484 calculate_shanten_3p over raw 34-histograms
 * (len_div3 = tile count / 3; -1 = complete hand).  Tables are generated at first use, on the host. */
int rmj_shanten(int device, const uint8_t* counts /*[n][34]*/, uint32_t n, int sanma, int8_t* out /*[n]*/);

/* shanten.rs:304-327 calculate_effective_tiles_with_discard / :525-548 _3p_with_discard: number of tile types whose
 * draw lowers the shanten (3n+1 hand), or the best such count over the discards that do not raise it (3n+2 hand).
 * The reference takes 136-ids and panics on a 3n hand; only types matter, a 3n hand yields 0xFFFFFFFF. */
int rmj_effective_tiles(int device, const uint8_t* counts /*[n][34]*/, uint32_t n, int sanma, uint32_t* out /*[n]*/);
/* shanten.rs:331-405 calculate_best_ukeire / :552-626 _3p: best, over the discards that do not raise the shanten, of the
 * number of live tiles (4 - visible - held, saturating) whose draw lowers it. */
int rmj_best_ukeire(int device, const uint8_t* counts /*[n][34]*/, const uint8_t* visible /*[n][34]*/, uint32_t n, int sanma,
                    uint32_t* out /*[n]*/);

/* ------------------------------------------------------------------ trainer-side device interface
 * Zero-copy views of the observation outputs for a policy that runs on the same GPU (riichienv-ml's PPO worker loop,
 * trainers/_ppo_worker.py:113-466, reads obs.mask() / obs.encode() per game on the host).  The pointers stay valid until
 * rmj_destroy; their contents are rewritten by every step / reset / apply call, in order, on `stream`. */
typedef struct RmjDeviceViews {
    uint32_t n_games, reserved;
    const uint32_t* status;   /* [n]        active_mask | phase << 8 | is_done << 16 */
    const uint8_t* nlegal;    /* [n][4]     */
    const uint64_t* legal;    /* [n][4][64] packed actions */
    const uint8_t* mask;      /* [n][4][82] action-id mask (first 60 ids in 3P) */
    const uint64_t* waits;    /* [n][4]     34-bit wait masks */
    void* stream;             /* hipStream_t of the handle */
} RmjDeviceViews;
int rmj_device_views(rmj_handle h, RmjDeviceViews* out);
/* step with the policy's action ids ([n][4] int32 on the device, -1 = no action): Observation.find_action
 * (observation/python.rs:119-122) + RiichiEnv.step; auto_reset != 0 restarts finished games like rmj_step_random */
int rmj_step_ids_device(rmj_handle h, const int32_t* d_action_ids, int auto_reset);
/* rmj_step_ids_device + rmj_encode_device(h, 2, d_out) as ONE launch: the step under the policy's ids, then Observation.encode() of
 * the seats that are to act next into the resident tensor d_out [n][4][74][34 | 27] (rows of the other seats untouched) - the
 * trainer loop's iteration (riichienv-ml trainers/_ppo_worker.py:151-239: step, then obs.encode() of the returned observations)
 * with one launch gap instead of two. */
int rmj_step_ids_encode_device(rmj_handle h, const int32_t* d_action_ids, int auto_reset, float* d_out);
/* rmj_sample_ids_device(h, d_logits, stride, seed, d_ids) + rmj_step_ids_encode_device(h, d_ids, auto_reset, d_out) as ONE launch (round 5):
 * every wave draws the ids of its own four games from the policy's logits (the same keyed draw: the ids are the ones the two calls
 * produce, and they are written to d_ids [n][4] for the caller's log-probabilities), steps under them and encodes the seats that act
 * next - the whole environment side of a trainer iteration (trainers/_ppo_worker.py:151-239) between two policy forward passes. */
int rmj_step_sample_encode_device(rmj_handle h, const float* d_logits, uint32_t stride, uint64_t seed, int auto_reset, int32_t* d_ids, float* d_out);
/* Masked categorical sampling for a policy on the same GPU (what riichienv-ml's PPO worker does per game on the host with
 * obs.mask(), trainers/_ppo_worker.py:164-239): for every seat that is to act, one action id drawn from
 * softmax(logits) restricted to the seat's legal ids (Gumbel-max on the resident mask slab); d_logits [n][4][stride] f32 on
 * the device (stride >= 82 / 60; masked entries are never read as candidates), NULL = uniform over the legal ids.
 * d_ids [n][4] int32, -1 for seats that do not act: the input of rmj_step_ids_device.  Counter-based noise: the same
 * (seed, state) gives the same ids.  Asynchronous on the handle's stream. */
int rmj_sample_ids_device(rmj_handle h, const float* d_logits, uint32_t stride, uint64_t seed, int32_t* d_ids);
/* Round boundaries and per-round score deltas for a trainer on the same GPU (what riichienv-ml's PPO worker computes on the host
 * between steps: trainers/_ppo_worker.py:100-116 GRP features, :240-266 the reward at a kyoku boundary, :283-291 rank rewards).
 * Call after every step (asynchronous on the handle's stream): d_ended [n] u8 = 0 the round goes on, 1 a round ended in this step and
 * the next one was dealt, 2 the round AND the game ended; for ended != 0, d_delta [n][4] i32 = the seats' scores now minus their
 * scores when that round was dealt, d_meta [n][4] i32 = round_wind, oya, honba, riichi_sticks at that deal (chang / ju / ben /
 * liqibang); zeros otherwise.  d_kyoku_idx [n] u8 = RiichiEnv.kyoku_idx.  Any output may be NULL.  The first call (and
 * rmj_round_track_reset) only takes the baseline; a finished game that was restarted (auto-reset / rmj_reset) re-opens without a
 * boundary.  Unlike the worker's kyoku_idx comparison a renchan counts as a boundary too (the wall's hand index moves). */
int rmj_round_track_device(rmj_handle h, uint8_t* d_ended, int32_t* d_delta, int32_t* d_meta, uint8_t* d_kyoku_idx);
int rmj_round_track_reset(rmj_handle h);
/* scores() (env.rs:401-404) into a device buffer [n][4]; d_event_counts [n] may be NULL */
int rmj_scores_device(rmj_handle h, int32_t* d_scores, uint32_t* d_event_counts);
/* RiichiEnv.points(rule_name) (riichienv-python/src/env.rs:691-727, ranks :673-689) of every game, computed on the device in f64
 * like the reference: (score - base) / 1000 * weight + uma[rank - 1]; rule 0 = "basic", 1 = "ouza-tyoujyo", 2 = "ouza-normal"
 * (3P: "basic" only; anything else -> RMJ_ERR_ARG like the reference's ValueError).  d_points / points: [n][4] f64, 0 for the
 * fourth seat of a 3P game.  The device version is asynchronous on the handle's stream (the reward of a trainer-side loop). */
int rmj_points_device(rmj_handle h, int rule, double* d_points);
int rmj_get_points(rmj_handle h, int rule, double* points);
int rmj_sync(rmj_handle h); /* wait for the handle's stream */
/* Issue all further work of the handle on the caller's HIP stream (e.g. the stream of the policy's framework), so that
 * kernels of the library and of the policy are ordered by the stream itself and no host synchronisation is needed
 * between them (NULL = the device's default stream, which is what frameworks use unless told otherwise); own != 0 returns
 * to the handle's own stream.  Work already issued is waited for first. */
int rmj_set_stream(rmj_handle h, void* hip_stream, int own);

/* ------------------------------------------------------------------ MJAI event ingestion
 * RiichiEnv.apply_event (riichienv-python/src/env.rs:880-887) -> GameState::apply_mjai_event
 * (state/event_handler.rs:18-330, state_3p/event_handler.rs:18-362) for every game at once: events[n][3] holds one MJAI
 * event per game as binary records (a start_kyoku is START_KYOKU + two TEHAI records; type NONE = no event for that
 * game).  Tile names are mapped to ids by the caller (parser.rs:336-385 mjai_to_tid; riichienv_amd/abi.py).  Afterwards
 * the observation outputs (status, legal lists, masks, waits) describe the new state like after rmj_step.
 * Bit-exact parity is claimed for full-information streams; a masked "?" tile is mapped to tile 0 by the host mapper on
 * request, like parse_mjai_tile (event_handler.rs:8-10), which leaves the masked seats in a garbage state in both
 * implementations (only the observing seat's outputs are meaningful).  The caller-side mjai_log recording of
 * env.rs:56-72 is not reproduced (start_game clears the device log, later events are not appended).
 * RMJ_EVF_REPLAY_PASS in the first record's `pad`: the bookkeeping of the reference's log walker on top of the event
 * (KyokuStepIterator, replay/mod.rs:129-177; apply_log_action, state/event_handler.rs:391-392): a seat that was offered Ron on
 * the last discard and does not win with this event has passed (same-turn furiten, permanent in riichi), a discard ends
 * the discarder's same-turn furiten, and the tile dealt after a kan (3P: after a kita too) is a rinshan draw (is_after_kan,
 * event_handler.rs:428: a win on it whose only yaku is rinshan kaihou is offered) - what (observation, action) datasets built
 * from logs need.  reach_accepted / dora events leave the published lists untouched (the claims on a riichi declaration tile
 * are decided after reach_accepted). */
#define RMJ_EVF_REPLAY_PASS 1u
int rmj_apply_events(rmj_handle h, const RmjEvent* events /*[n][3]*/);

/* Auxiliary feature blocks of an Observation that are not part of encode() / encode_extended(); absolute seat order,
 * public information only, so one block per game serves every observing seat (NP = 4, W = 34; 3P: NP = 3, W = 27):
 *   RMJ_AUX_KAWA_OVERVIEW     out[n][NP][7][W]   Observation.encode_kawa_overview (observation/python.rs:881-925,
 *                                                observation_3p/python.rs:759-810), incl. its red-five id / column quirks
 *   RMJ_AUX_YAKU_POSSIBILITY  out[n][NP][21][2]  Observation.encode_yaku_possibility (observation/python.rs:327-455,
 *                                                observation_3p/python.rs:275-400) over yaku_checker.rs:27-412
 *   RMJ_AUX_FURITEN_RON       out[n][NP][21]     Observation.encode_furiten_ron_possibility (observation/python.rs:251-293);
 *                                                all ones: the reference never fills tsumogiri_flags (observation/mod.rs:105)
 * `out` is a host pointer (rmj_encode_aux) or a device pointer written on the handle's stream (rmj_encode_aux_device). */
enum { RMJ_AUX_KAWA_OVERVIEW = 0, RMJ_AUX_YAKU_POSSIBILITY = 1, RMJ_AUX_FURITEN_RON = 2 };
int rmj_encode_aux(rmj_handle h, int which, float* out);
int rmj_encode_aux_device(rmj_handle h, int which, float* d_out);

/* Sequence (transformer) features, observation/sequence_features.rs (4-player games only, like the reference; spec
 * docs/SEQUENCE_FEATURE_ENCODING.md), for every (game, seat): Observation.encode_seq_sparse(game_style) (:331-378),
 * encode_seq_numeric (:447-471), encode_seq_candidates (:697-813) and, per game, encode_seq_progression (:503-671).
 * Arrays are padded to fixed lengths with the reference's padding values (441; (4,276,2,2,4); (279,2,2,3)), the real
 * lengths are returned beside them.  The reference derives these features from the MJAI strings an Observation
 * carries (`events`, the seat's log since its previous observation); the device defines them over the events of the
 * CURRENT ROUND, read from the binary event ring: the progression is GameState::round_seq_progression (the snapshot
 * the reference attaches with enable_seq_caching, state/mod.rs:257-260, 2150-2161), the drawn tile, the last
 * discarder and the round-start honba / deposits / scores are those of the round.  The ring must still hold the
 * round's start_kyoku (create the handle with event_ring >= 256): otherwise n_progression[g] = 0xFFFF and the
 * round-start numbers fall back to the current ones (sequence_features.rs:490).  Seats that are not to act get no
 * candidates. */
#define RMJ_SEQ_SPARSE 25
#define RMJ_SEQ_PROG 256
#define RMJ_SEQ_CAND 64
typedef struct RmjSeqBuffers {
    uint16_t* sparse;        /* [n][4][25]     token ids, padded with 441 */
    uint8_t* n_sparse;       /* [n][4]         */
    float* numeric;          /* [n][4][12]     */
    uint16_t* progression;   /* [n][256][5]    (actor, type, moqie, liqi, from), padded with (4,276,2,2,4) */
    uint16_t* n_progression; /* [n]            */
    uint16_t* candidates;    /* [n][4][64][4]  (type, moqie, liqi, from) in legal-list order, padded with (279,2,2,3) */
    uint8_t* n_candidates;   /* [n][4]         */
} RmjSeqBuffers;
/* The same features over the events of ONE OBSERVATION, as the reference's live environment computes them
 * (Observation.events = the seat's log since its previous observation, state/mod.rs:211-218; enable_seq_caching is off
 * outside the replay path): the progression holds only that delta (per seat), the drawn-tile token exists only while the
 * delta still contains the seat's tsumo, the round-start numbers come from a start_kyoku inside the delta (else the current
 * ones, sequence_features.rs:490), the last discarder is searched in the delta.  The library keeps the seats' event
 * cursors in the record: every publication of observations for an acting seat (reset, step) advances that seat's cursor
 * like get_observation does.  Seats that are not to act get empty outputs; n_progression = 0xFFFF if the ring no longer
 * holds the delta. */
#define RMJ_SEQ_DELTA_PROG 64
typedef struct RmjSeqDeltaBuffers {
    uint16_t* sparse;        /* [n][4][25]     */
    uint8_t* n_sparse;       /* [n][4]         */
    float* numeric;          /* [n][4][12]     */
    uint16_t* progression;   /* [n][4][64][5]  per seat: the delta's entries, padded with (4,276,2,2,4) */
    uint16_t* n_progression; /* [n][4]         */
    uint16_t* candidates;    /* [n][4][64][4]  */
    uint8_t* n_candidates;   /* [n][4]         */
} RmjSeqDeltaBuffers;
int rmj_encode_seq_delta(rmj_handle h, int game_style, const RmjSeqDeltaBuffers* out);          /* host arrays */
int rmj_encode_seq_delta_device(rmj_handle h, int game_style, const RmjSeqDeltaBuffers* d_out); /* device arrays, handle's stream */
int rmj_encode_seq(rmj_handle h, int game_style, const RmjSeqBuffers* out);          /* host arrays */
int rmj_encode_seq_device(rmj_handle h, int game_style, const RmjSeqBuffers* d_out); /* device arrays, handle's stream */

/* ------------------------------------------------------------------ scheduling */
/* Parts (HIP streams) a multi-step device rollout of this handle is cut into, 1..8 (default 4, or RMJ_STEP_STREAMS in
 * the environment when the handle is created); see rmj_step_random. */
int rmj_set_rollout_streams(rmj_handle h, int k);

/* Measurement entry points (rmj_bench_*, rmj_time_rollout*, rmj_total_full_path) and the test-only environment hooks are declared in
 * riichi_mi355x_bench.h: they are what bench.py, the profiles and the tests use, not part of the drop-in surface. */

#ifdef __cplusplus
}
#endif
#endif /* RIICHI_MI355X_H */
