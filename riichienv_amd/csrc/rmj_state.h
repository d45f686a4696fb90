// Device-resident game state layout (one 640-byte record per game) and the slabs around it.
//
// Layout rationale: the step path runs ONE WAVEFRONT PER GAME, so the
// coalesced access pattern is "64 lanes read one game's contiguous record" (40 lanes x 16 B),
// not "64 lanes read field f of 64 games".  State is therefore split by FIELD GROUP into
// separate HBM slabs (structure-of-arrays at slab granularity), each slab holding one
// contiguous, 16-byte-aligned record per game:
//     core  [B] x 640 B   GState   (hands, melds, discards, flags, scores, globals)
//     wall  [B] x 144 B            (136 tile ids, immutable within a kyoku; cursors live in core)
//     legal [B] x 4 x 64 x 8 B     (ordered legal-action lists = current_claims of the reference)
//     nlegal[B] x 4 B, mask [B] x 4 x 82 B, waits [B] x 4 x 8 B, status [B] x 4 B
//     events[B] x ring x 32 B      (binary MJAI records, host-formatted)
//
// Mirrors: GameState (state/mod.rs:31-91), PlayerState (state/player.rs:6-39),
// WallState (state/wall.rs:8-19) of the reference.
#pragma once
#include <stdint.h>

#define RMJ_WALL_STRIDE 144

// PState.flags bits (state/player.rs:19-37)
#define PF_RIICHI_DECLARED 1u
#define PF_RIICHI_STAGE 2u
#define PF_DOUBLE_RIICHI 4u
#define PF_MISSED_RIICHI 8u
#define PF_MISSED_DOUJUN 16u
#define PF_NAGASHI 32u
#define PF_IPPATSU 64u
#define PF_WAITS_VALID 128u /* derived cache bit: waits13 is current (never exported) */

struct alignas(16) PState {  // 128 bytes
    uint8_t hand[14];        // 136-ids; 13 sorted + drawn tile last (state/mod.rs:1575-1579)
    uint8_t hand_len;
    uint8_t n_melds;
    uint8_t meld_type[4];    // RMJ_MELD_*
    uint8_t meld_from[4];    // from_who, 0xFF = -1
    uint8_t meld_called[4];  // called_tile, 0xFF = None
    uint8_t meld_tiles[4][4];// sorted 136-ids (3 for chi/pon, 4 for kans)
    uint8_t n_discards;
    uint8_t flags;           // PF_*
    uint8_t pao37, pao50;    // liable seat for daisangen / daisuushi, 0xFF = none
    uint8_t n_forbidden;
    uint8_t forbidden[2];    // 136-ids, compared by type (legal_actions.rs:82-88)
    uint8_t riichi_decl_idx; // 0xFF = None
    uint8_t riichi_sutehai;  // 0xFF = None
    uint8_t last_tedashi;    // 0xFF = None
    uint8_t pad0[2];
    int32_t score, score_delta;
    uint32_t discard_from_hand_bits, discard_is_riichi_bits;
    uint64_t discard_type_mask;  // derived cache: bit t set iff some discard has type t
    uint8_t discards[32];
    uint64_t waits13;            // derived cache: get_waits of the 13-tile hand (valid iff PF_WAITS_VALID)
    uint8_t n_kita;              // 3P: kita_tiles (state_3p/player.rs:38)
    uint8_t kita[4];
    uint8_t sh13;                // derived cache: shanten number of the 13-tile hand (valid iff PF_WAITS_VALID)
    uint8_t pad1[2];
};

struct alignas(16) GState {  // 4*128 + 128 = 640 bytes
    PState p[4];
    uint64_t wall_seed;       // WallState.seed (episode seed)
    uint64_t pending_kan_action;
    uint32_t hand_index;      // WallState.hand_index
    uint32_t step_count;      // env.step calls that advanced this game
    uint32_t ev_count;        // MJAI records this game slot has emitted so far (stream position; the current game began at ev_base)
    uint32_t turn_count;
    uint32_t riichi_sticks;
    uint8_t current_player, phase, active_mask, is_done;
    uint8_t needs_tsumo, oya, honba, kyoku_idx;
    uint8_t round_wind, is_rinshan, is_first_turn, riichi_pending;  // riichi_pending: 0xFF none
    uint8_t drawn_tile, last_discard_pid, last_discard_tile, pending_kan_pid;  // 0xFF none
    // wall cursors over the fixed 136-array W (= WallState.tiles before any pop/remove):
    //   tiles.pop()      -> W[--live_end]
    //   tiles.remove(0)  -> W[rinshan_count++]
    //   tiles[i]         -> W[i + rinshan_count]   (so dora k = W[4+2k], ura k = W[5+2k])
    uint8_t live_end, rinshan_count, pending_kan_dora, drawable_count;
    uint8_t n_dora, dora[5];
    uint8_t ron_offer_mask;   // seats whose stored claim list contains Ron (state/mod.rs:902-917)
    uint8_t last_error_pid;   // 0xFF none (quirk Q9)
    uint8_t wall_total;       // 136 (4P) / 108 (3P)
    // Number of entries of each seat's stored claim list that are still in the reference's `current_claims`
    // (it is cleared only by _resolve_discard / all-pass / _initialize_round, so claims survive an accepted call and
    // resurface, in front of a chankan / kita Ron offer, if the caller kans or declares kita before discarding).
    uint8_t stale_n[4];
    uint8_t nlegal[4];        // copy of the nlegal slab row: the policy / validation need it as soon as the record arrives
    uint8_t replay_after_kan;   // is_after_kan of the reference's log walker (apply_log_action): set by rmj_apply_events in replay mode only
    uint8_t pad0_[2];
    uint32_t full_count;      // measurement only: steps of this game that took the full path of k_step (bench.py)
    // player_event_counts of the reference (state/mod.rs:65, 211-218): the events [obs_from[p], obs_upto[p]) are the delta
    // (Observation.events) of seat p's latest observation; advanced whenever observations are published for an acting seat
    uint32_t obs_from[4], obs_upto[4];
    uint8_t win_mask;         // seats with an entry in the win-result slab (win_results of the reference; cleared per round)
    // derived cache (never exported): the slots of seat tp_seat's 14-tile hand whose discard keeps the hand tenpai, as computed for
    // the Riichi entry of the list published in step tp_step - 1; the Riichi declaration of the next step reuses it for the
    // riichi-stage list (the hand cannot change in between).  Valid iff tp_step == step_count and tp_seat == current player.
    uint8_t tp_seat;
    uint16_t tp_mask;
    uint32_t tp_step;
    // WallState.salt / wall_digest (state/wall.rs:48-55): 1 after a shuffle under RMJ_RULE_REFERENCE_RNG (the salt is kept in bytes
    // 136..143 of the wall row; the digest is a function of salt and wall, evaluated when asked for: rmj_get_wall_digest); 2 once
    // that wall has been replaced (load_wall / poke leave salt and digest alone, state/wall.rs:69-80): the digest was evaluated
    // before the replacement and sits in Env::wall_dg; 0 = empty strings (never shuffled that way, or cleared by a start_kyoku
    // event, event_handler.rs:81-82)
    uint8_t wall_meta;
    uint8_t pad[3];
    // ev_count is a position in the game SLOT's record stream and never goes back: a restart (auto-reset, rmj_reset, a start_game
    // event) sets ev_base = ev_count instead of clearing the count, so the ring goes on behind the finished game's last records and a
    // drain cursor stays valid across restarts.  The current game's log (GameState.mjai_log: cleared by reset, state/mod.rs:171-187) is
    // the records [ev_base, ev_count); obs_from / obs_upto are stream positions too.  (u32, wraps after 2^32 records of one slot.)
    uint32_t ev_base;
};

#ifdef __cplusplus
static_assert(sizeof(PState) == 128, "PState must be 128 bytes");
static_assert(sizeof(GState) == 640, "GState must be 640 bytes");
#endif
