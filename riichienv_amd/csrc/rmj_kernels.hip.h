// Variant-dependent kernels of the step path; included once per variant (see rmj_step.hip.h) inside namespace RMJ_NS.
namespace RMJ_NS {

// RiichiEnv.reset defaults (env.rs:799-851) executed on device: reset() + _initialize_round(0,0,0,0,None,default scores)
__device__ __noinline__ void ol_env_reset_default(CtxV v) {
    CTX_FROM(v);
    c.S.ev_base = c.S.ev_count;  // GameState::reset clears the logs (state/mod.rs:171-187): the new game's log starts here
    if (c.lane < 4) { c.S.obs_from[c.lane] = c.S.ev_count; c.S.obs_upto[c.lane] = c.S.ev_count; }
    emit_simple(c, RMJ_EV_START_GAME);
    const int32_t st = KSANMA ? 35000 : 25000;  // state_3p/game_mode.rs:31-33
    const int32_t sc[4] = {st, st, st, st};
    shuffle_wall(c);
    init_round(c, 0, 0, 0, 0, sc);
}

// Full-featured step of one game, out of line: entered from k_step when the fast path met a rare transition (or the
// game must be reset).  It starts over from the HBM record - the fast path never stores before it is sure.
__device__ __noinline__ void ol_step_full(CtxV v, uint64_t mine, uint32_t flags) {
    CTX_FROM(v);
    flags = uni(flags);
    GState& S = c.S;
    if (flags & STEP_F_CONT_RYU) {
        // entered AT the exhaustive draw (k_step4): tier 0 has taken the step up to _deal_next's empty wall - the discard, its claims or
        // the seats' passes, riichi acceptance, the turn counter - on the record in LDS; what is left is state/mod.rs:1571-1574
        wave_sync();
        S.full_count += 1;
        S.is_rinshan = 0;
        trigger_ryukyoku(c, RMJ_RK_EXHAUSTIVE, 0);
        if (U(S.turn_count) >= (uint32_t)KNP) S.is_first_turn = 0;
    } else if (flags & STEP_F_CONT_CLAIMS) {
        // entered behind a discard tier 0 has made (pass 2 of a row that paused at the Ron check and then met something tier 0 does not
        // do): claim generation and the rest of _resolve_discard, state/mod.rs:1363-1413
        wave_sync();
        S.full_count += 1;
        resolve_discard_tail<false>(c, U((int)S.last_discard_pid));
    } else if (flags & STEP_F_CONT_FIN) {
        // the step is complete on the record in LDS (a round dealt by r4_round_end whose first list tier 0 cannot write): outputs only
        wave_sync();
        S.full_count += 1;
    } else {
        load_state(S, c.E.core + c.g, c.lane);
        TLF(c, 0);
        S.full_count += 1;
        if (S.is_done && (flags & STEP_F_AUTORESET)) ol_env_reset_default(v);
        else step_game<false>(c, mine, (flags & STEP_F_RANDOM) != 0);
    }
    TLF(c, 11);
    finalize_outputs<false>(c, true, true, (flags & STEP_F_ALLROWS) != 0u);
    TLF(c, 12);
    store_state(S, c.E.core + c.g, c.lane);
    TLF(c, 13);
}

#ifndef RMJ_STEP_WAVES
#define RMJ_STEP_WAVES 8
#endif
__global__ __launch_bounds__(64 * RMJ_STEP_WPB, RMJ_STEP_WAVES) void k_step(const Env* __restrict__ Ep, const uint64_t* __restrict__ actions, uint64_t policy_seed, uint32_t flags,
                                                                                uint32_t g_base, uint32_t g_end) {
    CEnv& E = *(CEnv*)Ep;  // device-resident record, read through the constant address space (see CEnv)
    __shared__ BlockSharedT<RMJ_STEP_WPB> sh;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t g = g_base + blockIdx.x * RMJ_STEP_WPB + wave;  // the launch covers games [g_base, g_end)
    if (g >= g_end) return;
    GState& S = sh.st[wave];
    PROF_START(sh.x[wave], lane);
    const bool device_policy = (flags & STEP_F_RANDOM) != 0;
    load_state(S, E.core + g, lane);
    Ctx c{S, E, sh.x[wave], g, lane, E.wall + (size_t)g * RMJ_WALL_STRIDE, E.legal + (size_t)g * 4 * RMJ_MAX_LEGAL};
    c.pf_live_end = S.live_end;
    c.pf_draw = c.W[S.live_end > 0 ? S.live_end - 1 : 0];
    PROF(c.X, lane, 0);
    // the seats' actions of this step: lane p (< 4) holds seat p's packed action (act_at reads them back)
    uint64_t mine = RMJ_NO_ACTION;
    if (device_policy) {
        // RandomAgent (random_agent.py:6-15), keyed per (game, step, seat): see rmj_step_random in the header
        // A loop over the acting seats; the modulo runs on the scalar unit (exact, by table), the chosen entry is loaded by
        // lane = seat.  The two hashes are pure arithmetic on wave-uniform values that the compiler would place on the
        // scalar unit; they are kept on the VECTOR unit on purpose (an opaque zero in a VGPR joins the sum): the step is
        // bound by the busier of the two issue ports, after the scalar branches of this round that is the scalar one
        // (642 SALU vs 539 VALU per step), and a 64-bit multiply costs 8 scalar but 5 vector instructions.
        // (Requesting the list heads together with the record, to spare the dependent trip to HBM, was measured
        // slower: latency is hidden by the other waves, the extra shuffle and traffic are not.)
        uint32_t vz;
        asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
        const uint64_t gs = sm64(policy_seed + E.game_offset + uni(g) + (uint64_t)vz);
        const uint32_t sc_u = uni((uint32_t)S.step_count);
        uint32_t am = S.is_done ? 0u : (uint32_t)S.active_mask;
        am = uni(am) & 0xFu;
        while (am) {
            const int p = __builtin_ctz(am);
            am &= am - 1u;
            const uint32_t n = uni((uint32_t)S.nlegal[p]);
            if (n == 0u) continue;
            const uint32_t ch = uni(policy_pick(policy_key32(gs, sc_u, (uint32_t)p), n > 64u ? 64u : n));
            if (lane == p) mine = c.Lg[p * RMJ_MAX_LEGAL + ch];
        }
    } else if (flags & STEP_F_IDS) {
        // Observation.find_action (observation/python.rs:119-122): the first legal action of the seat whose encoded id
        // equals the policy's id; lane = list entry.  No match = an action that fails validation (illegal action).
        const int32_t* ids = reinterpret_cast<const int32_t*>(actions);
        for (int p = 0; p < 4; p++) {
            const int id = U(ids[(size_t)g * 4 + p]);
            const int n = U((int)S.nlegal[p]);
            if (id < 0 || !((U((uint32_t)S.active_mask) >> p) & 1u) || n == 0 || U((int)S.is_done)) continue;
            uint64_t a = 0;
            bool hit = false;
            if (lane < n) {
                a = c.Lg[p * RMJ_MAX_LEGAL + lane];
                hit = (KSANMA ? a_encode_3p(a) : a_encode(a)) == id;
            }
            const uint64_t b = __ballot(hit);
            const uint64_t chosen = b ? act_at(a, __ffsll((long long)b) - 1) : mk_action(0x7F, RMJ_TILE_NONE, 0);
            if (lane == p) mine = chosen;
        }
    } else if (lane < 4) {
        uint64_t a = actions[(size_t)g * 4 + lane];
        mine = ((a & 0xFF) == 0xFF) ? RMJ_NO_ACTION : a_canon(a);
    }
    PROF(c.X, lane, 1);
    // Fast path: the common transitions, fully inline, nothing stored until it has succeeded.
    if (U((int)S.is_done) && (flags & STEP_F_AUTORESET)) c.bail = true;
    else {
        c.ev_stage = 0;
        step_game<true>(c, mine, device_policy);  // device policy: picked from the stored lists, valid by construction
        PROF(c.X, lane, 8);
        if (!c.bail) finalize_outputs<true>(c, true);
    }
    if (c.bail) {
        PROF(c.X, lane, 26);
#ifdef RMJ_PROFILE
        const uint64_t t_full = __builtin_readcyclecounter();
#endif
        ol_step_full(ctx_pack(c), mine, flags);
        PROF(c.X, lane, 25);
#ifdef RMJ_PROFILE
        if (lane == 0) { c.X.pacc[28] += (uint32_t)(__builtin_readcyclecounter() - t_full); c.X.pacc[32 + 28] += 1u; }
#endif
    } else {
        flush_events(c);
        store_state_partial(S, E.core + g, lane, U(c.dirty));
    }
    PROF(c.X, lane, 15);
    PROF_FLUSH(c.X, lane, g);
}

__device__ __noinline__ void freeze_wall_digest(const uint8_t* W, int total, uint32_t* out) {
    uint64_t salt = 0;
    for (int k = 0; k < 8; k++) salt |= (uint64_t)W[136 + k] << (8 * k);
    uint32_t dg[8];
    sha256_wall(W, total, salt, dg);
    for (int k = 0; k < 8; k++) out[k] = dg[k];
}
__global__ __launch_bounds__(256, 4) void k_reset(const Env* __restrict__ Ep, ResetArgs A) {
    CEnv& E = *(CEnv*)Ep;
    __shared__ BlockShared sh;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t g = blockIdx.x * WPB + wave;
    if (g >= E.n_games) return;
    if (!A.is_ctor && A.select && !A.select[g]) return;
    GState& S = sh.st[wave];
    if (A.is_ctor) {
        for (int i = lane; i < (int)(sizeof(GState) / 4); i += 64) reinterpret_cast<uint32_t*>(&S)[i] = 0u;
        wave_sync();
    } else {
        load_state(S, E.core + g, lane);
    }
    Ctx c{S, E, sh.x[wave], g, lane, E.wall + (size_t)g * RMJ_WALL_STRIDE, E.legal + (size_t)g * 4 * RMJ_MAX_LEGAL};
    const int32_t st0 = KSANMA ? 35000 : 25000;
    int32_t sc[4] = {st0, st0, st0, st0};
    if (A.is_ctor) {  // GameState::new, state/mod.rs:98-167
        S.wall_seed = A.seeds ? A.seeds[g] : sm64(A.base_seed + E.game_offset + g);  // shard.game_seed: decorrelated default seeds
        S.hand_index = 0;
        S.last_error_pid = 0xFF;
        S.pending_kan_pid = 0xFF;
        S.riichi_pending = 0xFF;
        S.drawn_tile = 0xFF;
        S.last_discard_pid = 0xFF;
        for (int p = 0; p < 4; p++) S.p[p].score = st0;
        emit_simple(c, RMJ_EV_START_GAME);
        shuffle_wall(c);
        init_round(c, 0, (int)E.ctor_round_wind, 0, 0, nullptr);
    } else {          // env.rs:799-851
        S.ev_base = S.ev_count;
        if (lane < 4) { S.obs_from[lane] = S.ev_count; S.obs_upto[lane] = S.ev_count; }
        emit_simple(c, RMJ_EV_START_GAME);
        if (A.scores)
            for (int p = 0; p < 4; p++) sc[p] = A.scores[(size_t)g * 4 + p];
        if (A.walls) {
            const int total = KSANMA ? 108 : 136;  // 3P: the first 108 entries of the [136] row
            for (int i = lane; i < total; i += 64) c.X.tiles[i] = A.walls[(size_t)g * 136 + (total - 1 - i)];  // load_wall: reverse
            if (lane < 8) c.X.tiles[136 + lane] = c.W[136 + lane];   // salt / digest stay what they were (state/wall.rs:69-80) ...
            if (S.wall_meta == 1) {                                   // ... so the digest of the wall that goes away is evaluated now
                if (lane == 0) freeze_wall_digest(c.W, total, E.wall_dg + (size_t)g * 8);
                S.wall_meta = 2;
            }
            wave_sync();
        } else {
            shuffle_wall(c);
        }
        init_round(c, A.oya ? A.oya[g] : 0, A.round_wind ? A.round_wind[g] : 0, A.honba ? A.honba[g] : 0,
                   A.kyotaku ? A.kyotaku[g] : 0u, sc);
    }
    finalize_outputs<false>(c, true);
    store_state(S, E.core + g, lane);
}

// rmj_apply_events: one MJAI event (up to three records) per game; games whose first record is NONE are left alone
__global__ __launch_bounds__(256, 4) void k_apply_event(const Env* __restrict__ Ep, const RmjEvent* __restrict__ ev) {
    CEnv& E = *(CEnv*)Ep;
    __shared__ BlockShared sh;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t g = blockIdx.x * WPB + wave;
    if (g >= E.n_games) return;
    const RmjEvent* mine = ev + (size_t)g * 3;
    if (mine[0].type == RMJ_EV_NONE || mine[0].type == RMJ_EV_END_GAME || mine[0].type == RMJ_EV_TEHAI) return;
    GState& S = sh.st[wave];
    load_state(S, E.core + g, lane);
    Ctx c{S, E, sh.x[wave], g, lane, E.wall + (size_t)g * RMJ_WALL_STRIDE, E.legal + (size_t)g * 4 * RMJ_MAX_LEGAL};
    // reach_accepted and dora move points and indicators only (event_handler.rs:311-319): whatever the seats were offered - the claims
    // on a riichi declaration tile, say - stands, and so do the published lists, masks and status
    const bool passive = mine[0].type == RMJ_EV_REACH_ACCEPTED || mine[0].type == RMJ_EV_DORA;
    apply_event(c, mine);
    if (!passive) finalize_outputs<false>(c, true, false);  // apply_event hands out no observation (env.rs:880-887)
    store_state(S, E.core + g, lane);
}

// recompute observation outputs of one game after rmj_poke_state
__global__ __launch_bounds__(64, 4) void k_refresh(const Env* __restrict__ Ep, uint32_t g) {
    CEnv& E = *(CEnv*)Ep;
    __shared__ GState st;
    __shared__ WaveScratch x;
    const int lane = threadIdx.x & 63;
    load_state(st, E.core + g, lane);
    Ctx c{st, E, x, g, lane, E.wall + (size_t)g * RMJ_WALL_STRIDE, E.legal + (size_t)g * 4 * RMJ_MAX_LEGAL};
    bool keep = st.phase == RMJ_WAIT_RESPONSE && st.pending_kan_pid != 0xFF;  // chankan claims are not reconstructible
    if (!keep) {
        finalize_outputs<false>(c, false, false);
        store_state(st, E.core + g, lane);
    }
}

// rmj_copy_games: the complete per-game state (record, wall, published lists / masks / waits / status, event ring, win results) of
// game src_idx[i] of `Sp` into game dst_idx[i] of `Dp`; one wave per pair, 4-byte words (every slab stride is a multiple of 4).
__device__ __forceinline__ void copy_words(void* dst, const void* src, size_t bytes, int lane) {
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
    for (size_t i = lane; i < bytes / 4; i += 64) d[i] = s[i];
}
__global__ __launch_bounds__(64) void k_copy_games(const Env* __restrict__ Dp, const Env* __restrict__ Sp, const uint32_t* __restrict__ dst_idx,
                                                   const uint32_t* __restrict__ src_idx, uint32_t n) {
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    CEnv& D = *(CEnv*)Dp;
    CEnv& S = *(CEnv*)Sp;
    const size_t a = dst_idx[i], b = src_idx[i];
    if (a >= D.n_games || b >= S.n_games) return;   // (device index arrays are not checked on the host)
    const int lane = threadIdx.x & 63;
    const size_t ring = (size_t)S.ring_mask + 1u;
    copy_words(D.core + a, S.core + b, sizeof(GState), lane);
    copy_words(D.wall + a * RMJ_WALL_STRIDE, S.wall + b * RMJ_WALL_STRIDE, RMJ_WALL_STRIDE, lane);
    copy_words(D.legal + a * 4 * RMJ_MAX_LEGAL, S.legal + b * 4 * RMJ_MAX_LEGAL, 4 * RMJ_MAX_LEGAL * sizeof(uint64_t), lane);
    copy_words(D.nlegal + a * 4, S.nlegal + b * 4, 4, lane);
    copy_words(D.mask + a * 328, S.mask + b * 328, 328, lane);
    copy_words(D.waits + a * 4, S.waits + b * 4, 4 * sizeof(uint64_t), lane);
    copy_words(D.status + a, S.status + b, sizeof(uint32_t), lane);
    copy_words(D.events + a * ring, S.events + b * ring, ring * sizeof(RmjEvent), lane);
    copy_words(D.win + a * 4, S.win + b * 4, 4 * sizeof(RmjWinResult), lane);
    copy_words(D.wall_dg + a * 8, S.wall_dg + b * 8, 32, lane);
}

}  // namespace RMJ_NS
