"""Behavioural KATs transcribed from the reference's own tests (file:line cited per scenario).  Each scenario
takes an env factory (oracle, HIP, or the dual adapter that cross-checks both after every call) and asserts the
values the reference's tests assert.  State is poked exactly like the reference's Python setters
(tests/env/helper.py:4-70 -> env.rs:134-622)."""
import json

from riichienv_amd import abi
from riichienv_amd.abi import (ANKAN, CHI, DAIMINKAN, DISCARD, KAKAN, KYUSHU, PASS, PON, RIICHI, RON, TSUMO, WAIT_ACT,
                               WAIT_RESPONSE, pack_action, unpack_action)

PON_M, CHI_M, DAIMINKAN_M, ANKAN_M, KAKAN_M = abi.MELD_PON, abi.MELD_CHI, abi.MELD_DAIMINKAN, abi.MELD_ANKAN, abi.MELD_KAKAN


def tiles(s):
    """'23m11p0s' -> 136-ids; 0 = red five (id 16/52/88); copies handed out in order (skipping the red copy)."""
    out, digits, used = [], [], {}
    for ch in s:
        if ch.isdigit():
            digits.append(int(ch))
        else:
            suit = "mpsz".index(ch)
            for d in digits:
                if d == 0:
                    out.append(suit * 36 + 16)
                    continue
                t = suit * 9 + d - 1
                k = used.get(t, 1 if (suit < 3 and d == 5) else 0)
                out.append(t * 4 + k)
                used[t] = k + 1
            digits = []
    return out


def set_meld(mv, mtype, mtiles, opened=True, from_who=-1, called=-1):
    mv.meld_type = mtype
    mv.n_tiles = len(mtiles)
    for i, t in enumerate(sorted(mtiles)):
        mv.tiles[i] = t
    mv.opened = 1 if opened else 0
    mv.from_who = from_who
    mv.called_tile = called


def setup(env, hands=None, melds=None, active_players=None, current_player=0, phase=WAIT_ACT, needs_tsumo=False,
          drawn_tile=None, wall=None, discards=None, riichi_declared=None, points=None, oya=None, round_wind=None,
          mutate=None, reset_kw=None):
    """helper_setup_env (tests/env/helper.py:4-70): reset(wall, oya) then overwrite state."""
    kw = dict(reset_kw or {})
    if wall is not None:
        kw["wall"] = wall
    if oya is not None:
        kw["oya"] = oya
    env.reset(**kw)
    v = env.peek()
    if hands is not None:
        for p in range(4):
            if hands[p] is not None:
                h = sorted(hands[p])
                v.players[p].hand_len = len(h)
                for i, t in enumerate(h):
                    v.players[p].hand[i] = t
    if melds is not None:
        for p in range(4):
            if melds[p]:
                v.players[p].n_melds = len(melds[p])
                for i, m in enumerate(melds[p]):
                    set_meld(v.players[p].melds[i], *m)
    if current_player is not None:
        v.current_player = current_player
    if active_players is not None:
        v.active_mask = sum(1 << p for p in active_players)
    elif current_player is not None and phase == WAIT_ACT:
        v.active_mask = 1 << current_player
    if phase is not None:
        v.phase = phase
    v.needs_tsumo = 1 if needs_tsumo else 0
    if drawn_tile is not None:
        v.drawn_tile = drawn_tile
        pl = v.players[current_player]
        h = list(pl.hand[: pl.hand_len]) + [drawn_tile]
        h = sorted(h)  # helper.py:52-55 appends then sorts
        pl.hand_len = len(h)
        for i, t in enumerate(h):
            pl.hand[i] = t
    else:
        v.drawn_tile = -1
    if discards is not None:
        for p in range(4):
            v.players[p].n_discards = len(discards[p])
            for i, t in enumerate(discards[p]):
                v.players[p].discards[i] = t
    if riichi_declared is not None:
        for p in range(4):
            v.players[p].riichi_declared = 1 if riichi_declared[p] else 0
    if points is not None:
        for p in range(4):
            v.players[p].score = points[p]
    if round_wind is not None:
        v.round_wind = round_wind
    if mutate:
        mutate(v)
    env.poke(v)
    return env


def evs(env, seat=-1):
    return [json.loads(s) for s in env.log(seat)]


def find(legal, atype, tile=None):
    for a in legal:
        t, tl, c = unpack_action(a)
        if t == atype and (tile is None or tl == tile):
            return a
    return None


# ---------------------------------------------------------------------------------------------------------
def sc_paishan_dora_indices(make):
    """tests/env/test_paishan.py:23-40,67-90: wall=list(range(136)) -> dora 131, ura 130; after one rinshan
    draw the kan-dora is 129 and the ura markers are 130, 128."""
    env = setup(make(), wall=list(range(136)))
    v = env.peek()
    assert v.n_dora == 1 and v.dora[0] == 131
    assert v.wall[5] == 130  # ura = tiles[5]
    # give seat 0 four 1m so that an ankan flips the kan dora; riichi first so that ura markers are emitted
    env2 = setup(make(), wall=list(range(136)), hands=[[0, 1, 2] + tiles("234p234s11z")[0:8] + [124, 125], None, None, None],
                 drawn_tile=3)
    a = find(env2.legal(0), ANKAN)
    assert a is not None
    env2.step({0: a})
    v = env2.peek()
    assert v.n_dora == 2 and v.dora[1] == 129 and v.rinshan_draw_count == 1
    assert [v.wall[5 - 1], v.wall[7 - 1]] == [130, 128]  # tiles[] shifted by one after remove(0)
    t = [e["type"] for e in evs(env2)]
    i = t.index("ankan")
    assert t[i: i + 3] == ["ankan", "dora", "tsumo"]  # tests/env/test_kan_dora_timing_events.py:16-66
    assert evs(env2)[i + 2]["pai"] == abi_mjai(135)  # rinshan = tiles[0] of the reversed wall


def abi_mjai(tid):
    from oracle import oracle

    return oracle.tid_to_mjai(tid)


def sc_kakan_dora_timing(make):
    """tests/env/test_kan_dora_timing_events.py:68-139: kakan -> tsumo -> dora -> dahai."""
    env = setup(make(), hands=[[4, 5, 6, 7, 8, 9, 10, 11, 12, 60], None, None, None], melds=[[(PON_M, [0, 1, 2], True, 1, 0)], [], [], []],
                drawn_tile=3)
    a = find(env.legal(0), KAKAN)
    assert a is not None and unpack_action(a) == (KAKAN, 3, [0, 1, 2])
    env.step({0: a})
    act, ph, dn = env.status()
    if ph == WAIT_RESPONSE:  # somebody may hold a chankan ron with a random deal; pass
        env.step({s: pack_action(PASS) for s in range(4) if (act >> s) & 1})
    d = find(env.legal(0), DISCARD)
    env.step({0: d})
    t = [e["type"] for e in evs(env)]
    k = t.index("kakan")
    rest = t[k:]
    assert rest.index("tsumo") < rest.index("dora") < rest.index("dahai")


def sc_daiminkan_dora_timing(make):
    """tests/env/test_kan_dora_timing_events.py:141-215: daiminkan -> tsumo -> dora -> dahai."""
    env = setup(make(), hands=[[4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], [72, 73, 74, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25], None, None],
                drawn_tile=75)
    env.step({0: pack_action(DISCARD, 75)})
    act, ph, dn = env.status()
    assert ph == WAIT_RESPONSE and (act >> 1) & 1
    k = find(env.legal(1), DAIMINKAN)
    assert k is not None and unpack_action(k) == (DAIMINKAN, 75, [72, 73, 74])
    acts = {s: pack_action(PASS) for s in range(4) if (act >> s) & 1}
    acts[1] = k
    env.step(acts)
    env.step({1: find(env.legal(1), DISCARD)})
    t = [e["type"] for e in evs(env)]
    i = t.index("daiminkan")
    rest = t[i:]
    assert rest.index("tsumo") < rest.index("dora") < rest.index("dahai")


def sc_south_round_tsumo(make):
    """tests/test_env_scoring.py:109-153: oya tsumo, South round: yaku {1,11,27,32}, deltas[0] == 18000."""
    env = make(round_wind=1)
    hand = [112, 113, 114] + [0, 4, 8, 5, 9, 12, 16, 20, 24] + [13]

    def mut(v):
        v.is_first_turn = 0
        v.players[0].n_discards = 1
        v.players[0].discards[0] = 0
        v.drawn_tile = 14  # NOT appended to the hand in the reference test

    setup(env, hands=[hand, None, None, None], mutate=mut, reset_kw={"round_wind": 1})
    sk = next(e for e in evs(env) if e["type"] == "start_kyoku")
    assert sk["bakaze"] == "S"
    env.step({0: pack_action(TSUMO)})
    hora = next(e for e in reversed(evs(env)) if e["type"] == "hora")
    assert hora["deltas"] == [18000, -6000, -6000, -6000]
    assert hora["tsumo"] is True and hora["actor"] == 0 and hora["target"] == 0


def sc_illegal_discard_penalty(make):
    """tests/env/test_illegal_actions.py:5-58: oya chombo -12000 / +4000 each, renchan honba 1, last_error set."""
    env = make(game_mode=1)
    env.reset()
    v = env.peek()
    hand = list(v.players[0].hand[: v.players[0].hand_len])
    bad = next(t for t in range(136) if t not in hand)
    env.step({0: pack_action(DISCARD, bad)})
    r = [e for e in evs(env) if e["type"] == "ryukyoku"][-1]
    assert r["reason"] == "Error: Illegal Action by Player 0"
    assert r["deltas"] == [-12000, 4000, 4000, 4000]
    assert env.scores() == [13000, 29000, 29000, 29000]
    v = env.peek()
    assert (v.is_done, v.oya, v.honba, v.kyoku_idx, v.players[0].hand_len, v.last_error_pid) == (0, 0, 1, 0, 14, 0)
    t = [e["type"] for e in evs(env)]
    assert t.index("ryukyoku") < t.index("end_kyoku")


def sc_illegal_out_of_turn(make):
    """tests/env/test_illegal_actions.py:60-121: non-dealer chombo -8000, oya +4000, others +2000; the lowest
    offending seat is punished."""
    env = make(game_mode=1)
    env.reset()
    v = env.peek()
    valid = v.players[0].hand[v.players[0].hand_len - 1]
    env.step({0: pack_action(DISCARD, valid), 1: pack_action(DISCARD, 0), 2: pack_action(DISCARD, 0)})
    assert env.scores() == [29000, 17000, 27000, 27000]
    v = env.peek()
    assert (v.honba, v.oya) == (1, 0)


def sc_claim_priority_pon_over_chi(make):
    """tests/env/rule_validation/test_claim_priority.py:11-56."""
    # (the reference pads the hands with repeated ids; here the padding is physically valid: <= 4 copies per type)
    env = setup(make(seed=1), hands=[[57] + list(range(0, 12)), [62, 65] + list(range(108, 119)),
                                    [56, 58] + list(range(120, 130)) + [131],
                                    [12, 16, 19, 21, 48, 59, 64, 77, 81, 89, 104, 130, 133]], current_player=0, active_players=[0],
                drawn_tile=100)
    env.step({0: pack_action(DISCARD, 57)})
    act, ph, dn = env.status()
    assert ph == WAIT_RESPONSE and act == 0b0110
    env.step({1: pack_action(CHI, 57, [62, 65]), 2: pack_action(PON, 57, [56, 58])})
    act, ph, dn = env.status()
    assert ph == WAIT_ACT and act == 0b0100
    assert evs(env)[-1]["type"] == "pon"


def sc_kuikae_suji(make):
    """tests/env/rule_validation/test_kuikae.py:7-42: after chi 1s with 2s3s, 4s (type 21) may not be discarded."""
    junk = [0, 12, 24, 36, 48, 60, 108, 112, 116]  # isolated tiles (the reference pads with nine copies of id 0)
    env = setup(make(), hands=[list(range(120, 133)), [72] + list(range(1, 12)) + [13], [79, 82, 85, 86] + junk,
                               [2, 3, 14, 15, 26, 27, 38, 39, 50, 51, 62, 63, 110]], current_player=1, wall=list(range(136)))
    env.step({1: pack_action(DISCARD, 72)})
    act, ph, dn = env.status()
    assert ph == WAIT_RESPONSE and (act >> 2) & 1
    assert find(env.legal(2), CHI) is not None
    acts = {s: pack_action(PASS) for s in range(4) if (act >> s) & 1}
    acts[2] = pack_action(CHI, 72, [79, 82])
    env.step(acts)
    for a in env.legal(2):
        t, tl, c = unpack_action(a)
        assert not (t == DISCARD and tl // 4 == 21)
        assert not (t == DISCARD and tl // 4 == 18)


def sc_kuikae_deadlock(make):
    """tests/env/rule_validation/test_kuikae.py:44-78 / src/tests.rs:274-310: chi is not offered when every
    remaining tile would be forbidden."""
    env = setup(make(), hands=[[0, 12, 24, 36], [72] + list(range(1, 12)) + [13], [79, 82, 85, 86],
                               [2, 3, 14, 15, 26, 27, 38, 39, 50, 51, 62, 63, 110]],
                melds=[[(PON_M, [108, 109, 110], True, 1, -1), (PON_M, [112, 113, 114], True, 1, -1), (PON_M, [116, 117, 118], True, 1, -1)], [], [], []],
                current_player=1, wall=list(range(136)))
    env.step({1: pack_action(DISCARD, 72)})
    act, ph, dn = env.status()
    assert ph == WAIT_ACT and act == 0b0100  # nobody could claim; seat 2 drew


def sc_sufuurenta(make):
    """tests/test_midway_draw.py:7-32."""
    scattered = [0, 4, 8, 36, 40, 44, 72, 76, 80, 112, 116, 120, 124]
    env = setup(make(), hands=[scattered[:] for _ in range(4)], wall=list(range(136)))
    winds = [108, 109, 110, 111]
    for i in range(4):
        v = env.peek()
        p = v.current_player
        pl = v.players[p]
        h = list(pl.hand[: pl.hand_len])
        h[0] = winds[i]
        for k, t in enumerate(h):
            pl.hand[k] = t
        v.drawn_tile = winds[i]
        env.poke(v)
        env.step({p: pack_action(DISCARD, winds[i])})
        assert bool(env.status()[2]) == (i == 3)
    assert any(e.get("reason") == "sufuurenta" for e in evs(env) if e["type"] == "ryukyoku")


def sc_suukansansen(make):
    """tests/test_midway_draw.py:34-56."""
    scattered = [0, 4, 8, 36, 40, 44, 72, 76, 80, 112, 116, 120, 124]
    env = setup(make(), hands=[scattered[:] for _ in range(4)],
                melds=[[(ANKAN_M, [0, 1, 2, 3], False), (ANKAN_M, [4, 5, 6, 7], False)],
                       [(ANKAN_M, [8, 9, 10, 11], False), (ANKAN_M, [12, 13, 14, 15], False)], [], []],
                current_player=1, drawn_tile=108, wall=list(range(136)))
    env.step({1: pack_action(DISCARD, 108)})
    assert env.status()[2] == 1
    assert any(e.get("reason") == "suukansansen" for e in evs(env) if e["type"] == "ryukyoku")


def sc_chankan_ron(make):
    """tests/env/agari/test_chankan.py:7-59: kakan of 1m, seat 1 waits on 1m-4m -> Ron offered, chankan yaku,
    single-round game ends."""
    h0 = tiles("02346789m01234p")
    env = setup(make(game_mode=0), hands=[h0, tiles("23m11p234s456s"), [], []],
                melds=[[(PON_M, [0, 1, 2], True, 1, -1)], [(PON_M, tiles("999m"), True, 0, -1)], [], []], drawn_tile=3)
    env.step({0: pack_action(KAKAN, 3, [0, 1, 2])})
    act, ph, dn = env.status()
    assert ph == WAIT_RESPONSE and act == 0b0010
    ron = find(env.legal(1), RON)
    assert ron is not None and unpack_action(ron)[1] == 3
    env.step({1: ron})
    assert env.status()[2] == 1
    hora = next(e for e in evs(env) if e["type"] == "hora")
    assert hora["actor"] == 1 and hora["target"] == 0
    # 2m3m +1m, 11p, 234s 456s, pon 999m, chankan only: 1 han 30 fu ko ron = 1000
    assert hora["deltas"] == [-1000, 1000, 0, 0]


def sc_chankan_pass(make):
    """tests/env/agari/test_chankan.py:61-110: chankan passed -> kakan resolves with a rinshan draw."""
    env = setup(make(game_mode=1), hands=[[4, 8, 12, 16, 20, 24, 28, 32, 36, 40], tiles("23m11p234s456s"), None, None],
                melds=[[(PON_M, [0, 1, 2], True, 1, -1)], [(PON_M, [132, 133, 134], True, 0, -1)], [], []], drawn_tile=3)
    env.step({0: pack_action(KAKAN, 3, [0, 1, 2])})
    act, ph, dn = env.status()
    assert ph == WAIT_RESPONSE and (act >> 1) & 1
    env.step({s: pack_action(PASS) for s in range(4) if (act >> s) & 1})
    act, ph, dn = env.status()
    assert ph == WAIT_ACT and act == 0b0001
    v = env.peek()
    assert v.players[1].missed_agari_doujun == 1  # state/mod.rs:902-917
    assert v.players[0].melds[0].meld_type == KAKAN_M and v.is_rinshan_flag == 1
    t = [e["type"] for e in evs(env)]
    assert t[-2:] == ["kakan", "tsumo"]


def _pao_setup(make, rule_bits, oya):
    env = make(rule_bits=rule_bits)
    setup(env, oya=oya, hands=[[132, 133, 0, 1, 2, 4, 5], [134] + list(range(40, 52)), None, None],
          melds=[[(PON_M, [124, 125, 126], True, 2, 124), (PON_M, [128, 129, 130], True, 3, 128)], [], [], []],
          current_player=1, active_players=[1], drawn_tile=None)
    env.step({1: pack_action(DISCARD, 134)})
    act, ph, dn = env.status()
    acts = {s: pack_action(PASS) for s in range(4) if (act >> s) & 1}
    acts[0] = pack_action(PON, 134, [132, 133])
    env.step(acts)
    assert env.peek().players[0].pao_daisangen == 1  # tests/env/agari/test_pao.py:45-46
    env.step({0: pack_action(DISCARD, 4)})
    return env


def sc_pao_daisangen_tsumo(make):
    """tests/env/agari/test_pao.py:7-76: oya daisangen tsumo with pao: liable seat pays all 48000."""
    env = _pao_setup(make, abi.RULE_TENHOU, 0)
    act, ph, dn = env.status()
    if ph == WAIT_RESPONSE:
        env.step({s: pack_action(PASS) for s in range(4) if (act >> s) & 1})
    v = env.peek()
    v.current_player = 0
    v.phase = WAIT_ACT
    v.active_mask = 1
    for p in range(1, 4):  # undo the draw the engine dealt to the next seat: keep 13-tile hands
        if v.players[p].hand_len == 14:
            v.players[p].hand_len = 13
    v.drawn_tile = 6
    pl = v.players[0]
    h = sorted(list(pl.hand[: pl.hand_len]) + [6])
    pl.hand_len = len(h)
    for i, t in enumerate(h):
        pl.hand[i] = t
    env.poke(v)
    env.step({0: find(env.legal(0), TSUMO)})
    hora = next(e for e in reversed(evs(env)) if e["type"] == "hora")
    assert hora["deltas"] == [48000, -48000, 0, 0]


def sc_pao_mjsoul_composite_tsumo(make):
    """src/tests.rs:1783-1884 (MjSoul liability-only split) realised on the state machine: ko winner (seat 0, oya 1)
    tsumo daisangen (pao by seat 3) + tsuuiisou: pao 32000 + ko share 8000 = 40000, oya 16000, other ko 8000."""
    env = make(rule_bits=abi.RULE_MJSOUL)
    # seat 0: pon haku, pon hatsu (from seat 2), pon chun from seat 3 -> pao 37 -> seat 3; hand EE SS + S wins
    hand0 = [132, 133, 108, 109, 112, 113, 114]
    setup(env, oya=1, hands=[hand0, list(range(60, 73)), None, [134] + list(range(40, 52))],
          melds=[[(PON_M, [124, 125, 126], True, 2, 124), (PON_M, [128, 129, 130], True, 2, 128)], [], [], []],
          current_player=3, active_players=[3])
    env.step({3: pack_action(DISCARD, 134)})
    act, ph, dn = env.status()
    acts = {s: pack_action(PASS) for s in range(4) if (act >> s) & 1}
    acts[0] = pack_action(PON, 134, [132, 133])
    env.step(acts)
    assert env.peek().players[0].pao_daisangen == 3
    env.step({0: pack_action(DISCARD, 114)})  # keep EE SS: shanpon-less tanki? -> 108,109,112,113 : wait E/S
    act, ph, dn = env.status()
    if ph == WAIT_RESPONSE:
        env.step({s: pack_action(PASS) for s in range(4) if (act >> s) & 1})
    v = env.peek()
    for p in range(1, 4):
        if v.players[p].hand_len == 14:
            v.players[p].hand_len = 13
    v.current_player = 0
    v.phase = WAIT_ACT
    v.active_mask = 1
    v.drawn_tile = 110
    pl = v.players[0]
    h = sorted(list(pl.hand[: pl.hand_len]) + [110])
    pl.hand_len = len(h)
    for i, t in enumerate(h):
        pl.hand[i] = t
    env.poke(v)
    ts = find(env.legal(0), TSUMO)
    assert ts is not None
    env.step({0: ts})
    hora = next(e for e in reversed(evs(env)) if e["type"] == "hora")
    # daisangen (pao) + tsuuiisou: total 2 yakuman
    assert hora["deltas"] == [64000, -16000, -8000, -40000]


def _poke_pao_case(make, hands, melds, pao, current_player, drawn_tile, riichi_sticks=0, rule_bits=abi.RULE_MJSOUL, game_mode=0, oya=0):
    """The state the tests of tests/env/test_majsoul_pao_scoring.py build with the reference's setters after
    RiichiEnv(seed=1, rule=default_mjsoul, game_mode="4p-red-single").reset(): oya 0, 25000 each, hands / melds / pao of the
    named seats replaced as given (hands stay in the given order, the drawn tile is NOT added to the hand), WaitAct for
    `current_player`.  pao: {winner: (field, liable seat)} with field "pao_daisangen" (yaku 37) or "pao_daisuushi" (yaku 50)."""
    env = make(game_mode=game_mode, seed=1, rule_bits=rule_bits)
    env.reset()
    v = env.peek()
    v.oya = oya
    for p in range(4):
        v.players[p].score = 25000 if game_mode < 3 else 35000
    for p, h in hands.items():
        v.players[p].hand_len = len(h)
        for i, t in enumerate(h):
            v.players[p].hand[i] = t
    for p, ms in melds.items():
        v.players[p].n_melds = len(ms)
        for i, (mt, mtiles, frm) in enumerate(ms):
            set_meld(v.players[p].melds[i], mt, mtiles, True, frm, -1)
    for p, (field, liable) in pao.items():
        setattr(v.players[p], field, liable)
    v.riichi_sticks = riichi_sticks
    v.current_player = current_player
    v.active_mask = 1 << current_player
    v.drawn_tile = drawn_tile
    v.needs_tsumo = 0
    v.phase = WAIT_ACT
    env.poke(v)
    return env


def _deltas(env):
    v = env.peek()
    return [v.players[p].score_delta for p in range(4)]


def sc_mjsoul_pao_tsumo_composite(make):
    """tests/env/test_majsoul_pao_scoring.py:5-83: dealer tsumo daisangen (pao by seat 2) + tsuuiisou under MjSoul rules: the
    liable seat pays the daisangen unit (48000) plus its normal share of the other yakuman (16000)."""
    env = _poke_pao_case(make, {0: [128, 129, 130, 132, 133, 134, 108, 109, 110, 112]}, {0: [(PON_M, [124, 125, 126], 1)]},
                         {0: ("pao_daisangen", 2)}, 0, 113)
    env.step({0: pack_action(TSUMO)})
    assert env.win_results()[0]["yakuman"]
    assert _deltas(env) == [96000, -16000, -64000, -16000]


def sc_mjsoul_pao_ron_composite(make):
    """tests/env/test_majsoul_pao_scoring.py:85-169: daisuushi (double, pao by seat 2) + tsuuiisou, Ron from seat 1: only the
    pao portion (96000) is split, the discarder pays 144000 - 48000."""
    melds0 = [(PON_M, [108, 109, 110], 1), (PON_M, [112, 113, 114], 1), (PON_M, [116, 117, 118], 1)]
    env = _poke_pao_case(make, {0: [120, 121, 124, 124], 1: [122, 1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13]}, {0: melds0},
                         {0: ("pao_daisuushi", 2)}, 1, 122)
    env.step({1: pack_action(DISCARD, 122)})
    assert env.status()[1] == WAIT_RESPONSE
    env.step({0: pack_action(RON, 122)})
    assert _deltas(env) == [144000, -96000, -48000, 0]


def sc_mjsoul_pao_ron_single(make):
    """tests/env/test_majsoul_pao_scoring.py:171-241: a single yakuman with pao, Ron: split 50/50 like Tenhou."""
    melds0 = [(PON_M, [128, 129, 130], 1), (PON_M, [132, 133, 134], 1), (PON_M, [124, 125, 126], 1)]
    env = _poke_pao_case(make, {0: [0, 1, 4, 4], 1: [4, 1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13]}, {0: melds0},
                         {0: ("pao_daisangen", 2)}, 1, 4)
    env.step({1: pack_action(DISCARD, 4)})
    assert env.status()[1] == WAIT_RESPONSE
    env.step({0: pack_action(RON, 4)})
    assert _deltas(env) == [48000, -24000, -24000, 0]


def _mjsoul_real_record(make, sticks):
    melds2 = [(PON_M, [124, 125, 126], 3), (PON_M, [116, 117, 118], 1), (PON_M, [128, 129, 130], 1), (PON_M, [132, 133, 134], 3)]
    env = _poke_pao_case(make, {2: [112], 0: [113, 1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13]}, {2: melds2}, {2: ("pao_daisangen", 3)}, 0, 113,
                         riichi_sticks=sticks)
    env.step({0: pack_action(DISCARD, 113)})
    assert env.status()[1] == WAIT_RESPONSE
    env.step({2: pack_action(RON, 113)})
    return env


def sc_mjsoul_pao_ron_real_record(make):
    """tests/env/test_majsoul_pao_scoring.py:243-315 (MjSoul game 251122-9051f1e9, round 11): ko Ron, daisangen (pao by seat 3) +
    tsuuiisou = 64000: the pao portion 32000 is halved, the discarder pays 48000."""
    assert _deltas(_mjsoul_real_record(make, 0)) == [-48000, 0, 64000, -16000]


def sc_mjsoul_pao_ron_real_record_with_riichi_stick(make):
    """tests/env/test_majsoul_pao_scoring.py:317-372: the same with one riichi stick on the table: the winner collects it."""
    assert _deltas(_mjsoul_real_record(make, 1)) == [-48000, 0, 65000, -16000]


def _kokushi_ankan_state(make, rule_bits):
    """setup_kokushi_scenario (tests/env/test_rules_chankan.py:4-45): on the constructor's first round (no reset), seat 0 holds
    four 9p (the fourth just drawn) + ten low tiles, seat 1 a kokushi tenpai waiting on 9p, seats 2 and 3 empty hands."""
    env = make(rule_bits=rule_bits)
    v = env.peek()
    _set_hand(v.players[0], sorted([68, 69, 70] + list(range(10)) + [71]))
    _set_hand(v.players[1], sorted(t * 4 for t in [0, 8, 9, 18, 26, 27, 28, 29, 30, 31, 32, 33] + [0]))
    _set_hand(v.players[2], [])
    _set_hand(v.players[3], [])
    v.current_player = 0
    v.active_mask = 1
    v.drawn_tile = 71
    v.phase = WAIT_ACT
    v.needs_tsumo = 0
    env.poke(v)
    return env


def sc_rules_chankan_kokushi_tenhou(make):
    """tests/env/test_rules_chankan.py:48-66: Tenhou rules: an Ankan cannot be robbed, not even by kokushi: the game goes on with
    seat 0's replacement draw.  (The action names another copy of the quad than the generated one: accepted, quirk Q13.)"""
    env = _kokushi_ankan_state(make, abi.RULE_TENHOU)
    env.step({0: pack_action(ANKAN, 71, [68, 69, 70, 71])})
    v = env.peek()
    assert v.current_player == 0 and v.phase == WAIT_ACT and v.drawn_tile >= 0


def sc_rules_chankan_kokushi_mjsoul(make):
    """tests/env/test_rules_chankan.py:69-90: MjSoul rules: kokushi may rob the Ankan: WaitResponse, seat 1 is offered Ron on 71."""
    env = _kokushi_ankan_state(make, abi.RULE_MJSOUL)
    env.step({0: pack_action(ANKAN, 71, [68, 69, 70, 71])})
    act, ph, _ = env.status()
    assert ph == WAIT_RESPONSE and (act >> 1) & 1
    ron = [a for a in env.legal(1) if unpack_action(a)[0] == RON]
    assert ron and unpack_action(ron[0])[1] == 71


def sc_rules_standard_chankan_kakan(make):
    """tests/env/test_rules_chankan.py:93-170: the ordinary chankan (Ron on an added kan) works under Tenhou rules as well."""
    env = make(rule_bits=abi.RULE_TENHOU)
    v = env.peek()
    _set_hand(v.players[0], sorted(list(range(10)) + [71]))
    v.players[0].n_melds = 1
    set_meld(v.players[0].melds[0], PON_M, [68, 69, 70], True, -1, -1)
    p1 = []
    for t in (0, 4, 8):
        p1 += [t * 4, t * 4 + 1, t * 4 + 2]
    _set_hand(v.players[1], sorted(p1 + [48, 49, 60, 64]))
    _set_hand(v.players[2], [])
    _set_hand(v.players[3], [])
    v.current_player = 0
    v.active_mask = 1
    v.drawn_tile = 71
    v.phase = WAIT_ACT
    v.needs_tsumo = 0
    env.poke(v)
    env.step({0: pack_action(KAKAN, 71, [68, 69, 70])})
    act, ph, _ = env.status()
    assert ph == WAIT_RESPONSE and (act >> 1) & 1
    ron = [a for a in env.legal(1) if unpack_action(a)[0] == RON]
    assert ron and unpack_action(ron[0])[1] == 71


def sc_game_modes_initialization_params(make):
    """tests/env/test_game_modes.py:9-32: reset(scores, kyotaku, honba) of a 4p-red-single game shows in the state and in the
    start_kyoku event."""
    env = make(game_mode=0, round_wind=0)
    env.reset(scores=[30000] * 4, kyotaku=1, honba=2)
    v = env.peek()
    assert [v.players[p].score for p in range(4)] == [30000] * 4 and v.riichi_sticks == 1
    sk = evs(env)[1]
    assert sk["type"] == "start_kyoku" and sk["bakaze"] == "E" and sk["honba"] == 2 and sk["kyotaku"] == 1


def sc_game_modes_south_round_wind(make):
    """tests/env/test_game_modes.py:34-43: round_wind=1 -> bakaze S"""
    env = make(game_mode=0, round_wind=1)
    env.reset(round_wind=1)
    assert evs(env)[1]["bakaze"] == "S"


def sc_riichi_setup_leaves_only_discards(make):
    """tests/env/actions/test_riichi_autoplay_pass.py:14-44: after the Riichi declaration the seat's list holds discards only,
    among them the drawn 5p and the 1z (every discard is listed, also those of a false riichi)."""
    env = setup(make(seed=42), hands=[tiles("111222333444m1z"), tiles("111222333444p2z"), tiles("111222333444s3z"), tiles("555666777888m4z")],
                current_player=0, drawn_tile=tiles("5p")[0])
    env.step({0: pack_action(RIICHI)})
    legal = [unpack_action(a) for a in env.legal(0)]
    assert legal and all(t == DISCARD for t, _, _ in legal)
    assert {tiles("5p")[0], tiles("1z")[0]} <= {tl for _, tl, _ in legal}


_NO_CLAIM_HANDS = [[0, 1, 4, 5, 8, 20, 20, 20, 20, 20, 20, 20, 20], [23, 35, 38, 61, 69, 70, 79, 83, 98, 123, 127, 128, 130],
                   [1, 4, 5, 8, 12, 17, 56, 59, 81, 94, 101, 106, 122]]


def sc_riichi_no_pon_claim(make):
    """tests/env/actions/test_riichi_no_claim.py:15-32: a seat in riichi is not offered Pon: the turn passes to it."""
    env = setup(make(seed=42), hands=_NO_CLAIM_HANDS + [[0] * 12 + [2]], current_player=3, active_players=[3],
                riichi_declared=[True, False, False, False], drawn_tile=2)
    env.step({3: pack_action(DISCARD, 2)})
    act, ph, _ = env.status()
    assert ph == WAIT_ACT and act == 1


def sc_riichi_no_chi_claim(make):
    """tests/env/actions/test_riichi_no_claim.py:34-54: a seat in riichi is offered Ron but not Chi."""
    env = setup(make(seed=42), hands=_NO_CLAIM_HANDS + [[0] * 12 + [11]], current_player=3, active_players=[3],
                riichi_declared=[True, False, False, False], drawn_tile=11)
    env.step({3: pack_action(DISCARD, 11)})
    act, ph, _ = env.status()
    assert ph == WAIT_RESPONSE and act == 1
    kinds = [unpack_action(a)[0] for a in env.legal(0)]
    assert CHI not in kinds and RON in kinds


def _riichi_autoplay(make, others, consistent=False):
    env = make(seed=42)
    env.reset()
    v = env.peek()
    v.players[0].riichi_declared = 1
    v.current_player = 3
    if consistent:
        v.active_mask = 1 << 3
    _set_hand(v.players[0], [1, 5, 9, 13, 17, 21, 25, 29, 33, 37, 41, 45, 49])
    _set_hand(v.players[1], others[0])
    _set_hand(v.players[2], others[1])
    _set_hand(v.players[3], others[2] + [100])
    v.wall_len = 40
    for i, t in enumerate([101, 102, 103, 104] * 10):
        v.wall[i] = t
    v.drawn_tile = -1
    env.poke(v)
    env.step({3: pack_action(DISCARD, 100)})
    v = env.peek()
    assert v.current_player == 0
    dt = v.drawn_tile
    assert dt >= 0
    env.step({0: pack_action(DISCARD, dt)})
    act, ph, _ = env.status()
    assert act == 2 and env.peek().current_player == 1
    e = evs(env)
    assert (e[-1]["type"], e[-1]["actor"]) == ("tsumo", 1)
    assert (e[-2]["type"], e[-2]["actor"], e[-2]["tsumogiri"]) == ("dahai", 0, True)
    assert (e[-3]["type"], e[-3]["actor"]) == ("tsumo", 0)


def sc_riichi_autoplay_waits_for_the_discard(make):
    """tests/test_riichi_autoplay.py:4-95: a seat in riichi that draws a dead tile is still asked for its discard (no auto-play);
    the discard is logged as tsumogiri and the next seat draws.  State as the test pokes it: seat 0 in riichi with a garbage
    hand, the wall replaced by 40 tiles, seat 3 (not the acting seat of the reset state) discards 100; the other seats hold
    thirteen copies of tile 0 (oracle only, see SCENARIOS_ORACLE_ONLY)."""
    _riichi_autoplay(make, [[0] * 13, [0] * 13, [0] * 12])


def sc_riichi_autoplay_waits_for_the_discard_possible_hands(make):
    """The same flow with hands a game can hold (other copies of 1m..4p for the seats that only watch) and with the discarder
    also named in active_players: runs on the GPU too.  (The reference test leaves active_players = [0] while seat 3 acts; the
    reference regenerates the legal actions of whoever sends an action, the HIP path validates against the lists it published
    for the active seats - DESIGN.md section 6, Q17.)"""
    _riichi_autoplay(make, [list(range(2, 54, 4)), list(range(3, 55, 4)), list(range(0, 48, 4))], consistent=True)


def sc_riichi_autoplay_unnamed_current_player(make):
    """tests/test_riichi_autoplay.py:4-95 as the reference test pokes it - active_players stays [0] while seat 3 is the current player
    and sends the discard: the reference regenerates the legal actions of whoever sends an action (state/mod.rs:339-402), so the discard
    is legal (quirk Q17).  Hands a game can hold (the reference's thirteen copies of one tile are outside the HIP path's 3-bit counters)."""
    _riichi_autoplay(make, [list(range(2, 54, 4)), list(range(3, 55, 4)), list(range(0, 48, 4))], consistent=False)


_KOKUSHI_WAIT_E = sorted([0, 32, 36, 68, 72, 104, 112, 116, 120, 124, 128, 132] + [1])   # thirteen kinds but East, 1m paired


def _ankan_east_state(make, game_mode, rule_bits, seat1_hand, seat1_melds=None):
    """tests/env/agari/test_chankan.py:101-219: after reset(), seat 0 holds EEE + ten low tiles and has drawn the fourth East (111),
    seat 1 the given hand; the other seats keep what the wall dealt."""
    env = make(seed=42, game_mode=game_mode, rule_bits=rule_bits)
    env.reset()
    v = env.peek()
    _set_hand(v.players[0], [108, 109, 110] + [4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 111])
    v.drawn_tile = 111
    _set_hand(v.players[1], seat1_hand)
    if seat1_melds:
        v.players[1].n_melds = len(seat1_melds)
        for i, m in enumerate(seat1_melds):
            set_meld(v.players[1].melds[i], *m)
    v.current_player = 0
    v.phase = WAIT_ACT
    v.active_mask = 1
    env.poke(v)
    return env


def sc_kokushi_ankan_ron(make):
    """tests/env/agari/test_chankan.py:101-147: MjSoul rules: kokushi robs the Ankan of East and wins (yaku 42), the game ends."""
    env = _ankan_east_state(make, 0, abi.RULE_MJSOUL, _KOKUSHI_WAIT_E)
    env.step({0: pack_action(ANKAN, 111, [108, 109, 110, 111])})
    act, ph, _ = env.status()
    assert ph == WAIT_RESPONSE and (act >> 1) & 1
    ron = [a for a in env.legal(1) if unpack_action(a)[0] == RON]
    assert ron and unpack_action(ron[0])[1] == 111
    env.step({1: ron[0]})
    assert env.status()[2]
    w = env.win_results()
    assert w[1]["is_win"] and 42 in w[1]["yaku"]


def sc_kokushi_ankan_ron_tenhou(make):
    """tests/env/agari/test_chankan.py:149-183: Tenhou rules: no Ron on an Ankan, seat 0 goes on."""
    env = _ankan_east_state(make, 1, abi.RULE_TENHOU, _KOKUSHI_WAIT_E)
    env.step({0: pack_action(ANKAN, 111, [108, 109, 110, 111])})
    act, ph, _ = env.status()
    assert ph == WAIT_ACT and act == 1


def sc_non_kokushi_ankan_no_ron(make):
    """tests/env/agari/test_chankan.py:185-219: an ordinary hand waiting on the tile cannot rob an Ankan: the kan resolves at once."""
    env = _ankan_east_state(make, 1, abi.RULE_TENHOU, [112, 113, 116, 116, 120, 120, 124, 124, 128, 128],
                            [(PON_M, [132, 133, 134], True, -1, -1)])
    env.step({0: pack_action(ANKAN, 111, [108, 109, 110, 111])})
    v = env.peek()
    assert v.phase == WAIT_ACT and v.current_player == 0 and v.players[0].n_melds == 1
    assert v.players[0].melds[0].meld_type == ANKAN_M and v.drawn_tile >= 0


def _four_1m_state(make, riichi):
    env = make(seed=42)
    env.reset()
    v = env.peek()
    _set_hand(v.players[0], [0, 1, 2, 3] + list(range(12, 22)))
    v.drawn_tile = 3
    v.active_mask = 1
    v.current_player = 0
    v.phase = WAIT_ACT
    if riichi:
        v.players[0].riichi_declared = 1
    env.poke(v)
    return env


def sc_ankan_generation(make):
    """tests/env/agari/test_chankan.py:221-243: four 1m in hand: the list offers the Ankan with all four tiles."""
    env = _four_1m_state(make, False)
    ankan = [unpack_action(a) for a in env.legal(0) if unpack_action(a)[0] == ANKAN]
    assert ankan and ankan[0][1] in (0, 1, 2, 3) and sorted(ankan[0][2]) == [0, 1, 2, 3]


def sc_ankan_generation_riichi(make):
    """tests/env/agari/test_chankan.py:245-268: the same after a riichi declaration (the kan of the drawn type that keeps the waits)."""
    env = _four_1m_state(make, True)
    ankan = [unpack_action(a) for a in env.legal(0) if unpack_action(a)[0] == ANKAN]
    assert ankan and ankan[0][1] == 0 and sorted(ankan[0][2]) == [0, 1, 2, 3]


def sc_chankan_stale_claims_repro(make):
    """tests/env/agari/test_chankan.py:270-330 (match 27, step 267): seat 0 passes a Pon offer, then seat 3 adds 6p to its Pon: seat 0,
    waiting on 3p-6p, must be offered Ron - its observation holds the stale Pon entry plus the Ron (stale current_claims, section 6)."""
    env = setup(make(seed=42), hands=[[4, 5, 6, 8, 9, 10, 12, 13, 14, 61, 62, 49, 53], [], [63, 64, 65, 66, 67, 68, 69, 70, 71, 72, 73, 74, 75], []],
                melds=[[], [], [], [(PON_M, [56, 57, 58], True, -1, -1)]], discards=[[60], [], [], []], current_player=2)
    env.step({2: pack_action(DISCARD, 63)})
    act, ph, _ = env.status()
    assert (act >> 0) & 1
    env.step({s: pack_action(PASS) for s in range(4) if (act >> s) & 1})
    v = env.peek()
    assert v.current_player == 3
    v.drawn_tile = 59
    _set_hand(v.players[3], [59])
    env.poke(v)
    env.step({3: pack_action(KAKAN, 59, [56, 57, 58])})
    act, ph, _ = env.status()
    assert (act >> 0) & 1, (act, ph)
    assert RON in [unpack_action(a)[0] for a in env.legal(0)]


def sc_env_scoring_ron_deltas(make):
    """tests/test_env_scoring.py:6-47: Ron on the discarded haku: zero-sum deltas in the hora event, scores updated."""
    env = make(seed=42)
    env.reset()
    v = env.peek()
    _set_hand(v.players[0], sorted([124, 125] + [0, 4, 8, 5, 9, 12, 16, 20, 24, 14, 15]))
    v.current_player = 1
    v.phase = WAIT_ACT
    v.active_mask = 2
    h1 = list(v.players[1].hand[: v.players[1].hand_len]) + [126]
    _set_hand(v.players[1], h1)
    if v.players[0].hand_len == 14:
        pass
    env.poke(v)
    env.step({1: pack_action(DISCARD, 126)})
    act, ph, _ = env.status()
    assert (act >> 0) & 1 and find(env.legal(0), RON) is not None
    env.step({0: pack_action(RON, 126)})
    hora = next(e for e in reversed(evs(env)) if e["type"] == "hora")
    d = hora["deltas"]
    assert d[1] < 0 < d[0] and sum(d) == 0
    v = env.peek()
    assert v.players[0].score == 25000 + d[0] and v.players[1].score == 25000 + d[1]


def sc_env_scoring_tsumo_deltas(make):
    """tests/test_env_scoring.py:49-85: a tsumo (haku triplet) outside the first turn: the event is flagged tsumo, everybody pays."""
    env = make(seed=42)
    env.reset()
    v = env.peek()
    _set_hand(v.players[0], sorted([124, 125, 126, 0, 4, 8, 5, 9, 13, 16, 20, 24, 12]))
    v.drawn_tile = 14
    v.current_player = 0
    v.is_first_turn = 0
    v.players[0].discards[v.players[0].n_discards] = 0
    v.players[0].n_discards += 1
    env.poke(v)
    assert find(env.legal(0), TSUMO) is not None
    env.step({0: pack_action(TSUMO)})
    hora = next(e for e in reversed(evs(env)) if e["type"] == "hora")
    d = hora["deltas"]
    assert hora["tsumo"] is True and d[0] > 0 and all(x < 0 for x in d[1:]) and sum(d) == 0
    assert env.peek().players[0].score == 25000 + d[0]


def sc_env_scoring_ura_markers(make):
    """tests/test_env_scoring.py:87-107: a tsumo in riichi shows the ura markers in the hora event."""
    env = make(seed=42)
    env.reset()
    v = env.peek()
    v.players[0].riichi_declared = 1
    _set_hand(v.players[0], sorted([124, 125, 126, 0, 1, 2, 4, 5, 6, 8, 9, 10, 12]))
    v.drawn_tile = 13
    v.current_player = 0
    env.poke(v)
    env.step({0: pack_action(TSUMO)})
    hora = next(e for e in reversed(evs(env)) if e["type"] == "hora")
    assert len(hora["ura_markers"]) > 0


def _poked_round(make, hands, current_player=0, drawn_tile=None, mutate=None, game_mode=0):
    """The fixture of tests/env/test_riichienv.py:197-650: RiichiEnv(seed=42).reset(), then hands replaced through the setter (given
    order kept sorted like the tests do), active_players = [current_player], drawn_tile set WITHOUT touching the hand.  Seats the
    reference test leaves to the wall get hands that cannot call anything here (the walls differ, section 6): honours and
    terminals of other suits, stated per scenario."""
    env = make(seed=42, game_mode=game_mode)
    env.reset()
    v = env.peek()
    for p, h in hands.items():
        _set_hand(v.players[p], h)
    v.current_player = current_player
    v.active_mask = 1 << current_player
    v.phase = WAIT_ACT
    if drawn_tile is not None:
        v.drawn_tile = drawn_tile
    if mutate:
        mutate(v)
    env.poke(v)
    return env


_IDLE2 = [73, 77, 81, 85, 89, 93, 97, 101, 105, 109, 113, 117, 121]      # sou / honours: no 1m-5m, no pairs
_IDLE3 = [74, 78, 82, 86, 90, 94, 98, 102, 106, 110, 114, 118, 122]


def sc_env_initialization(make):
    """tests/env/test_riichienv.py:8-64: after reset(): 83 wall tiles (69 live + 14 dead), 14 / 13 / 13 / 13 tiles, nothing melded
    or discarded, the dealer acts in WaitAct with fourteen discards to choose from, its log so far is start_game, start_kyoku, tsumo."""
    env = make(seed=42)
    v = env.peek()
    assert v.wall_len > 0
    env.reset()
    v = env.peek()
    assert v.wall_len == 83
    assert [v.players[p].hand_len for p in range(4)] == [14, 13, 13, 13]
    assert all(v.players[p].n_melds == 0 and v.players[p].n_discards == 0 for p in range(4))
    assert (v.current_player, v.turn_count, v.is_done, v.needs_tsumo) == (0, 0, 0, 0)
    act, ph, dn = env.status()
    assert (act, ph, dn) == (1, WAIT_ACT, 0)
    assert [e["type"] for e in evs(env, 0)] == ["start_game", "start_kyoku", "tsumo"]
    legal = [unpack_action(a) for a in env.legal(0)]
    assert len(legal) == 14 and legal[0][0] == DISCARD and 0 <= legal[0][1] < 136 and legal[0][2] == []


def sc_env_basic_step_processing(make):
    """tests/env/test_riichienv.py:66-127: discard, pass whatever is offered, the next seat draws; its own log masks the others' hands
    and draws, the environment's log does not."""
    env = make(seed=42)
    env.reset()
    v = env.peek()
    env.step({0: pack_action(DISCARD, v.players[0].hand[v.players[0].hand_len - 1])})
    passed = 0
    while env.status()[1] == WAIT_RESPONSE:
        act = env.status()[0]
        env.step({s: pack_action(PASS) for s in range(4) if (act >> s) & 1})
        passed += 1
    act, ph, dn = env.status()
    v = env.peek()
    assert (ph, v.current_player, act, dn) == (WAIT_ACT, 1, 2, 0) and v.players[1].hand_len == 14 and v.drawn_tile >= 0
    e1 = evs(env, 1)
    assert [e["type"] for e in e1[:3]] == ["start_game", "start_kyoku", "tsumo"]
    th = e1[1]["tehais"]
    assert th[0][0] == "?" and th[1][0] != "?" and th[2][0] == "?" and th[3][0] == "?"
    assert e1[2]["actor"] == 0 and e1[2]["pai"] == "?"
    full = evs(env)
    assert [e["type"] for e in full] == ["start_game", "start_kyoku", "tsumo", "dahai", "tsumo"]
    assert [(e["actor"], e["pai"] != "?") for e in full[2:]] == [(0, True), (0, True), (1, True)]


def sc_env_pon_claim(make):
    """tests/env/test_riichienv.py:197-239: seat 2 holds two 1m: Pon of seat 0's 1m, then it is seat 2's turn; the pon event."""
    env = _poked_round(make, {0: [0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48], 1: [13, 15, 17, 19, 21, 22, 23, 25, 26, 27, 29, 30, 31],
                              2: sorted([1, 2, 57, 61, 65, 69, 73, 77, 81, 85, 89, 93, 97]),
                              3: [101, 105, 109, 113, 117, 121, 125, 129, 133, 6, 10, 14, 18]}, drawn_tile=52)
    env.step({0: pack_action(DISCARD, 0)})
    act, ph, _ = env.status()
    assert ph == WAIT_RESPONSE and act == 4
    assert find(env.legal(2), PON) is not None
    env.step({2: pack_action(PON, 0, [1, 2])})
    v = env.peek()
    assert v.current_player == 2 and v.phase == WAIT_ACT
    last = evs(env)[-1]
    assert (last["type"], last["actor"], last["target"], last["pai"]) == ("pon", 2, 0, "1m")


def sc_env_pon_red_dora_claim(make):
    """tests/env/test_riichienv.py:241-280: Pon of the discarded red 5m with two plain 5m (seat 1 empty; seat 3 made harmless)."""
    env = _poked_round(make, {0: [0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48], 1: [],
                              2: sorted([17, 18, 21, 25, 29, 33, 37, 41, 45, 49, 53, 57, 61]), 3: _IDLE3}, drawn_tile=16)
    env.step({0: pack_action(DISCARD, 16)})
    act, ph, _ = env.status()
    assert ph == WAIT_RESPONSE and act == 4
    pon = find(env.legal(2), PON)
    assert pon is not None
    env.step({2: pon})
    v = env.peek()
    assert v.current_player == 2 and v.phase == WAIT_ACT
    last = evs(env)[-1]
    assert (last["type"], last["actor"], last["target"], last["pai"]) == ("pon", 2, 0, "5mr")


def _chi_state(make, h1):
    return _poked_round(make, {0: [8] + list(range(40, 52)), 1: sorted(h1), 2: _IDLE2, 3: _IDLE3}, drawn_tile=100)


def sc_env_chi_claim(make):
    """tests/env/test_riichienv.py:282-327: the next seat holds 4m 5m: Chi of the discarded 3m."""
    env = _chi_state(make, [12, 16] + list(range(60, 71)))
    env.step({0: pack_action(DISCARD, 8)})
    act, ph, _ = env.status()
    assert ph == WAIT_RESPONSE and (act >> 1) & 1
    assert find(env.legal(1), CHI) is not None
    env.step({1: pack_action(CHI, 8, [12, 16])})
    v = env.peek()
    assert v.current_player == 1 and v.phase == WAIT_ACT and evs(env)[-1]["type"] == "chi"


def sc_env_chi_claim_with_red_dora(make):
    """tests/env/test_riichienv.py:329-421: with 4m, 5m and 5mr in hand two Chi are offered and the seat chooses the five it uses."""
    for five, name in ((17, "5m"), (16, "5mr")):
        env = _chi_state(make, [12, 16, 17] + list(range(60, 71)))
        env.step({0: pack_action(DISCARD, 8)})
        act, ph, _ = env.status()
        assert ph == WAIT_RESPONSE and (act >> 1) & 1
        assert sum(unpack_action(a)[0] == CHI for a in env.legal(1)) == 2
        env.step({1: pack_action(CHI, 8, [12, five])})
        v = env.peek()
        assert v.current_player == 1 and v.phase == WAIT_ACT
        last = evs(env)[-1]
        assert (last["type"], last["pai"], last["consumed"]) == ("chi", "3m", ["4m", name])


def _illegal_ryukyoku(env):
    r = next(e for e in reversed(evs(env)) if e["type"] == "ryukyoku")
    assert "Error: Illegal Action" in r["reason"]


def sc_env_chi_claim_with_invalid_tile(make):
    """tests/env/test_riichienv.py:423-480: a Chi that names a tile the seat does not hold is an illegal action (penalty ryukyoku)."""
    env = _chi_state(make, [12, 16, 17] + list(range(60, 71)))
    env.step({0: pack_action(DISCARD, 8)})
    assert sum(unpack_action(a)[0] == CHI for a in env.legal(1)) == 2
    env.step({1: pack_action(CHI, 8, [12, 10])})
    _illegal_ryukyoku(env)


def sc_env_chi_claim_with_invalid_combo(make):
    """tests/env/test_riichienv.py:482-512: tiles in hand that form no sequence with the discard: illegal action."""
    env = _poked_round(make, {0: [8] + list(range(40, 52)), 1: sorted([12, 60] + list(range(61, 72))), 2: _IDLE2, 3: _IDLE3})
    env.step({0: pack_action(DISCARD, 8)})
    env.step({1: pack_action(CHI, 8, [12, 60])})
    _illegal_ryukyoku(env)


def sc_env_chi_multiple_patterns(make):
    """tests/env/test_riichienv.py:514-555: 123456789m + 5mr against a discarded 4m: five Chi (23, 35, 3-5r, 56, 5r-6)."""
    env = _poked_round(make, {0: [13] + list(range(40, 52)), 1: sorted([0, 4, 8, 12, 16, 17, 20, 24, 28, 32] + [100, 104, 108]),
                              2: _IDLE2, 3: _IDLE3}, drawn_tile=100)
    env.step({0: pack_action(DISCARD, 13)})
    chis = [set(unpack_action(a)[2]) for a in env.legal(1) if unpack_action(a)[0] == CHI]
    assert len(chis) == 5
    for want in ({4, 8}, {8, 17}, {8, 16}, {17, 20}, {16, 20}):
        assert want in chis


def sc_env_ron_claim(make):
    """tests/env/test_riichienv.py:557-597: 111222333444m + 5m waits on 5m: one Ron action, a hora event follows."""
    env = _poked_round(make, {1: [0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14, 16], 0: sorted([17] + list(range(40, 53))), 2: _IDLE2, 3: _IDLE3})
    env.step({0: pack_action(DISCARD, 17)})
    act, ph, _ = env.status()
    assert ph == WAIT_RESPONSE and act == 2
    ron = [a for a in env.legal(1) if unpack_action(a)[0] == RON]
    assert len(ron) == 1
    env.step({1: ron[0]})
    assert "hora" in [e["type"] for e in evs(env)[-3:]]


def sc_env_ankan_riichi_legality(make):
    """tests/env/test_riichienv.py:599-640: in riichi with 1s 1s 1s (+ drawn 1s) 2s 3s the Ankan would change the waits: not offered."""
    def riichi3(v):
        v.players[3].riichi_declared = 1

    env = _poked_round(make, {3: sorted([4, 5, 36, 60, 64, 68, 72, 73, 74, 75, 76, 80, 88, 92])}, current_player=3, drawn_tile=72, mutate=riichi3)
    assert not [a for a in env.legal(3) if unpack_action(a)[0] == ANKAN]


def _daiminkan_state(make, melds0, discarder, tile, consumed):
    """riichienv-core/src/tests.rs:1594-1743: seat 0 holds open Pon melds and the three further copies of `tile`; `discarder` has
    just discarded `tile` and seat 0 is offered the Daiminkan (the Rust tests call _resolve_kan directly)."""
    env = make(game_mode=2, seed=1)
    env.reset()
    v = env.peek()
    n = 13 - 3 * len(melds0) - 3
    _set_hand(v.players[0], sorted(consumed + [t for t in (0, 5, 10, 40, 45, 50, 80)][:n]))
    v.players[0].n_melds = len(melds0)
    for i, (mt, frm) in enumerate(melds0):
        set_meld(v.players[0].melds[i], PON_M, mt, True, frm, mt[0])
    for p in range(1, 4):
        if v.players[p].hand_len == 14:
            v.players[p].hand_len = 13
    v.drawn_tile = -1
    v.phase = WAIT_RESPONSE
    v.active_mask = 1
    v.current_player = discarder
    v.last_discard_pid = discarder
    v.last_discard_tile = tile
    env.poke(v)
    kan = find(env.legal(0), DAIMINKAN)
    assert kan is not None
    env.step({0: pack_action(DAIMINKAN, tile, consumed)})
    return env.peek().players[0]


def sc_daiminkan_pao_daisangen(make):
    """tests.rs:1594-1644: the third dragon set completed by a Daiminkan makes the discarder liable (yaku 37)."""
    p0 = _daiminkan_state(make, [([124, 125, 126], 1), ([128, 129, 130], 2)], 3, 132, [133, 134, 135])
    assert p0.pao_daisangen == 3 and p0.pao_daisuushi == -1


def sc_daiminkan_pao_daisuushii(make):
    """tests.rs:1646-1701: the fourth wind set by Daiminkan: liable for daisuushii (yaku 50)."""
    p0 = _daiminkan_state(make, [([108, 109, 110], 1), ([112, 113, 114], 2), ([116, 117, 118], 3)], 2, 120, [121, 122, 123])
    assert p0.pao_daisuushi == 2 and p0.pao_daisangen == -1


def sc_daiminkan_no_pao_insufficient_melds(make):
    """tests.rs:1703-1742: only the second dragon set: nobody is liable."""
    p0 = _daiminkan_state(make, [([124, 125, 126], 1)], 1, 128, [129, 130, 131])
    assert p0.pao_daisangen == -1 and p0.pao_daisuushi == -1


def sc_tenhou_tsumo_pao_composite(make):
    """tests.rs:1886-1917 on the state machine: Tenhou rules, ko tsumo of daisangen (pao by seat 3) + tsuuiisou: the liable seat
    pays both yakuman (64000), nobody else pays."""
    env = _poke_pao_case(make, {0: [128, 129, 130, 132, 133, 134, 108, 109, 110, 112]}, {0: [(PON_M, [124, 125, 126], 1)]},
                         {0: ("pao_daisangen", 3)}, 0, 113, rule_bits=abi.RULE_TENHOU, oya=1)
    env.step({0: pack_action(TSUMO)})
    assert _deltas(env) == [64000, 0, 0, -64000]


def sc_tenhou_ron_pao_composite(make):
    """tests.rs:1919-1950 (Tenhou: the WHOLE yakuman total is split in halves) on the state machine: dealer Ron of daisuushii (pao
    by seat 1) + tsuuiisou from seat 2.  Tenhou counts daisuushii once (no double yakuman, rule.rs), so the total is 2 x 48000."""
    melds0 = [(PON_M, [108, 109, 110], 1), (PON_M, [112, 113, 114], 1), (PON_M, [116, 117, 118], 1)]
    env = _poke_pao_case(make, {0: [120, 121, 124, 124], 2: [122, 1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13]}, {0: melds0},
                         {0: ("pao_daisuushi", 1)}, 2, 122, rule_bits=abi.RULE_TENHOU)
    env.step({2: pack_action(DISCARD, 122)})
    assert env.status()[1] == WAIT_RESPONSE
    env.step({0: pack_action(RON, 122)})
    assert _deltas(env) == [96000, -48000, -48000, 0]


def sc_mjsoul_3p_ron_pao_composite(make):
    """tests.rs:1951-1986 on the state machine: sanma, MjSoul rules, dealer Ron of daisangen (pao by seat 1) + tsuuiisou from seat
    2: only the daisangen unit is halved: 24000 / 72000."""
    melds0 = [(PON_M, [124, 125, 126], 1), (PON_M, [128, 129, 130], 1), (PON_M, [132, 133, 134], 1)]
    env = _poke_pao_case(make, {0: [108, 109, 112, 113], 2: [114, 36, 37, 40, 41, 44, 45, 48, 49, 53, 54, 56, 57]}, {0: melds0},
                         {0: ("pao_daisangen", 1)}, 2, 114, game_mode=3)
    env.step({2: pack_action(DISCARD, 114)})
    assert env.status()[1] == WAIT_RESPONSE
    env.step({0: pack_action(RON, 114)})
    assert _deltas(env)[:3] == [96000, -24000, -72000]


def sc_ryukyoku_deltas_are_reset_each_round_4p(make):
    """tests.rs:234-261: score deltas of an earlier settlement do not leak into the ryukyoku event of the next round (four East
    discards on the first turn: sufuurenta)."""
    env = make()
    env.reset()
    v = env.peek()
    for p, d in enumerate([2300, -2300, 0, 0]):
        v.players[p].score_delta = d
    env.poke(v)
    scattered = [0, 4, 8, 36, 40, 44, 72, 76, 80, 112, 116, 120, 124]
    setup(env, hands=[scattered[:] for _ in range(4)], wall=list(range(136)))      # reset() = _initialize_round: deltas cleared
    for i, w in enumerate([108, 109, 110, 111]):
        v = env.peek()
        p = v.current_player
        pl = v.players[p]
        pl.hand[0] = w
        v.drawn_tile = w
        env.poke(v)
        env.step({p: pack_action(DISCARD, w)})
    r = [e for e in evs(env) if e["type"] == "ryukyoku"][-1]
    assert r["reason"] == "sufuurenta" and r["deltas"] == [0, 0, 0, 0]


def sc_riichi_stage_only_tenpai_maintaining_discards(make):
    """tests.rs:936-1005: after the Riichi declaration (tile = None, MJAI style) the seat is in the riichi stage, not yet declared;
    its list holds only discards that keep 123456789m 12p 11s tenpai (the drawn 5sr among them) and no second Riichi."""
    env = make(game_mode=2)
    env.reset()
    v = env.peek()
    pid = v.current_player
    _set_hand(v.players[pid], sorted([0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 72, 73, 88]))
    v.players[pid].score = 25000
    v.drawn_tile = 88
    v.phase = WAIT_ACT
    v.active_mask = 1 << pid
    env.poke(v)
    env.step({pid: pack_action(RIICHI)})
    pl = env.peek().players[pid]
    assert pl.riichi_stage == 1 and pl.riichi_declared == 0
    legal = [unpack_action(a) for a in env.legal(pid)]
    discards = [tl for t, tl, _ in legal if t == DISCARD]
    assert discards and 88 in discards and RIICHI not in [t for t, _, _ in legal]
    from oracle import oracle

    hand = list(pl.hand[: pl.hand_len])
    for tl in discards:              # HandEvaluator(hand - tile).is_tenpai(): shanten 0 of the remaining thirteen
        rest = list(hand)
        rest.remove(tl)
        cnt = [[0] * 34]
        for t in rest:
            cnt[0][t // 4] += 1
        assert int(oracle.shanten(cnt)[0]) == 0, tl


def sc_reach_accepted_event_includes_actor(make):
    """tests.rs:857-934: Riichi (tile = None), the 5sr discard, everybody passes: the log holds a reach_accepted event with the
    declarer as actor and the pending acceptance is cleared."""
    env = make(game_mode=2)
    env.reset()
    v = env.peek()
    pid = v.current_player
    _set_hand(v.players[pid], sorted([0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 72, 73, 88]))
    v.players[pid].score = 25000
    v.drawn_tile = 88
    v.phase = WAIT_ACT
    v.active_mask = 1 << pid
    env.poke(v)
    env.step({pid: pack_action(RIICHI)})
    env.step({pid: pack_action(DISCARD, 88)})
    act, ph, _ = env.status()
    if ph == WAIT_RESPONSE:
        env.step({s: pack_action(PASS) for s in range(4) if (act >> s) & 1})
    ra = [e for e in evs(env) if e["type"] == "reach_accepted"]
    assert ra and ra[0]["actor"] == pid
    assert env.peek().riichi_pending_acceptance == -1


def sc_no_tobi_with_positive_scores(make):
    """tests.rs:1007-1040 (+ 270-311 for the bust case elsewhere): a hanchan goes on after a round when nobody is below zero."""
    env = make(game_mode=2)
    env.reset()
    v = env.peek()
    hand = list(v.players[0].hand[: v.players[0].hand_len])
    bad = next(t for t in range(136) if t not in hand)
    env.step({0: pack_action(DISCARD, bad)})                   # a penalty round end: 13000 / 29000 x 3, all positive
    v = env.peek()
    assert not v.is_done and all(v.players[p].score > 0 for p in range(4))


def sc_riichi_sequence(make):
    """docs/RULES.md:64-78, tests/env/rule_validation/test_riichi_sequence.py:118-215 (test_mjai_reach_then_dahai_sequence): reach -> dahai -> reach_accepted
    -> tsumo; riichi discard restricted to tenpai-keeping tiles; stick paid on acceptance; ippatsu tsumo."""
    # seat 0: 123m 456m 789m 234p 5p + draw 9s -> riichi by discarding 9s (wait 5p) ; wall fixed so nobody calls
    hand = tiles("123456789m2345p")
    env = setup(make(game_mode=1), hands=[hand, tiles("1199s1155z22266z")[0:13], tiles("147m258p369s1234z")[0:13], tiles("147m258p369s5677z")[0:13]],
                drawn_tile=107, wall=list(range(136)))
    r = find(env.legal(0), RIICHI)
    assert r is not None
    env.step({0: r})
    v = env.peek()
    assert v.players[0].riichi_stage == 1 and v.players[0].riichi_declared == 0
    disc = [unpack_action(a)[1] for a in env.legal(0) if unpack_action(a)[0] == DISCARD]
    # tenpai-keeping discards only: 9s (tanki 5p/…)/2p/5p
    assert 107 in disc and all(unpack_action(a)[0] == DISCARD for a in env.legal(0))
    env.step({0: pack_action(DISCARD, 107)})
    act, ph, dn = env.status()
    while ph == WAIT_RESPONSE:
        env.step({s: pack_action(PASS) for s in range(4) if (act >> s) & 1})
        act, ph, dn = env.status()
    t = [e["type"] for e in evs(env)]
    i = t.index("reach")
    assert t[i: i + 4] == ["reach", "dahai", "reach_accepted", "tsumo"]
    v = env.peek()
    assert v.players[0].score == 24000 and v.riichi_sticks == 1 and v.players[0].ippatsu_cycle == 1
    assert v.players[0].riichi_declared == 1 and v.players[0].double_riichi_declared == 1


def sc_riichi_stage_disables_ankan(make):
    """tests/env/rule_validation/test_riichi_sequence.py:25-117 (test_riichi_stage_disables_ankan) and :216-273 (test_legal_actions_consistency_with_mortal_state):
    333346m 23477p 345s + the fourth 3m: Riichi and Ankan are both offered; once riichi_stage is set (reach declared, discard pending) the list holds discards
    only - no Ankan, no second Riichi.  (The reference fills the other seats with thirteen copies of one tile; they do not matter here and hold ordinary hands.)"""
    hand = [8, 9, 10, 12, 20, 40, 44, 48, 60, 61, 80, 84, 88]
    others = [tiles("19m19p19s1234567z"), tiles("2468m2468p2468s5z"), tiles("357m357p357s1122z")[:13]]
    kw = dict(hands=others + [hand], current_player=3, active_players=[3], drawn_tile=11, riichi_declared=[False] * 4, points=[25000] * 4)
    env = setup(make(seed=42), **kw)
    types = {unpack_action(a)[0] for a in env.legal(3)}
    assert RIICHI in types and ANKAN in types
    env = setup(make(seed=42), mutate=lambda v: setattr(v.players[3], "riichi_stage", 1), **kw)
    types = [unpack_action(a)[0] for a in env.legal(3)]
    assert DISCARD in types and ANKAN not in types and RIICHI not in types and set(types) == {DISCARD}


def sc_kyushu_kyuhai(make):
    """tests/env/actions/test_kyushu_kyuhai.py:7-43 (test_kyushu_kyuhai_abortive_draw): 9 terminal kinds on the first turn -> KyushuKyuhai legal; abortive
    draw with renchan (honba+1); :44-74 (test_kyushu_kyuhai_not_available_after_meld): any meld on the table (here: a Pon of seat 1) takes the option away."""
    hand = tiles("19m19p19s1234z") + tiles("2m3p")[0:2]
    env2 = setup(make(game_mode=0, seed=42), hands=[tiles("19m19p19s123z") + [4, 5, 6, 7][:4], None, None, None], drawn_tile=12,
                 melds=[[], [(PON_M, [20, 21, 22], True, 0, 20)], [], []])
    assert find(env2.legal(0), KYUSHU) is None and env2.mask(0)[80] == 0
    env = setup(make(game_mode=1), hands=[hand[:13], None, None, None], drawn_tile=128)
    k = find(env.legal(0), KYUSHU)
    assert k is not None
    assert env.mask(0)[80] == 1
    env.step({0: k})
    r = [e for e in evs(env) if e["type"] == "ryukyoku"][-1]
    assert r["reason"] == "kyushu_kyuhai" and r["deltas"] == [0, 0, 0, 0]
    v = env.peek()
    assert (v.honba, v.oya, v.is_done) == (1, 0, 0)


def sc_double_ron_honba_sticks(make):
    """state/mod.rs:945-1142: two winners sorted by distance from the discarder; honba and riichi sticks only to
    the first; sanchaho not triggered with two."""
    hand2 = tiles("234m234p234s55p") + [96, 100]  # 7s8s -> waits 6s/9s
    hand3 = tiles("345m345p345s66p") + [96 + 1, 100 + 1]  # 7s8s -> waits 6s/9s
    env = setup(make(game_mode=1), hands=[tiles("19m19p1s1234567z")[0:12], None, hand2, hand3], drawn_tile=92,
                mutate=lambda v: (setattr(v, "honba", 2), setattr(v, "riichi_sticks", 1), setattr(v, "is_first_turn", 0)),
                wall=list(range(136)))
    env.step({0: pack_action(DISCARD, 92)})  # 6s
    act, ph, dn = env.status()
    assert ph == WAIT_RESPONSE and (act & 0b1100) == 0b1100
    acts = {s: pack_action(PASS) for s in range(4) if (act >> s) & 1}
    acts[2] = find(env.legal(2), RON)
    acts[3] = find(env.legal(3), RON)
    env.step(acts)
    h = [e for e in evs(env) if e["type"] == "hora"]
    assert [e["actor"] for e in h] == [2, 3]
    # seat 2: tanyao + pinfu + sanshoku(234) menzen ron = 4 han 30 fu -> 7700 (no kiriage) + honba 600 + stick 1000
    assert h[0]["deltas"] == [-8300, 0, 9300, 0]
    # seat 3: tanyao + pinfu + sanshoku(345) = 7700, no honba, no stick
    assert h[1]["deltas"] == [-7700, 0, 0, 7700]
    assert env.scores() == [25000 - 16000, 25000, 25000 + 9300, 25000 + 7700]


def sc_tobi_and_agariyame(make):
    """src/tests.rs:375-425: negative score ends the game; dealer top >= 30000 winning the last regular round ends it."""
    env = make(game_mode=2)
    hand = tiles("123456789m1134p")

    def mut(v):
        v.round_wind = 1
        v.oya = 3
        v.kyoku_idx = 3
        v.current_player = 3
        v.active_mask = 8
        v.is_first_turn = 0
        for p, s in enumerate([20000, 20000, 20000, 40000]):
            v.players[p].score = s
        v.players[3].n_discards = 1
        v.players[3].discards[0] = 108

    setup(env, hands=[None, None, None, hand], current_player=3, drawn_tile=tiles("2p")[0], mutate=mut)
    ts = find(env.legal(3), TSUMO)
    assert ts is not None
    env.step({3: ts})
    assert env.status()[2] == 1  # agari-yame
    assert [e["type"] for e in evs(env)][-2:] == ["end_kyoku", "end_game"]



# -- transcribed from tests/env/rule_validation/test_furiten_rules.py, test_temporary_furiten.py, test_valid_ankan.py,
#    tests/env/test_riichi_no_claims.py, tests/env/test_honba_reset.py --------------------------------------
_SAFE1 = tiles("147m258p369s2345z")[0:13]   # hands that cannot claim anything used below
_SAFE2 = tiles("147m258p369s4567z")[0:13]


def sc_furiten_ron(make):
    """test_furiten_rules.py:4-90 / :93-127: a wait tile among the own discards kills Ron on ANY wait; without it Ron is offered."""
    hand0 = [4, 8, 52, 53, 60, 61, 62, 92, 93, 94, 108, 109, 110]           # 23m 55p 666p 888s 111z: waits 1m / 4m
    for own_discards, expect_ron in (([0], False), ([], True)):
        env = setup(make(game_mode=2), hands=[hand0, _SAFE1, _SAFE2, tiles("19m19p19s1236677z")[0:13]],
                    current_player=1, active_players=[1], drawn_tile=12, discards=[own_discards, [], [], []], wall=list(range(136)))
        env.step({1: pack_action(DISCARD, 12)})                            # 4m
        act, ph, dn = env.status()
        if expect_ron:
            assert ph == WAIT_RESPONSE and (act & 1) and find(env.legal(0), RON) is not None
        else:
            assert not (act & 1)                                           # no Ron, and nothing else to claim


def sc_temporary_furiten(make):
    """test_temporary_furiten.py:11-66: passing an offered Ron sets missed_agari_doujun; the same tile from the next
    player is not offered again (same go-around)."""
    h3 = [4, 8, 12, 40, 44, 48, 76, 80, 84, 92, 93, 94, 96]                 # 234m 234p 234s 66s 67s: waits 5s / 8s
    h1 = _SAFE1[:12] + [88]
    h2 = _SAFE2[:12] + [89]
    env = setup(make(game_mode=2), hands=[tiles("19m19p19s1234567z")[0:13], h1, h2, h3], current_player=1,
                active_players=[1], drawn_tile=135, wall=list(range(136)), mutate=lambda v: setattr(v, "is_first_turn", 0))
    env.step({1: pack_action(DISCARD, 88)})                                 # 5s (red)
    act, ph, dn = env.status()
    assert ph == WAIT_RESPONSE and (act >> 3) & 1 and find(env.legal(3), RON) is not None
    env.step({s: pack_action(PASS) for s in range(4) if (act >> s) & 1})
    v = env.peek()
    assert v.players[3].missed_agari_doujun == 1
    assert v.current_player == 2 and v.phase == WAIT_ACT                    # seat 2 drew
    env.step({2: pack_action(DISCARD, 89)})                                 # 5s again, same go-around
    act, ph, dn = env.status()
    assert not ((act >> 3) & 1) or find(env.legal(3), RON) is None


def sc_valid_ankan_after_riichi(make):
    """test_valid_ankan.py:7-28: after riichi an ankan that keeps the waits is legal (hand 2m2m 9p9p9p(+1) 1s1s1s 5s5s5s 7s7s)."""
    h2 = [4, 5, 68, 69, 71, 73, 74, 75, 88, 89, 90, 97, 98]
    env = setup(make(game_mode=2), hands=[_SAFE1, _SAFE2, h2, tiles("19m19p19s1236677z")[0:13]], current_player=2,
                active_players=[2], drawn_tile=70, riichi_declared=[False, False, True, False], wall=list(range(136)))
    ank = [a for a in env.legal(2) if unpack_action(a)[0] == ANKAN]
    assert len(ank) == 1 and sorted(unpack_action(ank[0])[2]) == [68, 69, 70, 71]


def sc_no_claims_during_riichi(make):
    """test_riichi_no_claims.py:6-88: a riichi player is offered neither Chi nor Pon; without riichi Chi is offered."""
    for tile, h2, riichi, want in ((72, [76, 80, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44], True, None),
                                   (72, [76, 80, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44], False, CHI),
                                   (78, [76, 77, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44], True, None)):
        h = [list(range(13)) for _ in range(4)]
        h[1] = [tile] + list(range(12))
        h[2] = h2
        env = setup(make(game_mode=2), hands=h, current_player=1, active_players=[1], drawn_tile=tile,
                    riichi_declared=[False, False, riichi, False], wall=list(range(136)))
        env.step({1: pack_action(DISCARD, tile)})
        act, ph, dn = env.status()
        kinds = {unpack_action(a)[0] for a in env.legal(2)} if (act >> 2) & 1 else set()
        if want is None:
            assert CHI not in kinds and PON not in kinds
            assert ph == WAIT_ACT and act == 0b0100                        # seat 2 simply drew
        else:
            assert ph == WAIT_RESPONSE and want in kinds


def sc_honba_reset_and_increment(make):
    """test_honba_reset.py:4-73: a non-dealer win resets honba and rotates the dealer; a dealer win adds one."""
    h1 = [0, 1, 4, 5, 8, 9, 12, 13, 16, 17, 20, 21, 24]                      # chiitoi tanki 7m
    h0 = [2, 3, 6, 7, 10, 11, 14, 15, 18, 19, 22, 23, 26]
    env = setup(make(game_mode=2), hands=[h0, h1, _SAFE1, _SAFE2], drawn_tile=25, points=[60000, 25000, 25000, 25000],
                wall=list(range(136)), reset_kw={"oya": 0, "honba": 5})
    assert env.peek().honba == 5
    env.step({0: pack_action(DISCARD, 25)})
    act, ph, dn = env.status()
    assert ph == WAIT_RESPONSE and env.peek().last_discard_pid == 0 and env.peek().last_discard_tile == 25
    acts = {s: pack_action(PASS) for s in range(4) if (act >> s) & 1}
    acts[1] = find(env.legal(1), RON)
    env.step(acts)
    v = env.peek()
    assert v.honba == 0 and v.oya == 1
    env = setup(make(game_mode=2), hands=[h1, _SAFE1, _SAFE2, tiles("19m19p19s1236677z")[0:13]], drawn_tile=25,
                wall=list(range(136)), reset_kw={"oya": 0, "honba": 5})
    ts = find(env.legal(0), TSUMO)
    assert ts is not None
    env.step({0: ts})
    v = env.peek()
    assert v.honba == 6 and v.oya == 0

def sc_pao_ron_honba(make):
    """tests/env/agari/test_pao_honba.py:6-77: daisangen pao, Ron from a third party with 2 honba: the discarder pays half,
    the liable seat half plus the whole honba (32600 = 16000 + 16600)."""
    env = setup(make(game_mode=2), hands=[_SAFE1, _SAFE2, tiles("19m19p19s1236677z")[0:13], [0, 1, 2, 4, 99, 132, 133]],
                melds=[[], [], [], [(PON_M, [124, 125, 126], True, 0, -1), (PON_M, [128, 129, 130], True, 0, -1)]],
                current_player=0, phase=WAIT_RESPONSE, active_players=[3], wall=list(range(136)), reset_kw={"oya": 0, "honba": 2},
                mutate=lambda v: (setattr(v, "last_discard_pid", 0), setattr(v, "last_discard_tile", 134), setattr(v, "is_first_turn", 0)))
    pon = find(env.legal(3), PON, 134)
    assert pon is not None
    env.step({3: pon})
    assert env.peek().players[3].pao_daisangen == 0              # seat 0 is liable for the daisangen
    env.step({3: pack_action(DISCARD, 99)})
    act, ph, dn = env.status()
    if ph == WAIT_RESPONSE:
        env.step({s: pack_action(PASS) for s in range(4) if (act >> s) & 1})
    v = env.peek()                                               # now: seat 2 discards 2m (id 6) -> WaitResponse for seat 3
    for p in range(4):
        if v.players[p].hand_len + 3 * v.players[p].n_melds == 14:
            v.players[p].hand_len -= 1                           # undo the draw the engine dealt after the pass
    v.drawn_tile = -1
    v.current_player = 2
    v.phase = WAIT_RESPONSE
    v.active_mask = 0b1000
    v.last_discard_pid, v.last_discard_tile = 2, 6
    env.poke(v)
    ron = find(env.legal(3), RON, 6)
    assert ron is not None
    env.step({3: ron})
    hora = next(e for e in reversed(evs(env)) if e["type"] == "hora")
    assert hora["deltas"] == [-16600, 0, -16000, 32600]


def sc_doujun_cleared_by_call(make):
    """tests/env/test_m263_ron_mismatch.py:4-77: temporary furiten (missed_agari_doujun) ends with the seat's own call and
    discard; the next winning discard is offered as Ron again."""
    h1 = _SAFE1[:12] + [36]
    h2 = _SAFE2[:12] + [93]
    env = setup(make(game_mode=2), hands=[tiles("19m19p19s1236677z")[0:13], h1, h2, [37, 38, 40, 92]],
                melds=[[], [], [], [(PON_M, [124, 125, 126], True, 0, -1), (CHI_M, [24, 28, 32], True, 2, -1), (CHI_M, [60, 64, 68], True, 2, -1)]],
                current_player=1, active_players=[1], drawn_tile=135, wall=list(range(136)),
                mutate=lambda v: (setattr(v.players[3], "missed_agari_doujun", 1), setattr(v, "is_first_turn", 0)))
    env.step({1: pack_action(DISCARD, 36)})                               # 1p
    pon = find(env.legal(3), PON, 36)
    assert pon is not None
    act = env.status()[0]
    acts = {s: pack_action(PASS) for s in range(4) if (act >> s) & 1}
    acts[3] = pon
    env.step(acts)
    env.step({3: pack_action(DISCARD, 40)})                               # 2p; hand is now the single 7s
    act, ph, dn = env.status()
    if ph == WAIT_RESPONSE:
        env.step({s: pack_action(PASS) for s in range(4) if (act >> s) & 1})
    v = env.peek()
    assert v.players[3].missed_agari_doujun == 0
    assert v.current_player == 0 and v.phase == WAIT_ACT                  # seat 0 drew; let it tsumogiri, then seat 1, then seat 2
    for seat in (0, 1):
        v = env.peek()
        env.step({seat: pack_action(DISCARD, v.drawn_tile)})
        act, ph, dn = env.status()
        if ph == WAIT_RESPONSE:
            env.step({s: pack_action(PASS) for s in range(4) if (act >> s) & 1})
    assert env.peek().current_player == 2
    env.step({2: pack_action(DISCARD, 93)})                               # 7s
    act, ph, dn = env.status()
    assert ph == WAIT_RESPONSE and (act >> 3) & 1 and find(env.legal(3), RON, 93) is not None


def _exhaust(env, scores):
    """Put the game one discard before an exhaustive draw with everybody noten and no nagashi; returns after the draw."""
    v = env.peek()
    hands = [tiles("19m19p19s1236677z")[0:13], _SAFE1, _SAFE2, tiles("258m369p147s1234z")[0:13]]
    cur = v.current_player
    for p in range(4):
        h = sorted(hands[p] + ([135] if p == cur else []))
        v.players[p].hand_len = len(h)
        for i, t in enumerate(h):
            v.players[p].hand[i] = t
        v.players[p].n_melds = 0
        v.players[p].nagashi_eligible = 0
        v.players[p].score = scores[p]
        v.players[p].riichi_declared = 0
    v.drawn_tile = 135
    v.phase = WAIT_ACT
    v.active_mask = 1 << cur
    v.drawable_count = 0
    v.is_first_turn = 0
    env.poke(v)
    env.step({cur: pack_action(DISCARD, 135)})


def sc_sudden_death_west_round(make):
    """src/tests.rs:172-231: South 4 ends below 30000 -> the hanchan enters West 1; once somebody holds >= 30000 after a
    West hand the game ends (end_game is the last event)."""
    env = setup(make(game_mode=2), wall=list(range(136)), current_player=3, active_players=[3],
                mutate=lambda v: (setattr(v, "round_wind", 1), setattr(v, "kyoku_idx", 3), setattr(v, "oya", 3)))
    _exhaust(env, [25000] * 4)
    v = env.peek()
    assert not v.is_done and v.round_wind == 2 and v.kyoku_idx == 0 and v.oya == 0
    _exhaust(env, [31000, 25000, 24000, 20000])
    assert env.status()[2] == 1
    t = [e["type"] for e in evs(env)]
    assert t[-1] == "end_game" and "ryukyoku" in t

def sc_ron_after_call_clears_doujun(make):
    """test_m263_ron_mismatch.py:4-77: a seat in temporary furiten calls Pon and discards; the furiten is gone and the next
    discard of its wait is offered as Ron."""
    melds3 = [(abi.MELD_PON, [124, 125, 126], True, 0, 124), (abi.MELD_CHI, [24, 28, 32], True, 2, 24), (abi.MELD_CHI, [60, 64, 68], True, 2, 60)]
    h1 = [36] + _SAFE1[1:13]

    def mut(v):
        v.players[3].missed_agari_doujun = 1
        v.is_first_turn = 0

    env = setup(make(game_mode=2), hands=[_SAFE2, h1[:12], tiles("2378m2378p23s567z")[0:13][:12] + [93], [92, 37, 38, 40]],
                melds=[None, None, None, melds3], current_player=1, active_players=[1], drawn_tile=36 if 36 not in h1[:12] else None,
                wall=list(range(136)), mutate=mut)
    env.step({1: pack_action(DISCARD, 36)})                                 # 1p
    act, ph, dn = env.status()
    assert ph == WAIT_RESPONSE and (act >> 3) & 1 and find(env.legal(3), PON) is not None
    env.step({3: pack_action(PON, 36, [37, 38])})
    assert env.peek().players[3].missed_agari_doujun == 0                   # cleared by the call (state/mod.rs:1196)
    env.step({3: pack_action(DISCARD, 40)})
    while env.status()[1] == WAIT_RESPONSE:                                 # nobody needs the 2p
        a = env.status()[0]
        env.step({s: pack_action(PASS) for s in range(4) if (a >> s) & 1})
    v = env.peek()
    assert v.phase == WAIT_ACT and v.current_player == 0                    # seat 0 drew; force seat 2's turn with the 7s
    v.current_player, v.active_mask = 2, 1 << 2
    env.poke(v)
    env.step({2: pack_action(DISCARD, 93)})                                 # 7s
    act, ph, dn = env.status()
    assert ph == WAIT_RESPONSE and (act >> 3) & 1 and find(env.legal(3), RON) is not None


def sc_discard_type_tracking(make):
    """test_discard_type.py:5-61: discard_from_hand is False for a tsumogiri and True for a tedashi."""
    env = setup(make(game_mode=2), hands=[_SAFE1, _SAFE2, tiles("258m369p147s1234z")[0:13], tiles("369m147p258s4567z")[0:13]],
                drawn_tile=10, wall=list(range(136)))
    env.step({0: pack_action(DISCARD, 10)})
    p = env.peek().players[0]
    assert p.n_discards == 1 and p.discards[0] == 10 and (p.discard_from_hand_bits & 1) == 0
    while env.status()[1] == WAIT_RESPONSE:
        a = env.status()[0]
        env.step({s: pack_action(PASS) for s in range(4) if (a >> s) & 1})
    v = env.peek()
    cur = v.current_player
    tile = v.players[cur].hand[0]
    assert tile != v.drawn_tile
    env.step({cur: pack_action(DISCARD, tile)})
    q = env.peek().players[cur]
    assert q.discards[q.n_discards - 1] == tile and (q.discard_from_hand_bits >> (q.n_discards - 1)) & 1 == 1


def sc_riichi_markers(make):
    """test_riichi_markers.py:5-113: riichi_stage after the declaration; the declaring discard sets discard_is_riichi,
    riichi_declared and riichi_declaration_index; other seats stay None; reset() clears everything."""
    hand0 = [0, 1, 2, 12, 13, 14, 24, 25, 26, 40, 41, 42, 108]
    env = setup(make(game_mode=2), hands=[hand0, _SAFE1, _SAFE2, None], drawn_tile=109, wall=list(range(136)))
    r = find(env.legal(0), RIICHI)
    assert r is not None
    env.step({0: r})
    v = env.peek()
    assert v.players[0].riichi_stage == 1 and v.players[0].riichi_declared == 0
    discards = [a for a in env.legal(0) if unpack_action(a)[0] == DISCARD]
    assert discards and len(discards) == len(env.legal(0))                  # riichi stage: nothing but (tenpai-keeping) discards
    env.step({0: discards[0]})
    v = env.peek()
    p = v.players[0]
    assert p.n_discards == 1 and p.discards[0] == unpack_action(discards[0])[1]
    assert (p.discard_is_riichi_bits & 1) == 1 and p.riichi_stage == 0 and p.riichi_declared == 1
    assert p.riichi_declaration_index == 0 and v.players[1].riichi_declaration_index == -1
    env.reset()
    v = env.peek()
    assert v.players[0].riichi_declaration_index == -1 and v.players[0].n_discards == 0 and v.players[0].discard_is_riichi_bits == 0


def sc_daiminkan_rinshan_draw(make):
    """actions/test_daiminkan_rinshan_draw.py:8-53: an accepted Daiminkan makes the caller the current player in WaitAct
    with a rinshan tile drawn from the wall and a tsumo event logged."""
    h1 = [72, 73, 74] + [1, 5, 9, 13, 37, 41, 45, 49, 110, 114]
    env = setup(make(game_mode=2), hands=[tiles("19m19p19s1234567z")[0:13][:12] + [3], h1, _SAFE1, _SAFE2], drawn_tile=75,
                wall=list(range(136)))
    env.step({0: pack_action(DISCARD, 75)})
    act, ph, dn = env.status()
    assert ph == WAIT_RESPONSE and (act >> 1) & 1
    kan = find(env.legal(1), DAIMINKAN)
    assert kan is not None
    before = env.peek()
    env.step({1: kan})
    v = env.peek()
    assert v.current_player == 1 and v.phase == WAIT_ACT and v.drawn_tile >= 0
    assert v.rinshan_draw_count == before.rinshan_draw_count + 1
    last = evs(env)[-1]
    assert last["type"] == "tsumo" and last["actor"] == 1 and last["pai"] == abi_mjai(v.drawn_tile)


def sc_chi_needs_the_exact_copies(make):
    """actions/test_relaxed_red5.py:7-81: a Chi must name tiles the caller really holds - the red five by its own id, a plain
    five by the held copy; naming another copy is an illegal action (penalty ryukyoku, no meld)."""
    filler = [108, 109, 110, 112, 113, 114, 116, 117, 118, 120, 121]
    for five, named, ok in ((16, 16, True), (16, 20, False), (17, 18, False), (17, 17, True)):
        env = setup(make(game_mode=2), hands=[[five, 12] + filler, _SAFE1, _SAFE2, tiles("19m19p19s1234567z")[0:13][:12] + [1]],
                    current_player=3, active_players=[3], drawn_tile=8, wall=list(range(136)))
        env.step({3: pack_action(DISCARD, 8)})                              # 3m from kamicha
        act, ph, dn = env.status()
        assert ph == WAIT_RESPONSE and act & 1
        env.step({0: pack_action(CHI, 8, [named, 12])})
        p0 = env.peek().players[0]
        if ok:
            assert p0.n_melds == 1 and p0.melds[0].meld_type == abi.MELD_CHI and five in list(p0.melds[0].tiles[:3])
        else:
            assert p0.n_melds == 0

def sc_kakan_from_tsumo_and_from_hand(make):
    """actions/test_kakan.py:7-84: with a Pon of 1m, Kakan is offered for the 4th copy whether it was just drawn (a duplicate id
    in the reference's test: relaxed check) or sits in the hand; executing it turns the Pon into a Kakan of [0,1,2,3], removes
    the tile from the hand and logs a kakan event."""
    pon = [(abi.MELD_PON, [0, 1, 2], True, 1, 0)]
    env = setup(make(game_mode=2), hands=[[4, 5, 6, 7, 8, 9, 10, 11, 12, 60], _SAFE1, _SAFE2, None], melds=[pon, None, None, None],
                drawn_tile=2, wall=list(range(136)))
    assert find(env.legal(0), KAKAN) is not None
    env = setup(make(game_mode=2), hands=[[3, 4, 5, 6, 7, 8, 9, 10, 11, 12], _SAFE1, _SAFE2, None], melds=[pon, None, None, None],
                drawn_tile=13, wall=list(range(136)))
    k = [a for a in env.legal(0) if unpack_action(a)[0] == KAKAN]
    assert len(k) == 1 and unpack_action(k[0])[1] == 3 and sorted(unpack_action(k[0])[2]) == [0, 1, 2]
    env.step({0: k[0]})
    while env.status()[1] == WAIT_RESPONSE:                                  # nobody can rob it
        a = env.status()[0]
        env.step({q: pack_action(PASS) for q in range(4) if (a >> q) & 1})
    p = env.peek().players[0]
    assert p.n_melds == 1 and p.melds[0].meld_type == abi.MELD_KAKAN and sorted(p.melds[0].tiles[:4]) == [0, 1, 2, 3]
    assert 3 not in list(p.hand[: p.hand_len])
    assert any(e["type"] == "kakan" for e in evs(env))


def sc_riichi_player_tsumogiri(make):
    """actions/test_riichi_pass.py:11-54: a riichi player in WaitAct may discard the drawn tile or win, never Pass; the discard is
    logged as a tsumogiri and play moves on."""
    env = setup(make(game_mode=2), hands=[list(range(13)), _SAFE1, _SAFE2, tiles("258m369p147s1234z")[0:13]], drawn_tile=10,
                riichi_declared=[True, False, False, False], wall=list(range(136)))
    kinds = [unpack_action(a)[0] for a in env.legal(0)]
    assert DISCARD in kinds and PASS not in kinds and TSUMO in kinds
    assert [unpack_action(a)[1] for a in env.legal(0) if unpack_action(a)[0] == DISCARD] == [10]
    env.step({0: pack_action(DISCARD, 10)})
    v = env.peek()
    assert v.phase == WAIT_ACT and v.current_player == 1 and (v.last_discard_pid, v.last_discard_tile) == (0, 10)
    e = evs(env)
    assert e[-2]["type"] == "dahai" and e[-2]["actor"] == 0 and e[-2]["tsumogiri"] is True and e[-1]["type"] == "tsumo"

def sc_melds_with_red_fives(make):
    """actions/test_meld_aka.py:12-232: Chi in all three positions and Pon with a plain or a red 5m as the called tile, incl.
    a Chi that names the second copy of a tile (ids as riichienv.parse_hand assigns them; the two hands of the reference's
    test share ids, and the discarder holds the discarded id twice)."""
    h0 = [0, 4, 8, 12, 17, 20, 24, 28, 32, 36, 40, 44, 48]                    # 123456789m1234p
    pins = [36, 40, 44, 48, 53, 56, 60, 64, 68, 80]                          # 123456789p3s
    cases = [([8, 12, 13], 17, CHI, [8, 12]), ([8, 12, 13], 16, CHI, [8, 12]), ([12, 13, 20], 17, CHI, [12, 20]),
             ([12, 13, 20], 16, CHI, [12, 20]), ([24, 25, 20], 17, CHI, [20, 24]), ([24, 25, 20], 16, CHI, [20, 24]),
             ([24, 25, 20], 16, CHI, [20, 25]), ([17, 18, 20], 16, PON, [17, 18]), ([17, 18, 20], 19, PON, [17, 18])]
    for man, drawn, kind, consume in cases:
        env = setup(make(game_mode=2), hands=[h0, man + pins, _SAFE1, _SAFE2], drawn_tile=drawn, wall=list(range(136)))
        env.step({0: pack_action(DISCARD, drawn)})
        act, ph, dn = env.status()
        assert ph == WAIT_RESPONSE and act == 0b0010
        assert pack_action(kind, drawn, consume) in env.legal(1)
        env.step({1: pack_action(kind, drawn, consume)})
        act, ph, dn = env.status()
        assert ph == WAIT_ACT and act == 0b0010
        assert evs(env)[-1]["type"] == ("chi" if kind == CHI else "pon")

def sc_game_mode_round_transitions(make):
    """test_game_rules.py:11-100: after a kyushu kyuhai draw a single-round game is over; an east-only game goes on with the
    same dealer and one honba, announced by a start_kyoku event."""
    h = [0, 4, 8, 12, 36, 40, 44, 48, 72, 76, 80, 108, 112]
    for mode, expect_done in ((0, True), (1, False)):
        env = setup(make(game_mode=mode), hands=[h, tiles("19m19p19s1234567z")[0:13], list(h), list(h)], drawn_tile=1,
                    wall=list(range(136)))
        v = env.peek()
        assert v.oya == 0 and v.round_wind == 0 and v.honba == 0
        env.step({0: pack_action(DISCARD, 80)})
        act, ph, dn = env.status()
        assert (act >> 1) & 1 and find(env.legal(1), KYUSHU) is not None
        env.step({1: pack_action(KYUSHU)})
        act, ph, dn = env.status()
        assert bool(dn) == expect_done
        if not expect_done:
            v = env.peek()
            assert act & 1 and v.oya == 0 and v.round_wind == 0 and v.honba == 1
            e = evs(env)
            assert e[-2]["type"] == "start_kyoku" and e[-2]["oya"] == 0 and e[-2]["honba"] == 1

def sc_oyayame_needs_the_sole_or_seat_first_top(make):
    """_initialize_next_round (state/mod.rs:1606-1620; tests/test_oyayame_tiebreak.py:6-78): a dealer win in the last regular
    round ends the game only if the dealer is top with >= 30000, ties going to the LOWER seat - dealer seat 3 tied with seat
    0 plays on (renchan), one point more for the dealer ends it."""
    h3 = [4, 8, 12, 13, 17, 20, 56, 60, 64, 80, 81, 84, 92]                 # 234m 456m 678p 33s 46s: kanchan 5s, tanyao only
    h1 = _SAFE1[:12]
    for s0, expect_done in ((30000, False), (29900, True)):
        def mut(v):
            v.n_dora = 1
            v.dora[0] = 132
            v.is_first_turn = 0
        env = setup(make(game_mode=2), hands=[tiles("19m19p19s1234567z")[0:13], h1, _SAFE2, h3], current_player=1, active_players=[1],
                    drawn_tile=89, points=[s0, 22000, 20000, 28000], round_wind=1, wall=list(range(136)), mutate=mut,
                    reset_kw={"oya": 3, "round_wind": 1})
        env.step({1: pack_action(DISCARD, 89)})
        act, ph, dn = env.status()
        assert ph == WAIT_RESPONSE and (act >> 3) & 1 and find(env.legal(3), RON) is not None
        env.step({3: pack_action(RON, 89)})
        hora = [e for e in evs(env) if e["type"] == "hora"][-1]
        assert hora["deltas"] == [0, -2000, 0, 2000]                        # dealer 1 han 40 fu
        act, ph, dn = env.status()
        v = env.peek()
        assert bool(dn) == expect_done
        if not expect_done:
            assert v.oya == 3 and v.honba == 1 and v.round_wind == 1 and [p.score for p in v.players] == [30000, 20000, 20000, 30000]


def sc_win_results_of_the_final_round(make):
    """tests/env/test_riichienv_hora.py:5-100 (+ env.rs:606-607): win_results holds the winner's WinResult after the round
    that ended the game - chankan as the only yaku ([3]), then South-round yakuhai + toitoi + sanankou + honitsu
    ([11, 21, 22, 27], emission order); a new round clears it (state/mod.rs:1729)."""
    env = make(game_mode=0, seed=42)
    env.reset()
    assert env.win_results() == {}
    p1 = [0, 4, 8, 40, 44, 48, 56, 64, 72, 76, 80, 104, 105]

    def mut(v):
        v.players[0].n_melds = 1
        set_meld(v.players[0].melds[0], PON_M, [60, 61, 62], True)
        v.n_dora = 1
        v.dora[0] = 135

    setup(env, hands=[list(range(120, 132)), p1, None, None], current_player=0, drawn_tile=63, mutate=mut)
    env.step({0: pack_action(KAKAN, 63)})
    act, ph, dn = env.status()
    assert ph == WAIT_RESPONSE and (act >> 1) & 1
    ron = [a for a in env.legal(1) if unpack_action(a)[0] == RON]
    assert len(ron) == 1
    env.step({1: ron[0]})
    assert env.status()[2] == 1 and "hora" in [e["type"] for e in evs(env)[-3:]]
    w = env.win_results()
    assert list(w) == [1] and w[1]["yaku"] == [3] and w[1]["is_win"] and not w[1]["yakuman"] and w[1]["pao_payer"] is None
    # South 1: seat 3 (North seat) wins on 9m with a South pon: yakuhai of the round wind, toitoi, sanankou, honitsu
    env.reset(round_wind=1, oya=0, honba=0, kyotaku=0)
    assert env.win_results() == {}

    def mut2(v):
        v.players[3].n_melds = 1
        set_meld(v.players[3].melds[0], PON_M, [112, 113, 114], True)
        v.drawn_tile = -1
        v.n_dora = 1
        v.dora[0] = 135   # the reference's seed gives this round no dora in the winner's hand; keep the yaku list free of it

    setup(env, hands=[None, None, [33] + list(range(40, 52)), [0, 1, 2, 4, 5, 6, 8, 9, 10, 32]], current_player=2, mutate=mut2,
          reset_kw=dict(round_wind=1, oya=0, honba=0, kyotaku=0))
    env.step({2: pack_action(DISCARD, 33)})
    assert (env.status()[0] >> 3) & 1 and find(env.legal(3), RON) is not None
    env.step({3: pack_action(RON, 33)})
    assert env.win_results()[3]["yaku"] == [11, 21, 22, 27]


SCENARIOS = [sc_win_results_of_the_final_round, sc_paishan_dora_indices, sc_kakan_dora_timing, sc_daiminkan_dora_timing, sc_south_round_tsumo,
             sc_illegal_discard_penalty, sc_illegal_out_of_turn, sc_claim_priority_pon_over_chi, sc_kuikae_suji,
             sc_kuikae_deadlock, sc_sufuurenta, sc_suukansansen, sc_chankan_ron, sc_chankan_pass, sc_pao_daisangen_tsumo,
             sc_pao_mjsoul_composite_tsumo, sc_mjsoul_pao_tsumo_composite, sc_mjsoul_pao_ron_composite, sc_mjsoul_pao_ron_single,
             sc_mjsoul_pao_ron_real_record, sc_mjsoul_pao_ron_real_record_with_riichi_stick, sc_rules_chankan_kokushi_tenhou,
             sc_rules_chankan_kokushi_mjsoul, sc_rules_standard_chankan_kakan, sc_game_modes_initialization_params,
             sc_game_modes_south_round_wind, sc_riichi_setup_leaves_only_discards,
             sc_riichi_autoplay_waits_for_the_discard_possible_hands, sc_riichi_autoplay_unnamed_current_player, sc_kokushi_ankan_ron, sc_kokushi_ankan_ron_tenhou,
             sc_non_kokushi_ankan_no_ron, sc_ankan_generation, sc_ankan_generation_riichi, sc_chankan_stale_claims_repro,
             sc_env_scoring_ron_deltas, sc_env_scoring_tsumo_deltas, sc_env_scoring_ura_markers, sc_env_initialization,
             sc_env_basic_step_processing, sc_env_pon_claim, sc_env_pon_red_dora_claim, sc_env_chi_claim, sc_env_chi_claim_with_red_dora,
             sc_env_chi_claim_with_invalid_tile, sc_env_chi_claim_with_invalid_combo, sc_env_chi_multiple_patterns, sc_env_ron_claim,
             sc_env_ankan_riichi_legality, sc_daiminkan_pao_daisangen, sc_daiminkan_pao_daisuushii, sc_daiminkan_no_pao_insufficient_melds,
             sc_tenhou_tsumo_pao_composite, sc_tenhou_ron_pao_composite, sc_ryukyoku_deltas_are_reset_each_round_4p,
             sc_riichi_stage_only_tenpai_maintaining_discards, sc_reach_accepted_event_includes_actor, sc_no_tobi_with_positive_scores, sc_riichi_sequence, sc_riichi_stage_disables_ankan, sc_kyushu_kyuhai, sc_double_ron_honba_sticks,
             sc_tobi_and_agariyame,
             sc_furiten_ron, sc_temporary_furiten, sc_valid_ankan_after_riichi, sc_no_claims_during_riichi,
             sc_honba_reset_and_increment, sc_pao_ron_honba, sc_doujun_cleared_by_call,
             sc_sudden_death_west_round, sc_ron_after_call_clears_doujun, sc_discard_type_tracking, sc_riichi_markers,
             sc_daiminkan_rinshan_draw, sc_chi_needs_the_exact_copies, sc_kakan_from_tsumo_and_from_hand,
             sc_riichi_player_tsumogiri, sc_melds_with_red_fives, sc_game_mode_round_transitions,
             sc_oyayame_needs_the_sole_or_seat_first_top]


# ---------------------------------------------------------------------------------------------------------
# 3-player (sanma) KATs — riichienv-core/src/tests.rs:1042-1255, 2024-2340; tests/env/test_sanma.py
KITA = abi.KITA


def sc3_basics(make):
    """tests.rs:1067-1124: 35000 start, 108 tiles without 2m-8m, dealer 14 / others 13, 3 seats in the events."""
    env = make(game_mode=3)
    env.reset()
    v = env.peek()
    assert [v.players[p].score for p in range(3)] == [35000] * 3
    assert [v.players[p].hand_len for p in range(4)] == [14, 13, 13, 0]
    all_tiles = list(v.wall[: v.wall_len]) + [t for p in range(3) for t in v.players[p].hand[: v.players[p].hand_len]]
    assert len(all_tiles) == 108 and not any(1 <= t // 4 <= 7 for t in all_tiles)
    assert v.drawable_count == 54  # 108 - 39 dealt - 14, minus the dealer's draw
    sk = next(e for e in evs(env) if e["type"] == "start_kyoku")
    assert sk["scores"] == [35000] * 3 and len(sk["tehais"]) == 3
    assert [len(json.loads(s)["tehais"]) for s in env.log(1) if '"start_kyoku"' in s] == [3]


def sc3_no_chi(make):
    """tests.rs:1126-1156: a sequential hand is never offered Chi in sanma."""
    env = setup(make(game_mode=5), hands=[None, [36, 40, 44, 48, 52, 56, 60, 64, 68, 72, 76, 80, 84], None, None])
    v = env.peek()
    t = v.players[0].hand[0]
    env.step({0: pack_action(DISCARD, t)})
    act, ph, dn = env.status()
    if ph == WAIT_RESPONSE and (act >> 1) & 1:
        assert find(env.legal(1), CHI) is None


def sc3_kita(make):
    """tests.rs:1166-1187 + state_3p/sanma.rs:9-204: Kita is legal with a North tile, emits kita -> tsumo (rinshan, no
    dora), counts as nukidora, breaks the first turn; action id 59."""
    env = setup(make(game_mode=3), hands=[[0, 36, 40, 44, 48, 52, 56, 60, 64, 68, 72, 76], None, None, None], drawn_tile=120,
                wall=None)
    k = find(env.legal(0), KITA)
    assert k is not None and unpack_action(k) == (KITA, 120, [])
    assert env.mask(0)[59] == 1
    before = env.peek()
    env.step({0: k})
    v = env.peek()
    act, ph, dn = env.status()
    if ph == WAIT_ACT:  # nobody could ron the North tile
        assert v.players[0].n_kita == 1 and v.players[0].kita[0] == 120 and v.is_first_turn == 0
        assert v.rinshan_draw_count == 1 and v.is_rinshan_flag == 1 and v.n_dora == 1
        assert v.drawable_count == before.drawable_count - 1
        t = [e["type"] for e in evs(env)]
        assert t[-2:] == ["kita", "tsumo"]
        assert evs(env)[-2] == {"actor": 0, "pai": "N", "type": "kita"}


def sc3_oyayame_needs_40000(make):
    """tests.rs:1051-1065: last regular round, dealer top but below 40000 -> the game continues."""
    env = make(game_mode=5)
    hand = tiles("123456789p1134s")

    def mut(v):
        v.round_wind = 1
        v.oya = 2
        v.kyoku_idx = 2
        v.current_player = 2
        v.active_mask = 4
        v.is_first_turn = 0
        for p, s in enumerate([34000, 34000, 30000]):  # the win lifts the dealer to the top but below 40000
            v.players[p].score = s
        v.players[2].n_discards = 1
        v.players[2].discards[0] = 108

    setup(env, hands=[None, None, hand, None], current_player=2, drawn_tile=tiles("2s")[0], mutate=mut)
    ts = find(env.legal(2), TSUMO)
    assert ts is not None
    env.step({2: ts})
    assert env.status()[2] == 0  # continues: renchan in South 3
    v = env.peek()
    assert (v.oya, v.round_wind, v.honba) == (2, 1, 1)


def sc3_tsumo_payments_and_nukidora(make):
    """tests.rs:1207-1216 (3P tsumo: two payers) + hand_evaluator_3p.rs:126-147 (nukidora = kita count, yaku 34)."""
    env = make(game_mode=5)
    hand = tiles("234567p234567s9s")  # tanyao pinfu-shaped tenpai on 9s? keep simple: pair wait

    def mut(v):
        v.is_first_turn = 0
        v.players[1].n_discards = 1
        v.players[1].discards[0] = 108
        v.players[1].n_kita = 2
        v.players[1].kita[0] = 120
        v.players[1].kita[1] = 121
        v.current_player = 1
        v.active_mask = 2

    setup(env, oya=0, hands=[None, hand, None, None], current_player=1, drawn_tile=tiles("99s")[1], mutate=mut,
          wall=None)
    ts = find(env.legal(1), TSUMO)
    assert ts is not None
    env.step({1: ts})
    hora = next(e for e in reversed(evs(env)) if e["type"] == "hora")
    assert len(hora["deltas"]) == 3 and sum(hora["deltas"]) == 0
    # ko tsumo: dealer pays pay_tsumo_oya, the other ko pays pay_tsumo_ko
    assert hora["deltas"][0] < hora["deltas"][2] < 0 < hora["deltas"][1]


def sc3_exhaustive_draw_pool_2000(make):
    """state_3p/game_mode.rs:39-41 + state_3p/mod.rs:1791-1800: tenpai pool 2000 split over 3 seats."""
    env = make(game_mode=5)
    tenpai_hand = tiles("123456789p1239s")[0:13]

    def mut(v):
        v.drawable_count = 0
        v.is_first_turn = 0
        for p in range(3):
            v.players[p].nagashi_eligible = 0

    setup(env, hands=[tiles("19m19p19s1234567z")[0:13][:12] + [tiles("2p")[0]], tenpai_hand, tiles("147p258s369s1234z")[0:13], None],
          drawn_tile=tiles("8s")[0], mutate=mut)
    v = env.peek()
    env.step({0: pack_action(DISCARD, v.drawn_tile)})
    act, ph, dn = env.status()
    if ph == WAIT_RESPONSE:
        env.step({s: pack_action(PASS) for s in range(3) if (act >> s) & 1})
    r = [e for e in evs(env) if e["type"] == "ryukyoku"][-1]
    assert r["reason"] == "exhaustive_draw" and len(r["deltas"]) == 3
    assert sorted(r["deltas"]) in ([-1000, -1000, 2000], [-2000, 1000, 1000], [0, 0, 0])


def sc3_pon_and_rotation(make):
    """tests/env/test_sanma.py:119-207: turns rotate 0 -> 1 -> 2 -> 0, Chi never appears, Pon works and makes the caller the
    current player."""
    h0 = [36, 40, 44, 48, 52, 56, 60, 64, 68, 72, 76, 80, 84]
    h1 = sorted([37, 38, 49, 53, 57, 61, 65, 69, 73, 77, 81, 85, 89])
    env = setup(make(game_mode=5), hands=[h0, h1, tiles("19m19p19s1234567z")[0:13], None], drawn_tile=88)
    env.step({0: pack_action(DISCARD, 36)})                                 # 1p: seat 1 holds a pair
    act, ph, dn = env.status()
    assert ph == WAIT_RESPONSE and (act >> 1) & 1
    kinds = {unpack_action(a)[0] for q in range(3) if (act >> q) & 1 for a in env.legal(q)}
    assert PON in kinds and CHI not in kinds
    env.step({1: find(env.legal(1), PON)})
    v = env.peek()
    assert v.current_player == 1 and v.phase == WAIT_ACT and v.players[1].n_melds == 1
    # rotation from a fresh round: discard the drawn tile, pass every claim
    env = make(game_mode=5)
    env.reset()
    order = []
    for _ in range(4):
        v = env.peek()
        order.append(v.current_player)
        assert all(unpack_action(a)[0] != CHI for a in env.legal(v.current_player))
        env.step({v.current_player: pack_action(DISCARD, v.drawn_tile)})
        while env.status()[1] == WAIT_RESPONSE:
            a = env.status()[0]
            assert all(unpack_action(x)[0] != CHI for q in range(3) if (a >> q) & 1 for x in env.legal(q))
            env.step({q: pack_action(PASS) for q in range(3) if (a >> q) & 1})
    assert order == [0, 1, 2, 0]


def sc3_ron_deltas(make):
    """tests/env/test_sanma.py:434-466: a Ron in 3P moves points between winner and discarder only; the third seat is untouched."""
    p1 = sorted([36, 37, 38, 40, 41, 42, 44, 45, 46, 32, 33, 34, 0])       # 111p 222p 333p 999m 1m: tanki 1m
    h0 = sorted([48, 52, 56, 60, 64, 68, 72, 76, 80, 84, 88, 92, 96])
    env = setup(make(game_mode=5), hands=[h0, p1, tiles("19m19p19s1234567z")[0:13], None], drawn_tile=1,
                mutate=lambda v: setattr(v, "is_first_turn", 0))
    env.step({0: pack_action(DISCARD, 1)})
    act, ph, dn = env.status()
    assert ph == WAIT_RESPONSE and (act >> 1) & 1
    rons = [a for a in env.legal(1) if unpack_action(a)[0] == RON]
    assert len(rons) == 1
    env.step({1: rons[0]})
    d = [e for e in evs(env) if e["type"] == "hora"][-1]["deltas"]
    assert len(d) == 3 and d[1] > 0 and d[0] < 0 and d[2] == 0 and sum(d) == 0


def _poke3(env, fn):
    v = env.peek()
    fn(v)
    env.poke(v)
    return env.peek()


def _set_hand(pl, hand):
    pl.hand_len = len(hand)
    for i, t in enumerate(hand):
        pl.hand[i] = t


def _kita_state(make, seat, hand13, north, riichi=False):
    """The fixture of the kita-interaction tests (tests.rs:2024-2340): a fresh 3p-red-half round, the wall replaced by
    1p..5p (ids 36..55, rinshan draw = 36) with drawable_count = len - 14, `seat` to act holding `hand13` + the North tile
    it has just drawn."""
    env = make(game_mode=5)
    env.reset()

    def mut(v):
        v.wall_len = 20
        for i, t in enumerate(range(36, 56)):
            v.wall[i] = t
        v.drawable_count = 6
        pl = v.players[seat]
        _set_hand(pl, list(hand13) + ([north] if north is not None else []))
        if seat != 0:  # the dealer's first draw goes back: every seat holds what the reference test gives it
            _set_hand(v.players[0], list(v.players[0].hand[: 13]))
        if riichi:
            pl.riichi_declared = 1
            pl.riichi_declaration_index = 0
            pl.n_discards = 1
            pl.discards[0] = 131
            pl.discard_from_hand_bits = 1
            pl.discard_is_riichi_bits = 1
        v.drawn_tile = north if north is not None else -1
        v.current_player = seat
        v.phase = WAIT_ACT
        v.active_mask = 1 << seat
        v.needs_tsumo = 0

    _poke3(env, mut)
    return env


def _redraw(env, seat, tile):
    """`hand.push(tile); drawn_tile = Some(tile); current_player = seat; WaitAct` of the reference tests."""
    def mut(v):
        pl = v.players[seat]
        h = list(pl.hand[: pl.hand_len]) + [tile]
        _set_hand(pl, h)
        v.drawn_tile = tile
        v.current_player = seat
        v.phase = WAIT_ACT
        v.active_mask = 1 << seat
        v.needs_tsumo = 0
        for q in range(3):   # nobody else is in the middle of a turn
            if q != seat and v.players[q].hand_len + 3 * v.players[q].n_melds == 14:
                v.players[q].hand_len -= 1

    return _poke3(env, mut)


def sc3_kita_tile_none_removes_north(make):
    """tests.rs:2024-2080 (issue #179): a Kita action WITHOUT a tile removes the North tile from the hand (not tile 0),
    files it under kita_tiles, and the rinshan draw restores the hand size."""
    env = _kita_state(make, 0, [36, 40, 44, 48, 52, 56, 60, 64, 68, 72, 76, 80], 120)
    before = env.peek()
    assert any(t // 4 == 30 for t in before.players[0].hand[: before.players[0].hand_len])
    assert find(env.legal(0), KITA) is not None
    env.step({0: pack_action(KITA)})                       # tile = None (quirk Q13: validation accepts it)
    v = env.peek()
    hand = list(v.players[0].hand[: v.players[0].hand_len])
    assert not any(t // 4 == 30 for t in hand)
    assert v.players[0].n_kita == 1 and v.players[0].kita[0] // 4 == 30
    assert len(hand) == before.players[0].hand_len
    assert v.drawn_tile == 36 and evs(env)[-2:] == [{"actor": 0, "pai": "N", "type": "kita"},
                                                      {"actor": 0, "pai": "1p", "type": "tsumo"}]


def sc3_kita_with_correct_tile(make):
    """tests.rs:2192-2222: the same with tile = Some(120)."""
    env = _kita_state(make, 0, [36, 40, 44, 48, 52, 56, 60, 64, 68, 72, 76, 80], 120)
    env.step({0: pack_action(KITA, 120)})
    v = env.peek()
    hand = list(v.players[0].hand[: v.players[0].hand_len])
    assert 120 not in hand and list(v.players[0].kita[: v.players[0].n_kita]) == [120] and len(hand) == 13


def sc3_ankan_available_after_kita_in_riichi(make):
    """tests.rs:2084-2190: riichi seat draws North, declares Kita without a tile, discards the rinshan tile; when it later
    draws the fourth 4p, Ankan is offered (waits unchanged) and Kita is not (no North left)."""
    env = _kita_state(make, 1, [40, 41, 48, 49, 50, 52, 56, 60, 72, 73, 74, 100, 104], 121, riichi=True)
    env.step({1: pack_action(KITA)})
    v = env.peek()
    hand = list(v.players[1].hand[: v.players[1].hand_len])
    assert not any(t // 4 == 30 for t in hand) and len(hand) == 14
    env.step({1: pack_action(DISCARD, v.drawn_tile)})      # tsumogiri of the rinshan tile
    assert env.peek().players[1].hand_len == 13
    _redraw(env, 1, 51)
    v = env.peek()
    assert sum(1 for t in v.players[1].hand[: v.players[1].hand_len] if t // 4 == 12) == 4
    kinds = [unpack_action(a)[0] for a in env.legal(1)]
    assert ANKAN in kinds and KITA not in kinds


def sc3_reach_available_after_kita(make):
    """tests.rs:2224-2285: Riichi stays available after a tile-less Kita (tenpai hand, drawable_count > 0: quirk Q8)."""
    env = _kita_state(make, 1, [40, 44, 48, 52, 56, 60, 76, 80, 84, 88, 92, 96, 100], 122)
    env.step({1: pack_action(KITA)})
    env.step({1: pack_action(DISCARD, env.peek().drawn_tile)})
    _redraw(env, 1, 68)
    assert RIICHI in [unpack_action(a)[0] for a in env.legal(1)]


def sc3_tsumo_available_after_kita(make):
    """tests.rs:2287-2344: after a tile-less Kita the seat still wins by Tsumo on its pair wait."""
    env = _kita_state(make, 2, [40, 44, 48, 52, 56, 60, 76, 80, 84, 88, 92, 96, 108], 123)
    env.step({2: pack_action(KITA)})
    v = env.peek()
    assert v.players[2].hand_len == 14
    rinshan = v.drawn_tile

    def mut(v):   # hand.remove(rinshan); discards.push(rinshan); drawn_tile = None; then draw E (109)
        pl = v.players[2]
        h = [t for t in pl.hand[: pl.hand_len] if t != rinshan]
        _set_hand(pl, h + [109])
        pl.discards[pl.n_discards] = rinshan
        pl.n_discards += 1
        v.drawn_tile = 109
        v.current_player = 2
        v.phase = WAIT_ACT
        v.active_mask = 4

    _poke3(env, mut)
    assert TSUMO in [unpack_action(a)[0] for a in env.legal(2)]


def sc3_ryukyoku_deltas_are_reset_each_round(make):
    """tests.rs:1228-1256: score deltas of an earlier settlement do not leak into the ryukyoku event of the next round
    (suukansansen is kept in 3P, state_3p/mod.rs:1861-1888)."""
    env = make(game_mode=5)
    env.reset()

    def stale(v):
        for p, d in enumerate([2300, -2300, 0]):
            v.players[p].score_delta = d

    _poke3(env, stale)
    env.reset()                                            # _initialize_round resets the deltas
    scattered = [0, 32, 36, 40, 44, 72, 76, 80, 112, 116, 120, 124, 128]
    setup(env, hands=[scattered[:7], scattered[:7], scattered[:], None],
          melds=[[(ANKAN_M, [48, 49, 50, 51], False), (ANKAN_M, [52, 53, 54, 55], False)],
                 [(ANKAN_M, [56, 57, 58, 59], False), (ANKAN_M, [60, 61, 62, 63], False)], [], []],
          current_player=1, drawn_tile=108, reset_kw={})
    env.step({1: pack_action(DISCARD, 108)})
    r = [e for e in evs(env) if e["type"] == "ryukyoku"][-1]
    assert r["reason"] == "suukansansen" and r["deltas"] == [0, 0, 0]


def sc3_dora_wraps_between_1m_and_9m(make):
    """tests.rs:1192-1205 through the state machine: the indicator 1m makes 9m the dora (and 9m makes 1m), so a concealed
    999m / 111m triplet is worth three han at settlement: menzen tsumo + dora 3, 40 fu = mangan, ko win 4000 + 2000."""
    for ind, trip in ((0, "999m"), (32, "111m")):
        env = make(game_mode=5)
        hand = tiles(trip + "234567p234s9s")

        def mut(v, ind=ind):
            v.is_first_turn = 0
            v.n_dora = 1
            v.dora[0] = ind
            v.players[1].n_discards = 1
            v.players[1].discards[0] = 128
            v.current_player = 1
            v.active_mask = 2

        setup(env, oya=0, hands=[None, hand, None, None], current_player=1, drawn_tile=tiles("99s")[1], mutate=mut)
        env.step({1: find(env.legal(1), TSUMO)})
        hora = next(e for e in reversed(evs(env)) if e["type"] == "hora")
        assert hora["deltas"] == [-4000, 6000, -2000], (ind, hora)


def sc3_ankan_dora_before_rinshan(make):
    """tests/env/test_kan_dora_timing_events.py:210-256: ankan -> dora -> tsumo in 3P."""
    env = make(game_mode=3)
    env.reset()

    def mut(v):
        _set_hand(v.players[0], [36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48])
        v.active_mask = 1
        v.current_player = 0
        v.phase = WAIT_ACT
        v.needs_tsumo = 0
        v.drawn_tile = 39

    _poke3(env, mut)
    a = find(env.legal(0), ANKAN)
    assert a is not None
    env.step({0: a})
    t = [e["type"] for e in evs(env)]
    k = t.index("ankan")
    assert t[k + 1: k + 3] == ["dora", "tsumo"]


def sc3_kakan_dora_before_discard(make):
    """tests/env/test_kan_dora_timing_events.py:258-320: kakan -> tsumo -> dora -> dahai in 3P."""
    env = make(game_mode=3)
    env.reset()

    def mut(v):
        _set_hand(v.players[0], [39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49])
        v.players[0].n_melds = 1
        set_meld(v.players[0].melds[0], PON_M, [36, 37, 38], True)
        v.active_mask = 1
        v.current_player = 0
        v.phase = WAIT_ACT
        v.needs_tsumo = 0
        v.drawn_tile = 39

    _poke3(env, mut)
    a = find(env.legal(0), KAKAN)
    assert a is not None
    env.step({0: a})
    if env.status()[1] == WAIT_RESPONSE:   # a chankan offer: everybody passes
        env.step({s: pack_action(PASS) for s in range(3) if (env.status()[0] >> s) & 1})
    d = find(env.legal(0), DISCARD)
    assert d is not None
    env.step({0: d})
    t = [e["type"] for e in evs(env)]
    k = t.index("kakan")
    rest = t[k + 1:]
    assert rest.index("tsumo") < rest.index("dora") < rest.index("dahai")


def sc3_initialization(make):
    """tests/env/test_sanma.py:49-117: dealer 14 tiles, others 13, 68 tiles left, no 2m-8m anywhere, the log opens with
    start_game / start_kyoku / tsumo; every sanma game mode."""
    for mode in (3, 4, 5):
        env = make(game_mode=mode)
        env.reset()
        v = env.peek()
        assert [v.players[p].hand_len for p in range(3)] == [14, 13, 13] and v.players[3].hand_len == 0
        assert v.wall_len == 68 and v.phase == WAIT_ACT and v.current_player == 0 and env.status()[0] == 1
        seen = set(v.wall[: v.wall_len]) | set(v.dora[: v.n_dora])
        for p in range(3):
            seen |= set(v.players[p].hand[: v.players[p].hand_len])
        assert len(seen) == 108 and not any(1 <= t // 4 <= 7 for t in seen)
        assert [e["type"] for e in evs(env)[:3]] == ["start_game", "start_kyoku", "tsumo"]
        assert env.scores()[:3] == [35000] * 3


def sc3_tsumo_deltas(make):
    """tests/env/test_sanma.py:403-432: a Tsumo in 3P is paid by the two other seats, deltas sum to zero."""
    env = make(game_mode=5)
    env.reset()

    def mut(v):
        _set_hand(v.players[0], sorted([36, 37, 38, 40, 41, 42, 44, 45, 46, 32, 33, 34, 0]) + [1])
        v.drawn_tile = 1
        v.current_player = 0
        v.is_first_turn = 0
        v.players[0].discards[v.players[0].n_discards] = 100
        v.players[0].n_discards += 1

    _poke3(env, mut)
    assert find(env.legal(0), TSUMO) is not None
    env.step({0: pack_action(TSUMO)})
    hora = next(e for e in reversed(evs(env)) if e["type"] == "hora")
    d = hora["deltas"]
    assert hora["tsumo"] is True and len(d) == 3 and d[0] > 0 and d[1] < 0 and d[2] < 0 and sum(d) == 0


def sc3_play_full_round(make):
    """tests/env/test_sanma.py:480-499 (+514-530): Tsumo when offered, otherwise discard the last tile, pass every claim,
    until the single-round game is over; the seats' masked logs carry three tehais."""
    env = make(game_mode=3, seed=7)
    env.reset()
    turns = 0
    while not env.status()[2] and turns < 200:
        v = env.peek()
        pid = v.current_player
        ts = find(env.legal(pid), TSUMO)
        if ts is not None:
            env.step({pid: ts})
        else:
            env.step({pid: pack_action(DISCARD, v.players[pid].hand[v.players[pid].hand_len - 1])})
        while env.status()[1] == WAIT_RESPONSE and not env.status()[2]:
            env.step({s: pack_action(PASS) for s in range(3) if (env.status()[0] >> s) & 1})
        turns += 1
    assert env.status()[2] == 1
    sk = [json.loads(x) for x in env.log(1) if '"start_kyoku"' in x][0]
    assert len(sk["tehais"]) == 3 and sk["tehais"][0] == ["?"] * 13 and sk["tehais"][1] != ["?"] * 13
    assert evs(env)[-1]["type"] == "end_game"


# Reference tests whose fixture is a hand no game can reach - twelve or thirteen copies of ONE tile id (tests/env/actions/
# test_riichi_no_claim.py and tests/test_riichi_autoplay.py fill hands with [0] * 12 / [0] * 13): the oracle follows the reference there (fourteen 1m are a complete hand), the HIP path
# keeps type counts in 3-bit fields and is not run on them.
SCENARIOS_ORACLE_ONLY = [sc_riichi_no_pon_claim, sc_riichi_no_chi_claim, sc_riichi_autoplay_waits_for_the_discard]

SCENARIOS_3P = [sc_mjsoul_3p_ron_pao_composite, sc3_basics, sc3_no_chi, sc3_kita, sc3_oyayame_needs_40000, sc3_tsumo_payments_and_nukidora,
                sc3_exhaustive_draw_pool_2000, sc3_pon_and_rotation, sc3_ron_deltas, sc3_kita_tile_none_removes_north,
                sc3_kita_with_correct_tile, sc3_ankan_available_after_kita_in_riichi, sc3_reach_available_after_kita,
                sc3_tsumo_available_after_kita, sc3_ryukyoku_deltas_are_reset_each_round, sc3_dora_wraps_between_1m_and_9m,
                sc3_ankan_dora_before_rinshan, sc3_kakan_dora_before_discard, sc3_initialization, sc3_tsumo_deltas,
                sc3_play_full_round]
