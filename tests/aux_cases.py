"""Hand-built states for the auxiliary encoders (kawa overview / yaku possibility): shared by the oracle pinning test
and the GPU parity test.  Each case: (discards[4], melds[4] as (meld_type, tiles), dora indicators, round_wind, oya)."""
from riichienv_amd import abi

PON, CHI, ANKAN, DAIMINKAN = abi.MELD_PON, abi.MELD_CHI, abi.MELD_ANKAN, abi.MELD_DAIMINKAN

CASES = {
    # yaku_checker.rs:419-466 unit tests
    "no_melds": dict(),
    "pon_1m": dict(melds=[[(PON, [0, 1, 2])], [], [], []]),
    "pon_5m": dict(melds=[[(PON, [16, 17, 18])], [], [], []]),
    "chi_123m": dict(melds=[[(CHI, [0, 4, 8])], [], [], []]),
    # yakuhai: three whites visible (two own discards + indicator) -> impossible for that seat only; a set wins over it
    "white_dead": dict(discards=[[124, 125], [126], [], []], dora=[127]),
    "white_set_and_dead": dict(discards=[[], [], [], []], melds=[[(PON, [124, 125, 126])], [], [], []], dora=[127]),
    # winds: round = S, oya = 2 -> seat winds E at seat 2, S at 3, W at 0, N at 1
    "winds": dict(discards=[[116, 117, 118], [120, 121, 122], [108, 109, 110], [112, 113, 114]], round_wind=1, oya=2),
    # flushes
    "one_suit": dict(melds=[[(CHI, [36, 40, 44])], [(PON, [72, 73, 74]), (PON, [108, 109, 110])], [(CHI, [0, 4, 8]), (PON, [40, 41, 42])],
                            [(PON, [124, 125, 126])]]),
    # dragons: green fully visible (shousangen dead), red two visible without a set (daisangen dead)
    "dragons": dict(discards=[[128, 129, 130, 131], [132, 133], [], []], melds=[[], [], [(PON, [124, 125, 126])], []]),
    # kokushi: all four 9s visible in own river; kans; junchan / chanta / honroutou material
    "kokushi_dead": dict(discards=[[104, 105, 106, 107], [100], [], []], dora=[]),
    "outside": dict(melds=[[(CHI, [0, 4, 8]), (PON, [32, 33, 34])], [(CHI, [4, 8, 12])], [(DAIMINKAN, [108, 109, 110, 111])],
                           [(ANKAN, [68, 69, 70, 71]), (CHI, [96, 100, 104])]]),
    # kawa: repeated types (count channels), the reference's "red" ids 20 / 24 / 28 and the real reds 16 / 52 / 88
    "kawa": dict(discards=[[0, 1, 2, 3, 4, 20], [24, 28, 16, 52, 88], [33, 34, 35, 132, 133], [135, 134]]),
}


def apply_case(view, case, sanma=False):
    """write a case into a state view (tests/scenarios.setup mutate hook)"""
    for p in range(4):
        d = (case.get("discards") or [[]] * 4)[p]
        view.players[p].n_discards = len(d)
        for i, t in enumerate(d):
            view.players[p].discards[i] = t
        ms = (case.get("melds") or [[]] * 4)[p]
        view.players[p].n_melds = len(ms)
        for i, (mt, ts) in enumerate(ms):
            m = view.players[p].melds[i]
            m.meld_type, m.n_tiles, m.opened, m.from_who, m.called_tile = mt, len(ts), int(mt != ANKAN), (p + 1) % 4, -1
            for k, t in enumerate(ts):
                m.tiles[k] = t
        # keep hand + melds = 13 tiles so the state stays well-formed
        view.players[p].hand_len = 13 - 3 * len(ms)
    dora = case.get("dora", [])
    view.n_dora = len(dora)
    for i, t in enumerate(dora):
        view.dora[i] = t
    view.round_wind = case.get("round_wind", 0)
    view.oya = case.get("oya", 0)
