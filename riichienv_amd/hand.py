"""The scalar hand API of the reference's Python package under its names (src/riichienv/hand.py, parser.rs, _riichienv.pyi):
Conditions, HandEvaluator / HandEvaluator3P (.calc, .is_tenpai, .get_waits, .hand_from_text, .calc_from_text), parse_hand /
parse_tile, calculate_score, calculate_shanten(_3p), check_riichi_candidates.  Every call is a batch of one (or a few) through
the C-ABI's batched hand math on the GPU - rmj_eval_hands, rmj_calculate_score, rmj_shanten; code that evaluates many hands
should call those batches directly (riichienv_amd.vecenv.eval_hands, ...).  No CPU fallback: without the library these raise."""
from __future__ import annotations

import dataclasses
import enum
from dataclasses import dataclass

import numpy as np

from . import abi, vecenv
from .compat import Meld, MeldType


class Wind(enum.IntEnum):  # types.rs:28-33
    East = 0
    South = 1
    West = 2
    North = 3


WINDS = [Wind.East, Wind.South, Wind.West, Wind.North]


@dataclass
class Conditions:  # src/riichienv/hand.py:14-36, types.rs:193-210
    tsumo: bool = False
    riichi: bool = False
    double_riichi: bool = False
    ippatsu: bool = False
    haitei: bool = False
    houtei: bool = False
    rinshan: bool = False
    chankan: bool = False
    tsumo_first_turn: bool = False
    player_wind: int | Wind = 0
    round_wind: int | Wind = 0
    riichi_sticks: int = 0
    honba: int = 0
    kita_count: int = 0
    is_sanma: bool = False
    num_players: int = 4


class WinResult:  # types.rs:282-293
    __slots__ = ("is_win", "yakuman", "ron_agari", "tsumo_agari_oya", "tsumo_agari_ko", "yaku", "han", "fu", "pao_payer", "has_win_shape")

    def __init__(self, r: abi.HandResult):
        self.is_win, self.yakuman, self.has_win_shape = bool(r.is_win), bool(r.yakuman), bool(r.has_win_shape)
        self.ron_agari, self.tsumo_agari_oya, self.tsumo_agari_ko = int(r.ron_agari), int(r.tsumo_agari_oya), int(r.tsumo_agari_ko)
        self.yaku = [int(y) for y in r.yaku[: r.n_yaku]]
        self.han, self.fu = int(r.han), int(r.fu)
        self.pao_payer = None

    def yaku_list(self):
        """the Yaku entries of the ids in `yaku` (ids without an entry are skipped)"""
        from .yaku_table import get_yaku_by_id

        return [y for y in (get_yaku_by_id(i) for i in self.yaku) if y is not None]

    def __repr__(self):
        return f"WinResult(is_win={self.is_win}, han={self.han}, fu={self.fu}, yaku={self.yaku}, ron_agari={self.ron_agari})"


# ---- parser.rs:9-300: "123m456p789s111z2z", melds "(123m0)" chi, "(p5z1)" pon, "(k2z)" ankan / "(k2z1)" daiminkan, "(s3p2)" kakan
class _TileManager:
    """parser.rs:9-41: hands out the copies of a type; a plain five skips the red copy 0 while it can"""

    def __init__(self):
        self.used = [[False] * 4 for _ in range(34)]

    def get(self, t34, red):
        if t34 >= 34:
            raise ValueError(f"Invalid tile ID: {t34}")
        five = t34 in (4, 13, 22)
        order = (0,) if (five and red) else ((1, 2, 3, 0) if five else (0, 1, 2, 3))
        for k in order:
            if not self.used[t34][k]:
                self.used[t34][k] = True
                return t34 * 4 + k
        raise ValueError(f"No more copies of tile {t34}")


_SUIT = {"m": 0, "p": 9, "s": 18, "z": 27}


def _digit_tile(d, off):
    v = int(d)
    return (off + 4, True) if v == 0 else (off + v - 1, False)


def _parse_meld(content, tm):
    prefix = content[0] if content[:1] in ("p", "k", "s") else " "
    rest = content[1:] if prefix != " " else content
    i = 0
    while i < len(rest) and rest[i].isdigit():
        i += 1
    digits, suit = rest[:i], (rest[i] if i < len(rest) else " ")
    call_idx = int(rest[i + 1]) if i + 1 < len(rest) and rest[i + 1].isdigit() else 0
    if suit not in _SUIT:
        raise ValueError(f"Invalid suit in meld: {suit}")
    off = _SUIT[suit]
    if prefix == " ":
        if len(digits) != 3:
            raise ValueError("Chi meld requires 3 digits")
        return Meld(MeldType.Chi, sorted(tm.get(*_digit_tile(d, off)) for d in digits), True, -1, None)
    base, red = _digit_tile(digits[0], off)
    count = 3 if prefix == "p" else 4
    tiles, got_red = [], False
    if red:
        tiles.append(tm.get(base, True))
        got_red = True
    while len(tiles) < count:
        try:
            tiles.append(tm.get(base, False))
        except ValueError:
            if got_red:
                raise ValueError(f"Not enough tiles for meld of {base}")
            try:
                tiles.append(tm.get(base, True))
            except ValueError:
                raise ValueError(f"Not enough tiles for meld of {base}")
            got_red = True
    kind = MeldType.Pon if prefix == "p" else (MeldType.Kakan if prefix == "s" else (MeldType.Ankan if call_idx == 0 else MeldType.Daiminkan))
    return Meld(kind, sorted(tiles), kind != MeldType.Ankan, -1, None)


def parse_hand(text: str):
    """parser.rs:43-102 -> (tile ids, melds); ValueError like the binding's PyValueError"""
    tm = _TileManager()
    tiles, melds, pending = [], [], []
    i = 0
    while i < len(text):
        c = text[i]
        if c == "(":
            j = text.find(")", i)
            j = len(text) if j < 0 else j
            melds.append(_parse_meld(text[i + 1: j], tm))
            i = j + 1
            continue
        if c.isdigit() and c.isascii():
            pending.append(c)
        elif c in _SUIT:
            tiles += [tm.get(*_digit_tile(d, _SUIT[c])) for d in pending]
            pending = []
        i += 1
    if pending:
        raise ValueError("Pending digits without suit")
    return tiles, melds


def parse_tile(text: str) -> int:
    """parser.rs:111-135"""
    tiles, melds = parse_hand(text)
    if melds:
        raise ValueError("parse_tile expects a single tile, but found meld syntax in input")
    if not tiles:
        raise ValueError("No tile found in string")
    if len(tiles) != 1:
        raise ValueError(f"Expected exactly one tile, but found {len(tiles)} tiles in string")
    return tiles[0]


def _hand_case(tiles, melds, win_tile=0, dora=(), ura=(), cond: Conditions | None = None, sanma=False):
    c = cond or Conditions()
    hc = abi.HandCase()
    hc.n_tiles = len(tiles)
    for i, t in enumerate(tiles[:14]):
        hc.tiles[i] = t
    hc.n_melds = len(melds)
    for i, m in enumerate(melds[:4]):
        mv = hc.melds[i]
        mv.meld_type, mv.n_tiles, mv.opened = int(m.meld_type), len(m.tiles), 1 if m.opened else 0
        for j, t in enumerate(m.tiles[:4]):
            mv.tiles[j] = t
        mv.from_who = m.from_who if m.from_who is not None else -1
        mv.called_tile = -1 if m.called_tile is None else m.called_tile
    hc.win_tile = win_tile
    hc.n_dora, hc.n_ura = min(len(dora), 5), min(len(ura), 5)
    for i, t in enumerate(list(dora)[:5]):
        hc.dora[i] = t
    for i, t in enumerate(list(ura)[:5]):
        hc.ura[i] = t
    for k in ("tsumo", "riichi", "double_riichi", "ippatsu", "haitei", "houtei", "rinshan", "chankan", "tsumo_first_turn"):
        setattr(hc, k, 1 if getattr(c, k) else 0)
    hc.player_wind, hc.round_wind = int(c.player_wind) % 4, int(c.round_wind) % 4
    hc.honba, hc.kita_count, hc.is_sanma = int(c.honba), int(c.kita_count), 1 if sanma else 0
    return hc


class HandEvaluator:
    """hand_evaluator.rs:24-213 through rmj_eval_hands"""

    _SANMA = False

    def __init__(self, tiles, melds=None):
        self.tiles_136 = list(tiles)
        self.melds = list(melds or [])

    @classmethod
    def hand_from_text(cls, text: str):
        """src/riichienv/hand.py:45-66: 13 tiles plus one per kan"""
        tiles, melds = parse_hand(text)
        kans = sum(m.meld_type in (MeldType.Daiminkan, MeldType.Ankan, MeldType.Kakan) for m in melds)
        have = len(tiles) + sum(len(m.tiles) for m in melds)
        if have != 13 + kans:
            raise ValueError(f"Hand must have {13 + kans} tiles (got {have})")
        return cls(sorted(tiles), melds)

    @classmethod
    def calc_from_text(cls, text: str, dora_indicators: str | None = None, conditions: Conditions | None = None, ura_indicators: str | None = None):
        """src/riichienv/hand.py:92-131: the last standing tile of a 14-tile text is the winning tile"""
        tiles, melds = parse_hand(text)
        if not tiles and not melds:
            raise ValueError("Empty hand")
        if not tiles:
            raise ValueError("No standing tiles to check for win tile")
        win = tiles[-1]
        dora = sorted(parse_hand(dora_indicators)[0]) if dora_indicators else []
        ura = sorted(parse_hand(ura_indicators)[0]) if ura_indicators else []
        return cls(sorted(tiles), melds).calc(win, dora, conditions, ura)

    @staticmethod
    def _digit(t):
        return 0 if t in (16, 52, 88) else (t // 4) % 9 + 1 if t // 4 < 27 else t // 4 - 26

    def to_text(self) -> str:
        """src/riichienv/hand.py:68-90, 133-244: standing tiles grouped by suit (a red five is 0), then the melds as
        "(123m0)" / "(p1z0)" / "(k2z0)" / "(s3p0)" - the call index is not kept by a Meld and is written as 0"""
        out = ""
        for k, ch in enumerate("mpsz"):
            ds = [self._digit(t) for t in sorted(self.tiles_136) if t // 36 == k or (k == 3 and t >= 108)]
            if ds:
                out += "".join(map(str, ds)) + ch
        for m in self.melds:
            ch = "mpsz"[min(m.tiles[0] // 36, 3)]
            if m.meld_type == MeldType.Chi:
                out += "(" + "".join(str(self._digit(t)) for t in m.tiles) + ch + "0)"
            else:
                d = 0 if any(t in (16, 52, 88) for t in m.tiles) else self._digit(m.tiles[0])
                pre = {MeldType.Pon: "p", MeldType.Daiminkan: "k", MeldType.Kakan: "s", MeldType.Ankan: "k"}.get(m.meld_type, "")
                out += f"({pre}{d}{ch}0)"
        return out

    def calc(self, win_tile: int, dora_indicators=None, conditions: Conditions | None = None, ura_indicators=None) -> WinResult:
        """src/riichienv/hand.py:246-277: a 13-tile hand gets the winning tile added first"""
        tiles = self.tiles_136
        if (len(tiles) + sum(len(m.tiles) for m in self.melds)) % 3 == 1:
            tiles = sorted(tiles + [win_tile])
        cond = conditions or Conditions()
        if self._SANMA:
            cond = dataclasses.replace(cond, is_sanma=True, num_players=3)
        hc = _hand_case(tiles, self.melds, win_tile, dora_indicators or [], ura_indicators or [], cond, self._SANMA)
        return WinResult(vecenv.eval_hands([hc])[0])

    def _probe(self):
        return vecenv.eval_hands([_hand_case(self.tiles_136, self.melds, sanma=self._SANMA)])[0]

    def is_tenpai(self) -> bool:
        return bool(self._probe().is_tenpai)

    def get_waits(self):
        w = int(self._probe().waits)
        return [t for t in range(34) if (w >> t) & 1]

    get_waits_u8 = get_waits


class HandEvaluator3P(HandEvaluator):
    """hand_evaluator_3p.rs (sanma dora wrap, kita as dora, two payers)"""

    _SANMA = True


@dataclass
class Score:  # score.rs:5-11
    total: int
    pay_ron: int
    pay_tsumo_oya: int
    pay_tsumo_ko: int


def calculate_score(han: int, fu: int, is_oya: bool, is_tsumo: bool, honba: int, num_players: int = 4) -> Score:
    """score.rs:13-52 (rmj_calculate_score)"""
    r = vecenv.calculate_score([han], [fu], [int(is_oya)], [int(is_tsumo)], [honba], [num_players])[0]
    return Score(int(r[0]), int(r[1]), int(r[2]), int(r[3]))


def _counts(tiles):
    c = np.zeros((1, 34), np.uint8)
    for t in tiles:
        if t // 4 < 34:
            c[0, t // 4] += 1
    return c


def calculate_shanten(hand_tiles) -> int:
    """shanten.rs:250-261 (rmj_shanten): -1 = complete"""
    return int(vecenv.shanten(_counts(hand_tiles), False)[0])


def calculate_shanten_3p(hand_tiles) -> int:
    """shanten.rs:470-484"""
    return int(vecenv.shanten(_counts(hand_tiles), True)[0])


def check_riichi_candidates(tiles_136):
    """hand_evaluator.rs:263-284: the tiles whose discard leaves a tenpai hand (agari::is_tenpai of the other tiles), one GPU batch.
    A concealed hand of 14 goes through rmj_agari_counts; shorter hands (melds not given) are padded with honor triplets that
    cannot interact with the rest, which leaves the probe of the concealed part unchanged."""
    tiles = list(tiles_136)
    if not tiles:
        return []
    rest = [[t for j, t in enumerate(tiles) if j != i] for i in range(len(tiles))]
    if len(tiles) == 14:
        counts = np.zeros((14, 34), np.uint8)
        for i, r in enumerate(rest):
            for t in r:
                counts[i, t // 4] += 1
        tenpai = vecenv.agari_counts(counts)[1]
        return [int(t) for t, ok in zip(tiles, tenpai) if ok]
    missing = (13 - (len(tiles) - 1)) // 3
    free = [k for k in range(27, 34) if not any(t // 4 == k for t in tiles)][:missing]
    pads = [Meld(MeldType.Pon, [k * 4, k * 4 + 1, k * 4 + 2], True, -1, None) for k in free]
    res = vecenv.eval_hands([_hand_case(r, pads) for r in rest])
    return [int(t) for t, x in zip(tiles, res) if x.is_tenpai]
