"""Yaku ids and their names (yaku.rs:13-127, ids :131-180): `Yaku`, `get_yaku_by_id`, `get_all_yaku` under the reference's names.
id = the evaluator's yaku id = Mahjong Soul's fan id; tenhou_id = the index in Tenhou's yaku list (seat / round wind yakuhai:
the entry of East, 10 / 14 - Tenhou numbers the four winds separately)."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class Yaku:
    id: int
    name: str
    name_en: str
    tenhou_id: int
    mjsoul_id: int

    def __repr__(self):
        return f"Yaku(id={self.id}, name='{self.name}', name_en='{self.name_en}', tenhou_id={self.tenhou_id}, mjsoul_id={self.mjsoul_id})"


# (id, name, name_en, tenhou_id); mjsoul_id == id
_ROWS = (
    (1, "門前清自摸和", "Menzen Tsumo", 0), (2, "立直", "Riichi", 1), (3, "槍槓", "Chankan", 3), (4, "嶺上開花", "Rinshan Kaihou", 4),
    (5, "海底摸月", "Haitei Raoyue", 5), (6, "河底撈魚", "Houtei Raoyui", 6), (7, "役牌 白", "Yakuhai (haku)", 18), (8, "役牌 發", "Yakuhai (hatsu)", 19),
    (9, "役牌 中", "Yakuhai (chun)", 20), (10, "自風牌", "Yakuhai (seat wind)", 10), (11, "場風牌", "Yakuhai (round wind)", 14), (12, "断幺九", "Tanyao", 8),
    (13, "一盃口", "Iipeiko", 9), (14, "平和", "Pinfu", 7),
    (15, "混全帯幺九", "Chantai", 23), (16, "一気通貫", "Ittsu", 24), (17, "三色同順", "Sanshoku Doujun", 25), (18, "ダブル立直", "Double Riichi", 21),
    (19, "三色同刻", "Sanshoku Doukou", 26), (20, "三槓子", "San Kantsu", 27), (21, "対々和", "Toitoi", 28), (22, "三暗刻", "San Ankou", 29),
    (23, "小三元", "Shou Sangen", 30), (24, "混老頭", "Honroutou", 31), (25, "七対子", "Chiitoitsu", 22),
    (26, "純全帯幺九", "Junchan", 33), (27, "混一色", "Honitsu", 34), (28, "二盃口", "Ryanpeikou", 32),
    (29, "清一色", "Chinitsu", 35),
    (30, "一発", "Ippatsu", 2), (31, "ドラ", "Dora", 52), (32, "赤ドラ", "Aka Dora", 54), (33, "裏ドラ", "Ura Dora", 53), (34, "抜きドラ", "Nuki Dora", 52),
    (35, "天和", "Tenhou", 37), (36, "地和", "Chiihou", 38), (37, "大三元", "Dai Sangen", 39), (38, "四暗刻", "Su Ankou", 40), (39, "字一色", "Tsuu iisou", 42),
    (40, "緑一色", "Ryuu iisou", 43), (41, "清老頭", "Chinroutou", 44), (42, "国士無双", "Kokushi Musou", 47), (43, "小四喜", "Sho Suusi", 50),
    (44, "四槓子", "Su Kantsu", 51), (45, "九蓮宝燈", "Chuuren Poutou", 45),
    (47, "純正九蓮宝燈", "Junsei Chuuren Poutou", 46), (48, "四暗刻単騎", "Su Ankou Tanki", 41), (49, "国士無双十三面待ち", "Kokushi Musou 13-men", 48),
    (50, "大四喜", "Dai Suusi", 49),
)
_TABLE = tuple(Yaku(i, n, e, t, i) for i, n, e, t in _ROWS)
_BY_ID = {y.id: y for y in _TABLE}


def get_yaku_by_id(id_: int):
    return _BY_ID.get(int(id_))


def get_all_yaku():
    return list(_TABLE)
