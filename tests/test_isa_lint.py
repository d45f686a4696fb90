"""The ISA lint of round 6 (scripts/lint_isa_last_vgpr.py): gfx950 returns wrong results now and then for a 64-bit shift whose amount sits in the
last vector register a wave is allocated (scripts/micro/ballot_shift_hazard.hip, docs/journal_r06.md section 1).  `__graft_entry__.build()` refuses
to install a library that contains one; here: the lint recognises the pattern, and the library that ships does not have it."""
import os
import shutil

import pytest

from scripts import lint_isa_last_vgpr as lint

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(HERE), "riichienv_amd", "libriichi_mi355x.so")

ASM = """
\t.text
_ZN4rmj418step4_call_enc_oolILb0ELi0EEEvPK3Env:
\tv_cmp_ne_u32_e32 vcc, 0, v0
\ts_nop 1
\tv_lshrrev_b64 v[2:3], v{amount}, vcc
\tv_bfe_u32 v0, v2, v48, 16
\ts_setpc_b64 s[30:31]
\t.set .L_ZN4rmj418step4_call_enc_oolILb0ELi0EEEvPK3Env.num_vgpr, 88
k_step4_act_enc:
\ts_swappc_b64 s[30:31], s[0:1]
\ts_endpgm
\t.amdhsa_kernel k_step4_act_enc
\t\t.amdhsa_next_free_vgpr {next_free}
\t.end_amdhsa_kernel
\t.set k_step4_act_enc.num_vgpr, max(32, .L_ZN4rmj418step4_call_enc_oolILb0ELi0EEEvPK3Env.num_vgpr)
k_other:
\tv_lshrrev_b64 v[2:3], v{amount}, vcc
\ts_endpgm
\t.amdhsa_kernel k_other
\t\t.amdhsa_next_free_vgpr 120
\t.end_amdhsa_kernel
\t.set k_other.num_vgpr, 120
"""


@pytest.mark.parametrize("amount, next_free, findings", [(87, 88, 1),      # what round 5's -disable-machine-licm build of k_step4_act_enc had
                                                         (86, 88, 0),      # one register lower: never wrong in the probe
                                                         (87, 89, 0),      # the same instruction in a wave that is allocated 96 registers
                                                         (87, 81, 1),      # 81 named registers are 88 allocated
                                                         (119, 88, 1)])    # ... and the other kernel (120 allocated) shifts by v119
def test_lint_finds_the_shift_by_the_last_allocated_register(tmp_path, amount, next_free, findings):
    p = tmp_path / "case.s"
    p.write_text(ASM.format(amount=amount, next_free=next_free))
    summary, bad = lint.lint(str(p))
    assert len(bad) == findings, (summary, bad)
    if findings and amount == 87:
        assert "k_step4_act_enc" in bad[0][0] and "step4_call_enc_ool" in bad[0][0] and bad[0][1] == 87


@pytest.mark.skipif(not os.path.exists(lint.OBJDUMP) and not shutil.which("llvm-objdump"), reason="llvm-objdump not installed")
def test_the_library_that_ships_has_no_such_shift():
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    summary, bad = lint.lint(LIB)
    assert not bad, (summary, bad[:4])
    assert " 0 take the amount" in summary
