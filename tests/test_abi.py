"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol include/riichi_mi355x.h
declares, struct layouts agree with the header, and compute entry points fail loudly without a GPU."""
import ctypes as C
import os
import re

import pytest

from riichienv_amd import abi, vecenv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    # the drop-in surface + the measurement / test hooks (declared apart: riichi_mi355x_bench.h)
    hdr = open(os.path.join(ROOT, "include", "riichi_mi355x.h")).read() + open(os.path.join(ROOT, "include", "riichi_mi355x_bench.h")).read()
    product = set(re.findall(r"\b(rmj_[a-z_]+)\s*\(", open(os.path.join(ROOT, "include", "riichi_mi355x.h")).read()))
    assert not {s for s in product if s.startswith(("rmj_bench_", "rmj_time_"))}, "measurement hooks belong in riichi_mi355x_bench.h"
    declared = set(re.findall(r"\b(rmj_[a-z_]+)\s*\(", hdr))
    declared -= {"rmj_env"}
    assert declared == set(vecenv.EXPORTS), declared ^ set(vecenv.EXPORTS)
    lib = vecenv.load_lib()
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert b"gfx950" in lib.rmj_version()


def test_struct_sizes_match_header():
    # sizes implied by the C declarations (natural alignment)
    assert C.sizeof(abi.Event) == 32
    # compile-time check against the real header through the oracle's C++ translation unit sizes
    import subprocess
    import tempfile

    src = r'''
#include <cstdio>
#include "riichi_mi355x.h"
int main(){printf("%zu %zu %zu %zu %zu %zu\n", sizeof(RmjStateView), sizeof(RmjPlayerView), sizeof(RmjHandCase), sizeof(RmjHandResult), sizeof(RmjConfig), sizeof(RmjEvent));}
'''
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "s.cpp")
        open(p, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["g++", "-I", os.path.join(ROOT, "include"), p, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    got = [C.sizeof(x) for x in (abi.StateView, abi.PlayerView, abi.HandCase, abi.HandResult, abi.Config, abi.Event)]
    assert got == [int(x) for x in out], (got, out)


def test_pack_unpack_roundtrip():
    a = abi.pack_action(abi.CHI, 57, [65, 62])
    assert abi.unpack_action(a) == (abi.CHI, 57, [62, 65])  # Action::new sorts (action.rs:97-98)
    assert abi.unpack_action(abi.pack_action(abi.RIICHI)) == (abi.RIICHI, None, [])


def test_no_cpu_fallback():
    """Without a HIP device the product path must fail loudly (never route through the oracle)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = vecenv.load_lib()
    assert lib.rmj_device_count() == 0
    with pytest.raises(vecenv.RmjError):
        vecenv.VecRiichiEnv(4)
    with pytest.raises(vecenv.RmjError):
        vecenv.eval_hands([abi.HandCase()])
    src = open(os.path.join(ROOT, "riichienv_amd", "vecenv.py")).read()
    assert "oracle" not in src.replace("oracle/oracle.py", "")


def _build_c_example(tmp_path):
    import subprocess

    exe = tmp_path / "rollout"
    lib_dir = os.path.join(ROOT, "riichienv_amd")
    cmd = ["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "rollout.c"), "-o", str(exe),
           "-L" + lib_dir, "-l:libriichi_mi355x.so", "-Wl,-rpath," + lib_dir]
    subprocess.check_call(cmd)
    return str(exe)


def test_plain_c_program_links_against_the_boundary_and_fails_loudly_without_a_gpu(tmp_path):
    """examples/rollout.c sees nothing but include/riichi_mi355x.h (C, not C++): it must compile warning-free with gcc, link
    against the shared library, and - in this container, without a GPU - stop at rmj_create with the library's error text
    instead of computing anything on the CPU."""
    import subprocess

    import torch

    exe = _build_c_example(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 1 and "no HIP device" in p.stderr and "0 device(s)" in p.stdout


@pytest.mark.gpu
def test_plain_c_program_runs_a_rollout(tmp_path):
    import subprocess

    p = subprocess.run([_build_c_example(tmp_path)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert "env.step calls advanced a game" in p.stdout and '"type"' in p.stdout
