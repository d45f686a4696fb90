"""ISA lint (round 6): no 64-bit VALU shift may take its 32-bit shift amount from the LAST vector register a wave is allocated.

Why.  On gfx950 (MI355X, ROCm 7.2) `v_lshrrev_b64 vdst, vN, <64-bit source>` with vN = the wave's last allocated VGPR returns a
wrong result now and then in waves that are not the first of their SIMD (scripts/micro/ballot_shift_hazard.hip: ~1e-3 of the
executions in wave slots >= 3, none in slot 0, none with the amount one register lower or copied to another register first; the
allocation size does not matter, only "last").  The row ballot of the four-games-per-wave step used to compile to exactly that
instruction; the build with -mllvm -disable-machine-licm put the amount into v87 of the 88-register k_step4_act_enc and published legal
lists with entries missing (docs/journal_r06.md section 1).  The source no longer shifts ballots (row_ballot16, rmj_hand.hip.h: one
v_perm_b32), but wait masks and tile masks are still shifted by per-lane amounts all over the step, and which register the allocator
picks for an amount changes with every compiler flag - so every build is checked.

Two modes:
  * a shared library / code object (what `__graft_entry__.build()` and tests/test_isa_lint.py run on the library that ships): the gfx950
    code object is disassembled with llvm-objdump; a shift is reported when its amount register is the HIGHEST vector register its own
    function names and that index is 7 (mod 8).  Sound without a call graph: a function that names v0 .. vM runs only in kernels that
    allocate >= M + 1 registers, allocations are multiples of 8, so "amount = the kernel's last register" implies amount = M = 7 (mod 8).
    Conservative: a kernel that allocates more than the function needs is safe although the rule fires.
  * device assembly (`hipcc ... --cuda-device-only -S`): exact - per kernel alloc = next_free_vgpr rounded up to 8, every function the
    kernel reaches (the `.set <f>.num_vgpr, max(..., <callee>.num_vgpr)` expressions give the call graph) is scanned for the amount
    v<alloc - 1>.
Exit status 1 when anything is found.  usage: python scripts/lint_isa_last_vgpr.py <lib.so | file.co | file.s>"""
import os
import re
import struct
import subprocess
import sys
import tempfile

SHIFT = re.compile(r"^\s*(v_lshrrev_b64|v_lshlrev_b64|v_ashrrev_i64)\s+v\[\d+:\d+\],\s*v(\d+)\b")
OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")


def extract_code_object(path):
    """the gfx950 code object of a HIP fat binary (clang offload bundle inside the host library), or the file itself if it is one"""
    data = open(path, "rb").read()
    i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
    if i < 0:
        return data
    n, = struct.unpack_from("<Q", data, i + 24)
    off = i + 32
    for _ in range(n):
        o, s, t = struct.unpack_from("<QQQ", data, off)
        off += 24
        triple = data[off:off + t].decode()
        off += t
        if "gfx950" in triple:
            return data[i + o:i + o + s]
    raise RuntimeError(f"{path}: no gfx950 code object in the offload bundle")


def lint_object(path):
    co = extract_code_object(path)
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co)
        f.flush()
        dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", "--no-show-raw-insn", "--no-leading-addr", f.name], capture_output=True, text=True,
                             check=True).stdout
    funcs, cur = {}, None
    for line in dis.split("\n"):
        m = re.match(r"^(?:[0-9a-f]+ )?<(.+)>:$", line.strip())
        if m:
            cur = m.group(1)
            funcs[cur] = {"max": -1, "shifts": []}
            continue
        if cur is None:
            continue
        code = line.split("//")[0]
        f = funcs[cur]
        for r in re.finditer(r"\bv(\d+)\b", code):
            f["max"] = max(f["max"], int(r.group(1)))
        for r in re.finditer(r"\bv\[(\d+):(\d+)\]", code):
            f["max"] = max(f["max"], int(r.group(2)))
        m = SHIFT.match(code)
        if m:
            f["shifts"].append((int(m.group(2)), code.strip()))
    bad = [(name, reg, text) for name, f in sorted(funcs.items()) for reg, text in f["shifts"] if reg == f["max"] and reg % 8 == 7]
    n_shift = sum(len(f["shifts"]) for f in funcs.values())
    return len(funcs), n_shift, bad


def lint_assembly(path):
    funcs, cur, calls, kernels, kname = {}, None, {}, {}, None
    for ln, line in enumerate(open(path, errors="replace"), 1):
        m = re.match(r"^([A-Za-z_][\w$.]*):", line)
        if m and not line.startswith(".L"):
            cur = m.group(1)
            funcs.setdefault(cur, [])
            continue
        m = SHIFT.match(line)
        if m and cur:
            funcs[cur].append((ln, int(m.group(2)), line.strip().split(";")[0].strip()))
        m = re.match(r"^\s*\.set\s+(?:\.L)?([\w$.]+)\.num_vgpr,\s*(.*)$", line)
        if m:
            calls[m.group(1)] = set(re.findall(r"(?:\.L)?([A-Za-z_][\w$]*)\.num_vgpr", m.group(2)))
        m = re.match(r"^\s*\.amdhsa_kernel\s+(\S+)", line)
        if m:
            kname = m.group(1)
        m = re.match(r"^\s*\.amdhsa_next_free_vgpr\s+(\d+)", line)
        if m and kname:
            kernels[kname] = int(m.group(1))
    bad = []
    for k, nf in sorted(kernels.items()):
        last = (nf + 7) // 8 * 8 - 1
        seen, todo = set(), [k]
        while todo:
            f = todo.pop()
            if f not in seen:
                seen.add(f)
                todo.extend(calls.get(f, ()))
        for f in sorted(seen):
            for ln, reg, text in funcs.get(f, ()):
                if reg == last:
                    bad.append((f"{k} (next_free_vgpr {nf}) -> {f} line {ln}", reg, text))
    return len(kernels), sum(len(v) for v in funcs.values()), bad


def lint(path):
    """(summary line, findings): findings = [(where, register, instruction)]"""
    if path.endswith(".s"):
        n, n_shift, bad = lint_assembly(path)
        what = f"{n} kernels"
    else:
        n, n_shift, bad = lint_object(path)
        what = f"{n} functions"
    return f"{path}: {what}, {n_shift} 64-bit shifts by a vector register; {len(bad)} take the amount from the last register a wave may be allocated", bad


def main(path):
    summary, bad = lint(path)
    print(summary)
    for where, reg, text in bad[:40]:
        print(f"  {where}: {text}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
