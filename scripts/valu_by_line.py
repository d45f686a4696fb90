"""Static account of a kernel's vector instructions by SOURCE LINE (round 6): device assembly compiled with -gline-tables-only carries `.loc file line`
directives; every v_* instruction of the chosen function is charged to the innermost inlined-at source line it carries.  Static, not dynamic - but the
step function is mostly straight-line code executed once per call, so the table shows where the vector instructions of a step live.
usage: python scripts/valu_by_line.py <file.s> <function substring> [top N] [file filter, default rmj_]"""
import collections
import re
import sys


def main(path, func, top=60, ffilter="rmj_"):
    files, cur, loc, on = {}, None, None, False
    by_line, by_func_line = collections.Counter(), collections.Counter()
    total = 0
    for line in open(path, errors="replace"):
        m = re.match(r'^\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', line)
        if m:
            files[int(m.group(1))] = m.group(2)
            continue
        m = re.match(r"^([A-Za-z_][\w$.]*):", line)
        if m and not line.startswith(".L"):
            on = func in m.group(1)
            continue
        if not on:
            continue
        m = re.match(r"^\s*\.loc\s+(\d+)\s+(\d+)", line)
        if m:
            loc = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        m = re.match(r"^\s+(v_\w+)", line)
        if m and loc:
            total += 1
            if ffilter in loc[0]:
                by_line[loc] += 1
            else:
                by_line[("(" + loc[0] + ")", 0)] += 1
    print(f"{path}: function *{func}*: {total} vector instructions (static)")
    for (f, ln), c in by_line.most_common(top):
        print(f"  {c:5d}  {f}:{ln}")
    return by_line


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 60, sys.argv[4] if len(sys.argv) > 4 else "rmj_")
