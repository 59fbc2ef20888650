#!/usr/bin/env python3
"""Checks the gfx950 code of k_tile_rows (csrc/tile_rows.inc) for the one property its software
pipeline depends on and the compiler does not know about: between a pipelined load (inline-asm
`global_load_dword a<N>` into an accumulator register, or the returning ticket atomic) and the
first `s_waitcnt vmcnt(0)` behind it, no instruction may READ that register (a copy of a register
whose load is in flight copies garbage).  Straight-line scan per kernel: loads mark their
destination as pending, a vmcnt(0) clears all marks, any other mention of a pending register fails.
usage: check_tile_isa.py <file.s> ; exit code 1 on a violation."""
import re
import sys


def check(path):
    kernels, cur, name = {}, None, None
    for line in open(path):
        m = re.match(r"^(_ZN3bsp11k_tile_rows\w+):", line)
        if m:
            name, cur = m.group(1), []
            kernels[name] = cur
            continue
        if cur is not None:
            cur.append(line.rstrip())
            if "s_endpgm" in line:
                cur = None
    bad = 0
    for name, lines in kernels.items():
        pending = {}
        nloads = 0
        for no, line in enumerate(lines):
            t = line.strip()
            if not t or t.startswith(";") or t.startswith("."):
                continue
            if t.startswith("s_waitcnt") and "vmcnt(0)" in t:
                pending.clear()
                continue
            m = re.match(r"global_load_dword (a\d+),", t) or re.match(r"global_atomic_add (v\d+), .* sc0", t)
            if m:
                # the address operands of this very instruction must not be pending either
                rest = t[m.end():]
                for reg in pending:
                    if re.search(r"\b%s\b" % reg, rest):
                        print("%s: line %d reads pending %s: %s" % (name, no, reg, t))
                        bad += 1
                pending[m.group(1)] = no
                nloads += 1
                continue
            if t.endswith(":"):           # a label: control flow may merge here with loads in flight -- fine, marks stay
                continue
            for reg in pending:
                if re.search(r"\b%s\b" % reg, t) or re.search(r"\b[av]\[(\d+):(\d+)\]" , t) and any(
                        int(a) <= int(reg[1:]) <= int(b) and reg[0] == g for g, a, b in re.findall(r"\b([av])\[(\d+):(\d+)\]", t)):
                    print("%s: line %d touches pending %s (loaded at %d): %s" % (name, no, reg, pending[reg], t))
                    bad += 1
        if nloads == 0:
            print("%s: no pipelined loads found (pattern out of date?)" % name)
            bad += 1
    print("%d kernels checked, %d violations" % (len(kernels), bad))
    return 1 if (bad or not kernels) else 0


if __name__ == "__main__":
    sys.exit(check(sys.argv[1]))
