#!/usr/bin/env python3
"""Renames the vector registers of one kernel in a gfx950 assembly listing so that the source operands of its inner
loop's instructions come from different register banks.

A vector instruction of a wave64 reads its source VGPRs from four banks (register number mod 4), one operand per bank
per two cycles: an instruction whose distinct source registers share a bank issues in ~4 cycles instead of ~2.5
(tools/microbench/vgpr_bank.hip).  The register allocator does not know about banks.  A renaming that is applied to
every occurrence of a register inside the kernel is a permutation of names and changes nothing but the banks, so this
tool searches the permutations of the registers that may be renamed freely - those that never appear in a register
tuple (64-bit values, multi-dword loads keep their alignment and adjacency) and are not live at kernel entry - for one
that minimises the conflicts, instructions inside loops weighted by their nesting.

  vgpr_banks.py in.s out.s --kernel SYMBOL [--kernel SYMBOL ...] [--report]

Deterministic: same input, same output (fixed seed)."""
import argparse
import random
import re
import sys

REG = re.compile(r"(?<![\w.\[:])v(\d+)\b")
TUPLE = re.compile(r"\bv\[(\d+):(\d+)\]")


def body_range(lines, sym):
    start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return start, end


def parse(lines, start, end):
    """[(line index, weight, [source registers], is VALU)] and the set of pinned registers."""
    pinned = {0, 1, 2}  # work-item ids arrive in v0 (packed) - v1, v2 kept as well for kernels built without packing
    insts = []
    depth_weight = 1
    for i in range(start, end):
        raw = lines[i]
        code = raw.split(";")[0].strip()
        if "Loop Header: Depth=" in raw or "in Loop: Header=" in raw:
            depth_weight = 8
        elif re.match(r"^\.LBB\d+_\d+:", raw) and "Loop" not in raw:
            depth_weight = 1
        for m in TUPLE.finditer(code):
            pinned.update(range(int(m.group(1)), int(m.group(2)) + 1))
        if not code or code.endswith(":") or code.startswith("."):
            continue
        parts = code.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        if not op.startswith("v_"):
            continue
        srcs = ops[1:]
        if "fmac" in op or "_mac_" in op:
            srcs = srcs + ops[:1]
        regs = []
        for o in srcs:
            for m in REG.finditer(o):
                r = int(m.group(1))
                if r not in regs:
                    regs.append(r)
        insts.append((i, depth_weight, regs))
    return insts, pinned


def cost(insts, perm):
    c = 0
    for _, w, regs in insts:
        banks = [perm.get(r, r) & 3 for r in regs]
        c += w * (len(banks) - len(set(banks)))
    return c


def search(insts, free, seed=1):
    rnd = random.Random(seed)
    free = sorted(free)
    best = {r: r for r in free}
    best_c = cost(insts, best)
    for restart in range(6):
        perm = dict(best)
        cur = cost(insts, perm)
        temp = 3.0
        for it in range(6000):
            a, b = rnd.sample(free, 2)
            perm[a], perm[b] = perm[b], perm[a]
            c = cost(insts, perm)
            if c <= cur or rnd.random() < pow(2.718281828, (cur - c) / max(temp, 1e-3)):
                cur = c
                if c < best_c:
                    best_c, best = c, dict(perm)
            else:
                perm[a], perm[b] = perm[b], perm[a]
            temp *= 0.999
    return best, best_c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--kernel", action="append", required=True)
    ap.add_argument("--report", action="store_true")
    a = ap.parse_args()
    lines = open(a.src).read().split("\n")
    for sym in a.kernel:
        start, end = body_range(lines, sym)
        insts, pinned = parse(lines, start, end)
        used = set()
        for i in range(start, end):
            code = lines[i].split(";")[0]
            used.update(int(m.group(1)) for m in REG.finditer(code))
        free = used - pinned
        before = cost(insts, {})
        perm, after = search(insts, free)
        for i in range(start, end):
            head, sep, tail = lines[i].partition(";")
            lines[i] = REG.sub(lambda m: "v%d" % perm.get(int(m.group(1)), int(m.group(1))), head) + sep + tail
        if a.report:
            loop = [x for x in insts if x[1] > 1]
            print("%s: %d VGPRs renamed freely of %d used; weighted bank conflicts %d -> %d; loop instructions with a conflict %d -> %d of %d"
                  % (sym, len(free), len(used), before, after, sum(1 for _, _, r in loop if len(set(x & 3 for x in r)) < len(r)),
                     sum(1 for _, _, r in loop if len(set(perm.get(x, x) & 3 for x in r)) < len(r)), len(loop)), file=sys.stderr)
    open(a.dst, "w").write("\n".join(lines))


if __name__ == "__main__":
    main()
