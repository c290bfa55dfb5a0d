#!/usr/bin/env python3
"""Static per-phase instruction counts of one k_trace variant, from the assembly hipcc leaves with
`--save-temps -gline-tables-only` (the .loc lines name the innermost inlined source line of every instruction).

    tools/isa_attrib.py <file.s> <mangled kernel name> [--blocks]

Prints, per source phase of rt_kernels.hip (line ranges below), the VALU / SALU / branch / memory / LDS instructions
the kernel's code holds, and with --blocks the same per basic block (label, instruction mix, source lines), which is
what the dynamic attribution in profiles/ multiplies with the counting build's rounds, items, trips and steps.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm", "s_call")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier", "s_setprio")):
        return "wait"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime", "s_dcache")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def phases_of(src):
    """Phase table from marker comments `// @phase NAME` in the source: a marker opens a phase that lasts to the next one."""
    marks = []
    for n, line in enumerate(open(src), 1):
        m = re.search(r"@phase\s+(\S+)", line)
        if m:
            marks.append((n, m.group(1)))
    return marks


def phase_at(marks, line):
    name = "other"
    for n, p in marks:
        if n <= line:
            name = p
        else:
            break
    return name


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    want_blocks = "--blocks" in sys.argv
    src = [a for a in sys.argv[3:] if a.endswith(".hip")]
    marks = phases_of(src[0]) if src else []
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    cur_line, cur_file = 0, 1
    blocks = []          # (label, Counter by class, Counter of source lines, list of ops)
    blk = ["entry", collections.Counter(), collections.Counter(), []]
    per_phase = collections.defaultdict(collections.Counter)
    fp = collections.Counter()
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end") or s.startswith(".size"):
            break
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur_file, cur_line = int(m.group(1)), int(m.group(2))
            continue
        if re.match(r"^\.Ltmp\d+:|^\.Lfunc_begin\d+:", s):
            continue
        if re.match(r"^\.?L?[A-Za-z0-9_$.]+:", s):
            blocks.append(blk)
            blk = [s.split(":")[0], collections.Counter(), collections.Counter(), []]
            continue
        if not s or s.startswith((".", ";", "//")):
            continue
        op = s.split()[0]
        c = classify(op)
        blk[1][c] += 1
        blk[2][cur_line] += 1
        blk[3].append(op)
        ph = phase_at(marks, cur_line) if marks else "all"
        per_phase[ph][c] += 1
        if c == "valu":
            kind = ("f64" if re.search(r"_f64", op) and not op.startswith("v_cvt") else
                    "cvt" if op.startswith("v_cvt") else
                    "pk_f32" if op.startswith("v_pk_") else
                    "f32" if re.search(r"_f32", op) and not op.startswith("v_cmp") else
                    "cmp" if op.startswith("v_cmp") else
                    "lane" if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane", "ds_bpermute")) else
                    "move" if op.startswith(("v_mov", "v_cndmask", "v_accvgpr")) else "int")
            per_phase[ph]["valu:" + kind] += 1
    blocks.append(blk)
    cols = ["valu", "salu", "branch", "smem", "vmem", "scratch", "lds", "wait"]
    kinds = ["f64", "f32", "pk_f32", "cvt", "cmp", "lane", "move", "int"]
    print("%-12s" % "phase" + "".join("%8s" % c for c in cols) + "   |" + "".join("%7s" % k for k in kinds))
    tot = collections.Counter()
    for ph in sorted(per_phase, key=lambda p: -per_phase[p]["valu"]):
        c = per_phase[ph]
        tot.update(c)
        print("%-12s" % ph + "".join("%8d" % c[k] for k in cols) + "   |" + "".join("%7d" % c["valu:" + k] for k in kinds))
    print("%-12s" % "total" + "".join("%8d" % tot[k] for k in cols) + "   |" + "".join("%7d" % tot["valu:" + k] for k in kinds))
    if want_blocks:
        print()
        for label, cls, srcl, ops in blocks:
            n = sum(cls.values())
            if n == 0:
                continue
            top = ",".join("%d:%d" % (ln, k) for ln, k in srcl.most_common(4))
            ph = collections.Counter()
            for ln, k in srcl.items():
                ph[phase_at(marks, ln) if marks else "all"] += k
            print("%-14s n=%4d valu=%4d salu=%4d br=%2d vmem=%2d scr=%2d lds=%2d  %-24s lines %s" % (
                label, n, cls["valu"], cls["salu"], cls["branch"], cls["vmem"], cls["scratch"], cls["lds"],
                "/".join("%s:%d" % (p, k) for p, k in ph.most_common(2)), top))


if __name__ == "__main__":
    main()
