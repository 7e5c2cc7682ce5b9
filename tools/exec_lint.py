#!/usr/bin/env python3
"""tools/exec_lint.py FILE.s [--allow tools/exec_lint_allow.txt] [--list]

Cross-lane operations under a partial EXEC mask, found in hipcc's assembly output (compile with -gline-tables-only so that the
instructions carry `.loc` source lines).

Why: ds_bpermute / ds_permute / ds_swizzle / DPP modifiers / v_permlane read the registers of OTHER lanes, and a source lane that the current
EXEC mask has switched off delivers 0 (or leaves the old value): `cond ? __shfl(x, l) : 0` compiles to a branch on `cond`, and every lane that
reads from a lane outside `cond` silently gets 0.  That has been three bugs in five rounds (round 1: the rows routine's shuffle under a select;
round 5: k_apply_tiles<true>'s index entries -- 0.1 % of the update records of every bin of 257 .. 511 tiles were dropped for two rounds and the
N = 1e10 bench numbers of rounds 3 .. 4 were timed on a kernel that skipped work).  The rule of the code base is: cross-lane operations are
executed by ALL lanes, selects come after.  This script checks it mechanically.

How: per kernel, basic blocks from labels and branches; a forward data flow over them whose state is the STACK of open EXEC restrictions:
  s_*_saveexec_b64 D, ..       push D            (if / else / loop entry: SI_IF, SI_ELSE)
  s_andn2_b64 exec, exec, X    push X            (lanes that left a loop)
  s_and_b64 exec, exec, X      push "&"          (narrowed for good inside the enclosing region)
  s_or_b64 exec, exec, X       pop down to and including X   (SI_END_CF: the region's lanes are back)
  s_mov_b64 exec, -1           clear; s_mov_b64 exec, X: pop to X if X is open, else push
  s_mov_b64 D, exec            push the marker ~D (no restriction): the `s_or_b64 exec, exec, D` that ends such a region pops down to it
  s_xor_b64 S, exec, X         (S not exec) the open region X goes by the name S from here (its else-half opens with `s_andn2_saveexec_b64 D, S`)
  s_xor_b64 exec, ..           unchanged (the other half of an if / else: still partial)
joins take the longer stack (a loop header inherits the back edge's restrictions).  A cross-lane operation in a block position whose stack is not
empty is reported as (kernel, source file:line, mnemonic).  v_readlane / v_readfirstlane / v_writelane are NOT reported: they ignore EXEC and
move the register as it is (the compiler's SGPR spills use them everywhere).

The compiler cannot know that a condition is the same for all 64 lanes (a wave's window number, `while (__ballot(..))`), so code under such a
condition is reported too: the allow list (tools/exec_lint_allow.txt) holds every site that has been LOOKED AT and is meant to run that way --
`mangled-kernel-name  file:line  count  reason`, count = how many reported instructions that (kernel, line) pair has.  An inlined helper's line
(wave_incl_scan, __shfl) stands for all its call sites in a kernel, so a NEW call site under a condition raises the count and fails the gate: a
ratchet.  `--update` rewrites the list from the code as it stands (counts refreshed, reasons kept, new sites marked UNREVIEWED, which fails the
gate until a reason has been written).  Exit code 1 on any site not covered.
"""
import re
import sys
from collections import defaultdict

CROSS = re.compile(r"^(ds_bpermute_b32|ds_permute_b32|ds_swizzle_b32|v_permlane\w*)\b")
DPP_MOD = re.compile(r"\b(quad_perm:|row_shl:|row_shr:|row_ror:|wave_shl:|wave_shr:|wave_rol:|wave_ror:|row_mirror|row_half_mirror|row_bcast:|row_newbcast:|row_share:|row_xmask:)")
BRANCH = re.compile(r"^(s_cbranch_\w+|s_branch)\s+(\S+)")
LABEL = re.compile(r"^([.\w$]+):")


def demangle_hint(sym):
    """k_scan<1,0,1> out of _ZN4lime6k_scanILi1ELi0ELi1EEEvNS_8ScanArgsE (good enough to read; the regexes of the allow list match the mangled name)"""
    m = re.match(r"_ZN4lime(\d+)", sym)
    if not m:
        return sym
    n = int(m.group(1)); start = m.end()
    name = sym[start:start + n]
    targs = re.findall(r"L[ib](\d+)E", sym[start + n:].split("EEv")[0]) if sym[start + n:].startswith("I") else []
    return name + ("<" + ",".join(targs) + ">" if targs else "")


def parse_functions(txt):
    """-> {symbol: [(kind, payload, loc)]} with kind in label / insn"""
    files, funcs, cur, loc = {}, {}, None, None
    for raw in txt.split("\n"):
        t = raw.strip()
        if not t:
            continue
        m = re.match(r"\.file\s+(\d+)\s+(?:\"([^\"]*)\"\s+)?\"([^\"]*)\"", t)
        if m:
            files[int(m.group(1))] = m.group(3).split("/")[-1]
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            loc = (files.get(int(m.group(1)), "file%s" % m.group(1)), int(m.group(2)))
            continue
        m = re.match(r"\.type\s+(\S+),@function", t)
        if m:
            cur = m.group(1); funcs[cur] = []; loc = None
            continue
        if t.startswith(".size") or t.startswith(".end_amdhsa_kernel") or t.startswith(".section"):
            if t.startswith(".size") and cur and t.split()[1].rstrip(",") == cur:
                cur = None
            continue
        if cur is None or t.startswith((";", "#", "//")):
            continue
        m = LABEL.match(t)
        if m and not t.startswith(".L") or (m and t.startswith(".LBB")):
            if m.group(1) != cur:
                funcs[cur].append(("label", m.group(1), loc))
            continue
        if t.startswith("."):
            continue
        funcs[cur].append(("insn", t.split(";")[0].strip(), loc))
    return funcs


def blocks_of(items):
    """basic blocks: list of dicts {labels, insns [(text, loc)], succ [block indices]}"""
    blocks, cur = [], {"labels": [], "insns": []}
    for kind, payload, loc in items:
        if kind == "label":
            if cur["insns"] or cur["labels"]:
                if cur["insns"]:
                    blocks.append(cur); cur = {"labels": [], "insns": []}
            cur["labels"].append(payload)
        else:
            cur["insns"].append((payload, loc))
            if BRANCH.match(payload) or payload.startswith(("s_endpgm", "s_setpc_b64")):
                blocks.append(cur); cur = {"labels": [], "insns": []}
    if cur["insns"] or cur["labels"]:
        blocks.append(cur)
    at = {}
    for i, b in enumerate(blocks):
        for l in b["labels"]:
            at[l] = i
    for i, b in enumerate(blocks):
        succ = []
        last = b["insns"][-1][0] if b["insns"] else ""
        m = BRANCH.match(last)
        if m:
            if m.group(2) in at:
                succ.append(at[m.group(2)])
            if m.group(1) != "s_branch" and i + 1 < len(blocks):
                succ.append(i + 1)
        elif not last.startswith(("s_endpgm", "s_setpc_b64")) and i + 1 < len(blocks):
            succ.append(i + 1)
        b["succ"] = succ
    return blocks


def step(stack, insn):
    """EXEC restriction stack after one instruction"""
    if "exec" not in insn:
        return stack
    m = re.match(r"s_\w+_saveexec_b64\s+(\S+?),\s*(\S+)", insn)
    if m:
        d, src = m.group(1), m.group(2)
        if d in stack:
            return stack                                                    # (a loop's body re-entered: the same save, not a deeper one)
        if src in stack:                                                    # the switch to an else-half whose mask was renamed below: the same region
            return tuple(d if x == src else x for x in stack)
        return stack + (d,)
    m = re.match(r"s_xor_b64\s+(\S+?),\s*exec,\s*(\S+)$", insn)
    if m and m.group(1) != "exec":
        # S = exec ^ X right after `s_and_saveexec X`: S holds the if's OTHER lanes; when S is another register than X, the else-half opens with
        # `s_andn2_saveexec D, S` and the region closes with `s_or exec, exec, D` -- X is never mentioned again, so the open region goes by S from here
        return tuple(m.group(1) if x == m.group(2) else x for x in stack)
    m = re.match(r"s_mov_b64\s+(\S+?),\s*exec$", insn)
    if m:
        # a plain copy of EXEC (`s_mov D, exec; s_mov exec, X; ..; s_or exec, exec, D`: a region run by a precomputed lane set): no restriction by
        # itself -- a marker "~D" that the closing s_or finds, so that what was pushed in between is popped with it
        return stack if ("~" + m.group(1)) in stack else stack + ("~" + m.group(1),)
    m = re.match(r"(s_\w+_b64)\s+exec,\s*(\S+?)(?:,\s*(\S+))?$", insn)
    if not m:
        return stack
    op, a, b = m.group(1), m.group(2), m.group(3)
    other = b if a == "exec" else a
    if op == "s_or_b64":
        for key in (other, "~" + other):
            if key in stack:
                return stack[:len(stack) - 1 - stack[::-1].index(key)]
        return stack
    if op == "s_andn2_b64":
        return stack if other in stack else stack + (other,)
    if op == "s_and_b64":
        return stack if stack and stack[-1] == "&" else stack + ("&",)      # (once: around a loop the same narrowing comes again and again)
    if op == "s_mov_b64":
        if a == "-1":
            return ()
        if a in stack:
            return stack[:len(stack) - 1 - stack[::-1].index(a)]
        return stack if ("=" + a) in stack else stack + ("=" + a,)
    return stack            # s_xor_b64 exec and anything else: as restricted as before


def lint_function(items):
    blocks = blocks_of(items)
    if not blocks:
        return []
    entry = [None] * len(blocks)
    entry[0] = ()
    work = [0]
    rounds = 0
    while work and rounds < 200000:
        rounds += 1
        i = work.pop()
        st = entry[i]
        for insn, _ in blocks[i]["insns"]:
            st = step(st, insn)
        for j in blocks[i]["succ"]:
            if entry[j] is None or len(st) > len(entry[j]):
                entry[j] = st
                work.append(j)
    found = []
    for i, b in enumerate(blocks):
        st = entry[i]
        if st is None:
            continue            # unreachable in this model
        for insn, loc in b["insns"]:
            if any(not x.startswith("~") for x in st) and (CROSS.match(insn) or DPP_MOD.search(insn)):
                found.append((loc, insn.split()[0], len(st)))
            st = step(st, insn)
    return found


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    allow_path = None
    if "--allow" in sys.argv:
        allow_path = sys.argv[sys.argv.index("--allow") + 1]
        args = [a for a in args if a != allow_path]
    txt = open(args[0]).read()
    allow, head = [], []
    if allow_path:
        try:
            for ln in open(allow_path):
                t = ln.strip()
                if not t or t.startswith("#"):
                    if not allow:
                        head.append(ln.rstrip("\n"))
                    continue
                sym, where, count, reason = t.split(None, 3)
                allow.append((sym, where, int(count), reason))
        except FileNotFoundError:
            pass
    funcs = parse_functions(txt)
    sites = defaultdict(int)
    for sym, items in funcs.items():
        for loc, op, depth in lint_function(items):
            where = "%s:%d" % loc if loc else "?:0"
            sites[(sym, where, op)] += 1
    per_line, ops_of = defaultdict(int), defaultdict(set)
    for (sym, where, op), n in sites.items():
        per_line[(sym, where)] += n
        ops_of[(sym, where)].add(op)
    if "--update" in sys.argv:
        # the list as the code stands now: counts refreshed, reasons kept where (kernel, line) was listed before, new sites marked UNREVIEWED (the gate
        # fails on them until somebody has looked at the site and written down why it may run with lanes switched off)
        known = {(a[0], a[1]): a[3] for a in allow}
        with open(allow_path, "w") as f:
            f.write("\n".join(head) + ("\n" if head else ""))
            for (sym, where), n in sorted(per_line.items()):
                f.write(f"{sym}  {where}  {n}  {known.get((sym, where), 'UNREVIEWED ' + ', '.join(sorted(ops_of[(sym, where)])))}\n")
        print(f"{allow_path}: {len(per_line)} site(s) written")
        return
    bad = []
    cover = {(a[0], a[1]): a for a in allow}
    for (sym, where), n in sorted(per_line.items()):
        a = cover.get((sym, where))
        ops = ", ".join(sorted(ops_of[(sym, where)]))
        if "--list" in sys.argv:
            print(f"{demangle_hint(sym):40s} {where:32s} {n:3d}  {ops}   [{sym}]")
        if a is None or n > a[2]:
            bad.append(f"{demangle_hint(sym)}: {n} cross-lane operation(s) ({ops}) under a partial EXEC mask at {where}" + (f" (the allow list covers {a[2]})" if a else "") + f"   [{sym}]")
        elif a[3].startswith("UNREVIEWED"):
            bad.append(f"{demangle_hint(sym)}: the site {where} is in the allow list without a reason (UNREVIEWED)   [{sym}]")
    if bad:
        print("EXEC LINT FAILED -- cross-lane operations that run with lanes switched off (execute them with all lanes and select afterwards; or, if it is\n"
              "meant, list the site in tools/exec_lint_allow.txt with the reason -- `tools/exec_lint.py FILE.s --allow tools/exec_lint_allow.txt --update` refreshes the counts):\n  " + "\n  ".join(bad))
        sys.exit(1)
    stale = [f"{a[0]} {a[1]}" for a in allow if per_line.get((a[0], a[1]), 0) < a[2]]
    print(f"exec lint ok ({len(funcs)} functions, {sum(per_line.values())} listed cross-lane operation(s) under a possibly partial EXEC mask" + (f"; {len(stale)} entries of the allow list are larger than needed" if stale else "") + ")")


if __name__ == "__main__":
    main()
