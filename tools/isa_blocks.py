#!/usr/bin/env python3
"""tools/isa_blocks.py FILE.s KERNEL_SUBSTR -- basic blocks of one kernel in hipcc's assembly output with their
instruction mix (VALU / SALU / LDS / VMEM / other), loop depth and the LIMEMARK phase comments (-DLIME_MARK).
A reading aid for counting the vector instructions a window of the scan executes."""
import re, sys
src, want = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN4lime\w+:", l) and want in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks = []; cur = {"name": "entry", "v": 0, "s": 0, "lds": 0, "vm": 0, "o": 0, "depth": 0, "marks": [], "line": start, "br": []}
def cls(op):
    if op.startswith("v_"): return "v"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "vm"
    if op.startswith("s_"): return "s"
    return "o"
for i in range(start + 1, end):
    l = lines[i]
    m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
    if m:
        blocks.append(cur)
        d = re.search(r"Depth=(\d+)", m.group(2))
        cur = {"name": m.group(1), "v": 0, "s": 0, "lds": 0, "vm": 0, "o": 0, "depth": int(d.group(1)) if d else 0, "marks": [], "line": i, "br": []}
        continue
    t = l.strip()
    if t.startswith("; LIMEMARK"): cur["marks"].append(t.split()[-1]); continue
    if not t or t.startswith((";", ".", "#")): continue
    op = t.split()[0]
    if op in ("s_nop", "s_waitcnt"): cur["o"] += 1; continue
    cur[cls(op)] += 1
    if op.startswith(("s_cbranch", "s_branch")): cur["br"].append(t.split()[-1])
blocks.append(cur)
tot = {k: sum(b[k] for b in blocks) for k in ("v", "s", "lds", "vm", "o")}
print("kernel", want, "blocks", len(blocks), "totals", tot)
for b in blocks:
    print(f"{b['name']:>12} L{b['line']:<6} d{b['depth']} V{b['v']:<4} S{b['s']:<4} LDS{b['lds']:<3} VM{b['vm']:<3} marks={','.join(b['marks'])} -> {' '.join(b['br'])}")
