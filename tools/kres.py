#!/usr/bin/env python3
"""tools/kres.py FILE.s [regex] -- per-kernel resources from the amdhsa metadata of hipcc's assembly output:
VGPRs, SGPRs, spills, LDS bytes, scratch; waves per SIMD the register count allows."""
import re, sys, json
txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
md = txt[txt.index("amdhsa.kernels:"):]
out = {}
for blk in re.split(r"\n  - \.agpr_count:", md)[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, None])[1]
    name = g("name")
    if pat and not pat.search(name): continue
    v = int(g("vgpr_count")); alloc = (v + 7) // 8 * 8
    out[name] = dict(vgpr=v, sgpr=int(g("sgpr_count")), sgpr_spill=int(g("sgpr_spill_count")), vgpr_spill=int(g("vgpr_spill_count")),
                     lds=int(g("group_segment_fixed_size")), scratch=int(g("private_segment_fixed_size")), waves_by_vgpr=min(8, 512 // alloc))
    print(f"{name[:60]:60s} vgpr {v:4d} sgpr {out[name]['sgpr']:4d} sspill {out[name]['sgpr_spill']:3d} vspill {out[name]['vgpr_spill']:3d} lds {out[name]['lds']:6d} scratch {out[name]['scratch']:4d} waves/simd {out[name]['waves_by_vgpr']}")
if len(sys.argv) > 3: json.dump(out, open(sys.argv[3], "w"), indent=1)
if "--gate" in sys.argv:
    bad = []
    for name, r in out.items():
        if "k_scanI" not in name: continue
        ebwt = "k_scanILi1" in name
        want_waves = 4 if (not ebwt or "k_scanILi1ELi0ELi1" in name or "k_scanILi1ELi0ELi2" in name) else 3     # EBWT=1: the record-emitting scans run 16 waves per CU, the compare-and-swap one 12
        if r["vgpr_spill"] or r["scratch"]: bad.append(f"{name}: VGPR spills / scratch")
        if r["waves_by_vgpr"] < want_waves: bad.append(f"{name}: {r['vgpr']} VGPRs allow {r['waves_by_vgpr']} waves per SIMD, planned {want_waves}")
        if r["lds"] > 160 * 1024: bad.append(f"{name}: {r['lds']} B of LDS do not fit a CU (one workgroup per CU)")
        if r["sgpr_spill"] > (12 if (ebwt or "ELi0ELi2E" in name or "ELi0ELi1E" in name) else 2): bad.append(f"{name}: {r['sgpr_spill']} SGPR spills")     # (SGPR spills go to VGPR lanes: harmless, DESIGN_HISTORY 4.10; the scans that write finished records -- <., 0, 2>, round 6 -- and the queue drains of three and more sub-regions (<., 0, 1>: a scalar pair of waiting counts) keep their line counters in scalar registers: 8 .. 11 spilled, outside the rounds)
    # k_part_lines: two workgroups per CU at the 477 bins of a 1 GB table (N = 1e10), 32- and 64-bit positions -- launch_part's rule
    # 2 x (part_lines_lds(nb, p64) + static + 512) <= 160 KB; round 5 lost the 64-bit instance to k_part (8.8 against 5.6 ms) over 512 bytes of static LDS
    for name, r in out.items():
        if "k_part_linesILb" not in name: continue
        p64 = "ILb1E" in name
        nb, tasks, cs = 477, 8192 // 16 + 16, 17
        dyn = nb * (16 + 4 * cs) + (tasks + 2 * nb) * 4 + (4 * nb if p64 else 0)
        if 2 * (dyn + r["lds"] + 512) > 160 * 1024: bad.append(f"{name}: {r['lds']} B static + {dyn} B of line buffers at {nb} bins do not fit a CU twice")
        if r["waves_by_vgpr"] < 4: bad.append(f"{name}: {r['vgpr']} VGPRs: two workgroups of 8 waves do not fit a CU")
    # the v_writelane of the scan's mask words sits in inline assembly with its own wait states (a VALU write of an SGPR / VCC
    # followed at once by a v_writelane reading it took the PREVIOUS value; the compiler's hazard recognizer does not see into the assembly):
    # every v_writelane_b32 must follow an s_nop or another v_writelane directly, and no DPP operation or v_readlane may follow one at once
    # (only the hand-written ones, between the compiler's ;;#ASMSTART / ;;#ASMEND marks: its own v_writelanes -- SGPR spills -- get their wait states from it)
    in_asm, prev, chk_next = False, "", False
    for l in txt.split("\n"):
        t = l.strip()
        if "#ASMSTART" in t: in_asm = True; continue
        if "#ASMEND" in t: in_asm = False; continue
        if not t or t.startswith((";", ".", "#")) or t.endswith(":"): continue
        if chk_next:
            if "_dpp" in t or t.startswith("v_readlane"): bad.append(f"`{t.split()[0]}` directly behind a hand-written v_writelane_b32")
            chk_next = False
        if in_asm and t.startswith("v_writelane_b32"):
            if not prev.startswith(("s_nop", "v_writelane_b32")): bad.append(f"hand-written v_writelane_b32 directly behind `{prev.split()[0] if prev else '?'}` (no wait states)")
            chk_next = True
        prev = t
    if bad:
        print("RESOURCE GATE FAILED:\n  " + "\n  ".join(sorted(set(bad)))); sys.exit(1)
    print("resource gate ok")
