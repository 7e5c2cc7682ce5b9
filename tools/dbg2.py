import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lime_amd
from oracle import oracle_py as O
ctx = lime_amd.Context()
rng = np.random.default_rng(1)
def run(name, lcp, da, eb, nr, ng):
    cl, nc, ml = O.detect(lcp, da, nr, 16)
    for e, tag in ((eb, "e1"), (None, "e0")):
        exp = O.score(da, e, cl, nr, ng, threads=4)
        try:
            sim, gnc, gml = ctx.fused(lcp, da, e, nr, ng, 16)
            s, _ = ctx.stats()
            bad = int((sim != exp).sum())
            print(f"{name} {tag}: clusters {nc} maxlen {ml} n_med {list(s.n_med)} n_big {s.n_big} diffcells {bad} sum_gpu {int(sim.astype(np.int64).sum())} sum_exp {int(exp.astype(np.int64).sum())}")
        except Exception as ex:
            print(name, tag, "ERR", ex)
def mk(n, nr, ng, p_run, p_read, syms=b"ACGT"):
    hi = rng.random(n) < p_run
    lcp = np.where(hi, 20, 3).astype(np.uint32); lcp[0] = 0
    da = np.where(rng.random(n) < p_read, rng.integers(0, nr, n), nr + rng.integers(0, ng, n)).astype(np.uint32)
    eb = rng.choice(np.frombuffer(syms, np.uint8), n).astype(np.uint8)
    return lcp, da, eb
run("small_nodup", *mk(50000, 20000, 20000, 0.3, 0.3), 20000, 20000)
run("small_dup", *mk(50000, 3, 3, 0.3, 0.5), 3, 3)
run("med_nodup", *mk(50000, 20000, 20000, 0.75, 0.4), 20000, 20000)
run("med_dup", *mk(50000, 4, 4, 0.75, 0.5), 4, 4)
run("iupac", *mk(50000, 20000, 20000, 0.5, 0.4, b"ACGTRYN\x00$a"), 20000, 20000)
run("one_window", *mk(400, 2000, 2000, 0.5, 0.4), 2000, 2000)
run("two_windows", *mk(900, 2000, 2000, 0.5, 0.4), 2000, 2000)
print("---- dense patterns")
def pat(n, period, nr, ng):
    lcp = np.full(n, 20, np.uint32); lcp[::period] = 0
    da = np.where(rng.random(n) < 0.5, rng.integers(0, nr, n), nr + rng.integers(0, ng, n)).astype(np.uint32)
    eb = rng.choice(np.frombuffer(b"ACGT", np.uint8), n).astype(np.uint8)
    return lcp, da, eb
for period in (2, 3, 4, 5, 8):
    for n in (512, 1024, 5000):
        run(f"period{period}_n{n}", *pat(n, period, 30000, 30000), 30000, 30000)
