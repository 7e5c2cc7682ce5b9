#!/bin/bash
# sweep: ring vs direct, chunk sizes, at one collection size
export C_N=${C_N:-1000000000}
python3 - <<'PY'
import json, os, subprocess, sys, tempfile, time
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
import torch, lime_amd
n = int(float(os.environ["C_N"])); nr, ng, alpha = 1_000_000, 500, 16
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    base = os.path.join(td, "S.fasta")
    ctx = lime_amd.Context(0); dev = torch.device("cuda:0")
    lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp)
    ctx.synth_dev(42, 0, n, nr, ng, alpha, 0, lcp, da, None); torch.cuda.synchronize()
    lcp.cpu().numpy().tofile(base + ".lcp"); da.cpu().numpy().tofile(base + ".da")
    del lcp, da; ctx.close(); torch.cuda.empty_cache()
    for chunk in (4 << 20, 16 << 20, 64 << 20):
        for staging in ("ring", "direct"):
            for thr in (8,):
                env = dict(os.environ, LIME_DETECT_CHUNK=str(chunk))
                if staging == "direct": env["LIME_NO_STAGING"] = "1"
                best = 1e9
                for rep in range(2):
                    t0 = time.perf_counter()
                    subprocess.run([f"{ROOT}/lime_amd/bin/ClusterLCP", base, str(nr), str(ng), str(alpha), str(thr)], check=True, capture_output=True, env=env, cwd=td)
                    best = min(best, time.perf_counter() - t0)
                print(f"chunk {chunk>>20}Mi {staging} t{thr}: {best:.3f} s  {8*n/best/1e9:.2f} GB/s", flush=True)
PY
