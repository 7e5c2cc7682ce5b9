import os, subprocess, sys, tempfile
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
import torch, lime_amd
n = int(float(os.environ.get("C_N", 1e9))); nr, ng, alpha = 1_000_000, 500, 16
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    base = os.path.join(td, "S.fasta")
    ctx = lime_amd.Context(0); dev = torch.device("cuda:0")
    lcp = torch.empty(n, dtype=torch.int32, device=dev); da = torch.empty_like(lcp); eb = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.synth_dev(42, 0, n, nr, ng, alpha, 0, lcp, da, eb); torch.cuda.synchronize()
    lcp.cpu().numpy().tofile(base + ".lcp"); da.cpu().numpy().tofile(base + ".da"); eb.cpu().numpy().tofile(base + ".ebwt")
    del lcp, da, eb; ctx.close(); torch.cuda.empty_cache()
    r = subprocess.run([f"{ROOT}/lime_amd/bin/ClusterLCP", base, str(nr), str(ng), str(alpha), "8"], capture_output=True, cwd=td)
    print("LCP rc", r.returncode)
    for env_extra in ({"LIME_NO_STAGING": "1", "LIME_SCORE_CHUNK": "4194304"}, {"LIME_SCORE_CHUNK": "67108864"}, {"LIME_SCORE_CHUNK": "4194304", "LIME_IO_THREADS": "1"}, {}):
        r = subprocess.run([f"{ROOT}/lime_amd/bin/ClusterBWT_DA", base, "100", "0.25", "8"], capture_output=True, cwd=td, env=dict(os.environ, **env_extra))
        print(env_extra, "BWT rc", r.returncode, r.stderr.decode()[-200:].replace("\n", " | "), flush=True)
