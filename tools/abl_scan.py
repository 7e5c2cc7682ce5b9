#!/usr/bin/env python3
"""tools/abl_scan.py WORKLOAD[,WORKLOAD..] [cuts] -- where the scan's time goes: a library built with -DLIME_ABLATE_BUILD (select it with LIME_LIB) cuts
k_scan after phase LIME_ABLATE=k (RESULTS INVALID): 1 = loads + staging, 3 = + chunk acceptance, 4 = + cluster lists, 12 = + the rounds of a dense
window's 2-symbol clusters, 10 = + the other clusters' lengths, 11 = + 2-4-symbol scoring, 8 = everything but the record drains, 0 = everything.
  make -C lime_amd/csrc EXTRA=-DLIME_ABLATE_BUILD OUT=/tmp/abl ...   (or hipcc by hand into variants/lib_abl.so)
  LIME_LIB=$PWD/variants/lib_abl.so python3 tools/abl_scan.py text_spread,c3"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import lime_amd  # noqa: E402
from lime_amd import dist as ldist  # noqa: E402

names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["text_spread"]
cuts = sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "3", "4", "12", "10", "11", "8", "0"]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for name in names:
    wl = bench.WORKLOADS[name]
    out = []
    for k in cuts:
        os.environ["LIME_ABLATE"] = k
        r = bench.run_pass_series(torch, lime_amd, ldist, wl, wl["n"], 5 if wl["n"] >= 10_000_000_000 else 12, 2, 1, 0, dev, None, overlap=False,
                                  options={"update_path": os.environ.get("ABL_PATH", "bin"), "no_probe": "1"})      # (ABL_PATH=cas: the compare-and-swap kernels)
        out.append((k, round(r["parts"]["scan"] * 1e3, 1)))
        del r
    print(name, " ".join(f"cut {k}: {us} us" for k, us in out), flush=True)
