import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The library reads its tuning / test knobs from the environment only in a process that says it is a test (lime_init, include/lime_hip.h):
# every LIME_<KNOB> a test sets with monkeypatch.setenv before it creates a Context -- and the drop-in programs the tests start -- depend on it.
# Under it the library also fills every block it recycles from its process-wide cache with 0xA5: nothing may rely on what a fresh allocation holds.
os.environ.setdefault("LIME_TEST_HOOKS", "1")

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                      if not os.path.basename(p).startswith(("classify_", "example_")))   # those: tests/test_classify_cpu.py, tests/test_example_gpu.py


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def pytest_sessionstart(session):
    """Built artefacts are kept out of git: if the library, a drop-in program or the oracle is missing in
    this checkout, build them once (make; hipcc cross-compiles without a GPU) before any test imports them."""
    need = [os.path.join(ROOT, "lime_amd", "liblime_hip.so"), os.path.join(ROOT, "oracle", "liblime_oracle.so")]
    need += [os.path.join(ROOT, "lime_amd", "bin", b) for b in ("ClusterLCP", "ClusterBWT_DA", "Classify", "EGSAtoBCR")]
    if all(os.path.exists(p) for p in need):
        return
    import subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "lime_amd", "csrc"), "-s"], check=True)
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {k: z[k] for k in z.files}
    g["n_reads"], g["n_refs"], g["alpha"], g["read_len"] = (int(x) for x in g["params"])
    g["beta"] = float(g["beta"])
    return g


@pytest.fixture(params=GOLDEN_CASES)
def golden(request):
    g = load_golden(request.param)
    g["name"] = request.param
    return g


@pytest.fixture(scope="module", autouse=True)
def _trim_block_cache_per_module(request):
    """contexts leave their large device blocks with the library (lime_trim_cache): hand them back between test modules, so that the
    full-size tests of a later module find the whole device; within a module the recycling (poisoned under LIME_TEST_HOOKS) is what is tested"""
    yield
    try:
        import torch
        if torch.cuda.is_available():
            import lime_amd
            lime_amd.trim_cache()
    except Exception:
        pass
