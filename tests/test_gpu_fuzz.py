"""A few seconds of tools/fuzz_spmm.py (random shapes, row-length distributions, panel counts, layouts, dtypes and
kernels against a dense numpy product) and tools/fuzz_structure.py (merges, gather, slices, sort, SpMV, CSR op vector
against the oracle, bit for bit) inside the GPU suite; run the tools themselves for longer sweeps."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_spmm_fuzz_short(gpu, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_spmm.py"), "6", str(seed)], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "fuzz OK" in r.stdout


@pytest.mark.gpu
def test_structure_fuzz_short(gpu):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_structure.py"), "6", "21"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "fuzz OK" in r.stdout


@pytest.mark.gpu
def test_r_shim_fuzz_short(gpu):
    """tools/fuzz_r_shim.py: edge shapes through the `.Call` shim by routine name on the mock R runtime (gctorture on)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_r_shim.py"), "6", "31"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "shim fuzz OK" in r.stdout


@pytest.mark.gpu
def test_sharded_export_fuzz_short(gpu):
    """tools/fuzz_sharded.py: random geometry of the sharded export (this GPU listed 1 .. 8 times) against the unsharded call"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_sharded.py"), "8", "41"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "sharded fuzz OK" in r.stdout
