"""The measurement tooling must not rot: every script and example parses, and the instrumented build of the library
(`make stamp`: in-kernel stamps of the persistent sweeps, never loaded by the product) still compiles for gfx950."""
import py_compile
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_scripts_and_examples_parse(tmp_path):
    files = sorted((ROOT / "scripts").glob("*.py")) + sorted((ROOT / "examples").glob("*.py")) + [ROOT / "bench.py", ROOT / "__graft_entry__.py"]
    assert len(files) > 8
    for f in files:
        py_compile.compile(str(f), cfile=str(tmp_path / (f.name + "c")), doraise=True)


@pytest.mark.skipif(shutil.which("hipcc") is None and not Path("/opt/rocm/bin/hipcc").exists(), reason="hipcc not available")
def test_instrumented_build_compiles(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = ROOT / "julia-newtonraphsonhank_amd" / "csrc" / "hank_hip.hip"
    out = tmp_path / "stamp.o"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-DHANK_XSTAMP", "-c", "--cuda-device-only",
                        "-o", str(out), str(src)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.stat().st_size > 100_000
