"""Structural guarantees: the oracle is test infrastructure only; the product never imports it."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "julia-newtonraphsonhank_amd"


def test_product_never_touches_the_oracle():
    for f in list(PKG.rglob("*.py")) + list(PKG.rglob("*.hip")) + list(PKG.rglob("*.h")) + [ROOT / "hank_amd.py"]:
        text = f.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f
        assert "libhank_oracle" not in text and "hank_oracle" not in text, f


def test_no_reference_reads_at_runtime():
    for f in [ROOT / "bench.py", ROOT / "__graft_entry__.py"] + list(PKG.rglob("*.py")):
        if f.exists():
            text = f.read_text()
            for line in text.splitlines():
                if "/root/reference" in line:
                    # only the build step may look at the reference (to compile oracle/_ref)
                    assert f.name == "__graft_entry__.py", (f, line)


def test_oracle_header_says_test_infrastructure():
    head = (ROOT / "oracle" / "hank_oracle.c").read_text()[:1500]
    assert "TEST INFRASTRUCTURE ONLY" in head and "parity unpinned" in head
