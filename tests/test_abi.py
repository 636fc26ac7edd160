"""The C-ABI library loads and exports every symbol include/hank_hip.h declares; with no GPU the
product path fails loudly (no CPU fallback)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "hank_hip.h").read_text()
    return sorted(set(re.findall(r"\b(hank_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree(hank):
    from hank_amd import hip
    assert _declared_symbols() == sorted(hip.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(hank):
    from hank_amd import hip
    lib = ctypes.CDLL(str(hip.library_path()))
    for name in _declared_symbols():
        assert getattr(lib, name) is not None, name


def test_every_entry_point_cites_the_reference():
    text = (ROOT / "include" / "hank_hip.h").read_text()
    for cite in ("BackwardIteration.jl", "ForwardIteration.jl", "KrusellSmith.jl", "GeneralStructures.jl", "NewtonRaphson.jl"):
        assert cite in text


def test_no_device_fails_loudly(hank):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import numpy as np
    with pytest.raises(hank.HankHIPError):
        hank.HouseholdBlock(np.linspace(0, 1, 5), np.ones(2), np.full((2, 2), 0.5), 0.98, 2.0, 0.0, 10)
