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


def test_every_family_has_its_oracle_and_its_gpu_parity_test():
    """a value-function family is an enum value of the library (the reference resolves any function by name,
    ModelParser.jl:338-342, :404-413); every HANK_VF_* must come with its oracle hook and a GPU parity test that names it."""
    header = (ROOT / "include" / "hank_hip.h").read_text()
    ids = dict((name, int(val)) for name, val in re.findall(r"\b(HANK_VF_[A-Z_0-9]+)\s*=\s*(\d+)", header))
    assert len(ids) >= 2 and sorted(ids.values()) == list(range(len(ids)))
    # id -> (oracle entry point of the family's ValueFunction, the GPU test that pins the family against it, the host plugin module)
    families = {
        "HANK_VF_KRUSELL_SMITH": ("orc_value_function", "tests/test_gpu_steps.py::test_backward_step_matches_value_function", "KrusellSmith.py"),
        "HANK_VF_ONE_ASSET_HANK": ("orc_value_function_tr", "tests/test_gpu_hank.py::test_household_block_matches_oracle", "OneAssetHANK.py"),
    }
    assert set(ids) == set(families), f"a family without its row here (INTEGRATION.md 5a): {set(ids) ^ set(families)}"
    oracle_c = (ROOT / "oracle" / "hank_oracle.c").read_text()
    julia = (ROOT / "julia" / "HankHIP.jl").read_text()
    for name, (orc_fn, test, plugin) in families.items():
        assert re.search(r"\bFN\(" + orc_fn + r"\)", oracle_c), (name, orc_fn)
        f, fn = test.split("::")
        assert re.search(r"^def " + fn + r"\(", (ROOT / f).read_text(), re.M), (name, test)
        plug = (ROOT / "julia-newtonraphsonhank_amd" / plugin).read_text()
        assert re.search(r"value_fn_id\s*=\s*(HANK_VF_[A-Z_]+|\d+)", plug), (name, plugin)
        assert re.search(r"=>\s*\(" + str(ids[name]) + r",", julia), (name, "julia/HankHIP.jl:_FAMILIES")
