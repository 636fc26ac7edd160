"""Import alias: the package directory is `julia-newtonraphsonhank_amd/` (not a valid Python
identifier), so `import hank_amd` loads it under this name."""
import importlib.util
import sys
from pathlib import Path

_pkg_dir = Path(__file__).resolve().parent / "julia-newtonraphsonhank_amd"
_spec = importlib.util.spec_from_file_location("hank_amd", _pkg_dir / "__init__.py",
                                               submodule_search_locations=[str(_pkg_dir)])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["hank_amd"] = _mod
_spec.loader.exec_module(_mod)
