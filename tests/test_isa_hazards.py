"""The one hand-written instruction of the library — the 16-byte write-through store `global_store_dwordx4 ... sc1` of
hank_kernels.h, issued from inline asm — relies on an `s_nop 1` inside the same asm statement so that the compiler's next
instruction cannot overwrite the data registers before the store has read them. Nothing in the language pins that: this
test compiles the kernels to gfx950 assembly (hipcc cross-compiles without a GPU) and checks every occurrence."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "julia-newtonraphsonhank_amd" / "csrc"


@pytest.mark.skipif(shutil.which("hipcc") is None and not Path("/opt/rocm/bin/hipcc").exists(), reason="hipcc not available")
def test_every_sc1_dwordx4_store_is_followed_by_its_nop(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = tmp_path / "hank.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
                    "-o", str(out), str(CSRC / "hank_hip.hip")], check=True, capture_output=True, timeout=600)
    lines = [l.strip() for l in out.read_text().splitlines()]
    code = [l for l in lines if l and not l.startswith((";", ".", "//")) and not l.endswith(":")]
    stores = [i for i, l in enumerate(code) if l.startswith("global_store_dwordx4") and " sc1" in l]
    assert stores, "the write-through 16-byte stores are gone: update this test with the kernels"
    for i in stores:
        assert re.match(r"s_nop\s+[1-9]", code[i + 1]), f"no s_nop after `{code[i]}` (next: `{code[i + 1]}`)"
    # the persistent sweeps must not have picked up scratch (a register spill would sit on their critical path)
    txt = out.read_text()
    for kern in ("k_xtan_back", "k_xtan_fwd", "k_xprimal_back", "k_xprimal_fwd"):
        blocks = re.findall(r"\.amdhsa_kernel (\S*" + kern + r"\S*)\n(.*?)\.end_amdhsa_kernel", txt, re.S)
        assert blocks, kern
        for name, body in blocks:
            if "ILi768E" not in name and "Li768E" not in name:
                continue            # the 1024-thread variants (n_e > 11) are allowed a few spilled registers
            m = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body)
            assert m and int(m.group(1)) == 0, f"{name} uses {m.group(1) if m else '?'} bytes of scratch"
