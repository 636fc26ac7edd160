"""What the language does not pin, checked in the gfx950 assembly (hipcc cross-compiles without a GPU):
1. the one hand-written store of the library — `global_store_dwordx4 ... sc1` of hank_kernels.h, issued from inline asm —
   relies on an `s_nop 1` inside the same asm statement so that the compiler's next instruction cannot overwrite the data
   registers before the store has read them;
2. the persistent forward sweep drains its stores with a COUNTED wait (`s_waitcnt vmcnt(N)` behind a batch of exactly N
   unconditional loads, hank_xsweep.h:k_xfwd): were the compiler to drop, merge or branch around one of those loads, the wait
   would return before the stores have completed and a neighbour would read a stale row."""
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
    for kern in ("k_xtan_back", "k_xfwd", "k_xprimal_back", "k_xdual_back"):
        blocks = re.findall(r"\.amdhsa_kernel (\S*" + kern + r"\S*)\n(.*?)\.end_amdhsa_kernel", txt, re.S)
        assert blocks, kern
        for name, body in blocks:
            if "ILi768E" not in name and "Li768E" not in name:
                continue            # the 1024-thread variants (n_e > 11) are allowed a few spilled registers
            m = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body)
            assert m and int(m.group(1)) == 0, f"{name} uses {m.group(1) if m else '?'} bytes of scratch"


@pytest.mark.skipif(shutil.which("hipcc") is None and not Path("/opt/rocm/bin/hipcc").exists(), reason="hipcc not available")
def test_counted_drain_of_the_forward_sweep_matches_its_batch(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = tmp_path / "hank.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
                    "-o", str(out), str(CSRC / "hank_hip.hip")], check=True, capture_output=True, timeout=600)
    name, inside, ops, ctl, pending, seen = None, False, 0, [], False, 0
    for raw in out.read_text().splitlines():
        l = raw.strip()
        m = re.match(r"^(_ZN4hank6k_xfwd\w+):", raw)
        if m:
            name = m.group(1)
        if "XFWD_BATCH_BEGIN" in l:
            inside, ops, ctl = True, 0, []
            continue
        if "XFWD_BATCH_END" in l:
            inside, pending = False, True
            continue
        op = l.split()[0] if l else ""
        if inside:
            if re.match(r"(global_|buffer_|scratch_|flat_)", op):
                ops += 1
            if op.startswith("s_cbranch") or op == "s_branch" or op == "s_barrier":
                ctl.append(op)
        elif pending and op == "s_waitcnt":
            n = int(re.search(r"vmcnt\((\d+)\)", l).group(1))
            seen += 1
            assert not ctl, f"{name}: control flow inside the counted batch: {ctl}"
            # fewer instructions than the immediate = the wait returns before the stores are done (a race); more (a spilled
            # register in the 1024-thread variants) only waits for part of the batch as well
            assert ops >= n, f"{name}: vmcnt({n}) behind a batch of {ops} vector-memory instructions"
            if "Li768E" in name:
                assert ops == n, f"{name}: vmcnt({n}) behind a batch of {ops} vector-memory instructions"
            pending = False
    assert seen >= 7, f"only {seen} counted drains found: update this test with the kernel"
