"""profiles/<tag>_pmc_{FETCH,WRITE}_SIZE_summary.txt -> profiles/pmc_latest.json (what bench.py reports as roofline.traffic)."""
import json, re, sys
from pathlib import Path

tag = sys.argv[1]
root = Path(__file__).resolve().parent.parent / "profiles"
p = root / "pmc_latest.json"
d = json.loads(p.read_text()) if p.exists() else {}

if (root / f"{tag}_pmc_P1.txt").exists():       # a wide-batch pass (scripts/dev_pmc_wide.sh <tag>): P1 = FETCH_SIZE, P2 = WRITE_SIZE, P3 / P4 = SQ counters
    def wmean(which, counter, kernel):
        t = (root / f"{tag}_pmc_{which}.txt").read_text()
        m = re.search(r"^" + re.escape(kernel) + r"<[^\n]*\n(?:\s+\S+\s+mean\s+[\d.]+\n)*?\s+" + counter + r"\s+mean\s+([\d.]+)", t, re.M)
        return float(m.group(1)) if m else None
    wl = d.setdefault("ks_2000x11_T300_N256", {})
    for k in ("k_wide_back", "k_wide_fwd", "k_fused_back", "k_fused_fwd"):
        f, w = wmean("P1", "FETCH_SIZE", k), wmean("P2", "WRITE_SIZE", k)
        if f is None or w is None:
            continue
        waves, valu, wcyc, wany, avalu = (wmean("P3", n, k) for n in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU"))
        c = {}
        if waves and valu:
            c["valu_insts_per_wave"] = valu / waves
        if wcyc and wany is not None:
            c["parked_frac"] = wany / wcyc
        if wcyc and avalu is not None:
            c["active_valu_frac"] = avalu / wcyc
        wl[k] = {"fetch_bytes": 2.0 * f * 1024, "write_bytes": w * 1024, "hbm_bytes": 2.0 * f * 1024 + w * 1024, "profile_tag": tag, "counters": c,
                 "note": f"rocprofv3 --pmc (separate passes, profiles/{tag}_pmc_P1..P4.txt, scripts/dev_pmc_wide.sh), per launch; FETCH_SIZE x2, see the N32 entry"}
    sys.path.insert(0, str(root.parent))
    from bench import kernel_source_sha16  # noqa: E402
    wl["kernel_source_sha16"] = kernel_source_sha16()
    wl["profile_tag"] = tag
    p.write_text(json.dumps(d, indent=1))
    print(json.dumps({k: v["hbm_bytes"] for k, v in wl.items() if isinstance(v, dict)}))
    sys.exit(0)


def mean_kb(counter, kernel):
    t = (root / f"{tag}_pmc_{counter}_summary.txt").read_text()
    m = re.search(r"^" + re.escape(kernel) + r"<[^\n]*\n\s+" + counter + r"\s+mean\s+([\d.]+)", t, re.M)
    return float(m.group(1))


note = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, profiles/{tag}_pmc_*_summary.txt, scripts/profile_round.sh), "
        "KB -> bytes, per launch; FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B: scripts/ubench/fetch_calib.hip reads 1 GiB "
        "with 8 B/lane, 16 B/lane and 256-B row-gather accesses and FETCH_SIZE reports 0.5 GiB for each, WRITE_SIZE is exact: "
        "profiles/r01f_calib_*_summary.txt)")
wl = d.setdefault("ks_2000x11_T300_N32", {})
found = []
for k in ("k_fused_back", "k_fused_fwd", "k_xdual_back", "k_xfwd"):       # the launches' kernels (tag ...l) or the persistent Dual pass's
    try:
        f, w = 2.0 * mean_kb("FETCH_SIZE", k) * 1024, mean_kb("WRITE_SIZE", k) * 1024
    except AttributeError:      # not in this profile
        continue
    wl[k] = {"fetch_bytes": f, "write_bytes": w, "hbm_bytes": f + w, "note": note, "profile_tag": tag}
    found.append(k)


def sq(counter, kernel, which):
    """mean of an SQ counter per launch of `kernel` from the SQ pass `which` (None where the pass or the counter is missing)."""
    f = root / f"{tag}_pmc_{which}_summary.txt"
    if not f.exists():
        return None
    blocks = re.split(r"^(?=\S)", f.read_text(), flags=re.M)
    for b in blocks:
        if b.startswith(kernel + "<") or b.startswith(kernel + " "):
            m = re.search(r"^\s+" + counter + r"\s+mean\s+([\d.]+)", b, re.M)
            if m:
                return float(m.group(1))
    return None


# measured issue and wait shares of the persistent Dual pass's kernels (the bench line prints them next to its model ceiling):
# VALU wave-instructions per wave and period, the share of wave-cycles parked at s_waitcnt / a barrier, LDS bank conflicts
P_HEADLINE = 299
for k in found:
    c = {}
    waves, valu, wcyc, wany, winst = (sq(n, k, "SQ1") for n in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"))
    conf, lact, wlds, vrd, vwr = (sq(n, k, "SQ2") for n in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
    avalu, avmem = (sq(n, k, "SQ3") for n in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM"))
    persistent = k.startswith("k_x")
    if waves and valu:
        c["valu_insts_per_wave" + ("_period" if persistent else "")] = valu / waves / (P_HEADLINE if persistent else 1)
    if wcyc and wany is not None:
        c["parked_frac"] = wany / wcyc
    if wcyc and winst is not None:
        c["issue_stall_frac"] = winst / wcyc
    if lact and conf is not None:
        c["lds_bank_conflict_frac"] = conf / lact
    if wcyc and wlds is not None:
        c["wait_lds_frac"] = wlds / wcyc
    if waves and vrd is not None:
        c["vmem_rd_per_wave" + ("_period" if persistent else "")] = vrd / waves / (P_HEADLINE if persistent else 1)
    if wcyc and avalu is not None:
        c["active_valu_frac"] = avalu / wcyc
    if wcyc and avmem is not None:
        c["active_vmem_frac"] = avmem / wcyc
    c["source"] = f"profiles/{tag}_pmc_SQ1/SQ2/SQ3_summary.txt (rocprofv3 --pmc, per launch; SQ_* cycle counters are quad-cycles, ratios are unit-free)"
    wl[k]["counters"] = c
sys.path.insert(0, str(root.parent))
from bench import kernel_source_sha16  # noqa: E402
wl["kernel_source_sha16"] = kernel_source_sha16()
wl["profile_tag"] = tag
p.write_text(json.dumps(d, indent=1))
print(json.dumps({k: wl[k]["hbm_bytes"] for k in found}))
