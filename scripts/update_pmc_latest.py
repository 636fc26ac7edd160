"""profiles/<tag>_pmc_{FETCH,WRITE}_SIZE_summary.txt -> profiles/pmc_latest.json (what bench.py reports as roofline.traffic)."""
import json, re, sys
from pathlib import Path

tag = sys.argv[1]
root = Path(__file__).resolve().parent.parent / "profiles"
p = root / "pmc_latest.json"
d = json.loads(p.read_text()) if p.exists() else {}


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
sys.path.insert(0, str(root.parent))
from bench import kernel_source_sha16  # noqa: E402
wl["kernel_source_sha16"] = kernel_source_sha16()
wl["profile_tag"] = tag
p.write_text(json.dumps(d, indent=1))
print(json.dumps({k: wl[k]["hbm_bytes"] for k in found}))
