"""Builds profiles/pmc_traffic.json from the per-group summaries that scripts/gpu_pmc.sh leaves under
profiles/r01/<pmc dir>/g*/**/*summary.txt (rocprofv3 --pmc passes of bench.py, averaged per dispatch).
usage: make_pmc_traffic.py precision=dir [precision=dir ...]      (round 2: bf16x3=profiles/r02/pmc_bf16x3)"""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP = ["FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "TCC_HIT_sum", "TCC_MISS_sum",
        "SQ_INSTS_MFMA", "SQ_INSTS_VALU"]


def parse(d):
    tab = {}
    for f in glob.glob(os.path.join(d, "**", "*summary.txt"), recursive=True):
        for line in open(f):
            m = re.match(r"(\S.*?)\s+(\S+)\s+avg_per_dispatch=(\S+) n=\d+", line.strip())
            if not m:
                continue
            kern = re.sub(r"<.*", "", m.group(1)).strip()
            tab.setdefault(kern, {})[m.group(2)] = float(m.group(3))
    out = {}
    for k, c in tab.items():
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        e = {"FETCH_SIZE_KiB": c["FETCH_SIZE"], "WRITE_SIZE_KiB": c["WRITE_SIZE"],
             "hbm_bytes_per_launch": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0}
        for n in KEEP[2:]:
            if n in c:
                e[n] = c[n]
        if c.get("GRBM_GUI_ACTIVE"):
            e["mfma_util"] = (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0) / (c["GRBM_GUI_ACTIVE"] / 8.0)
        out[k] = e
    return out


def main():
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    res = json.load(open(path)) if os.path.exists(path) else {}
    srcs = []
    for a in sys.argv[1:]:
        prec, d = a.split("=", 1)
        res[prec] = parse(os.path.join(ROOT, d) if not os.path.isabs(d) else d)
        srcs.append(f"{prec}: {d}")
    sys.path.insert(0, ROOT)
    import bench
    res["_csrc_sha"] = bench.csrc_sha()          # the kernel sources these passes were collected on (bench.py refuses a stale table)
    res["_note"] = ("hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: gfx950 counts the 128-B requests of wide coalesced "
                    "reads as 64 B (MI355X_MICROARCH.md, HBM section); calibrated on the streaming update kernel, which reads 4 x 1.206 GB "
                    "and reports FETCH_SIZE = 2.35e6 KiB. mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs. "
                    "Sources (rocprofv3 --pmc passes of bench.py at cfg2, scripts/gpu_pmc.sh + scripts/make_pmc_traffic.py): " + "; ".join(srcs))
    json.dump(res, open(path, "w"), indent=1)
    for prec in res:
        if prec.startswith("_"):
            continue
        for k, e in res[prec].items():
            print(prec, k, "%.3f GB" % (e["hbm_bytes_per_launch"] / 1e9), "mfma %.2f" % e.get("mfma_util", 0.0))


if __name__ == "__main__":
    main()
