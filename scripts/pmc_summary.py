"""Summarise the rocprofv3 --pmc passes of scripts/profile_round.sh into hbm_traffic.json — MERGED into what the file
already holds (a partial refresh keeps the other workloads) and CHECKED for completeness.

usage: pmc_summary.py <gpurun_out/tag> [required workload ...]
       (expects <workload>_FETCH_SIZE/, <workload>_WRITE_SIZE/, calib_FETCH_SIZE/ under the directory)
FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3.  The unit of FETCH_SIZE on gfx950 is only documented for
16-byte coalesced streaming reads (it reports half the bytes); the calibration pass measures it for the access shapes
the scan kernels use, and every workload's fetch is reported raw and scaled by the factor of its dominant shape.
A workload that launches several kernels per evaluation (the gather path) is the SUM of its kernels."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
required = sys.argv[2:]
CALIB_BYTES = 1 << 30
path = os.path.join(root, "hbm_traffic.json")


def medians(pattern):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(root, pattern, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: sorted(v)[len(v) // 2] for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


try:
    out = json.load(open(path))
except (OSError, ValueError):
    out = {}
out["_comment"] = ("HBM bytes per launch of the dominant kernel(s) from rocprofv3 --pmc (one pass per counter, "
                   "scripts/profile_round.sh); FETCH_SIZE / WRITE_SIZE are KB, medians over launches; fetch_factor = real "
                   "bytes per FETCH_SIZE byte measured by scripts/pmc_calibrate.py for the kernel's dominant access shape; "
                   "traffic_bytes = FETCH_SIZE*1024*fetch_factor + WRITE_SIZE*1024.  Merged across partial refreshes.")
calib, _ = medians("calib_FETCH_SIZE")
factors = dict(out.get("fetch_size_calibration", {}).get("real_bytes_per_reported_byte", {}))
for k, kb in calib.items():
    shape = "scattered8" if "scattered8" in k else ("coalesced%s" % k.split("<")[1].split(">")[0] if "<" in k else k)
    if kb:
        factors[shape] = CALIB_BYTES / (kb * 1024.0)
out["fetch_size_calibration"] = {"bytes_per_launch": CALIB_BYTES, "real_bytes_per_reported_byte": factors}
# dominant access shape of each kernel family
SHAPE = {"k_str_pred": "coalesced8", "k_like_lean": "coalesced8", "k_like_flat": "coalesced16", "k_like_scanall": "coalesced8", "k_fixed_pred_reg": "coalesced4",
         "k_fixed_pred": "coalesced16", "k_fixed_chain": "coalesced4", "k_fixed_gather": "coalesced16",
         "k_sel_entry_counts": "coalesced8", "k_scan_": "coalesced8"}
FAMILIES = ("k_fixed_pred_reg", "k_fixed_pred", "k_fixed_chain", "k_fixed_gather", "k_sel_entry_counts", "k_scan_", "k_str_pred",
            "k_like_lean", "k_like_flat", "k_like_scanall")
SUMMED = {"gather_10pct"}  # every kernel of the workload belongs to one evaluation
for d in sorted(glob.glob(os.path.join(root, "*_FETCH_SIZE"))):
    wl = os.path.basename(d)[: -len("_FETCH_SIZE")]
    if wl == "calib":
        continue
    fetch, n = medians(wl + "_FETCH_SIZE")
    write, _ = medians(wl + "_WRITE_SIZE")
    entry = {}
    for k, kb in fetch.items():
        fam = next((f for f in FAMILIES if f in k), None)
        if fam is None:
            continue
        factor = factors.get(SHAPE[fam]) or 2.0
        wkb = write.get(k, 0.0)
        short = k.split("(lc::")[0].replace("void lc::(anonymous namespace)::", "")  # kernel + template arguments
        entry[short] = {"FETCH_SIZE_KB": kb, "WRITE_SIZE_KB": wkb, "launches": n[k], "fetch_shape": SHAPE[fam],
                        "fetch_factor": factor, "traffic_bytes": int(kb * 1024 * factor + wkb * 1024),
                        "traffic_bytes_raw_counter": int(kb * 1024 + wkb * 1024)}
    if not entry:
        continue
    kernels = [v for v in entry.values() if isinstance(v, dict)]
    if wl in SUMMED:
        # (the kernels of the looped evaluation: those launched as often as the most-launched one — the predicate pass that
        # makes the selection runs once and is not part of the gather)
        top = max(v["launches"] for v in kernels)
        entry["traffic_bytes"] = sum(v["traffic_bytes"] for v in kernels if v["launches"] == top)
        entry["_launches"] = top
    else:
        # the workload's figure is that of the kernel the timed loop launches (the one with the most launches; the
        # byte-accounting instantiation of k_str_pred and the plan's trial launch run once per process)
        best = max(kernels, key=lambda v: v["launches"])
        entry["traffic_bytes"] = best["traffic_bytes"]
        entry["_launches"] = best["launches"]
    out[wl] = entry
json.dump(out, open(path, "w"), indent=1)
print(json.dumps({k: (v.get("traffic_bytes") if k != "fetch_size_calibration" else v)
                  for k, v in out.items() if isinstance(v, dict)}, indent=1))
missing = [w for w in required if not isinstance(out.get(w), dict) or "traffic_bytes" not in out[w]]
if missing or not factors:
    print("INCOMPLETE profile set: no traffic figure for %s%s" % (missing, "" if factors else "; no calibration factors"))
    sys.exit(3)
print("profile set complete: %d workloads, calibration %s" % (len(required), {k: round(v, 5) for k, v in factors.items()}))
