"""Summarise the rocprofv3 --pmc passes of scripts/profile_round.sh into hbm_traffic.json.

usage: pmc_summary.py <gpurun_out/tag>   (expects <workload>_FETCH_SIZE/, <workload>_WRITE_SIZE/, calib_FETCH_SIZE/)
FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3.  The unit of FETCH_SIZE on gfx950 is only documented for
16-byte coalesced streaming reads (it reports half the bytes); the calibration pass measures it for the access shapes
the scan kernels use, and every workload's fetch is reported raw and scaled by the factor of its dominant shape."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
CALIB_BYTES = 1 << 30


def medians(pattern):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(root, pattern, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: sorted(v)[len(v) // 2] for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


out = {"_comment": "HBM bytes per launch of the dominant kernel from rocprofv3 --pmc (one pass per counter, "
                   "scripts/profile_round.sh); FETCH_SIZE / WRITE_SIZE are KB, medians over launches; "
                   "fetch_factor = real bytes per FETCH_SIZE byte measured by scripts/pmc_calibrate.py for the kernel's "
                   "dominant access shape; traffic_bytes = FETCH_SIZE*1024*fetch_factor + WRITE_SIZE*1024."}
calib, _ = medians("calib_FETCH_SIZE")
factors = {}
for k, kb in calib.items():
    shape = "scattered8" if "scattered8" in k else ("coalesced%s" % k.split("<")[1].split(">")[0] if "<" in k else k)
    factors[shape] = CALIB_BYTES / (kb * 1024.0) if kb else None
out["fetch_size_calibration"] = {"bytes_per_launch": CALIB_BYTES, "real_bytes_per_reported_byte": factors}
# dominant access shape of each kernel family
SHAPE = {"k_str_pred": "coalesced8", "k_like_lean": "coalesced8", "k_fixed_pred_reg": "coalesced4", "k_fixed_pred": "coalesced16",
         "k_fixed_chain": "coalesced4"}
for d in sorted(glob.glob(os.path.join(root, "*_FETCH_SIZE"))):
    wl = os.path.basename(d)[: -len("_FETCH_SIZE")]
    if wl == "calib":
        continue
    fetch, n = medians(wl + "_FETCH_SIZE")
    write, _ = medians(wl + "_WRITE_SIZE")
    for k, kb in fetch.items():
        fam = next((f for f in ("k_fixed_pred_reg", "k_fixed_pred", "k_fixed_chain", "k_str_pred", "k_like_lean") if f in k), None)
        if fam is None:
            continue
        factor = factors.get(SHAPE[fam]) or 2.0
        wkb = write.get(k, 0.0)
        short = k.split("(lc::")[0].replace("void lc::(anonymous namespace)::", "")  # kernel + template arguments
        out.setdefault(wl, {})[short] = {"FETCH_SIZE_KB": kb, "WRITE_SIZE_KB": wkb, "launches": n[k],
                                         "fetch_shape": SHAPE[fam], "fetch_factor": factor,
                                         "traffic_bytes": int(kb * 1024 * factor + wkb * 1024),
                                         "traffic_bytes_raw_counter": int(kb * 1024 + wkb * 1024)}
        # the workload's figure is that of the kernel the timed loop launches (the one with the most launches; the
        # byte-accounting instantiation of k_str_pred runs once per process and also reads the fingerprints)
        if n[k] >= out[wl].get("_launches", 0):
            out[wl]["_launches"] = n[k]
            out[wl]["traffic_bytes"] = int(kb * 1024 * factor + wkb * 1024)
json.dump(out, open(os.path.join(root, "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
