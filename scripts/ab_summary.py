"""Key kernel times of the A/B runs written by scripts/ab_fixed.sh (one JSON bench line per variant)."""
import glob
import json
import os
import sys


def pick(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def main():
    rows = []
    for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
        try:
            line = [l for l in open(f) if l.startswith("{")][-1]
            j = json.loads(line)
        except Exception as e:  # noqa: BLE001
            print(os.path.basename(f), "no bench line:", e)
            continue
        sec = j.get("secondary", {})
        r = {"variant": os.path.basename(f)[:-5],
             "main_w12_us": round(1e3 * pick(j, "roofline", "kernel_ms"), 2),
             "main_w12_cold_us": round(1e3 * (pick(j, "roofline", "kernel_ms_l3_cold") or 0), 2)}
        for name in ("int64_gt_w62", "date32_gt_w12", "int16_gt_w12", "decimal_gt_w4", "int64_gt_w17"):
            ms = pick(sec, name, "kernel_ms")
            cold = pick(sec, name, "kernel_ms_l3_cold")
            r[name] = "%s/%s" % (round(1e3 * ms, 1) if ms else None, round(1e3 * cold, 1) if cold else None)
        for g in ("10pct", "0.1pct"):
            ms = pick(sec, "int64_gt_w62", "get_with_selection", g, "ms")
            r["gather_" + g] = round(1e3 * ms, 1) if ms else None
        q6 = sec.get("tpch_q6_pushdown", {})
        r["q6"] = {k: (round(1e3 * v["ms"], 1) if isinstance(v, dict) and "ms" in v else v)
                   for k, v in q6.items() if isinstance(v, dict) and "ms" in v} or q6.get("error")
        rows.append(r)
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
