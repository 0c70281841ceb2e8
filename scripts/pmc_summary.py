"""Summarise rocprofv3 --pmc CSVs: per kernel name, mean counter value per launch."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            k = "k_str_pred" if "k_str_pred" in k else ("k_fixed_pred" if "k_fixed_pred" in k else k[:40])
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in sorted(acc.items()):
    if not any(s in k for s in ("k_str_pred", "k_fixed_pred")):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        sv = sorted(v)
        print("   %-24s median/launch %.5g  mean %.5g  min %.5g  max %.5g  (n=%d)" % (c, sv[len(sv) // 2], sum(v) / len(v), sv[0], sv[-1], len(v)))
