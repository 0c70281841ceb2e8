"""Print the LIKE needle-class table of a bench line: python scripts/needle_table.py <bench.json>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", d["config"].get("evaluation_path"), "value %.3e step %.2f us" % (d["value"], d["ms_per_step"] * 1e3))
for k, v in d.get("like_needle_classes", {}).items():
    print("%-16s %-44s hot %8.1f us cold %8.1f us hits %9d ok %s | %s" % (
        k, v["predicate"][:44], v["kernel_us_hot"], v["kernel_us_l3_cold"] or 0, v["hits"], v["mask_equals_cpu_oracle"], v["path"][:70]))
