"""Point profiles/pmc_index.json's in-frame entries at the counter files of a round tag:
   python scripts/pmc_index_update.py r06      (profiles/r06_pmc_frame_{f32,fp16,bf16}.json must exist)
bench.py reads the index to fill roofline.traffic / mfma_busy_frac and to decide counters_stale."""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
path = os.path.join(root, "profiles", "pmc_index.json")
doc = json.load(open(path))
entry = {}
for mode in ("f32", "bf16", "fp16"):
    name = f"{tag}_pmc_frame_{mode}.json"
    if not os.path.exists(os.path.join(root, "profiles", name)):
        sys.exit(f"profiles/{name} is missing")
    entry[mode] = name
doc["in_frame"] = entry
json.dump(doc, open(path, "w"), indent=1)
print("in_frame ->", entry)
