"""Merge the per-configuration hbm_traffic.json files written by tools/profile_bench.sh into the file bench.py reads:
    python tools/merge_traffic.py profiles/r02_hbm_traffic.json gpurun_out/prof_<tag>/hbm_traffic.json ...
An entry replaces the one with the same key ("<config>/<wide|binary>")."""
import json
import os
import sys

dst = sys.argv[1]
doc = json.load(open(dst)) if os.path.exists(dst) else {"note": "fabric bytes per ray of every kernel, from separate rocprofv3 --pmc "
                                                                "FETCH_SIZE / WRITE_SIZE passes (tools/profile_bench.sh)", "configs": {}}
for path in sys.argv[2:]:
    e = json.load(open(path))
    doc["configs"][e.pop("config_key")] = e
json.dump(doc, open(dst, "w"), indent=1)
print(sorted(doc["configs"]))
