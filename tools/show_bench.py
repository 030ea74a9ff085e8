"""Print the headline fields of a bench.py line: show_bench.py <file with the json line>"""
import json
import sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j.get("roofline", {})
print(j["ms_per_step"], "ms/step", j["value"], j["unit"], "rays/sample", j.get("rays_per_sample"))
print({k: r.get(k) for k in ("achieved", "frac", "traffic", "algorithmic_GBps", "algorithmic_frac_of_hbm_peak", "bound", "avg_launch_ms", "traffic_source")})
if "cpu_baseline" in j:
    c = j["cpu_baseline"]
    print("cpu", c["value"], c["unit"], c.get("build"), "| parity build", c.get("parity_build", {}).get("value"))
