#!/usr/bin/env python3
"""usage (here, after tools/collect_profiles.sh ran on the GPU box): python tools/keep_profiles.py <tag>
Copies the gathered result files from gpurun_out/profiles_<tag>/ to profiles/ and merges the per-configuration traffic
files into profiles/<tag>_hbm_traffic.json, the file bench.py reads."""
import glob
import os
import shutil
import subprocess
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"profiles_{tag}")
parts = []
for f in sorted(glob.glob(os.path.join(src, f"{tag}_*"))):
    base = os.path.basename(f)
    if base.endswith(".log"):
        continue
    if base.endswith("_hbm_traffic.json"):
        parts.append(f)
        continue
    shutil.copy(f, os.path.join(root, "profiles", base))
    print("kept", base)
if parts:
    subprocess.run([sys.executable, os.path.join(root, "tools", "merge_traffic.py"), os.path.join(root, "profiles", f"{tag}_hbm_traffic.json")] + parts, check=True)
