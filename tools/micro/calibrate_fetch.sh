#!/bin/bash
# FETCH_SIZE calibration on random 64-byte record gathers (tools/micro/gather_bench): known useful bytes per dispatch
# vs what the counter reports.  Usage on the GPU box: tools/micro/calibrate_fetch.sh
set -e
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cal && timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/cal -- $root/tools/micro/bin/gather_bench > /tmp/cal.log 2>&1
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/cal/**/*counter_collection.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "gather" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
lines = [l for l in open("/tmp/cal.log") if l.startswith("table")]
# two dispatches (repetitions) per printed line
for i, l in enumerate(lines):
    r = rows[2 * i + 1]
    parts = l.split()
    mb, mode, occ = float(parts[1]), int(parts[4]), int(parts[6].rstrip(":"))
    lane_loads = occ * 256 * 256 * 2000 * (1 if mode == 0 else 4)
    useful = lane_loads * 16
    rep = float(r["Counter_Value"]) * 1024
    print(f"table {mb:9.3f} MB mode {mode} occ {occ}: useful {useful/1e9:8.2f} GB  FETCH_SIZE {rep/1e9:8.2f} GB  ratio {rep/useful:6.3f}")
PY
