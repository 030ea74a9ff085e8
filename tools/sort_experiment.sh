#!/bin/bash
# usage (GPU box, repo root): tools/sort_experiment.sh [launch]   -> traversal kernel time per ordering, from rocprofv3
# Needs a developer build of the library that can write one launch's rays to a file (here, before the GPU call):
#   tools/build_variant.sh dump -DPT_DEV_DUMP_RAYS      and run with PTGPU_LIB=$PWD/pathtracer_amd/lib/variants/libptgpu_dump.so
set -e
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/sortexp
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 python3 -u $root/tools/sort_experiment.py dump ${1:-5} 2>&1 | grep -v amdgpu
IFS=";" read -ra modes <<< "${MODES:-plain;sorted od;sorted fine}"
for mode in "${modes[@]}"; do
  tag=$(echo $mode | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag -- python3 $root/tools/sort_experiment.py $mode > $out/$tag.log 2>&1
  tail -1 $out/$tag.log
  python3 - "$out/$tag" "$mode" <<'PY'
import sys, glob, csv
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "traverse_kernel" in r["Name"] or "tlas_kernel" in r["Name"]:
        print("   ", sys.argv[2], r["Name"][:40], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1))
PY
  find $out/$tag -name "*kernel_trace.csv" -delete
done
