#!/bin/bash
# usage (GPU box, repo root): tools/ab_variants.sh <variantA> <variantB> ...
# Alternates the libraries pathtracer_amd/lib/variants/libptgpu_<variant>.so (tools/build_variant.sh), three default bench
# runs each, and prints frame and kernel times: runs of one call share a machine, so 1 % differences can be told apart.
for i in 1 2 3; do
  for v in "$@"; do
    PTGPU_LIB=$PWD/pathtracer_amd/lib/variants/libptgpu_$v.so timeout -k 10 200 python bench.py --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/ab_$v.json 2>/dev/null || exit 1
    python - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_%s.json" % sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(sys.argv[1], d["ms_per_step"], "trav", r["avg_launch_ms"], "tlas", r["instance_pass"]["avg_launch_ms"], "shade share", r["shade_share_of_step"], flush=True)
PY
  done
done
