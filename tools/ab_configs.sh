#!/bin/bash
# usage (GPU box, repo root): tools/ab_configs.sh <variantA> <variantB> ...
# tools/ab_variants.sh over the other bench configurations (binary nodes, ambient occlusion, material frame, stress frame at
# reduced sample counts), two rounds each.
# AB_ONLY=<text> keeps the configurations whose arguments contain the text.
for args in "--binary-nodes --spp 100" "--config ao_1080p" "--config material_4k --spp 256" "--config stress_4k --spp 16"; do
  case "$args" in *"${AB_ONLY:-}"*) ;; *) continue ;; esac
  echo "== $args"
  for i in 1 2; do
    for v in "$@"; do
      PTGPU_LIB=$PWD/pathtracer_amd/lib/variants/libptgpu_$v.so timeout -k 10 300 python bench.py --no-cpu-baseline $args --steps 2 --warmup 1 > gpurun_out/abc_$v.json 2>/dev/null || exit 1
      python - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/abc_%s.json" % sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(sys.argv[1], d["ms_per_step"], "trav", r["avg_launch_ms"], "tlas", r["instance_pass"]["avg_launch_ms"], "shade share", r["shade_share_of_step"], flush=True)
PY
    done
  done
done
