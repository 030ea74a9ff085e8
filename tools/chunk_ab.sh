#!/bin/bash
# usage (GPU box, repo root): bash tools/chunk_ab.sh  -- frame time of every bench configuration with the default batch size and with one of 256 Mi samples
for cfg in readme_pbr ao_1080p material_4k stress_4k; do
  steps=3; [ $cfg = stress_4k ] && steps=1
  for chunk in default 268435456 134217728; do
    if [ $chunk = default ]; then unset PTGPU_CHUNK; else export PTGPU_CHUNK=$chunk; fi
    timeout -k 10 300 python bench.py --config $cfg --steps $steps --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/chunk_${cfg}_${chunk}.json 2>/dev/null || exit 1
    python - $cfg $chunk <<'PY'
import json, sys
d = json.loads(open("gpurun_out/chunk_%s_%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], d["ms_per_step"], d["value"], flush=True)
PY
  done
done
