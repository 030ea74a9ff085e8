#!/bin/bash
# usage (GPU box, repo root): BENCH_ARGS="--config stress_4k --spp 16" tools/ab_env.sh "name:VAR=value VAR2=value" "other:VAR=value" ...
# Alternates environment settings (the PTGPU_* variables a new context starts from, or PTGPU_LIB for another build of the
# library) around the same bench command on one box, ROUNDS (default 2) rounds, and prints frame time and the per-launch
# times of the traversal kernel and the instance pass for each.
mkdir -p gpurun_out
for i in $(seq 1 ${ROUNDS:-2}); do
  for spec in "$@"; do
    name=${spec%%:*}; vars=${spec#*:}
    env $vars timeout -k 10 ${AB_TIMEOUT:-300} python bench.py --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/abe_$name.json 2> gpurun_out/abe_$name.err || { echo "$name failed"; tail -3 gpurun_out/abe_$name.err; exit 1; }
    python - "$name" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/abe_%s.json" % sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(sys.argv[1], "frame ms", d["ms_per_step"], "Mrays/s", d["value"], "trav", r["avg_launch_ms"], "tlas", r["instance_pass"]["avg_launch_ms"],
      "shade share", r["shade_share_of_step"], "rays/launch", r["rays_per_launch"], flush=True)
PY
  done
done
