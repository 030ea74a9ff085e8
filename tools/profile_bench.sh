#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_bench.sh <tag> [<config> [<spp for the counter passes>]]
# Pass 1: kernel trace + stats of the bench command of <config> (default: the default bench command).
# Passes 2 and 3: fabric fetch / write byte counters, each in a run of its own (the two do not fit one pass), over one
# untimed frame of the same workload — at a reduced sample count when given: bytes per ray do not depend on it, and the
# per-dispatch counter collection of a 70-G-ray frame would not end.
set -e
tag=$1
config=${2:-readme_pbr}
pmc_spp=${3:-0}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
steps=2
if [ "$config" = "stress_4k" ] || [ "$config" = "material_4k" ]; then steps=1; fi
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --config $config --steps $steps --warmup 1 > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json
spp_arg=""
if [ "$pmc_spp" != "0" ]; then spp_arg="--spp $pmc_spp"; fi
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $root/bench.py --config $config $spp_arg --steps 1 --warmup 0 --no-roofline --no-cpu-baseline > $out/fetch.json 2> $out/fetch.err
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $root/bench.py --config $config $spp_arg --steps 1 --warmup 0 --no-roofline --no-cpu-baseline > $out/write.json 2> $out/write.err
python3 $root/tools/profile_summary.py $out $config > $out/summary.txt
cat $out/summary.txt
# keep only the small files: the per-dispatch traces are tens of MB
find $out -name "*kernel_trace.csv" -delete
find $out -name "*counter_collection.csv" -delete
find $out -name "*agent_info.csv" -delete
