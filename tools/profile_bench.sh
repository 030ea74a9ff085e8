#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_bench.sh <tag>
# Pass 1: kernel trace + stats of the default bench command.  Passes 2 and 3: HBM fetch / write byte counters,
# each in a run of its own (the two do not fit one pass), over one untimed frame of the same workload.
set -e
tag=$1
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --steps 2 --warmup 1 > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $root/bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline > $out/fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $root/bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline > $out/write.log 2>&1
python3 $root/tools/profile_summary.py $out > $out/summary.txt
cat $out/summary.txt
# keep only the small files: the per-dispatch traces are tens of MB
find $out -name "*kernel_trace.csv" -delete
find $out -name "*counter_collection.csv" -delete
find $out -name "*agent_info.csv" -delete
