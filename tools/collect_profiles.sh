#!/bin/bash
# usage (GPU box, repo root): tools/collect_profiles.sh <tag> <name>:<config>:<pmc spp>:<counter spp> ...
# e.g. tools/collect_profiles.sh r03 readme:readme_pbr:100:100 stress:stress_4k:8:8
# For every configuration: the kernel trace + FETCH_SIZE / WRITE_SIZE passes (tools/profile_bench.sh) and the four utilisation
# counter passes (tools/counters_report.sh); the small result files are gathered under gpurun_out/profiles_<tag>/ with the
# names profiles/ uses (copy them there afterwards: tools/keep_profiles.py <tag>).
set -e
tag=$1; shift
dst=$GRAFT_REPO_ROOT/gpurun_out/profiles_$tag
mkdir -p $dst
for spec in "$@"; do
  IFS=: read name config pmc_spp cnt_spp <<< "$spec"
  echo "== $name ($config)"
  bash tools/profile_bench.sh ${tag}_$name $config $pmc_spp > $dst/${tag}_${name}_profile.log 2>&1 || { tail -5 $dst/${tag}_${name}_profile.log; exit 1; }
  p=$GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_$name
  cp $p/summary.txt $dst/${tag}_${name}_summary.txt
  cp $p/hbm_traffic.json $dst/${tag}_${name}_hbm_traffic.json
  tail -1 $p/bench.json > $dst/${tag}_bench_${name}.json
  f=$(find $p/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $dst/${tag}_${name}_kernel_stats.csv
  cd $GRAFT_REPO_ROOT
  bash tools/counters_report.sh ${tag}_$name $config $cnt_spp > $dst/${tag}_${name}_counters.log 2>&1 || { tail -5 $dst/${tag}_${name}_counters.log; exit 1; }
  cp $GRAFT_REPO_ROOT/gpurun_out/counters_${tag}_$name/summary.txt $dst/${tag}_counters_${name}.txt
  cd $GRAFT_REPO_ROOT
  tail -2 $dst/${tag}_${name}_summary.txt
done
