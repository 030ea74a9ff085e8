#!/bin/bash
set -e
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 python3 $root/tools/sort_experiment.py dump 12
for m in "plain" "sorted od" "sorted do"; do
  tag=$(echo $m | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/se_$tag -- python3 $root/tools/sort_experiment.py $m > /tmp/se_$tag.log 2>&1
  tail -3 /tmp/se_$tag.log
  grep -h "traverse_kernel\|tlas_kernel" $(find /tmp/se_$tag -name "*kernel_stats.csv") | cut -c1-30,150-400
done
