#!/bin/bash
# usage: tools/pmc.sh <tag> <counters...> -- runs tools/prof_extend.py under one rocprofv3 --pmc pass
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/tools/prof_extend.py 2 2 1.0 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
