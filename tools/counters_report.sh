#!/bin/bash
# usage (GPU box, repo root): tools/counters_report.sh <tag> [<config> [<spp>]]
# Per-kernel utilisation counters of one frame of a bench configuration (default: the README workload at 8 spp), four
# rocprofv3 --pmc passes, each a run of its own.
set -e
tag=$1
config=${2:-readme_pbr}
spp=${3:-8}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/counters_$tag
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 $root/bench.py --config $config --spp $spp --steps 1 --warmup 0 --no-roofline --no-cpu-baseline > $out/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LEVEL_WAVES SQ_CYCLES GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE
run tcp TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE
echo "config $config at $spp spp" > $out/summary.txt
python3 $root/tools/counters_summary.py $out >> $out/summary.txt
cat $out/summary.txt
find $out -name "*.csv" -delete
