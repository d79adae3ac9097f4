#!/bin/bash
# kernel durations + SQ counters of the tall-K micro-benchmark for one library build:
#   bash profiles/tk_prof.sh NAME   (exp_libs/libpsvae_NAME.so) -> gpurun_out/tk_NAME/
set -e
NAME=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tk_$NAME; mkdir -p $O
export PSVAE_HIP_LIB=$R/exp_libs/libpsvae_$NAME.so
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/tests/bench_tallk.py 10 > $O/kt.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc -o pmc -- python3 $R/tests/bench_tallk.py 4 > $O/pmc.log 2>&1
find $O -name '*agent_info*' -delete
grep -h tallk $O/kt/*kernel_stats.csv || true
grep -h reduce $O/kt/*kernel_stats.csv || true
