#!/bin/bash
# Counter passes over ONE replayed conv product (95 167 x 300 x 300, forward form X W^T + b with ReLU and
# the input-gradient form dY W), each pass in a run of its own:  bash profiles/conv_pmc.sh <tag>
# -> gpurun_out/<tag>/pmc_<mode>_<pass>/...counter_collection.csv
set -e
TAG=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for MODE in fwd dx; do
  timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/pmc_${MODE}_lds -o p -- python3 $R/tests/bench_gemm_one.py 95167 300 300 $MODE 4 > $O/pmc_${MODE}_lds.log 2>&1
  timeout -s KILL 90 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${MODE}_mfma -o p -- python3 $R/tests/bench_gemm_one.py 95167 300 300 $MODE 4 > $O/pmc_${MODE}_mfma.log 2>&1
done
find $O -name '*agent_info*' -delete
ls -R $O | head -30
