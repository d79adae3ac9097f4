#!/bin/bash
# Evidence for one build state (run on the GPU box through gpurun):
#   bash profiles/run_profile.sh r02_v1
# writes gpurun_out/<tag>/: the bench line, rocprofv3 kernel statistics of the one-stream and the
# two-stream run, and three separate counter passes (FETCH_SIZE, WRITE_SIZE, MFMA busy cycles).
set -e
TAG=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout -k 10 400 python bench.py --gemm-shapes > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-sub-records"
PSVAE_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/one -o one -- python3 $R/bench.py $ARGS > $O/one.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/two -o two -- python3 $R/bench.py $ARGS > $O/two.log 2>&1
SHORT="$ARGS --steps 4 --warmup 2"
PSVAE_STREAMS=1 timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- python3 $R/bench.py $SHORT > $O/fetch.log 2>&1
PSVAE_STREAMS=1 timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- python3 $R/bench.py $SHORT > $O/write.log 2>&1
PSVAE_STREAMS=1 timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o mfma -- python3 $R/bench.py $SHORT > $O/mfma.log 2>&1
find $O -name '*kernel_trace*' -size +4M -delete
find $O -name '*agent_info*' -delete
ls -R $O | head -40
