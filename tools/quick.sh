#!/bin/bash
# One gpurun call: GPU tests (optional filter), bench line, one-stream rocprof kernel statistics.
#   tools/quick.sh <tag> [pytest -k filter | "none"]
TAG=$1; FILTER=$2
CMD='O=$GRAFT_REPO_ROOT/gpurun_out/'$TAG'; mkdir -p $O; '
if [ "$FILTER" != "none" ]; then
  if [ -n "$FILTER" ]; then K="-k \"$FILTER\""; else K=""; fi
  CMD+='timeout -k 10 800 python -u -m pytest tests -m gpu -q -x --timeout 240 --timeout-method thread '$K' > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log; '
fi
CMD+='timeout -k 10 250 python bench.py --no-cpu-baseline --no-sub-records > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "import json; d=json.load(open(\"$O/bench.json\")); print(d[\"value\"], d[\"ms_per_step\"], d[\"ms_per_step_without_event_brackets\"], \"gemm\", d[\"roofline\"][\"frac\"], \"fwd\", d[\"roofline_hbm\"][\"frac\"], d[\"roofline_hbm\"][\"avg_launch_us\"], \"bwd\", d[\"roofline_hbm_bwd\"][\"frac\"], d[\"roofline_hbm_bwd\"][\"avg_launch_us\"])"; cd /tmp && export TMPDIR=/tmp; PSVAE_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/one -o one -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sub-records > $O/one.log 2>&1; echo rc=$?; find $O -name "*kernel_trace*" -delete; find $O -name "*agent_info*" -delete'
/usr/local/graft/bin/gpurun --timeout 1000 -- "$CMD" 2>&1 | grep -v "^\[gpurun\] every\|amdgpu.ids\|^\[gpurun\] sending"
python - <<PY
import csv
rows = list(csv.DictReader(open('gpurun_out/$TAG/one/one_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
steps = 25
print('total kernel ms/step (one stream)', round(tot/steps/1e6, 3))
for r in rows[:${TOP:-24}]:
    print(f"{float(r['TotalDurationNs'])/steps/1e3:9.1f} us/step  n {int(r['Calls'])/steps:5.1f}  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}")
PY
