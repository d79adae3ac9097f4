# Same-box A/B of library builds: the bench step time of each build, twice, in one gpurun call
#   bash profiles/ab_libs.sh name=path [name=path ...]
run() { PSVAE_HIP_LIB=$1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-sub-records --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', d['ms_per_step'], d['roofline']['frac'])"; }
for i in 1 2; do
  for spec in "$@"; do run "${spec#*=}" "${spec%%=*}"; done
done
