# Same-box A/B of an environment switch: bench step time with and without it, twice, in one gpurun call
#   bash profiles/ab_env.sh VAR=value
run() { env $1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-sub-records --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"; }
for i in 1 2; do run "PSVAE_AB_NONE=1" "default   "; run "$1" "$1"; done
