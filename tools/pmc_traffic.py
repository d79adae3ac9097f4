"""Turn two rocprofv3 counter passes over bench.py (--pmc FETCH_SIZE and --pmc WRITE_SIZE,
csv output) into per-launch HBM traffic of the three profiled kernel groups.
    python tools/pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv out.json
FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B... rocprofv3 reports kilobytes; on
gfx950 FETCH_SIZE counts 128-B requests of 16-B-per-lane loads at 64 B, so it is doubled
(all three kernels load float4 per lane); WRITE_SIZE is exact for 16-B-per-lane stores."""
import collections
import csv
import json
import sys


def group(name):
    if 'gemm_panel_kernel' in name or 'gemm_generic_kernel' in name:
        return 'gemm_f32_kernel'
    if 'gine_aggregate_fwd' in name:
        return 'gine_aggregate_fwd_kernel'
    if 'gine_aggregate_bwd' in name:
        return 'gine_aggregate_bwd_kernel'
    return None


def load(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = group(r['Kernel_Name'])
        if k:
            agg[k][0] += 1
            agg[k][1] += float(r['Counter_Value'])
    return agg


fetch, write = load(sys.argv[1]), load(sys.argv[2])
out = {}
for k in fetch:
    f = 2.0 * fetch[k][1] / fetch[k][0] * 1024.0
    w = write[k][1] / write[k][0] * 1024.0
    out[k] = {'launches_sampled': fetch[k][0], 'fetch_bytes_per_launch': round(f), 'write_bytes_per_launch': round(w),
              'hbm_bytes_per_launch': round(f + w),
              'note': 'FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE as read; KB -> bytes x1024'}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out, indent=1))
