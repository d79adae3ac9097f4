"""Reduce the rocprofv3 counter passes over bench.py (profiles/run_profile.sh: --pmc FETCH_SIZE,
--pmc WRITE_SIZE and --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE, csv output) to
per-launch figures of the three profiled kernel groups.
    python tools/pmc_traffic.py <dir with fetch/ write/ mfma/> out.json --build TAG --batch B --gemm-calls N
A "launch" of the matrix product is one psvae_gemm_f32 call (the bracket bench.py times): its
kernels are the panel / generic / tall-K kernel plus the split-K reduction that may follow, so the
group's bytes are divided by the number of calls (N per step x steps, steps = launches of the
collation fill kernel in the pass).  FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950
FETCH_SIZE counts 128-B requests of 16-B-per-lane loads at 64 B, so it is doubled (all these
kernels load float4 per lane); WRITE_SIZE is exact for 16-B-per-lane stores.
MFMA busy fraction of a group = sum SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x sum GRBM_GUI_ACTIVE / 8):
GRBM_GUI_ACTIVE is summed over the 8 XCDs, so an eighth of it is the kernel's cycles."""
import argparse
import collections
import csv
import glob
import json
import os


def group(name):
    if any(k in name for k in ('gemm_panel_kernel', 'gemm_generic_kernel', 'gemm_tallk_kernel', 'gemm_splitk_reduce')):
        return 'gemm_f32_kernel'
    if 'gine_aggregate_fwd' in name:
        return 'gine_aggregate_fwd_kernel'
    if 'gine_aggregate_bwd' in name or 'gine_table_reduce' in name:
        return 'gine_aggregate_bwd_kernel'
    return None


def is_main(name):   # the kernel that counts as the group's launch (helpers only add their bytes)
    return 'splitk_reduce' not in name and 'table_reduce' not in name


def load(path):
    """{group: {counter: sum}}, {group: main-kernel launches}, steps in the pass"""
    sums = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    seen = set()
    steps = set()
    for r in csv.DictReader(open(path)):
        name = r['Kernel_Name']
        if 'collate_kernel<true>' in name:
            steps.add(r['Dispatch_Id'])
        k = group(name)
        if not k:
            continue
        sums[k][r['Counter_Name']] += float(r['Counter_Value'])
        if is_main(name) and r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id'])
            launches[k] += 1
    return sums, launches, len(steps)


def one(pattern):
    files = glob.glob(pattern)
    if len(files) != 1:
        raise SystemExit(f'{pattern}: {len(files)} files')
    return files[0]


ap = argparse.ArgumentParser()
ap.add_argument('root')
ap.add_argument('out')
ap.add_argument('--build', required=True)
ap.add_argument('--batch', type=int, default=4096)
ap.add_argument('--gemm-calls', type=int, required=True, help='psvae_gemm_f32 calls per train step')
args = ap.parse_args()

fetch, fl, fsteps = load(one(os.path.join(args.root, 'fetch', '*counter_collection.csv')))
write, wl, wsteps = load(one(os.path.join(args.root, 'write', '*counter_collection.csv')))
mfma_path = glob.glob(os.path.join(args.root, 'mfma', '*counter_collection.csv'))
mfma = load(mfma_path[0])[0] if mfma_path else {}
out = {'_meta': {'build': args.build, 'batch': args.batch, 'steps_in_fetch_pass': fsteps, 'steps_in_write_pass': wsteps,
                 'gemm_calls_per_step': args.gemm_calls,
                 'note': 'FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE as read; KiB -> bytes x1024; '
                         'a gemm launch = one psvae_gemm_f32 call (its split-K reduction included)'}}
for k in fetch:
    n_f = args.gemm_calls * fsteps if k == 'gemm_f32_kernel' else fl[k]
    n_w = args.gemm_calls * wsteps if k == 'gemm_f32_kernel' else wl[k]
    f = 2.0 * fetch[k]['FETCH_SIZE'] / n_f * 1024.0
    w = write[k]['WRITE_SIZE'] / n_w * 1024.0
    out[k] = {'launches_sampled': n_f, 'fetch_bytes_per_launch': round(f), 'write_bytes_per_launch': round(w),
              'hbm_bytes_per_launch': round(f + w)}
    if k in mfma and mfma[k].get('GRBM_GUI_ACTIVE'):
        cycles = mfma[k]['GRBM_GUI_ACTIVE'] / 8.0
        out[k]['mfma_busy_frac'] = round(mfma[k]['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cycles), 4)
        out[k]['sq_busy_frac'] = round(mfma[k]['SQ_BUSY_CYCLES'] / (32.0 * cycles), 4) if mfma[k].get('SQ_BUSY_CYCLES') else None
json.dump(out, open(args.out, 'w'), indent=1)
print(json.dumps(out, indent=1))
