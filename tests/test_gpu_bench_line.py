"""bench.py prints one JSON line that carries the agreed fields, with roofline fractions that are fractions."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_fields_and_fractions():
    run = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '2', '--batch', '1024',
                          '--no-cpu-baseline', '--no-sub-records'], capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline'):
        assert key in rec, key
    assert rec['n_gpus'] == 1 and rec['steps'] == 3 and rec['warmup'] == 2 and rec['dtype'] == 'f32' and rec['scaling'] == 'weak'
    assert rec['value'] > 0 and abs(rec['value'] - 1024 / (rec['ms_per_step'] * 1e-3)) <= 0.01 * rec['value']
    assert 'workload' in rec['config'] and 'model' not in rec['config']
    for key in ('roofline', 'roofline_mfma', 'roofline_hbm', 'roofline_hbm_bwd'):
        r = rec[key]
        assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s')
        assert 0.0 < r['frac'] <= 1.05, (key, r)
        assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
        assert abs(r['achieved'] - r['work_per_launch'] / (r['avg_launch_us'] * 1e-6) / (1e9 if r['unit'] == 'GB/s' else 1e12)) \
            <= 0.01 * r['achieved'], (key, r)
    assert rec['roofline']['kernel'] == 'gemm_f32_kernel'
    assert rec['roofline']['traffic'] is None               # counters exist for the headline batch size only
