"""bench.py's output contract: exactly one line on stdout, a JSON object with the agreed keys (the driver parses it),
whatever the libraries underneath print; run here at a small batch with a live one-rank RCCL group, whose version
banner goes to stdout when the communicator comes up."""
import json
import os
import subprocess
import sys

import pytest

gpu = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@gpu
def test_bench_prints_one_json_line():
    env = dict(os.environ, VCA_FORCE_EXCHANGE='1', MASTER_PORT='29641')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--batch', '2',
                        '--no-alt', '--no-cpu-baseline'], capture_output=True, text=True, timeout=280, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[:2000]
    res = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert k in res, k
    assert res['n_gpus'] == 1 and res['steps'] == 2 and res['warmup'] == 1 and res['unit'] == 'frames/s'
    assert res['higher_is_better'] is True and res['scaling'] == 'weak' and res['vs_baseline'] is None
    assert abs(res['value'] - 2 * 75 / (res['ms_per_step'] * 1e-3)) <= 1e-3 * res['value']
    assert 'workload' in res['config'] and 'model' not in res['config']
    r = res['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert '3 hip-graphs + 2 all-reduces' in res['launch']
