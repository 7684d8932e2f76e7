"""bench.py's driver contract, as far as it can be checked without a GPU: the committed bench line of the last profiled
round carries every field the contract names, the roofline figures are self-consistent, the helper that reads the PMC
passes agrees with the committed summaries, and bench.py fails loudly without a HIP device."""

import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _last_bench_line():
    files = sorted((ROOT / 'profiles').glob('r*_bench.json'))
    assert files, 'no committed bench line under profiles/'
    return json.loads(files[-1].read_text().strip().splitlines()[-1])


def test_committed_bench_line_has_the_contract_fields():
    line = _last_bench_line()
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in line, key
    assert line['unit'] == 'solves/s' and line['higher_is_better'] is True and line['scaling'] == 'weak' and line['data'] == 'synthetic'
    assert line['vs_baseline'] is None  # BASELINE.md holds no published number for this metric
    assert 'workload' in line['config'] and 'model' not in line['config']
    assert abs(line['value'] - line['config']['batch_per_gpu'] * line['n_gpus'] / (line['ms_per_step'] * 1e-3)) < 1e-6 * line['value']
    r = line['roofline']
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert key in r, key
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    assert abs(r['achieved'] - r['alg_bytes_per_launch'] / (r['avg_launch_us'] * 1e-6) / 1e9) < 1e-6 * r['achieved']
    c = line['cpu_baseline']
    for key in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert key in c, key
    assert c['kind'] in ('reference', 'port') and c['cores'] >= 1 and c['unit'] == line['unit']
    assert line['check']['adders_match_oracle'] is True


def test_pmc_traffic_helper_reads_the_committed_passes():
    sys.path.insert(0, str(ROOT))
    import bench

    per_chain = bench.pmc_traffic_per_chain()
    assert per_chain and per_chain > 0
    r = _last_bench_line()['roofline']
    # within 1 %: the PMC passes were collected once more after the committed bench line was produced
    assert abs(per_chain * r['chains_per_launch'] - r['traffic']) < 1e-2 * r['traffic']
    assert 'c3_256x256_int8_batch64_single_chain' in bench.WORKLOADS and bench.WORKLOADS['c3_256x256_int8_batch64_single_chain'][:3] == (256, 256, 64)
    ks = bench.make_batch(4, 3, 2, first_seed=5)
    assert len(ks) == 2 and ks[0].shape == (4, 3) and ks[0].dtype.name == 'float32' and abs(ks[0]).max() <= 128


def test_bench_needs_a_gpu():
    from da4ml_amd import _binary

    if _binary.device_count() > 0:
        import pytest

        pytest.skip('a GPU is present')
    r = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--steps', '1', '--warmup', '0'], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'needs a HIP device' in (r.stdout + r.stderr)
