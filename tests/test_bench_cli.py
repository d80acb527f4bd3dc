"""CPU: bench.py's command line (the driver's contract flags + the per-row switch) and the pieces of it that run without a device."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *args], capture_output=True, text=True, cwd=ROOT)


def test_flags_and_loud_failure_without_a_device():
    r = _run('--help')
    assert r.returncode == 0
    for flag in ('--gpus', '--steps', '--warmup', '--row'):
        assert flag in r.stdout
    if not torch.cuda.is_available():
        for extra in ([], ['--row', 'vocoder'], ['--row', 'train'], ['--row', 'fs2'], ['--split']):
            r = _run('--steps', '1', '--warmup', '0', *extra)
            assert r.returncode != 0 and 'needs an MI355X' in (r.stderr + r.stdout)      # no CPU fallback for the product path


def test_vocoder_row_accounting_and_cpu_leg():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.vocoder_flop_per_frame(bench.VOC_CONFIG) == 38_510_592
    cb = bench.cpu_baseline_vocoder(budget_s=1.0)
    assert cb['kind'] == 'port' and cb['unit'] == 'mel-frames/s' and cb['value'] > 0 and cb['cores'] >= 1 and 'forwards' in cb['sample']
    cf = bench.cpu_baseline_fs2(budget_s=1.0)
    assert cf['kind'] == 'port' and cf['value'] > 0 and 'teacher-forced' in cf['sample']
    ct = bench.cpu_baseline_train(budget_s=1.0)
    assert ct['kind'] == 'port' and ct['unit'] == 'frames/s' and ct['value'] > 0 and 'forward + backward' in ct['sample']


def test_gpus_n_without_a_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with WORLD_SIZE unset spawns 2 ranks through torch.distributed.run on 127.0.0.1; here (no device) each
    rank then fails loudly - what is under test is that the spawn happens and the exit code propagates."""
    r = _run('--help')
    assert '--config' in r.stdout
    if torch.cuda.is_available():
        return
    r = _run('--gpus', '2', '--steps', '1', '--warmup', '0')
    assert r.returncode != 0 and 'needs an MI355X node with at least 2 visible devices' in (r.stderr + r.stdout)
    env = dict(os.environ, DSD_BENCH_SPAWN_ANYWAY='1')
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], capture_output=True, text=True,
                       cwd=ROOT, env=env, timeout=300)
    out = r.stderr + r.stdout
    assert r.returncode != 0
    assert 'torch.distributed.run' in out and '--nproc-per-node=2' in out
    assert '[rank 0 of 2]' in out and '[rank 1 of 2]' in out
