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
    # the launcher tears the other rank down as soon as one has failed: under load only one of the two may get to print its line
    assert '[rank 0 of 2]' in out or '[rank 1 of 2]' in out


def test_evidence_refuses_a_summary_of_another_binary(tmp_path, monkeypatch):
    """every round-5 PMC summary under profiles/ was stamped with a round-4 commit.  Summaries now carry the build id of the
    library they were measured on and the hash of their kernel's device code; bench.py quotes one only if it matches the library it has
    loaded (same build id, or the same device code of that kernel after a rebuild) - else `traffic: null` and the reason."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    from diffsinger_amd import _lib
    from diffsinger_amd.build import kernel_isa, write_kernel_isa
    cur = _lib.build_id()
    if not kernel_isa(cur):
        write_kernel_isa()
    isa = kernel_isa(cur)
    name = 'k_loop_wino<1, 4>(LoopWinoParams)'
    assert name in isa and len(isa) > 100
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    os.makedirs(tmp_path / 'profiles')
    js = {'kernel_tag': 'k_loop_wino<1, 4>', 'frames': 8192, 'round': 'rX', 'hbm_bytes_per_launch': {'total': 7.0, 'note': 'n'}, 'build_id': cur}
    put = lambda: json.dump(js, open(tmp_path / 'profiles' / 'loop_pmc.json', 'w'))
    put()
    t, src = bench.pmc_traffic('k_loop_wino<1, 4>', 8192)
    assert t == 7.0 and 'round rX' in src and cur[:12] in src
    assert bench.pmc_traffic('k_loop_wino<1, 4>', 4096)[0] is None and bench.pmc_traffic('k_loop<1>', 8192)[0] is None      # another shape / kernel
    js['build_id'] = 'deadbeef' * 8
    put()
    t, src = bench.pmc_traffic('k_loop_wino<1, 4>', 8192)
    assert t is None and src.startswith('stale:') and 'deadbeefdead' in src
    js.update(kernel_isa_name=name, kernel_isa=isa[name])           # the library was rebuilt for an edit elsewhere: this kernel's code is the same
    put()
    t, src = bench.pmc_traffic('k_loop_wino<1, 4>', 8192)
    assert t == 7.0 and 'device code' in src
    js['kernel_isa'] = '0' * 40
    put()
    assert bench.pmc_traffic('k_loop_wino<1, 4>', 8192)[0] is None
    os.remove(tmp_path / 'profiles' / 'loop_pmc.json')
    assert bench.pmc_traffic('k_loop_wino<1, 4>', 8192) == (None, 'no profiles/loop_pmc.json')


def test_the_n1_line_has_rows_offshape_and_inservice_noise_legs():
    """the driver-run line carries compact figures for the rows around the path, the off-shapes and the in-service noise path.
    CPU: the legs exist, are wired into the N = 1 line, can be switched off, and their FLOP accounting is what DESIGN section 10 says."""
    sys.path.insert(0, ROOT)
    import bench
    r = _run('--help')
    assert '--no-extras' in r.stdout
    src = open(os.path.join(ROOT, 'bench.py')).read()
    for key in ("res['rows'] = quick_rows(", "res['offshape'] = offshape(", "res['inservice_noise'] = inservice_noise("):
        assert key in src
    assert bench.F_TRAIN_EXEC == 3 * (26_427_392 - 15_728_640 // 3) and bench.F_CONV_LAYERS == 15_728_640
    assert bench.WEIGHT_BYTES == 60_345_664 and bench.F_EVAL_EXEC_WINO == 15_810_560
    assert not [f for f in os.listdir(os.path.join(ROOT, 'tools')) if f.startswith('gpu_') and f.endswith('.sh')], 'ONE GPU script: tools/gpu.sh'
