"""bench.py's host-side helpers (no GPU): the stamp that says whether the committed profiles belong to the kernel sources
in the tree, the self-launch command of `--gpus N`, and the traffic tool's difference-of-two-runs arithmetic."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_sources_digest_is_stable_and_sees_kernel_edits(tmp_path, monkeypatch):
    d0 = bench.csrc_digest()
    assert d0 == bench.csrc_digest() and len(d0) == 16
    # a profile of a round that recorded no digest: unknown; of a round whose digest matches: current; else stale
    assert bench.profile_is_current(os.path.join(ROOT, 'profiles', 'r03_traffic_config3.json')) is None
    assert bench.profile_is_current(None) is None
    fake_root = tmp_path
    (fake_root / 'profiles').mkdir()
    (fake_root / 'profiles' / 'r77_sources.sha16').write_text(d0 + '\n')
    (fake_root / 'profiles' / 'r78_sources.sha16').write_text('0' * 16 + '\n')
    real_join = os.path.join
    monkeypatch.setattr(bench, 'csrc_digest', lambda: d0)
    monkeypatch.setattr(bench, 'ROOT', str(fake_root))
    assert bench.profile_is_current(real_join('profiles', 'r77_traffic_config3.json')) is True
    assert bench.profile_is_current(real_join('profiles', 'r78_traffic_config3.json')) is False


def test_committed_round_profiles_carry_a_digest():
    """Every round from 4 on records what its profiles were measured on (tools/final_measure.sh)."""
    import glob
    rounds = {os.path.basename(f)[:3] for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic_config*.json'))}
    for r in sorted(rounds):
        if int(r[1:]) >= 4:
            stamp = os.path.join(ROOT, 'profiles', r + '_sources.sha16')
            assert os.path.exists(stamp) and len(open(stamp).read().split()[0]) == 16, r


def test_gpus_n_starts_its_own_ranks_as_a_child_process(monkeypatch):
    """`python bench.py --gpus 4` without WORLD_SIZE: torch.distributed.run with 4 ranks on 127.0.0.1, the same arguments,
    as a CHILD process (subprocess.run, never exec), dmabuf IPC kept in its environment."""
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen['cmd'], seen['env'] = cmd, env

        class R:
            returncode = 7
        return R()

    monkeypatch.setattr(subprocess, 'run', fake_run)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3', '--warmup', '1'])
    assert bench.launch_ranks(4) == 7
    cmd = seen['cmd']
    assert cmd[0] == sys.executable and cmd[1:3] == ['-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and int(cmd[cmd.index('--master-port') + 1]) > 0
    assert cmd[-6:] == ['--gpus', '4', '--steps', '3', '--warmup', '1'] and cmd[-7].endswith('bench.py')
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def _write_pass(d, counter, rows):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, 'x_counter_collection.csv'), 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Kernel_Name', 'Counter_Name', 'Counter_Value'])
        for name, val in rows:
            w.writerow([name, counter, val])


def test_traffic_per_iteration_is_the_difference_of_two_runs(tmp_path):
    """Set-up launches (here: one big launch of kernel A and one small of B in both runs) cancel; bytes per launch is the
    largest dispatch; FETCH_SIZE counts double (gfx950)."""
    A, B = 'void (anonymous namespace)::k_a<float>(int)', 'void (anonymous namespace)::k_b<float>(int)'
    for tag, steps in (('long', 10), ('short', 4)):
        fetch = [(A, 1000.0)] + [(A, 100.0)] * steps + [(B, 5.0)]       # KB
        write = [(A, 500.0)] + [(A, 50.0)] * steps + [(B, 1.0)]
        _write_pass(str(tmp_path / tag / 'FETCH_SIZE'), 'FETCH_SIZE', fetch)
        _write_pass(str(tmp_path / tag / 'WRITE_SIZE'), 'WRITE_SIZE', write)
    out = tmp_path / 't.json'
    subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'traffic_from_pmc.py'), str(tmp_path / 'long'), '3', '10',
                    str(out), str(tmp_path / 'short'), '4'], check=True, capture_output=True)
    d = json.load(open(out))
    per_it = (2 * 100.0 + 50.0) * 1024
    assert abs(d['iteration_bytes'] - per_it) < 1e-6 * per_it               # k_b cancels entirely, k_a's set-up launch too
    assert d['per_kernel'][A]['bytes_per_launch'] == (2 * 1000.0 + 500.0) * 1024 and d['per_kernel'][A]['launches_per_iteration'] == 1.0
    assert d['per_kernel'][B]['bytes_per_iteration'] == 0.0
