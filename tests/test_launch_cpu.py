"""CPU: the rank launcher behind `python bench.py --gpus N` / `main.py --gpus N` / `create_data.py --gpus N`
(efficient-nerf_amd/launch.py; no reference counterpart: main.py:473 there renders on one GPU).  Fresh child processes with
the torchrun environment, first non-zero exit code wins, a failing rank or the time limit stops the others, and the launcher
itself imports neither torch nor the package."""
import importlib.util
import json
import os
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch():
    spec = importlib.util.spec_from_file_location('r2l_launch', os.path.join(ROOT, 'efficient-nerf_amd', 'launch.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _script(tmp_path, body):
    p = tmp_path / 'rank.py'
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_wants_spawn_only_outside_a_launched_job():
    L = _launch()
    assert L.wants_spawn(2, {}) and L.wants_spawn(8, {'PATH': ''})
    assert not L.wants_spawn(1, {}) and not L.wants_spawn(0, {})
    assert not L.wants_spawn(8, {'WORLD_SIZE': '8'})          # a rank under torchrun or under this launcher


def test_ranks_get_the_torchrun_environment_and_rank0_owns_stdout(tmp_path):
    script = _script(tmp_path, '''
        import json, os, sys
        print(json.dumps({k: os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
                         | {'argv': sys.argv[1:]}), flush=True)
    ''')
    # run the launcher in its own interpreter so that the ranks' inherited stdout / stderr can be captured
    code = ('import importlib.util, sys\n'
            'spec = importlib.util.spec_from_file_location("l", %r)\n'
            'm = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n'
            'rc = m.spawn_ranks(%r, ["--x", "1"], 3, timeout=60)\n'
            'assert "torch" not in sys.modules and "efficient_nerf_amd" not in sys.modules\n'
            'sys.exit(rc)\n') % (os.path.join(ROOT, 'efficient-nerf_amd', 'launch.py'), script)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith('{')]
    err = [json.loads(ln) for ln in r.stderr.splitlines() if ln.startswith('{')]
    assert len(out) == 1 and out[0]['RANK'] == '0'                       # one line on stdout: rank 0's
    assert sorted(e['RANK'] for e in err) == ['1', '2']                  # the other ranks' stdout went to stderr
    for e in out + err:
        assert e['WORLD_SIZE'] == e['LOCAL_WORLD_SIZE'] == '3' and e['LOCAL_RANK'] == e['RANK'] and e['MASTER_ADDR'] == '127.0.0.1'
        assert e['argv'] == ['--x', '1'] and e['MASTER_PORT'] == out[0]['MASTER_PORT'] and int(e['MASTER_PORT']) > 0


def test_eight_ranks(tmp_path):
    """the driver's N = 8: eight children, ranks 0..7 each exactly once, one port, rank 0's line alone on stdout"""
    script = _script(tmp_path, '''
        import json, os
        print(json.dumps({k: os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}), flush=True)
    ''')
    code = ('import importlib.util, sys\n'
            'spec = importlib.util.spec_from_file_location("l", %r)\n'
            'm = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n'
            'sys.exit(m.spawn_ranks(%r, [], 8, timeout=60))\n') % (os.path.join(ROOT, 'efficient-nerf_amd', 'launch.py'), script)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith('{')]
    err = [json.loads(ln) for ln in r.stderr.splitlines() if ln.startswith('{')]
    assert [e['RANK'] for e in out] == ['0'] and sorted(int(e['RANK']) for e in err) == list(range(1, 8))
    assert {e['WORLD_SIZE'] for e in out + err} == {'8'} and len({e['MASTER_PORT'] for e in out + err}) == 1
    assert all(e['LOCAL_RANK'] == e['RANK'] for e in out + err)


def test_json_only_keeps_library_chatter_off_stdout(tmp_path):
    """bench.py's launcher: the driver parses ONE JSON line from stdout; what gloo / RCCL print there while connecting
    ('[Gloo] Rank 0 is connected to ...' on the GPU box) is relayed to stderr"""
    script = _script(tmp_path, '''
        import os
        print('[Gloo] Rank %s is connected to 1 peer ranks.' % os.environ['RANK'], flush=True)
        print('{"rank": %s}' % os.environ['RANK'], flush=True)
    ''')
    code = ('import importlib.util, sys\n'
            'spec = importlib.util.spec_from_file_location("l", %r)\n'
            'm = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n'
            'sys.exit(m.spawn_ranks(%r, [], 2, timeout=60, json_only=True))\n') % (os.path.join(ROOT, 'efficient-nerf_amd', 'launch.py'), script)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == ['{"rank": 0}'], r.stdout
    assert '[Gloo] Rank 0' in r.stderr and '[Gloo] Rank 1' in r.stderr and '{"rank": 1}' in r.stderr


def test_first_failing_rank_stops_the_others_and_sets_the_exit_code(tmp_path):
    script = _script(tmp_path, '''
        import os, sys, time
        if os.environ['RANK'] == '1':
            time.sleep(0.3)
            sys.exit(3)
        time.sleep(120)
    ''')
    L = _launch()
    msgs = []
    t0 = time.monotonic()
    rc = L.spawn_ranks(script, [], 3, timeout=100, log=msgs.append)
    assert rc == 3 and time.monotonic() - t0 < 30
    assert any('rank 1 exited with code 3' in m for m in msgs)


def test_time_limit_stops_every_rank(tmp_path):
    script = _script(tmp_path, '''
        import signal, time
        signal.signal(signal.SIGTERM, signal.SIG_IGN)     # a rank that ignores the first signal is still killed
        time.sleep(120)
    ''')
    L = _launch()
    t0 = time.monotonic()
    old, L._stop.__defaults__ = L._stop.__defaults__, (1.0,)
    try:
        rc = L.spawn_ranks(script, [], 2, timeout=1.0, log=lambda m: None)
    finally:
        L._stop.__defaults__ = old
    assert rc == 124 and time.monotonic() - t0 < 30


def test_ranks_die_with_a_killed_launcher(tmp_path):
    pidfile = tmp_path / 'pids'
    script = _script(tmp_path, '''
        import os, time
        open(%r, 'a').write(str(os.getpid()) + '\\n')
        time.sleep(120)
    ''' % str(pidfile))
    code = ('import importlib.util, sys\n'
            'spec = importlib.util.spec_from_file_location("l", %r)\n'
            'm = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n'
            'sys.exit(m.spawn_ranks(%r, [], 2, timeout=100))\n') % (os.path.join(ROOT, 'efficient-nerf_amd', 'launch.py'), script)
    p = subprocess.Popen([sys.executable, '-c', code])
    t_end = time.monotonic() + 30
    while time.monotonic() < t_end and (not pidfile.exists() or len(pidfile.read_text().split()) < 2):
        time.sleep(0.05)
    pids = [int(x) for x in pidfile.read_text().split()]
    assert len(pids) == 2
    p.kill()                      # SIGKILL: no handler runs; PR_SET_PDEATHSIG takes the ranks down
    p.wait()
    t_end = time.monotonic() + 10
    alive = pids
    while alive and time.monotonic() < t_end:
        alive = [q for q in alive if os.path.exists('/proc/%d' % q) and 'Z' not in open('/proc/%d/stat' % q).read().split(')')[-1].split()[0]]
        time.sleep(0.05)
    assert not alive, alive


def test_entry_scripts_launch_before_importing_torch():
    """bench.py / main.py / create_data.py hand over to the launcher before `import torch`: with a python whose torch import
    fails, `--gpus 2` still reaches the launcher (and the ranks, which then fail on the import: exit code != 0, no hang)"""
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, 'tests', '_no_torch'))
    os.makedirs(os.path.join(ROOT, 'tests', '_no_torch'), exist_ok=True)
    with open(os.path.join(ROOT, 'tests', '_no_torch', 'torch.py'), 'w') as f:
        f.write('import os, sys\nsys.stderr.write("TORCH-IMPORT rank=%s\\n" % os.environ.get("RANK"))\nraise ImportError("no torch here")\n')
    try:
        for script, extra in (('bench.py', ['--steps', '1', '--warmup', '0']), ('main.py', []), ('create_data.py', [])):
            env.pop('WORLD_SIZE', None)
            r = subprocess.run([sys.executable, os.path.join(ROOT, script), '--gpus', '2'] + extra, env=env, capture_output=True,
                               text=True, timeout=120)
            assert r.returncode != 0
            assert 'TORCH-IMPORT rank=None' not in r.stderr, (script, r.stderr[-800:])      # the launcher never imported it
            assert 'TORCH-IMPORT rank=0' in r.stderr or 'TORCH-IMPORT rank=1' in r.stderr, (script, r.stderr[-800:])
            assert '[launch] rank' in r.stderr
    finally:
        import shutil
        shutil.rmtree(os.path.join(ROOT, 'tests', '_no_torch'), ignore_errors=True)
