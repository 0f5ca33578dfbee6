"""Starts the N ranks of a one-node run: `python bench.py --gpus N`, `python main.py --gpus N …`,
`python create_data.py --gpus N …` without torchrun.  The reference has no counterpart (it renders on one GPU,
main.py:473); SURVEY §8(e).

Standard library only, and loaded by file path (`load_launcher` in the entry scripts) so that the parent imports
neither torch nor the package: it never touches the GPU, it only starts fresh child processes of the same script with
the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT), waits for
them, and exits with the first non-zero child code.  A child that fails, or the time limit, takes the other ranks down
(SIGTERM, then SIGKILL); a parent that dies takes its children with it (PR_SET_PDEATHSIG).  Rank 0 owns stdout, the other
ranks' stdout goes to stderr; with `json_only` (bench.py: the driver parses ONE JSON line) only rank 0's lines that start
with `{` reach stdout and whatever else a library prints there (gloo's and RCCL's connection notes) goes to stderr.  Under
torchrun (WORLD_SIZE set) nothing here runs.
"""
import os
import signal
import socket
import subprocess
import sys
import threading
import time


def wants_spawn(n_ranks, environ=None):
    """True when this process was asked for N > 1 ranks and is not itself a rank of a launched job"""
    environ = os.environ if environ is None else environ
    return n_ranks > 1 and 'WORLD_SIZE' not in environ


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    try:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]
    finally:
        s.close()


def _die_with_parent():
    # child side, between fork and exec: SIGKILL when the launcher goes away (PR_SET_PDEATHSIG = 1)
    try:
        import ctypes
        ctypes.CDLL(None).prctl(1, signal.SIGKILL)
    except Exception:
        pass


def rank_env(rank, world, port, environ=None):
    env = dict(os.environ if environ is None else environ)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC only on this pool (dist.py)
    return env


def _stop(procs, grace=10.0):
    for p in procs:
        if p.poll() is None:
            try:
                p.terminate()
            except OSError:
                pass
    t_end = time.monotonic() + grace
    for p in procs:
        while p.poll() is None and time.monotonic() < t_end:
            time.sleep(0.05)
        if p.poll() is None:
            try:
                p.kill()
            except OSError:
                pass
            p.wait()


def _relay_json_lines(pipe):
    for raw in iter(pipe.readline, b''):
        line = raw.decode('utf-8', 'replace')
        dst = sys.stdout if line.lstrip().startswith('{') else sys.stderr
        dst.write(line)
        dst.flush()
    pipe.close()


def spawn_ranks(script, argv, n_ranks, timeout=None, poll=0.1, log=None, json_only=False):
    """Run `python script argv…` as ranks 0 … n_ranks − 1 of one job and return the job's exit code: 0 when every rank
    returned 0, else the first non-zero code seen (a rank killed by a signal: 128 + signal), 124 on the time limit."""
    log = log or (lambda m: print(m, file=sys.stderr, flush=True))
    port = free_port()
    procs = []
    got = {'sig': None}

    def on_signal(signum, _frame):
        got['sig'] = signum

    old = {s: signal.signal(s, on_signal) for s in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    relay = None
    try:
        for r in range(n_ranks):
            out = (subprocess.PIPE if json_only else None) if r == 0 else sys.stderr
            procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=rank_env(r, n_ranks, port),
                                          stdout=out, preexec_fn=_die_with_parent))
        if json_only:
            relay = threading.Thread(target=_relay_json_lines, args=(procs[0].stdout,), daemon=True)
            relay.start()
        t_end = None if not timeout else time.monotonic() + timeout
        left = set(range(n_ranks))
        while left:
            for r in sorted(left):
                c = procs[r].poll()
                if c is None:
                    continue
                left.discard(r)
                if c != 0 and rc == 0:
                    rc = c if c > 0 else 128 - c
                    log('[launch] rank %d exited with code %d: stopping the other %d rank(s)' % (r, c, len(left)))
            if rc or not left:
                break
            if got['sig'] is not None:
                rc = 128 + got['sig']
                log('[launch] signal %d: stopping %d rank(s)' % (got['sig'], len(left)))
                break
            if t_end is not None and time.monotonic() > t_end:
                rc = 124
                log('[launch] time limit of %g s: stopping %d rank(s)' % (timeout, len(left)))
                break
            time.sleep(poll)
    finally:
        _stop(procs)
        if relay is not None:
            relay.join(timeout=10)
        for s, h in old.items():
            signal.signal(s, h)
    return rc
