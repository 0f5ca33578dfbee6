#!/usr/bin/env python
"""Entry point with the reference's `utils/create_data.py --create_data rand` command line;
see efficient-nerf_amd/create_data.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _maybe_launch():
    """`--gpus N` with N > 1 and no torchrun environment: start the N ranks (fresh child processes of this script) before
    torch or the package is imported, so that the launcher itself never touches the GPU (efficient-nerf_amd/launch.py)"""
    import argparse
    import importlib.util
    ap = argparse.ArgumentParser(add_help=False, allow_abbrev=False)
    ap.add_argument('--gpus', type=int, default=0)
    ap.add_argument('--launch_timeout', type=float, default=0.)
    own, _ = ap.parse_known_args(sys.argv[1:])
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location('r2l_launch', os.path.join(here, 'efficient-nerf_amd', 'launch.py'))
    launch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(launch)
    if launch.wants_spawn(own.gpus):
        sys.exit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], own.gpus, timeout=own.launch_timeout or None))


if __name__ == '__main__':
    _maybe_launch()
import _pkg  # noqa: E402

_pkg.load()
from efficient_nerf_amd.create_data import main  # noqa: E402

if __name__ == '__main__':
    sys.exit(main())
