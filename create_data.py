#!/usr/bin/env python
"""Entry point with the reference's `utils/create_data.py --create_data rand` command line;
see efficient-nerf_amd/create_data.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _pkg  # noqa: E402

_pkg.load()
from efficient_nerf_amd.create_data import main  # noqa: E402

if __name__ == '__main__':
    sys.exit(main())
