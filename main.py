#!/usr/bin/env python
"""Entry point with the reference's command line for the render-only path
(`--render_only [--render_test]`, `--config configs/*.txt`, `--pretrained_ckpt X.tar`);
see efficient-nerf_amd/frontend.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _pkg  # noqa: E402

_pkg.load()
from efficient_nerf_amd.frontend import main  # noqa: E402

if __name__ == '__main__':
    sys.exit(main())
