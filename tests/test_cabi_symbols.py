"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/r2l_hip.h declares; the ctypes table covers the same set.  No compute calls."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'r2l_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    names = re.findall(r'\b((?:r2l|nerf)_[a-z0-9_]+)\s*\(', src)
    return sorted(set(names))


def test_header_declares_expected_entry_points():
    names = declared_symbols()
    for must in ['r2l_create', 'r2l_load_weights', 'r2l_render', 'r2l_render_rays', 'r2l_destroy',
                 'r2l_sample_embed', 'nerf_create', 'nerf_load_weights', 'nerf_render', 'nerf_raw2outputs',
                 'nerf_sample_pdf', 'nerf_merge_sorted']:
        assert must in names


def test_library_exports_every_declared_symbol(built_lib):
    L = ctypes.CDLL(built_lib)
    for name in declared_symbols():
        assert hasattr(L, name), f'{name} declared in include/r2l_hip.h but not exported'


def test_ctypes_table_matches_header(pkg, built_lib):
    from efficient_nerf_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    L = _lib.lib()
    assert L.r2l_last_error() is not None
    assert L.r2l_device_count() >= 0  # 0 on a CPU-only box: every compute entry point then fails loudly


def test_no_oracle_import_in_product():
    """The product path must not route through the oracle (or any CPU fallback)."""
    pk = os.path.join(ROOT, 'efficient-nerf_amd')
    for dp, _, fs in os.walk(pk):
        for f in fs:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dp, f)).read()
                assert 'oracle' not in txt.replace('the oracle', ''), f
