"""efficient-nerf_amd: MI355X (gfx950) native renderer for the R2L / NeRF-teacher
ray-batched inference path of MingSun-Tse/Efficient-NeRF.

The package directory name carries a hyphen (it is the repo's required layout); import it
through ``_pkg.load()`` at the repo root, which registers it as ``efficient_nerf_amd``.

Everything computational lives in ``libr2l_hip.so`` (hand-written HIP, C-ABI declared in
``include/r2l_hip.h``).  The Python here mirrors the reference's call surface for the
path (same names, argument meaning, error behaviour) and is plumbing only: it hands
PyTorch-ROCm device pointers and the current HIP stream to the library.  There is no CPU
or eager fallback: if the library is missing or no gfx950 device is visible, calls raise.
"""
from ._lib import lib, R2LError, PREC_FP16X3, PREC_FP16X1, PREC_FP16_FP8, PREC_FP16_E4M3, PREC_FP16X3_ASM, PREC_FP16_SPLIT, PREC_FP16_SPLIT8, PRECISIONS  # noqa: F401
from .r2l import PointSampler, PositionalEmbedder, R2LEngine, NeRF_v3_2, render_func, PREC_NAMES  # noqa: F401
from .teacher import NeRFEngine, get_rays, ndc_rays, raw2outputs, sample_pdf, merge_sorted, render  # noqa: F401
