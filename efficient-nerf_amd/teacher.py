"""Host-side mirror of the reference's NeRF-teacher call surface, backed by libr2l_hip.so.

Reference interface mirrored (MingSun-Tse/Efficient-NeRF):
  get_rays                 utils/run_nerf_raybased_helpers.py:231-257
  raw2outputs              main.py:556-621  (== helpers:77-144, model/nerf_raybased.py:226-295)
  sample_pdf               utils/run_nerf_raybased_helpers.py:283-330
  render / render_rays     main.py:107-186, 624-756   (and utils/create_data.py:80-176, 405-544)
  create_nerf's networks   main.py:425-453 (NeRF D=8 W=256, skips=[4], use_viewdirs)

Same names and argument meaning; tensors are float32 on the HIP device; errors raise
R2LError.  No CPU fallback.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import R2LError, PREC_FP16X1, PREC_FP16X3, check, current_stream, dptr, lib
from .r2l import _c2w_host, _dev


def _f32(t, device):
    return torch.as_tensor(t).to(device=device, dtype=torch.float32).contiguous()


def get_rays(H, W, focal, c2w, trans_origin='', focal_scale=1, rows=None, device=None):
    """helpers:231-257 (trans_origin='' only).  Returns rays_o, rays_d [rows, W, 3]."""
    if trans_origin:
        raise NotImplementedError('trans_origin variants are ablation paths (out of scope)')
    dev = _dev(device)
    focal = float(focal) * focal_scale
    r0, r1 = (0, H) if rows is None else rows
    c = _c2w_host(c2w)
    n = (r1 - r0) * W
    ro = torch.empty((n, 3), dtype=torch.float32, device=dev)
    rd = torch.empty((n, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib().nerf_get_rays(int(H), int(W), focal, C.c_void_p(c.data_ptr()), r0, r1, dptr(ro), dptr(rd),
                                  current_stream()))
    return ro.view(r1 - r0, W, 3), rd.view(r1 - r0, W, 3)


def _raw_noise(shape, raw_noise_std, pytest, dev):
    """main.py:592-598: torch.randn * std; with pytest the numpy stream the reference switches to"""
    if pytest:
        np.random.seed(0)
        return torch.Tensor(np.random.rand(*list(shape)) * raw_noise_std).to(dev)
    return torch.randn(shape, device=dev) * raw_noise_std


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False, verbose=False):
    """main.py:556-621.  raw [n,S,4], z_vals [n,S], rays_d [n,3] ->
    rgb_map [n,3], disp_map [n], acc_map [n], weights [n,S], depth_map [n]."""
    dev = raw.device
    raw, rays_d = _f32(raw, dev), _f32(rays_d, dev)
    n, S = raw.shape[0], raw.shape[1]
    noise = _f32(_raw_noise((n, S), raw_noise_std, pytest, dev), dev) if raw_noise_std > 0. else None
    z_vals = _f32(z_vals.expand(n, S) if z_vals.dim() == 2 else z_vals, dev)
    rgb = torch.empty((n, 3), dtype=torch.float32, device=dev)
    disp, acc, depth = (torch.empty((n,), dtype=torch.float32, device=dev) for _ in range(3))
    weights = torch.empty((n, S), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib().nerf_raw2outputs_noise(dptr(raw), dptr(z_vals), dptr(rays_d), dptr(noise), n, S, int(bool(white_bkgd)),
                                           dptr(rgb), dptr(disp), dptr(acc), dptr(weights), dptr(depth), current_stream()))
    return rgb, disp, acc, weights, depth


def _sample_pdf_u(n, N_samples, det, pytest, dev):
    """helpers:293-307: the uniforms sample_pdf inverts the cdf at, drawn the way the reference draws them.
    Returns (u, per_ray)."""
    if pytest:
        np.random.seed(0)
        if det:
            return torch.Tensor(np.linspace(0., 1., N_samples)).to(dev), False   # np.linspace in float64, then float32
        return torch.Tensor(np.random.rand(n, N_samples)).to(dev), True
    if det:
        return torch.linspace(0., 1., steps=N_samples).to(dev), False            # evaluated on the host as the reference does
    return torch.rand((n, N_samples), device=dev), True


def sample_pdf(bins, weights, N_samples, det=False, pytest=False, u=None, taps=False):
    """helpers:283-330.  bins [n,B], weights [n,B-1] -> samples [n,N_samples].  `u` (optional, [N] or
    [n,N]) replaces the draw; taps=True also returns (cdf [n,B], inds [n,N] int32 = searchsorted(cdf, u, right))."""
    dev = bins.device
    bins, weights = _f32(bins, dev), _f32(weights, dev)
    n, B = bins.shape
    if weights.shape != (n, B - 1):
        raise R2LError(f'weights must be [n, {B - 1}]; got {tuple(weights.shape)}')
    if u is None:
        u, per_ray = _sample_pdf_u(n, N_samples, det, pytest, dev)
    else:
        per_ray = u.dim() == 2
    u = _f32(u, dev)
    if u.shape[-1] != N_samples or (per_ray and u.shape[0] != n):
        raise R2LError(f'u must be [{N_samples}] or [{n}, {N_samples}]; got {tuple(u.shape)}')
    out = torch.empty((n, N_samples), dtype=torch.float32, device=dev)
    cdf = torch.empty((n, B), dtype=torch.float32, device=dev) if taps else None
    inds = torch.empty((n, N_samples), dtype=torch.int32, device=dev) if taps else None
    with torch.cuda.device(dev):
        check(lib().nerf_sample_pdf_ex(dptr(bins), dptr(weights), n, B, dptr(u), int(per_ray), N_samples, dptr(out),
                                       dptr(cdf), None if inds is None else C.c_void_p(inds.data_ptr()), current_stream()))
    return (out, cdf, inds) if taps else out


def merge_sorted(z_vals, z_samples):
    """main.py:730-732: torch.sort(torch.cat([z_vals, z_samples], -1), -1)[0] for rows that
    are each already ascending."""
    dev = z_samples.device
    z_samples = _f32(z_samples, dev)
    n = z_samples.shape[0]
    z_vals = _f32(z_vals.expand(n, z_vals.shape[-1]), dev)
    out = torch.empty((n, z_vals.shape[1] + z_samples.shape[1]), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib().nerf_merge_sorted(dptr(z_vals), z_vals.shape[1], dptr(z_samples), z_samples.shape[1], n,
                                      dptr(out), current_stream()))
    return out


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """utils/run_nerf_raybased_helpers.py:260-279 on the GPU: [..., 3] rays -> projected rays."""
    o = rays_o.reshape(-1, 3).to(rays_o.device if rays_o.is_cuda else _dev(None), torch.float32).contiguous()
    d = rays_d.reshape(-1, 3).to(o.device, torch.float32).contiguous()
    oo, dd = torch.empty_like(o), torch.empty_like(d)
    with torch.cuda.device(o.device):
        check(lib().nerf_ndc_rays(int(H), int(W), float(focal), float(near), dptr(o), dptr(d), o.shape[0], dptr(oo),
                                  dptr(dd), current_stream()))
    return oo.view(rays_o.shape), dd.view(rays_d.shape)


def rebalanced_state(sd, maxima, keep=2, target=8.0):
    """An exact reparametrisation of NeRF(D=8, W=256, use_viewdirs) (model/nerf_raybased.py:377-401) that brings the inputs of its layers
    to the range the bf6 layer chain converts them with (fixed exponents: |a| <= 14 for the values, <= 8 for their fp16 residuals;
    csrc/gen/isa.py ACT_EXP / RES_EXP; which `target` serves that best is measured: NeRFEngine.REBALANCE_TARGET): relu is positively
    homogeneous and feature_linear is linear, so the output of trunk layer i
    divided by 2^s_i, feature by 2^s_f and the view layer's output by 2^s_v -- with the consumers' weight columns multiplied back -- is the
    same function, and every factor a power of two makes it the same function in float32 arithmetic too (no rounding: each weight and
    bias is scaled exactly).  `maxima`: {'h0' .. 'h7', 'feature', 'views': largest |value| measured on the job's own points
    (NeRFEngine.fine_activation_maxima)}; s = ceil(log2(max / target)) so that max / 2^s lies in (target / 2, target]; the outputs of the
    first `keep` trunk layers stay as they are (R2L_PREC_FP16_MIX runs the layers that read them in three fp16 passes).  Returns (new
    state dict, {name: s})."""
    import math
    sd = {(k[7:] if k.startswith('module.') else k): v.detach().to('cpu', torch.float32).clone() for k, v in sd.items()}
    sh = lambda m: 0 if not (m > 0 and math.isfinite(m)) else int(math.ceil(math.log2(m / target)))
    s = {f'h{i}': (0 if i < keep else sh(float(maxima[f'h{i}']))) for i in range(8)}
    s['feature'], s['views'] = sh(float(maxima['feature'])), sh(float(maxima['views']))
    for i in range(8):
        prev = s[f'h{i - 1}'] if i else 0
        W = sd[f'pts_linears.{i}.weight']
        if i == 5:          # cat([input_pts, h]) (:385): the first 63 columns read the embedding
            W[:, :63] *= 2.0 ** (-s['h5'])
            W[:, 63:] *= 2.0 ** (prev - s['h5'])
        elif i == 0:
            W *= 2.0 ** (-s['h0'])
        else:
            W *= 2.0 ** (prev - s[f'h{i}'])
        sd[f'pts_linears.{i}.bias'] *= 2.0 ** (-s[f'h{i}'])
    sd['alpha_linear.weight'] *= 2.0 ** s['h7']
    sd['feature_linear.weight'] *= 2.0 ** (s['h7'] - s['feature'])
    sd['feature_linear.bias'] *= 2.0 ** (-s['feature'])
    sd['views_linears.0.weight'][:, :256] *= 2.0 ** (s['feature'] - s['views'])
    sd['views_linears.0.weight'][:, 256:] *= 2.0 ** (-s['views'])
    sd['views_linears.0.bias'] *= 2.0 ** (-s['views'])
    sd['rgb_linear.weight'] *= 2.0 ** s['views']
    return sd, s


class NeRFEngine:
    """One nerf_ctx: coarse + fine NeRF(D=8, W=256, use_viewdirs) and the render_rays
    pipeline (include/r2l_hip.h)."""

    STATE_NAMES = [f'pts_linears.{i}.{k}' for i in range(8) for k in ('weight', 'bias')] + [
        'views_linears.0.weight', 'views_linears.0.bias', 'feature_linear.weight', 'feature_linear.bias',
        'alpha_linear.weight', 'alpha_linear.bias', 'rgb_linear.weight', 'rgb_linear.bias']

    def __init__(self, H, W, focal, near=2., far=6., N_samples=64, N_importance=128, multires=10, multires_views=4,
                 white_bkgd=True, precision=PREC_FP16X3, device=None, z_coarse=None, u=None, ndc=False, lindisp=False):
        """ndc: render(..., ndc=True) of the reference (forward-facing scenes; pass near=0, far=1 as
        main.py:917-918 does); lindisp: coarse depths linear in inverse depth (main.py:679-680)."""
        self.device = _dev(device)
        self.H, self.W, self.focal = int(H), int(W), float(focal)
        self.N_samples, self.N_importance = int(N_samples), int(N_importance)
        self.near, self.far = float(near), float(far)
        self.white_bkgd = bool(white_bkgd)
        self._ctx = C.c_void_p()
        from ._lib import PREC_FP16_MIX, PREC_FP16X3_ASM
        mix = int(precision) == PREC_FP16_MIX      # a mode of the fine network: created in three passes, the pair is set below
        with torch.cuda.device(self.device):
            check(lib().nerf_create(C.byref(self._ctx), self.H, self.W, self.focal, float(near), float(far),
                                    self.N_samples, self.N_importance, int(multires), int(multires_views),
                                    int(bool(white_bkgd)), PREC_FP16X3_ASM if mix else int(precision)))
        self.precision = self.precision_coarse = PREC_FP16X3_ASM if mix else int(precision)
        if mix:
            self.set_precision(PREC_FP16_MIX)
        # main.py:676-678 / helpers:293 evaluated with the host's torch, as the reference does
        if z_coarse is None:
            t_vals = torch.linspace(0., 1., steps=self.N_samples)
            if not lindisp:
                z_coarse = float(near) * (1. - t_vals) + float(far) * (t_vals)
            else:
                nt, ft = torch.tensor(float(near)), torch.tensor(float(far))  # tensor arithmetic as in main.py:679-680
                z_coarse = 1. / (1. / nt * (1. - t_vals) + 1. / ft * (t_vals))      # (near = 0 gives inf / nan there too)
        if u is None:
            u = torch.linspace(0., 1., steps=self.N_importance)
        self.set_sampling(z_coarse, u)
        self.ndc = bool(ndc)
        if ndc:
            with torch.cuda.device(self.device):
                check(lib().nerf_set_ndc(self._ctx, 1, 1.0))  # ndc_rays(H, W, focal, 1., ...)  (main.py:162)

    def set_sampling(self, z_coarse, u):
        z = torch.as_tensor(z_coarse).detach().to('cpu', torch.float32).contiguous()
        uu = torch.as_tensor(u).detach().to('cpu', torch.float32).contiguous()
        with torch.cuda.device(self.device):
            check(lib().nerf_set_sampling(self._ctx, C.c_void_p(z.data_ptr()), z.numel(), C.c_void_p(uu.data_ptr()),
                                          uu.numel()))
        self.z_coarse, self.u = z, uu

    def close(self):
        if getattr(self, '_ctx', None) and self._ctx.value and _lib._lib is not None:
            lib().nerf_destroy(self._ctx)
            self._ctx = C.c_void_p()

    __del__ = close

    def _load(self, which, state_dict):
        sd = {(k[7:] if k.startswith('module.') else k): v for k, v in state_dict.items()}
        missing = [n for n in self.STATE_NAMES if n not in sd]
        if missing:
            raise R2LError(f'teacher state_dict lacks {missing[:3]} ...')
        keep, arr = _lib.host_ptrs([sd[n] for n in self.STATE_NAMES])
        want = {'pts_linears.0.weight': (256, 63), 'pts_linears.5.weight': (256, 319),
                'views_linears.0.weight': (128, 283), 'alpha_linear.weight': (1, 256), 'rgb_linear.weight': (3, 128)}
        for n, t in zip(self.STATE_NAMES, keep):
            exp = want.get(n)
            if exp is None:
                exp = (256, 256) if n.endswith('weight') else ((128,) if n.startswith('views') else
                                                                 (1,) if n.startswith('alpha') else
                                                                 (3,) if n.startswith('rgb') else (256,))
            if tuple(t.shape) != exp:
                raise R2LError(f'{n}: shape {tuple(t.shape)} != {exp}')
        with torch.cuda.device(self.device):
            check(lib().nerf_load_weights(self._ctx, which, arr, len(keep)))

    def load_state_dicts(self, network_fn_state_dict, network_fine_state_dict):
        """The `.tar`'s 'network_fn_state_dict' / 'network_fine_state_dict' (main.py:1516-1542)."""
        self._load(0, network_fn_state_dict)
        self._load(1, network_fine_state_dict)
        self._fine_state = network_fine_state_dict        # a reference: rebalance_fine re-packs the fine network from it
        self.fine_shifts = None
        return self

    def fine_activation_maxima(self, rays_o, rays_d, z_vals, max_points=1 << 18):
        """largest |value| of every hidden activation of the FINE network (outputs of pts_linears.0-7, feature_linear, views_linears.0)
        on up to `max_points` of the points rays_o + rays_d z_vals ([n, S]: the sample positions a render of these rays used), evaluated
        layer by layer with the library's fp32 layer kernels (generic.Linear = r2l_linear_forward; the embedding by nerf_embed): what
        rebalanced_state needs.  Synchronous; once per weight load."""
        from .generic import Linear, _view
        sd = {(k[7:] if k.startswith('module.') else k): v for k, v in self._fine_state.items()}
        dev = self.device
        n, S = z_vals.shape
        step = max(1, (n * S) // int(max_points))
        ro, rd, z = rays_o.reshape(-1, 3)[::step].contiguous(), rays_d.reshape(-1, 3)[::step].contiguous(), z_vals[::step].contiguous()
        n = ro.shape[0]
        m = n * S
        pts = torch.empty((m, 3), dtype=torch.float32, device=dev)
        cat = torch.empty((m, 63 + 256), dtype=torch.float32, device=dev)
        views = torch.empty((m, 256 + 27), dtype=torch.float32, device=dev)
        work = [torch.empty((m, 256), dtype=torch.float32, device=dev) for _ in range(2)]
        lin = {k: Linear(sd[k + '.weight'], sd[k + '.bias'], dev) for k in [f'pts_linears.{i}' for i in range(8)] + ['feature_linear', 'views_linears.0']}
        mx = {}
        with torch.cuda.device(dev):
            check(lib().r2l_sample_points(dptr(ro), dptr(rd), n, dptr(z), S, 1, dptr(pts), current_stream()))
            op, ldo = _view(cat[:, :63])
            check(lib().nerf_embed(dptr(pts), 3, m, 3, 10, op, ldo, current_stream()))
            dirs = (rd / torch.norm(rd, dim=-1, keepdim=True))[:, None, :].expand(n, S, 3).reshape(m, 3).contiguous()      # main.py:148-157, 76-77
            op, ldo = _view(views[:, 256:])
            check(lib().nerf_embed(dptr(dirs), 3, m, 3, 4, op, ldo, current_stream()))
            x, pp = cat[:, :63], 0
            for i in range(8):
                y = cat[:, 63:] if i == 4 else work[pp]         # h = cat([input_pts, h]) behind layer 4 (model/nerf_raybased.py:385)
                if i != 4:
                    pp ^= 1
                lin[f'pts_linears.{i}'](x, y, act='relu')
                mx[f'h{i}'] = float(y.abs().max())
                x = cat if i == 4 else y
            lin['feature_linear'](x, views[:, :256])
            mx['feature'] = float(views[:, :256].abs().max())
            hv = work[pp][:, :128]
            lin['views_linears.0'](views, hv, act='relu')
            mx['views'] = float(hv.abs().max())
        for l in lin.values():
            l.close()
        return mx

    # The largest value of every rescaled activation lands in (REBALANCE_TARGET / 2, REBALANCE_TARGET].  Measured, not derived
    # (tools/rebalance_sweep.py, profiles/r06_rebalance_sweep.txt: targets 2 .. 80 on two trained teachers, fp16_mix against three passes
    # over three whole frames): the worst ray falls from 1.1e-4 / 2.2e-4 at target 2 to a flat minimum at 28 .. 40 (2.4e-5 / 4.4-5.0e-5;
    # 8, the chain's nominal range, gives 2.9e-5 / 6.4e-5) and rises again from 56.  Trained activations are heavy-tailed: at 32 the bulk sits
    # two binades further inside bf6's seven-binade normal range, and the few values beyond the conversion's largest number saturate in the
    # CORRECTION terms only (the fp16 pass carries them whole) -- an error of the size of a bf6 rounding step on those terms.
    REBALANCE_TARGET = 32.0

    def rebalance_fine(self, rays_o, rays_d, z_vals, target=None):
        """Re-pack the fine network from an exact power-of-two reparametrisation whose hidden activations suit the bf6 chain's fixed
        conversion exponents on these points (rebalanced_state, REBALANCE_TARGET): the calibration of R2L_PREC_FP16_MIX / _FP16_FP8 without
        a kernel change.  The function the network computes is unchanged bit for bit in float32; three passes render the same image.
        Returns the shifts {activation: s} (also kept in `fine_shifts`)."""
        mx = self.fine_activation_maxima(rays_o, rays_d, z_vals)
        sd, sh = rebalanced_state(self._fine_state, mx, target=self.REBALANCE_TARGET if target is None else float(target))
        try:
            self._load(1, sd)
        except R2LError as e:          # a rescaled layer outside what the chain's weight split packs (max|w| beyond 2^-12 .. 2^6): as loaded
            self._load(1, self._fine_state)
            self.fine_shifts, self.fine_maxima, self.rebalance_note = None, mx, str(e)
            return None
        self.fine_shifts, self.fine_maxima = sh, mx
        return sh

    def set_precision(self, precision):
        from ._lib import PREC_FP16_MIX, PREC_FP16X3_ASM
        if int(precision) == PREC_FP16_MIX:       # a mode of the FINE network: the coarse one, which steers sample_pdf, runs three passes
            return self.set_precision_pair(PREC_FP16X3_ASM, PREC_FP16_MIX)
        with torch.cuda.device(self.device):
            check(lib().nerf_set_precision(self._ctx, int(precision)))
        self.precision = self.precision_coarse = int(precision)

    def set_skip_rgb0(self, on=True):
        """nerf_set_skip_rgb0 (include/r2l_hip.h): the render pipeline's coarse pass without its view branch whenever the coarse network runs
        fp16x3_asm -- rgb0 is not computed (render(..., extras=True) then returns no 'rgb0'), every other output is bit for bit what it
        was.  For callers that drop render()'s extras, as main.py:277-282 and utils/create_data.py:824-831 do (frontend.render_path and
        create_data.create_rand switch it on)."""
        with torch.cuda.device(self.device):
            check(lib().nerf_set_skip_rgb0(self._ctx, int(bool(on))))
        self.skip_rgb0 = bool(on)
        return self

    def _rgb0_skipped(self):
        from ._lib import PREC_FP16X3_ASM
        return getattr(self, 'skip_rgb0', False) and self.precision_coarse == PREC_FP16X3_ASM

    def set_precision_pair(self, coarse, fine):
        """one mode per network (include/r2l_hip.h nerf_set_precision_pair): the coarse pass decides where the fine samples go
        (sample_pdf), which on rays that graze an object depends on weights at the 1e-4 level -- a trained teacher needs it at
        fp32 grade (fp16x3) whatever the fine pass runs in"""
        with torch.cuda.device(self.device):
            check(lib().nerf_set_precision_pair(self._ctx, int(coarse), int(fine)))
        self.precision, self.precision_coarse = int(fine), int(coarse)

    #: `--precision auto`: largest difference of rgb / acc (and depth, below) from fp16x3 on probes of the caller's own rays that still
    #: selects a faster mode.  The contract is 1e-4 against the reference on every ray; fp16x3 is within 2e-7 of it.  A probe is 2.5 %
    #: of a frame's rays, so the limits keep a factor of three (single fp16 pass) / five (bf6 chain) to the contract:
    #: measured over whole 400 x 400 frames of three poses (tools/teacher_x1_error.py, profiles/r04_teacher_x1.txt) the single
    #: fp16 pass is 0.6-1.6e-5 from fp16x3 on the synthetic teachers (3.6e-5 with the trunk weights doubled), the bf6 chain 1e-6.
    AUTO_MAX_DIFF = 2e-5           # fp16_fp8: the layer chain's bf6 terms run under FIXED activation exponents (no calibration)
    AUTO_MAX_DIFF_X1 = 3e-5        # fp16x1: one fp16 pass, no correction terms
    #: depth_map = sum(weights * z) carries the weights' error times z: its limit is the rgb limit times max(1, far) (scene units).
    #: disp_map = 1 / max(1e-10, depth / acc) is ill-conditioned on empty rays in the reference itself and is not compared.
    WATCH_KEYS = ('rgb_map', 'acc_map', 'depth_map')
    LADDER = ('fp16x1', 'fp16_fp8', 'fp16_mix', 'fp16x3_asm', 'fp16x3')
    #: fp16_mix (round 6): the coarse network in fp16x3_asm, the FINE network on the bf6 chain with its first two trunk layers in three
    #: passes (R2L_PREC_FP16_MIX: ~2.0 instead of 3.0 pass-equivalents on 75 % of the points).  The rung for trained teachers: their sharp
    #: density tail amplifies what the early layers get wrong, so bf6 terms in every layer leave the fine pass 1.2-1.6e-4 from three passes
    #: (at fixed sample positions: the coarse pass is the same), with L1 and L2 exact 2-4e-5 (profiles/r06_teacher_mixed_study.txt).
    #: Measured per checkpoint against fp16x3_asm / fp16x3_asm -- the same coarse pass, hence the same sample positions, hence a plain
    #: comparison of the maps -- on up to MIX_PROBE_RAYS rays of every probe set; watched the same way (spot_check).
    AUTO_MAX_DIFF_MIX = 5e-5
    MIX_PROBE_RAYS = 65536
    AUTO_MIX = True
    AUTO_REBALANCE = True
    #: fp16x3_asm (the generated three-pass chain) against fp16x3 (the compiler-scheduled kernel every probe compares with), round 6
    #:: both are fp32-grade, and on a TRAINED teacher two fp32-grade evaluations differ by more than any tight limit on the
    #: rays where sample_pdf is discontinuous -- a coarse weight that differs in its last float32 bits moves a fine sample by up to a
    #: coarse bin (helpers:312-326; the fp32 reference itself is > 1e-4 from float64 there: profiles/r06_teacher_whole_frame.txt).  The
    #: comparison is therefore made stage by stage: the COARSE maps (rgb0 / acc0: continuous in the weights; measured 7.8e-7 apart over
    #: whole frames of the trained-like teacher) on every ray, and the fine network + compositing AT THE SAME SAMPLE POSITIONS (run_network on
    #: the reference's z_vals: measured 5.4e-7 apart): even samples that move by 2e-5 change a pixel by 2e-4 where sigma ~ 200.  Tight limit: both are fp32-grade.
    AUTO_MAX_DIFF_X3ASM = 5e-6
    #: rays of a spot check (spot_check): < 0.5 % of a 100-pose save group, < 2 % of one 400 x 400 frame
    WATCH_RAYS = 2048

    def _limits(self, limit):
        return {'rgb_map': limit, 'acc_map': limit, 'depth_map': limit * max(1., float(self.far))}

    @staticmethod
    def _strided(n, k, device):
        return torch.arange(0, n, max(1, n // k), device=device)[:k]

    def choose_precision(self, rays_o, rays_d=None, max_diff=None, max_diff_x1=None, max_diff_mix=None):
        """`--precision auto` for the teacher, measured on THESE weights and rays.  `rays_o, rays_d`: one ray set, or a list of
        (rays_o, rays_d) probe sets spanning what the job will render (create_data: top-down to horizontal poses, focal x 1 ... x 2;
        render_path: the first, middle and last pose); up to 4,096 rays of every set, spread over it, are rendered coarse + fine in
        fp16x3 and in the candidates, fastest first:
          fp16x1    the generated layer chain WITHOUT correction terms: one fp16 pass on the 256-wide sources (1.0 pass-equivalents).
                    Eleven layers and the compositing over 192 samples average its rounding errors to ~1e-5 on rgb where the
                    88-layer R2L student ends at 3.5e-4 -- for the teacher it is the fast mode (33.4-35 ms per 400 x 400 frame);
          fp16_fp8  the generated layer chain, fp16 + bf6 correction terms (1.5 pass-equivalents, ~1e-6), for weights whose
                    single-pass error is too large; its fixed exponents (|a| x 16 / 2^3 within bf6's +-28) are what is measured;
          fp16x3_asm  three fp16 passes on hi / lo fragments of both operands: fp16x3's arithmetic on the generated layer chain
                    (nerf_chain_kernel<false, 2, true>, 17 % faster than the compiler-scheduled fp16x3 it is tested against; raw within
                    1.5e-5 of it, rgb 2e-7 on smooth teachers).  Unconditional: every TRAINED teacher ends here -- sharp densities
                    amplify a single pass's error to 3e-3 and the fine samples follow the coarse weights discontinuously on rays
                    that graze an object (profiles/r05_trained_like.txt).
        A candidate is kept when rgb, acc and depth agree with fp16x3 within its limits on EVERY probe set.  Returns (name, its
        largest rgb / acc difference); `auto_diffs` keeps the per-candidate maxima, `auto_detail` the per-set, per-output ones.
        The choice is not final: spot_check / step_down keep watching what is rendered afterwards (create_data per save group,
        render_path every few frames).  Synchronous, once per weight load."""
        from ._lib import PREC_FP16_FP8
        if getattr(self, 'skip_rgb0', False):
            self.set_skip_rgb0(False)           # the measurements below compare the coarse maps too; the render loops switch it on again
        max_diff = self.AUTO_MAX_DIFF if max_diff is None else float(max_diff)
        max_diff_x1 = self.AUTO_MAX_DIFF_X1 if max_diff_x1 is None else float(max_diff_x1)
        sets = [(rays_o, rays_d)] if rays_d is not None else list(rays_o)
        probes = []
        for ro, rd in sets:
            ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
            idx = self._strided(ro.shape[0], 4096, ro.device)
            probes.append((ro[idx].contiguous(), rd[idx].contiguous()))
        self.set_precision(PREC_FP16X3)
        refs = []
        for ro, rd in probes:
            r = self.render_rays(ro, rd)
            refs.append({k: r[k].clone() for k in self.WATCH_KEYS})
        self.auto_diffs, self.auto_detail = {}, {}
        diff = float('nan')
        for name, prec, limit in (('fp16x1', PREC_FP16X1, max_diff_x1), ('fp16_fp8', PREC_FP16_FP8, max_diff)):
            self.set_precision(prec)
            lim = self._limits(limit)
            per_set, ok = [], True
            for (ro, rd), ref in zip(probes, refs):
                got = self.render_rays(ro, rd)
                d = {k: float((got[k] - ref[k]).abs().max()) for k in ref}
                per_set.append(d)
                ok = ok and all(d[k] <= lim[k] for k in d)               # NaN fails
            diff = max(max(d['rgb_map'], d['acc_map']) for d in per_set)
            self.auto_diffs[name] = diff
            self.auto_detail[name] = per_set
            if ok:
                return name, diff
        from ._lib import PREC_FP16X3_ASM, PREC_FP16_MIX
        if self.AUTO_MIX:
            # coarse fp16x3_asm + fine fp16_mix against fp16x3_asm for both: identical coarse passes, so no fine sample moves
            big = []
            for ro, rd in sets:
                ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
                idx = self._strided(ro.shape[0], self.MIX_PROBE_RAYS, ro.device)
                big.append((ro[idx].contiguous(), rd[idx].contiguous()))
            self.set_precision(PREC_FP16X3_ASM)
            if self.AUTO_REBALANCE and not self.ndc:      # (NDC renders: the points the fine network sees are not ro + rd z of the given rays)
                # the fine network's activations on the probes' own fine sample positions -> exact rescaling into the bf6 chain's range
                # (measured on the trained-like teacher: worst rgb difference over whole frames 4.1e-5 -> 2.4e-5)
                zs, rs = [], []
                for ro, rd in probes:
                    zs.append(self.render_rays(ro, rd, extras=True)['z_vals'].clone())
                self.rebalance_fine(torch.cat([p[0] for p in probes]), torch.cat([p[1] for p in probes]), torch.cat(zs))
            refs_m = []
            for ro, rd in big:
                r = self.render_rays(ro, rd, extras=True)
                refs_m.append(dict({k: r[k].clone() for k in self.WATCH_KEYS}, raw_last=r['raw'][:, -1:, :].clone()))
                del r
            self.set_precision(PREC_FP16_MIX)
            lim = self._limits(self.AUTO_MAX_DIFF_MIX if max_diff_mix is None else float(max_diff_mix))
            per_set, ok = [], True
            for (ro, rd), ref in zip(big, refs_m):
                got = self.render_rays(ro, rd, extras=True)
                d = self._map_diffs(got, ref, self._far_ties(got['raw'], ref['raw_last']))
                del got
                per_set.append(d)
                ok = ok and all(d[k] <= lim[k] for k in self.WATCH_KEYS)
            self.auto_diffs['fp16_mix'] = max(max(d['rgb_map'], d['acc_map']) for d in per_set)
            self.auto_detail['fp16_mix'] = per_set
            del refs_m
            if ok:
                # ... provided the generated three-pass chain it leans on agrees with fp16x3 (below); checked on the small probes
                self.set_precision(PREC_FP16X3)
                refs_x = [{k: v.clone() for k, v in self.render_rays(ro, rd, extras=True).items() if k in self.WATCH_KEYS + ('rgb0', 'acc0', 'z_vals', 'raw')}
                          for ro, rd in probes]
                self.set_precision(PREC_FP16X3_ASM)
                checks = [self._x3_pair(ro, rd, ref) for (ro, rd), ref in zip(probes, refs_x)]
                self.auto_diffs['fp16x3_asm'] = max(max(d['rgb_map'], d['acc_map'], d['rgb0'], d['acc0']) for d, _ in checks)
                self.auto_detail['fp16x3_asm'] = [d for d, _ in checks]
                if all(g for _, g in checks):
                    self.set_precision(PREC_FP16_MIX)
                    return 'fp16_mix', self.auto_diffs['fp16_mix']
        self.set_precision(PREC_FP16X3)
        refs_x = [self.render_rays(ro, rd, extras=True) for ro, rd in probes]
        refs_x = [{k: v.clone() for k, v in r.items() if k in self.WATCH_KEYS + ('rgb0', 'acc0', 'z_vals', 'raw')} for r in refs_x]
        self.set_precision(PREC_FP16X3_ASM)
        per_set, ok = [], True
        for (ro, rd), ref in zip(probes, refs_x):
            d, good = self._x3_pair(ro, rd, ref)
            per_set.append(d)
            ok = ok and good
        self.auto_diffs['fp16x3_asm'] = max(max(d['rgb_map'], d['acc_map'], d['rgb0'], d['acc0']) for d in per_set)
        self.auto_detail['fp16x3_asm'] = per_set
        if ok:
            return 'fp16x3_asm', self.auto_diffs['fp16x3_asm']
        self.set_precision(PREC_FP16X3)          # the generated chain disagrees with the mode everything is measured against: that mode
        return 'fp16x3', self.auto_diffs['fp16x3_asm']

    #: The far plane is the reference's second discontinuity: raw2outputs gives the LAST sample of a ray the distance 1e10 (main.py:571-573),
    #: so its alpha is 0 or 1 by the SIGN of its raw density -- a density of +-1e-6 there, which float32-grade arithmetic cannot pin, moves
    #: acc by the ray's whole remaining transmittance and depth by far x that (measured on the second trained-like teacher: one probe ray,
    #: acc 0.82, depth 4.9, rgb unchanged because the sample is as white as the background).  Two evaluations at the same sample positions
    #: are therefore compared on acc / depth only where the far sample's density has the same sign in both or is not within FAR_TIE of
    #: zero in both; rgb is compared on every ray (a ray whose colour flips there fails the candidate).
    FAR_TIE = 1e-3

    def _far_ties(self, raw_a, raw_b):
        """[n] mask of rays whose last sample's raw density (raw[:, -1, 3]) is a far-plane tie between two evaluations"""
        a, b = raw_a[:, -1, 3], raw_b[:, -1, 3]
        return ((a > 0) != (b > 0)) & (a.abs() < self.FAR_TIE) & (b.abs() < self.FAR_TIE)

    def _map_diffs(self, got, ref, tie=None):
        """{output: largest difference} over WATCH_KEYS; acc / depth without the rays of `tie` (their number as 'far_plane_ties')"""
        d = {}
        for k in self.WATCH_KEYS:
            e = (got[k] - ref[k]).abs()
            if tie is not None and k != 'rgb_map' and bool(tie.any()):
                e = e[~tie]
            d[k] = float(e.max()) if e.numel() else 0.0
        if tie is not None:
            d['far_plane_ties'] = int(tie.sum())
        return d

    def _x3_pair(self, ro, rd, ref):
        """fp16x3_asm (the current mode) against `ref` = what fp16x3 rendered for these rays with extras: ({output: largest difference},
        ok) -- stage by stage, see AUTO_MAX_DIFF_X3ASM: the coarse maps of a full render, then the FINE network and its compositing at
        the reference's own sample positions (run_network + raw2outputs on ref['z_vals'])"""
        lim = self._limits(self.AUTO_MAX_DIFF_X3ASM)
        lim['rgb0'] = lim['acc0'] = self.AUTO_MAX_DIFF_X3ASM
        got = self.render_rays(ro, rd, extras=True)
        d = {k: float((got[k] - ref[k]).abs().max()) for k in ('rgb0', 'acc0')}
        raw = self.run_network(1, ro, rd, ref['z_vals'])
        rgb, _, acc, _, depth = raw2outputs(raw, ref['z_vals'], rd, white_bkgd=self.white_bkgd)
        tie = self._far_ties(raw, ref['raw']) if 'raw' in ref else None
        d.update(self._map_diffs({'rgb_map': rgb, 'acc_map': acc, 'depth_map': depth}, ref, tie))
        return d, all(d[k] <= lim[k] for k in lim)       # NaN fails

    @property
    def precision_name(self):
        from ._lib import PRECISIONS
        return next(k for k, v in PRECISIONS.items() if v == self.precision)

    def spot_check(self, rays_o, rays_d, got, n_rays=None):
        """The watch behind `--precision auto`'s one-time choice: `got` is what this engine just rendered for (rays_o, rays_d) in
        its current fast mode; up to `n_rays` (default WATCH_RAYS) of those rays, spread over the set, are rendered again in
        fp16x3 and rgb / acc / depth compared under the mode's limits (a ray's result does not depend on the batch it is in).
        Returns (ok, {output: largest difference}).  fp16x3 itself: (True, {}).  Synchronous (one small render + three maxima)."""
        from ._lib import PREC_FP16_FP8, PREC_FP16X3_ASM
        cur = self.precision
        from ._lib import PREC_FP16_MIX
        if cur not in (PREC_FP16X1, PREC_FP16_FP8, PREC_FP16X3_ASM, PREC_FP16_MIX):
            return True, {}
        ro, rd = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
        idx = self._strided(ro.shape[0], int(n_rays or self.WATCH_RAYS), ro.device)
        if cur == PREC_FP16X3_ASM:        # two fp32-grade modes: coarse maps, then the fine pass at the same sample positions (_x3_pair)
            ro_s, rd_s = ro[idx].contiguous(), rd[idx].contiguous()
            skipped = getattr(self, 'skip_rgb0', False)
            if skipped:
                self.set_skip_rgb0(False)       # the comparison of the coarse maps includes rgb0
            self.set_precision(PREC_FP16X3)
            try:
                ref = {k: v.clone() for k, v in self.render_rays(ro_s, rd_s, extras=True).items()}
            finally:
                self.set_precision(cur)
            try:
                d, good = self._x3_pair(ro_s, rd_s, ref)
            finally:
                if skipped:
                    self.set_skip_rgb0(True)
            self.watch_checks = getattr(self, 'watch_checks', 0) + 1
            return good, d
        lim = self._limits(self.AUTO_MAX_DIFF_X1 if cur == PREC_FP16X1 else self.AUTO_MAX_DIFF_MIX if cur == PREC_FP16_MIX else self.AUTO_MAX_DIFF)
        if cur == PREC_FP16_MIX:
            # against three passes for BOTH networks on the generated chain: the same coarse pass, so no fine sample moves and the maps
            # compare directly -- acc / depth without the far-plane ties (FAR_TIE), which needs the sample's raw of both renders
            ro_s, rd_s = ro[idx].contiguous(), rd[idx].contiguous()
            mine = {k: v.clone() for k, v in self.render_rays(ro_s, rd_s, extras=True).items() if k in self.WATCH_KEYS + ('raw',)}
            self.set_precision(PREC_FP16X3_ASM)
            try:
                ref = self.render_rays(ro_s, rd_s, extras=True)
            finally:
                self.set_precision(cur)
            d = self._map_diffs(mine, ref, self._far_ties(mine['raw'], ref['raw']))
            # (what was rendered for the whole set is this mode's render of these rays: a ray's result does not depend on its batch)
            d['rgb_map'] = max(d['rgb_map'], float((got['rgb_map'].reshape(-1, 3)[idx] - ref['rgb_map']).abs().max()))
            self.watch_checks = getattr(self, 'watch_checks', 0) + 1
            return all(d[k] <= lim[k] for k in self.WATCH_KEYS), d
        self.set_precision(PREC_FP16X3)
        try:
            ref = self.render_rays(ro[idx].contiguous(), rd[idx].contiguous())
        finally:
            self.set_precision(cur)
        d = {k: float((got[k].reshape((-1,) + tuple(ref[k].shape[1:]))[idx] - ref[k]).abs().max()) for k in self.WATCH_KEYS}
        self.watch_checks = getattr(self, 'watch_checks', 0) + 1
        return all(d[k] <= lim[k] for k in d), d

    def step_down(self):
        """one rung down the ladder fp16x1 -> fp16_fp8 -> fp16x3_asm -> fp16x3 (after a failed spot_check); returns the new mode's name"""
        from ._lib import PRECISIONS
        name = self.precision_name
        nxt = self.LADDER[min(self.LADDER.index(name) + 1, len(self.LADDER) - 1)] if name in self.LADDER else 'fp16x3'
        self.set_precision(PRECISIONS[nxt])
        self.watch_fallbacks = getattr(self, 'watch_fallbacks', 0) + 1
        return nxt

    def timing(self, on=True):
        """HIP events around every MLP launch (nerf_chain_kernel / nerf_mlp_kernel), on its launch stream"""
        check(lib().nerf_timing_enable(self._ctx, int(on)))

    def kernel_time_ms(self, reset=True):
        """(sum of the MLP launches' durations since the last reset in ms, their number); synchronises those events"""
        tot, n = C.c_double(), C.c_int()
        check(lib().nerf_kernel_time_ms(self._ctx, C.byref(tot), C.byref(n), int(reset)))
        return tot.value, n.value

    def _outs(self, n):
        dev = self.device
        return (torch.empty((n, 3), dtype=torch.float32, device=dev), torch.empty((n,), dtype=torch.float32, device=dev),
                torch.empty((n,), dtype=torch.float32, device=dev), torch.empty((n,), dtype=torch.float32, device=dev))

    def _extras(self, n):
        S1 = self.N_samples + self.N_importance
        dev = self.device
        rgb0 = None if self._rgb0_skipped() else torch.empty((n, 3), dtype=torch.float32, device=dev)
        zs = torch.empty((n, self.N_importance), dtype=torch.float32, device=dev)
        zv = torch.empty((n, S1), dtype=torch.float32, device=dev)
        raw = torch.empty((n, S1, 4), dtype=torch.float32, device=dev)
        check(lib().nerf_copy_extras(self._ctx, n, dptr(rgb0), dptr(zs), dptr(zv), dptr(raw), current_stream()))
        disp0, acc0, z_std = (torch.empty((n,), dtype=torch.float32, device=dev) for _ in range(3))
        check(lib().nerf_copy_extras0(self._ctx, n, dptr(disp0), dptr(acc0), dptr(z_std), current_stream()))
        # main.py:743-750: rgb0, disp0, acc0, z_std (+ raw with retraw); z_samples / z_vals for the parity tests
        ret = {'disp0': disp0, 'acc0': acc0, 'z_std': z_std, 'z_samples': zs, 'z_vals': zv, 'raw': raw}
        if rgb0 is not None:            # (set_skip_rgb0: the coarse pass ran without its view branch)
            ret['rgb0'] = rgb0
        return ret

    def render(self, c2w, rows=None, extras=False):
        """render(H, W, focal, c2w=c2w[:3,:4]) of main.py:107-186 for rows [r0,r1):
        dict(rgb_map [n,3], disp_map [n], acc_map [n], depth_map [n])."""
        r0, r1 = (0, self.H) if rows is None else (int(rows[0]), int(rows[1]))
        c = _c2w_host(c2w)
        n = (r1 - r0) * self.W
        rgb, disp, acc, depth = self._outs(n)
        with torch.cuda.device(self.device):
            check(lib().nerf_render(self._ctx, C.c_void_p(c.data_ptr()), r0, r1, dptr(rgb), dptr(disp), dptr(acc),
                                    dptr(depth), current_stream()))
            ret = {'rgb_map': rgb, 'disp_map': disp, 'acc_map': acc, 'depth_map': depth}
            if extras:
                ret.update(self._extras(n))
        return ret

    def render_rays(self, rays_o, rays_d, extras=False, perturb=0., raw_noise_std=0., pytest=False):
        """render(..., rays=(rays_o, rays_d)) / render_rays of main.py:624-756.  perturb > 0: stratified jitter of
        the coarse depths (main.py:684-699) and random sample_pdf uniforms (det=False); raw_noise_std > 0: noise on
        the densities (main.py:592-600); pytest=True: the reference's fixed numpy streams.  The numbers are drawn
        here, in the reference's order, and handed to the library."""
        rays_o, rays_d = _f32(rays_o, self.device).view(-1, 3), _f32(rays_d, self.device).view(-1, 3)
        n = rays_o.shape[0]
        rgb, disp, acc, depth = self._outs(n)
        zc = u = n0 = n1 = None
        dev = self.device
        if perturb > 0.:
            # evaluated on the host (n x N_samples floats): device elementwise kernels contract a + b*c into an fma
            z = self.z_coarse.expand(n, self.N_samples)
            mids = .5 * (z[..., 1:] + z[..., :-1])
            upper = torch.cat([mids, z[..., -1:]], -1)
            lower = torch.cat([z[..., :1], mids], -1)
            if pytest:
                np.random.seed(0)
                t_rand = torch.Tensor(np.random.rand(n, self.N_samples))
            else:
                t_rand = torch.rand((n, self.N_samples))
            zc = _f32(lower + (upper - lower) * t_rand, dev)
        if raw_noise_std > 0.:
            n0 = _f32(_raw_noise((n, self.N_samples), raw_noise_std, pytest, dev), dev)
        if perturb > 0.:
            u = _f32(_sample_pdf_u(n, self.N_importance, False, pytest, dev)[0], dev)
        elif pytest:  # det with the pytest stream: u = float32(np.linspace) on every ray
            u = _f32(_sample_pdf_u(n, self.N_importance, True, True, dev)[0].expand(n, self.N_importance), dev)
        if raw_noise_std > 0.:
            n1 = _f32(_raw_noise((n, self.N_samples + self.N_importance), raw_noise_std, pytest, dev), dev)
        with torch.cuda.device(self.device):
            check(lib().nerf_render_rays_ex(self._ctx, dptr(rays_o), dptr(rays_d), n, dptr(zc), dptr(u), dptr(n0), dptr(n1),
                                            dptr(rgb), dptr(disp), dptr(acc), dptr(depth), current_stream()))
            ret = {'rgb_map': rgb, 'disp_map': disp, 'acc_map': acc, 'depth_map': depth}
            if extras:
                ret.update(self._extras(n))
        return ret

    def run_network(self, which, rays_o, rays_d, z_vals):
        """network_query_fn(pts, viewdirs, network) of main.py:447-453 with
        pts = rays_o + rays_d * z_vals: raw [n,S,4]."""
        rays_o, rays_d = _f32(rays_o, self.device), _f32(rays_d, self.device)
        n = rays_o.shape[0]
        z = _f32(z_vals, self.device)
        shared = z.dim() == 1
        S = z.shape[-1]
        raw = torch.empty((n, S, 4), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().nerf_run_network(self._ctx, int(which), dptr(rays_o), dptr(rays_d), dptr(z), 0 if shared else S,
                                         S, n, dptr(raw), current_stream()))
        return raw


def render(H, W, focal, chunk=1024 * 32, rays=None, c2w=None, ndc=False, near=2., far=6., use_viewdirs=True,
           engine=None, **kwargs):
    """main.py:107-186 call shape: returns [rgb_map, disp_map, acc_map, extras_dict] reshaped
    to the ray batch.  `engine` is the NeRFEngine holding network_fn / network_fine (the
    reference passes them through **kwargs).  `chunk` does not affect results (main.py:124)."""
    if ndc != bool(getattr(engine, 'ndc', False)):
        raise R2LError(f'render(ndc={ndc}) on an engine built with ndc={getattr(engine, "ndc", False)}: '
                       'construct NeRFEngine(..., ndc=True, near=0., far=1.) for forward-facing scenes')
    if not use_viewdirs:
        raise R2LError('the teacher path is built for use_viewdirs=True networks')
    if engine is None:
        raise R2LError('pass engine=NeRFEngine(...) (holds the coarse/fine networks)')
    if c2w is not None:
        out = engine.render(c2w)
        sh = (engine.H, engine.W)
    else:
        rays_o, rays_d = rays
        sh = tuple(rays_d.shape[:-1])
        out = engine.render_rays(rays_o, rays_d)
    rgb = out['rgb_map'].view(*sh, 3)
    disp, acc = out['disp_map'].view(*sh), out['acc_map'].view(*sh)
    return [rgb, disp, acc, {'depth_map': out['depth_map'].view(*sh)}]
