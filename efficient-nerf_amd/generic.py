"""Generic fp32 layer path: every NeRF_v3_2 / NeRF network the reference's constructors build, composed of one
r2l_linear_forward launch per nn.Linear (csrc/r2l_generic.hip, include/r2l_hip.h).

The fused kernels (r2l.R2LEngine, teacher.NeRFEngine) are built for the README's shapes -- R2L W256 with 1008 inputs, the
8 x 256 NeRF.  The reference's constructors accept more (model/nerf_raybased.py:483-537, 339-401): other widths,
--layerwise_netwidths, trial.n_learnable != 2, other n_sample_per_ray / multires, odd mlp depths, other teacher depths and
widths, no view directions.  Those run here: fp32 products and accumulation on the fp32 MFMA (the reference's own
precision: nothing to calibrate, L_inf vs the reference ~2e-7), activations through HBM.  The README's own networks run 11 x (R2L
W256D88: 5.5e6 rays/s) and 17 x (8 x 256 teacher: 2.4e5 rays/s) slower here than on the fused kernels
(profiles/r04_generic_time.txt); the front end takes this path only for what those refuse.

The module structure (which state_dict key is which Linear, where activations and residuals sit) is restated from the
constructors cited at each function; tests/golden/make_golden_generic.py pins it against the reference's own classes.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import R2LError, check, current_stream, dptr, lib

ACT_CODES = {'none': 0, 'relu': 1, 'lrelu': 2, 'sigmoid': 3}


def _act_code(name):
    name = 'none' if name is None else str(name).lower()
    if name not in ACT_CODES:
        raise R2LError(f'activation {name!r}: the reference knows relu, lrelu and none (model/nerf_raybased.py:468-476)')
    return ACT_CODES[name]


def _view(t, width=None):
    """(pointer, row stride in floats) of a 2-D float32 device view whose rows are contiguous."""
    if t is None:
        return None, 0
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and (t.shape[1] == 1 or t.stride(1) == 1)):
        raise R2LError(f'expected a [n, c] float32 device view with contiguous rows; got {tuple(t.shape)} strides {t.stride()} {t.dtype} on {t.device}')
    if width is not None and t.shape[1] != width:
        raise R2LError(f'expected {width} columns, got {t.shape[1]}')
    return C.c_void_p(t.data_ptr()), int(t.stride(0))


class Linear:
    """One nn.Linear on the device (r2l_linear_create): weight [out, in], bias [out] or None, as in the state_dict."""

    def __init__(self, weight, bias=None, device=None):
        w = torch.as_tensor(weight).detach().to('cpu', torch.float32).contiguous()
        if w.dim() != 2:
            raise R2LError(f'Linear weight must be [out, in]; got {tuple(w.shape)}')
        b = None if bias is None else torch.as_tensor(bias).detach().to('cpu', torch.float32).contiguous()
        if b is not None and tuple(b.shape) != (w.shape[0],):
            raise R2LError(f'Linear bias must be [{w.shape[0]}]; got {tuple(b.shape)}')
        self.out_dim, self.in_dim = int(w.shape[0]), int(w.shape[1])
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib().r2l_linear_create(C.byref(self._h), C.c_void_p(w.data_ptr()), None if b is None else C.c_void_p(b.data_ptr()),
                                          self.out_dim, self.in_dim))

    def close(self):
        if getattr(self, '_h', None) and self._h.value and _lib is not None and _lib._lib is not None:
            _lib._lib.r2l_linear_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __call__(self, x, y, act='none', res=None, res_scale=1.0, post=None):
        """y = post + act((x W^T + b) * res_scale + res); x [n, in], y [n, out] device views (row strides free)."""
        n = x.shape[0]
        xp, ldx = _view(x, self.in_dim)
        yp, ldy = _view(y, self.out_dim)
        rp, ldr = _view(res, self.out_dim if res is not None else None)
        pp, ldp = _view(post, self.out_dim if post is not None else None)
        if y.shape[0] != n or (res is not None and res.shape[0] != n) or (post is not None and post.shape[0] != n):
            raise R2LError('row counts of x / y / res / post differ')
        with torch.cuda.device(self.device):
            check(lib().r2l_linear_forward(self._h, xp, ldx, n, yp, ldy, rp, ldr, float(res_scale), _act_code(act), pp, ldp,
                                           current_stream()))
        return y


def _strip(sd):
    return {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}


def v3_2_plan(netdepth, netwidth, input_dim, output_dim=3, layerwise_netwidths='', act='relu', use_residual=True, trial=None):
    """The Linear layers of NeRF_v3_2 in execution order (model/nerf_raybased.py:483-544), each as
    dict(key, in_dim, out_dim, act, block_in (this layer opens a residual block), block_out (it closes one: its output is
    outact(lin * res_scale + block input)), res_scale).  `trial`: None (no --trial.ON: the plain body of :497-501) or a mapping /
    namespace with body_arch, n_block, n_learnable, res_scale, inact, outact."""
    D, W = int(netdepth), int(netwidth)
    if layerwise_netwidths:
        Ws = [int(v) for v in str(layerwise_netwidths).split(',')] + [output_dim]       # :489-491
    else:
        Ws = [W] * (D - 1) + [output_dim]
    if len(Ws) < D or D < 3:
        raise R2LError(f'netdepth={D} with {len(Ws) - 1} layer widths: the constructor indexes Ws[0 .. netdepth - 2] (model/nerf_raybased.py:497-537)')
    get = (lambda k, d=None: d) if trial is None else ((lambda k, d=None: trial.get(k, d)) if isinstance(trial, dict) else (lambda k, d=None: getattr(trial, k, d)))
    _act_code(act)
    plan = [dict(key='head.0', in_dim=int(input_dim), out_dim=Ws[0], act=str(act).lower())]
    arch = 'mlp' if trial is None else str(get('body_arch', 'resmlp'))
    if arch == 'mlp':                                        # :497-501 / :515-518: nn.Sequential(Linear, act, Linear, act, ...)
        if str(act).lower() == 'none':
            raise R2LError('a plain MLP body with --act none: the reference cannot build that network either (nn.Sequential of None)')
        for i in range(1, D - 1):
            plan.append(dict(key=f'body.{2 * (i - 1)}', in_dim=Ws[i - 1], out_dim=Ws[i], act=str(act).lower()))
    elif arch == 'resmlp':                                   # :503-514 with ResMLP :443-465
        n_block = int(get('n_block', -1))
        if n_block <= 0:
            n_block = (D - 2) // 2
        n_learn = int(get('n_learnable', 2))
        inact, outact = str(get('inact', 'relu')).lower(), str(get('outact', 'none')).lower()
        _act_code(inact), _act_code(outact)
        if n_learn < 1:
            raise R2LError(f'trial.n_learnable={n_learn}')
        rs = float(get('res_scale', 1.0))
        for bi in range(n_block):
            sub = 0
            for j in range(n_learn):
                last = j == n_learn - 1
                plan.append(dict(key=f'body.{bi}.body.{sub}', in_dim=W, out_dim=W, act=outact if last else inact, block_in=j == 0,
                                 block_out=last, res_scale=rs))
                sub += 1 if (last or inact == 'none') else 2      # nn.Sequential index: the activation module takes a slot (:450-454)
    else:
        raise R2LError(f'--trial.body_arch {arch}: resmlp or mlp (model/nerf_raybased.py:503-518)')
    n_body = len(plan) - 1
    plan.append(dict(key='tail.0', in_dim=Ws[D - 2], out_dim=int(output_dim), act='sigmoid'))     # :534-537
    for a, b in zip(plan[:-1], plan[1:]):
        if a['out_dim'] != b['in_dim']:
            raise R2LError(f"layer {b['key']} takes {b['in_dim']} inputs but {a['key']} produces {a['out_dim']}: the reference's forward "
                           f'would fail the same way (model/nerf_raybased.py:539-544)')
    if use_residual and n_body and plan[n_body]['out_dim'] != plan[0]['out_dim']:
        raise R2LError('--use_residual adds the head output to the body output: their widths differ')
    return plan


class GenericR2L:
    """NeRF_v3_2 of any shape the constructor accepts + PointSampler + PositionalEmbedder for any n_sample / L
    (model/nerf_raybased.py:76-126, 191-208, 480-544), as launches of the generic kernels.  The call surface R2LEngine has:
    render / render_batch / render_rays."""

    precision_name = 'fp32'
    precision = -1          # none of the library's fp16 modes: nothing to calibrate, agree on or watch (dist.agree_act_exponents)
    n_block = 0

    def render_checked(self, render, log=None, check=None):
        """R2LEngine.render_checked's shape: fp32 has no operand ranges to leave, so one render and no re-render"""
        return render(), 0

    def __init__(self, H, W, focal, near=2., far=6., n_sample=16, L=10, netdepth=88, netwidth=256, layerwise_netwidths='',
                 act='relu', use_residual=True, trial=None, chunk=1 << 16, device=None, z_vals=None):
        if not torch.cuda.is_available():
            raise R2LError('no HIP device visible to torch: the R2L path has no CPU fallback')
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.H, self.W, self.focal = int(H), int(W), float(focal)
        self.n_sample, self.L = int(n_sample), int(L)
        if not 1 <= self.L <= 16 or self.n_sample < 1:
            raise R2LError(f'n_sample={n_sample} multires={L}')
        self.input_dim = 3 * self.n_sample * (2 * self.L + 1)
        self.use_residual = bool(use_residual)
        self.plan = v3_2_plan(netdepth, netwidth, self.input_dim, 3, layerwise_netwidths, act, use_residual, trial)
        self.chunk = int(chunk)
        if z_vals is None:       # model/nerf_raybased.py:88-90, on the host as the reference does (torch.linspace's last ulp)
            t_vals = torch.linspace(0., 1., steps=self.n_sample)
            z_vals = float(near) * (1 - t_vals) + float(far) * t_vals
        self.z_vals = torch.as_tensor(z_vals).detach().to('cpu', torch.float32).contiguous()
        self._z_dev = self.z_vals.to(self.device)
        self.layers = None
        self._buf_n = 0

    @property
    def flops_per_ray(self):
        return 2 * sum(p['in_dim'] * p['out_dim'] for p in self.plan)

    def state_names(self):
        return [f"{p['key']}.{kind}" for p in self.plan for kind in ('weight', 'bias')]

    def load_state_dict(self, state_dict):
        sd = _strip(state_dict)
        missing = [k for k in self.state_names() if k not in sd]
        if missing:
            raise R2LError(f'state_dict lacks {len(missing)} tensors, e.g. {missing[:3]} (has e.g. {sorted(sd)[:4]})')
        for p in self.plan:
            if tuple(sd[p['key'] + '.weight'].shape) != (p['out_dim'], p['in_dim']):
                raise R2LError(f"{p['key']}.weight is {tuple(sd[p['key'] + '.weight'].shape)}, the flags describe ({p['out_dim']}, {p['in_dim']})")
        with torch.cuda.device(self.device):
            self.layers = [Linear(sd[p['key'] + '.weight'], sd[p['key'] + '.bias'], self.device) for p in self.plan]
        return self

    def _buffers(self, n):
        if n > self._buf_n:
            wmax = max(p['out_dim'] for p in self.plan[:-1])
            self._h0 = torch.empty((n, self.plan[0]['out_dim']), dtype=torch.float32, device=self.device)
            self._work = [torch.empty((n, wmax), dtype=torch.float32, device=self.device) for _ in range(3)]
            self._pts = torch.empty((n, 3 * self.n_sample), dtype=torch.float32, device=self.device)
            self._emb = torch.empty((n, self.input_dim), dtype=torch.float32, device=self.device)
            self._buf_n = n

    def forward(self, emb, out=None):
        """NeRF_v3_2.forward (model/nerf_raybased.py:539-544) on embedded inputs [n, input_dim] -> rgb [n, 3]."""
        if self.layers is None:
            raise R2LError('forward before load_state_dict')
        n = emb.shape[0]
        self._buffers(n)
        if out is None:
            out = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        h0 = self._h0[:n]
        self.layers[0](emb, h0, act=self.plan[0]['act'])
        cur, x = h0, h0            # cur: the stream a residual block adds to; x: the running activation
        body = list(zip(self.plan[1:-1], self.layers[1:-1]))
        if not body and self.use_residual:
            raise R2LError('an empty body with --use_residual (netdepth < 4) is not built')
        for idx, (p, lin) in enumerate(body):
            last_of_body = idx == len(body) - 1
            if p.get('block_in'):
                cur = x
            # never the layer's own input, never the stream the block adds to (the head output h0 is not a work buffer: it stays
            # intact for the global skip)
            y = next(wb for wb in self._work if wb.data_ptr() not in (x.data_ptr(), cur.data_ptr()))[:n, :p['out_dim']]
            lin(x, y, act=p['act'], res=cur if p.get('block_out') else None, res_scale=p.get('res_scale', 1.0),
                post=h0 if (last_of_body and self.use_residual) else None)
            x = y
        self.layers[-1](x, out, act='sigmoid')
        return out

    def _render_rays_chunk(self, ro, rd, out):
        n = ro.shape[0]
        self._buffers(n)
        pts, emb = self._pts[:n], self._emb[:n]
        with torch.cuda.device(self.device):
            check(lib().r2l_sample_points(dptr(ro), dptr(rd), n, dptr(self._z_dev), self.n_sample, 0, dptr(pts), current_stream()))
            check(lib().r2l_embed(dptr(pts), n, 3 * self.n_sample, self.L, dptr(emb), current_stream()))
        return self.forward(emb, out)

    def render_rays(self, rays_o, rays_d, out=None):
        """Given rays (main.py:220-230; PointSampler.sample_train without perturbation): [n, 3] device f32 -> rgb [n, 3]."""
        n = rays_o.shape[0]
        if out is None:
            out = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        ro, rd = rays_o.contiguous(), rays_d.contiguous()
        for s in range(0, n, self.chunk):
            self._render_rays_chunk(ro[s:s + self.chunk], rd[s:s + self.chunk], out[s:s + self.chunk])
        return out

    def _rows(self, rows):
        return (0, self.H) if rows is None else (int(rows[0]), int(rows[1]))

    def render(self, c2w, rows=None, out=None):
        """rgb [rows * W, 3] of one pose (render_func, main.py:401-404)."""
        from .r2l import _c2w_host
        r0, r1 = self._rows(rows)
        c = _c2w_host(c2w)
        n = (r1 - r0) * self.W
        if out is None:
            out = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        rows_per = max(1, self.chunk // self.W)
        ro = torch.empty((min(n, rows_per * self.W), 3), dtype=torch.float32, device=self.device)
        rd = torch.empty_like(ro)
        for a in range(r0, r1, rows_per):
            b = min(r1, a + rows_per)
            m = (b - a) * self.W
            with torch.cuda.device(self.device):     # same dirs rule and summation order as PointSampler (model/nerf_raybased.py:80-99)
                check(lib().nerf_get_rays(self.H, self.W, self.focal, C.c_void_p(c.data_ptr()), a, b, dptr(ro[:m]), dptr(rd[:m]), current_stream()))
            self._render_rays_chunk(ro[:m], rd[:m], out[(a - r0) * self.W:(a - r0) * self.W + m])
        return out

    def render_batch(self, c2w_dev, rows=None, out=None):
        """rgb [P, rows * W, 3] for P poses ([P, 3, 4]); the poses go through the host (this path is not the fast one)."""
        if c2w_dev.dim() == 2:
            c2w_dev = c2w_dev[None]
        P = c2w_dev.shape[0]
        r0, r1 = self._rows(rows)
        if out is None:
            out = torch.empty((P, (r1 - r0) * self.W, 3), dtype=torch.float32, device=self.device)
        host = c2w_dev.detach().to('cpu', torch.float32)
        for i in range(P):
            self.render(host[i], rows=(r0, r1), out=out[i])
        return out


def nerf_plan(D, W, input_ch, input_ch_views, output_ch, skips=(4,), use_viewdirs=True):
    """The Linear layers of NeRF (model/nerf_raybased.py:357-375) as (key, in_dim, out_dim)."""
    plan = [(f'pts_linears.{i}', (input_ch if i == 0 else (W + input_ch if (i - 1) in skips else W)), W) for i in range(D)]
    if use_viewdirs:
        plan += [('alpha_linear', W, 1), ('feature_linear', W, W), ('views_linears.0', input_ch_views + W, W // 2), ('rgb_linear', W // 2, 3)]
    else:
        plan += [('output_linear', W, output_ch)]
    return plan


class _NeRFNet:
    """One NeRF module (model/nerf_raybased.py:339-401) as generic layer launches.  The reference's torch.cat inputs are column
    slices of wider buffers: [input_pts | h] behind a skip layer, [feature | input_views] in front of views_linears."""

    def __init__(self, sd, D, W, input_ch, input_ch_views, output_ch, skips, use_viewdirs, device):
        sd = _strip(sd)
        self.D, self.W, self.input_ch, self.input_ch_views, self.skips, self.use_viewdirs = D, W, input_ch, input_ch_views, tuple(skips), use_viewdirs
        self.output_ch = output_ch
        self.lin = {}
        for key, i, o in nerf_plan(D, W, input_ch, input_ch_views, output_ch, skips, use_viewdirs):
            if key + '.weight' not in sd:
                raise R2LError(f"state_dict lacks {key}.weight (has e.g. {sorted(sd)[:4]})")
            if tuple(sd[key + '.weight'].shape) != (o, i):
                raise R2LError(f"{key}.weight is {tuple(sd[key + '.weight'].shape)}, the flags describe ({o}, {i})")
            self.lin[key] = Linear(sd[key + '.weight'], sd[key + '.bias'], device)
        self.macs = sum(i * o for _, i, o in nerf_plan(D, W, input_ch, input_ch_views, output_ch, skips, use_viewdirs))

    def forward(self, cat, views, work, raw):
        """cat [m, input_ch + W]: the embedded points in its first input_ch columns (kept: the skip layer reads the whole row);
        views [m, W + input_ch_views]: the embedded directions in its LAST input_ch_views columns; work: two [m, W] buffers;
        raw [m, 4 | output_ch] receives [rgb, alpha] (:396-399)."""
        ic, W = self.input_ch, self.W
        x = cat[:, :ic]
        pp = 0
        for i in range(self.D):
            into_cat = i in self.skips           # h = cat([input_pts, h]): this layer's output lands behind the points
            y = cat[:, ic:] if into_cat else work[pp]
            if not into_cat:
                pp ^= 1
            self.lin[f'pts_linears.{i}'](x, y, act='relu')
            x = cat if into_cat else y
        if x is cat and x.shape[1] != W:
            raise R2LError('the last pts_linears layer is a skip layer: alpha_linear / feature_linear take W inputs (the reference fails too)')
        if not self.use_viewdirs:
            self.lin['output_linear'](x, raw)
            return raw
        self.lin['alpha_linear'](x, raw[:, 3:4])
        self.lin['feature_linear'](x, views[:, :W])
        h = work[pp][:, :W // 2]
        if h.data_ptr() == x.data_ptr():
            h = work[pp ^ 1][:, :W // 2]
        self.lin['views_linears.0'](views, h, act='relu')
        self.lin['rgb_linear'](h, raw[:, :3])
        return raw


class GenericNeRF:
    """render / render_rays of main.py:107-186, 624-756 at test time (perturb = 0, raw_noise_std = 0) for ANY pair of networks
    create_nerf builds (main.py:407-453): the scan stages are the library's stand-alone kernels (nerf_raw2outputs,
    nerf_sample_pdf, nerf_merge_sorted: the ones NeRFEngine fuses behind its 8 x 256 MLP kernel), the networks are generic
    layer launches.  Returns what NeRFEngine returns."""

    precision_name = 'fp32'

    def __init__(self, H, W, focal, near=2., far=6., N_samples=64, N_importance=128, multires=10, multires_views=4, i_embed=0,
                 netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256, use_viewdirs=True, white_bkgd=False, lindisp=False,
                 ndc=False, chunk=1 << 13, device=None):
        if not torch.cuda.is_available():
            raise R2LError('no HIP device visible to torch: the teacher path has no CPU fallback')
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.H, self.W, self.focal, self.near, self.far = int(H), int(W), float(focal), float(near), float(far)
        self.N_samples, self.N_importance = int(N_samples), int(N_importance)
        self.multires, self.multires_views, self.i_embed = int(multires), int(multires_views), int(i_embed)
        if self.i_embed not in (0, -1):
            raise R2LError(f'i_embed={i_embed}: 0 (positional encoding) or -1 (none), utils/run_nerf_raybased_helpers.py:59-74')
        self.use_viewdirs, self.white_bkgd, self.lindisp, self.ndc = bool(use_viewdirs), bool(white_bkgd), bool(lindisp), bool(ndc)
        self.shape = ((int(netdepth), int(netwidth)), (int(netdepth_fine), int(netwidth_fine)))
        self.input_ch = 3 if self.i_embed == -1 else 3 * (2 * self.multires + 1)
        self.input_ch_views = 0 if not self.use_viewdirs else (3 if self.i_embed == -1 else 3 * (2 * self.multires_views + 1))
        self.output_ch = 5 if self.N_importance > 0 else 4          # main.py:426
        # rays per call, sized from the widest network so that a call's per-point buffers stay near 2^18 points x ~1,100 floats
        # at W = 256 (1.2 GB) and shrink as the width grows (8,192 rays x 192 samples x W floats grew linearly with netwidth)
        wmax = max(int(netwidth), int(netwidth_fine))
        pts_cap = max(1 << 14, (1 << 18) * 256 // max(wmax, 256))
        self.chunk = max(64, min(int(chunk), pts_cap // max(1, self.N_samples + self.N_importance)))
        self._pt_buffers = {}
        # main.py:673-682 on the host as the reference's first ray computes it (near, far are the same for every ray)
        t = torch.linspace(0., 1., steps=self.N_samples)
        nr, fr = torch.tensor([self.near]), torch.tensor([self.far])
        z = nr * (1. - t) + fr * t if not self.lindisp else 1. / (1. / nr * (1. - t) + 1. / fr * t)
        self.z_coarse = z.to(torch.float32).contiguous()
        self._z_dev = self.z_coarse.to(self.device)
        self.nets = None

    @property
    def flops_per_ray(self):
        m0, m1 = self.nets[0].macs, self.nets[1].macs if self.nets[1] is not None else 0
        return 2 * (m0 * self.N_samples + m1 * (self.N_samples + self.N_importance if self.N_importance > 0 else 0))

    def load_state_dicts(self, network_fn_state_dict, network_fine_state_dict=None):
        with torch.cuda.device(self.device):
            mk = lambda sd, dw: _NeRFNet(sd, dw[0], dw[1], self.input_ch, self.input_ch_views, self.output_ch, (4,), self.use_viewdirs, self.device)
            coarse = mk(network_fn_state_dict, self.shape[0])
            fine = None
            if self.N_importance > 0:
                if network_fine_state_dict is None:
                    raise R2LError('N_importance > 0 needs network_fine_state_dict (main.py:436-445)')
                fine = mk(network_fine_state_dict, self.shape[1])
            self.nets = (coarse, fine)
        return self

    def _embed(self, x, L, out):
        """get_embedder(L, i_embed) of x [m, 3] into the view `out` [m, 3 (2 L + 1)] (or [m, 3] for i_embed = -1)"""
        if self.i_embed == -1:
            out.copy_(x)
            return
        op, ldo = _view(out)
        check(lib().nerf_embed(dptr(x), 3, x.shape[0], 3, L, op, ldo, current_stream()))

    def run_network(self, which, rays_o, rays_d, z_vals, viewdirs=None):
        """network_query_fn(pts, viewdirs, network) of main.py:447-453 with pts = rays_o + rays_d * z_vals: raw [n, S, 4]
        (or output_ch without view directions)."""
        net = self.nets[which]
        if net is None:
            raise R2LError('no fine network (N_importance = 0)')
        n = rays_o.shape[0]
        shared = z_vals.dim() == 1
        S = z_vals.shape[-1]
        m = n * S
        dev = self.device
        W = net.W
        # per-point buffers of a call, cached on the engine by (points, width) like GenericR2L._buffers (~7 GB at W = 256
        # were allocated twice per chunk): the coarse and the fine pass of a chunk each keep one set
        key = (m, W)
        buf = self._pt_buffers.get(key)
        if buf is None:
            if len(self._pt_buffers) >= 4:          # other chunk sizes seen before (ragged last chunk, another frame size): drop them
                self._pt_buffers.clear()
            buf = dict(pts=torch.empty((m, 3), dtype=torch.float32, device=dev),
                       cat=torch.empty((m, self.input_ch + W), dtype=torch.float32, device=dev),
                       views=torch.empty((m, W + self.input_ch_views), dtype=torch.float32, device=dev) if self.use_viewdirs else None,
                       work=[torch.empty((m, W), dtype=torch.float32, device=dev) for _ in range(2)])
            self._pt_buffers[key] = buf
        pts, cat, views, work = buf['pts'], buf['cat'], buf['views'], buf['work']
        raw = torch.empty((m, 4 if self.use_viewdirs else self.output_ch), dtype=torch.float32, device=dev)   # returned to the caller
        with torch.cuda.device(dev):
            check(lib().r2l_sample_points(dptr(rays_o), dptr(rays_d), n, dptr(z_vals.contiguous()), S, 0 if shared else 1, dptr(pts),
                                          current_stream()))
            self._embed(pts, self.multires, cat[:, :self.input_ch])
            if self.use_viewdirs:
                if viewdirs is None:
                    viewdirs = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)            # main.py:148-157
                dirs = viewdirs[:, None, :].expand(n, S, 3).reshape(m, 3).contiguous()       # main.py:76-77
                self._embed(dirs, self.multires_views, views[:, W:])
            net.forward(cat, views, work, raw)
        return raw.view(n, S, -1)

    def _raw4(self, raw):
        return raw if raw.shape[-1] == 4 else raw[..., :4].contiguous()       # raw2outputs reads channels 0..3 (main.py:588-600)

    def _render_rays_chunk(self, ro, rd, viewdirs):
        from .teacher import merge_sorted, raw2outputs, sample_pdf
        n = ro.shape[0]
        z = self._z_dev
        raw0 = self.run_network(0, ro, rd, z, viewdirs)
        rgb, disp, acc, weights, depth = raw2outputs(self._raw4(raw0), z.expand(n, self.N_samples), rd, white_bkgd=self.white_bkgd)
        ret = {}
        if self.N_importance > 0:
            ret.update(rgb0=rgb, disp0=disp, acc0=acc)
            z_mid = .5 * (z[1:] + z[:-1])
            z_samples = sample_pdf(z_mid.expand(n, self.N_samples - 1), weights[:, 1:-1], self.N_importance, det=True)
            z_all = merge_sorted(z.expand(n, self.N_samples), z_samples)
            raw = self.run_network(1, ro, rd, z_all, viewdirs)
            rgb, disp, acc, weights, depth = raw2outputs(self._raw4(raw), z_all, rd, white_bkgd=self.white_bkgd)
            ret.update(z_samples=z_samples, z_vals=z_all, z_std=torch.std(z_samples, dim=-1, unbiased=False), raw=raw)
        else:
            ret.update(raw=raw0)
        ret.update(rgb_map=rgb, disp_map=disp, acc_map=acc, depth_map=depth)
        return ret

    def render_rays(self, rays_o, rays_d, extras=False, perturb=0., raw_noise_std=0., pytest=False):
        if perturb > 0. or raw_noise_std > 0.:
            raise R2LError('the generic teacher path renders the test-time configuration (perturb = 0, raw_noise_std = 0); the jittered '
                           'paths exist in NeRFEngine (8 x 256)')
        ro = torch.as_tensor(rays_o).to(self.device, torch.float32).reshape(-1, 3).contiguous()
        rd = torch.as_tensor(rays_d).to(self.device, torch.float32).reshape(-1, 3).contiguous()
        vd = None
        if self.use_viewdirs:
            vd = rd / torch.norm(rd, dim=-1, keepdim=True)                   # main.py:148-157, before the NDC projection
        if self.ndc:                                                         # main.py:160-162
            from .teacher import ndc_rays
            ro, rd = ndc_rays(self.H, self.W, self.focal, 1., ro, rd)
        outs = [self._render_rays_chunk(ro[s:s + self.chunk], rd[s:s + self.chunk], None if vd is None else vd[s:s + self.chunk])
                for s in range(0, ro.shape[0], self.chunk)]
        keep = ('rgb_map', 'disp_map', 'acc_map', 'depth_map') + (tuple(k for k in outs[0] if k not in ('rgb_map', 'disp_map', 'acc_map', 'depth_map'))
                                                                   if extras else ())
        return {k: (outs[0][k] if len(outs) == 1 else torch.cat([o[k] for o in outs], 0)) for k in keep}

    def render(self, c2w, rows=None, extras=False):
        from .teacher import get_rays
        ro, rd = get_rays(self.H, self.W, self.focal, c2w, rows=None if rows is None else (int(rows[0]), int(rows[1])), device=self.device)
        return self.render_rays(ro.reshape(-1, 3), rd.reshape(-1, 3), extras=extras)
