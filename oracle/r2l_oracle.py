"""CPU oracle for the R2L / NeRF-teacher ray-batched inference path.

TEST INFRASTRUCTURE ONLY.  This file is a plain PyTorch-CPU fp32 restatement of the
reference's algorithm (MingSun-Tse/Efficient-NeRF, paths cited per function as
``file:line`` relative to the reference checkout).  It is the *checker* for the HIP
path: only ``tests/``, ``__graft_entry__.smoke()``, ``bench.py``'s ``cpu_baseline``
leg and the fixture / measurement scripts under ``tools/`` (test infrastructure too: e.g.
``tools/train_like.py`` fits the trained-like fixture with autograd on these functions)
may import it.  Nothing under ``efficient-nerf_amd/`` imports it and the product
path never falls back to it.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference's own
``model/nerf_raybased.py`` and ``utils/run_nerf_raybased_helpers.py`` on CPU, runs them
on seeded inputs and commits the inputs/outputs under ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` asserts every function here reproduces those vectors
(bit-exact where the op sequence is the same, which is everywhere except the MLP
GEMMs whose summation order is the BLAS's).

All tensors are torch.float32 on CPU unless a function says otherwise.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# poses / camera (dataset/load_blender.py:10-28, 82, 106-109, 327-368)
# --------------------------------------------------------------------------------------
LEGO_CAMERA_ANGLE_X = 0.6911112070083618  # transforms_*.json of nerf_synthetic/lego


def focal_from_angle(W, camera_angle_x=LEGO_CAMERA_ANGLE_X):
    """dataset/load_blender.py:82  focal = .5 * W / np.tan(.5 * camera_angle_x)."""
    return .5 * W / np.tan(.5 * camera_angle_x)


def pose_spherical(theta, phi, radius):
    """dataset/load_blender.py:10-28 (trans_t, rot_phi, rot_theta, pose_spherical)."""
    trans_t = torch.Tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius],
                            [0, 0, 0, 1]]).float()
    ph = phi / 180. * np.pi
    rot_phi = torch.Tensor([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0],
                            [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1]]).float()
    th = theta / 180. * np.pi
    rot_theta = torch.Tensor([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0],
                              [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]]).float()
    c2w = trans_t
    c2w = rot_phi @ c2w
    c2w = rot_theta @ c2w
    c2w = torch.Tensor([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0],
                        [0, 0, 0, 1]]) @ c2w
    return c2w


def novel_poses(n_pose, phi=-30., radius=4.):
    """dataset/load_blender.py:327-333 (get_novel_poses, int case): even thetas."""
    thetas = np.linspace(-180, 180, n_pose + 1)[:-1]
    return torch.stack([pose_spherical(t, phi, radius) for t in thetas], 0)


def rand_poses(n_pose, seed=0):
    """dataset/load_blender.py:359-368 (get_rand_pose) driven by one numpy stream."""
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n_pose):
        theta = -180 + rs.rand() * 360
        phi = -90 + rs.rand() * 90
        out.append(pose_spherical(theta, phi, 4))
    return torch.stack(out, 0)


# --------------------------------------------------------------------------------------
# R2L: PointSampler / PositionalEmbedder / NeRF_v3_2  (model/nerf_raybased.py)
# --------------------------------------------------------------------------------------
def camera_dirs(H, W, focal):
    """model/nerf_raybased.py:80-86 == utils/run_nerf_raybased_helpers.py:233-240.

    dirs[h, w] = ((w - W/2)/focal, -(h - H/2)/focal, -1), no half-pixel offset."""
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W), torch.linspace(0, H - 1, H),
                          indexing='ij')
    i, j = i.t(), j.t()
    return torch.stack([(i - W * .5) / focal, -(j - H * .5) / focal,
                        -torch.ones_like(i)], dim=-1)  # [H, W, 3]


def sampler_z_vals(n_sample, near, far):
    """model/nerf_raybased.py:88-90."""
    t_vals = torch.linspace(0., 1., steps=n_sample)
    return near * (1 - t_vals) + far * (t_vals)


def rays_from_dirs(dirs, c2w):
    """model/nerf_raybased.py:95-99 / helpers:243-247: rays_d[k] = sum_j dirs[j]*c2w[k,j]."""
    rays_d = torch.sum(dirs.unsqueeze(dim=-2) * c2w[:3, :3], dim=-1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def get_rays(H, W, focal, c2w):
    """utils/run_nerf_raybased_helpers.py:231-257 (trans_origin='', focal_scale=1)."""
    return rays_from_dirs(camera_dirs(H, W, focal), c2w)


def sample_test(dirs, z_vals, c2w):
    """model/nerf_raybased.py:94-102 (PointSampler.sample_test): [H*W, n_sample*3]."""
    rays_o, rays_d = rays_from_dirs(dirs, c2w)
    rays_o, rays_d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
    return sample_rays(rays_o, rays_d, z_vals)


def sample_rays(rays_o, rays_d, z_vals):
    """model/nerf_raybased.py:114-126 (sample_train, perturb=0) == :100-102."""
    z = z_vals[None, :].expand(rays_o.shape[0], z_vals.shape[0])
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
    return pts.reshape(pts.shape[0], -1)


def positional_embed(x, L=10, include_input=True):
    """model/nerf_raybased.py:191-208 (PositionalEmbedder.__call__).

    out[r, c*(2L+1) + (l | L+l | 2L)] = (sin(x*2^l) | cos(x*2^l) | x)."""
    weights = (2**torch.linspace(0, L - 1, steps=L)).to(x.device)      # evaluated on the host as the reference does
    y = x[..., None] * weights
    y = torch.cat([torch.sin(y), torch.cos(y)], dim=-1)
    if include_input:
        y = torch.cat([y, x.unsqueeze(dim=-1)], dim=-1)
    return y.reshape(y.shape[0], -1)


def r2l_state_names(n_block=43):
    names = ['head.0.weight', 'head.0.bias']
    for i in range(n_block):
        for j in (0, 2):
            names += [f'body.{i}.body.{j}.weight', f'body.{i}.body.{j}.bias']
    names += ['tail.0.weight', 'tail.0.bias']
    return names


def make_r2l_state(seed=0, netdepth=88, netwidth=256, input_dim=1008, body_gain=1.0, inact='relu'):
    """Seeded synthetic W{netwidth}D{netdepth} state_dict with nn.Linear default init.

    Reproduces, RNG draw for RNG draw, what ``NeRF_v3_2.__init__`` does under
    ``torch.manual_seed(seed)`` (model/nerf_raybased.py:483-537): head Linear, then the
    D-2 plain Linear layers of the first ``body`` list (created and discarded when
    ``--trial.body_arch resmlp`` replaces it, :503-524), then (D-2)//2 ResMLP blocks of
    two Linear each (:443-457), then the tail Linear (:534-537).  With trial.inact = none the block's nn.Sequential holds
    no activation module and its second Linear sits at index 1, not 2 (:450-454): the keys follow."""
    j2 = 1 if inact.lower() == 'none' else 2
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    D, W = netdepth, netwidth
    sd = OrderedDict()
    head = nn.Linear(input_dim, W)
    sd['head.0.weight'], sd['head.0.bias'] = head.weight.data, head.bias.data
    for _ in range(1, D - 1):
        nn.Linear(W, W)  # first body list, discarded (same RNG consumption)
    n_block = (D - 2) // 2
    for i in range(n_block):
        l0, l2 = nn.Linear(W, W), nn.Linear(W, W)
        sd[f'body.{i}.body.0.weight'], sd[f'body.{i}.body.0.bias'] = l0.weight.data, l0.bias.data
        sd[f'body.{i}.body.{j2}.weight'], sd[f'body.{i}.body.{j2}.bias'] = l2.weight.data, l2.bias.data
    tail = nn.Linear(W, 3)
    sd['tail.0.weight'], sd['tail.0.bias'] = tail.weight.data, tail.bias.data
    torch.random.set_rng_state(g)
    if body_gain != 1.0:
        for k in sd:
            if k.startswith('body.') and k.endswith('weight'):
                sd[k] = sd[k] * body_gain
    return sd


def _activation(name):
    """model/nerf_raybased.py:468-476 get_activation: relu | lrelu (nn.LeakyReLU default slope 0.01) | none"""
    name = name.lower()
    if name == 'relu':
        return F.relu
    if name == 'lrelu':
        return F.leaky_relu
    if name == 'none':
        return lambda t: t
    raise NotImplementedError(name)


def r2l_forward(sd, x, use_residual=True, res_scale=1.0, dtype=torch.float32,
                return_layers=False, act='relu', inact='relu', outact='none'):
    """model/nerf_raybased.py:539-544 (NeRF_v3_2.forward) with ResMLP blocks :461-465.

    head: ReLU(W x + b); body: x = (W2 ReLU(W1 x + b1) + b2)*res_scale + x, no outact;
    global skip body(x)+x when use_residual; tail: sigmoid(W x + b)."""
    n_block = sum(1 for k in sd if k.endswith('body.0.weight'))
    j2 = 2 if 'body.0.body.2.weight' in sd or n_block == 0 else 1     # trial.inact = none: the second Linear is body.{i}.body.1
    c = lambda t: t.to(dtype)
    x = c(x)
    a_head, a_in, a_out = _activation(act), _activation(inact), _activation(outact)   # :497, :443-465
    h = a_head(F.linear(x, c(sd['head.0.weight']), c(sd['head.0.bias'])))
    layers = [h]
    h0 = h
    for i in range(n_block):
        t = a_in(F.linear(h, c(sd[f'body.{i}.body.0.weight']), c(sd[f'body.{i}.body.0.bias'])))
        h = a_out(F.linear(t, c(sd[f'body.{i}.body.{j2}.weight']), c(sd[f'body.{i}.body.{j2}.bias'])).mul(res_scale) + h)
        if return_layers:
            layers.append(h)
    if use_residual:
        h = h + h0
    out = torch.sigmoid(F.linear(h, c(sd['tail.0.weight']), c(sd['tail.0.bias'])))
    return (out, layers) if return_layers else out


def r2l_forward_mlp(sd, x, use_residual=True, act='relu', dtype=torch.float32):
    """NeRF_v3_2.forward with `trial.body_arch = mlp` (model/nerf_raybased.py:497-518, 539-544): head Linear + act, body
    nn.Sequential(Linear, act, Linear, act, ...) under state_dict keys body.{0,2,4,...}, global skip, sigmoid tail."""
    c = lambda t: t.to(dtype)
    a = _activation(act)
    h0 = a(F.linear(c(x), c(sd['head.0.weight']), c(sd['head.0.bias'])))
    h = h0
    k = 0
    while f'body.{k}.weight' in sd:
        h = a(F.linear(h, c(sd[f'body.{k}.weight']), c(sd[f'body.{k}.bias'])))
        k += 2
    if use_residual:
        h = h + h0
    return torch.sigmoid(F.linear(h, c(sd['tail.0.weight']), c(sd['tail.0.bias'])))


def make_r2l_mlp_state(seed=0, netdepth=8, W=256, input_dim=1008):
    """state_dict of NeRF_v3_2 with body_arch = mlp: nn.Linear default init in the constructor's order (model/nerf_raybased.py:
    497-518: head, the first body list -- built and then replaced by an identical one under the trial flags: same RNG
    consumption --, the body that is kept, tail)"""
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    sd = OrderedDict()
    lin = torch.nn.Linear(input_dim, W)
    sd['head.0.weight'], sd['head.0.bias'] = lin.weight.detach().clone(), lin.bias.detach().clone()
    for _ in range(netdepth - 2):
        torch.nn.Linear(W, W)        # the first list, discarded
    for i in range(netdepth - 2):
        lin = torch.nn.Linear(W, W)
        sd[f'body.{2 * i}.weight'], sd[f'body.{2 * i}.bias'] = lin.weight.detach().clone(), lin.bias.detach().clone()
    lin = torch.nn.Linear(W, 3)
    sd['tail.0.weight'], sd['tail.0.bias'] = lin.weight.detach().clone(), lin.bias.detach().clone()
    torch.random.set_rng_state(g)
    return sd



def _v3_2_widths(netdepth, netwidth, layerwise_netwidths=''):
    """model/nerf_raybased.py:488-493"""
    if layerwise_netwidths:
        return [int(x) for x in layerwise_netwidths.split(',')] + [3]
    return [netwidth] * (netdepth - 1) + [3]


def make_v3_2_state(seed, netdepth, netwidth, input_dim, layerwise_netwidths='', act='relu', trial=None):
    """state_dict of ANY NeRF_v3_2 the constructor builds (model/nerf_raybased.py:483-537), nn.Linear default init, RNG draw for
    RNG draw: head; the first body list (always built, :499-501); under the trial flags the body that replaces it (:503-518:
    n_block ResMLP blocks of n_learnable Linear layers at netwidth, or the plain list again); tail.  `trial`: None or a dict
    with body_arch, n_block, n_learnable, inact.  Keys follow nn.Sequential's indices (an activation module takes a slot)."""
    D, W = netdepth, netwidth
    Ws = _v3_2_widths(D, W, layerwise_netwidths)
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    sd = OrderedDict()

    def put(key, lin):
        sd[key + '.weight'], sd[key + '.bias'] = lin.weight.detach().clone(), lin.bias.detach().clone()

    put('head.0', nn.Linear(input_dim, Ws[0]))
    first = [nn.Linear(Ws[i - 1], Ws[i]) for i in range(1, D - 1)]
    step = 2 if act.lower() != 'none' else 1
    if trial is None:
        for i, lin in enumerate(first):
            put(f'body.{step * i}', lin)
    elif trial['body_arch'] == 'resmlp':
        n_block = trial.get('n_block', -1)
        if n_block <= 0:
            n_block = (D - 2) // 2
        sub_step = 1 if trial.get('inact', 'relu').lower() == 'none' else 2
        for b in range(n_block):
            for j in range(trial.get('n_learnable', 2)):
                put(f'body.{b}.body.{sub_step * j}', nn.Linear(W, W))
    else:
        for i in range(1, D - 1):
            put(f'body.{step * (i - 1)}', nn.Linear(Ws[i - 1], Ws[i]))
    put('tail.0', nn.Linear(Ws[D - 2], 3))
    torch.random.set_rng_state(g)
    return sd


def v3_2_forward(sd, x, netdepth, act='relu', use_residual=True, trial=None):
    """NeRF_v3_2.forward (model/nerf_raybased.py:539-544) for the state_dicts of make_v3_2_state: the Linear layers are taken
    in key order; ResMLP.forward (:461-465) under trial.body_arch = resmlp."""
    a = _activation(act)
    h0 = a(F.linear(x, sd['head.0.weight'], sd['head.0.bias']))
    h = h0
    if trial is not None and trial['body_arch'] == 'resmlp':
        inact, outact = _activation(trial.get('inact', 'relu')), _activation(trial.get('outact', 'none'))
        rs = trial.get('res_scale', 1.0)
        sub_step = 1 if trial.get('inact', 'relu').lower() == 'none' else 2
        b = 0
        while f'body.{b}.body.0.weight' in sd:
            t, j = h, 0
            while f'body.{b}.body.{sub_step * j}.weight' in sd:
                if j > 0:
                    t = inact(t)
                t = F.linear(t, sd[f'body.{b}.body.{sub_step * j}.weight'], sd[f'body.{b}.body.{sub_step * j}.bias'])
                j += 1
            h = outact(t.mul(rs) + h)
            b += 1
    else:
        keys = sorted({int(k.split('.')[1]) for k in sd if k.startswith('body.')})
        for k in keys:
            h = a(F.linear(h, sd[f'body.{k}.weight'], sd[f'body.{k}.bias']))
    if use_residual:
        h = h + h0
    return torch.sigmoid(F.linear(h, sd['tail.0.weight'], sd['tail.0.bias']))


def r2l_render(sd, H, W, focal, c2w, near=2., far=6., n_sample=16, L=10, chunk=40000,
               rows=None, dtype=torch.float32):
    """main.py:401-404 render_func == main.py:300-309: model(embed(sample_test(c2w)))."""
    dirs = camera_dirs(H, W, focal)
    if rows is not None:   # (r0, r1) or (r0, r1, step): a strided subset of the frame's rows (bench.py's far-pose parity)
        dirs = dirs[rows[0]:rows[1]:(rows[2] if len(rows) > 2 else 1)]
    z = sampler_z_vals(n_sample, near, far)
    pts = sample_test(dirs, z, c2w[:3, :4])
    outs = []
    with torch.no_grad():
        for s in range(0, pts.shape[0], chunk):
            outs.append(r2l_forward(sd, positional_embed(pts[s:s + chunk], L), dtype=dtype))
    return torch.cat(outs, 0).float()


# --------------------------------------------------------------------------------------
# NeRF teacher: Embedder / NeRF / run_network / raw2outputs / sample_pdf / render_rays
# --------------------------------------------------------------------------------------
def nerf_embed(x, multires):
    """utils/run_nerf_raybased_helpers.py:24-56: [x, sin(2^0 x), cos(2^0 x), ...]."""
    freq_bands = 2.**torch.linspace(0., multires - 1, steps=multires)
    outs = [x]
    for freq in freq_bands:
        outs.append(torch.sin(x * freq))
        outs.append(torch.cos(x * freq))
    return torch.cat(outs, -1)


def teacher_state_names():
    names = []
    for i in range(8):
        names += [f'pts_linears.{i}.weight', f'pts_linears.{i}.bias']
    names += ['views_linears.0.weight', 'views_linears.0.bias', 'feature_linear.weight',
              'feature_linear.bias', 'alpha_linear.weight', 'alpha_linear.bias',
              'rgb_linear.weight', 'rgb_linear.bias']
    return names


def make_teacher_state(seed=0, D=8, W=256, input_ch=63, input_ch_views=27, skips=(4,),
                       sigma_bias_shift=0.5):
    """Seeded synthetic NeRF(D=8,W=256,use_viewdirs) state_dict, nn.Linear default init,
    module creation order of model/nerf_raybased.py:357-375."""
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    sd = OrderedDict()
    lins = [nn.Linear(input_ch, W)] + [
        nn.Linear(W, W) if i not in skips else nn.Linear(W + input_ch, W) for i in range(D - 1)]
    views = nn.Linear(input_ch_views + W, W // 2)
    feature, alpha, rgb = nn.Linear(W, W), nn.Linear(W, 1), nn.Linear(W // 2, 3)
    for i, l in enumerate(lins):
        sd[f'pts_linears.{i}.weight'], sd[f'pts_linears.{i}.bias'] = l.weight.data, l.bias.data
    sd['views_linears.0.weight'], sd['views_linears.0.bias'] = views.weight.data, views.bias.data
    sd['feature_linear.weight'], sd['feature_linear.bias'] = feature.weight.data, feature.bias.data
    sd['alpha_linear.weight'], sd['alpha_linear.bias'] = alpha.weight.data, alpha.bias.data + sigma_bias_shift
    sd['rgb_linear.weight'], sd['rgb_linear.bias'] = rgb.weight.data, rgb.bias.data
    torch.random.set_rng_state(g)
    return sd


def teacher_forward(sd, x, input_ch=63, skips=(4,), dtype=torch.float32):
    """model/nerf_raybased.py:377-401 (NeRF.forward, use_viewdirs=True)."""
    c = lambda t: t.to(dtype)
    x = c(x)
    input_pts, input_views = x[..., :input_ch], x[..., input_ch:]
    h = input_pts
    for i in range(8):
        h = F.relu(F.linear(h, c(sd[f'pts_linears.{i}.weight']), c(sd[f'pts_linears.{i}.bias'])))
        if i in skips:
            h = torch.cat([input_pts, h], -1)
    alpha = F.linear(h, c(sd['alpha_linear.weight']), c(sd['alpha_linear.bias']))
    feature = F.linear(h, c(sd['feature_linear.weight']), c(sd['feature_linear.bias']))
    h = torch.cat([feature, input_views], -1)
    h = F.relu(F.linear(h, c(sd['views_linears.0.weight']), c(sd['views_linears.0.bias'])))
    rgb = F.linear(h, c(sd['rgb_linear.weight']), c(sd['rgb_linear.bias']))
    return torch.cat([rgb, alpha], -1)


def run_network(sd, pts, viewdirs, multires=10, multires_views=4, netchunk=1024 * 64,
                dtype=torch.float32):
    """main.py:65-87 (run_network): embed pts + expanded viewdirs, MLP in netchunk slices."""
    inputs_flat = pts.reshape(-1, 3)
    embedded = nerf_embed(inputs_flat, multires)
    input_dirs = viewdirs[:, None].expand(pts.shape).reshape(-1, 3)
    embedded = torch.cat([embedded, nerf_embed(input_dirs, multires_views)], -1)
    outs = [teacher_forward(sd, embedded[i:i + netchunk], dtype=dtype)
            for i in range(0, embedded.shape[0], netchunk)]
    return torch.cat(outs, 0).reshape(list(pts.shape[:-1]) + [4]).float()


def raw_noise(shape, raw_noise_std, pytest=False):
    """main.py:592-598: randn * std, or with pytest the numpy stream np.random.seed(0); rand(*shape) * std"""
    if pytest:
        np.random.seed(0)
        return torch.Tensor(np.random.rand(*list(shape)) * raw_noise_std)
    return torch.randn(shape) * raw_noise_std


def raw2outputs(raw, z_vals, rays_d, white_bkgd=False, raw_noise_std=0., pytest=False, noise=None):
    """main.py:556-621: alpha-compositing along the ray (`noise`: explicit [n,S] tensor instead of a draw)."""
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.Tensor([1e10]).to(dists.device).expand(dists[..., :1].shape)], -1)
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    if noise is None:
        noise = raw_noise(raw[..., 3].shape, raw_noise_std, pytest) if raw_noise_std > 0. else 0.
    alpha = 1. - torch.exp(-F.relu(raw[..., 3] + noise) * dists)
    weights = alpha * torch.cumprod(
        torch.cat([torch.ones((alpha.shape[0], 1), device=alpha.device), 1. - alpha + 1e-10], -1), -1)[:, :-1]
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    depth_map = torch.sum(weights * z_vals, -1)
    disp_map = 1. / torch.max(1e-10 * torch.ones_like(depth_map),
                              depth_map / torch.sum(weights, -1))
    acc_map = torch.sum(weights, -1)
    if white_bkgd:
        rgb_map = rgb_map + (1. - acc_map[..., None])
    return rgb_map, disp_map, acc_map, weights, depth_map


def sample_pdf_u(shape, N_samples, det=True, pytest=False):
    """the uniforms of helpers:293-307: linspace (det) / torch.rand, or with pytest the numpy stream"""
    new_shape = list(shape) + [N_samples]
    if pytest:
        np.random.seed(0)
        u = np.broadcast_to(np.linspace(0., 1., N_samples), new_shape) if det else np.random.rand(*new_shape)
        return torch.Tensor(u)
    if det:
        return torch.linspace(0., 1., steps=N_samples).expand(new_shape)
    return torch.rand(new_shape)


def sample_pdf(bins, weights, N_samples, det=True, pytest=False, u=None, taps=False):
    """utils/run_nerf_raybased_helpers.py:283-330.  `u`: explicit [n, N_samples] uniforms instead of a draw;
    taps=True also returns (cdf, inds)."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if u is None:
        u = sample_pdf_u(cdf.shape[:-1], N_samples, det, pytest)
    u = u.to(cdf.device).expand(list(cdf.shape[:-1]) + [N_samples]).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.max(torch.zeros_like(inds - 1), inds - 1)
    above = torch.min((cdf.shape[-1] - 1) * torch.ones_like(inds), inds)
    inds_g = torch.stack([below, above], -1)
    matched_shape = [inds_g.shape[0], inds_g.shape[1], cdf.shape[-1]]
    cdf_g = torch.gather(cdf.unsqueeze(1).expand(matched_shape), 2, inds_g)
    bins_g = torch.gather(bins.unsqueeze(1).expand(matched_shape), 2, inds_g)
    denom = (cdf_g[..., 1] - cdf_g[..., 0])
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_g[..., 0]) / denom
    samples = bins_g[..., 0] + t * (bins_g[..., 1] - bins_g[..., 0])
    return (samples, cdf, inds) if taps else samples


def perturb_z_vals(z_vals, pytest=False, t_rand=None):
    """main.py:684-699: stratified jitter of the coarse depths"""
    mids = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
    upper = torch.cat([mids, z_vals[..., -1:]], -1)
    lower = torch.cat([z_vals[..., :1], mids], -1)
    if t_rand is None:
        t_rand = torch.rand(z_vals.shape, device=z_vals.device)
        if pytest:
            np.random.seed(0)
            t_rand = torch.Tensor(np.random.rand(*list(z_vals.shape)))
    return lower + (upper - lower) * t_rand


def coarse_z_vals(near, far, N_samples, n_rays):
    """main.py:676-682 (lindisp=False, perturb=0)."""
    t_vals = torch.linspace(0., 1., steps=N_samples)
    z_vals = near * (1. - t_vals) + far * (t_vals)
    return z_vals.expand([n_rays, N_samples])


def merge_z(z_vals, z_samples):
    """main.py:730-732: sort(cat(coarse, fine))."""
    return torch.sort(torch.cat([z_vals, z_samples], -1), -1)[0]


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """utils/run_nerf_raybased_helpers.py:260-279: origins moved to the near plane, then the
    perspective projection of origin and direction (Python-float constants times f32 tensors)."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    p = rays_o + t[..., None] * rays_d
    sx, sy = -1. / (W / (2. * focal)), -1. / (H / (2. * focal))
    o = torch.stack([sx * p[..., 0] / p[..., 2], sy * p[..., 1] / p[..., 2], 1. + 2. * near / p[..., 2]], -1)
    d = torch.stack([sx * (rays_d[..., 0] / rays_d[..., 2] - p[..., 0] / p[..., 2]),
                     sy * (rays_d[..., 1] / rays_d[..., 2] - p[..., 1] / p[..., 2]),
                     -2. * near / p[..., 2]], -1)
    return o, d


def render_rays(sd_coarse, sd_fine, rays_o, rays_d, near=2., far=6., N_samples=64,
                N_importance=128, white_bkgd=True, dtype=torch.float32, viewdirs=None, lindisp=False,
                perturb=0., raw_noise_std=0., pytest=False):
    """main.py:624-756 (render_rays) + main.py:148-157 (viewdirs)."""
    if viewdirs is None:
        viewdirs = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    n = rays_o.shape[0]
    near_t = near * torch.ones_like(rays_d[..., :1])
    far_t = far * torch.ones_like(rays_d[..., :1])
    t_vals = torch.linspace(0., 1., steps=N_samples).to(rays_d.device)      # the host's linspace, as in the reference
    if not lindisp:
        z_vals = near_t * (1. - t_vals) + far_t * (t_vals)  # [n, N_samples] (main.py:673-682)
    else:
        z_vals = 1. / (1. / near_t * (1. - t_vals) + 1. / far_t * (t_vals))
    if perturb > 0.:
        z_vals = perturb_z_vals(z_vals, pytest)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
    raw0 = run_network(sd_coarse, pts, viewdirs, dtype=dtype)
    rgb0, disp0, acc0, weights0, depth0 = raw2outputs(raw0, z_vals, rays_d, white_bkgd, raw_noise_std, pytest)
    z_mid = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
    z_samples = sample_pdf(z_mid, weights0[..., 1:-1], N_importance, det=(perturb == 0.), pytest=pytest)
    z_samples = z_samples.detach()       # main.py:729 (no gradient through the sample positions; a no-op for inference)
    z_all = merge_z(z_vals, z_samples)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_all[..., :, None]
    raw = run_network(sd_fine, pts, viewdirs, dtype=dtype)
    rgb, disp, acc, weights, depth = raw2outputs(raw, z_all, rays_d, white_bkgd, raw_noise_std, pytest)
    return dict(rgb_map=rgb, disp_map=disp, acc_map=acc, depth_map=depth, rgb0=rgb0,
                disp0=disp0, acc0=acc0, weights0=weights0, z_samples=z_samples, z_vals=z_all,
                raw0=raw0, raw=raw, z_std=torch.std(z_samples, dim=-1, unbiased=False))


def render_rays_taps(sd_coarse, sd_fine, rays_o, rays_d, dtype=torch.float32, near=2., far=6., N_samples=64, N_importance=128,
                     white_bkgd=True, z_samples=None):
    """render_rays (perturb = 0) with EVERY step evaluated in `dtype` and the sample_pdf taps returned: float32 = render_rays'
    arithmetic; float64 = the same functions on the same float32 inputs (rays, the host's float32 linspace values of t and u) with
    ~1e-16 rounding, the stand-in for exact arithmetic the whole-frame classification measures both the fp32 reference and the HIP
    path against (tools/teacher_whole_frame.py).  `z_samples` (optional, [n, N_importance]) replaces sample_pdf's output: the fine
    pass at GIVEN sample positions.  main.py:624-756, helpers:283-330."""
    ro, rd = rays_o.to(dtype), rays_d.to(dtype)
    viewdirs = rd / torch.norm(rd, dim=-1, keepdim=True)
    t_vals = torch.linspace(0., 1., steps=N_samples).to(rd.device, dtype)
    z_vals = (near * torch.ones_like(rd[..., :1])) * (1. - t_vals) + (far * torch.ones_like(rd[..., :1])) * t_vals

    def net(sd, pts):
        e = torch.cat([nerf_embed(pts.reshape(-1, 3), 10), nerf_embed(viewdirs[:, None].expand(pts.shape).reshape(-1, 3), 4)], -1)
        return teacher_forward(sd, e, dtype=dtype).reshape(list(pts.shape[:-1]) + [4])

    raw0 = net(sd_coarse, ro[..., None, :] + rd[..., None, :] * z_vals[..., :, None])
    rgb0, disp0, acc0, weights0, depth0 = raw2outputs(raw0, z_vals, rd, white_bkgd)
    z_mid = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
    u = torch.linspace(0., 1., steps=N_importance).to(rd.device, dtype).expand(ro.shape[0], N_importance)
    zs, cdf, inds = sample_pdf(z_mid, weights0[..., 1:-1], N_importance, u=u, taps=True)
    below, above = (inds - 1).clamp(min=0), inds.clamp(max=cdf.shape[-1] - 1)
    denom = torch.gather(cdf, 1, above) - torch.gather(cdf, 1, below)
    if z_samples is not None:
        zs = z_samples.to(dtype)
    z_all = merge_z(z_vals, zs)
    raw = net(sd_fine, ro[..., None, :] + rd[..., None, :] * z_all[..., :, None])
    rgb, disp, acc, weights, depth = raw2outputs(raw, z_all, rd, white_bkgd)
    return dict(rgb_map=rgb, disp_map=disp, acc_map=acc, depth_map=depth, rgb0=rgb0, acc0=acc0, weights0=weights0, cdf=cdf, inds=inds,
                denom=denom, u=u, z_samples=zs, z_vals=z_all, raw0=raw0, raw=raw)


def teacher_render(sd_coarse, sd_fine, H, W, focal, c2w, rows=None, chunk=4096, ndc=False, **kw):
    """main.py:107-186 (render, c2w given, use_viewdirs=True) over a row range; ndc=True projects the
    rays with ndc_rays(H, W, focal, 1., ...) after the view directions were taken (main.py:148-162)."""
    rays_o, rays_d = get_rays(H, W, focal, c2w[:3, :4])
    if rows is not None:
        rays_o, rays_d = rays_o[rows[0]:rows[1]], rays_d[rows[0]:rows[1]]
    rays_o, rays_d = rays_o.reshape(-1, 3).float(), rays_d.reshape(-1, 3).float()
    vd = None
    if ndc:
        vd = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
        rays_o, rays_d = ndc_rays(H, W, focal, 1., rays_o, rays_d)
    outs = []
    with torch.no_grad():
        for s in range(0, rays_o.shape[0], chunk):
            outs.append(render_rays(sd_coarse, sd_fine, rays_o[s:s + chunk], rays_d[s:s + chunk],
                                    viewdirs=None if vd is None else vd[s:s + chunk], **kw))
    return {k: torch.cat([o[k] for o in outs], 0) for k in outs[0]}


# --------------------------------------------------------------------------------------
# metrics (utils/run_nerf_raybased_helpers.py:19-20)

# -- any NeRF the reference's constructor builds (netdepth / netwidth / multires / use_viewdirs / N_importance variants) -----
def make_nerf_state(seed, D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=(4,), use_viewdirs=True,
                    sigma_bias_shift=0.5):
    """state_dict of NeRF(D, W, input_ch, input_ch_views, output_ch, skips, use_viewdirs) with nn.Linear default init in the
    module creation order of model/nerf_raybased.py:357-375: pts_linears, views_linears (always built), then feature / alpha /
    rgb under use_viewdirs, output_linear otherwise.  The density bias is shifted so the scan paths see non-trivial weights."""
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    sd = OrderedDict()

    def put(key, lin):
        sd[key + '.weight'], sd[key + '.bias'] = lin.weight.detach().clone(), lin.bias.detach().clone()

    lins = [nn.Linear(input_ch, W)] + [nn.Linear(W, W) if i not in skips else nn.Linear(W + input_ch, W) for i in range(D - 1)]
    views = nn.Linear(input_ch_views + W, W // 2)
    for i, l in enumerate(lins):
        put(f'pts_linears.{i}', l)
    put('views_linears.0', views)
    if use_viewdirs:
        feature, alpha, rgb = nn.Linear(W, W), nn.Linear(W, 1), nn.Linear(W // 2, 3)
        put('feature_linear', feature), put('alpha_linear', alpha), put('rgb_linear', rgb)
        sd['alpha_linear.bias'] = sd['alpha_linear.bias'] + sigma_bias_shift
    else:
        put('output_linear', nn.Linear(W, output_ch))
        sd['output_linear.bias'][3] += sigma_bias_shift
    torch.random.set_rng_state(g)
    return sd


def nerf_forward(sd, x, input_ch=63, skips=(4,), use_viewdirs=True):
    """NeRF.forward (model/nerf_raybased.py:377-401) for any depth: the pts_linears are counted from the state_dict"""
    input_pts, input_views = x[..., :input_ch], x[..., input_ch:]
    h = input_pts
    i = 0
    while f'pts_linears.{i}.weight' in sd:
        h = F.relu(F.linear(h, sd[f'pts_linears.{i}.weight'], sd[f'pts_linears.{i}.bias']))
        if i in skips:
            h = torch.cat([input_pts, h], -1)
        i += 1
    if not use_viewdirs:
        return F.linear(h, sd['output_linear.weight'], sd['output_linear.bias'])
    alpha = F.linear(h, sd['alpha_linear.weight'], sd['alpha_linear.bias'])
    feature = F.linear(h, sd['feature_linear.weight'], sd['feature_linear.bias'])
    h = torch.cat([feature, input_views], -1)
    h = F.relu(F.linear(h, sd['views_linears.0.weight'], sd['views_linears.0.bias']))
    return torch.cat([F.linear(h, sd['rgb_linear.weight'], sd['rgb_linear.bias']), alpha], -1)


def run_network_generic(sd, pts, viewdirs, multires=10, multires_views=4, i_embed=0, use_viewdirs=True, skips=(4,)):
    """main.py:65-87 with get_embedder(multires, i_embed) (helpers:59-74: i_embed = -1 is the identity)"""
    flat = pts.reshape(-1, 3)
    emb = flat if i_embed == -1 else nerf_embed(flat, multires)
    input_ch = emb.shape[-1]
    if use_viewdirs:
        dirs = viewdirs[:, None].expand(pts.shape).reshape(-1, 3)
        emb = torch.cat([emb, dirs if i_embed == -1 else nerf_embed(dirs, multires_views)], -1)
    out = nerf_forward(sd, emb, input_ch, skips, use_viewdirs)
    return out.reshape(list(pts.shape[:-1]) + [out.shape[-1]])


def render_rays_generic(sd_coarse, sd_fine, rays_o, rays_d, near=2., far=6., N_samples=64, N_importance=128, white_bkgd=True,
                        lindisp=False, viewdirs=None, **net):
    """main.py:624-756 at test time (perturb = 0, raw_noise_std = 0) for any network create_nerf builds (main.py:407-453):
    N_importance = 0 ends behind the coarse pass; network_fine = None runs the coarse network twice (:737)."""
    use_viewdirs = net.get('use_viewdirs', True)
    if viewdirs is None and use_viewdirs:
        viewdirs = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    near_t, far_t = near * torch.ones_like(rays_d[..., :1]), far * torch.ones_like(rays_d[..., :1])
    t_vals = torch.linspace(0., 1., steps=N_samples)
    z_vals = near_t * (1. - t_vals) + far_t * (t_vals) if not lindisp else 1. / (1. / near_t * (1. - t_vals) + 1. / far_t * (t_vals))
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
    raw = run_network_generic(sd_coarse, pts, viewdirs, **net)
    rgb, disp, acc, weights, depth = raw2outputs(raw, z_vals, rays_d, white_bkgd)
    ret = dict(raw0=raw)
    if N_importance > 0:
        ret.update(rgb0=rgb, disp0=disp, acc0=acc)
        z_mid = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
        z_samples = sample_pdf(z_mid, weights[..., 1:-1], N_importance, det=True)
        z_vals = merge_z(z_vals, z_samples)
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
        raw = run_network_generic(sd_coarse if sd_fine is None else sd_fine, pts, viewdirs, **net)
        rgb, disp, acc, weights, depth = raw2outputs(raw, z_vals, rays_d, white_bkgd)
        ret.update(z_samples=z_samples, z_std=torch.std(z_samples, dim=-1, unbiased=False))
    ret.update(rgb_map=rgb, disp_map=disp, acc_map=acc, depth_map=depth, raw=raw, z_vals=z_vals)
    return ret

# --------------------------------------------------------------------------------------
def mse2psnr(mse):
    return -10. * math.log10(max(float(mse), 1e-30))


def psnr(a, b):
    return mse2psnr(torch.mean((a.double() - b.double())**2))


def perturbed_state(sd, seed=1234, rel=0.05):
    """Stand-in ground truth for the PSNR-delta measurement (no real GT exists offline): the same network
    with every weight matrix multiplied element-wise by 1 + rel * N(0, 1), fixed seed.  Its render plays the role the
    ground-truth image plays in the reference's mse2psnr(img2mse(rgb, gt)) (utils/run_nerf_raybased_helpers.py:19-20)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        out[k] = v * (1 + rel * torch.randn(v.shape, generator=g)) if k.endswith('weight') else v.clone()
    return out


def redistributed_state(sd, kind, seed=5, body_gain=1.0):
    """Synthetic weights that do not look like nn.Linear's uniform init (no trained checkpoint exists offline):
    every weight matrix of `sd` replaced by one of the same scale with another distribution -- 'laplace' (heavy tails, same
    standard deviation), 'sparse' (half the weights zero, the rest x sqrt 2), 'outlier' (0.05 % of the weights x 12) --,
    'uniform' keeps it; then every body weight x body_gain (the lever that moves the activation range).  Test / tool
    infrastructure: tests/test_r2l_gpu.py, tools/range_sweep_dists.py."""
    gen = torch.Generator().manual_seed(seed)
    out = {}
    for k, w in sd.items():
        if not k.endswith('weight') or kind == 'uniform':
            out[k] = w.clone()
        elif kind == 'laplace':
            u = torch.rand(w.shape, generator=gen) - 0.5
            out[k] = (-torch.sign(u) * torch.log1p(-2 * u.abs()) * w.std() / np.sqrt(2)).float()
        elif kind == 'sparse':
            out[k] = w * (torch.rand(w.shape, generator=gen) < 0.5).float() * float(np.sqrt(2))
        elif kind == 'outlier':
            w2 = w.clone()
            w2.view(-1)[torch.randint(0, w.numel(), (max(1, w.numel() // 2000),), generator=gen)] *= 12.0
            out[k] = w2
        else:
            raise ValueError(kind)
        if k.startswith('body.') and k.endswith('weight') and body_gain != 1.0:
            out[k] = out[k] * body_gain
    return out
