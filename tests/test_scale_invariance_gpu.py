"""Per-layer weight scales of the fp16 + bf6 arithmetic (R2L body: csrc/gen/body_gen.py, teacher chain:
csrc/gen/nerf_gen.py): relu is positively homogeneous, so multiplying one layer (weights and bias) by 2^k and the next
layer's weights by 2^-k leaves the network function unchanged while the layer exponents (the E8M0 scale bytes of the
correction terms) and the activation magnitudes in between move by k.  The rendered images must stay inside the 1e-4
contract of the reference comparison (model/nerf_raybased.py:443-465, :377-401) and agree with the unscaled network."""
import pytest
import torch

from oracle import r2l_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('k', [-5, 3])
def test_r2l_body_layer_scales(pkg, k):
    from efficient_nerf_amd import PREC_FP16_FP8, R2LEngine
    H, nb = 48, 43
    focal = O.focal_from_angle(H)
    sd = O.make_r2l_state(seed=11, netdepth=2 + 2 * nb)
    sd2 = {n: v.clone() for n, v in sd.items()}
    for b in range(0, nb, 2):                      # every second block: h = relu(W1 x + b1) scaled by 2^k, W2 by 2^-k
        sd2[f'body.{b}.body.0.weight'] *= 2.0 ** k
        sd2[f'body.{b}.body.0.bias'] *= 2.0 ** k
        sd2[f'body.{b}.body.2.weight'] *= 2.0 ** -k
    c2w = O.pose_spherical(40., -20., 4.)
    ref = O.r2l_render(sd, H, H, focal, c2w)
    out = []
    for s in (sd, sd2):
        eng = R2LEngine(H, H, focal, n_block=nb, precision=PREC_FP16_FP8).load_state_dict(s)
        out.append(eng.render(c2w).cpu())
        eng.close()
    e0, e1 = (out[0] - ref).abs().max().item(), (out[1] - ref).abs().max().item()
    print(f'R2L layer scale 2^{k}: L_inf vs oracle {e0:.2e} (as is), {e1:.2e} (rescaled)')
    assert e0 <= 1e-4 and e1 <= 1e-4


@pytest.mark.parametrize('k', [-7, -4, 3, 5])
def test_teacher_chain_layer_scales(pkg, k):
    from efficient_nerf_amd import NeRFEngine, PREC_FP16_FP8
    H = 20
    focal = O.focal_from_angle(H)
    sds = [O.make_teacher_state(1), O.make_teacher_state(2)]
    scaled = []
    for sd in sds:
        s2 = {n: v.clone() for n, v in sd.items()}
        for i in (1, 3, 6):                        # pts_linears.i scaled by 2^k, its consumer by 2^-k
            s2[f'pts_linears.{i}.weight'] *= 2.0 ** k
            s2[f'pts_linears.{i}.bias'] *= 2.0 ** k
            w = s2[f'pts_linears.{i + 1}.weight']
            if i + 1 == 5:
                w[:, 63:] *= 2.0 ** -k             # layer 5 sees cat([input_pts, h]) (model/nerf_raybased.py:385)
            else:
                w *= 2.0 ** -k
        scaled.append(s2)
    c2w = O.pose_spherical(10., -35., 4.)
    ro, rd = O.get_rays(H, H, focal, c2w)
    ref = O.render_rays(sds[0], sds[1], ro.reshape(-1, 3).float(), rd.reshape(-1, 3).float(), white_bkgd=True)['rgb_map']
    out = []
    for pair in (sds, scaled):
        eng = NeRFEngine(H, H, focal, precision=PREC_FP16_FP8).load_state_dicts(pair[0], pair[1])
        out.append(eng.render(c2w)['rgb_map'].cpu())
        eng.close()
    e0, e1 = (out[0] - ref).abs().max().item(), (out[1] - ref).abs().max().item()
    print(f'teacher layer scale 2^{k}: L_inf vs oracle {e0:.2e} (as is), {e1:.2e} (rescaled)')
    assert e0 <= 1e-4 and e1 <= 1e-4


def test_r2l_exponents_are_measured_once_and_can_be_fixed(pkg):
    """r2l_get/set_act_exponents (include/r2l_hip.h): the first fp16_fp8 render after loading measures the exponents on its
    own rays; they track the activation ranges (hidden activations 2^-5 smaller -> exponents 5 lower); fixed exponents
    are what the kernel then uses, and None re-arms the measurement."""
    from efficient_nerf_amd import PREC_FP16_FP8, R2LEngine
    H, nb = 32, 5
    focal = O.focal_from_angle(H)
    sd = O.make_r2l_state(seed=3, netdepth=2 + 2 * nb)
    sd2 = {n: v.clone() for n, v in sd.items()}
    for b in range(nb):
        sd2[f'body.{b}.body.0.weight'] *= 2.0 ** -5
        sd2[f'body.{b}.body.0.bias'] *= 2.0 ** -5
        sd2[f'body.{b}.body.2.weight'] *= 2.0 ** 5
    c2w = O.pose_spherical(0., -30., 4.)
    ref = O.r2l_render(sd, H, H, focal, c2w)
    eng = R2LEngine(H, H, focal, n_block=nb, precision=PREC_FP16_FP8).load_state_dict(sd)
    assert eng.act_exponents() == [3] * (2 * nb + 1)          # defaults until a render has measured them
    img = eng.render(c2w).cpu()
    e1 = eng.act_exponents()
    assert len(e1) == 2 * nb + 1 and all(-4 <= e <= 8 for e in e1) and e1[-1] == e1[0]
    assert torch.equal(eng.render(c2w).cpu(), img)            # measured once: the second render uses the same exponents
    eng.load_state_dict(sd2)
    img2 = eng.render(c2w).cpu()
    e2 = eng.act_exponents()
    assert [a - b for a, b in zip(e1[1:-1:2], e2[1:-1:2])] == [5] * nb       # h sets
    assert e1[0:-1:2] == e2[0:-1:2]                                            # x sets unchanged
    assert (img - ref).abs().max() <= 1e-4 and (img2 - ref).abs().max() <= 1e-4
    eng.set_act_exponents([9] * (2 * nb + 1))                                  # far too coarse: the terms lose their bits
    assert eng.act_exponents()[:-1] == [9] * (2 * nb)
    coarse = (eng.render(c2w).cpu() - ref).abs().max().item()
    eng.set_act_exponents(None)
    again = eng.render(c2w).cpu()
    assert eng.act_exponents() == e2 and torch.equal(again, img2)
    print(f'exponents {e1} -> {e2}; L_inf with all exponents 9: {coarse:.2e}')
    eng.close()
