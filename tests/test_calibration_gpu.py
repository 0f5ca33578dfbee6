"""The bf6 activation exponents of R2L_PREC_FP16_FP8 are measured on the first render's own rays (include/r2l_hip.h).
A first call with fewer rays than the sample must not freeze exponents taken from a handful of rays: the measurement
stays open and later calls only add to it."""
import pytest
import torch

from oracle import r2l_oracle as O

pytestmark = pytest.mark.gpu


def test_thin_first_call_keeps_the_measurement_open(pkg):
    from efficient_nerf_amd import PREC_FP16_FP8, R2LEngine
    H, nb = 48, 8
    focal = O.focal_from_angle(H)
    sd = O.make_r2l_state(seed=21, netdepth=2 + 2 * nb)
    gain = 1.4                               # growing activations: the ranges differ from ray to ray
    for k in sd:
        if k.startswith('body.') and k.endswith('weight'):
            sd[k] = sd[k] * gain
    c2w = O.pose_spherical(15., -40., 4.)
    ro, rd = O.get_rays(H, H, focal, c2w)
    ro, rd = ro.reshape(-1, 3).float().contiguous().cuda(), rd.reshape(-1, 3).float().contiguous().cuda()
    ref = O.r2l_render(sd, H, H, focal, c2w)
    eng = R2LEngine(H, H, focal, n_block=nb, precision=PREC_FP16_FP8).load_state_dict(sd)
    one = eng.render_rays(ro[:1].contiguous(), rd[:1].contiguous()).cpu()        # 1 ray: a thin sample
    e1 = eng.act_exponents()
    assert (one - ref[:1]).abs().max().item() <= 2e-4
    full = eng.render_rays(ro, rd).cpu()                                        # 2,304 rays: fills the sample
    e2 = eng.act_exponents()
    assert all(b >= a for a, b in zip(e1, e2)), (e1, e2)
    assert (full - ref).abs().max().item() <= 2e-4
    # closed now: other rays do not move the exponents, and a repeated call is bit-identical
    eng.render_rays(ro[500:700].contiguous(), rd[500:700].contiguous())
    assert eng.act_exponents() == e2
    assert torch.equal(eng.render_rays(ro, rd).cpu(), full)
    # a fresh context that sees the full call first ends at exponents no larger than the accumulated ones
    eng2 = R2LEngine(H, H, focal, n_block=nb, precision=PREC_FP16_FP8).load_state_dict(sd)
    eng2.render_rays(ro, rd)
    e3 = eng2.act_exponents()
    assert all(b >= a for a, b in zip(e3, e2)), (e3, e2)
    eng.close()
    eng2.close()


def test_auto_precision_follows_the_measured_ranges(pkg):
    """`--precision auto` (R2LEngine.choose_precision): fp16_fp8 for the standard synthetic W256D88 weights; for the stress set
    (body weights x 1.3), whose residual stream is too large for bf6 correction terms throughout to hold 1e-4 on rgb,
    the rungs behind the activation limits: fp16_split with a measured split, or fp16x3_asm; either way the contract holds."""
    from efficient_nerf_amd import PREC_FP16_FP8, PREC_FP16_SPLIT, PREC_FP16_SPLIT8, PREC_FP16X3_ASM, R2LEngine
    H = 48
    focal = O.focal_from_angle(H)
    c2w = O.pose_spherical(50., -30., 4.)
    for gain, want in ((1.0, 'fp16_fp8'), (1.3, 'fp16x3_asm')):
        sd = O.make_r2l_state(seed=0)
        for k in sd:
            if k.startswith('body.') and k.endswith('weight'):
                sd[k] = sd[k] * gain
        eng = R2LEngine(H, H, focal).load_state_dict(sd)
        name, top = eng.choose_precision(c2w=c2w)
        assert name == want or (want == 'fp16x3_asm' and name in ('fp16_split', 'fp16_split8') and eng.auto_split[name][eng.split_block] <= eng.AUTO_SPLIT_MAX_DIFF), (gain, name, top)
        assert eng.precision == {'fp16_fp8': PREC_FP16_FP8, 'fp16_split': PREC_FP16_SPLIT, 'fp16_split8': PREC_FP16_SPLIT8, 'fp16x3_asm': PREC_FP16X3_ASM}[name]
        assert (top <= eng.AUTO_MAX_EXP) == (want == 'fp16_fp8')
        ref = O.r2l_render(sd, H, H, focal, c2w)
        assert (eng.render(c2w).cpu() - ref).abs().max().item() <= 1e-4, gain
        eng.close()


def test_auto_precision_on_weights_the_generated_kernels_cannot_pack(pkg):
    """A layer scaled far outside the range the fp16 + residual weight split covers (max|w| = 2^8; the next layer undoes it):
    the split modes refuse the weights with a message, `--precision auto` takes the compiler-scheduled fp16x3 (per-layer
    scales) and says why; the render meets the contract."""
    from efficient_nerf_amd import PREC_FP16X3, PREC_FP16_FP8, R2LEngine, R2LError
    H = 24
    focal = O.focal_from_angle(H)
    c2w = O.pose_spherical(10., -30., 4.)
    sd = O.make_r2l_state(seed=3, netdepth=10)
    sd['body.1.body.0.weight'] = sd['body.1.body.0.weight'] * 4096.
    sd['body.1.body.0.bias'] = sd['body.1.body.0.bias'] * 4096.
    sd['body.1.body.2.weight'] = sd['body.1.body.2.weight'] / 4096.
    eng = R2LEngine(H, H, focal, n_block=4).load_state_dict(sd)
    with pytest.raises(R2LError, match='outside the range'):
        eng.set_precision(PREC_FP16_FP8)
    name, top = eng.choose_precision(c2w=c2w)
    assert (name, top) == ('fp16x3', None) and eng.precision == PREC_FP16X3 and 'outside the range' in eng.auto_note
    ref = O.r2l_render(sd, H, H, focal, c2w)
    assert (eng.render(c2w).cpu() - ref).abs().max().item() <= 1e-4
    assert eng.check_ranges() is None
    eng.close()
