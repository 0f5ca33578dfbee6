"""include/r2l_hip.h: the library is re-entrant per context and enqueues on the caller's stream.  Two contexts driven
from two host threads on two streams at the same time must give the results of a serial run, bit for bit (ctypes drops
the GIL inside the calls, so the C side really runs concurrently)."""
import threading

import pytest
import torch

from oracle import r2l_oracle as O

pytestmark = pytest.mark.gpu


def test_two_contexts_two_threads_two_streams(pkg):
    from efficient_nerf_amd import NeRFEngine, PREC_FP16_FP8, R2LEngine
    H, nb = 96, 6
    focal = O.focal_from_angle(H)
    sd = O.make_r2l_state(seed=4, netdepth=2 + 2 * nb)
    t1, t2 = O.make_teacher_state(1), O.make_teacher_state(2)
    poses = [torch.as_tensor(O.pose_spherical(float(t), -30., 4.))[:3, :4].float().contiguous() for t in range(0, 360, 30)]
    r2l = R2LEngine(H, H, focal, n_block=nb, precision=PREC_FP16_FP8).load_state_dict(sd)
    tea = NeRFEngine(32, 32, O.focal_from_angle(32), precision=PREC_FP16_FP8).load_state_dicts(t1, t2)
    serial_r = [r2l.render(p).clone() for p in poses]
    serial_t = [tea.render(p)['rgb_map'].clone() for p in poses]
    torch.cuda.synchronize()
    got_r, got_t, errs = [None] * len(poses), [None] * len(poses), []

    def work(fn, dst):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for rep in range(3):
                    for i, p in enumerate(poses):
                        dst[i] = fn(p)
            s.synchronize()
        except Exception as e:      # surfaced in the main thread
            errs.append(e)

    th = [threading.Thread(target=work, args=(lambda p: r2l.render(p), got_r)),
          threading.Thread(target=work, args=(lambda p: tea.render(p)['rgb_map'], got_t))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    torch.cuda.synchronize()
    for i in range(len(poses)):
        assert torch.equal(got_r[i], serial_r[i]), i
        assert torch.equal(got_t[i], serial_t[i]), i
    r2l.close()
    tea.close()
