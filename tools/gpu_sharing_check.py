#!/usr/bin/env python
"""Two processes time-sharing ONE GPU (what the N-rank rehearsals on a one-GPU box do; NOT the deployment, which is one process per
GPU): a heavy process loops one kind of kernel for 12 s, a light process beside it runs nerf_get_rays and a torch elementwise chain
40 times and compares both with the CPU bit for bit.
    python tools/gpu_sharing_check.py MODE      t: teacher frames   c: nerf_chain_kernel only   s: the scan kernels only
                                                g: get_rays only    r: R2L frames               m: torch fp16 matmuls
Round 4 finding (profiles/r04_gpu_sharing.txt): beside the chain kernel -- and only beside it -- a build of nerf_get_rays_kernel in
which the SLP vectorizer had formed packed-fp32 ops (v_pk_mul_f32 / v_pk_add_f32 with SGPR-pair operands) returned wrong d.x in
groups of 16 lanes; the scalar build (-fno-slp-vectorize, csrc/Makefile) does not.  Cause (profiles/r04_coresidency.txt,
tools/coresidency_probe.hip): the chain kernel leaves 112 registers per SIMD lane free, so the light kernel's waves run between its
v_mfma_f32_16x16x32_f16, beside which a v_pk_*_f32 with op_sel:[0,1] loses the hi half of src1 in lanes 48..63 -- one process with two
streams shows the same.  (GS_LIGHT_FIRST=1 starts the light process first; that order showed nothing -- the kernels did not overlap.)
The fp16x3 frame of the light process (r2l_resmlp_kernel, compiler-scheduled; GS_NO_X3=1 skips it) is compared with its own first
render.  Exit code 1 when anything differs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def worker(rank, world, mode):
    sys.path.insert(0, ROOT)
    import torch, time
    if rank == 0 and os.environ.get('GS_LIGHT_FIRST'):
        time.sleep(4)          # GS_LIGHT_FIRST=1: the light process initialises on a quiet card first -- in that order nothing was ever wrong
    import _pkg; _pkg.load()
    from efficient_nerf_amd import NeRFEngine, R2LEngine, PREC_FP16_FP8
    from efficient_nerf_amd.create_data import RandStream
    from efficient_nerf_amd.teacher import get_rays
    from oracle import r2l_oracle as O
    H = 400
    focal = O.focal_from_angle(H)
    if rank == 0:     # the heavy process: teacher chain (mode t), R2L body (mode r), or a big torch matmul loop (mode m)
        if mode in ('c', 's', 'g'):
            from efficient_nerf_amd.teacher import raw2outputs, sample_pdf, merge_sorted
            eng = NeRFEngine(H, H, focal, precision=PREC_FP16_FP8).load_state_dicts(O.make_teacher_state(1), O.make_teacher_state(2))
            pose = O.novel_poses(1)[0]
            ro, rd = get_rays(H, H, focal, pose[:3, :4], device='cuda')
            ro, rd = ro.reshape(-1, 3).contiguous(), rd.reshape(-1, 3).contiguous()
            z = torch.linspace(2., 6., 64, device='cuda').expand(ro.shape[0], 64).contiguous()
            raw = torch.randn(ro.shape[0], 64, 4, device='cuda')
            t0 = time.time()
            while time.time() - t0 < float(os.environ.get("GS_HEAVY_SECONDS", 12)):
                if mode == 'c':      # the chain kernel alone
                    eng.run_network(0, ro, rd, z)
                elif mode == 's':    # the scan kernels alone
                    rgb, disp, acc, w, depth = raw2outputs(raw, z, rd, white_bkgd=True)
                    zm = .5 * (z[:, 1:] + z[:, :-1])
                    zs = sample_pdf(zm, w[:, 1:-1], 128, det=True)
                    merge_sorted(z, zs)
                else:                # get_rays alone
                    for _ in range(50):
                        get_rays(H, H, focal, pose[:3, :4], device='cuda')
                torch.cuda.synchronize()
        elif mode == 't':
            eng = NeRFEngine(H, H, focal, precision=PREC_FP16_FP8).load_state_dicts(O.make_teacher_state(1), O.make_teacher_state(2))
            pose = O.novel_poses(1)[0][:3, :4]
            t0 = time.time()
            while time.time() - t0 < float(os.environ.get("GS_HEAVY_SECONDS", 12)):
                eng.render(pose)
                torch.cuda.synchronize()
        elif mode == 'r':
            eng = R2LEngine(800, 800, O.focal_from_angle(800), precision=PREC_FP16_FP8).load_state_dict(O.make_r2l_state(0))
            pose = O.novel_poses(1)[0][:3, :4]
            t0 = time.time()
            while time.time() - t0 < float(os.environ.get("GS_HEAVY_SECONDS", 12)):
                for _ in range(5):
                    eng.render(pose)
                torch.cuda.synchronize()
        else:
            a = torch.randn(8192, 8192, device='cuda', dtype=torch.float16)
            t0 = time.time()
            while time.time() - t0 < float(os.environ.get("GS_HEAVY_SECONDS", 12)):
                for _ in range(10):
                    b = a @ a
                torch.cuda.synchronize()
        print('heavy process done', flush=True)
        return
    # the light process: references while the card is still its own (the heavy one needs ~3 s to load), then the loop beside it
    bad_x3 = bad_x8 = 0
    if not os.environ.get('GS_NO_X3'):
        # ... and a frame of the generated kernels (head launch + r2l_body_kernel), against its own first render
        x8 = R2LEngine(200, 200, O.focal_from_angle(200), n_block=8, precision=PREC_FP16_FP8).load_state_dict(O.make_r2l_state(seed=2, netdepth=18))
        x8_ref = x8.render(O.novel_poses(3)[1][:3, :4]).clone()
        x3 = R2LEngine(200, 200, O.focal_from_angle(200), n_block=8).load_state_dict(O.make_r2l_state(seed=2, netdepth=18))   # fp16x3: compiler-scheduled
        x3_pose = O.novel_poses(3)[1][:3, :4]
        x3_ref = x3.render(x3_pose).clone()
    time.sleep(3)
    st = RandStream()
    bad_rays = bad_torch = 0
    n = 0
    xs = torch.arange(0, 160000 * 3, device='cuda', dtype=torch.float32)
    want = (torch.arange(0, 160000 * 3, dtype=torch.float32) * 1.5 + 2.0) * 0.25 - 1.0
    for k in range(40):
        p, f = st.rand_pose(), focal * st.rand_focal_scale()
        ro, rd = get_rays(H, H, f, p[:3, :4], device='cuda')
        y = (xs * 1.5 + 2.0) * 0.25 - 1.0
        cro, crd = O.get_rays(H, H, f, p[:3, :4])
        bad_rays += (rd.reshape(-1, 3).cpu() != crd.reshape(-1, 3)).any(1).sum().item()
        bad_torch += (y.cpu() != want).sum().item()
        if not os.environ.get('GS_NO_X3'):       # GS_NO_X3=1: get_rays and the torch chain only (the round-4 reproduction)
            bad_x3 += (x3.render(x3_pose) != x3_ref).any(1).sum().item()
            bad_x8 += (x8.render(x3_pose) != x8_ref).any(1).sum().item()
        n += 1
    print(f'mode {mode}: light process, {n} iterations beside the heavy one: get_rays rays differing from the CPU oracle {bad_rays}; '
          f'torch elementwise values differing from the CPU {bad_torch}; rays of a 200x200 fp16x3 frame (r2l_resmlp_kernel) differing from '
          f'its own first render {bad_x3}; of a 200x200 fp16_fp8 frame (generated head + body) {bad_x8}', flush=True)
    if bad_rays or bad_torch or bad_x3 or bad_x8:
        sys.exit(1)

if __name__ == '__main__':
    import torch.multiprocessing as mp
    if os.environ.get('GS_HEAVY_ONLY'):      # only the heavy process (tools/pk_sgpr_run.sh brings its own light one)
        worker(0, 2, sys.argv[1])
        sys.exit(0)
    mp.spawn(worker, args=(2, sys.argv[1]), nprocs=2, join=True)
