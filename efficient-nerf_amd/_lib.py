"""ctypes binding of libr2l_hip.so (include/r2l_hip.h).  Fails loudly when absent."""
import ctypes as C
import os

PREC_FP16X3 = 0
PREC_FP16X1 = 1
PREC_FP16_FP8 = 2  # fp16 main pass + both correction terms in OCP bf6 (e3m2) on the block-scaled MFMA at 4x the fp16 rate:
                   # generated head / body kernels (R2L), generated layer chain (teacher); the name is historical
PREC_FP16_E4M3 = 3  # R2L only: the same with both correction terms in OCP e4m3 (2.0 pass-equivalents, half the error)
PREC_FP16X3_ASM = 4  # R2L only: fp16x3's three fp16 passes on the generated body kernel's machine (no scales, nothing to calibrate)
PREC_FP16_SPLIT = 5  # R2L only: head + blocks [0, split) in three passes, blocks [split, n_block) with bf6 terms (r2l_set_split_block)
PREC_FP16_SPLIT8 = 6  # ... with e4m3 terms behind the split (2.0 pass-equivalents, 0.44 x the bf6 terms' error)
PREC_FP16_MIX = 7  # teacher only: fp16_fp8's layer chain with its first two trunk layers in three fp16 passes (the fine pass of trained teachers)
PRECISIONS = {'fp16x3': PREC_FP16X3, 'fp16x1': PREC_FP16X1, 'fp16_fp8': PREC_FP16_FP8, 'fp16_e4m3': PREC_FP16_E4M3,
              'fp16x3_asm': PREC_FP16X3_ASM, 'fp16_split': PREC_FP16_SPLIT, 'fp16_split8': PREC_FP16_SPLIT8, 'fp16_mix': PREC_FP16_MIX}

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('R2L_LIB_PATH', os.path.join(_HERE, 'libr2l_hip.so'))  # override: ablation builds (tools/)


class R2LError(RuntimeError):
    pass


_f = C.POINTER(C.c_float)
_vp = C.c_void_p


class RangeStatus(C.Structure):
    """include/r2l_hip.h r2l_range_status"""
    _fields_ = [('h0_max', C.c_float), ('h0_fill', C.c_float), ('worst_fill', C.c_float), ('worst_set', C.c_int),
                ('saturated', C.c_int), ('beyond_calibration', C.c_int), ('format_top', C.c_float), ('stream_max', C.c_float), ('launches', C.c_longlong),
                ('guarded_launches', C.c_longlong)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# name -> (restype, argtypes); every symbol include/r2l_hip.h declares
SIGNATURES = {
    'r2l_last_error': (C.c_char_p, []),
    'r2l_device_count': (C.c_int, []),
    'r2l_comm_available': (C.c_int, []),
    'r2l_comm_unique_id': (C.c_int, [_vp]),
    'r2l_comm_create': (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, _vp]),
    'r2l_comm_destroy': (None, [_vp]),
    'r2l_gather_image': (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    'r2l_create': (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_double, C.c_float, C.c_float,
                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'r2l_destroy': (None, [_vp]),
    'r2l_load_weights': (C.c_int, [_vp, C.POINTER(_vp), C.c_int]),
    'r2l_set_precision': (C.c_int, [_vp, C.c_int]),
    'r2l_set_split_block': (C.c_int, [_vp, C.c_int]),
    'r2l_set_activations': (C.c_int, [_vp, C.c_float, C.c_float, C.c_float]),
    'r2l_set_network_form': (C.c_int, [_vp, C.c_float, C.c_float, C.c_float, C.c_int]),
    'r2l_set_z_vals': (C.c_int, [_vp, _vp, C.c_int]),
    'r2l_render': (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    'r2l_render_rays': (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp]),
    'r2l_sample_embed': (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    'r2l_embed': (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    'r2l_debug_pack_host': (C.c_longlong, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, _vp, C.c_longlong]),
    'r2l_debug_pack_body_host': (C.c_longlong, [C.POINTER(_vp), C.c_int, C.c_int, _vp, C.c_longlong,
                                                C.POINTER(C.c_longlong)]),
    'r2l_debug_pack_body_format': (C.c_int, [C.c_int]),
    'r2l_set_act_exponents': (C.c_int, [_vp, C.POINTER(C.c_int), C.c_int]),
    'r2l_get_act_exponents': (C.c_int, [_vp, C.POINTER(C.c_int), C.c_int]),
    'r2l_set_guard_period': (C.c_int, [_vp, C.c_int]),
    'r2l_get_range_status': (C.c_int, [_vp, C.POINTER(RangeStatus), C.c_int]),
    'r2l_recalibrate': (C.c_int, [_vp, _vp]),
    'r2l_debug_body': (C.c_int, [_vp, _vp, _vp, C.c_int, _vp]),
    'r2l_debug_set_fused_tail': (C.c_int, [_vp, C.c_int]),
    'r2l_flops_per_ray': (C.c_longlong, [_vp]),
    'r2l_kernel_flops_per_ray': (C.c_longlong, [_vp]),
    'r2l_weight_image_bytes': (C.c_longlong, [_vp]),
    'r2l_rays_per_tile': (C.c_int, [_vp]),
    'r2l_timing_enable': (C.c_int, [_vp, C.c_int]),
    'r2l_kernel_time_ms': (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int]),
    'nerf_create': (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_double, C.c_float, C.c_float,
                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'nerf_destroy': (None, [_vp]),
    'nerf_load_weights': (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.c_int]),
    'nerf_set_precision': (C.c_int, [_vp, C.c_int]),
    'nerf_set_precision_pair': (C.c_int, [_vp, C.c_int, C.c_int]),
    'nerf_set_skip_rgb0': (C.c_int, [_vp, C.c_int]),
    'nerf_set_sampling': (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int]),
    'nerf_get_rays': (C.c_int, [C.c_int, C.c_int, C.c_double, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    'nerf_set_ndc': (C.c_int, [_vp, C.c_int, C.c_float]),
    'nerf_ndc_rays': (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_float, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    'nerf_debug_pack_chain_host': (C.c_longlong, [C.POINTER(_vp), C.c_int, C.c_int, _vp, C.c_longlong, C.POINTER(C.c_longlong)]),
    'nerf_run_network': (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    'nerf_sample_pdf_u': (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp]),
    'nerf_render': (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    'nerf_render_rays': (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    'nerf_last_extras': (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    'nerf_copy_extras': (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    'nerf_raw2outputs': (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    'nerf_sample_pdf': (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    'nerf_sample_pdf_ex': (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    'nerf_raw2outputs_noise': (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    'nerf_copy_extras0': (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp]),
    'nerf_render_rays_ex': (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'nerf_debug_set_split_scans': (C.c_int, [_vp, C.c_int]),
    'nerf_debug_set_x1_col_tiles': (C.c_int, [_vp, C.c_int]),
    'nerf_debug_set_x1_stream_embed': (C.c_int, [_vp, C.c_int]),
    'nerf_timing_enable': (C.c_int, [_vp, C.c_int]),
    'nerf_kernel_time_ms': (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int]),
    'r2l_np_legacy_permutation': (C.c_int, [_vp, C.POINTER(C.c_int), C.c_longlong, _vp]),
    'nerf_merge_sorted': (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp]),
    'r2l_linear_create': (C.c_int, [C.POINTER(_vp), _vp, _vp, C.c_int, C.c_int]),
    'r2l_linear_destroy': (None, [_vp]),
    'r2l_linear_forward': (C.c_int, [_vp, _vp, C.c_longlong, C.c_int, _vp, C.c_longlong, _vp, C.c_longlong, C.c_float, C.c_int, _vp,
                                     C.c_longlong, _vp]),
    'r2l_sample_points': (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp]),
    'nerf_embed': (C.c_int, [_vp, C.c_longlong, C.c_int, C.c_int, C.c_int, _vp, C.c_longlong, _vp]),
}

_lib = None


def lib():
    """The loaded library.  Raises R2LError (never falls back) when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise R2LError(f'{LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                           f'(or `make -C efficient-nerf_amd/csrc`). There is no CPU fallback.')
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:  # e.g. libamdhip64 not found
            raise R2LError(f'cannot load {LIB_PATH}: {e}') from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise R2LError(f'[{rc}] ' + lib().r2l_last_error().decode('utf-8', 'replace'))


def dptr(t):
    """data_ptr of a contiguous float32 CUDA(HIP) tensor, or None."""
    if t is None:
        return None
    import torch
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise R2LError(f'expected a contiguous float32 device tensor, got {t.dtype} on {t.device} '
                       f'(contiguous={t.is_contiguous()})')
    return C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def host_ptrs(tensors):
    """float32 contiguous CPU copies + a (void*)[] over them (keep the first alive)."""
    import torch
    keep = [t.detach().to('cpu', torch.float32).contiguous() for t in tensors]
    arr = (C.c_void_p * len(keep))(*[C.c_void_p(t.data_ptr()) for t in keep])
    return keep, arr
