"""
ctypes binding of libcosmomap2_hip.so (the C ABI in include/cosmomap2.h).

There is NO CPU fallback: if the shared library has not been built
(`python -m cosmomap2_amd.build`) or no GPU is present, the first call that needs
the device raises.  Pointers are passed as integers (torch ``data_ptr()``).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CM2_LIB_PATH: load another build of the same library (kernel experiments: several builds of one
# source with different -D switches, timed in one process each)
LIB_PATH = os.environ.get("CM2_LIB_PATH") or os.path.join(_HERE, "libcosmomap2_hip.so")

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_dbl = ctypes.c_double

# name -> argtypes (restype is int unless listed in _RESTYPE)
PROTOTYPES = {
    "cm2_last_error": [],
    "cm2_abi_version": [],
    "cm2_device_info": [_int, ctypes.c_char_p, ctypes.POINTER(_int), ctypes.POINTER(_dbl)],
    "cm2_release_cached_memory": [],
    "cm2_device_memory_info": [ctypes.POINTER(_i64)],
    "cm2_set_exact_order": [_int],
    "cm2_pointing_create": [ctypes.POINTER(_vp), _vp, _vp, _vp, _i64, _i64, _int, _vp],
    "cm2_pointing_destroy": [_vp],
    "cm2_pointing_info": [_vp, ctypes.POINTER(_i64)],
    "cm2_pointing_build_sell": [_vp, _vp],
    "cm2_P_apply": [_vp, _vp, _vp, _vp],
    "cm2_Pt_apply": [_vp, _vp, _vp, _vp],
    "cm2_pointing_set_weights": [_vp, _vp, _vp],
    "cm2_PtNP_diag_apply": [_vp, _vp, _vp, _vp],
    "cm2_tiles_create": [ctypes.POINTER(_vp), _vp, _vp, _vp, _i64, _i64, _int, _int, _i64, _vp],
    "cm2_tiles_destroy": [_vp],
    "cm2_tiles_info": [_vp, ctypes.POINTER(_i64)],
    "cm2_tiles_pixel_range": [_vp, _i64, _i64, ctypes.POINTER(_i64)],
    "cm2_tiles_group_tiles": [_vp, _int, ctypes.POINTER(_i64)],
    "cm2_tiles_set_pt_order": [_vp, _int],
    "cm2_tiles_prepare_pt": [_vp, _vp],
    "cm2_tiles_pt_parts": [_vp, ctypes.POINTER(_i64)],
    "cm2_P_tiles_apply": [_vp, _vp, _vp, _vp],
    "cm2_Pt_tiles_apply": [_vp, _vp, _vp, _vp],
    "cm2_i32_time_to_tiles": [_vp, _vp, _vp, _vp],
    "cm2_Pt_tiles_apply_range": [_vp, _vp, _vp, _i64, _i64, _vp],
    "cm2_tod_time_to_tiles": [_vp, _vp, _vp, _vp],
    "cm2_tod_tiles_to_time": [_vp, _vp, _vp, _vp],
    "cm2_noise_create_diag": [ctypes.POINTER(_vp), ctypes.POINTER(_dbl), ctypes.POINTER(_i64), _i64],
    "cm2_noise_create_toeplitz": [ctypes.POINTER(_vp), ctypes.POINTER(_dbl), _i64,
                                  ctypes.POINTER(_i64), _i64, _int, _vp],
    "cm2_noise_destroy": [_vp],
    "cm2_noise_apply": [_vp, _vp, _vp, _vp],
    "cm2_noise_apply_tiles": [_vp, _vp, _vp, _vp, _vp],
    "cm2_noise_prepare_tiles": [_vp, _vp, _vp],
    "cm2_noise_expand_diag": [_vp, _vp, _vp],
    "cm2_noise_info": [_vp, ctypes.POINTER(_i64)],
    "cm2_noise_tile_kernel_info": [_vp, ctypes.POINTER(_i64), ctypes.POINTER(_dbl)],
    "cm2_weights_accumulate": [_int, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cm2_pixel_mask": [_int, _i64, _vp, _vp, _vp, _vp, _dbl, _vp, _vp],
    "cm2_pixel_compact": [_i64, _vp, _vp, ctypes.POINTER(_i64), _vp],
    "cm2_compact_f64": [_i64, _vp, _vp, _vp, _vp],
    "cm2_compact_i64": [_i64, _vp, _vp, _vp, _vp],
    "cm2_flag_samples": [_i64, _vp, _vp, _vp],
    "cm2_bd_det_mask": [_int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cm2_bdprecond_apply": [_int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cm2_bd_apply": [_int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cm2_reduce_work_doubles": [],
    "cm2_dot": [_i64, _vp, _vp, _vp, _vp, _vp],
    "cm2_axpy": [_i64, _dbl, _vp, _vp, _vp],
    "cm2_scal": [_i64, _dbl, _vp, _vp],
    "cm2_xmy": [_i64, _vp, _vp, _vp, _vp],
    "cm2_pcg_update_p": [_i64, _vp, _vp, _vp, _vp, _vp],
    "cm2_pcg_update_xr": [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cm2_Zt_apply": [_i64, _int, _vp, _vp, _vp, _vp, _vp],
    "cm2_pcg": [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _int, _dbl, _dbl, _i64, _vp, _vp,
                ctypes.POINTER(_i64), ctypes.POINTER(_int), _vp],
    "cm2_pcg_sharded": [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _int, _dbl, _dbl, _i64, _vp, _vp, _int, _vp, _vp,
                        ctypes.POINTER(_i64), ctypes.POINTER(_int), _vp],
    "cm2_arnoldi": [_i64, _vp, _vp, _vp, _vp, _dbl, _int, _vp, ctypes.POINTER(_dbl),
                    ctypes.POINTER(_int), _vp],
    "cm2_PtNP_tiles_apply": [_vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cm2_Z_apply": [_i64, _int, _vp, _vp, _vp, _vp],
    "cm2_Z_axpy": [_i64, _int, _vp, _vp, _dbl, _vp, _vp],
    "cm2_gemm_tn_work_doubles": [_int, _int],
    "cm2_gemm_tn": [_i64, _int, _int, _vp, _vp, _vp, _vp, _vp],
    "cm2_small_matvec": [_int, _vp, _vp, _vp, _vp],
    "cm2_gemm_atbt": [_i64, _i64, _i64, _vp, _vp, _vp, _vp],
    "cm2_transpose": [_i64, _i64, _vp, _vp, _vp],
    "cm2_panel_gemm": [_i64, _int, _int, _vp, _vp, _vp, _int, _vp],
    "cm2_cos_sin_2phi": [_i64, _vp, _vp, _vp, _vp],
    "cm2_m2_finish": [_int, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                      _vp, _vp, _vp],
    "cm2_filter_create": [ctypes.POINTER(_vp), _i64, _i64, _vp, _vp, _vp, _int, _vp, _vp, _i64,
                          _vp],
    "cm2_filter_destroy": [_vp],
    "cm2_filter_info": [_vp, _vp],
    "cm2_filter_apply": [_vp, _vp, _vp, _vp],
    "cm2_filter_apply_tiles": [_vp, _vp, _vp, _vp, _vp, _vp],
    "cm2_cutsky_to_fullsky": [_int, _i64, _vp, _vp, _i64, _vp, _vp],
    "cm2_fullsky_to_cutsky": [_int, _i64, _vp, _vp, _i64, _vp, _vp],
    "cm2_ground_bin_sums": [_i64, _int, _vp, _vp, _vp, _vp],
    "cm2_ground_subtract": [_i64, _vp, _vp, _vp, _vp, _vp],
}
_RESTYPE = {"cm2_last_error": ctypes.c_char_p, "cm2_reduce_work_doubles": _i64,
            "cm2_gemm_tn_work_doubles": _i64}

_lib = None


class HipError(RuntimeError):
    pass


def load():
    """dlopen the library and bind every symbol of include/cosmomap2.h."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipError(
                "libcosmomap2_hip.so is not built (%s); run `python -m cosmomap2_amd.build`. "
                "There is no CPU fallback for the map-making kernels." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, args in PROTOTYPES.items():
            fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
            fn.argtypes = args
            fn.restype = _RESTYPE.get(name, _int)
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        msg = load().cm2_last_error()
        raise HipError(msg.decode() if msg else "libcosmomap2_hip call failed (rc=%d)" % rc)


def _free_torch_cache():
    """torch's caching allocator and the library's block cache share the device and do not see each
    other's idle memory (device._alloc handles the other direction)."""
    try:
        import torch
        if not torch.cuda.is_available():
            return False
        torch.cuda.empty_cache()
        return True
    except Exception:          # pragma: no cover
        return False


ERR_OUT_OF_MEMORY = 3          # CM2_ERR_OUT_OF_MEMORY of include/cosmomap2.h

# Entry points that may be called again after an out-of-memory failure: they build a new object (a failed
# build frees what it had made) or overwrite their outputs from their inputs.  NOT in the list: the in-place
# updates (cm2_axpy, cm2_scal, cm2_Z_axpy, cm2_panel_gemm with accumulate, cm2_pcg_update_*,
# cm2_flag_samples, cm2_compact_*), which a second run would apply twice, and the drivers with callbacks
# (cm2_pcg, cm2_pcg_sharded, cm2_arnoldi).
RESTARTABLE = frozenset([
    "cm2_pointing_create", "cm2_pointing_build_sell", "cm2_pointing_set_weights",
    "cm2_P_apply", "cm2_Pt_apply", "cm2_PtNP_diag_apply",
    "cm2_tiles_create", "cm2_tiles_prepare_pt", "cm2_P_tiles_apply", "cm2_Pt_tiles_apply",
    "cm2_Pt_tiles_apply_range", "cm2_i32_time_to_tiles", "cm2_tod_time_to_tiles", "cm2_tod_tiles_to_time",
    "cm2_noise_create_diag", "cm2_noise_create_toeplitz", "cm2_noise_apply", "cm2_noise_apply_tiles",
    "cm2_noise_prepare_tiles", "cm2_noise_expand_diag", "cm2_PtNP_tiles_apply",
    "cm2_weights_accumulate", "cm2_pixel_mask", "cm2_pixel_compact",
    "cm2_bd_det_mask", "cm2_bdprecond_apply", "cm2_bd_apply",
    "cm2_dot", "cm2_xmy", "cm2_Zt_apply", "cm2_Z_apply", "cm2_gemm_tn", "cm2_small_matvec", "cm2_gemm_atbt",
    "cm2_transpose", "cm2_cos_sin_2phi", "cm2_m2_finish",
    "cm2_filter_create", "cm2_filter_apply", "cm2_filter_apply_tiles",
    "cm2_cutsky_to_fullsky", "cm2_fullsky_to_cutsky", "cm2_ground_bin_sums", "cm2_ground_subtract",
])


def call(name, *args):
    """Call an int-status entry point and raise HipError with cm2_last_error().  A RESTARTABLE call that
    returned CM2_ERR_OUT_OF_MEMORY is tried once more after torch has returned its idle blocks to the driver
    (the library has released its own cached blocks before reporting the failure)."""
    fn = getattr(load(), name)
    rc = fn(*args)
    if rc == ERR_OUT_OF_MEMORY and name in RESTARTABLE and _free_torch_cache():
        rc = fn(*args)
    check(rc)
