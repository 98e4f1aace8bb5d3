"""ctypes binding of ``libptychohip.so`` (C ABI: ``include/ptycho_hip.h``).

Replaces the SWIG / pybind11 extension module ``libtike.cufft.ptychofft``
(``/root/reference/src/cuda/swig/ptychofft.i:1-26``,
``/root/reference/src/cuda/pybind11/ptychofft.cxx:1-27``).  There is no
fallback: if the shared library is missing this module raises at import.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTYCHO_HIP_LIB") or os.path.join(_HERE, "libptychohip.so")

#: every symbol ``include/ptycho_hip.h`` declares
SYMBOLS = ("ptycho_create", "ptycho_free", "ptycho_destroy", "ptycho_get",
           "ptycho_fwd", "ptycho_adj", "ptycho_fft2", "ptycho_set_option",
           "ptycho_profile", "ptycho_profile_read",
           "ptycho_cg_fwd_cols", "ptycho_cg_stats", "ptycho_cg_project",
           "ptycho_cg_adj_cols", "ptycho_cg_linesearch",
           "ptycho_cg_project_multi",
           "ptycho_cg_intensity_modes", "ptycho_cg_linesearch_modes",
           "ptycho_cg_cross", "ptycho_cg_argmax", "ptycho_cg_zoom",
           "ptycho_cg_fwd_cols_modes", "ptycho_cg_linesearch_chunk",
           "ptycho_cg_obj_begin", "ptycho_cg_obj_grad", "ptycho_cg_obj_dir", "ptycho_cg_ls_next",
           "ptycho_cg_reg_prepare",
           "ptycho_cg_obj_finish", "ptycho_cg_prb_grad", "ptycho_cg_prb_dir", "ptycho_cg_prb_finish",
           "ptycho_cg_ls_begin", "ptycho_cg_ls_obj_chunk", "ptycho_cg_ls_prb_pass", "ptycho_cg_ls_decide",
           "ptycho_cg_cross_dev", "ptycho_cg_obj_begin2", "ptycho_cg_obj_dir2",
           "ptycho_last_error", "ptycho_version")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libtike.hipfft: %s not found -- build it with "
        "`make -C libtike-cufft_amd/csrc` (or `python -c 'import "
        "__graft_entry__ as g; g.build()'`). There is no CPU fallback." % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH)

_vp, _sz, _i, _ll = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_longlong


def _sig(name, res, *args):
    fn = getattr(lib, name)
    fn.restype = res
    fn.argtypes = list(args)
    return fn


create = _sig("ptycho_create", _i, ctypes.POINTER(_vp), _sz, _sz, _sz, _sz, _sz, _sz)
free = _sig("ptycho_free", _i, _vp)
destroy = _sig("ptycho_destroy", _i, _vp)
get = _sig("ptycho_get", _ll, _vp, _i)
fwd = _sig("ptycho_fwd", _i, _vp, _vp, _vp, _vp, _vp, _vp)
adj = _sig("ptycho_adj", _i, _vp, _vp, _vp, _vp, _vp, _i, _vp)
fft2 = _sig("ptycho_fft2", _i, _vp, _vp, _vp, _sz, _i, _vp)
set_option = _sig("ptycho_set_option", _i, _vp, ctypes.c_char_p, _ll)
cg_fwd_cols = _sig("ptycho_cg_fwd_cols", _i, _vp, _i, _vp, _vp, _vp, _vp)
cg_stats = _sig("ptycho_cg_stats", _i, _vp, _i, _vp, _vp, _vp)
cg_project = _sig("ptycho_cg_project", _i, _vp, _i, _i, _vp, _vp, _vp, _vp)
cg_adj_cols = _sig("ptycho_cg_adj_cols", _i, _vp, _i, _vp, _vp, _vp, _i, _vp)
cg_linesearch = _sig("ptycho_cg_linesearch", _i, _vp, _i, _i, _vp, _vp, ctypes.c_double, _i, _vp, _vp)
cg_project_multi = _sig("ptycho_cg_project_multi", _i, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp)
cg_intensity_modes = _sig("ptycho_cg_intensity_modes", _i, _vp, _i, _vp, _vp, _vp, _vp)
cg_linesearch_modes = _sig("ptycho_cg_linesearch_modes", _i, _vp, _i, _i, _vp, _vp, _vp, ctypes.c_double, _i, _vp, _vp)
cg_cross = _sig("ptycho_cg_cross", _i, _vp, _i, _i, ctypes.c_double, _vp, _vp)
cg_argmax = _sig("ptycho_cg_argmax", _i, _vp, _i, _vp, _vp)
cg_zoom = _sig("ptycho_cg_zoom", _i, _vp, _vp, _vp, _vp, _vp, _i, _i, ctypes.c_double, _vp, _vp)
_d = ctypes.c_double
cg_fwd_cols_modes = _sig("ptycho_cg_fwd_cols_modes", _i, _vp, _i, _i, _vp, _vp, ctypes.POINTER(_vp), _i, _i, _vp)
cg_linesearch_chunk = _sig("ptycho_cg_linesearch_chunk", _i, _vp, _i, _vp, _vp, _d, _i, _vp, _vp)
cg_obj_begin = _sig("ptycho_cg_obj_begin", _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp)
cg_obj_grad = _sig("ptycho_cg_obj_grad", _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp)
cg_obj_dir = _sig("ptycho_cg_obj_dir", _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp)
cg_ls_next = _sig("ptycho_cg_ls_next", _i, _vp, _vp, _i, _i, _vp, _i, _vp)
cg_reg_prepare = _sig("ptycho_cg_reg_prepare", _i, _vp, _vp, _vp, _vp, _vp, _vp)
cg_obj_finish = _sig("ptycho_cg_obj_finish", _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _d, _vp)
cg_prb_grad = _sig("ptycho_cg_prb_grad", _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp)
cg_prb_dir = _sig("ptycho_cg_prb_dir", _i, _vp, _vp, _i, _d, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp)
cg_prb_finish = _sig("ptycho_cg_prb_finish", _i, _vp, _vp, _vp, _vp, _vp)
cg_ls_begin = _sig("ptycho_cg_ls_begin", _i, _vp, _vp, _i, _vp)
cg_ls_obj_chunk = _sig("ptycho_cg_ls_obj_chunk", _i, _vp, _vp, _i, _vp, _vp, ctypes.POINTER(_vp), _vp, _vp, _vp)
cg_ls_prb_pass = _sig("ptycho_cg_ls_prb_pass", _i, _vp, _vp, _i, _vp, _vp, _vp)
cg_ls_decide = _sig("ptycho_cg_ls_decide", _i, _vp, _vp, _i, _i, _vp)
cg_cross_dev = _sig("ptycho_cg_cross_dev", _i, _vp, _i, _i, _vp, _vp, _vp)
cg_obj_begin2 = _sig("ptycho_cg_obj_begin2", _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp)
cg_obj_dir2 = _sig("ptycho_cg_obj_dir2", _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp)
#: word offsets of the device-resident CG state (enum PTYCHO_ST_* in include/ptycho_hip.h)
ST_A, ST_B, ST_COST, ST_COST2 = 0, 1, 2, 3
ST_GAMMA_PSI, ST_GAMMA_PRB, ST_LS_FAILED, ST_HINT, ST_COSTS, ST_WORDS = 12, 13, 19, 20, 24, 160
ST_NCOSTS = 7 * 17
profile = _sig("ptycho_profile", _i, _vp, _i)
profile_read = _sig("ptycho_profile_read", _i, _vp, ctypes.POINTER(ctypes.c_double),
                    ctypes.POINTER(_ll), _i)
KERNEL_NAMES = ("k_cols<FWD>", "k_rows<fwd>", "k_rows<inv>", "k_cols<ADJ_OBJ>",
                "k_cols<ADJ_PRB>", "k_cols<PLAIN>", "sort_positions",
                "k_rows_fused<STATS>", "k_rows_fused<PROJECT>", "k_rows_fused<LINESEARCH>",
                "k_cg_scalars", "k_fwd_fused256", "k_cg_update",
                "k_rows_fused<CROSS>", "k_cols_argmax", "k_zoom_argmax",
                "k_fwd_tile", "k_adjprb_tile")
last_error = _sig("ptycho_last_error", ctypes.c_char_p)
version = _sig("ptycho_version", ctypes.c_char_p)


class PtychoHipError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise PtychoHipError("libptychohip error %d: %s"
                             % (rc, last_error().decode("utf-8", "replace")))
