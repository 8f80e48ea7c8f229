"""ctypes binding of the C-ABI in include/smhip.h (stereomatch_amd/lib/libsmhip.so).

This is plumbing for Python callers (tests, bench.py); the compute is entirely in the
HIP library.  There is no CPU fallback: if the library is missing or a call fails, an
exception is raised.
"""
import contextlib
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsmhip.so")

# every symbol include/smhip.h declares
SYMBOLS = [
    "smhip_init", "smhip_shutdown", "smhip_last_error", "smhip_set_stream", "smhip_set_option", "smhip_sync",
    "smhip_workspace_bytes", "smhip_malloc", "smhip_free", "smhip_memcpy_h2d",
    "smhip_memcpy_d2h", "smhip_memcpy_d2d", "smhip_memcpy2d_h2d", "smhip_memcpy2d_d2h", "smhip_match_outside",
    "smhip_raw_costs", "smhip_box_filter", "smhip_min_filter", "smhip_min_filter_to",
    "smhip_convolve_14641", "smhip_convolve_14641_to",
    "smhip_diffusion_iter", "smhip_pad_costs", "smhip_wta", "smhip_fused_box_wta_keys",
    "smhip_keys_decode", "smhip_fused_box_wta", "smhip_profile_begin", "smhip_profile_end", "smhip_profile_get", "smhip_fused_box_wta_noalloc",
    "smhip_subpixel_fit", "smhip_collapse", "smhip_refine", "smhip_match_quality", "smhip_box_filter_border",
    "smhip_diffusion_run", "smhip_convolve_14641_run",
    "smhip_fused_box_dsi", "smhip_min_pad_wta_keys", "smhip_preprocess_blur",
    "smhip_opt_so", "smhip_bayesian_iter", "smhip_opt_dp", "smhip_bayesian_run", "smhip_pad_wta_keys",
    "smhip_last_pass_a_kernel", "smhip_fused_iter_wta_keys", "smhip_fused_box_wta_keys_merge",
]


class SmhipError(RuntimeError):
    """Non-zero status from the C-ABI; the host driver turns these into CError
    (Error.h:23-32), here they are Python exceptions carrying smhip_last_error()."""

    def __init__(self, code, msg):
        super().__init__("smhip error %d: %s" % (code, msg))
        self.code = code


class MatchParams(C.Structure):
    """`smhip_match_params` (replaces LineProcessStruct, CudaRawCosts.h:97-114)."""

    _fields_ = [
        ("width", C.c_int), ("height", C.c_int), ("n_bands", C.c_int),
        ("disp_min", C.c_int), ("disp_num", C.c_int), ("disp_den", C.c_int),
        ("disp_n", C.c_int), ("frame_diff_sign", C.c_int),
        ("match_fn", C.c_int), ("match_interp", C.c_int), ("match_max", C.c_int),
        ("match_interval", C.c_int), ("match_interpolated", C.c_int),
    ]


def match_params(width, height, n_bands, disp_n, disp_min=0, disp_num=1, disp_den=1,
                 frame_diff_sign=1, match_fn=1, match_interp=3, match_max=1000,
                 match_interval=0, match_interpolated=0):
    return MatchParams(width, height, n_bands, disp_min, disp_num, disp_den, disp_n,
                       frame_diff_sign, match_fn, match_interp, match_max, match_interval,
                       match_interpolated)


_lib = None


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.  Two HIP runtimes in one process do
    not share the GPU: whichever loads second finds no device (measured: `import torch` after
    libsmhip.so had pulled in /opt/rocm's runtime gives torch.cuda.is_available() == False).
    So when a torch with a bundled runtime is installed, that copy is loaded first and
    libsmhip.so binds to it (same soname) -- the import order of torch and this module then does
    not matter.  Nothing of torch itself is imported here."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(cand):
        return None
    try:
        return C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except OSError:
        return None


def load():
    """Loads libsmhip.so and declares prototypes.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    _share_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, sz, ci, cf = C.c_void_p, C.c_size_t, C.c_int, C.c_float
    pp = C.POINTER(MatchParams)
    L.smhip_init.argtypes = [ci, C.POINTER(vp)]
    L.smhip_shutdown.argtypes = [vp]
    L.smhip_last_error.restype = C.c_char_p
    L.smhip_last_error.argtypes = []
    L.smhip_set_stream.argtypes = [vp, vp]
    L.smhip_set_option.argtypes = [vp, ci, ci]
    L.smhip_sync.argtypes = [vp]
    L.smhip_workspace_bytes.restype = sz
    L.smhip_workspace_bytes.argtypes = [vp]
    L.smhip_malloc.argtypes = [vp, sz, C.POINTER(vp)]
    L.smhip_free.argtypes = [vp, vp]
    L.smhip_memcpy_h2d.argtypes = [vp, vp, vp, sz]
    L.smhip_memcpy_d2h.argtypes = [vp, vp, vp, sz]
    L.smhip_memcpy_d2d.argtypes = [vp, vp, vp, sz]
    L.smhip_memcpy2d_h2d.argtypes = [vp, vp, sz, vp, sz, sz, sz]
    L.smhip_memcpy2d_d2h.argtypes = [vp, vp, sz, vp, sz, sz, sz]
    L.smhip_match_outside.restype = cf
    L.smhip_match_outside.argtypes = [pp]
    L.smhip_raw_costs.argtypes = [vp, vp, vp, sz, pp, vp, sz]
    L.smhip_box_filter.argtypes = [vp, vp, ci, ci, ci, sz, ci, ci, ci]
    L.smhip_min_filter.argtypes = [vp, vp, ci, ci, ci, sz, ci, ci]
    L.smhip_convolve_14641.argtypes = [vp, vp, ci, ci, ci, sz]
    L.smhip_min_filter_to.argtypes = [vp, vp, vp, ci, ci, ci, sz, ci, ci]
    L.smhip_convolve_14641_to.argtypes = [vp, vp, vp, ci, ci, ci, sz]
    L.smhip_diffusion_iter.argtypes = [vp, vp, vp, vp, ci, ci, ci, sz, cf, cf]
    L.smhip_bayesian_iter.argtypes = [vp, vp, vp, vp, ci, ci, ci, sz, cf, cf, cf, cf, ci]
    L.smhip_bayesian_run.argtypes = [vp, vp, vp, vp, ci, ci, ci, sz, cf, cf, cf, cf, ci, C.POINTER(ci)]
    L.smhip_pad_costs.argtypes = [vp, vp, sz, pp]
    L.smhip_wta.argtypes = [vp, vp, ci, ci, ci, sz, vp, vp]
    L.smhip_fused_box_wta_keys.argtypes = [vp, vp, vp, sz, pp, ci, ci, ci, vp]
    L.smhip_fused_box_wta_keys_merge.argtypes = [vp, vp, vp, sz, pp, ci, ci, ci, vp]
    L.smhip_keys_decode.argtypes = [vp, vp, ci, ci, vp, vp]
    L.smhip_fused_box_wta.argtypes = [vp, vp, vp, sz, pp, ci, vp, vp]
    L.smhip_fused_box_wta_noalloc.argtypes = [vp, vp, vp, sz, pp, ci, vp, vp]
    L.smhip_subpixel_fit.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, sz, cf]
    L.smhip_collapse.argtypes = [vp, vp, vp, ci, ci, ci, sz, ci, ci, sz, cf, vp, vp]
    L.smhip_refine.argtypes = [vp, vp, vp, vp, ci, ci, ci, sz, cf, cf, ci, vp]
    L.smhip_box_filter_border.argtypes = [vp, vp, ci, ci, ci, sz, ci, ci, ci, ci]
    L.smhip_match_quality.argtypes = [vp, vp, vp, vp, ci, ci, ci, sz, sz, cf, cf, ci, ci, ci, ci, vp, vp]
    L.smhip_diffusion_run.argtypes = [vp, vp, vp, vp, ci, ci, ci, sz, cf, cf, ci, C.POINTER(ci)]
    L.smhip_convolve_14641_run.argtypes = [vp, vp, vp, ci, ci, ci, sz, ci, C.POINTER(ci)]
    L.smhip_fused_box_dsi.argtypes = [vp, vp, vp, sz, pp, ci, ci, ci, vp, sz]
    L.smhip_min_pad_wta_keys.argtypes = [vp, vp, sz, pp, ci, ci, ci, vp]
    L.smhip_pad_wta_keys.argtypes = [vp, vp, sz, pp, ci, ci, vp]
    L.smhip_preprocess_blur.argtypes = [vp, vp, sz, ci, ci, ci, ci]
    L.smhip_opt_so.argtypes = [vp, vp, ci, ci, ci, sz, vp, sz, ci, cf, cf, cf, vp, vp]
    L.smhip_opt_dp.argtypes = [vp, vp, ci, ci, ci, sz, vp, sz, ci, cf, cf, cf, ci, vp, vp]
    L.smhip_profile_begin.argtypes = [vp, ci]
    L.smhip_profile_end.argtypes = [vp, C.POINTER(ci), C.POINTER(cf), C.POINTER(cf)]
    L.smhip_profile_get.argtypes = [vp, ci, ci, C.POINTER(cf), C.POINTER(cf)]
    L.smhip_fused_iter_wta_keys.argtypes = [vp, vp, vp, sz, pp, ci, ci, cf, cf, ci, ci, vp]
    L.smhip_last_pass_a_kernel.argtypes = [vp]
    L.smhip_last_pass_a_kernel.restype = C.c_char_p
    for name in SYMBOLS:
        f = getattr(L, name)
        if name not in ("smhip_last_error", "smhip_workspace_bytes", "smhip_match_outside", "smhip_last_pass_a_kernel"):
            f.restype = ci
    _lib = L
    return L


class DeviceBuffer:
    """A device allocation owned by a Context (or a borrowed pointer, e.g. a torch tensor)."""

    def __init__(self, ctx, ptr, nbytes, owned=True):
        self.ctx, self.ptr, self.nbytes, self.owned = ctx, ptr, nbytes, owned

    def free(self):
        if self.owned and self.ptr:
            self.ctx._chk(self.ctx.L.smhip_free(self.ctx.h, self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, DeviceBuffer):
        return x.ptr
    return int(x)


# enum smhip_option (include/smhip.h)
OPTIONS = {"narrow": 0, "tyshift": 1, "chunk": 2, "bt_prepass": 3, "tseg": 4, "maxk": 5, "float_a": 6, "keybuf": 7, "store_policy": 8, "paired": 9, "a_lds_kb": 10, "b_waves": 11, "stage_lds_kb": 12, "b_ahead": 13, "multi_txt": 14, "tap_slow": 15, "multi_px": 16, "iter_fused": 17, "iter_groups": 18, "a_nw": 19, "b_lred": 20, "iter_direct": 21}


class Context:
    """One device context (`smhip_ctx`); not thread-safe, like the reference."""

    def __init__(self, device=0):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.smhip_init(device, C.byref(h))
        if rc:
            raise SmhipError(rc, self.L.smhip_last_error().decode())
        self.h = h

    def _chk(self, rc):
        if rc:
            raise SmhipError(rc, self.L.smhip_last_error().decode())

    def close(self):
        if self.h:
            self._chk(self.L.smhip_shutdown(self.h))
            self.h = None

    def set_stream(self, stream_ptr):
        self._chk(self.L.smhip_set_stream(self.h, stream_ptr))

    def sync(self):
        self._chk(self.L.smhip_sync(self.h))

    def set_option(self, name, value):
        """Tuning / test knob of this context (`enum smhip_option`); value < 0 = the library's policy."""
        self._chk(self.L.smhip_set_option(self.h, OPTIONS[name], int(value)))

    @contextlib.contextmanager
    def options(self, **kw):
        """`with ctx.options(chunk=1, tyshift=3): ...` -- knobs set for the block, policy restored after."""
        try:
            for k, v in kw.items():
                self.set_option(k, v)
            yield self
        finally:
            for k in kw:
                self.set_option(k, -1)

    def workspace_bytes(self):
        return self.L.smhip_workspace_bytes(self.h)

    # -- memory
    def alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self.L.smhip_malloc(self.h, nbytes, C.byref(p)))
        return DeviceBuffer(self, p.value, nbytes)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        buf = self.alloc(max(arr.nbytes, 1))
        if arr.nbytes:
            self._chk(self.L.smhip_memcpy_h2d(self.h, buf.ptr, arr.ctypes.data, arr.nbytes))
        return buf

    def download(self, buf, shape, dtype):
        out = np.empty(shape, dtype)
        self._chk(self.L.smhip_memcpy_d2h(self.h, out.ctypes.data, _ptr(buf), out.nbytes))
        return out

    def copy_d2d(self, dst, src, nbytes):
        self._chk(self.L.smhip_memcpy_d2d(self.h, _ptr(dst), _ptr(src), nbytes))

    def upload_pitched(self, arr2d_bytes, d_pitch):
        """arr2d_bytes: uint8 [rows, h_pitch] host array; copies width=min pitch."""
        rows, h_pitch = arr2d_bytes.shape
        buf = self.alloc(rows * d_pitch)
        self._chk(self.L.smhip_memcpy2d_h2d(self.h, buf.ptr, d_pitch, arr2d_bytes.ctypes.data,
                                            h_pitch, min(h_pitch, d_pitch), rows))
        return buf

    def download_pitched(self, buf, d_pitch, rows, h_pitch, width_bytes):
        out = np.zeros((rows, h_pitch), np.uint8)
        self._chk(self.L.smhip_memcpy2d_d2h(self.h, out.ctypes.data, h_pitch, _ptr(buf), d_pitch,
                                            width_bytes, rows))
        return out

    # -- operators (arguments as in include/smhip.h)
    def preprocess_blur(self, d_img, img_pitch, w, h, nb, n_iter):
        self._chk(self.L.smhip_preprocess_blur(self.h, _ptr(d_img), img_pitch, w, h, nb, n_iter))

    def match_outside(self, p):
        return float(self.L.smhip_match_outside(C.byref(p)))

    def raw_costs(self, d_ref, d_mtc, img_pitch, p, d_cost, row_elems):
        self._chk(self.L.smhip_raw_costs(self.h, _ptr(d_ref), _ptr(d_mtc), img_pitch, C.byref(p),
                                         _ptr(d_cost), row_elems))

    def box_filter(self, d_cost, w, h, nb, row_elems, x_width, y_width=None, average=True):
        y_width = x_width if y_width is None else y_width
        self._chk(self.L.smhip_box_filter(self.h, _ptr(d_cost), w, h, nb, row_elems, x_width,
                                          y_width, int(average)))

    def min_filter(self, d_cost, w, h, nb, row_elems, x_width, y_width=None):
        y_width = x_width if y_width is None else y_width
        self._chk(self.L.smhip_min_filter(self.h, _ptr(d_cost), w, h, nb, row_elems, x_width,
                                          y_width))

    def convolve_14641(self, d_cost, w, h, nb, row_elems):
        self._chk(self.L.smhip_convolve_14641(self.h, _ptr(d_cost), w, h, nb, row_elems))

    def min_filter_to(self, d_src, d_dst, w, h, nb, row_elems, x_width, y_width=None):
        y_width = x_width if y_width is None else y_width
        self._chk(self.L.smhip_min_filter_to(self.h, _ptr(d_src), _ptr(d_dst), w, h, nb, row_elems,
                                             x_width, y_width))

    def convolve_14641_to(self, d_src, d_dst, w, h, nb, row_elems):
        self._chk(self.L.smhip_convolve_14641_to(self.h, _ptr(d_src), _ptr(d_dst), w, h, nb, row_elems))

    def diffusion_iter(self, d_src, d_dst, d_cost0, w, h, nb, row_elems, lam, beta):
        self._chk(self.L.smhip_diffusion_iter(self.h, _ptr(d_src), _ptr(d_dst), _ptr(d_cost0),
                                              w, h, nb, row_elems, lam, beta))

    def bayesian_iter(self, d_cost, d_cost2, d_cost0, w, h, nb, row_elems, scale_cost, mu, sigma_p,
                      eps_p, it):
        """One AggrBayesian iteration (StcDiffusion.cpp:139-308), E in place; tolerance mode."""
        self._chk(self.L.smhip_bayesian_iter(self.h, _ptr(d_cost), _ptr(d_cost2), _ptr(d_cost0), w, h,
                                             nb, row_elems, scale_cost, mu, sigma_p, eps_p, it))

    def bayesian_run(self, d_cost, d_cost2, d_cost0, w, h, nb, row_elems, scale_cost, mu, sigma_p, eps_p,
                     n_iter):
        """n_iter AggrBayesian iterations in one call; returns True when E is in d_cost2."""
        flag = C.c_int()
        self._chk(self.L.smhip_bayesian_run(self.h, _ptr(d_cost), _ptr(d_cost2), _ptr(d_cost0), w, h, nb,
                                            row_elems, scale_cost, mu, sigma_p, eps_p, n_iter, C.byref(flag)))
        return bool(flag.value)

    def diffusion_run(self, d_a, d_b, d_cost0, w, h, nb, row_elems, lam, beta, n_iter):
        """n_iter iterations, up to four per pass; returns True when the result is in d_b."""
        flag = C.c_int()
        self._chk(self.L.smhip_diffusion_run(self.h, _ptr(d_a), _ptr(d_b), _ptr(d_cost0), w, h, nb,
                                             row_elems, lam, beta, n_iter, C.byref(flag)))
        return bool(flag.value)

    def convolve_14641_run(self, d_a, d_b, w, h, nb, row_elems, n_iter):
        flag = C.c_int()
        self._chk(self.L.smhip_convolve_14641_run(self.h, _ptr(d_a), _ptr(d_b), w, h, nb, row_elems,
                                                  n_iter, C.byref(flag)))
        return bool(flag.value)

    def pad_costs(self, d_cost, row_elems, p):
        self._chk(self.L.smhip_pad_costs(self.h, _ptr(d_cost), row_elems, C.byref(p)))

    def wta(self, d_cost, w, h, nb, row_elems, d_disp, d_best):
        self._chk(self.L.smhip_wta(self.h, _ptr(d_cost), w, h, nb, row_elems, _ptr(d_disp),
                                   _ptr(d_best)))

    def fused_box_wta_keys(self, d_ref, d_mtc, img_pitch, p, window, d_first, d_count, d_keys):
        self._chk(self.L.smhip_fused_box_wta_keys(self.h, _ptr(d_ref), _ptr(d_mtc), img_pitch,
                                                  C.byref(p), window, d_first, d_count,
                                                  _ptr(d_keys)))

    def fused_box_wta_keys_merge(self, d_ref, d_mtc, img_pitch, p, window, d_first, d_count, d_keys):
        """As fused_box_wta_keys, but d_keys receives min(d_keys, this call's keys)."""
        self._chk(self.L.smhip_fused_box_wta_keys_merge(self.h, _ptr(d_ref), _ptr(d_mtc), img_pitch,
                                                        C.byref(p), window, d_first, d_count, _ptr(d_keys)))

    def fused_box_dsi(self, d_ref, d_mtc, img_pitch, p, window, d_first, d_count, d_cost, row_elems):
        self._chk(self.L.smhip_fused_box_dsi(self.h, _ptr(d_ref), _ptr(d_mtc), img_pitch, C.byref(p),
                                             window, d_first, d_count, _ptr(d_cost), row_elems))

    def min_pad_wta_keys(self, d_cost, row_elems, p, min_window, d_first, d_count, d_keys):
        self._chk(self.L.smhip_min_pad_wta_keys(self.h, _ptr(d_cost), row_elems, C.byref(p), min_window,
                                                d_first, d_count, _ptr(d_keys)))

    def fused_iter_wta_keys(self, d_ref, d_mtc, img_pitch, p, aggr_fn, n_iter, d_first, d_count, d_keys,
                            diff_lambda=0.15, diff_beta=0.5):
        """Raw costs -> n_iter iterations of aggr_fn (2 binomial, 3 diffusion, 4 membrane) -> PadCosts -> WTA keys."""
        self._chk(self.L.smhip_fused_iter_wta_keys(self.h, _ptr(d_ref), _ptr(d_mtc), img_pitch, C.byref(p), aggr_fn,
                                                   n_iter, diff_lambda, diff_beta, d_first, d_count, _ptr(d_keys)))

    def pad_wta_keys(self, d_cost, row_elems, p, d_first, d_count, d_keys):
        self._chk(self.L.smhip_pad_wta_keys(self.h, _ptr(d_cost), row_elems, C.byref(p), d_first, d_count,
                                            _ptr(d_keys)))

    def opt_so(self, d_cost, w, h, nb, row_elems, d_ref, img_pitch, img_bands, smoothness, grad_thresh,
               grad_penalty, d_disp, d_best):
        self._chk(self.L.smhip_opt_so(self.h, _ptr(d_cost), w, h, nb, row_elems, _ptr(d_ref), img_pitch,
                                      img_bands, smoothness, grad_thresh, grad_penalty, _ptr(d_disp),
                                      _ptr(d_best)))

    def opt_dp(self, d_cost, w, h, nb, row_elems, d_ref, img_pitch, img_bands, smoothness, grad_thresh,
               grad_penalty, occlusion_cost, d_disp, d_best):
        self._chk(self.L.smhip_opt_dp(self.h, _ptr(d_cost), w, h, nb, row_elems, _ptr(d_ref), img_pitch,
                                      img_bands, smoothness, grad_thresh, grad_penalty, int(occlusion_cost),
                                      _ptr(d_disp), _ptr(d_best)))

    def keys_decode(self, d_keys, width, height, d_disp, d_best):
        """d_keys: width*height int64, column-major (pixel (x, y) at x*height + y)."""
        self._chk(self.L.smhip_keys_decode(self.h, _ptr(d_keys), width, height, _ptr(d_disp),
                                           _ptr(d_best)))

    def fused_box_wta(self, d_ref, d_mtc, img_pitch, p, window, d_disp, d_best):
        self._chk(self.L.smhip_fused_box_wta(self.h, _ptr(d_ref), _ptr(d_mtc), img_pitch,
                                             C.byref(p), window, _ptr(d_disp), _ptr(d_best)))

    def fused_box_wta_noalloc(self, d_ref, d_mtc, img_pitch, p, window, d_disp, d_best):
        """smhip_fused_box_wta on the device's static scratch array: no device allocation in the call."""
        self._chk(self.L.smhip_fused_box_wta_noalloc(self.h, _ptr(d_ref), _ptr(d_mtc), img_pitch,
                                                     C.byref(p), window, _ptr(d_disp), _ptr(d_best)))

    def subpixel_fit(self, d_src, d_dst, d_min, d_cert, w, h, nb, row_elems, match_outside):
        self._chk(self.L.smhip_subpixel_fit(self.h, _ptr(d_src), _ptr(d_dst), _ptr(d_min), _ptr(d_cert),
                                            w, h, nb, row_elems, match_outside))

    def collapse(self, d_cost1, d_min1, w, h, nb1, row_elems1, df, nb2, row_elems2, disp_step,
                 d_cost2, d_min2):
        self._chk(self.L.smhip_collapse(self.h, _ptr(d_cost1), _ptr(d_min1), w, h, nb1, row_elems1, df,
                                        nb2, row_elems2, disp_step, _ptr(d_cost2), _ptr(d_min2)))

    def refine(self, d_disp, d_cost, d_min, w, h, nb, row_elems, disp_step, disp_min, mode, d_fdisp):
        self._chk(self.L.smhip_refine(self.h, _ptr(d_disp), _ptr(d_cost), _ptr(d_min), w, h, nb,
                                      row_elems, disp_step, disp_min, mode, _ptr(d_fdisp)))

    def box_filter_border(self, d_cost, w, h, nb, row_elems, window, average, border_x, border_y):
        self._chk(self.L.smhip_box_filter_border(self.h, _ptr(d_cost), w, h, nb, row_elems, window, int(average),
                                                 border_x, border_y))

    def match_quality(self, d_cost, d_cert, d_fdisp, w, h, nb, row_elems, cert_row_elems, disp_min,
                      disp_step_inv, quality_mode, use_fit, norm_bands, take_sqrt, d_fcost, d_fcert):
        self._chk(self.L.smhip_match_quality(self.h, _ptr(d_cost), _ptr(d_cert), _ptr(d_fdisp), w, h, nb,
                                             row_elems, cert_row_elems, disp_min, disp_step_inv, quality_mode,
                                             int(use_fit), norm_bands, int(take_sqrt), _ptr(d_fcost), _ptr(d_fcert)))

    def profile_begin(self, max_calls):
        self._chk(self.L.smhip_profile_begin(self.h, max_calls))

    def profile_get(self, first, count):
        """Per-call (ms of pass A, ms of pass B) lists of the profiled calls [first, first + count)."""
        a, b = (C.c_float * count)(), (C.c_float * count)()
        self._chk(self.L.smhip_profile_get(self.h, first, count, a, b))
        return list(a), list(b)

    def last_pass_a_kernel(self):
        """Name of the pass A kernel the most recent fused call launched."""
        return self.L.smhip_last_pass_a_kernel(self.h).decode()

    def profile_end(self):
        """Returns (n_calls, total ms of pass A, total ms of pass B)."""
        n, a, b = C.c_int(), C.c_float(), C.c_float()
        self._chk(self.L.smhip_profile_end(self.h, C.byref(n), C.byref(a), C.byref(b)))
        return n.value, a.value, b.value
