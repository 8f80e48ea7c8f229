"""Python mirror of the reference's pipeline driver for the hot path.

`StereoMatcher` keeps the reference's parameter names (StereoParameters.h:50-242) and stage
names (StereoMatcher.h:78-109: RawCosts / Aggregate / Optimize / Refine) and sequences the
C-ABI calls of include/smhip.h exactly as ComputeCorrespondence (StereoMatcher.cpp:155-177)
sequences the CPU stages.  The DSI stays resident on the device between stages.  All compute
runs in libsmhip.so; this file holds no arithmetic beyond the O(W*H) depth-map scaling that
the reference also does on the host (StcRefine.cpp:34-39, StereoMatcher.cpp:177).
"""
import numpy as np

from . import capi

# enums, StereoParameters.h:33-136
eAD, eSD = 1, 2
eLinear, eCubic = 1, 3
eBox, eBinomial, eDiffusion, eMembrane, eBayesian = 1, 2, 3, 4, 5
eNoOpt, eWTA, eDynamicProg, eScanlineOpt = 0, 1, 2, 3


def _round(x):  # ROUND, StereoMatcher.cpp:36
    return int(x + 0.5) if x >= 0 else int(x - 0.5)


class StereoParameters:
    """Algorithm + sequence parameters with the defaults of ReInitializeSeqParams /
    ReInitializeAlgParams (StereoParameters.cpp:33-116); only the hot-path subset."""

    def __init__(self, **kw):
        self.ReInitializeSeqParams()
        self.ReInitializeAlgParams()
        for k, v in kw.items():
            if not hasattr(self, k):
                raise KeyError("unknown parameter %s" % k)
            setattr(self, k, v)

    def ReInitializeSeqParams(self):
        self.frame_ref, self.frame_match = 0, 1
        self.disp_min, self.disp_max, self.disp_step = 0, 7, 1.0
        self.disp_n, self.disp_scale = 0, 1.0
        self.verbose = 3

    def ReInitializeAlgParams(self):
        self.match_fn, self.match_interp, self.match_max = eAD, eCubic, 1000
        self.match_interval, self.match_interpolated = 0, 0
        self.aggr_fn, self.aggr_window_size, self.aggr_iter = eBox, 7, 1
        self.aggr_minfilter, self.aggr_subpixel, self.aggr_collapse = 0, 0, 0
        self.diff_lambda, self.diff_beta = 0.15, 0.5
        self.diff_scale_cost, self.diff_mu = 0.01, 0.5       # StereoParameters.cpp:80-83
        self.diff_sigmaP, self.diff_epsP = 0.4, 0.01
        self.opt_fn = eWTA
        self.opt_smoothness, self.opt_grad_thresh, self.opt_grad_penalty = 1.0, 5.0, 1.0
        self.opt_occlusion_cost = 20
        self.refine_subpix = 0
        self.eval_match_quality = 0
        self.preproc_blur_iter = 0


class StereoMatcher(StereoParameters):
    """Device-resident pipeline: images in, int32 disparity + best cost + uint8 depth out."""

    def __init__(self, ctx=None, device=0, **kw):
        super().__init__(**kw)
        self.ctx = ctx if ctx is not None else capi.Context(device)
        self.allow_fused = True
        self.used_fused = False
        self._slab = None                  # (whole-volume params, d_first, d_count, key buffer) in ComputeSlabKeys
        self._bufs = {}

    # -- device buffers (re-used across calls while the size fits)
    def _buf(self, name, nbytes):
        b = self._bufs.get(name)
        if b is None or b.nbytes < nbytes:
            if b is not None:
                b.free()
            b = self.ctx.alloc(nbytes)
            self._bufs[name] = b
        return b

    def _derive(self, shape):
        # StereoMatcher.cpp:60-82
        h, w, nb = shape
        step = float(np.float32(self.disp_step))        # the reference's disp_step is a float
        min_precision = float(np.float32(1e-3))
        if (step <= 0.0
                or (step < 1.0 and abs(1.0 / step - _round(1.0 / step)) > min_precision)
                or (step > 1.0 and abs(step - _round(step)) > min_precision)):
            raise ValueError("ComputeCorrespondence: disp_step must integer N or 1.0/N")
        self.m_frame_diff_sign = 1 if self.frame_match - self.frame_ref > 0 else -1
        self.m_disp_num = 1 if self.disp_step < 1.0 else _round(self.disp_step)
        self.m_disp_den = _round(1.0 / self.disp_step) if self.disp_step < 1.0 else 1
        self.disp_step = float(np.float32(self.m_disp_num) / np.float32(self.m_disp_den))
        inv = np.float32(self.m_disp_den) / np.float32(self.m_disp_num)
        self.disp_n = int(np.float32(inv * np.float32(self.disp_max - self.disp_min))) + 1
        self.m_disp_n = self.disp_n
        if self.m_disp_n < 2:
            raise ValueError("ComputeCorrespondence: too few disparity levels (%d)" % self.m_disp_n)
        self.m_params = capi.match_params(
            w, h, nb, self.m_disp_n, disp_min=self.disp_min, disp_num=self.m_disp_num,
            disp_den=self.m_disp_den, frame_diff_sign=self.m_frame_diff_sign,
            match_fn=self.match_fn, match_interp=self.match_interp, match_max=self.match_max,
            match_interval=1 if self.match_interval else 0,
            match_interpolated=self.match_interpolated)
        self.m_match_outside = self.ctx.match_outside(self.m_params)

    def _fusable(self):
        # box x1 + WTA and nobody needs the full DSI afterwards
        return (self.allow_fused and self.aggr_fn == eBox and self.aggr_iter == 1
                and self.aggr_minfilter <= 1 and self.opt_fn == eWTA and not self.refine_subpix
                and not self.aggr_subpixel and not self.aggr_collapse
                and not self.eval_match_quality and self.m_disp_den == 1)

    def _iter_fusable(self):
        # binomial iterations + WTA and nobody needs the DSI afterwards: smhip_fused_iter_wta_keys (aggr_iter <= 4
        # as one kernel that moves no DSI, more between its raw-cost and its WTA end)
        return (self.allow_fused and self.aggr_fn == eBinomial and self.aggr_iter >= 1
                and self.aggr_minfilter <= 1 and self.opt_fn == eWTA and not self.refine_subpix
                and not self.aggr_subpixel and not self.aggr_collapse
                and not self.eval_match_quality and self.m_disp_den == 1 and self.m_params.disp_n <= 65535)

    def _box_dsi_fusable(self):
        # RawCosts + one AggrBox as smhip_fused_box_dsi (12 instead of 20 bytes per voxel)
        p = self.m_params
        return (self.allow_fused and self.aggr_fn == eBox and self.aggr_iter == 1
                and self.m_disp_den == 1 and p.disp_n >= 33 and p.disp_n % 4 == 0
                and (p.n_bands == 1 or self.m_pitch % 4 == 0) and not self.eval_match_quality)

    def _min_wta_fusable(self):
        # AggrMin + PadCosts + OptWTA as smhip_min_pad_wta_keys (the filtered DSI is not written)
        return (self.allow_fused and self.aggr_minfilter in (3, 5, 7, 9, 11) and self.opt_fn == eWTA
                and self.m_disp_den == 1 and self.m_params.disp_n % 4 == 0 and not self.refine_subpix
                and not self.aggr_subpixel and not self.aggr_collapse and not self.eval_match_quality)

    def ComputeCorrespondence(self, ref, mtc):
        """ref, mtc: uint8 [H, W, nBands] host arrays.  Returns (disparity, best_cost, depth)."""
        ref = np.ascontiguousarray(ref, np.uint8)
        mtc = np.ascontiguousarray(mtc, np.uint8)
        if ref.ndim == 2:
            ref, mtc = ref[:, :, None], mtc[:, :, None]
        if ref.shape != mtc.shape:
            raise ValueError("ComputeCorrespondence: reference and matching not the same size")
        self._derive(ref.shape)
        h, w, nb = ref.shape
        self.m_pitch = w * nb
        self.d_ref = self._buf("ref", ref.nbytes)
        self.d_mtc = self._buf("mtc", mtc.nbytes)
        c = self.ctx
        c._chk(c.L.smhip_memcpy_h2d(c.h, self.d_ref.ptr, ref.ctypes.data, ref.nbytes))
        c._chk(c.L.smhip_memcpy_h2d(c.h, self.d_mtc.ptr, mtc.ctypes.data, mtc.nbytes))
        self.PreProcess()
        self.d_disp = self._buf("disp", w * h * 4)
        self.d_best = self._buf("best", w * h * 4)
        self.used_fused = self._fusable()
        self.m_disp_step = np.float32(self.disp_step)
        self.d_mind = None
        if self.used_fused:
            c.fused_box_wta(self.d_ref, self.d_mtc, self.m_pitch, self.m_params,
                            self.aggr_window_size, self.d_disp, self.d_best)
            disp = c.download(self.d_disp, (h, w), np.int32)
            self.m_float_disparity = self._scale_and_offset(disp)
        elif self._iter_fusable():
            self.used_fused = True
            keys = self._buf("keys", w * h * 8)
            c.fused_iter_wta_keys(self.d_ref, self.d_mtc, self.m_pitch, self.m_params, self.aggr_fn, self.aggr_iter,
                                  0, self.m_params.disp_n, keys, self.diff_lambda, self.diff_beta)
            c.keys_decode(keys, w, h, self.d_disp, self.d_best)
            disp = c.download(self.d_disp, (h, w), np.int32)
            self.m_float_disparity = self._scale_and_offset(disp)
        else:
            self.RawCosts()
            self.Aggregate()
            self.Optimize()
            disp = c.download(self.d_disp, (h, w), np.int32)
            self.m_float_disparity = self.Refine()
        best = c.download(self.d_best, (h, w), np.float32)
        return disp, best, self._depth_image(self.m_float_disparity)

    def ComputeSlabKeys(self, ref, mtc, d_first, d_count):
        """One rank's share of ComputeCorrespondence under disparity-slab sharding: levels
        [d_first, d_first + d_count) of the same parameters through PreProcess / RawCosts / Aggregate /
        PadCosts / OptWTA, result = the device buffer of per-pixel keys (column-major int64, global
        levels) whose MIN over the slabs is OptWTA's answer on the whole DSI.  Every aggregator but the
        Bayesian one treats the levels independently (BoxFilter.cpp:119-125, Convolve.cpp:166-254,
        StcDiffusion.cpp:28-137, MinFilter.cpp:122-202), so the slab is a DSI of its own whose first level
        has disparity disp_min + d_first * step."""
        ref = np.ascontiguousarray(ref, np.uint8)
        mtc = np.ascontiguousarray(mtc, np.uint8)
        if ref.ndim == 2:
            ref, mtc = ref[:, :, None], mtc[:, :, None]
        if ref.shape != mtc.shape:
            raise ValueError("ComputeCorrespondence: reference and matching not the same size")
        self._derive(ref.shape)
        if (self.opt_fn != eWTA or self.aggr_fn == eBayesian or self.m_disp_den != 1 or self.refine_subpix
                or self.aggr_subpixel or self.aggr_collapse):
            raise ValueError("slab sharding needs per-level aggregation, integer steps and opt_fn WTA")
        if d_first < 0 or d_count < 2 or d_first + d_count > self.m_disp_n:
            raise ValueError("slab [%d, %d) must hold >= 2 of the %d levels" % (d_first, d_first + d_count, self.m_disp_n))
        h, w, nb = ref.shape
        self.m_pitch = w * nb
        self.d_ref = self._buf("ref", ref.nbytes)
        self.d_mtc = self._buf("mtc", mtc.nbytes)
        c = self.ctx
        c._chk(c.L.smhip_memcpy_h2d(c.h, self.d_ref.ptr, ref.ctypes.data, ref.nbytes))
        c._chk(c.L.smhip_memcpy_h2d(c.h, self.d_mtc.ptr, mtc.ctypes.data, mtc.nbytes))
        self.PreProcess()
        keys = self._buf("keys", w * h * 8)
        whole = self.m_params
        self.used_fused = self._fusable()
        if self.used_fused:
            c.fused_box_wta_keys(self.d_ref, self.d_mtc, self.m_pitch, whole, self.aggr_window_size,
                                 d_first, d_count, keys)
            return keys
        if self._iter_fusable():
            self.used_fused = True
            c.fused_iter_wta_keys(self.d_ref, self.d_mtc, self.m_pitch, whole, self.aggr_fn, self.aggr_iter,
                                  d_first, d_count, keys, self.diff_lambda, self.diff_beta)
            return keys
        self.m_params = capi.match_params(
            w, h, nb, d_count, disp_min=self.disp_min + d_first * self.m_disp_num, disp_num=self.m_disp_num,
            disp_den=1, frame_diff_sign=self.m_frame_diff_sign, match_fn=self.match_fn,
            match_interp=self.match_interp, match_max=self.match_max,
            match_interval=1 if self.match_interval else 0, match_interpolated=self.match_interpolated)
        self._slab = (whole, d_first, d_count, keys)
        try:
            self.RawCosts()
            self.Aggregate()
        finally:
            self._slab = None
            self.m_params = whole
        return keys

    def PreProcess(self):  # StcPreProcess.cpp:31-56, on the device copies of the two frames
        if self.preproc_blur_iter > 0:
            p = self.m_params
            for d_img in (self.d_ref, self.d_mtc):
                self.ctx.preprocess_blur(d_img, self.m_pitch, p.width, p.height, p.n_bands,
                                         self.preproc_blur_iter)

    def RawCosts(self):  # StcRawCosts.cpp:229
        p = self.m_params
        self.m_disp_n = p.disp_n
        self.m_row_elems = p.width * p.disp_n
        self.d_cost = self._buf("cost", self.m_row_elems * p.height * 4)
        self.m_keys_done = False
        self.m_raw_deferred = self._box_dsi_fusable()       # raw costs are then made inside AggrBox
        if not self.m_raw_deferred:
            self.ctx.raw_costs(self.d_ref, self.d_mtc, self.m_pitch, p, self.d_cost, self.m_row_elems)

    def Aggregate(self):  # StcAggregate.cpp:255-365
        p, c = self.m_params, self.ctx
        w, h, nb, re = p.width, p.height, p.disp_n, self.m_row_elems
        nbytes = re * h * 4
        beta = 0.0 if self.aggr_fn == eDiffusion else self.diff_beta   # :305
        cost0 = None
        if self.aggr_fn == eMembrane and beta > 0.0 and self.aggr_iter > 0:
            cost0 = self._buf("cost0", nbytes)                          # :262
            self._copy(cost0, self.d_cost, nbytes)
        # the aggr_iter loop (StcAggregate.cpp:265-343); the iterated stencils run up to four
        # iterations per pass over the DSI (smhip_*_run), with the reference's ping-pong of
        # m_cost / m_cost2 (StcDiffusion.cpp:58-63)
        if self.aggr_fn == eBox:
            if self.m_raw_deferred:
                c.fused_box_dsi(self.d_ref, self.d_mtc, self.m_pitch, p, self.aggr_window_size, 0, nb,
                                self.d_cost, re)
            else:
                for _ in range(self.aggr_iter):
                    c.box_filter(self.d_cost, w, h, nb, re, self.aggr_window_size)
        elif self.aggr_fn in (eBinomial, eDiffusion, eMembrane):
            if self.aggr_iter > 0:
                other = self._buf("cost2", nbytes)
                if self.aggr_fn == eBinomial:
                    in_b = c.convolve_14641_run(self.d_cost, other, w, h, nb, re, self.aggr_iter)
                else:
                    in_b = c.diffusion_run(self.d_cost, other, cost0, w, h, nb, re, self.diff_lambda,
                                           beta, self.aggr_iter)
                if in_b:
                    self._bufs["cost"], self._bufs["cost2"] = other, self.d_cost
                    self.d_cost = other
        elif self.aggr_fn == eBayesian:                                   # :320-330, tolerance mode
            if self.aggr_iter > 0:
                es = self._buf("cost2", nbytes)
                e0 = self._buf("cost0", nbytes)
                in_b = c.bayesian_run(self.d_cost, es, e0, w, h, nb, re, self.diff_scale_cost, self.diff_mu,
                                      self.diff_sigmaP, self.diff_epsP, self.aggr_iter)
                if in_b:
                    self._bufs["cost"], self._bufs["cost2"] = es, self.d_cost
                    self.d_cost = es
        else:
            raise ValueError("CStereoMatcher::Aggregate(): unknown aggregation function")
        if self._slab is not None:                                       # :351-365 + OptWTA on a slab -> keys
            whole, d_first, d_count, keys = self._slab
            if self.aggr_minfilter > 1 and self._min_wta_fusable():
                c.min_pad_wta_keys(self.d_cost, re, whole, self.aggr_minfilter, d_first, d_count, keys)
                return
            if self.aggr_minfilter > 1:
                other = self._buf("cost2", nbytes)
                c.min_filter_to(self.d_cost, other, w, h, nb, re, self.aggr_minfilter)
                self._bufs["cost"], self._bufs["cost2"] = other, self.d_cost
                self.d_cost = other
            c.pad_wta_keys(self.d_cost, re, whole, d_first, d_count, keys)
            return
        if self.aggr_minfilter > 1 and self._min_wta_fusable():          # :351-365 + OptWTA, no DSI write
            keys = self._buf("keys", w * h * 8)
            c.min_pad_wta_keys(self.d_cost, re, p, self.aggr_minfilter, 0, nb, keys)
            c.keys_decode(keys, w, h, self.d_disp, self.d_best)
            self.m_keys_done = True
            return
        if self.aggr_minfilter > 1:                                      # :351-360
            other = self._buf("cost2", nbytes)
            c.min_filter_to(self.d_cost, other, w, h, nb, re, self.aggr_minfilter)
            self._bufs["cost"], self._bufs["cost2"] = other, self.d_cost
            self.d_cost = other
        c.pad_costs(self.d_cost, re, p)                                  # :365
        if self.aggr_subpixel:                                           # :368-385
            self.AggrSubPixelFit()
        if self.aggr_collapse:                                           # :388-400
            self.AggrCollapse()

    def AggrSubPixelFit(self):  # StcAggregate.cpp:66-132
        p, c = self.m_params, self.ctx
        nbytes = self.m_row_elems * p.height * 4
        other = self._buf("cost2", nbytes)
        self.d_mind = self._buf("mind", nbytes)
        self.d_cert = self._buf("cert", nbytes)
        c.subpixel_fit(self.d_cost, other, self.d_mind, self.d_cert, p.width, p.height, self.m_disp_n,
                       self.m_row_elems, self.m_match_outside)
        self._bufs["cost"], self._bufs["cost2"] = other, self.d_cost
        self.d_cost = other
        self.m_cert_row_elems = self.m_row_elems         # m_sub_pixel_cert is not collapsed later

    def AggrCollapse(self):  # StcAggregate.cpp:134-230
        if self.m_disp_step >= 1.0:
            return
        p, c = self.m_params, self.ctx
        df = self.m_disp_den
        n2 = (self.disp_max - self.disp_min) + 1
        re2 = p.width * n2
        cost2 = self._buf("cost_c", re2 * p.height * 4)
        mind2 = self._buf("mind_c", re2 * p.height * 4)
        c.collapse(self.d_cost, self.d_mind, p.width, p.height, self.m_disp_n, self.m_row_elems, df, n2,
                   re2, float(self.m_disp_step), cost2, mind2)
        self.d_cost, self.d_mind = cost2, mind2          # :227-229
        self.m_disp_n, self.m_row_elems, self.m_disp_step = n2, re2, np.float32(1.0)

    def Optimize(self):  # StcOptimize.cpp:162-280: WTA and the two scan-line optimisers
        p = self.m_params
        if self.opt_fn == eScanlineOpt:                                  # OptSO, StcOptSO.cpp
            self.ctx.opt_so(self.d_cost, p.width, p.height, self.m_disp_n, self.m_row_elems, self.d_ref,
                            self.m_pitch, p.n_bands, self.opt_smoothness, self.opt_grad_thresh,
                            self.opt_grad_penalty, self.d_disp, self.d_best)
            return
        if self.opt_fn == eDynamicProg:                                  # OptDP, StcOptDP.cpp
            self.ctx.opt_dp(self.d_cost, p.width, p.height, self.m_disp_n, self.m_row_elems, self.d_ref,
                            self.m_pitch, p.n_bands, self.opt_smoothness, self.opt_grad_thresh,
                            self.opt_grad_penalty, self.opt_occlusion_cost, self.d_disp, self.d_best)
            return
        if self.opt_fn != eWTA:
            raise ValueError("CStereoMatcher::Optimize(): only opt_fn 1 (WTA), 2 (dynamic programming) "
                             "and 3 (scan-line optimisation) are on the device path")
        if self.m_keys_done:
            return
        self.ctx.wta(self.d_cost, p.width, p.height, self.m_disp_n, self.m_row_elems, self.d_disp,
                     self.d_best)

    def Refine(self):  # StcRefine.cpp:28-100, on the resident DSI
        p, c = self.m_params, self.ctx
        mode = 0
        if self.refine_subpix and self.m_disp_n >= 3:
            from_fit = self.aggr_subpixel or (self.aggr_collapse and np.float32(self.disp_step) < 1.0)
            mode = 1 if from_fit else 2
        d_f = self._buf("fdisp", p.width * p.height * 4)
        c.refine(self.d_disp, self.d_cost, self.d_mind if mode == 1 else None, p.width, p.height,
                 self.m_disp_n, self.m_row_elems, float(self.m_disp_step), float(self.disp_min), mode, d_f)
        return c.download(d_f, (p.height, p.width), np.float32)

    def ComputeMatchQuality(self):  # StcEvaluate.cpp:760-833, on the resident DSI
        """(final_cost, certainty) float32 [H, W] for the last ComputeCorrespondence; needs
        eval_match_quality 1 or 2 (it keeps the DSI: no fused route)."""
        if not self.eval_match_quality or self.used_fused or self.m_keys_done:
            raise ValueError("ComputeMatchQuality: set eval_match_quality before ComputeCorrespondence")
        p, c = self.m_params, self.ctx
        n = p.width * p.height * 4
        d_f, d_fc, d_ce = self._buf("fdisp", n), self._buf("fcost", n), self._buf("fcert", n)
        fd = np.ascontiguousarray(self.m_float_disparity, np.float32)
        c._chk(c.L.smhip_memcpy_h2d(c.h, d_f.ptr, fd.ctypes.data, fd.nbytes))
        inv = float(np.float32(self.m_disp_den) / np.float32(self.m_disp_num)) if self.m_disp_step != 1.0 else 1.0
        c.match_quality(self.d_cost, self.d_cert if self.aggr_subpixel else None, d_f, p.width, p.height,
                        self.m_disp_n, self.m_row_elems, self.m_cert_row_elems if self.aggr_subpixel else 0,
                        float(self.disp_min), inv, self.eval_match_quality, bool(self.aggr_subpixel),
                        1 if p.n_bands == 1 else 3, self.match_fn == 2, d_fc, d_ce)
        return (c.download(d_fc, (p.height, p.width), np.float32),
                c.download(d_ce, (p.height, p.width), np.float32))

    def _scale_and_offset(self, disp):  # StcRefine.cpp:34-39 (Convert.cpp:31-40)
        f32 = np.float32
        d_off = f32(self.disp_min)
        fd = disp.astype(f32)
        if f32(self.m_disp_step) != 1.0 or d_off != 0.0:
            fd = fd * f32(self.m_disp_step) + d_off
        return fd

    def _depth_image(self, fd):  # StereoMatcher.cpp:177 (Convert.cpp:25-53)
        f32 = np.float32
        off = f32(np.float64(f32(-self.disp_min) * f32(self.disp_scale)) + 0.5)
        v = fd
        if f32(self.disp_scale) != 1.0 or off != 0.0:
            v = fd * f32(self.disp_scale) + off
        return np.clip(v, f32(0), f32(255)).astype(np.uint8)

    def download_cost(self):
        p = self.m_params
        return self.ctx.download(self.d_cost, (p.height, p.width, self.m_disp_n), np.float32)

    def _copy(self, dst, src, nbytes):
        self.ctx.copy_d2d(dst, src, nbytes)
