"""ctypes binding of libhipcanny.so plus Python mirrors of the reference operator classes.

`CannyEdge` and `cvPipeline` keep the names, argument meaning and error behaviour of
cvp::cuda::CannyEdge (src/cvp/cannyEdgeH.hpp:17-32) and cvp::cvPipeline (src/cvp/cvPipeline.hpp:20-39)
so the parity tests read like tests of the reference.  numpy arrays play the role of cv::Mat:
(H, W) uint8 == CV_8UC1, (H, W, 3) uint8 == CV_8UC3 (BGR).

There is no CPU fallback: if the shared library or a GPU is missing, construction raises.
"""
import ctypes as C
import enum
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HIPCANNY_LIB") or os.path.join(_HERE, "libhipcanny.so")  # override: kernel experiments only


class CannyStage(enum.IntEnum):
    """cvp::CannyStage, src/cvp/define.hpp:9-17"""
    MONO = 0
    GAUSSIAN = 1
    GRADIENT = 2
    NMS = 3
    THRESH = 4
    HYSTER = 5


# cvp::CANNY_STAGES, src/cvp/define.hpp:27-34 (display strings double as timer names)
CANNY_STAGES = {
    CannyStage.MONO: "1/6 Mono Conversion",
    CannyStage.GAUSSIAN: "2/6 Gaussian Noise Removal",
    CannyStage.GRADIENT: "3/6 Gradient Computation",
    CannyStage.NMS: "4/6 Non Maximum Suppression",
    CannyStage.THRESH: "5/6 Double Threshold",
    CannyStage.HYSTER: "6/6 Hysteresis",
}

MODE_R, MODE_O = 0, 1
OPT_NMS_SATURATE = 1
OPT_PIPELINE = 2
OPT_PER_CHANNEL = 3
OPT_FRONT_SPLIT = 4
OPT_L2_GRADIENT = 5
OPT_DEBUG_TAPS = 6
OPT_FRONT_HALF = 7
OPT_FRONT_DENSE = 8
OPT_COPY_STREAMS = 9
OPT_PIPELINE_SLOTS = 10
OPT_FRONT_WPB = 11
OPT_FRONT_MX = 12
OPT_TEST_HYST_LATE_GRID, OPT_TEST_HYST_LOOP, OPT_TEST_HYST_DIAG, OPT_TEST_HYST_GEOM, OPT_TEST_DENSE_ENTER, OPT_TEST_DENSE_LEAVE = 100, 101, 102, 103, 104, 105   # test / diagnostic hooks
TAP_BLUR, TAP_THRESH = 1, 2

# every symbol include/hipcanny.h declares
ABI_SYMBOLS = [
    "hc_create", "hc_destroy", "hc_set_thresholds", "hc_get_thresholds", "hc_upload", "hc_run", "hc_run_device",
    "hc_hysteresis_device", "hc_download", "hc_sync", "hc_set_stream", "hc_enable_profiling", "hc_stage_time_ms", "hc_profile_get",
    "hc_device_ptrs", "hc_last_hysteresis_info", "hc_hysteresis_stats", "hc_set_tuning", "hc_set_option", "hc_selftest", "hc_last_error", "hc_version",
    "hc_host_alloc", "hc_host_free", "hc_profile_get_front", "hc_debug_tap", "hc_use_own_stream", "hc_profile_get_intervals", "hc_last_run_info", "hc_pipeline_depth", "hc_pipeline_slots_in_use", "hc_front_waves_per_workgroup",
    "hc_profile_get_front_each", "hc_hysteresis_totals", "hc_download_begin", "hc_download_end",
]

_lib = None
_lib_legacy = None
LEGACY_LIB_PATH = os.path.join(_HERE, "libhipcanny_legacy.so")


class HipCannyError(RuntimeError):
    pass


def preload_hip_runtime():
    """One HIP runtime per process.  The PyTorch wheel bundles its own libamdhip64 (file name
    libamdhip64.so, SONAME libamdhip64.so.7 -- the same SONAME as /opt/rocm's).  If libhipcanny
    pulled in /opt/rocm's copy first, a later `import torch` would load the bundled one as well and
    the second runtime finds no GPU.  Loading torch's copy first (without importing torch) makes
    every later user -- this library, torch, oracle/_ref -- resolve to that single instance."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return None
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        return C.CDLL(cand, mode=C.RTLD_GLOBAL)
    return None


def load_library(legacy=False):
    """Loads libhipcanny.so (fails loudly when it has not been built).  legacy: libhipcanny_legacy.so instead -- the same
    sources plus the round-1 front kernels of Mode R (HC_OPT_FRONT_SPLIT 1 / 0), which the product library no longer
    contains; parity tests run every case through them as independent implementations."""
    global _lib, _lib_legacy
    if legacy and _lib_legacy is not None:
        return _lib_legacy
    if not legacy and _lib is not None:
        return _lib
    preload_hip_runtime()
    path = LEGACY_LIB_PATH if legacy else LIB_PATH
    if not os.path.exists(path):
        raise HipCannyError(f"{path} is missing: run `python -m cudacam_amd.build` (hipcc, gfx950). There is no CPU fallback.")
    L = C.CDLL(path)
    vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
    L.hc_create.restype = vp
    L.hc_create.argtypes = [i, i, i, i, i, i]
    L.hc_destroy.restype = None
    L.hc_destroy.argtypes = [vp]
    L.hc_set_thresholds.argtypes = [vp, i, i]
    L.hc_get_thresholds.argtypes = [vp, C.POINTER(i), C.POINTER(i)]
    L.hc_upload.argtypes = [vp, vp, sz, sz, i]
    L.hc_run.argtypes = [vp, i, i]
    L.hc_run_device.argtypes = [vp, vp, sz, sz, vp, sz, sz, i, i]
    L.hc_hysteresis_device.argtypes = [vp, vp, sz, sz, vp, sz, sz, i]
    L.hc_download.argtypes = [vp, vp, sz, sz, i]
    L.hc_download_begin.argtypes = [vp, vp, sz, sz, i]
    L.hc_download_end.argtypes = [vp]
    L.hc_sync.argtypes = [vp]
    L.hc_set_stream.argtypes = [vp, vp]
    L.hc_use_own_stream.argtypes = [vp]
    L.hc_enable_profiling.argtypes = [vp, i]
    L.hc_stage_time_ms.argtypes = [vp, i, C.POINTER(C.c_float)]
    L.hc_profile_get.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_long), i]
    L.hc_profile_get_front.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_long)]
    L.hc_device_ptrs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
    L.hc_last_hysteresis_info.argtypes = [vp, C.POINTER(i), C.POINTER(i)]
    L.hc_hysteresis_stats.argtypes = [vp, C.POINTER(C.c_uint), i]
    L.hc_set_tuning.argtypes = [vp, i, i]
    L.hc_set_option.argtypes = [vp, i, i]
    L.hc_selftest.argtypes = [i]
    L.hc_debug_tap.argtypes = [vp, i, vp, sz, sz, i]
    L.hc_last_run_info.argtypes = [vp, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    L.hc_pipeline_depth.argtypes = [vp, i]
    L.hc_pipeline_slots_in_use.argtypes = [vp]
    L.hc_front_waves_per_workgroup.argtypes = [vp]
    L.hc_profile_get_intervals.argtypes = [vp, C.POINTER(C.c_float), i, C.POINTER(i)]
    L.hc_profile_get_front_each.argtypes = [vp, C.POINTER(C.c_float), i, C.POINTER(i)]
    L.hc_hysteresis_totals.argtypes = [vp, C.POINTER(C.c_ulonglong), i]
    L.hc_host_alloc.restype = vp
    L.hc_host_alloc.argtypes = [sz]
    L.hc_host_free.restype = None
    L.hc_host_free.argtypes = [vp]
    L.hc_last_error.restype = C.c_char_p
    L.hc_version.restype = C.c_char_p
    for name in ABI_SYMBOLS:
        getattr(L, name)
    if legacy:
        _lib_legacy = L
    else:
        _lib = L
    return L


def last_error():
    msg = load_library().hc_last_error().decode()
    if _lib_legacy is not None:   # (a context of the test library keeps its error text there)
        other = _lib_legacy.hc_last_error().decode()
        if other and other != msg:
            msg = (msg + " | " if msg else "") + other
    return msg


def _ck(rc):
    if rc != 0:
        raise HipCannyError(f"hipcanny error {rc}: {last_error()}")


class Context:
    """Thin RAII wrapper of hc_ctx (one device, one stream)."""

    def __init__(self, width, height, channels=1, max_batch=1, mode=MODE_R, device=0, front_split=None):
        """front_split: HC_OPT_FRONT_SPLIT for the context's whole life (None: the library's default, k_front8 / k_front8o).
        The round-1 front kernels of Mode R (1: k_blur + k_nms, 0: the 4-px k_front) are not part of the product library:
        such a context is created in libhipcanny_legacy.so (parity tests, bench.py --front split / fused4)."""
        self.lib = load_library(legacy=front_split in (0, 1) and int(mode) == MODE_R)
        self.w, self.h, self.c, self.max_batch = int(width), int(height), int(channels), int(max_batch)
        self.handle = self.lib.hc_create(int(device), self.w, self.h, self.c, self.max_batch, int(mode))
        if not self.handle:
            raise HipCannyError(f"hc_create failed: {last_error()}")
        if front_split is not None:
            self.set_option(OPT_FRONT_SPLIT, front_split)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.hc_destroy(self.handle)
            self.handle = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_thresholds(self, low, high):
        _ck(self.lib.hc_set_thresholds(self.handle, int(low), int(high)))

    def get_thresholds(self):
        lo, hi = C.c_int(), C.c_int()
        _ck(self.lib.hc_get_thresholds(self.handle, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def set_tuning(self, chunk_rows=0, hyst_launches=0):
        _ck(self.lib.hc_set_tuning(self.handle, int(chunk_rows), int(hyst_launches)))

    def set_option(self, option, value):
        _ck(self.lib.hc_set_option(self.handle, int(option), int(value)))
        if int(option) == OPT_PER_CHANNEL:
            self._maps_per_frame = 3 if value else 1

    def set_stream(self, stream_handle):
        """Run on the caller's hipStream_t (0 = the null stream, e.g. torch's default current stream)."""
        _ck(self.lib.hc_set_stream(self.handle, C.c_void_p(stream_handle)))

    def use_own_stream(self):
        _ck(self.lib.hc_use_own_stream(self.handle))

    def enable_profiling(self, on):
        _ck(self.lib.hc_enable_profiling(self.handle, int(bool(on))))

    def stage_time_ms(self, stage):
        ms = C.c_float()
        _ck(self.lib.hc_stage_time_ms(self.handle, int(stage), C.byref(ms)))
        return ms.value

    def profile_get(self, reset=True):
        """(sum_ms[3], nruns) over all profiled runs since the last reset: stage 0 / front / hysteresis+expand."""
        sums = (C.c_double * 3)()
        n = C.c_long()
        _ck(self.lib.hc_profile_get(self.handle, sums, C.byref(n), int(bool(reset))))
        return [sums[0], sums[1], sums[2]], n.value

    def profile_intervals(self, cap=4096):
        """End-of-run to end-of-run times (ms) of consecutive profiled runs since the last reset."""
        buf = (C.c_float * cap)()
        n = C.c_int()
        _ck(self.lib.hc_profile_get_intervals(self.handle, buf, cap, C.byref(n)))
        return [buf[k] for k in range(min(cap, n.value))]

    def profile_front_each(self, cap=4096):
        """The front kernels' time (ms) of every profiled HYSTER run since the last reset, in run order."""
        buf = (C.c_float * cap)()
        n = C.c_int()
        _ck(self.lib.hc_profile_get_front_each(self.handle, buf, cap, C.byref(n)))
        return [buf[k] for k in range(min(cap, n.value))]

    def hysteresis_totals(self, reset=False):
        """(runs, runs continued from the host, launches with work, launches queued) since creation / the last reset."""
        t = (C.c_ulonglong * 4)()
        _ck(self.lib.hc_hysteresis_totals(self.handle, t, int(bool(reset))))
        return tuple(int(v) for v in t)

    def profile_get_front(self):
        """([k_blur_ms_sum, k_nms_ms_sum], nruns) of the profiled runs that took the split front path; call before
        profile_get(reset=True)."""
        sums = (C.c_double * 2)()
        n = C.c_long()
        _ck(self.lib.hc_profile_get_front(self.handle, sums, C.byref(n)))
        return [sums[0], sums[1]], n.value

    def upload(self, frames):
        """frames: (n,H,W) / (n,H,W,3) uint8, or a single frame."""
        a = np.ascontiguousarray(frames, dtype=np.uint8)
        if a.ndim == (2 if self.c == 1 else 3):
            a = a[None]
        exp = (self.h, self.w) if self.c == 1 else (self.h, self.w, 3)
        if a.shape[1:] != exp:
            raise HipCannyError(f"Cannot load image to GPU, specs different since initialization: {a.shape[1:]} vs {exp}")
        row = self.w * self.c
        _ck(self.lib.hc_upload(self.handle, a.ctypes.data, row, row * self.h, a.shape[0]))
        self._keep = a
        return a.shape[0]

    def run(self, final_stage=CannyStage.HYSTER, nframes=1):
        _ck(self.lib.hc_run(self.handle, int(final_stage), int(nframes)))

    def download(self, nframes=1):
        """nframes output images (in per-channel mode: 3 per input frame)."""
        out = np.empty((nframes, self.h, self.w), np.uint8)
        _ck(self.lib.hc_download(self.handle, out.ctypes.data, self.w, self.w * self.h, nframes))
        return out

    def sync(self):
        _ck(self.lib.hc_sync(self.handle))

    def run_device(self, d_in, in_pitch, in_fs, d_out, out_pitch, out_fs, nframes, final_stage=CannyStage.HYSTER):
        _ck(self.lib.hc_run_device(self.handle, C.c_void_p(d_in), in_pitch, in_fs, C.c_void_p(d_out), out_pitch, out_fs,
                                   int(nframes), int(final_stage)))

    def hysteresis_device(self, d_thr, in_pitch, in_fs, d_out, out_pitch, out_fs, nframes):
        _ck(self.lib.hc_hysteresis_device(self.handle, C.c_void_p(d_thr), in_pitch, in_fs, C.c_void_p(d_out), out_pitch,
                                          out_fs, int(nframes)))

    def front_waves_per_workgroup(self):
        """Waves per workgroup of the most recent k_front8 launch (hc_front_waves_per_workgroup): 4, 1 or 3."""
        r = self.lib.hc_front_waves_per_workgroup(self.handle)
        if r < 0:
            _ck(r)
        return r

    def pipeline_slots_in_use(self):
        """Slots of the ring the most recent pipelined run used (hc_pipeline_slots_in_use)."""
        r = self.lib.hc_pipeline_slots_in_use(self.handle)
        if r < 0:
            _ck(r)
        return r

    def pipeline_depth(self, nframes):
        """Runs of `nframes` frames kept in flight in pipelined mode (hc_pipeline_depth): the size of the output-buffer ring."""
        r = self.lib.hc_pipeline_depth(self.handle, int(nframes))
        if r < 0:
            _ck(r)
        return r

    def last_run_info(self):
        """(input_staged, output_staged, front_form) of the last run: see hc_last_run_info."""
        a, b, f = C.c_int(), C.c_int(), C.c_int()
        _ck(self.lib.hc_last_run_info(self.handle, C.byref(a), C.byref(b), C.byref(f)))
        return bool(a.value), bool(b.value), f.value

    def hysteresis_info(self):
        a, b = C.c_int(), C.c_int()
        _ck(self.lib.hc_last_hysteresis_info(self.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def hysteresis_stats(self, launches=6):
        """[(sweeps_sum, sweeps_max, active_tiles)] per queued hysteresis launch of the last run."""
        buf = (C.c_uint * (3 * launches))()
        _ck(self.lib.hc_hysteresis_stats(self.handle, buf, 3 * launches))
        return [(buf[3 * k], buf[3 * k + 1], buf[3 * k + 2]) for k in range(launches)]

    def debug_tap(self, what, nframes=1):
        """The FAST path's own intermediate (TAP_BLUR / TAP_THRESH) of the last HYSTER run; needs OPT_DEBUG_TAPS."""
        out = np.empty((nframes, self.h, self.w), np.uint8)
        _ck(self.lib.hc_debug_tap(self.handle, int(what), out.ctypes.data, self.w, self.w * self.h, int(nframes)))
        return out

    def process(self, frames, final_stage=CannyStage.HYSTER):
        """upload -> run -> download convenience."""
        n = self.upload(frames)
        self.run(final_stage, n)
        return self.download(n * getattr(self, "_maps_per_frame", 1))   # per-channel mode: three maps per input frame


def selftest(device=0):
    _ck(load_library().hc_selftest(int(device)))


class TimerManager:
    """timerManager singleton, src/utils/timer.hpp:13-67 (running averages keyed by stage name)."""
    _inst = None

    def __init__(self):
        self._timers = {}

    @classmethod
    def Get(cls):
        if cls._inst is None:
            cls._inst = cls()
        return cls._inst

    def createTimer(self, name):
        self._timers.setdefault(name, [0.0, 0])

    def addTime(self, name, t):
        if name in self._timers:
            self._timers[name][0] += t
            self._timers[name][1] += 1
        else:
            print(f"Timer {name} unknown", file=sys.stderr)

    def getAverageTime(self, name):
        t = self._timers.get(name)
        if t and t[1] > 0:
            return t[0] / t[1]
        print(f"Timer {name} unknown", file=sys.stderr)
        return 0.0

    def timers(self):
        return dict(self._timers)


def _mat_type(mat):
    if not isinstance(mat, np.ndarray) or mat.dtype != np.uint8:
        return None
    if mat.ndim == 2:
        return "CV_8UC1"
    if mat.ndim == 3 and mat.shape[2] == 3:
        return "CV_8UC3"
    return None


class CannyEdge:
    """Mirror of cvp::cuda::CannyEdge (src/cvp/cannyEdgeH.hpp:17-32).

    pbo: the reference's OpenGL pixel-buffer id; a headless MI355X has no GL, so it must be 0 and
    the result is read with output() instead.
    """

    def __init__(self, pbo, imageWidth, imageHeight, imageNbChannels, mode=MODE_R, device=0):
        if pbo != 0:
            raise HipCannyError("GL interop is not available on MI355X: pass pbo=0 and read output()")
        self._ctx = Context(imageWidth, imageHeight, imageNbChannels, 1, mode, device)
        self.m_inputW, self.m_inputH, self.m_inputNbChannels = imageWidth, imageHeight, imageNbChannels
        self._ctx.set_thresholds(10, 40) if mode == MODE_R else None  # cannyEdgeH.cu:22-23
        self._profiling = True                                          # cannyEdgeH.cu:24
        self._ctx.enable_profiling(True)
        self._out = None
        tm = TimerManager.Get()
        for name in CANNY_STAGES.values():                              # cannyEdgeH.cu:35-37
            tm.createTimer(name)

    def run(self, input, finalStage):
        """cannyEdgeH.cu:49-120.  Size/channel mismatch: logged, frame not processed (:124-130)."""
        exp = (self.m_inputH, self.m_inputW) if self.m_inputNbChannels == 1 else (self.m_inputH, self.m_inputW, 3)
        if input.shape != exp:
            print("Cannot load image to GPU, specs different since initialization", file=sys.stderr)
            return
        try:
            stage = CannyStage(int(finalStage))
        except ValueError:
            print("Canny Stage Not Recognized", file=sys.stderr)
            return
        self._ctx.upload(input)
        self._ctx.run(stage, 1)
        self._out = self._ctx.download(1)[0]
        if self._profiling:
            tm = TimerManager.Get()
            for st in CannyStage:  # one sample per stage that ran (cannyEdgeH.cu:415-430); -1 = the run did not execute it
                ms = self._ctx.stage_time_ms(st)
                if ms >= 0:
                    tm.addTime(CANNY_STAGES[st], ms)

    def output(self):
        return self._out

    def setLowThreshold(self, low):
        lo, hi = self._ctx.get_thresholds()
        self._ctx.set_thresholds(min(int(low) & 0xFF, hi), hi)     # cannyEdgeH.hpp:25

    def getLowThreshold(self):
        return self._ctx.get_thresholds()[0]

    def setHighThreshold(self, high):
        lo, hi = self._ctx.get_thresholds()
        self._ctx.set_thresholds(lo, max(int(high) & 0xFF, lo))    # cannyEdgeH.hpp:28

    def getHighThreshold(self):
        return self._ctx.get_thresholds()[1]

    def enableKernelProfiling(self, profiling):
        self._profiling = bool(profiling)
        self._ctx.enable_profiling(self._profiling)

    def isKernelProfilingEnabled(self):
        return self._profiling


class cvPipeline:
    """Mirror of cvp::cvPipeline (src/cvp/cvPipeline.hpp:20-39, cvPipeline.cpp:9-96)."""

    def __init__(self, pbo, inputImageCols, inputImageRows, inputImageNbChannels, mode=MODE_R, device=0):
        self.m_cudaCannyEdge = CannyEdge(pbo, inputImageCols, inputImageRows, inputImageNbChannels, mode, device)

    def process(self, inputImage, finalStage):
        if self.m_cudaCannyEdge is None:
            print("Cannot process the webcam stream, Cuda is not ready.", file=sys.stderr)
            return False
        if inputImage is None or getattr(inputImage, "size", 0) == 0:
            print("Blank frame grabbed", file=sys.stderr)                       # cvPipeline.cpp:27-31
            return False
        if _mat_type(inputImage) is None:
            print("Only supporting CV_8UC3 and CV_8UC1 input types for now", file=sys.stderr)  # :32-36
            return False
        self.m_cudaCannyEdge.run(inputImage, finalStage)
        return True

    def output(self):
        return self.m_cudaCannyEdge.output()

    def setLowThreshold(self, low):
        self.m_cudaCannyEdge.setLowThreshold(low)

    def getLowThreshold(self):
        return self.m_cudaCannyEdge.getLowThreshold()

    def setHighThreshold(self, high):
        self.m_cudaCannyEdge.setHighThreshold(high)

    def getHighThreshold(self):
        return self.m_cudaCannyEdge.getHighThreshold()

    def enableCudaProfiling(self, profiling):
        self.m_cudaCannyEdge.enableKernelProfiling(profiling)

    def isCudaProfilingEnabled(self):
        return self.m_cudaCannyEdge.isKernelProfilingEnabled() if self.m_cudaCannyEdge else False
