"""orb-slam3_amd -- MI355X-native ORB front-end (extract + match) for ORB-SLAM3.

Python host mirror of the reference's C++ operator interface, for tests / bench / scripting:
    ORBextractor  <->  ORB_SLAM3::ORBextractor   (include/ORBextractor.h:49-83)
    ORBmatcher    <->  ORB_SLAM3::ORBmatcher     (include/ORBmatcher.h:35-111)  -- flattened-array form
Everything computes through the C ABI of liborbslam3_amd.so (include/orbx.h, include/orbm.h), i.e. the
hand-written gfx950 kernels.  There is NO CPU fallback: a missing library or device raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liborbslam3_amd.so")
_LIB = None

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])

HOST, DEVICE = 0, 1
TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30          # ORBmatcher.cc:36-38

EXPORTS = [
    # include/orbx.h
    "orbx_create", "orbx_destroy", "orbx_last_error", "orbx_max_keypoints", "orbx_extract", "orbx_extract_batch",
    "orbx_extract_batch_async", "orbx_sync", "orbx_result_device", "orbx_result_fetch", "orbx_level_size",
    "orbx_level_image", "orbx_scale_tables", "orbx_features_per_level", "orbx_level_candidates",
    "orbx_level_selected", "orbx_last_timings", "orbx_algorithmic_bytes", "orbx_stream", "orbx_debug_stamps", "orbx_dev_alloc",
    "orbx_dev_free", "orbx_memcpy_h2d", "orbx_memcpy_d2h", "orbx_device_count",
    # include/orbm.h
    "orbm_create", "orbm_destroy", "orbm_last_error", "orbm_sync", "orbm_stream", "orbm_hamming",
    "orbm_three_maxima", "orbm_knn2_batch", "orbm_knn2_batch_async", "orbm_last_timing",
]


class OrbError(RuntimeError):
    pass


def build(force=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc")] + (["-B"] if force else []))
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise OrbError("liborbslam3_amd.so is not built (run __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        vp, ci, i32p = C.c_void_p, C.c_int, C.POINTER(C.c_int32)
        L.orbx_create.argtypes = [C.POINTER(vp), ci, C.c_float, ci, ci, ci, ci, ci, ci, ci]
        L.orbx_destroy.argtypes = [vp]
        L.orbx_last_error.restype = C.c_char_p
        L.orbx_max_keypoints.argtypes = [vp]
        L.orbx_extract.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, vp, ci, i32p]
        L.orbx_extract_batch.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, vp, vp]
        L.orbx_extract_batch_async.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp]
        L.orbx_sync.argtypes = [vp]
        L.orbx_result_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32p]
        L.orbx_result_fetch.argtypes = [vp, ci, vp, vp, ci, i32p]
        L.orbx_level_size.argtypes = [vp, ci, i32p, i32p]
        L.orbx_level_image.argtypes = [vp, ci, ci, ci, vp, ci]
        L.orbx_scale_tables.argtypes = [vp, vp, vp, vp, vp]
        L.orbx_features_per_level.argtypes = [vp, vp]
        L.orbx_level_candidates.argtypes = [vp, ci, ci, vp, ci]
        L.orbx_level_selected.argtypes = [vp, ci, ci, vp, ci]
        L.orbx_last_timings.argtypes = [vp, vp]
        L.orbx_algorithmic_bytes.restype = C.c_int64
        L.orbx_algorithmic_bytes.argtypes = [vp, C.POINTER(C.c_int64)]
        L.orbx_debug_stamps.argtypes = [vp, vp, ci]
        L.orbx_stream.restype = vp
        L.orbx_stream.argtypes = [vp]
        L.orbx_dev_alloc.restype = vp
        L.orbx_dev_alloc.argtypes = [C.c_size_t]
        L.orbx_dev_free.argtypes = [vp]
        L.orbx_memcpy_h2d.argtypes = [vp, vp, C.c_size_t]
        L.orbx_memcpy_d2h.argtypes = [vp, vp, C.c_size_t]
        L.orbm_create.argtypes = [C.POINTER(vp), ci]
        L.orbm_destroy.argtypes = [vp]
        L.orbm_last_error.restype = C.c_char_p
        L.orbm_sync.argtypes = [vp]
        L.orbm_stream.restype = vp
        L.orbm_stream.argtypes = [vp]
        L.orbm_hamming.argtypes = [vp, vp]
        L.orbm_three_maxima.argtypes = [vp, ci, vp]
        L.orbm_knn2_batch.argtypes = [vp, ci, vp, ci, vp, vp, ci, vp, ci, vp, vp]
        L.orbm_knn2_batch_async.argtypes = [vp, vp, ci, vp, vp, ci, vp, ci, ci, vp, vp]
        L.orbm_last_timing.argtypes = [vp, vp]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _chk(rc, what):
    if rc < 0:
        L = lib()
        msg = (L.orbx_last_error() or b"").decode() or (L.orbm_last_error() or b"").decode()
        raise OrbError("%s failed with code %d: %s" % (what, rc, msg))
    return rc


class DeviceBuffer:
    """hipMalloc'd bytes owned by Python (keeps tests/bench free of any other HIP binding)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = lib().orbx_dev_alloc(max(1, self.nbytes))
        if not self.ptr:
            raise OrbError("device allocation of %d bytes failed" % nbytes)

    def upload(self, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        _chk(lib().orbx_memcpy_h2d(self.ptr + offset, _p(arr), arr.nbytes), "h2d")
        return self

    def download(self, dtype, count, offset=0):
        out = np.empty(count, dtype)
        _chk(lib().orbx_memcpy_d2h(_p(out), self.ptr + offset, out.nbytes), "d2h")
        return out

    def __del__(self):
        try:
            lib().orbx_dev_free(self.ptr)
        except Exception:
            pass


class ORBextractor:
    """ORB_SLAM3::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) on one MI355X.

    __call__(image, lapping) -> (monoIndex, keypoints, descriptors) mirrors operator()
    (src/ORBextractor.cc:1534-1659): keypoints is a structured array with cv::KeyPoint's layout,
    descriptors an (n,32) uint8 array, monoIndex the reference's return value (-1 for an empty image).
    """

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7,
                 device=0, max_size=(752, 480), max_batch=1):
        self.L = lib()
        h = C.c_void_p()
        _chk(self.L.orbx_create(C.byref(h), nfeatures, scale_factor, nlevels, ini_th, min_th, device,
                                int(max_size[0]), int(max_size[1]), int(max_batch)), "orbx_create")
        self.h = h
        self.nfeatures, self.nlevels, self.max_batch = nfeatures, nlevels, max_batch
        self.cap = self.L.orbx_max_keypoints(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.L.orbx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- getters (include/ORBextractor.h:61-79)
    def GetLevels(self):
        return self.nlevels

    def _tables(self):
        n = self.nlevels
        t = [np.zeros(n, np.float32) for _ in range(4)]
        self.L.orbx_scale_tables(self.h, *[_p(a) for a in t])
        return t

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def GetScaleFactor(self):
        return float(self._tables()[0][1]) if self.nlevels > 1 else 1.0

    def features_per_level(self):
        out = np.zeros(self.nlevels, np.int32)
        self.L.orbx_features_per_level(self.h, _p(out))
        return out

    # ---- operator()
    def __call__(self, image, lapping=(0, 0)):
        if image is None or image.size == 0:
            return -1, None, None
        img = np.ascontiguousarray(image)
        assert img.dtype == np.uint8 and img.ndim == 2, "image must be CV_8UC1"
        h, w = img.shape
        kps = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        mono = C.c_int32(0)
        n = _chk(self.L.orbx_extract(self.h, _p(img), w, h, w, int(lapping[0]), int(lapping[1]), _p(kps), _p(desc),
                                     self.cap, C.byref(mono)), "orbx_extract")
        return mono.value, kps[:n].copy(), desc[:n].copy()

    # ---- frame-parallel batch
    def extract_batch(self, images, lapping=None):
        """images: list of equal-size uint8 arrays (host).  Returns list of (mono, kps, desc)."""
        imgs = [np.ascontiguousarray(i, np.uint8) for i in images]
        h, w = imgs[0].shape
        ptrs = (C.c_void_p * len(imgs))(*[i.ctypes.data for i in imgs])
        lap = None if lapping is None else np.ascontiguousarray(np.asarray(lapping, np.int32).reshape(-1))
        n_out = np.zeros(len(imgs), np.int32); m_out = np.zeros(len(imgs), np.int32)
        _chk(self.L.orbx_extract_batch(self.h, ptrs, HOST, len(imgs), w, h, w, None if lap is None else _p(lap),
                                       _p(n_out), _p(m_out)), "orbx_extract_batch")
        return [self.fetch(i) for i in range(len(imgs))]

    def enqueue_device(self, dev_ptrs, w, h, stride, lapping=None):
        """dev_ptrs: ctypes array of device pointers; enqueue only (bench hot loop)."""
        lap = None if lapping is None else np.ascontiguousarray(np.asarray(lapping, np.int32).reshape(-1))
        self._keep = (dev_ptrs, lap)
        _chk(self.L.orbx_extract_batch_async(self.h, dev_ptrs, DEVICE, len(dev_ptrs), w, h, stride,
                                             None if lap is None else _p(lap)), "orbx_extract_batch_async")

    def sync(self):
        _chk(self.L.orbx_sync(self.h), "orbx_sync")

    def fetch(self, i):
        kps = np.zeros(self.cap, KP_DTYPE); desc = np.zeros((self.cap, 32), np.uint8)
        mono = C.c_int32(0)
        n = _chk(self.L.orbx_result_fetch(self.h, i, _p(kps), _p(desc), self.cap, C.byref(mono)), "orbx_result_fetch")
        return mono.value, kps[:n].copy(), desc[:n].copy()

    def result_device(self):
        k, d, c, m = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        cap = C.c_int32()
        _chk(self.L.orbx_result_device(self.h, C.byref(k), C.byref(d), C.byref(c), C.byref(m), C.byref(cap)), "result_device")
        return dict(kps=k.value, desc=d.value, counts=c.value, monos=m.value, cap=cap.value)

    # ---- stage introspection
    def level_size(self, level):
        w, h = C.c_int32(), C.c_int32()
        _chk(self.L.orbx_level_size(self.h, level, C.byref(w), C.byref(h)), "level_size")
        return w.value, h.value

    def level_image(self, level, frame=0, blurred=False):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        _chk(self.L.orbx_level_image(self.h, frame, level, int(blurred), _p(out), w), "level_image")
        return out

    def level_candidates(self, level, frame=0):
        n = _chk(self.L.orbx_level_candidates(self.h, frame, level, None, 0), "level_candidates")
        out = np.zeros((max(n, 1), 3), np.int32)
        self.L.orbx_level_candidates(self.h, frame, level, _p(out), n)
        return out[:n]

    def level_selected(self, level, frame=0):
        n = _chk(self.L.orbx_level_selected(self.h, frame, level, None, 0), "level_selected")
        out = np.zeros((max(n, 1), 3), np.int32)
        self.L.orbx_level_selected(self.h, frame, level, _p(out), n)
        return out[:n]

    def timings(self):
        t = np.zeros(7, np.float32)
        _chk(self.L.orbx_last_timings(self.h, _p(t)), "timings")
        return dict(zip(["pyramid", "fast", "quadtree", "slots", "blur", "orient_desc", "total"], t.tolist()))

    def algorithmic_bytes(self):
        f = C.c_int64()
        a = self.L.orbx_algorithmic_bytes(self.h, C.byref(f))
        return int(a), int(f.value)


class ORBmatcher:
    """Flattened-array form of ORB_SLAM3::ORBmatcher (include/ORBmatcher.h:35-111)."""

    TH_HIGH, TH_LOW, HISTO_LENGTH = TH_HIGH, TH_LOW, HISTO_LENGTH

    def __init__(self, nnratio=0.6, check_ori=True, device=0):
        self.L = lib()
        self.nnratio, self.check_ori = float(nnratio), bool(check_ori)
        h = C.c_void_p()
        _chk(self.L.orbm_create(C.byref(h), device), "orbm_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.orbm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def DescriptorDistance(a, b):
        a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
        return lib().orbm_hamming(_p(a), _p(b))

    @staticmethod
    def ComputeThreeMaxima(bin_sizes):
        c = np.ascontiguousarray(bin_sizes, np.int32); out = np.zeros(3, np.int32)
        lib().orbm_three_maxima(_p(c), c.shape[0], _p(out))
        return out

    def knn2(self, q, t):
        """2-NN of every row of q among the rows of t (BFMatcher.knnMatch k=2, Frame.cc:1458)."""
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32); t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        nq = np.array([q.shape[0]], np.int32); nt = np.array([t.shape[0]], np.int32)
        idx = np.zeros((max(q.shape[0], 1), 2), np.int32); dist = np.zeros_like(idx)
        if q.shape[0] == 0:
            return idx[:0], dist[:0]
        tt = t if t.shape[0] else np.zeros((1, 32), np.uint8)
        _chk(self.L.orbm_knn2_batch(self.h, HOST, _p(q), q.shape[0], _p(nq), _p(tt), tt.shape[0], _p(nt), 1,
                                    _p(idx), _p(dist)), "orbm_knn2_batch")
        return idx, dist

    def sync(self):
        _chk(self.L.orbm_sync(self.h), "orbm_sync")

    def timing_ms(self):
        t = C.c_float()
        _chk(self.L.orbm_last_timing(self.h, C.byref(t)), "orbm_last_timing")
        return t.value
