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
LIB_PATH = os.environ.get("ORB_LIB") or os.path.join(_HERE, "liborbslam3_amd.so")   # ORB_LIB: A/B builds of the same ABI
_LIB = None

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])

HOST, DEVICE = 0, 1
TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30          # ORBmatcher.cc:36-38

EXPORTS = [
    # include/orbx.h
    "orbx_create", "orbx_destroy", "orbx_last_error", "orbx_max_keypoints", "orbx_extract", "orbx_extract_batch",
    "orbx_extract_batch_async", "orbx_sync", "orbx_result_device", "orbx_result_fetch", "orbx_result_fetch_all", "orbx_level_size",
    "orbx_level_image", "orbx_pyramid_fetch", "orbx_pyramid_map", "orbx_scale_tables", "orbx_features_per_level", "orbx_level_candidates",
    "orbx_level_selected", "orbx_last_timings", "orbx_set_stage_timing", "orbx_mean_timings", "orbx_stream_wait_results", "orbx_stream_wait_other", "orbx_guard_results", "orbx_gray_from_color", "orbx_remap_linear", "orbx_clahe", "orbx_algorithmic_bytes", "orbx_blur_in_pass", "orbx_stream", "orbx_dev_alloc",
    "orbx_dev_free", "orbx_memcpy_h2d", "orbx_memcpy_d2h", "orbx_device_count",
    "orbx_capture_begin", "orbx_capture_end", "orbx_graph_launch", "orbx_result_download_async", "orbx_result_block_layout", "orbx_block_attach", "orbx_block_detach_all", "orbx_download_sync",
    "orbx_host_alloc", "orbx_host_free", "orbx_set_result_block", "orbx_mark", "orbx_mark_elapsed_ms",
    # include/orbm.h
    "orbm_create", "orbm_destroy", "orbm_last_error", "orbm_sync", "orbm_stream", "orbm_set_stream", "orbm_hamming",
    "orbm_three_maxima", "orbm_knn2_batch", "orbm_knn2_batch_async", "orbm_knn2_ratio_batch_async", "orbm_last_timing",
    "orbm_stereo_batch_async", "orbm_bow_nodes_batch_async", "orbm_triangulation_batch_async",
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
        L.orbx_result_fetch_all.argtypes = [vp, vp, vp, ci, vp, vp]
        L.orbx_level_size.argtypes = [vp, ci, i32p, i32p]
        L.orbx_level_image.argtypes = [vp, ci, ci, ci, vp, ci]
        L.orbx_scale_tables.argtypes = [vp, vp, vp, vp, vp]
        L.orbx_features_per_level.argtypes = [vp, vp]
        L.orbx_level_candidates.argtypes = [vp, ci, ci, vp, ci]
        L.orbx_level_selected.argtypes = [vp, ci, ci, vp, ci]
        L.orbx_last_timings.argtypes = [vp, vp]
        L.orbx_algorithmic_bytes.restype = C.c_int64
        L.orbx_blur_in_pass.argtypes = [vp]
        L.orbx_algorithmic_bytes.argtypes = [vp, C.POINTER(C.c_int64)]
        L.orbx_mean_timings.argtypes = [vp, vp, i32p]
        L.orbx_stream_wait_results.argtypes = [vp, vp]
        L.orbx_stream_wait_other.argtypes = [vp, vp]
        L.orbx_guard_results.argtypes = [vp, vp]
        L.orbx_set_stage_timing.argtypes = [vp, C.c_int]
        L.orbx_clahe.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, vp, C.c_int]
        L.orbx_remap_linear.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, vp, C.c_int]
        L.orbx_gray_from_color.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
        L.orbx_stream.restype = vp
        L.orbx_stream.argtypes = [vp]
        L.orbx_dev_alloc.restype = vp
        L.orbx_dev_alloc.argtypes = [C.c_size_t]
        L.orbx_dev_free.argtypes = [vp]
        L.orbx_memcpy_h2d.argtypes = [vp, vp, C.c_size_t]
        L.orbx_memcpy_d2h.argtypes = [vp, vp, C.c_size_t]
        L.orbx_capture_begin.argtypes = [vp, ci]
        L.orbx_capture_end.argtypes = [vp]
        L.orbx_graph_launch.argtypes = [vp, ci]
        L.orbx_result_download_async.argtypes = [vp, vp]
        L.orbx_result_block_layout.argtypes = [vp] + [C.POINTER(C.c_size_t)] * 5
        L.orbx_download_sync.argtypes = [vp]
        L.orbx_block_attach.argtypes = [vp, ci, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.orbx_block_detach_all.argtypes = [vp]
        L.orbx_set_result_block.argtypes = [vp, ci]
        L.orbx_host_alloc.restype = vp
        L.orbx_host_alloc.argtypes = [C.c_size_t]
        L.orbx_host_free.argtypes = [vp]
        L.orbx_mark.argtypes = [vp, ci]
        L.orbx_mark_elapsed_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.orbm_create.argtypes = [C.POINTER(vp), ci]
        L.orbm_destroy.argtypes = [vp]
        L.orbm_last_error.restype = C.c_char_p
        L.orbm_sync.argtypes = [vp]
        L.orbm_stream.restype = vp
        L.orbm_stream.argtypes = [vp]
        L.orbm_set_stream.argtypes = [vp, vp]
        L.orbm_hamming.argtypes = [vp, vp]
        L.orbm_three_maxima.argtypes = [vp, ci, vp]
        L.orbm_knn2_batch.argtypes = [vp, ci, vp, ci, vp, vp, ci, vp, ci, vp, vp]
        L.orbm_knn2_batch_async.argtypes = [vp, vp, ci, vp, vp, ci, vp, ci, ci, vp, vp]
        L.orbm_knn2_ratio_batch_async.argtypes = [vp, vp, ci, vp, vp, ci, vp, ci, C.c_double, vp, vp, vp]
        L.orbm_last_timing.argtypes = [vp, vp]
        L.orbm_stereo_batch_async.argtypes = [vp, vp, ci, ci, ci, vp, vp, vp, ci, C.c_float, C.c_float, vp, vp, vp, vp]
        L.orbm_bow_nodes_batch_async.argtypes = [vp, vp, vp, ci, ci, vp]
        L.orbm_triangulation_batch_async.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, vp, vp, ci, ci, ci, vp, vp]
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


class PinnedBuffer:
    """hipHostMalloc'd bytes with a numpy view (destination of orbx_result_download_async)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = lib().orbx_host_alloc(max(1, self.nbytes))
        if not self.ptr:
            raise OrbError("pinned allocation of %d bytes failed" % nbytes)
        self.bytes = np.ctypeslib.as_array((C.c_uint8 * max(1, self.nbytes)).from_address(self.ptr))

    def view(self, dtype, shape):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return self.bytes[:n].view(dtype).reshape(shape)

    def __del__(self):
        try:
            self.bytes = None
            lib().orbx_host_free(self.ptr)
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
        return self.fetch_all()

    def fetch_all(self):
        """Results of every frame of the last batch with one device-to-host copy per array."""
        B = self.L.orbx_result_fetch_all(self.h, None, None, self.cap, None, None)
        _chk(B, "orbx_result_fetch_all")
        kps = np.zeros((B, self.cap), KP_DTYPE); desc = np.zeros((B, self.cap, 32), np.uint8)
        n = np.zeros(B, np.int32); m = np.zeros(B, np.int32)
        _chk(self.L.orbx_result_fetch_all(self.h, _p(kps), _p(desc), self.cap, _p(n), _p(m)), "orbx_result_fetch_all")
        return [(int(m[i]), kps[i, :n[i]].copy(), desc[i, :n[i]].copy()) for i in range(B)]

    def gray_from_color(self, images, blue_first=False, coef_bits=15):
        """cv::cvtColor(..., COLOR_*2GRAY) of equal-size HxWx{3,4} uint8 host images on the GPU (Tracking.cc:1264-1290).
        Returns (DeviceBuffer, stride): the gray images back to back, usable by enqueue_device / extract."""
        imgs = [np.ascontiguousarray(i, np.uint8) for i in images]
        h, w, ch = imgs[0].shape
        stride = (w + 63) // 64 * 64
        buf = DeviceBuffer(stride * h * len(imgs))
        srcs = (C.c_void_p * len(imgs))(*[i.ctypes.data for i in imgs])
        dsts = (C.c_void_p * len(imgs))(*[buf.ptr + k * stride * h for k in range(len(imgs))])
        _chk(self.L.orbx_gray_from_color(self.h, srcs, HOST, len(imgs), w, h, w * ch, ch, 1 if blue_first else 0, int(coef_bits), dsts, stride),
             "orbx_gray_from_color")
        self.sync()                                        # enqueue-only in the C ABI; synced here because callers download the buffer
        return buf, stride

    def clahe(self, images, clip_limit=3.0, tiles=(8, 8)):
        """cv::createCLAHE(clip, Size(tiles))->apply on equal-size uint8 host images, in place on the GPU
        (mono_tum_vi.cc:101-109).  Returns (DeviceBuffer, stride)."""
        imgs = [np.ascontiguousarray(i, np.uint8) for i in images]
        h, w = imgs[0].shape
        stride = (w + 63) // 64 * 64
        buf = DeviceBuffer(stride * h * len(imgs))
        pad = np.zeros((h, stride), np.uint8)
        for k, im in enumerate(imgs):
            pad[:, :w] = im
            buf.upload(pad, offset=k * stride * h)
        ptrs = (C.c_void_p * len(imgs))(*[buf.ptr + k * stride * h for k in range(len(imgs))])
        _chk(self.L.orbx_clahe(self.h, ptrs, len(imgs), w, h, stride, float(clip_limit), int(tiles[0]), int(tiles[1]), ptrs, stride), "orbx_clahe")
        self.sync()
        return buf, stride

    def remap_linear(self, images, mapx, mapy):
        """cv::remap(img, M1, M2, INTER_LINEAR) of equal-size uint8 host images on the GPU (stereo_euroc.cc:168-169).
        Returns (DeviceBuffer, stride) with the rectified images back to back."""
        imgs = [np.ascontiguousarray(i, np.uint8) for i in images]
        sh, sw = imgs[0].shape
        mx = np.ascontiguousarray(mapx, np.float32); my = np.ascontiguousarray(mapy, np.float32)
        dh, dw = mx.shape
        sstride = (sw + 63) // 64 * 64; dstride = (dw + 63) // 64 * 64
        src = DeviceBuffer(sstride * sh * len(imgs)); dst = DeviceBuffer(dstride * dh * len(imgs))
        pad = np.zeros((sh, sstride), np.uint8)
        for k, im in enumerate(imgs):
            pad[:, :sw] = im
            src.upload(pad, offset=k * sstride * sh)
        dmx = DeviceBuffer(mx.nbytes).upload(mx); dmy = DeviceBuffer(my.nbytes).upload(my)
        sp = (C.c_void_p * len(imgs))(*[src.ptr + k * sstride * sh for k in range(len(imgs))])
        dp = (C.c_void_p * len(imgs))(*[dst.ptr + k * dstride * dh for k in range(len(imgs))])
        _chk(self.L.orbx_remap_linear(self.h, sp, len(imgs), sw, sh, sstride, dmx.ptr, dmy.ptr, dw, dh, dp, dstride), "orbx_remap_linear")
        self.sync()                                        # the source / map buffers above are released when this returns
        return dst, dstride

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

    def result_block_layout(self):
        """(off_kps, off_desc, off_counts, off_monos, bytes) of a result block (include/orbx.h: orbx_result_block_layout)."""
        v = [C.c_size_t() for _ in range(5)]
        _chk(self.L.orbx_result_block_layout(self.h, *[C.byref(x) for x in v]), "result_block_layout")
        return tuple(int(x.value) for x in v)

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

    def pyramid(self, frame=0):
        """every unblurred level of one frame with one call (orbx_pyramid_fetch)."""
        outs = [np.zeros(self.level_size(l)[::-1], np.uint8) for l in range(self.GetLevels())]
        ptrs = (C.c_void_p * len(outs))(*[o.ctypes.data for o in outs])
        strides = (C.c_int * len(outs))(*[o.shape[1] for o in outs])
        self.L.orbx_pyramid_fetch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _chk(self.L.orbx_pyramid_fetch(self.h, frame, ptrs, strides), "pyramid_fetch")
        return outs

    def pyramid_views(self, frame=0):
        """zero-copy views of every level in the handle's pinned memory (orbx_pyramid_map); valid until the next call on the handle."""
        n = self.GetLevels()
        ptrs = (C.c_void_p * n)(); pitch = (C.c_int * n)()
        self.L.orbx_pyramid_map.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _chk(self.L.orbx_pyramid_map(self.h, frame, ptrs, pitch), "pyramid_map")
        out = []
        for l in range(n):
            w, h = self.level_size(l)
            if not ptrs[l]:
                out.append(None); continue
            buf = (C.c_uint8 * (pitch[l] * h)).from_address(ptrs[l])
            out.append(np.frombuffer(buf, np.uint8).reshape(h, pitch[l])[:, :w])
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
        t = np.zeros(8, np.float32)
        _chk(self.L.orbx_last_timings(self.h, _p(t)), "timings")
        return dict(zip(["pyramid", "fast", "quadtree", "slots", "blur", "orient_desc", "total", "pyramid_fast_span"], t.tolist()))

    def mean_timings(self):
        t = np.zeros(8, np.float32); n = C.c_int32()
        _chk(self.L.orbx_mean_timings(self.h, _p(t), C.byref(n)), "mean_timings")
        return dict(zip(["pyramid", "fast", "quadtree", "slots", "blur", "orient_desc", "total", "pyramid_fast_span"], t.tolist())), n.value

    def blur_in_pass(self):
        """True when the blur is scheduled inside the pyramid+FAST pass (include/orbx.h: orbx_blur_in_pass)."""
        return bool(self.L.orbx_blur_in_pass(self.h))

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


# ---------------------------------------------------------------------------------------------------------------
# Flattened Frame / KeyFrame views for the searches (include/orbm.h).  FrameView mirrors the members of
# ORB_SLAM3::Frame the matcher reads: mvKeysUn, mDescriptors, mvuRight, mGrid + bounds (include/Frame.h).
# ---------------------------------------------------------------------------------------------------------------
GRID_COLS, GRID_ROWS = 64, 48


class _CFrame(C.Structure):
    _fields_ = [("n", C.c_int32), ("kps", C.c_void_p), ("desc", C.c_void_p), ("uright", C.c_void_p),
                ("min_x", C.c_float), ("min_y", C.c_float), ("inv_w", C.c_float), ("inv_h", C.c_float),
                ("grid_start", C.c_void_p), ("grid_idx", C.c_void_p)]


PAIR_GATE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)      # orbm_pair_gate_fn (include/orbm.h)


def _bind_search(L, prefix):
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    g = lambda n: getattr(L, prefix + n)
    h = [vp] if prefix == "orbm_" else []
    g("grid_build").argtypes = h + [vp, ci, cf, cf, cf, cf, vp, vp]
    g("search_by_projection_frame").argtypes = h + [vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, cf, ci, ci, cf, ci, vp]
    g("search_by_projection_points").argtypes = h + [vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, cf, cf, vp]
    g("search_for_initialization").argtypes = h + [vp, vp, vp, ci, cf, ci, vp]
    g("search_for_triangulation").argtypes = h + [ci, vp, vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, vp, ci, vp, vp, vp,
                                                  vp, cf, cf, vp, vp, ci, ci, ci, vp]
    g("search_by_bow").argtypes = h + [ci, vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, ci, vp, vp, vp, cf, ci, vp]
    g("search_for_triangulation_legacy").argtypes = g("search_for_triangulation").argtypes
    g("search_for_triangulation_gated").argtypes = h + [ci, vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, PAIR_GATE, vp, ci, vp]
    g("search_by_projection_kf").argtypes = h + [vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, cf, ci, ci, vp]
    g("search_by_projection_sim3").argtypes = h + [vp, vp, vp, ci, vp, vp, vp, vp, vp, ci, cf, vp]
    g("search_by_projection_frame_fisheye").argtypes = h + [vp, vp, vp, vp, vp, ci] + [vp] * 9 + [cf, ci, ci, ci, vp, vp]
    g("search_by_projection_points_fisheye").argtypes = h + [vp] * 7 + [ci] + [vp] * 12 + [cf, cf, vp, vp]
    g("search_by_bow_fisheye").argtypes = h + [ci, vp, vp, vp, ci, vp, vp, vp, ci, ci, vp, vp, ci, vp, vp, vp, cf, ci, vp]
    g("search_by_sim3").argtypes = h + [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, cf, vp]
    g("fuse").argtypes = h + [vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, cf, ci, vp]
    g("search_by_bow_kf").argtypes = h + [ci, vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, cf, ci, vp]


class FrameView:
    """Plain-array view of a Frame; `backend` is an object exposing grid_build (ORBmatcher here, the oracle in tests)."""

    def __init__(self, kps, desc, width, height, uright=None, backend=None):
        self.kps = np.ascontiguousarray(kps); self.desc = np.ascontiguousarray(desc, np.uint8)
        self.n = len(self.kps)
        self.uright = None if uright is None else np.ascontiguousarray(uright, np.float32)
        # undistortion-free bounds (Frame::ComputeImageBounds with k1 == 0, Frame.cc:1005-1021)
        self.min_x, self.min_y = np.float32(0), np.float32(0)
        self.inv_w = np.float32(GRID_COLS) / (np.float32(width) - self.min_x)
        self.inv_h = np.float32(GRID_ROWS) / (np.float32(height) - self.min_y)
        self.grid_start = np.zeros(GRID_COLS * GRID_ROWS + 1, np.int32)
        self.grid_idx = np.zeros(max(self.n, 1), np.int32)
        if backend is not None:
            backend.grid_build(self)

    def cstruct(self):
        return _CFrame(self.n, self.kps.ctypes.data, self.desc.ctypes.data,
                       None if self.uright is None else self.uright.ctypes.data,
                       float(self.min_x), float(self.min_y), float(self.inv_w), float(self.inv_h),
                       self.grid_start.ctypes.data, self.grid_idx.ctypes.data)


class ResidentFrame:
    """A Frame kept in HBM across searches (include/orbm.h: orbm_frame_create).  `view` is a FrameView (host arrays are
    uploaded once) or a dict of device pointers (kps, desc, n -- e.g. one frame of an extractor's result block, adopted in place)."""

    def __init__(self, matcher, view=None, device=None, width=752, height=480, uright=None):
        L = lib()
        L.orbm_frame_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_void_p)]
        L.orbm_frame_destroy.argtypes = [C.c_void_p]
        L.orbm_frame_size.argtypes = [C.c_void_p]
        rf = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 8
        L.orbm_search_by_projection_frame_resident.argtypes = rf + [C.c_float, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.orbm_search_by_projection_points_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_float, C.c_float, C.c_void_p]
        self.L, self.m = L, matcher
        h = C.c_void_p()
        if view is not None:
            self.n = view.n
            self._keep = view
            _chk(L.orbm_frame_create(matcher.h, HOST, view.n, _p(view.kps), _p(view.desc), None if view.uright is None else _p(view.uright),
                                     float(view.min_x), float(view.min_y), float(view.inv_w), float(view.inv_h), C.byref(h)), "orbm_frame_create")
        else:
            self.n = int(device["n"])
            inv_w = np.float32(GRID_COLS) / np.float32(width); inv_h = np.float32(GRID_ROWS) / np.float32(height)
            _chk(L.orbm_frame_create(matcher.h, DEVICE, self.n, device["kps"], device["desc"], uright, 0.0, 0.0, float(inv_w), float(inv_h), C.byref(h)),
                 "orbm_frame_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.orbm_frame_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def feature_vector_csr(node_of_feature):
    """DBoW2::FeatureVector (map<NodeId, vector<unsigned>>) as CSR: nodes ascending, features ascending inside."""
    node_of_feature = np.asarray(node_of_feature, np.int64)
    order = np.argsort(node_of_feature, kind="stable").astype(np.int32)
    nodes, counts = np.unique(node_of_feature, return_counts=True)
    start = np.zeros(len(nodes) + 1, np.int32); start[1:] = np.cumsum(counts)
    return nodes.astype(np.int32), start, order


class _SearchMixin:
    """Search entry points shared by the product matcher and (in tests) the oracle wrapper: same argument lists."""
    _prefix = "orbm_"

    def _call(self, name, *args):
        f = getattr(self.L, self._prefix + name)
        a = ([self.h] if self._prefix == "orbm_" else []) + list(args)
        rc = f(*a)
        if rc < 0 and self._prefix == "orbm_":
            _chk(rc, name)
        return rc

    def grid_build(self, fv):
        fv.placed = self._call("grid_build", _p(fv.kps), fv.n, float(fv.min_x), float(fv.min_y), float(fv.inv_w), float(fv.inv_h),
                                _p(fv.grid_start), _p(fv.grid_idx))
        return fv.placed

    def SearchByProjectionFrame(self, cur, cur_blocked, scale_factors, valid, u, v, invzc, octave, angle, qdesc, mp_obs,
                                th, forward=False, backward=False, mbf=0.0, check_ori=True):
        match = np.full(cur.n, -1, np.int32)
        cs = cur.cstruct()
        arr = [np.ascontiguousarray(a, t) for a, t in ((cur_blocked, np.uint8), (scale_factors, np.float32), (valid, np.uint8),
               (u, np.float32), (v, np.float32), (invzc, np.float32), (octave, np.int32), (angle, np.float32),
               (qdesc, np.uint8), (mp_obs, np.uint8))]
        n = self._call("search_by_projection_frame", C.byref(cs), _p(arr[0]), _p(arr[1]), len(arr[2]), *[_p(a) for a in arr[2:]],
                       float(th), int(forward), int(backward), float(mbf), int(check_ori), _p(match))
        return n, match

    def SearchByProjectionFrameResident(self, cur, cur_blocked, scale_factors, valid, u, v, invzc, octave, angle, qdesc, mp_obs,
                                        th, forward=False, backward=False, mbf=0.0, check_ori=True):
        """M4 against a ResidentFrame: only the queries travel (include/orbm.h: orbm_search_by_projection_frame_resident)."""
        match = np.full(cur.n, -1, np.int32)
        arr = [np.ascontiguousarray(a, t) for a, t in ((cur_blocked, np.uint8), (scale_factors, np.float32), (valid, np.uint8),
               (u, np.float32), (v, np.float32), (invzc, np.float32), (octave, np.int32), (angle, np.float32),
               (qdesc, np.uint8), (mp_obs, np.uint8))]
        n = self._call("search_by_projection_frame_resident", cur.h, _p(arr[0]), _p(arr[1]), len(arr[2]), *[_p(a) for a in arr[2:]],
                       float(th), int(forward), int(backward), float(mbf), int(check_ori), _p(match))
        return n, match

    def SearchByProjectionPointsResident(self, f, blocked, scale_factors, in_view, px, py, pxr, view_cos, level, qdesc, mp_obs, th, nnratio):
        match = np.full(f.n, -1, np.int32)
        arr = [np.ascontiguousarray(a, t) for a, t in ((blocked, np.uint8), (scale_factors, np.float32), (in_view, np.uint8),
               (px, np.float32), (py, np.float32), (pxr, np.float32), (view_cos, np.float32), (level, np.int32),
               (qdesc, np.uint8), (mp_obs, np.uint8))]
        n = self._call("search_by_projection_points_resident", f.h, _p(arr[0]), _p(arr[1]), len(arr[2]), *[_p(a) for a in arr[2:]],
                       float(th), float(nnratio), _p(match))
        return n, match

    def SearchByProjectionPoints(self, f, blocked, scale_factors, in_view, px, py, pxr, view_cos, level, qdesc, mp_obs, th, nnratio):
        match = np.full(f.n, -1, np.int32)
        cs = f.cstruct()
        arr = [np.ascontiguousarray(a, t) for a, t in ((blocked, np.uint8), (scale_factors, np.float32), (in_view, np.uint8),
               (px, np.float32), (py, np.float32), (pxr, np.float32), (view_cos, np.float32), (level, np.int32),
               (qdesc, np.uint8), (mp_obs, np.uint8))]
        n = self._call("search_by_projection_points", C.byref(cs), _p(arr[0]), _p(arr[1]), len(arr[2]), *[_p(a) for a in arr[2:]],
                       float(th), float(nnratio), _p(match))
        return n, match

    def SearchForInitialization(self, f1, f2, prev_matched, window, nnratio, check_ori=True):
        prev = np.ascontiguousarray(prev_matched, np.float32).copy()
        m12 = np.full(f1.n, -1, np.int32)
        c1, c2 = f1.cstruct(), f2.cstruct()
        n = self._call("search_for_initialization", C.byref(c1), C.byref(c2), _p(prev), int(window), float(nnratio), int(check_ori), _p(m12))
        return n, m12, prev

    def SearchByProjectionKF(self, cur, blocked, scale_factors, valid, u, v, level, angle, qdesc, th, orb_dist, check_ori=True):
        match = np.full(cur.n, -1, np.int32)
        cs = cur.cstruct()
        arr = [np.ascontiguousarray(a, t) for a, t in ((blocked, np.uint8), (scale_factors, np.float32), (valid, np.uint8),
               (u, np.float32), (v, np.float32), (level, np.int32), (angle, np.float32), (qdesc, np.uint8))]
        n = self._call("search_by_projection_kf", C.byref(cs), _p(arr[0]), _p(arr[1]), len(arr[2]), *[_p(a) for a in arr[2:]],
                       float(th), int(orb_dist), int(check_ori), _p(match))
        return n, match

    def SearchByProjectionSim3(self, kf, matched_in, scale_factors, valid, u, v, level, qdesc, th, ratio_hamming):
        match = np.full(kf.n, -1, np.int32)
        cs = kf.cstruct()
        arr = [np.ascontiguousarray(a, t) for a, t in ((matched_in, np.uint8), (scale_factors, np.float32), (valid, np.uint8),
               (u, np.float32), (v, np.float32), (level, np.int32), (qdesc, np.uint8))]
        n = self._call("search_by_projection_sim3", C.byref(cs), _p(arr[0]), _p(arr[1]), len(arr[2]), *[_p(a) for a in arr[2:]],
                       int(th), float(ratio_hamming), _p(match))
        return n, match

    def SearchByProjectionFrameFisheye(self, cur_l, cur_r, blocked_l, blocked_r, scale_factors, valid, u, v, ur, vr, octave, angle,
                                       qdesc, mp_obs, th, forward=False, backward=False, check_ori=True):
        ml = np.full(cur_l.n, -1, np.int32); mr = np.full(cur_r.n, -1, np.int32)
        cl, cr = cur_l.cstruct(), cur_r.cstruct()
        arr = [np.ascontiguousarray(a, t) for a, t in ((blocked_l, np.uint8), (blocked_r, np.uint8), (scale_factors, np.float32),
               (valid, np.uint8), (u, np.float32), (v, np.float32), (ur, np.float32), (vr, np.float32), (octave, np.int32),
               (angle, np.float32), (qdesc, np.uint8), (mp_obs, np.uint8))]
        n = self._call("search_by_projection_frame_fisheye", C.byref(cl), C.byref(cr), _p(arr[0]), _p(arr[1]), _p(arr[2]), len(arr[3]),
                       *[_p(a) for a in arr[3:]], float(th), int(forward), int(backward), int(check_ori), _p(ml), _p(mr))
        return n, ml, mr

    def SearchByProjectionPointsFisheye(self, f_l, f_r, blocked_l, blocked_r, l2r, r2l, scale_factors, left, right, qdesc, mp_obs, th, nnratio):
        """left / right: dicts with in_view, px, py, view_cos, level for the two cameras."""
        ml = np.full(f_l.n, -1, np.int32); mr = np.full(f_r.n, -1, np.int32)
        cl, cr = f_l.cstruct(), f_r.cstruct()
        def pack(q):
            return [np.ascontiguousarray(q["in_view"], np.uint8), np.ascontiguousarray(q["px"], np.float32), np.ascontiguousarray(q["py"], np.float32),
                    np.ascontiguousarray(q["view_cos"], np.float32), np.ascontiguousarray(q["level"], np.int32)]
        a = [np.ascontiguousarray(blocked_l, np.uint8), np.ascontiguousarray(blocked_r, np.uint8), np.ascontiguousarray(l2r, np.int32),
             np.ascontiguousarray(r2l, np.int32), np.ascontiguousarray(scale_factors, np.float32)]
        L_, R_ = pack(left), pack(right)
        qd = np.ascontiguousarray(qdesc, np.uint8); ob = np.ascontiguousarray(mp_obs, np.uint8)
        n = self._call("search_by_projection_points_fisheye", C.byref(cl), C.byref(cr), *[_p(x) for x in a], len(L_[0]),
                       *[_p(x) for x in L_], *[_p(x) for x in R_], _p(qd), _p(ob), float(th), float(nnratio), _p(ml), _p(mr))
        return n, ml, mr

    def SearchByBoWFisheye(self, kkf, dkf, kf_good, fvk, kf_, df, nleft, fvf, nnratio, check_ori=True):
        fm = np.full(len(kf_), -1, np.int32)
        kkf = np.ascontiguousarray(kkf); kf_ = np.ascontiguousarray(kf_)
        dkf = np.ascontiguousarray(dkf, np.uint8); df = np.ascontiguousarray(df, np.uint8)
        good = np.ascontiguousarray(kf_good, np.uint8)
        n = self._call("search_by_bow_fisheye", len(kkf), _p(kkf), _p(dkf), _p(good), len(fvk[0]), _p(fvk[0]), _p(fvk[1]), _p(fvk[2]),
                       len(kf_), int(nleft), _p(kf_), _p(df), len(fvf[0]), _p(fvf[0]), _p(fvf[1]), _p(fvf[2]),
                       float(nnratio), int(check_ori), _p(fm))
        return n, fm

    def SearchBySim3(self, kf1, kf2, sf1, sf2, q1, q2, th):
        """q1 / q2: dicts with valid, u, v, level, qdesc for the KF1->KF2 and KF2->KF1 projections."""
        m12 = np.full(kf1.n, -1, np.int32)
        c1, c2 = kf1.cstruct(), kf2.cstruct()
        s1 = np.ascontiguousarray(sf1, np.float32); s2 = np.ascontiguousarray(sf2, np.float32)
        def pack(q):
            return [np.ascontiguousarray(q["valid"], np.uint8), np.ascontiguousarray(q["u"], np.float32), np.ascontiguousarray(q["v"], np.float32),
                    np.ascontiguousarray(q["level"], np.int32), np.ascontiguousarray(q["qdesc"], np.uint8)]
        a1, a2 = pack(q1), pack(q2)
        n = self._call("search_by_sim3", C.byref(c1), C.byref(c2), _p(s1), _p(s2), *[_p(a) for a in a1], *[_p(a) for a in a2], float(th), _p(m12))
        return n, m12

    def Fuse(self, kf, scale_factors, inv_sigma2, valid, u, v, ur, level, qdesc, th, chi2_gate=True):
        best = np.full(len(valid), -1, np.int32)
        cs = kf.cstruct()
        arr = [np.ascontiguousarray(a, t) for a, t in ((scale_factors, np.float32), (inv_sigma2, np.float32), (valid, np.uint8),
               (u, np.float32), (v, np.float32), (ur, np.float32), (level, np.int32), (qdesc, np.uint8))]
        n = self._call("fuse", C.byref(cs), _p(arr[0]), _p(arr[1]), len(arr[2]), *[_p(a) for a in arr[2:]],
                       float(th), int(chi2_gate), _p(best))
        return n, best

    def SearchByBoWKF(self, k1, d1, good1, fv1, k2, d2, good2, fv2, nnratio, check_ori=True):
        m12 = np.full(len(k1), -1, np.int32)
        k1 = np.ascontiguousarray(k1); k2 = np.ascontiguousarray(k2)
        d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
        g1 = np.ascontiguousarray(good1, np.uint8); g2 = np.ascontiguousarray(good2, np.uint8)
        n = self._call("search_by_bow_kf", len(k1), _p(k1), _p(d1), _p(g1), len(fv1[0]), _p(fv1[0]), _p(fv1[1]), _p(fv1[2]),
                       len(k2), _p(k2), _p(d2), _p(g2), len(fv2[0]), _p(fv2[0]), _p(fv2[1]), _p(fv2[2]),
                       float(nnratio), int(check_ori), _p(m12))
        return n, m12

    def SearchForTriangulation(self, k1, d1, has_mp1, ur1, fv1, k2, d2, has_mp2, ur2, fv2, F12, ep, sf2, sigma2_2,
                               only_stereo=False, coarse=False, check_ori=False, legacy=False):
        m12 = np.full(len(k1), -1, np.int32)
        k1 = np.ascontiguousarray(k1); k2 = np.ascontiguousarray(k2)
        d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
        h1 = np.ascontiguousarray(has_mp1, np.uint8); h2 = np.ascontiguousarray(has_mp2, np.uint8)
        u1 = None if ur1 is None else np.ascontiguousarray(ur1, np.float32)
        u2 = None if ur2 is None else np.ascontiguousarray(ur2, np.float32)
        F = np.ascontiguousarray(F12, np.float32).reshape(9); sf2 = np.ascontiguousarray(sf2, np.float32)
        sg = np.ascontiguousarray(sigma2_2, np.float32)
        n = self._call("search_for_triangulation_legacy" if legacy else "search_for_triangulation", len(k1), _p(k1), _p(d1), _p(h1), None if u1 is None else _p(u1),
                       len(fv1[0]), _p(fv1[0]), _p(fv1[1]), _p(fv1[2]),
                       len(k2), _p(k2), _p(d2), _p(h2), None if u2 is None else _p(u2),
                       len(fv2[0]), _p(fv2[0]), _p(fv2[1]), _p(fv2[2]),
                       _p(F), float(ep[0]), float(ep[1]), _p(sf2), _p(sg), int(only_stereo), int(coarse), int(check_ori), _p(m12))
        return n, m12

    def SearchForTriangulationGated(self, k1, d1, has_mp1, fv1, k2, d2, has_mp2, fv2, gate, check_ori=False):
        """M10 with a second camera / M12 (ORBmatcher.cc:1388-1629, 1632-1821): `gate(idx1, idx2) -> bool` stands for the
        camera model's epipolarConstrain_ / matchAndtriangulate and is called where the reference calls it."""
        m12 = np.full(len(k1), -1, np.int32)
        k1 = np.ascontiguousarray(k1); k2 = np.ascontiguousarray(k2)
        d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
        h1 = np.ascontiguousarray(has_mp1, np.uint8); h2 = np.ascontiguousarray(has_mp2, np.uint8)
        cb = PAIR_GATE(lambda user, a, b: int(bool(gate(a, b))))
        n = self._call("search_for_triangulation_gated", len(k1), _p(k1), _p(d1), _p(h1), len(fv1[0]), _p(fv1[0]), _p(fv1[1]), _p(fv1[2]),
                       len(k2), _p(k2), _p(d2), _p(h2), len(fv2[0]), _p(fv2[0]), _p(fv2[1]), _p(fv2[2]), cb, None, int(check_ori), _p(m12))
        return n, m12

    def SearchByBoW(self, kkf, dkf, kf_good, fvk, kf_, df, fvf, nnratio, check_ori=True):
        fm = np.full(len(kf_), -1, np.int32)
        kkf = np.ascontiguousarray(kkf); kf_ = np.ascontiguousarray(kf_)
        dkf = np.ascontiguousarray(dkf, np.uint8); df = np.ascontiguousarray(df, np.uint8)
        good = np.ascontiguousarray(kf_good, np.uint8)
        n = self._call("search_by_bow", len(kkf), _p(kkf), _p(dkf), _p(good), len(fvk[0]), _p(fvk[0]), _p(fvk[1]), _p(fvk[2]),
                       len(kf_), _p(kf_), _p(df), len(fvf[0]), _p(fvf[0]), _p(fvf[1]), _p(fvf[2]),
                       float(nnratio), int(check_ori), _p(fm))
        return n, fm


def _install_search():
    L = lib()
    _bind_search(L, "orbm_")
    L.orbm_window_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_int] + [C.c_void_p] * 3
    L.orbm_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    for name in ("grid_build", "SearchByProjectionFrame", "SearchByProjectionPoints", "SearchForInitialization",
                 "SearchForTriangulation", "SearchForTriangulationGated", "SearchByBoW", "SearchByProjectionKF", "SearchByBoWKF", "SearchByProjectionSim3", "Fuse", "SearchBySim3", "SearchByProjectionFrameFisheye",
                 "SearchByProjectionPointsFisheye", "SearchByBoWFisheye", "SearchByProjectionFrameResident", "SearchByProjectionPointsResident", "_call"):
        setattr(ORBmatcher, name, getattr(_SearchMixin, name))
    ORBmatcher._prefix = "orbm_"

    def window_candidates(self, f, qx, qy, qr, min_level, max_level, qdesc, cap, q_ur=None, q_er=None):
        nq = len(qx)
        arr = [np.ascontiguousarray(a, t) for a, t in ((qx, np.float32), (qy, np.float32), (qr, np.float32),
               (min_level, np.int32), (max_level, np.int32))]
        ur = None if q_ur is None else np.ascontiguousarray(q_ur, np.float32)
        er = None if q_er is None else np.ascontiguousarray(q_er, np.float32)
        qd = np.ascontiguousarray(qdesc, np.uint8)
        cnt = np.zeros(nq, np.int32); idx = np.zeros((nq, cap), np.int32); dist = np.zeros((nq, cap), np.int32)
        cs = f.cstruct()
        _chk(self.L.orbm_window_candidates(self.h, C.byref(cs), nq, *[_p(a) for a in arr], None if ur is None else _p(ur),
                                           None if er is None else _p(er), _p(qd), cap, _p(cnt), _p(idx), _p(dist)), "window_candidates")
        return cnt, idx, dist

    def stereo_matches(self, ex_left, ex_right, kl, dl, kr, dr, mb, mbf, frame_l=0, frame_r=0):
        kl = np.ascontiguousarray(kl); kr = np.ascontiguousarray(kr)
        dl = np.ascontiguousarray(dl, np.uint8); dr = np.ascontiguousarray(dr, np.uint8)
        ur = np.zeros(max(len(kl), 1), np.float32); dp = np.zeros(max(len(kl), 1), np.float32)
        n = _chk(self.L.orbm_stereo_matches(self.h, ex_left.h, frame_l, ex_right.h, frame_r, len(kl), _p(kl), _p(dl), len(kr), _p(kr), _p(dr),
                                            float(mb), float(mbf), _p(ur), _p(dp)), "stereo_matches")
        return n, ur[:len(kl)], dp[:len(kl)]

    L.orbm_grid_build_batch_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                              C.c_float, C.c_void_p, C.c_void_p]
    L.orbm_track_window_batch_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_float,
                                                C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orbm_search_by_projection_batch_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                        C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_float,
                                                        C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int,
                                                        C.c_void_p, C.c_void_p]
    L.orbm_vocab_load_text.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_char_p]
    L.orbm_vocab_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orbm_vocab_destroy.argtypes = [C.c_void_p]
    L.orbm_vocab_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4
    L.orbm_bow_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orbm_bow_vectors.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int),
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    ORBmatcher.window_candidates = window_candidates
    ORBmatcher.ComputeStereoMatches = stereo_matches
    _bind_frame_geometry(L, "orbm_")
    for name, fn in _frame_geometry_methods("orbm_").items():
        setattr(ORBmatcher, name, fn)


def _bind_frame_geometry(L, prefix):
    """argtypes of the undistort / image-bounds / frustum entry points (same argument lists on the product and the oracle,
    except the leading handle + space of the product)."""
    vp, f, i = C.c_void_p, C.c_float, C.c_int
    lead = [vp, i] if prefix == "orbm_" else []
    getattr(L, prefix + "undistort_keypoints").argtypes = lead + [vp, i, vp, vp, i, vp, vp]
    getattr(L, prefix + "image_bounds").argtypes = ([vp] if prefix == "orbm_" else []) + [i, i, vp, vp, i, vp, vp]
    getattr(L, prefix + "is_in_frustum").argtypes = lead + [i, vp, vp, vp, vp, vp, vp, vp, vp, vp, f, f, f, i, vp, vp, vp, vp, vp, vp, vp]


def _frame_geometry_methods(prefix):
    def _lead(self, with_space=True):
        return ([self.h, HOST] if with_space else [self.h]) if prefix == "orbm_" else []

    def UndistortKeyPoints(self, kps, K, dist, newK=None):
        """Frame::UndistortKeyPoints (Frame.cc:924-970).  K / newK = (fx, fy, cx, cy); dist = (k1, k2, p1, p2[, k3])."""
        kps = np.ascontiguousarray(kps, KP_DTYPE); out = np.zeros(max(len(kps), 1), KP_DTYPE)
        K = np.ascontiguousarray(K, np.float32); nk = K if newK is None else np.ascontiguousarray(newK, np.float32)
        d = np.ascontiguousarray(dist, np.float32)
        rc = getattr(self.L, prefix + "undistort_keypoints")(*_lead(self), _p(kps), len(kps), _p(K), _p(d), len(d), _p(nk), _p(out))
        if rc < 0:
            raise OrbError("undistort_keypoints failed with code %d" % rc)
        return out[:len(kps)]

    def ComputeImageBounds(self, cols, rows, K, dist, newK=None):
        """Frame::ComputeImageBounds (Frame.cc:977-1021) -> (minX, maxX, minY, maxY)."""
        K = np.ascontiguousarray(K, np.float32); nk = K if newK is None else np.ascontiguousarray(newK, np.float32)
        d = np.ascontiguousarray(dist, np.float32); b = np.zeros(4, np.float32)
        rc = getattr(self.L, prefix + "image_bounds")(*_lead(self, False), int(cols), int(rows), _p(K), _p(d), len(d), _p(nk), _p(b))
        if rc < 0:
            raise OrbError("image_bounds failed with code %d" % rc)
        return b

    def isInFrustum(self, Pw, normal, min_dist, max_dist, Rcw, tcw, Ow, K, bounds, bf, viewing_cos_limit, log_scale_factor, n_levels):
        """Frame::isInFrustum for n map points (Frame.cc:603-671) -> dict of the MapPoint tracking members."""
        Pw = np.ascontiguousarray(Pw, np.float32).reshape(-1, 3); n = len(Pw)
        nm = np.ascontiguousarray(normal, np.float32).reshape(-1, 3)
        mn = np.ascontiguousarray(min_dist, np.float32); mx = np.ascontiguousarray(max_dist, np.float32)
        R = np.ascontiguousarray(Rcw, np.float32).reshape(9); t = np.ascontiguousarray(tcw, np.float32).reshape(3)
        O = np.ascontiguousarray(Ow, np.float32).reshape(3); K = np.ascontiguousarray(K, np.float32); B = np.ascontiguousarray(bounds, np.float32)
        out = {"in_view": np.zeros(max(n, 1), np.uint8), "proj_x": np.zeros(max(n, 1), np.float32), "proj_y": np.zeros(max(n, 1), np.float32),
               "proj_xr": np.zeros(max(n, 1), np.float32), "depth": np.zeros(max(n, 1), np.float32), "level": np.full(max(n, 1), -1, np.int32),
               "view_cos": np.zeros(max(n, 1), np.float32)}
        rc = getattr(self.L, prefix + "is_in_frustum")(*_lead(self), n, _p(Pw), _p(nm), _p(mn), _p(mx), _p(R), _p(t), _p(O), _p(K), _p(B),
                                                       float(bf), float(viewing_cos_limit), float(log_scale_factor), int(n_levels),
                                                       _p(out["in_view"]), _p(out["proj_x"]), _p(out["proj_y"]), _p(out["proj_xr"]),
                                                       _p(out["depth"]), _p(out["level"]), _p(out["view_cos"]))
        if rc < 0:
            raise OrbError("is_in_frustum failed with code %d" % rc)
        return rc, {k: v[:n] for k, v in out.items()}

    return {"UndistortKeyPoints": UndistortKeyPoints, "ComputeImageBounds": ComputeImageBounds, "isInFrustum": isInFrustum}


EXPORTS += ["orbm_undistort_keypoints", "orbm_image_bounds", "orbm_is_in_frustum"]
EXPORTS += ["orbm_frame_create", "orbm_frame_destroy", "orbm_frame_size", "orbm_search_by_projection_frame_resident", "orbm_search_by_projection_points_resident"]
EXPORTS += ["orbm_grid_build", "orbm_window_candidates", "orbm_search_by_projection_frame", "orbm_search_by_projection_points",
            "orbm_search_for_initialization", "orbm_search_for_triangulation", "orbm_search_by_bow", "orbm_stereo_matches",
            "orbm_search_by_projection_kf", "orbm_search_by_bow_kf", "orbm_search_for_triangulation_legacy", "orbm_search_for_triangulation_gated",
            "orbm_search_by_projection_sim3", "orbm_fuse", "orbm_search_by_sim3",
            "orbm_grid_build_batch_async", "orbm_track_window_batch_async", "orbm_search_by_projection_batch_async", "orbm_search_by_projection_frame_fisheye",
            "orbm_search_by_projection_points_fisheye", "orbm_search_by_bow_fisheye",
            "orbm_vocab_load_text", "orbm_vocab_create", "orbm_vocab_destroy", "orbm_vocab_info", "orbm_bow_transform", "orbm_bow_vectors"]
_orig_lib = lib
_search_ready = False


def lib():                                         # noqa: F811  (binds the search entry points on first use)
    global _search_ready
    L = _orig_lib()
    if not _search_ready:
        _search_ready = True
        _install_search()
    return L


class ORBVocabulary:
    """DBoW2 ORB vocabulary resident in HBM (include/ORBVocabulary.h typedef; TemplatedVocabulary.h)."""

    def __init__(self, matcher, path):
        self.L = lib(); self.m = matcher
        h = C.c_void_p()
        if isinstance(path, dict):                                  # arrays: k, L, parent, is_leaf, desc, weight (orbm_vocab_create)
            a = path
            arr = [np.ascontiguousarray(a["parent"], np.int32), np.ascontiguousarray(a["is_leaf"], np.uint8),
                   np.ascontiguousarray(a["desc"], np.uint8), np.ascontiguousarray(a["weight"], np.float64)]
            _chk(self.L.orbm_vocab_create(matcher.h, C.byref(h), int(a["k"]), int(a["L"]), len(arr[0]), *[_p(x) for x in arr]), "orbm_vocab_create")
        else:
            _chk(self.L.orbm_vocab_load_text(matcher.h, C.byref(h), path.encode()), "orbm_vocab_load_text")
        self.h = h

    def info(self):
        v = [C.c_int() for _ in range(4)]
        self.L.orbm_vocab_info(self.h, *[C.byref(x) for x in v])
        return dict(zip(["k", "L", "nnodes", "nwords"], [x.value for x in v]))

    def transform(self, desc, levelsup=4):
        """Frame::ComputeBoW: returns (BowVector ids, values), FeatureVector CSR (nodes, start, idx), per-feature arrays."""
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32); n = desc.shape[0]
        w = np.zeros(max(n, 1), np.int32); nd = np.zeros(max(n, 1), np.int32); wt = np.zeros(max(n, 1), np.float64)
        _chk(self.L.orbm_bow_transform(self.m.h, self.h, _p(desc), n, levelsup, _p(w), _p(nd), _p(wt)), "orbm_bow_transform")
        return bow_vectors(self.L.orbm_bow_vectors, n, w[:n], nd[:n], wt[:n]) + (w[:n], nd[:n], wt[:n])

    def __del__(self):
        try:
            self.L.orbm_vocab_destroy(self.h)
        except Exception:
            pass


def bow_vectors(fn, n, w, nd, wt):
    w = np.ascontiguousarray(w, np.int32); nd = np.ascontiguousarray(nd, np.int32); wt = np.ascontiguousarray(wt, np.float64)
    bi = np.zeros(max(n, 1), np.int32); bv = np.zeros(max(n, 1), np.float64); nb = C.c_int()
    fn_ = np.zeros(max(n, 1), np.int32); fs = np.zeros(max(n, 1) + 1, np.int32); fi = np.zeros(max(n, 1), np.int32); nf = C.c_int()
    fn(n, _p(w), _p(nd), _p(wt), _p(bi), _p(bv), C.byref(nb), _p(fn_), _p(fs), _p(fi), C.byref(nf))
    return (bi[:nb.value], bv[:nb.value]), (fn_[:nf.value], fs[:nf.value + 1], fi[:fs[nf.value]])
