#!/usr/bin/env python3
"""bench.py -- ORB extract+match frames/sec (BASELINE.json metric) on N MI355X.

One step = one pass of the hot path over one batch of synthetic frames already resident in HBM.  Workloads (--config):
  c2 (default, the headline: configs[1])  752x480, 1000 features, 512 frames/GPU/step: extraction of every frame (8-level
      pyramid, FAST, quadtree, orientation, blur, rBRIEF) + dense 2-NN Hamming match of each frame against the previous one.
  c5  1920x1080, 4000 features, 64 frames/GPU/step, same step (the roofline run of SURVEY 8(d)).
  c4  TUM-VI geometry: 8 fisheye stereo pairs of 512x512, 1500 features per image: 2 extractions + the brute-force 2-NN of
      ComputeStereoFishEyeMatches (Frame.cc:1440-1480) per pair.
  c3  EuRoC stereo: pairs of 752x480, 1200 features: 2 extractions + ComputeStereoMatches (Frame.cc:1027-1276) per pair + one
      SearchForTriangulation bucket match against the previous pair (LocalMapping.cc:592).
`value` counts what SURVEY 8(d) specifies: all kernels + the copy of every frame's keypoints / descriptors / counts to (pinned)
host memory, which runs on a copy stream beside the next batch; `value_device_resident` is the same loop without that copy.

N GPUs = N processes (torch.distributed over RCCL), each with its own frames ("weak" scaling); the only collective is the
all_gather of per-frame keypoint counts.  `python bench.py --gpus N` starts the N ranks itself (fresh child processes, spawned
before this process touches torch or HIP); under torchrun (RANK / WORLD_SIZE set) it is one of the ranks.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline"     -- pyramid+FAST(+blur) pass: algorithmic bytes / live HIP-event span of those launches
  "cpu_baseline" -- the CPU oracle (a port, 1 thread) on a bounded sample of the same frames (+ an nproc-thread context figure).
"""
import argparse
import ctypes as C
import gc
import importlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X spec (MI355X_MICROARCH.md): 8.0 TB/s; 6.29 TB/s measured copy peak

CONFIGS = {
    #        W     H     nF    frames/step  lapping       what
    # frames/step is this build's choice (BASELINE.json fixes it only for configs[3]: 8 pairs): big enough that the ~85 us of kernel
    # boundaries and graph-launch latency a step pays once are a few per cent of it (752x480: 512 frames 273 k frames/s, 1024 286 k,
    # 2048 no further gain); 1024 frames hold 0.4 GB of images, 2.4 GB of pyramids and 8 x 62 MB of result blocks
    "c2": (752, 480, 1000, 1024, (0, 1000), "configs[1]"),
    "c5": (1920, 1080, 4000, 128, (0, 1000), "configs[4]"),
    "c4": (512, 512, 1500, 16, (0, 511), "configs[3]"),
    "c3": (752, 480, 1200, 512, (0, 0), "configs[2]"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--batch", type=int, default=0, help="frames per GPU per step (0 = the config's default)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--nfeatures", type=int, default=0)
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic frames generated per rank")
    ap.add_argument("--launch", choices=["auto", "graph", "eager"], default="auto",
                    help="graph: the step's whole enqueue sequence is captured once and replayed with one hipGraphLaunch per step; "
                         "auto: graph from 64 frames per step on, eager below (a small step's graph is placed differently from one "
                         "instantiation to the next: 0.24 or 0.32 ms per 16-frame step, eager 0.245 every time)")
    ap.add_argument("--match", choices=["window", "knn2"], default="window",
                    help="c2/c5 match leg: 'window' (default) = the monocular tracker's own matcher, SearchByProjection(frame, previous frame) "
                         "with final matches on the device (Tracking.cc:3203-3211); 'knn2' = dense brute-force 2-NN (Frame.cc:1440-1480)")
    ap.add_argument("--texture", choices=["dense", "sparse", "lowcontrast", "mixed"], default="dense",
                    help="synthetic input class (orb-slam3_amd/synth.py): dense = SURVEY 8(d)'s generator (the headline; ~6 %% of the level-0 "
                         "pixels are FAST corners), sparse = camera-like density (~2 %%), lowcontrast = corners mostly below iniThFAST "
                         "(per-cell threshold retry), mixed = the three classes side by side in every batch")
    ap.add_argument("--cpu-sample", type=int, default=96, help="frames timed through the CPU oracle (0 = skip)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` outside torchrun starts N fresh rank processes (nothing GPU-related has been imported here)
# ----------------------------------------------------------------------------------------------------------------------
def launch_ranks(n, argv):
    import tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    # a rank that dies takes the job down: the others would wait for it in the rendezvous or in a collective
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            for p in procs:
                if p.poll() is None:
                    p.terminate()                       # exactly the children started above
            break
        time.sleep(0.1)
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=30))
        except subprocess.TimeoutExpired:
            p.kill()
            rcs.append(p.wait())
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    if failed or any(rc != 0 for rc in rcs):
        sys.stderr.write("bench.py: ranks failed, exit codes %s\n" % rcs)
        return 1
    return 0


# ----------------------------------------------------------------------------------------------------------------------
# the workloads
# ----------------------------------------------------------------------------------------------------------------------
class StubWorkload:
    """ORB_BENCH_STUB=1: no GPU, no HIP library -- the launcher / rendezvous / max-over-ranks / count-gather path on CPU."""
    kind = "stub"

    def __init__(self, args, rank, local_rank):
        self.B = args.batch or 4
        self.rank = rank
        if os.environ.get("ORB_BENCH_STUB_FAIL_RANK") == str(rank):      # launcher test: a rank that dies before the rendezvous
            raise SystemExit(3)

    def prime(self):
        pass

    def step(self):
        time.sleep(0.02 if os.environ.get("ORB_BENCH_STUB_SLOW_RANK") == str(self.rank) else 0.002)   # (a straggler, for the per-rank times)

    def flush(self):
        pass

    def sync(self):
        pass

    def counts(self):
        import numpy as np
        return np.full(self.B, 1000 + self.rank, np.int32)


class OrbWorkload:
    """Extraction of a batch + the config's match leg, all through the C ABI (ctypes)."""
    kind = "orb"

    def __init__(self, args, rank, local_rank):
        import numpy as np
        self.np = np
        pkg = importlib.import_module("orb-slam3_amd")
        synth = importlib.import_module("orb-slam3_amd.synth")
        self.pkg, self.L = pkg, pkg.lib()
        L = self.L
        cw, ch, cnf, cb, lap, _ = CONFIGS[args.config]
        self.W, self.H = args.width or cw, args.height or ch
        self.nF, self.B = args.nfeatures or cnf, args.batch or cb
        self.cfg, self.args, self.lap = args.config, args, lap
        W, H, B = self.W, self.H, self.B
        if self.cfg in ("c3", "c4") and B % 2:
            raise SystemExit("stereo configs take an even number of frames per step")
        self.ex = pkg.ORBextractor(self.nF, 1.2, 8, 20, 7, device=local_rank, max_size=(W, H), max_batch=B)
        self.mt = pkg.ORBmatcher(0.7, device=local_rank)
        ex, mt = self.ex, self.mt
        self.cap = cap = ex.cap
        # ---- inputs resident in HBM before the timed region
        self.stride = stride = (W + 63) // 64 * 64
        nd = max(1, min(args.distinct, B))
        if self.cfg in ("c3", "c4"):                            # stereo: frames 0..B/2-1 are the left images, B/2.. the right ones
            pairs = [synth.gen_stereo_pair(W, H, 1000 * rank + 100 + i) for i in range(max(1, nd // 2))]
            lefts = [p[0] for p in pairs]; rights = [p[1] for p in pairs]
            self.host_imgs = [lefts[i % len(lefts)] for i in range(B // 2)] + [rights[i % len(rights)] for i in range(B // 2)]
        else:
            kinds = {"dense": ("textured",), "sparse": ("sparse",), "lowcontrast": ("lowcontrast",), "mixed": ("textured", "sparse", "lowcontrast")}[args.texture]
            base = [synth.gen_image(W, H, 1000 * rank + 1 + i, kinds[i % len(kinds)]) for i in range(nd)]
            self.host_imgs = [base[i % nd] for i in range(B)]
        self.dev = pkg.DeviceBuffer(stride * H * B)
        padded = np.zeros((H, stride), np.uint8)
        for i in range(B):
            padded[:, :W] = self.host_imgs[i]
            self.dev.upload(padded, offset=i * stride * H)
        self.ptrs = (C.c_void_p * B)(*[int(self.dev.ptr + i * stride * H) for i in range(B)])
        self.laps = np.tile(np.array(lap, np.int32), B)
        self.nblk = 8
        self.res = []                                           # device pointers of the result blocks (walked round-robin step by step)
        for blk in range(self.nblk):
            L.orbx_set_result_block(ex.h, blk)
            self.res.append(ex.result_device())
        L.orbx_set_result_block(ex.h, 0)
        if args.match == "window":
            self.sf_host = ex.GetScaleFactors()
            self.inv_w = float(np.float32(64) / np.float32(W)); self.inv_h = float(np.float32(48) / np.float32(H))   # Frame.cc:401-402
        # ---- where the results land on the host: pinned, one block per array (reused every step, like a Frame's mvKeys)
        self.layout = ex.result_block_layout()                  # one block = [kps | desc | counts | monos], moved by ONE copy
        self.host_bytes = self.layout[4]
        if self.cfg == "c3":
            self._setup_c3()                                    # attaches the matcher's per-block outputs behind the block
        else:
            self._setup_match_outputs()                         # likewise: the match leg's results travel to the host with the block
        self.h_blk = [pkg.PinnedBuffer(self.host_bytes) for _ in range(self.nblk)]
        self.download = True
        self.graph = args.launch == "graph" or (args.launch == "auto" and B >= 64)
        self.nslots = 8
        self.captured = False
        self.measure_match = False
        self.k = 0
        # the matcher's kernels run on the extractor's stream (batch i's match, then batch i+1's extraction, in order): they are
        # then part of the captured sequence too
        assert L.orbm_set_stream(mt.h, L.orbx_stream(ex.h)) == 0
        L.orbx_set_stage_timing(ex.h, 0)

    def _setup_match_outputs(self):
        """Per result block the match leg's outputs (c2 / c5: 2-NN lists or SearchByProjection match rows of every frame against the
        previous one; c4: 2-NN lists of left x right), attached to the block: ONE copy takes keypoints, descriptors, counts and
        matches to the host."""
        pkg, L, B, cap = self.pkg, self.L, self.B, self.cap
        self.mo, self.mo_off = [], None
        for blk in range(self.nblk):
            if self.cfg in ("c2", "c5") and self.args.match == "window":
                d = dict(match=pkg.DeviceBuffer(B * cap * 4), nm=pkg.DeviceBuffer(B * 4), gs=pkg.DeviceBuffer(B * 3073 * 4), gi=pkg.DeviceBuffer(B * cap * 4))
                keys = ("match", "nm")
            else:
                rows = B if self.cfg in ("c2", "c5") else B // 2
                d = dict(idx2=pkg.DeviceBuffer(rows * cap * 2 * 4), dist2=pkg.DeviceBuffer(rows * cap * 2 * 4))
                keys = ("idx2", "dist2")
                if self.cfg == "c4":                                # ComputeStereoFishEyeMatches keeps the Lowe-ratio survivors (Frame.cc:1465)
                    d["good"] = pkg.DeviceBuffer(rows * cap)
                    keys = ("idx2", "dist2", "good")
            offs = {}
            for key in keys:
                o = C.c_size_t()
                rc = L.orbx_block_attach(self.ex.h, blk, d[key].ptr, d[key].nbytes, C.byref(o))
                assert rc == 0, rc
                offs[key] = int(o.value)
                self.host_bytes = max(self.host_bytes, offs[key] + ((d[key].nbytes + 255) & ~255))
            self.mo_off = offs
            self.mo.append(d)

    # ---- one step's kernels (what a graph slot holds): extraction into result block `blk` + the match leg on that block
    def _enqueue(self, blk):
        L, ex, mt, r, B, cap = self.L, self.ex, self.mt, self.res[blk], self.B, self.cap
        L.orbx_set_result_block(ex.h, blk)
        ex.enqueue_device(self.ptrs, self.W, self.H, self.stride, self.laps)
        if self.measure_match:                                  # untimed eager pass of report(): stamps around the match leg
            L.orbx_mark(ex.h, 0)
        if self.cfg in ("c2", "c5"):
            mo = self.mo[blk]
            if self.args.match == "window" and B > 1:
                # the monocular tracker's own matcher (Tracking.cc:3203-3211): AssignFeaturesToGrid of every frame, then
                # SearchByProjection(frame i, frame i-1, th = 15, mono) end to end on the device: final match rows + counts
                rc = L.orbm_grid_build_batch_async(mt.h, r["kps"], r["counts"], B, cap, 0.0, 0.0, self.inv_w, self.inv_h, mo["gs"].ptr, mo["gi"].ptr)
                assert rc == 0, rc
                rc = L.orbm_search_by_projection_batch_async(mt.h, r["kps"], r["desc"], r["counts"], cap, mo["gs"].ptr, mo["gi"].ptr,
                                                             0.0, 0.0, self.inv_w, self.inv_h, 1, 0, B - 1, 15.0,
                                                             self.sf_host.ctypes.data_as(C.c_void_p), 8, 0.0, 0.0, None, None, 1,
                                                             mo["match"].ptr, mo["nm"].ptr)
                assert rc == 0, (rc, L.orbm_last_error())
            elif B > 1:          # dense 2-NN of frame i (query) against frame i-1 (train): B-1 pairs
                rc = L.orbm_knn2_batch_async(mt.h, r["desc"] + cap * 32, cap, r["counts"] + 4, r["desc"], cap, r["counts"],
                                             B - 1, cap, mo["idx2"].ptr + cap * 8, mo["dist2"].ptr + cap * 8)
                assert rc == 0, rc
        elif self.cfg == "c4":   # ComputeStereoFishEyeMatches: left descriptors (query) x right descriptors (train) of the same pair
            P = B // 2
            mo = self.mo[blk]
            rc = L.orbm_knn2_ratio_batch_async(mt.h, r["desc"], cap, r["counts"], r["desc"] + P * cap * 32, cap, r["counts"] + 4 * P,
                                               P, 0.7, mo["idx2"].ptr, mo["dist2"].ptr, mo["good"].ptr)
            assert rc == 0, rc
        else:
            self._enqueue_c3(blk)
        if self.measure_match:
            L.orbx_mark(ex.h, 1)

    # ---- config C3 (EuRoC stereo): ComputeStereoMatches per pair, ComputeBoW buckets of the left images, and one
    # SearchForTriangulation_ of every left image against the left image of the previous step (the KeyFrame before it)
    MBF = 47.90639384423901
    MB = MBF / 435.2046959714599                                # Examples/Stereo/EuRoC.yaml:9,28

    def _setup_c3(self):
        np, pkg, L, B, cap = self.np, self.pkg, self.L, self.B, self.cap
        P = B // 2
        # a seeded vocabulary of ORBvoc's shape: k = 10, L = 6 (1 111 111 nodes, 35.6 MB of node descriptors), used as Frame::ComputeBoW
        # uses the real one: levelsup = 4, i.e. the 100 nodes of tree level 2 are the FeatureVector buckets (Frame.cc:905-918).
        # ORBvoc.txt itself is a missing blob in the reference snapshot.
        self.voc_tree = importlib.import_module("orb-slam3_amd.synth").gen_vocabulary(10, 6, seed=7)
        self.voc = pkg.ORBVocabulary(self.mt, self.voc_tree)
        self.levelsup = 4
        self.F12 = np.array([0, 0, 0, 0, 0, 0.11, 0, -0.11, 0], np.float32)    # a fixed small sideways motion between identical pinhole cameras
        self.ep = (1.0e4, 240.0)
        self.sf = self.ex.GetScaleFactors(); self.sig2 = self.ex.GetScaleSigmaSquares()
        self.inv_w = float(np.float32(64) / np.float32(self.W)); self.inv_h = float(np.float32(48) / np.float32(self.H))   # Frame.cc:401-402
        # per result block: mvuRight, mvDepth, SAD scratch, kept counts, bucket ids of the left images, match lists; they travel to the host with the block
        self.c3 = []
        self.c3_off = None
        for blk in range(self.nblk):
            d = dict(ur=pkg.DeviceBuffer(P * cap * 4), dp=pkg.DeviceBuffer(P * cap * 4), sad=pkg.DeviceBuffer(P * cap * 4), kept=pkg.DeviceBuffer(P * 4),
                     nodes=pkg.DeviceBuffer(B * cap * 4), m12=pkg.DeviceBuffer(P * cap * 4), nm=pkg.DeviceBuffer(P * 4),
                     gs=pkg.DeviceBuffer(P * 3073 * 4), gi=pkg.DeviceBuffer(P * cap * 4))
            offs = {}
            for key in ("ur", "dp", "kept", "m12", "nm", "gs", "gi"):
                o = C.c_size_t()
                rc = L.orbx_block_attach(self.ex.h, blk, d[key].ptr, d[key].nbytes, C.byref(o))
                assert rc == 0, rc
                offs[key] = int(o.value)
                self.host_bytes = max(self.host_bytes, offs[key] + ((d[key].nbytes + 255) & ~255))
            self.c3_off = offs
            self.c3.append(d)

    def _enqueue_c3(self, blk):
        L, mt, ex, r, cap, P, B = self.L, self.mt, self.ex, self.res[blk], self.cap, self.B // 2, self.B
        cur, prev = self.c3[blk], self.c3[(blk - 1) % self.nblk]
        rp = self.res[(blk - 1) % self.nblk]
        rc = L.orbm_stereo_batch_async(mt.h, ex.h, 0, P, P, r["kps"], r["desc"], r["counts"], cap, self.MB, self.MBF, cur["ur"].ptr, cur["dp"].ptr, cur["sad"].ptr, cur["kept"].ptr)
        assert rc == 0, (rc, L.orbm_last_error())
        # AssignFeaturesToGrid of the left images (Frame.cc:446-480; rectified EuRoC images: mvKeysUn = mvKeys, bounds = the image)
        rc = L.orbm_grid_build_batch_async(mt.h, r["kps"], r["counts"], P, cap, 0.0, 0.0, self.inv_w, self.inv_h, cur["gs"].ptr, cur["gi"].ptr)
        assert rc == 0, (rc, L.orbm_last_error())
        rc = L.orbm_bow_nodes_batch_async(mt.h, self.voc.h, r["desc"], B * cap, self.levelsup, cur["nodes"].ptr)     # ComputeBoW buckets, left and right images
        assert rc == 0, (rc, L.orbm_last_error())
        # KeyFrame 1 of pair p = its left image now; KeyFrame 2 = the RIGHT image of the same pair as the previous step left it in its
        # result block: another view of the same scene whose epipolar geometry is F12's (sideways baseline), standing in for the
        # neighbouring KeyFrame LocalMapping matches against.  Right images carry no mvuRight (mono features).
        rc = L.orbm_triangulation_batch_async(mt.h, P, cap, r["kps"], r["desc"], r["counts"], cur["nodes"].ptr, cur["ur"].ptr,
                                              rp["kps"] + P * cap * 28, rp["desc"] + P * cap * 32, rp["counts"] + 4 * P, prev["nodes"].ptr + P * cap * 4, None,
                                              self.F12.ctypes.data_as(C.c_void_p), self.ep[0], self.ep[1], self.sf.ctypes.data_as(C.c_void_p),
                                              self.sig2.ctypes.data_as(C.c_void_p), 8, 0, 0, cur["m12"].ptr, cur["nm"].ptr)
        assert rc == 0, (rc, L.orbm_last_error())

    def _capture(self):
        L, ex = self.L, self.ex
        for slot in range(self.nslots):                         # slot s writes result block s
            rc = L.orbx_capture_begin(ex.h, slot)
            assert rc == 0, (rc, L.orbx_last_error())
            self._enqueue(slot % self.nblk)
            rc = L.orbx_capture_end(ex.h)
            assert rc == 0, (rc, L.orbx_last_error())
        self.captured = True

    def prime(self):
        """Untimed setup: every code path of the timed loop runs once (eager enqueue, graph capture + replay, download)."""
        graph, self.graph = self.graph, False
        for _ in range(2):
            self.step()
        self.sync()
        self.graph = graph
        if self.graph:
            self._capture()
            for _ in range(self.nslots):
                self.step()
            self.sync()

    def step(self):
        """Batch k goes into result block k % 8; its results leave for the host (copy thread + copy stream) beside batch k+1,
        which writes the next block; batch k+8 waits for that copy before it rewrites the block."""
        blk = self.k % self.nblk
        if self.graph:
            rc = self.L.orbx_graph_launch(self.ex.h, self.k % self.nslots)
            assert rc == 0, rc
        else:
            self._enqueue(blk)
        if self.download:
            rc = self.L.orbx_result_download_async(self.ex.h, self.h_blk[blk].ptr)
            assert rc == 0, rc
        self.k += 1

    def flush(self):
        pass

    def sync(self):
        self.ex.sync()
        self.mt.sync()

    def counts(self):
        """per-frame keypoint counts of the most recent batch, straight from the device"""
        out = self.np.zeros(self.B, self.np.int32)
        self.L.orbx_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.res[(self.k - 1) % self.nblk]["counts"], 4 * self.B)
        return out

    def mark(self, which):
        self.L.orbx_mark(self.ex.h, which)

    def mark_ms(self):
        t = C.c_float()
        rc = self.L.orbx_mark_elapsed_ms(self.ex.h, C.byref(t))
        assert rc == 0, rc
        return t.value


def timed_loop(wl, steps, barrier, pre=0):
    """EXACTLY `steps` steps between two barriers; returns (wall seconds, host seconds spent enqueueing).  `pre`: the last few of the
    UNTIMED warm-up steps, run here behind the garbage collection -- a collection idles the GPU for milliseconds, and the first
    step after an idle spell is a slow one (clocks); with them the opening barrier finds a busy device."""
    gc.collect()
    gc.disable()
    try:
        for _ in range(pre):
            wl.step()
        barrier()
        if hasattr(wl, "mark"):
            wl.mark(0)
        t0 = time.perf_counter()
        for _ in range(steps):
            wl.step()
        wl.flush()
        t_enq = time.perf_counter() - t0
        if hasattr(wl, "mark"):
            wl.mark(1)
        wl.sync()
        t_own = time.perf_counter() - t0                    # this rank's own K steps, before it waits for the others
        barrier()
        dt = time.perf_counter() - t0
    finally:
        gc.enable()
    timed_loop.last_own = t_own
    return dt, t_enq


def cpu_baseline(wl, args):
    """The oracle (test infrastructure, a port of the reference's CPU algorithm) timed on this box's host cores."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orbref
    pkg = wl.pkg
    W, H, nF, lap = wl.W, wl.H, wl.nF, wl.lap
    ns = args.cpu_sample
    imgs = wl.host_imgs
    out = {}

    def run_stream(ref, frames, match=True):
        prev = None
        OM = orbref._oracle_matcher_class()() if args.match == "window" else None
        for i in frames:
            n, kps, desc, mono = ref(imgs[i % len(imgs)], lap)
            if match and prev is not None:
                if args.match == "knn2":
                    orbref.knn2(desc, prev[1])
                else:                                   # the oracle's SearchByProjection(Frame,Frame) in C over the same windows
                    fv = pkg.FrameView(prev[0], prev[1], W, H, backend=OM)
                    nq = len(kps)
                    OM.SearchByProjectionFrame(fv, np.zeros(fv.n, np.uint8), ref.tables()["sf"], np.ones(nq, np.uint8), kps["x"], kps["y"],
                                               np.zeros(nq, np.float32), kps["octave"], kps["angle"], desc, np.zeros(nq, np.uint8), 15.0)
            prev = (kps, desc)

    if wl.cfg in ("c2", "c5"):
        ref = orbref.Extractor(nF, 1.2, 8, 20, 7)
        tc = time.perf_counter()
        run_stream(ref, range(ns))
        tcpu = time.perf_counter() - tc
        out["cpu_baseline"] = {"value": ns / tcpu, "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": "%d frames of the same synthetic stream through oracle/liborbref.so "
                                         "(extract + %s match vs previous frame), 1 thread" % (ns, args.match),
                               "stage_ms_per_frame": {k: v / ns for k, v in ref.stage_ms().items()}}
        # context figure: frame-parallel over every host core of this box, one oracle extractor per thread
        nthr = max(1, min(os.cpu_count() or 1, 64))
        per = max(2, ns // 4)
        refs = [orbref.Extractor(nF, 1.2, 8, 20, 7) for _ in range(nthr)]
        thr = [threading.Thread(target=run_stream, args=(refs[t], range(t * per, (t + 1) * per))) for t in range(nthr)]
        tc = time.perf_counter()
        for t in thr:
            t.start()
        for t in thr:
            t.join()
        tpar = time.perf_counter() - tc
        out["cpu_context_all_cores"] = {"value": nthr * per / tpar, "unit": "frames/s", "cores": nthr, "kind": "port",
                                        "sample": "%d threads x %d frames each (every thread its own stream and oracle extractor)" % (nthr, per)}
    elif wl.cfg == "c4":
        # the reference starts the two extractions of a stereo frame on two threads (Frame.cc:132-137 / 1363-1364); then the
        # brute-force 2-NN of left x right (Frame.cc:1458)
        P = max(1, ns // 8)
        refs = [orbref.Extractor(nF, 1.2, 8, 20, 7) for _ in range(2)]
        half = wl.B // 2
        tc = time.perf_counter()
        for p in range(P):
            res = [None, None]

            def one(side, idx):
                res[side] = refs[side](imgs[idx], lap)
            th = [threading.Thread(target=one, args=(0, p % half)), threading.Thread(target=one, args=(1, half + p % half))]
            for t in th:
                t.start()
            for t in th:
                t.join()
            orbref.knn2(res[0][2], res[1][2])
        tcpu = time.perf_counter() - tc
        out["cpu_baseline"] = {"value": 2 * P / tcpu, "unit": "frames/s", "cores": 2, "kind": "port",
                               "sample": "%d stereo pairs of the same synthetic stream through oracle/liborbref.so: two extractions on two "
                                         "threads (Frame.cc:1363-1364), then the 2-NN of left x right on one" % P}
    elif wl.cfg == "c3":
        # per stereo pair: two extractions on two threads (Frame.cc:132-137), ComputeStereoMatches, ComputeBoW of the left image and
        # one SearchForTriangulation_ against the previous pair's left image -- the oracle's restatements, same inputs as the GPU step
        half = wl.B // 2
        P = max(2, min(half, ns // 8))
        refs = [orbref.Extractor(nF, 1.2, 8, 20, 7) for _ in range(2)]
        OM = orbref._oracle_matcher_class()()
        voc = orbref.Vocabulary(wl.voc_tree)
        nmatch = []
        tc = time.perf_counter()
        for p in range(P):
            res = [None, None]

            def one(side, idx):
                res[side] = refs[side](imgs[idx], lap)
            th = [threading.Thread(target=one, args=(0, p % half)), threading.Thread(target=one, args=(1, half + p % half))]
            for t in th:
                t.start()
            for t in th:
                t.join()
            (nl, kl, dl, _), (nr, kr, dr, _) = res
            kept, ur, dp = OM.ComputeStereoMatches(refs[0], refs[1], kl, dl, kr, dr, wl.MB, wl.MBF)
            pkg.FrameView(kl, dl, W, H, backend=OM)                                  # AssignFeaturesToGrid
            fv = c3_feature_vector(voc.transform(dl, wl.levelsup)[3])
            fvr = c3_feature_vector(voc.transform(dr, wl.levelsup)[3])
            n_t, m12 = OM.SearchForTriangulation(kl, dl, np.zeros(nl, np.uint8), ur, fv, kr, dr, np.zeros(nr, np.uint8), None, fvr,
                                                 wl.F12, wl.ep, wl.sf, wl.sig2, only_stereo=False, coarse=False, check_ori=False)
            nmatch.append(n_t)
        tcpu = time.perf_counter() - tc
        out["cpu_baseline"] = {"value": 2 * P / tcpu, "unit": "frames/s", "cores": 2, "kind": "port",
                               "sample": "%d stereo pairs of the same synthetic stream through oracle/liborbref.so: two extractions on two threads "
                                         "(Frame.cc:132-137), then ComputeStereoMatches, ComputeBoW buckets of both images and SearchForTriangulation_ (left image vs the right image as the neighbouring view) on one" % P,
                               "triangulation_matches_per_pair": float(np.mean(nmatch)) if nmatch else None}
    return out


def c3_feature_vector(nodes):
    """DBoW2 FeatureVector (std::map<node, vector<idx>>) as CSR: nodes ascending, indices in feature order."""
    import numpy as np
    order = np.argsort(nodes, kind="stable")
    un, start = np.unique(nodes[order], return_index=True)
    return un.astype(np.int32), np.append(start, len(nodes)).astype(np.int32), order.astype(np.int32)


def c3_verify(wl, blk):
    """Outside the timed region: the last step's stereo points and triangulation match lists, as they arrived in the pinned host
    block, against the oracle on the same images (pair 0 and the last pair)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orbref
    P, cap, half = wl.B // 2, wl.cap, wl.B // 2
    hb = wl.h_blk[blk].bytes
    off = wl.c3_off
    ur_h = hb[off["ur"]:off["ur"] + 4 * P * cap].view(np.float32).reshape(P, cap)
    dp_h = hb[off["dp"]:off["dp"] + 4 * P * cap].view(np.float32).reshape(P, cap)
    kept_h = hb[off["kept"]:off["kept"] + 4 * P].view(np.int32)
    m12_h = hb[off["m12"]:off["m12"] + 4 * P * cap].view(np.int32).reshape(P, cap)
    nm_h = hb[off["nm"]:off["nm"] + 4 * P].view(np.int32)
    gs_h = hb[off["gs"]:off["gs"] + 4 * P * 3073].view(np.int32).reshape(P, 3073)
    gi_h = hb[off["gi"]:off["gi"] + 4 * P * cap].view(np.int32).reshape(P, cap)
    OM = orbref._oracle_matcher_class()()
    voc = orbref.Vocabulary(wl.voc_tree)
    ok = True
    cache = {}

    def pair(p):
        if p not in cache:
            ol, orr = orbref.Extractor(wl.nF, 1.2, 8, 20, 7), orbref.Extractor(wl.nF, 1.2, 8, 20, 7)
            nl, kl, dl, _ = ol(wl.host_imgs[p], wl.lap); nr, kr, dr, _ = orr(wl.host_imgs[half + p], wl.lap)
            kept, ur, dp = OM.ComputeStereoMatches(ol, orr, kl, dl, kr, dr, wl.MB, wl.MBF)
            cache[p] = (nl, kl, dl, kept, ur, dp, c3_feature_vector(voc.transform(dl, wl.levelsup)[3]), nr, kr, dr, c3_feature_vector(voc.transform(dr, wl.levelsup)[3]))
        return cache[p]
    for p in sorted({0, P - 1}):
        nl, kl, dl, kept, ur, dp, fv, nr, kr, dr, fvr = pair(p)
        ok = ok and kept == int(kept_h[p]) and ur_h[p, :nl].tobytes() == ur.tobytes() and dp_h[p, :nl].tobytes() == dp.tobytes()
        n_t, m12 = OM.SearchForTriangulation(kl, dl, np.zeros(nl, np.uint8), ur, fv, kr, dr, np.zeros(nr, np.uint8), None, fvr,
                                             wl.F12, wl.ep, wl.sf, wl.sig2, only_stereo=False, coarse=False, check_ori=False)
        ok = ok and n_t == int(nm_h[p]) and np.array_equal(m12_h[p, :nl], m12)
        fv_ = wl.pkg.FrameView(kl, dl, wl.W, wl.H, backend=OM)                       # the oracle's AssignFeaturesToGrid
        placed = int(fv_.grid_start[-1])
        ok = ok and np.array_equal(gs_h[p], fv_.grid_start) and np.array_equal(gi_h[p, :placed], fv_.grid_idx[:placed])
    return bool(ok), int(kept_h.sum()), int(nm_h.sum())


def pin_to_gpu_numa_node(local_rank):
    """Host threads of a rank (enqueue loop, copy thread, RCCL proxy) stay on the CPUs of the NUMA node its GPU hangs off:
    /sys/class/drm/card*/device of the AMD GPUs (vendor 0x1002, in PCI bus order) -> numa_node -> that node's cpulist.  Best effort:
    returns what was done, None where the topology cannot be read (containers without /sys, single-node hosts report -1)."""
    if os.environ.get("ORB_BENCH_NO_PIN") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        import glob
        cards = []
        for d in glob.glob("/sys/class/drm/card[0-9]*/device"):
            try:
                if open(os.path.join(d, "vendor")).read().strip() == "0x1002":
                    cards.append((os.path.basename(os.path.realpath(d)), d))        # PCI address orders them as the runtime does
            except OSError:
                continue
        cards.sort()
        if not cards:
            return None
        d = cards[local_rank % len(cards)][1]
        node = int(open(os.path.join(d, "numa_node")).read().strip())
        if node < 0:
            return {"numa_node": node, "pinned": False}
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)                          # never widen what the launcher / cgroup allows
        if not cpus:
            return {"numa_node": node, "pinned": False}
        os.sched_setaffinity(0, cpus)
        return {"numa_node": node, "pinned": True, "cpus": len(cpus)}
    except Exception:
        return None


def match_verify(wl, blk):
    """Outside the timed region: the match leg's outputs of the last step, as they arrived in the pinned host block, against the
    oracle on the same images (first and last pair of the batch)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orbref
    B, cap = wl.B, wl.cap
    hb = wl.h_blk[blk].bytes
    off = wl.mo_off
    ok = True

    def feat(i):
        return orbref.Extractor(wl.nF, 1.2, 8, 20, 7)(wl.host_imgs[i], wl.lap)
    if wl.cfg in ("c2", "c5"):
        if B < 2:
            return True
        for p in sorted({0, B - 2}):
            nt, kt, dt, _ = feat(p); nq, kq, dq, _ = feat(p + 1)
            if wl.args.match == "window":
                OM = orbref._oracle_matcher_class()()
                m_h = hb[off["match"]:off["match"] + 4 * B * cap].view(np.int32).reshape(B, cap)
                n_h = hb[off["nm"]:off["nm"] + 4 * B].view(np.int32)
                fv = wl.pkg.FrameView(kt, dt, wl.W, wl.H, backend=OM)
                n_ref, m_ref = OM.SearchByProjectionFrame(fv, np.zeros(nt, np.uint8), wl.sf_host, np.ones(nq, np.uint8), kq["x"], kq["y"], np.zeros(nq, np.float32),
                                                          kq["octave"], kq["angle"], dq, np.ones(nq, np.uint8), 15.0, check_ori=True)
                ok = ok and int(n_h[p]) == n_ref and np.array_equal(m_h[p, :nt], m_ref)
            else:
                i_h = hb[off["idx2"]:off["idx2"] + 8 * B * cap].view(np.int32).reshape(B, cap, 2)
                d_h = hb[off["dist2"]:off["dist2"] + 8 * B * cap].view(np.int32).reshape(B, cap, 2)
                ri, rd = orbref.knn2(dq, dt)
                ok = ok and np.array_equal(i_h[p + 1, :nq], ri) and np.array_equal(d_h[p + 1, :nq], rd)
    elif wl.cfg == "c4":
        P = B // 2
        i_h = hb[off["idx2"]:off["idx2"] + 8 * P * cap].view(np.int32).reshape(P, cap, 2)
        d_h = hb[off["dist2"]:off["dist2"] + 8 * P * cap].view(np.int32).reshape(P, cap, 2)
        for p in sorted({0, P - 1}):
            nl, kl, dl, _ = feat(p); nr, kr, dr, _ = feat(P + p)
            ri, rd = orbref.knn2(dl, dr)
            g_h = hb[off["good"]:off["good"] + P * cap].reshape(P, cap)
            want = np.array([1 if (a >= 0 and b >= 0 and float(np.float32(a)) < float(np.float32(b)) * 0.7) else 0 for a, b in rd.tolist()], np.uint8)
            ok = ok and np.array_equal(i_h[p, :nl], ri) and np.array_equal(d_h[p, :nl], rd) and np.array_equal(g_h[p, :nl], want)
    return bool(ok)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    stub = os.environ.get("ORB_BENCH_STUB") == "1"

    pin = pin_to_gpu_numa_node(local_rank) if not stub else None     # before torch / HIP start their threads: they inherit the mask
    import numpy as np
    import torch                       # plumbing: device sync + torch.distributed (RCCL); imported BEFORE the
    import torch.distributed as dist   # HIP library so both share one libamdhip64 runtime
    backend = os.environ.get("ORB_BENCH_BACKEND", "gloo" if stub else "nccl")   # "nccl" IS RCCL on ROCm; "gloo" for rehearsals
    if not stub:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
        ndev = torch.cuda.device_count()
        local_rank = local_rank % max(ndev, 1)             # rehearsals may put several ranks on one GPU (gloo only)
        torch.cuda.set_device(local_rank)
    cdev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize()

    wl = (StubWorkload if stub else OrbWorkload)(args, rank, local_rank)
    B = wl.B
    wl.prime()
    pre = min(2, args.warmup)                                # the last warm-up steps run inside timed_loop, right before its opening barrier
    for _ in range(args.warmup - pre):
        wl.step()
    wl.sync()
    dt, t_enq = timed_loop(wl, args.steps, barrier, pre)     # the metric: kernels + results to the host
    dt_own = timed_loop.last_own
    gpu_wall = wl.mark_ms() if not stub else None
    extra = {}
    if not stub:
        # HIP-event times of the timed steps (recorded on the streams the kernels were launched on; in graph mode the latest
        # replay of every graph slot, i.e. the last steps of the timed region)
        tm, nsamp = wl.ex.mean_timings()
        span_ms, total_ms = tm["pyramid_fast_span"], tm["total"]
        pf_only_ms = tm["fast"]                            # (stage timing off: first launch -> end of the last FAST launch)
        match_ms = wl.mt.timing_ms()
        # the same loop with the results left in HBM (round-1's figure)
        wl.download = False
        for _ in range(2):
            wl.step()
        wl.sync()
        dt_res, t_enq_res = timed_loop(wl, args.steps, barrier, 2)
        wl.download = True
        extra = dict(span_ms=span_ms, pf_only_ms=pf_only_ms, total_ms=total_ms, match_ms=match_ms, nsamp=nsamp, dt_res=dt_res, t_enq_res=t_enq_res)

    n_host = wl.counts()
    dmod = importlib.import_module("orb-slam3_amd.dist")
    tmax = torch.tensor([dt, extra.get("dt_res", 0.0)], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    # RCCL: the path's only exchange -- per-frame keypoint counts of every rank
    gathered = dmod.gather_counts(n_host, device=cdev)
    total_kp = int(torch.stack([g.sum() for g in gathered]).sum().item())
    # every rank's own time for its K steps, measured before the closing barrier (a straggler shows here; `value` uses the
    # barrier-to-barrier maximum)
    mine = torch.tensor([dt_own], dtype=torch.float64, device=cdev)
    if world > 1:
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_ms = [float(t.item()) / args.steps * 1e3 for t in allr]
    else:
        per_rank_ms = [dt_own / args.steps * 1e3]
    dt, dt_res = float(tmax[0].item()), float(tmax[1].item())

    if rank == 0:
        frames = B * args.steps * world
        out = {
            "metric": None, "value": frames / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic", "per_rank_ms_per_step": per_rank_ms, "cpu_affinity_rank0": pin,
        }
        if stub:
            out["metric"] = "stub (no GPU work): launcher / rendezvous / gather rehearsal"
            out["config"] = {"workload": "stub", "frames_per_step_per_gpu": B, "keypoints_last_batch": total_kp}
            print(json.dumps(out))
        else:
            out.update(report(wl, args, world, dt, dt_res, t_enq, gpu_wall, extra, total_kp, frames))
            print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def report(wl, args, world, dt, dt_res, t_enq, gpu_wall, extra, total_kp, frames):
    import numpy as np
    import torch
    ex, mt, B, W, H, nF = wl.ex, wl.mt, wl.B, wl.W, wl.H, wl.nF
    L = wl.L
    # per-stage breakdown: the same batch again, eagerly, with every stage event recorded (untimed)
    L.orbx_set_stage_timing(ex.h, 1)
    was_graph, wl.graph = wl.graph, False
    wl.measure_match = True
    for _ in range(max(3, min(8, args.steps))):
        wl.step()
    wl.sync()
    match_ms = wl.mark_ms()                                     # the match leg of the last of those steps (stamps on the shared stream)
    wl.measure_match = False
    wl.graph = was_graph
    stage, _ = ex.mean_timings()
    stage["pyramid_fast_span_staged"] = stage["pyramid_fast_span"]
    stage["pyramid_fast_span"] = extra["span_ms"]; stage["total"] = extra["total_ms"]       # the live figures of the timed steps
    stage["match"] = match_ms
    L.orbx_set_stage_timing(ex.h, 0)

    alg, fused = ex.algorithmic_bytes()
    # the blur's compulsory traffic is 2 * (sum of level pixels) = 2 * (alg - fused); it belongs to the pass when
    # the blur is scheduled inside it (default: matrix-core blur behind the resize chain, beside FAST) -- the span then ends
    # with the later of FAST and blur
    blur_in = ex.blur_in_pass()
    alg_pf = alg
    if blur_in:
        alg = alg + 2 * (alg - fused)
    pf_ms = extra["span_ms"]                                    # the pass: first launch -> later of (last FAST launch, in-pass blur)
    pfo_ms = extra["pf_only_ms"] or pf_ms                      # first launch -> end of the last FAST launch (pyramid+FAST proper)
    achieved = alg * B / (pf_ms * 1e-3) / 1e9
    achieved_pf = alg_pf * B / (pfo_ms * 1e-3) / 1e9
    traffic, traffic_pf, traffic_src = None, None, None        # HBM bytes per launch group from the committed rocprofv3 --pmc passes
    for name in ("r03_traffic_%s.json" % wl.cfg, "r03_traffic.json", "r02_traffic_%s.json" % wl.cfg, "r02_traffic.json"):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", name)))
            c = tj["config"]
            if (c["width"], c["height"], c["nfeatures"]) == (W, H, nF):
                traffic_pf = tj["pyramid_fast_bytes_per_frame"] * B
                traffic = (tj["pyramid_fast_bytes_per_frame"] + (tj["blur_bytes_per_frame"] if blur_in else 0)) * B
                traffic_src = "profiles/" + name + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command at batch %d, per frame x frames per launch)" % c["batch"]
                break
        except Exception:
            continue
    # the other roofline of this pass: integer VALU issue.  Wave-instructions per frame from the committed SQ_INSTS_VALU pass, issue
    # rate from the committed micro-benchmark; the span is the live one of this run.
    valu = None
    for name in ("r03_valu_%s.json" % wl.cfg, "r03_valu.json"):
        try:
            vj = json.load(open(os.path.join(ROOT, "profiles", name)))
            c = vj["config"]
            if (c["width"], c["height"], c["nfeatures"]) == (W, H, nF) and args.texture == "dense":   # (the PMC pass ran on the dense class: FAST's count follows the corner density)
                pp = vj["pass_per_frame"]
                wi = (pp["resize"] + pp["fast"] + (pp["blur"] if blur_in else 0.0)) * B
                valu = {"wave_instr_per_launch": wi, "achieved": wi / (pf_ms * 1e-3), "peak": vj["peak_wave_instr_per_s"], "unit": "wave-instr/s",
                        "frac": wi / (pf_ms * 1e-3) / vj["peak_wave_instr_per_s"],
                        "kernels": "k_resize2 x7 + k_fast4 x3 + k_fast_fix" + (" + k_blur3" if blur_in else "") + " over the pass span (launch_ms)",
                        "source": "profiles/%s (SQ_INSTS_VALU, rocprofv3 --pmc at batch %d, per frame x frames per launch); peak: %s" % (name, c["batch"], vj["peak_source"])}
                break
        except Exception:
            continue
    # secondary denominator: the device-to-device copy rate this GPU actually reaches, measured here with a
    # 1 GiB torch copy (read + write bytes / time), outside the timed region
    copy_gbs = None
    try:
        n = 1 << 30
        a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
        b.copy_(a); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            b.copy_(a)
        e1.record(); torch.cuda.synchronize()
        copy_gbs = 2.0 * n * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a, b
    except Exception:
        copy_gbs = None
    cw, ch, cnf, cb, _, cname = CONFIGS[wl.cfg]
    named = (W, H, nF) == (cw, ch, cnf)
    legs = {"c2": "dense 2-NN Hamming match (int8 MFMA) of every frame against the previous one",
            "c5": "dense 2-NN Hamming match (int8 MFMA) of every frame against the previous one",
            "c4": "%d fisheye stereo pairs: brute-force 2-NN of left x right descriptors per pair + Lowe ratio 0.7 in the kernel's epilogue (ComputeStereoFishEyeMatches, Frame.cc:1458-1465)" % (B // 2),
            "c3": "%d stereo pairs (the stereo Frame constructor, Frame.cc:103-200, + what LocalMapping does with a new KeyFrame): ComputeStereoMatches and AssignFeaturesToGrid per pair, ComputeBoW buckets (seeded vocabulary of ORBvoc's shape: k=10, L=6, levelsup=4) and one SearchForTriangulation_ per pair "
                  "(its left image against the right image of the previous step as the neighbouring KeyFrame)" % (B // 2)}
    if args.match == "window":
        legs["c2"] = legs["c5"] = ("AssignFeaturesToGrid + SearchByProjection(frame, previous frame, th=15, mono; ORBmatcher.cc:2469-2711) of every frame, "
                                   "final match rows (claims, TH_HIGH, rotation-histogram cull) on the device")
    d2h = wl.host_bytes
    out = {
        "metric": "ORB extract+match frames/sec @%dx%d, %d feat" % (W, H, nF),
        "config": {"workload": (cname + ": " if named else "other size (not a named config): ") +
                               "%dx%d grayscale (synthetic, texture class '%s', %d distinct frames per rank), %d features, 8 levels, scale 1.2, FAST 20/7; batch of %d frames/GPU/step resident in HBM; "
                               "extract + %s; keypoints, descriptors, counts AND the match leg's outputs of every batch copied to pinned host memory inside the timed region"
                               % (W, H, args.texture, max(1, min(args.distinct, B)), nF, B, legs[wl.cfg]),
                   "frames_per_step_per_gpu": B, "launch": "hipGraph replay (one hipGraphLaunch per step)" if wl.graph else "eager enqueue",
                   "keypoints_last_batch": total_kp, "result_bytes_to_host_per_step": d2h},
        "value_device_resident": frames / dt_res, "ms_per_step_device_resident": dt_res / args.steps * 1e3,
        # host time of the calls that enqueue a step (no waiting involved: measured in the device-resident loop), and of the
        # timed loop itself, which also holds the flow control of the result ring (a block is rewritten only after its copy landed)
        "host_enqueue_ms_per_step": extra["t_enq_res"] / args.steps * 1e3,
        "host_loop_ms_per_step": t_enq / args.steps * 1e3,
        "gpu_wall_ms_per_step": gpu_wall / args.steps,
        "gpu_total_ms_per_step": extra["total_ms"] + match_ms,
        # Headline figure = pyramid+FAST proper (SURVEY 8(d)'s object): its algorithmic bytes over the live span from the first launch to
        # the end of the last FAST launch.  What BINDS the pass is integer VALU issue, not HBM: `valu` holds that roofline (whole pass,
        # blur included, because the blur shares the window and the pipes), `with_blur` the HBM figure of the whole pass.
        "roofline": {"bound": ("valu" if valu and valu["frac"] > achieved / HBM_PEAK_GBS else "hbm"),
                     "kernel": "pyramid+FAST pass (k_resize2 x7 on stream 2 + k_fast4 x3 and k_fast_fix on stream 1; wall span by HIP events / on-stream stamps, "
                               "first launch -> end of the last FAST launch" + ("; k_blur3 runs beside it on stream 2)" if blur_in else ")"),
                     "achieved": achieved_pf, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_pf / HBM_PEAK_GBS,
                     "traffic": traffic_pf, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": alg_pf * B, "frames_per_launch": B, "algorithmic_bytes_per_frame": alg_pf,
                     "fused_lower_bound_per_frame": fused, "launch_ms": pfo_ms, "launch_samples": extra["nsamp"],
                     "measured_copy_peak_GBps": copy_gbs,
                     "frac_of_measured_copy": (achieved_pf / copy_gbs) if copy_gbs else None,
                     "valu": valu,
                     "with_blur": ({"kernel": "the same pass with k_blur3 (scheduled inside it): span to the later of FAST and blur",
                                    "algorithmic_bytes_per_frame": alg, "launch_ms": pf_ms, "achieved": achieved, "frac": achieved / HBM_PEAK_GBS,
                                    "traffic": traffic} if blur_in else None)},
        "stage_ms_per_step": stage,
    }
    # host copy == device results (outside the timed region): one more step, then compare its pinned copy with per-frame fetches
    wl.step()
    wl.sync()
    blk = (wl.k - 1) % wl.nblk
    ok_, od_, on_, om_, _ = wl.layout
    hb = wl.h_blk[blk].bytes
    n_pin = hb[on_:on_ + 4 * B].view(np.int32); m_pin = hb[om_:om_ + 4 * B].view(np.int32)
    k_pin = hb[ok_:ok_ + 28 * wl.cap * B].reshape(B, wl.cap, 28); d_pin = hb[od_:od_ + 32 * wl.cap * B].reshape(B, wl.cap, 32)
    ok = bool(np.array_equal(wl.counts(), n_pin))
    for i in sorted({0, B // 2, B - 1}):
        mono, kps, desc = ex.fetch(i)
        n = len(kps)
        ok = ok and n == int(n_pin[i]) and mono == int(m_pin[i]) and np.array_equal(kps.view(np.uint8).reshape(n, 28), k_pin[i, :n]) and np.array_equal(desc, d_pin[i, :n])
    out["host_copy_matches_device"] = ok
    if wl.cfg != "c3":
        out["match_leg_matches_oracle"] = match_verify(wl, blk)
    if wl.cfg == "c3":
        ok3, nst, ntri = c3_verify(wl, blk)
        out["stereo_and_triangulation_match_oracle"] = ok3
        out["config"]["stereo_points_last_step"] = nst
        out["config"]["triangulation_matches_last_step"] = ntri
    if world == 1 and args.cpu_sample > 0:
        out.update(cpu_baseline(wl, args))
    return out


if __name__ == "__main__":
    main()
