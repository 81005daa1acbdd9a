#!/usr/bin/env python3
"""bench.py -- ORB extract+match frames/sec @752x480, 1000 features (BASELINE.json metric).

One step = one pass of the hot path over one batch of B synthetic frames already resident in HBM:
  extraction of every frame (8-level pyramid, FAST, quadtree, orientation, blur, rBRIEF) followed by the
  Hamming match of each frame's descriptors against the previous frame's (dense 2-NN, orbm_knn2).
N GPUs = N processes (torch.distributed / RCCL), each with its own B frames ("weak" scaling); the only
collective is the all_gather of per-frame keypoint counts after the timed steps' last batch.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline"     -- pyramid+FAST pass: algorithmic bytes / HIP-event time of those launches
  "cpu_baseline" -- the CPU oracle (a port, 1 thread) on a bounded sample of the same frames.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

HBM_PEAK_GBS = 8000.0       # MI355X spec (MI355X_MICROARCH.md): 8.0 TB/s; 6.29 TB/s measured copy peak


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="frames per GPU per step (a larger batch amortises the dependent resize chain and the latency-bound quadtree)")
    ap.add_argument("--width", type=int, default=752)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic frames generated per rank")
    ap.add_argument("--match-stream", choices=["ex", "own"], default="ex",
                    help="where the matcher's kernels run when one handle is in flight (see the step loop)")
    ap.add_argument("--match", choices=["window", "knn2"], default="knn2",
                    help="match leg: 'window' = the mono SearchByProjection window search of every frame's keypoints in the previous "
                         "frame (grid gather + Hamming, ORBmatcher.cc:2543-2612); 'knn2' = dense brute-force 2-NN (Frame.cc:1440-1480)")
    ap.add_argument("--handles", type=int, default=1, help="extractor handles kept in flight per GPU (the batch is split over them)")
    ap.add_argument("--cpu-sample", type=int, default=96, help="frames timed through the CPU oracle (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch                       # plumbing: device sync + torch.distributed (RCCL); imported BEFORE the
    import torch.distributed as dist   # HIP library so both share one libamdhip64 runtime
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    local_rank = local_rank % max(ndev, 1)             # rehearsals may put several ranks on one GPU (gloo only)
    torch.cuda.set_device(local_rank)
    backend = os.environ.get("ORB_BENCH_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm; "gloo" only for single-GPU rehearsals
    cdev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    pkg = importlib.import_module("orb-slam3_amd")
    synth = importlib.import_module("orb-slam3_amd.synth")
    L = pkg.lib()

    W, H, B = args.width, args.height, args.batch
    # The batch of B frames is split over `--handles` extractor handles (own streams + scratch each) that are kept in
    # flight together: while one half-batch is in its latency-bound quadtree/descriptor kernels the other one runs the
    # VALU-bound FAST kernel.  No host sync inside a step: matcher streams wait on extractor streams through events.
    NH = max(1, min(args.handles, B))
    sizes = [B // NH + (1 if i < B % NH else 0) for i in range(NH)]
    exs = [pkg.ORBextractor(args.nfeatures, 1.2, 8, 20, 7, device=local_rank, max_size=(W, H), max_batch=sz) for sz in sizes]
    mts = [pkg.ORBmatcher(0.7, device=local_rank) for _ in range(NH)]
    ex = exs[0]
    cap = ex.cap

    # ---- inputs resident in HBM before the timed region
    stride = (W + 63) // 64 * 64
    host_imgs = [synth.gen_image(W, H, 1000 * rank + 1 + i) for i in range(min(args.distinct, B))]
    dev = pkg.DeviceBuffer(stride * H * B)
    padded = np.zeros((H, stride), np.uint8)
    for i in range(B):
        padded[:, :W] = host_imgs[i % len(host_imgs)]
        dev.upload(padded, offset=i * stride * H)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
    ptrs = [(C.c_void_p * sizes[h])(*[int(dev.ptr + int(offs[h] + i) * stride * H) for i in range(sizes[h])]) for h in range(NH)]
    laps = [np.tile(np.array([0, 1000], np.int32), sizes[h]) for h in range(NH)]   # monocular call: vLapping = {0,1000} (Frame.cc:361)
    res = [e.result_device() for e in exs]
    idx2 = [pkg.DeviceBuffer(max(1, sizes[h]) * cap * 2 * 4) for h in range(NH)]
    dist2 = [pkg.DeviceBuffer(max(1, sizes[h]) * cap * 2 * 4) for h in range(NH)]
    xidx = pkg.DeviceBuffer(NH * cap * 2 * 4); xdist = pkg.DeviceBuffer(NH * cap * 2 * 4)
    gstart = [pkg.DeviceBuffer(max(1, sizes[h]) * 3073 * 4) for h in range(NH)]
    gidx = [pkg.DeviceBuffer(max(1, sizes[h]) * cap * 4) for h in range(NH)]
    sdist = [pkg.DeviceBuffer(max(1, sizes[h]) * cap * 4) for h in range(NH)]
    sf_host = ex.GetScaleFactors()
    inv_w = float(np.float32(64) / np.float32(W)); inv_h = float(np.float32(48) / np.float32(H))   # Frame.cc:401-402, identity undistortion

    # --match-stream ex: the matcher runs on the extractor's stream (batch i's match, then batch i+1's extraction, in order);
    # own: on its own stream behind batch i's extraction, beside batch i+1's pyramid/FAST phase -- batch i+1 waits for it only
    # before its first kernel that writes the result block (orbx_guard_results)
    same_stream = NH == 1 and args.match_stream == "ex"
    guard = NH == 1 and args.match_stream == "own"
    if same_stream:
        assert L.orbm_set_stream(mts[0].h, L.orbx_stream(exs[0].h)) == 0

    def step():
        for h in range(NH):
            # the next batch may overwrite result buffers the matcher of the previous step still reads
            if guard:
                L.orbx_guard_results(exs[h].h, L.orbm_stream(mts[h].h))
            elif not same_stream:
                L.orbx_stream_wait_other(exs[h].h, L.orbm_stream(mts[h].h))
            if h > 0:
                L.orbx_stream_wait_other(exs[h].h, L.orbm_stream(mts[h - 1].h))
            exs[h].enqueue_device(ptrs[h], W, H, stride, laps[h])
        for h in range(NH):
            ms = L.orbm_stream(mts[h].h)
            if not same_stream:
                L.orbx_stream_wait_results(exs[h].h, ms)
            r = res[h]
            if args.match == "window":
                # Frame grid (M14) for every frame, then each frame's keypoints search the previous frame inside the
                # SearchByProjection window th=15 (mono, Tracking.cc:3203-3208); frames of other handles are skipped here
                rc = L.orbm_grid_build_batch_async(mts[h].h, r["kps"], r["counts"], sizes[h], cap, 0.0, 0.0, inv_w, inv_h,
                                                   gstart[h].ptr, gidx[h].ptr)
                assert rc == 0, rc
                if sizes[h] > 1:
                    rc = L.orbm_track_window_batch_async(mts[h].h, r["kps"], r["desc"], r["counts"], cap, gstart[h].ptr, gidx[h].ptr,
                                                         0.0, 0.0, inv_w, inv_h, 1, 0, sizes[h] - 1, 15.0,
                                                         sf_host.ctypes.data_as(C.c_void_p), 8, 0.0, 0.0,
                                                         idx2[h].ptr, dist2[h].ptr, sdist[h].ptr)
                    assert rc == 0, rc
                continue
            # dense 2-NN Hamming match of frame i (query) against frame i-1 (train) inside the half-batch ...
            if sizes[h] > 1:
                rc = L.orbm_knn2_batch_async(mts[h].h, r["desc"] + cap * 32, cap, r["counts"] + 4, r["desc"], cap, r["counts"],
                                             sizes[h] - 1, cap, idx2[h].ptr + cap * 8, dist2[h].ptr + cap * 8)
                assert rc == 0, rc
            # ... and of its first frame against the last frame of the previous half-batch (B-1 pairs in total)
            if h > 0:
                L.orbx_stream_wait_results(exs[h - 1].h, ms)
                p = res[h - 1]
                rc = L.orbm_knn2_batch_async(mts[h].h, r["desc"], cap, r["counts"], p["desc"] + (sizes[h - 1] - 1) * cap * 32, cap,
                                             p["counts"] + 4 * (sizes[h - 1] - 1), 1, cap, xidx.ptr + h * cap * 8, xdist.ptr + h * cap * 8)
                assert rc == 0, rc

    def sync_all():
        for e in exs:
            e.sync()
        for m in mts:
            m.sync()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the timed steps record only the dependency events (start, end of the pyramid+FAST pass, end of the batch): the
    # roofline span is measured live on them; the per-stage breakdown comes from a short untimed pass afterwards
    for e in exs:
        L.orbx_set_stage_timing(e.h, 0)
    for _ in range(args.warmup):
        step()
    sync_all()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    barrier()
    dt = time.perf_counter() - t0

    # HIP-event times of the timed steps (events recorded on the streams the kernels were launched on)
    t_all = []
    for e, sz in zip(exs, sizes):
        tm, ns = e.mean_timings()
        t_all.append((tm, sz))
    span_timed = [tm["pyramid_fast_span"] for tm, _ in t_all]
    total_timed = [tm["total"] for tm, _ in t_all]
    for e in exs:                                             # stage breakdown: same steps again with every stage event, untimed
        L.orbx_set_stage_timing(e.h, 1)
    for _ in range(max(3, min(8, args.steps))):
        step()
    sync_all()
    t_all = []
    for e, sz, sp_t, tot_t in zip(exs, sizes, span_timed, total_timed):
        tm, ns = e.mean_timings()
        tm["pyramid_fast_span_staged"] = tm["pyramid_fast_span"]
        tm["pyramid_fast_span"] = sp_t; tm["total"] = tot_t     # the live figures of the timed steps
        t_all.append((tm, sz))
    t_pyr = [sum(tm["pyramid_fast_span"] for tm, _ in t_all) / len(t_all)]
    t_fast = [0.0]
    mt = mts[0]
    n_host = np.zeros(B, np.int32)                         # keypoint counts of the last batch
    for h in range(NH):
        part = np.zeros(sizes[h], np.int32)
        L.orbx_memcpy_d2h(part.ctypes.data_as(C.c_void_p), res[h]["counts"], 4 * sizes[h])
        n_host[offs[h]:offs[h + 1]] = part
    dmod = importlib.import_module("orb-slam3_amd.dist")
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    # RCCL: the path's only exchange -- per-frame keypoint counts of every rank
    gathered = dmod.gather_counts(n_host, device=cdev)
    total_kp = torch.stack([g.sum() for g in gathered]).sum()
    dt = float(tmax.item())

    if rank == 0:
        frames = B * args.steps * world
        alg, fused = ex.algorithmic_bytes()
        # the blur's compulsory traffic is 2 * (sum of level pixels) = 2 * (alg - fused); it belongs to the pass
        # when the blur is scheduled inside it (default: matrix-core blur behind the resize chain, beside FAST) -- the span
        # then ends with the later of FAST and blur
        blur_in = ex.blur_in_pass()
        alg_pf = alg
        if blur_in:
            alg = alg + 2 * (alg - fused)
        # every handle processes its share of the batch concurrently: per-launch-group figure = bytes of ONE handle's
        # frames / that handle's own pyramid+FAST wall span (conservative: the spans overlap other handles' kernels)
        pf_ms = float(np.mean(t_pyr) + np.mean(t_fast))
        achieved = alg * (B / NH) / (pf_ms * 1e-3) / 1e9
        traffic = None                                      # HBM bytes/launch-group from the committed rocprofv3 --pmc passes
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            c = tj["config"]
            if (c["width"], c["height"], c["nfeatures"]) == (W, H, args.nfeatures):
                traffic = (tj["pyramid_fast_bytes_per_frame"] + (tj["blur_bytes_per_frame"] if blur_in else 0)) * (B / NH)
        except Exception:
            traffic = None
        stage = {k: float(np.mean([tm[k] for tm, _ in t_all])) for k in t_all[0][0]}
        stage["match_" + args.match] = mt.timing_ms()
        # secondary denominator: the device-to-device copy rate this GPU actually reaches, measured here with a
        # 1 GiB torch copy (read + write bytes / time), outside the timed region
        copy_gbs = None
        try:
            n = 1 << 30
            a = torch.empty(n, dtype=torch.uint8, device=cdev); b = torch.empty_like(a)
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
        out = {
            "metric": "ORB extract+match frames/sec @%dx%d, %d feat" % (W, H, args.nfeatures),
            "value": frames / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": ("configs[1]: " if (W, H, args.nfeatures) == (752, 480, 1000) else "other size (not the headline config): ") +
                                   "%dx%d grayscale, %d features, 8 levels, scale 1.2, FAST 20/7; "
                                   "batch of %d frames/GPU/step resident in HBM; extract + %s against the previous frame"
                                   % (W, H, args.nfeatures, B, "Frame grid build and SearchByProjection window match (th=15) of every keypoint"
                                      if args.match == "window" else "dense 2-NN Hamming match (int8 MFMA)"),
                       "frames_per_step_per_gpu": B, "handles_in_flight": NH, "keypoints_last_batch": int(total_kp.item())},
            "roofline": {"bound": "hbm", "kernel": ("pyramid+FAST+blur pass (k_resize2 x7 then k_blur3 on stream 2 + k_fast3 x3 and k_fast_fix on stream 1; wall span by HIP events)"
                                    if blur_in else "pyramid+FAST pass (k_resize2 x7 on stream 2 + k_fast3 x3 and k_fast_fix on stream 1; wall span by HIP events)"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes_per_launch": alg * (B / NH), "frames_per_launch": B / NH, "algorithmic_bytes_per_frame": alg, "fused_lower_bound_per_frame": fused,
                         "launch_ms": pf_ms, "measured_copy_peak_GBps": copy_gbs,
                         # for continuity with the pyramid+FAST-only figure: its own bytes over the span to the end of the last FAST
                         # launch (stage events of the untimed staged pass; FAST is stretched by the blur running beside it)
                         "pyramid_fast_only": ({"algorithmic_bytes_per_frame": alg_pf, "span_ms": stage["fast"],
                                                "achieved": alg_pf * (B / NH) / (stage["fast"] * 1e-3) / 1e9,
                                                "frac": alg_pf * (B / NH) / (stage["fast"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                                               if blur_in and stage.get("fast", 0) > 0 else None),
                         "frac_of_measured_copy": (achieved / copy_gbs) if copy_gbs else None},
            "stage_ms_per_step": stage,
        }
        if world == 1 and args.cpu_sample > 0:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import orbref                                   # the checker, timed as the CPU baseline ("port")
            ref = orbref.Extractor(args.nfeatures, 1.2, 8, 20, 7)
            OM = orbref._oracle_matcher_class()()
            ns = args.cpu_sample
            prev = None
            tc = time.perf_counter()
            for i in range(ns):
                n, kps, desc, mono = ref(host_imgs[i % len(host_imgs)], (0, 1000))
                if prev is not None:
                    if args.match == "knn2":
                        orbref.knn2(desc, prev[1])
                    else:                                   # the oracle's SearchByProjection(Frame,Frame) in C over the same windows
                        fv = pkg.FrameView(prev[0], prev[1], W, H, backend=OM)
                        nq = len(kps)
                        OM.SearchByProjectionFrame(fv, np.zeros(fv.n, np.uint8), ref.tables()["sf"], np.ones(nq, np.uint8), kps["x"], kps["y"],
                                                   np.zeros(nq, np.float32), kps["octave"], kps["angle"], desc, np.zeros(nq, np.uint8), 15.0)
                prev = (kps, desc)
            tcpu = time.perf_counter() - tc
            out["cpu_baseline"] = {"value": ns / tcpu, "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": "%d frames of the same synthetic stream through oracle/liborbref.so "
                                             "(extract + %s match vs previous frame), 1 thread" % (ns, args.match),
                                   "stage_ms_per_frame": {k: v / ns for k, v in ref.stage_ms().items()}}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
