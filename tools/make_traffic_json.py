#!/usr/bin/env python3
"""Derives profiles/r03_traffic[_<cfg>].json (HBM bytes of the pyramid+FAST(+blur) pass per frame) from two rocprofv3 --pmc passes:
    bash tools/pmc.sh pmc_fetch FETCH_SIZE ; bash tools/pmc.sh pmc_write WRITE_SIZE      (on the GPU box; PMC_BENCH_ARGS selects the config)
    python tools/make_traffic_json.py gpurun_out/pmc_fetch gpurun_out/pmc_write <width> <height> <nfeatures> <batch> [suffix]
Correction, calibrated this round on known byte counts in this pass's own access shapes (tools/ubench/fetch_calib.hip,
profiles/r02_fetch_calibration.txt): on gfx950 FETCH_SIZE = 1/2 x the 128-byte lines actually fetched, for the wide 1-KiB-per-wave
stream of the guide AND for 160-B / 512-B row runs of 16-B pieces (k_blur3 / k_fast3) AND for overlapping unaligned 8-byte loads
(k_resize2: 0.54) -- memory is fetched in 128-B lines and the counter tallies each at 64 B.  So every FETCH_SIZE is doubled;
WRITE_SIZE is exact (MI355X_MICROARCH.md, HBM section).  Counters are in KB."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(d, counter):
    f = max(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)   # newest run in the directory
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]].append(float(r["Counter_Value"]))
    return f, acc


ff, F = per_kernel(sys.argv[1], "FETCH_SIZE")
wf, W = per_kernel(sys.argv[2], "WRITE_SIZE")
width, height, nfeat, B = (int(x) for x in sys.argv[3:7])
suffix = sys.argv[7] if len(sys.argv) > 7 else ""
alg_pf, alg_blur = int(sys.argv[8]) if len(sys.argv) > 8 else 2963001, int(sys.argv[9]) if len(sys.argv) > 9 else 2234734
steps = len(F["k_resize2"]) / 7.0          # seven resize launches per step (levels 1..7); FAST is 3 launches + k_fast_fix
def fast_sum(D):
    return sum(sum(v) for k, v in D.items() if k.startswith("k_fast"))       # k_fast4 / k_fast3 launches + k_fast_fix


fast_fetch = fast_sum(F) / steps; rz_fetch = (sum(F["k_resize2"]) + sum(F.get("k_resize", [0]))) / steps
fast_write = fast_sum(W) / steps; rz_write = (sum(W["k_resize2"]) + sum(W.get("k_resize", [0]))) / steps
traffic = (2.0 * fast_fetch + 2.0 * rz_fetch + fast_write + rz_write) * 1024 / B
bl_fetch = sum(F.get("k_blur3", [0])) / steps; bl_write = sum(W.get("k_blur3", [0])) / steps
blur_traffic = (2.0 * bl_fetch + bl_write) * 1024 / B
out = {"config": {"width": width, "height": height, "nfeatures": nfeat, "batch": B},
       "units": "bytes per frame for the pyramid+FAST pass (k_resize2 x7 + k_fast4 x3 + k_fast_fix) and for k_blur3, rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate passes (KB)",
       "raw_kb_per_step": {"k_fast3_fetch": fast_fetch, "k_resize_fetch": rz_fetch, "k_fast3_write": fast_write, "k_resize_write": rz_write,
                           "k_blur3_fetch": bl_fetch, "k_blur3_write": bl_write},
       "correction": "every FETCH_SIZE doubled: calibrated on known byte counts in this pass's access shapes (profiles/r02_fetch_calibration.txt): the counter reads 0.50-0.54 x "
                     "the 128-byte lines really fetched; WRITE_SIZE exact",
       "pyramid_fast_bytes_per_frame": traffic, "algorithmic_bytes_per_frame": alg_pf,
       "blur_bytes_per_frame": blur_traffic, "blur_algorithmic_bytes_per_frame": alg_blur,
       "ratio_pyramid_fast": traffic / alg_pf, "ratio_blur": blur_traffic / alg_blur}
name = "r03_traffic%s.json" % (("_" + suffix) if suffix else "")
json.dump(out, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
shutil.copy(ff, os.path.join(ROOT, "profiles", "r03_pmc_fetch%s_counter_collection.csv" % (("_" + suffix) if suffix else "")))
shutil.copy(wf, os.path.join(ROOT, "profiles", "r03_pmc_write%s_counter_collection.csv" % (("_" + suffix) if suffix else "")))
print(name, json.dumps(out["raw_kb_per_step"]), round(traffic), round(blur_traffic), round(out["ratio_pyramid_fast"], 3), round(out["ratio_blur"], 3))
