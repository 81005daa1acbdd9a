#!/usr/bin/env python3
"""Derives profiles/r01_traffic.json (HBM bytes of the pyramid+FAST pass per frame) from two rocprofv3 --pmc passes:
    bash tools/pmc.sh pmc_fetch FETCH_SIZE ; bash tools/pmc.sh pmc_write WRITE_SIZE      (on the GPU box)
    python tools/make_traffic_json.py gpurun_out/pmc_fetch gpurun_out/pmc_write
Correction as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE of a wide (16 B/lane) coalesced stream reads
half of the real bytes on gfx950 -> doubled for k_fast3; 4 B/lane loads (k_resize2) are uncalibrated -> taken as is;
WRITE_SIZE is exact.  Counters are in KB."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(d, counter):
    f = max(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)   # newest run in the directory
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0].split("::")[-1]].append(float(r["Counter_Value"]))
    return f, acc


ff, F = per_kernel(sys.argv[1], "FETCH_SIZE")
wf, W = per_kernel(sys.argv[2], "WRITE_SIZE")
B = 128
steps = len(F["k_resize2"]) / 7.0          # seven resize launches per step (levels 1..7); FAST is 3 launches + k_fast_fix
fast_fetch = (sum(F["k_fast3"]) + sum(F.get("k_fast_fix", [0]))) / steps; rz_fetch = (sum(F["k_resize2"]) + sum(F.get("k_resize", [0]))) / steps
fast_write = (sum(W["k_fast3"]) + sum(W.get("k_fast_fix", [0]))) / steps; rz_write = (sum(W["k_resize2"]) + sum(W.get("k_resize", [0]))) / steps
traffic = (2.0 * fast_fetch + rz_fetch + fast_write + rz_write) * 1024 / B
bl_fetch = sum(F.get("k_blur3", [0])) / steps; bl_write = sum(W.get("k_blur3", [0])) / steps     # 16 B/lane loads: doubled like k_fast3
blur_traffic = (2.0 * bl_fetch + bl_write) * 1024 / B
out = {"config": {"width": 752, "height": 480, "nfeatures": 1000, "batch": B},
       "units": "bytes per frame for the pyramid+FAST pass (k_resize2 x7 + k_fast3 x3 + k_fast_fix), rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate passes (KB)",
       "raw_kb_per_step": {"k_fast3_fetch": fast_fetch, "k_resize_fetch": rz_fetch, "k_fast3_write": fast_write, "k_resize_write": rz_write,
                           "k_blur3_fetch": bl_fetch, "k_blur3_write": bl_write},
       "correction": "k_fast3 loads 16 B/lane: FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B); k_resize2 loads 4 B/lane: uncalibrated, as is; WRITE_SIZE exact",
       "pyramid_fast_bytes_per_frame": traffic, "algorithmic_bytes_per_frame": 2963001,
       "blur_bytes_per_frame": blur_traffic, "blur_algorithmic_bytes_per_frame": 2234734}
json.dump(out, open(os.path.join(ROOT, "profiles", "r01_traffic.json"), "w"), indent=1)
shutil.copy(ff, os.path.join(ROOT, "profiles", "r01_pmc_fetch_counter_collection.csv"))
shutil.copy(wf, os.path.join(ROOT, "profiles", "r01_pmc_write_counter_collection.csv"))
print(json.dumps(out["raw_kb_per_step"]), round(traffic))
