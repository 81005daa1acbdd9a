#!/usr/bin/env python3
"""profiles/r03_valu.json: VALU wave-instructions per frame of every extractor kernel, from a rocprofv3 --pmc SQ_INSTS_VALU pass
(bash tools/pmc.sh <dir> SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU  -- batch 128, eager), and the integer issue
rate of the committed micro-benchmark (profiles/r01_valu_rate_ubench.txt, tools/ubench/valu_rate.hip: 8 workgroups of 4 waves per CU).
    python tools/make_valu_json.py gpurun_out/<dir> <width> <height> <nfeatures> <batch> [suffix]"""
import collections, csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, width, height, nfeat, B = sys.argv[1], *(int(x) for x in sys.argv[2:6])
suffix = sys.argv[6] if len(sys.argv) > 6 else ""
f = max(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "SQ_INSTS_VALU":
        acc[r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]].append(float(r["Counter_Value"]))
steps = len(acc["k_resize2"]) / 7.0                       # seven resize launches per step
per_frame = {k: sum(v) / steps / B for k, v in acc.items() if k.startswith("k_")}
rates = []
for line in open(os.path.join(ROOT, "profiles", "r01_valu_rate_ubench.txt")):
    m = re.match(r"(k_\w+)\s+[\d.]+ ms\s+([\d.]+) T lane-instr/s", line)
    if m and m.group(1) in ("k_pkmin", "k_pkmax_i", "k_pkadd", "k_pksub_i", "k_perm", "k_align", "k_min", "k_max_i", "k_andor", "k_bfe", "k_lshlor"):
        rates.append(float(m.group(2)))
rate = sorted(rates)[len(rates) // 2] * 1e12 / 64.0       # median over the integer ops these kernels issue; wave-instructions per second, whole chip
fast = sum(v for k, v in per_frame.items() if k.startswith("k_fast"))
out = {"config": {"width": width, "height": height, "nfeatures": nfeat, "batch": B},
       "units": "SQ_INSTS_VALU (wave-instructions) per frame; rocprofv3 --pmc, one pass, eager launches",
       "per_frame": per_frame,
       "pass_per_frame": {"resize": per_frame.get("k_resize2", 0.0), "fast": fast, "blur": per_frame.get("k_blur3", 0.0)},
       "peak_wave_instr_per_s": rate,
       "peak_source": "profiles/r01_valu_rate_ubench.txt: median of v_pk_min/max/add/sub_u16, v_perm, v_alignbyte, v_min/max, and/or, bfe, lshl_or "
                      "at 8 waves per SIMD (%.1f T lane-ops/s = 4.5 cycles per wave-instruction and SIMD)" % (rate * 64 / 1e12)}
name = "r03_valu%s.json" % (("_" + suffix) if suffix else "")
json.dump(out, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
print(name, {k: round(v) for k, v in out["pass_per_frame"].items()}, "peak %.3g wave-instr/s" % rate)
