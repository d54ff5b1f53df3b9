#!/usr/bin/env python3
"""Summarise a tools/profile.sh run (gpurun_out/prof_TAG) into profiles/: a kernel-stats CSV copy,
a counters JSON and profiles/traffic.json (HBM bytes per launch for bench.py's roofline.traffic).

FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts
128-byte requests at 64 bytes for wide coalesced streaming reads, so the read side is doubled;
WRITE_SIZE is exact for 16-byte-per-lane streaming stores.  Both raw and corrected values are kept."""
import csv
import glob
import os
import json
import shutil
import sys
from collections import defaultdict
from pathlib import Path

tag, rnd = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "r01"
src = Path("gpurun_out") / f"prof_{tag}"
dst = Path("profiles")
dst.mkdir(exist_ok=True)
summary = {"tag": tag}
def newest(pattern):
    """the most recently written match (gpurun merges every call's files into gpurun_out/: older runs of a tag stay beside the new one)"""
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]


for f in newest(str(src / "kt" / "*" / "*_kernel_stats.csv")):
    shutil.copy(f, dst / f"{rnd}_{tag}_kernel_stats.csv")
    rows = list(csv.DictReader(open(f)))
    summary["kernel_stats"] = [{k: r[k][:120] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "Percentage")}
                               for r in rows[:3]]
for f in newest(str(src / "kt" / "*" / "*_kernel_trace.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "lutr::" in r["Kernel_Name"] and "make_lat16" not in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    summary["dispatch_us"] = [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in rows]
counters = defaultdict(list)
kernel = None
# the kernel that carries the launch: the lutr kernel with the most time in the kernel-trace pass (helpers such as the
# fp16 lattice builder also live in namespace lutr)
main = None
if summary.get("kernel_stats"):
    cand = [k for k in summary["kernel_stats"] if "lutr::" in k["Name"]]
    if cand:
        main = max(cand, key=lambda k: float(k["Calls"]) * float(k["AverageNs"]))["Name"].split("(")[0]
summary["kernel_stats"] = [k for k in summary.get("kernel_stats", [])]
for sub in ("fetch", "write", "sq", "tcc", "gui"):
    for f in newest(str(src / sub / "*" / "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "lutr::" in r["Kernel_Name"] and (main is None or r["Kernel_Name"].startswith(main)):
                kernel = r["Kernel_Name"].split("(")[0]
                counters[r["Counter_Name"]].append(float(r["Counter_Value"]))
                summary.setdefault("dispatch", {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size",
                                                                    "Scratch_Size", "VGPR_Count", "SGPR_Count")})
summary["kernel"] = kernel
summary["counters_mean_per_dispatch"] = {k: sum(v) / len(v) for k, v in counters.items()}
c = summary["counters_mean_per_dispatch"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    summary["hbm_bytes_per_launch"] = {
        "fetch_raw": c["FETCH_SIZE"] * 1024, "write_raw": c["WRITE_SIZE"] * 1024,
        "fetch_corrected_x2": 2 * c["FETCH_SIZE"] * 1024,
        "total_corrected": 2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024,
        "note": "FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md (HBM); WRITE_SIZE exact"}
line = (src / "bench_line.json")
steps = None
if line.exists() and line.read_text().strip():
    b = json.loads(line.read_text())
    summary["bench"] = {k: b[k] for k in ("value", "ms_per_step", "roofline", "config")}
    summary["workload_tag"] = tag
    steps = int(b.get("steps", 0)) or None
# The profiler's own average (kernel_stats AverageNs) runs over EVERY dispatch of the kernel, including bench.py's untimed
# clock-ramp and warm-up launches, which are 10-25 % slower (the GPU leaves its idle clock after ~20 ms of load).  What the
# bench line times is the LAST `steps` dispatches: report those, and the median, so that the figures here and the driver's
# step time describe the same launches.
d = summary.get("dispatch_us") or []
if d:
    timed = d[-steps:] if steps and len(d) >= steps else d
    srt = sorted(d)
    summary["kernel_time_us"] = {
        "dispatches": len(d), "mean_all": round(sum(d) / len(d), 1), "median_all": srt[len(srt) // 2], "min": srt[0],
        "timed_dispatches": len(timed), "mean_timed": round(sum(timed) / len(timed), 1),
        "median_timed": sorted(timed)[len(timed) // 2],
        "note": "mean_timed = the last `steps` dispatches (the ones bench.py times); mean_all includes clock-ramp launches"}
    if "bench" in summary:
        alg = summary["bench"]["roofline"].get("algorithmic_bytes_per_launch")
        if alg:
            t = summary["kernel_time_us"]["mean_timed"] * 1e-6
            summary["kernel_time_us"]["roofline_frac_from_profile"] = round(alg / t / 1e9 / summary["bench"]["roofline"]["peak"], 4)
# VALU issue: the tile kernels are persistent (every wave lives as long as the launch), SQ_WAVE_CYCLES counts in units of four
# shader cycles, and one SIMD issues one wave-instruction per 2.5 cycles at best.
c = summary.get("counters_mean_per_dispatch") or {}
if "bench" in summary and all(k in c for k in ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAVES")) and c["SQ_WAVES"] > 0:
    px = summary["bench"]["value"] * 1e6 * summary["bench"]["ms_per_step"] * 1e-3
    cycles = 4.0 * c["SQ_WAVE_CYCLES"] / c["SQ_WAVES"]
    # GRBM_GUI_ACTIVE (summed over the 8 XCDs) is the count the 2.5-cycle issue rate was calibrated in (tools/ubench/cycles.sh): use it
    # when the run has it.  (SQ_WAVE_CYCLES x 4 / SQ_WAVES agrees with it at the product's 4 waves per SIMD, not at the 8 of the micro-benchmarks.)
    cycles_sq = cycles
    if "GRBM_GUI_ACTIVE" in c:
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0
    simds = 1024
    summary["valu_issue"] = {
        "valu_insts_per_px": round(c["SQ_INSTS_VALU"] * 64 / px, 1),
        "lds_insts_per_px": round(c.get("SQ_INSTS_LDS", 0) * 64 / px, 2),
        "shader_cycles_per_launch": round(cycles),
        "cycles_source": "GRBM_GUI_ACTIVE / 8 XCDs" if "GRBM_GUI_ACTIVE" in c else "SQ_WAVE_CYCLES x 4 / SQ_WAVES",
        "wave_lifetime_cycles_sq": round(cycles_sq),
        "shader_clock_GHz": round(cycles / (summary["kernel_time_us"]["mean_all"] * 1e3), 2) if "kernel_time_us" in summary else None,
        "cycles_per_valu_inst_at_best": 2.5,
        "frac_of_issue_limit": round(c["SQ_INSTS_VALU"] * 2.5 / (simds * cycles), 3),
        "note": "SQ_INSTS_VALU x 2.5 cycles / (1024 SIMDs x wave lifetime): how close the launch is to one VALU instruction per 2.5 "
                "cycles and SIMD, the best the unit does with any instruction mix (tools/ubench/op_rates2.hip)"}
(dst / f"{rnd}_{tag}_counters.json").write_text(json.dumps(summary, indent=1) + "\n")
if "hbm_bytes_per_launch" in summary and len(sys.argv) > 3:
    tpath = dst / "traffic.json"
    t = json.loads(tpath.read_text()) if tpath.exists() else {}
    t[sys.argv[3]] = int(summary["hbm_bytes_per_launch"]["total_corrected"])
    tpath.write_text(json.dumps(t, indent=1) + "\n")
    if "valu_issue" in summary:
        ipath = dst / "issue.json"
        t = json.loads(ipath.read_text()) if ipath.exists() else {}
        t[sys.argv[3]] = {k: summary["valu_issue"][k] for k in ("valu_insts_per_px", "lds_insts_per_px", "shader_clock_GHz", "frac_of_issue_limit")}
        t[sys.argv[3]]["source"] = f"profiles/{rnd}_{tag}_counters.json"
        ipath.write_text(json.dumps(t, indent=1) + "\n")
print(json.dumps(summary, indent=1)[:3000])
