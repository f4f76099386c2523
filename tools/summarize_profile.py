#!/usr/bin/env python3
"""Condenses a tools/profile.sh output directory into profiles/<tag>_kernel_stats.csv and
profiles/<tag>_pmc_summary.json, and refreshes profiles/pmc_traffic.json for the workload.
   python tools/summarize_profile.py gpurun_out/prof_r01c r01c batched255 <units_per_launch> <ntaps> [bytes_per_unit flops_per_unit launches_per_step kernel_name_part]
(a unit is an output sample, 16 bytes and 2*ntaps flops, unless the two optional figures say otherwise:
the wavelet bank counts input samples, 528 bytes and 2*34385 flops each; several kernels matching
"fir_f64" in one step, as in the bank, are added up per step)"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

src, tag, workload, outputs, ntaps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
bytes_per_unit = float(sys.argv[6]) if len(sys.argv) > 6 else 16.0
flops_per_unit = float(sys.argv[7]) if len(sys.argv) > 7 else 2.0 * ntaps
launches_per_step = int(sys.argv[8]) if len(sys.argv) > 8 else 1
pattern = sys.argv[9] if len(sys.argv) > 9 else "fir_f64"   # kernels of the step
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)
shutil.copy(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0], os.path.join(prof, f"{tag}_kernel_stats.csv"))
out = {"command": "tools/profile.sh: rocprofv3 --kernel-trace --stats, then one --pmc pass per counter group, each around "
                  "`python3 bench.py --steps 5|2 --warmup 1 --no-cpu-baseline`",
       "workload": workload, "per_launch": {}, "kernel": None}
for name in ("sq1", "sq2", "fetch", "write", "grbm", "mfma"):
    found = glob.glob(os.path.join(src, name, "*", "*_counter_collection.csv"))
    if not found:
        continue   # older profile directories have no matrix-pipe pass
    f = found[0]
    agg, disp, spans = collections.defaultdict(float), set(), {}
    for r in csv.DictReader(open(f)):
        if pattern in r["Kernel_Name"]:
            out["kernel"] = re.search(pattern + r"\w*(<[^>]*>)?", r["Kernel_Name"]).group(0)
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            disp.add(r["Dispatch_Id"])
            spans[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if name == "grbm":
        out["grbm_pass_avg_launch_ns"] = sum(spans.values()) / len(spans) * launches_per_step
    for c, v in agg.items():
        out["per_launch"][c] = v / len(disp) * launches_per_step   # per step when a step is several launches
pl = out["per_launch"]
out["derived"] = {
    "hbm_read_bytes_per_launch": pl["FETCH_SIZE"] * 1024 * 2,
    "hbm_write_bytes_per_launch": pl["WRITE_SIZE"] * 1024,
    "note_fetch": "FETCH_SIZE is in KB and counts 64 B per 128-B request for 16-byte-per-lane streaming reads on gfx950: doubled",
    "algorithmic_bytes_per_launch": bytes_per_unit * outputs,
    "algorithmic_flops_per_launch": flops_per_unit * outputs,
    "lds_conflict_fraction": pl["SQ_LDS_BANK_CONFLICT"] / max(pl["SQ_LDS_IDX_ACTIVE"], 1),
    "clock_ghz_from_grbm": pl["GRBM_GUI_ACTIVE"] / 8 / out["grbm_pass_avg_launch_ns"],
    "note_clock": "GRBM_GUI_ACTIVE sums the 8 XCDs; divided by 8 and by the launch duration in the same pass",
}
if "SQ_VALU_MFMA_BUSY_CYCLES" in pl:
    simds = 1024.0   # 256 CUs x 4
    clock_hz = out["derived"]["clock_ghz_from_grbm"] * 1e9
    out["derived"]["mfma_busy_fraction"] = pl["SQ_VALU_MFMA_BUSY_CYCLES"] / (simds * clock_hz * out["grbm_pass_avg_launch_ns"] * 1e-9)
    out["derived"]["mfma_flops_issued_per_launch"] = pl["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0
    out["derived"]["note_mfma"] = ("SQ_VALU_MFMA_BUSY_CYCLES counts the cycles a SIMD's matrix pipe is occupied (16 per "
                                   "v_mfma_f64_4x4x4_4b_f64); the fraction is that over SIMDs x clock x launch duration. "
                                   "MOPS_F64 x 512 = flops issued, the band's zero columns included")
json.dump(out, open(os.path.join(prof, f"{tag}_pmc_summary.json"), "w"), indent=1)
tpath = os.path.join(prof, "pmc_traffic.json")
tr = json.load(open(tpath)) if os.path.exists(tpath) else {}
# the figure belongs to the kernel source it was profiled with: bench.py reports it only while that source is unchanged
import hashlib
ksrc = os.path.join(root, "dsp_amd", "csrc", "iir_kernels.hip" if workload == "iir_biquad" else "fir_kernels.hip")
tr[workload] = {"hbm_bytes_per_launch": out["derived"]["hbm_read_bytes_per_launch"] + out["derived"]["hbm_write_bytes_per_launch"],
                "source": f"profiles/{tag}_pmc_summary.json (FETCH_SIZE x2 + WRITE_SIZE, KB -> bytes)",
                "kernel_source_sha256": hashlib.sha256(open(ksrc, "rb").read()).hexdigest()}
json.dump(tr, open(tpath, "w"), indent=1)
print(open(os.path.join(prof, f"{tag}_kernel_stats.csv")).read()[:700])
print(json.dumps(out["derived"], indent=1))
