#!/usr/bin/env python3
"""profiles/r<N>_pmc_bench.json from one tools/profile_round.sh directory: per-launch HBM traffic of the dominant scan kernel
(2 x FETCH_SIZE + WRITE_SIZE: gfx950 reports half of the fetched bytes of wide streaming reads, MI355X_MICROARCH.md, HBM) and
the `binding` block bench.py prints - what actually bounds the kernel, from the SQ counters and the kernel trace."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_hash  # noqa: E402   (the hash bench.py checks before it quotes this file)

root = sys.argv[1]
pm = json.load(open(os.path.join(root, "pmc_summary.json")))
line = json.loads(open(os.path.join(root, "bench_line_under_rocprof.json")).read().strip().splitlines()[-1])
n_scans_pmc = 3                                      # scans in each PMC run of tools/profile_round.sh (bench.py --steps 2 --warmup 1)
scan_keys = [k for k in pm if "scan8_kernel" in k or k.startswith("void scan_kernel")]
main = max(scan_keys, key=lambda k: pm[k].get("SQ_INSTS_VALU", {}).get("mean", 0.0) * pm[k].get("SQ_INSTS_VALU", {}).get("launches", 0))
# one scan = every launch of the scan kernels between two steps (sample + bulk of scan8_kernel, and the - empty - hand-over
# launches): counters are summed over the launches of one scan, not averaged per launch
c = {}
for k in scan_keys:
    for n, v in pm[k].items():
        c[n] = c.get(n, 0.0) + v["mean"] * v["launches"] / n_scans_pmc
scan = main
kt_scans = 4                                         # the kernel-trace run: bench.py --steps 3 --warmup 1
tot_ns, names = 0.0, []
for row in csv.DictReader(open(os.path.join(root, "kernel_stats.csv"))):
    if "scan8_kernel" in row["Name"] or row["Name"].startswith("void scan_kernel") or "finish_rows_kernel" in row["Name"]:
        tot_ns += float(row["TotalDurationNs"])
        names.append("%s x%s avg %.3f ms" % (row["Name"].split("(")[0], row["Calls"], float(row["AverageNs"]) * 1e-6))
# wall time of the scan launches of one step: the union of their intervals in the kernel trace (a long scan runs its last
# sixteenth as a second launch on another stream: intervals of different launches may touch or overlap)
iv = []
ktp = os.path.join(root, "kernel_trace.csv")
if os.path.exists(ktp):
    for row in csv.DictReader(open(ktp)):
        nm = row["Kernel_Name"]
        if "scan8_kernel" in nm or nm.startswith("void scan_kernel") or "finish_rows_kernel" in nm:
            iv.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
if iv:
    iv.sort()
    tot_ns, (a0, b0) = 0.0, iv[0]
    for a1, b1 in iv[1:]:
        if a1 <= b0: b0 = max(b0, b1)
        else: tot_ns += b0 - a0; a0, b0 = a1, b1
    tot_ns += b0 - a0
dur_ns = (tot_ns / kt_scans,)
cfg = line["config"]
n_win = cfg["candidate_windows_per_gpu"]
# SQ_* cycle counters are in quad-cycles summed over waves; 256 CUs x 4 SIMDs
simds = 256 * 4
t_cycles = dur_ns[0] * 1e-9 * 2.4e9
valu_busy = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (simds * t_cycles) if "SQ_ACTIVE_INST_VALU" in c else None
f64_per_pos = 28.0          # FP64 VALU instructions per scored position in the ISA of scan8_kernel<256,20,*>
pos_per_win = 4993.0
fp64_lane_ops = n_win * pos_per_win * f64_per_pos / (dur_ns[0] * 1e-9)      # one lane scores one position
out = {
    "kernel": scan.split("(")[0].strip(),
    "csrc_sha256": csrc_hash(),            # frisk_amd/csrc/* + include/*.h these counters were taken from
    "workload_bases_per_gpu": cfg["bases_per_gpu"], "candidate_windows_per_gpu": n_win,
    "scan_ms_per_step_rocprofv3": dur_ns[0] * 1e-6, "scan_launches": names,
    "FETCH_SIZE_KB_per_launch": c.get("FETCH_SIZE"), "WRITE_SIZE_KB_per_launch": c.get("WRITE_SIZE"),
    "hbm_bytes_per_launch": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 if "FETCH_SIZE" in c and "WRITE_SIZE" in c else None,
    "hbm_bytes_note": "per scan = all launches of the scan kernels in one step; 2 x FETCH_SIZE + WRITE_SIZE, KB = 1024 B",
    "binding": {
        "source": "rocprofv3 --pmc passes (SQ counters summed over waves, quad-cycles) and --kernel-trace of bench.py, offline",
        "valu_instr_per_window": c["SQ_INSTS_VALU"] / n_win,
        "salu_instr_per_window": c["SQ_INSTS_SALU"] / n_win,
        "valu_busy_frac_of_simd_cycles": valu_busy,
        "wave_issue_active_frac": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
        "wave_wait_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
        "wave_issue_stall_frac": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
        "lds_bank_conflict_frac_of_lds_cycles": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"],
        "lds_active_quadcycles_per_window": c["SQ_LDS_IDX_ACTIVE"] / n_win,
        # scratch traffic, by counter: vector-memory instructions per window against what the hot path's ISA holds per window (four
        # waves: ~8 vectorised sequence loads in stage 1 + 20 ring loads each; stores: the parking wave's 20, a fresh window's 80 one
        # time in 16, the row's 6) - spilled registers reloaded at run time would come on top of these
        "vmem_rd_instr_per_window": c["SQ_INSTS_VMEM_RD"] / n_win if "SQ_INSTS_VMEM_RD" in c else None,
        "vmem_wr_instr_per_window": c["SQ_INSTS_VMEM_WR"] / n_win if "SQ_INSTS_VMEM_WR" in c else None,
        "vmem_instr_expected_from_hot_isa_per_window": {"rd": 4 * (8 + 20), "wr": 20 + 80 / 16.0 + 6},
        "fp64_lane_ops_per_s": fp64_lane_ops,
        "fp64_lane_ops_peak_per_s": 78.6e12 / 2.0,       # 78.6 TFLOP/s vector FP64 = 39.3 T fused multiply-add lanes per second
        "fp64_frac_of_vector_peak": fp64_lane_ops / (78.6e12 / 2.0),
        "note": "between VALU issue and latency: the SIMDs' VALU is busy most of the time, the waves of a window are short dependent "
                "stages between four barriers, three workgroups per CU; LDS and HBM are far from their limits",
    },
}
json.dump(out, sys.stdout, indent=1)
