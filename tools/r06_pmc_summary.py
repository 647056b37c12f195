#!/usr/bin/env python3
"""gpurun_out/r06/pmc_scan/pmc_summary.json (tools/r06_pmc_scan.sh) -> profiles/r06/scan_piece_pmc.json in the shape of the r05 file,
and the scan kernel's entry of profiles/traffic.json (what bench.py quotes as roofline.traffic / binds / issue)."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = json.load(open(os.path.join(ROOT, "gpurun_out", "r06", "pmc_scan", "pmc_summary.json")))
k, r = dict(src["scan_piece_kernel"]), dict(src["sp_refine_kernel"])
host = src["host"][0]
num = lambda name: int(re.search(name + r" (\d+)", host).group(1))
walked, skipped, slots = num("walked half paths"), num("rows skipped"), num("walked slots")
N, NNZ = 576289, 42512334
alg = 4 * walked + 24 * (NNZ - skipped) + 16 * N + 12 * slots
kms = k.pop("kernel_ms"); rms = r.pop("kernel_ms")
fetch, wr = k["FETCH_SIZE"] * 1024, k["WRITE_SIZE"] * 1024
total = 2 * fetch + wr
ms = sum(kms) / len(kms)
cyc = k["GRBM_GUI_ACTIVE"] / 8.0 if False else None
out = {
    "command": "rocprofv3 --kernel-trace --pmc <set> --output-format csv -- python3 tools/r06_scan_one.py  (one counter set per run: tools/r06_pmc_scan.sh)",
    "launch": "scan_piece_kernel<256, false, true, true> (the body compiled for the main launch's tables + the per-column pack): ONE launch over the "
              "live columns of the ppa-like graph at bar 2.8758 with skipped heads (beta 0.5, 16384 hub rows): " + host,
    "kernel_ms_under_pmc": kms,
    "counters": k,
    "refine_kernel": {"kernel_ms_under_pmc": rms, "counters": r},
    "fabric_traffic_bytes": {"FETCH_SIZE_bytes": fetch, "WRITE_SIZE_bytes": wr, "corrected_total": total,
                             "how": "as in r05 (profiles/r05/fetch_size_calibration.json): FETCH_SIZE on gfx950 tallies 128-byte requests at 64 B -> doubled"},
    "derived": {
        "kernel_ms_mean": ms,
        "algorithmic_bytes_walked": alg,
        "traffic_over_algorithmic_bytes": total / alg,
        "fabric_TBps": total / (ms * 1e-3) / 1e12,
        "fabric_frac_of_8TBps": total / (ms * 1e-3) / 8e12,
        "valu_busy_of_simd_time": k["SQ_ACTIVE_INST_VALU"] / 8.0 / k["SQ_BUSY_CYCLES"] if k.get("SQ_BUSY_CYCLES") else None,
        "wave_issue_share": k["SQ_ACTIVE_INST_ANY"] / k["SQ_WAVE_CYCLES"],
        "waves_waiting_share": k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"],
        "lds_bank_conflict_share_of_lds_cycles": k["SQ_LDS_BANK_CONFLICT"] / k["SQ_LDS_IDX_ACTIVE"],
        "l2_hit_rate": k["TCC_HIT_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"]),
        "valu_lane_instructions_per_walked_half_path": k["SQ_INSTS_VALU"] * 64 / walked,
        "vs_r05": "r05's launch (same paths, same pieces, generic body, row-record gathers): 9.52 ms under the counters, VALU 3.57 G, SALU 2.31 G "
                  "wave-instructions, fabric 33.03 GB = 1.85x the walked bytes (3.5 TB/s)",
    },
}
os.makedirs(os.path.join(ROOT, "profiles", "r06"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", "r06", "scan_piece_pmc.json"), "w"), indent=1)
tpath = os.path.join(ROOT, "profiles", "traffic.json")
t = json.load(open(tpath))
d = out["derived"]
t["scan_piece_kernel/ppa_like/576289"] = {
    "traffic": total, "source": "profiles/r06/scan_piece_pmc.json",
    "binds": "per-piece latency chains at 4 waves per SIMD (a wave issues %.0f %% of its cycles and waits %.0f %%); fabric traffic %.1f GB per launch = %.2fx "
             "the walked algorithmic bytes = %.1f TB/s, %.2f of the HBM peak (r05: 33.0 GB = 1.85x: the per-column pack took the 128-byte row-record "
             "line per walked row out of the set-up)" % (100 * d["wave_issue_share"], 100 * d["waves_waiting_share"], total / 1e9,
                                                          d["traffic_over_algorithmic_bytes"], d["fabric_TBps"], d["fabric_frac_of_8TBps"]),
    "issue": {"valu_busy_of_simd_time": d["valu_busy_of_simd_time"], "lds_bank_conflict_share": d["lds_bank_conflict_share_of_lds_cycles"],
              "valu_wave_instructions": k["SQ_INSTS_VALU"], "salu_wave_instructions": k["SQ_INSTS_SALU"], "lds_wave_instructions": k["SQ_INSTS_LDS"],
              "waves_waiting_share": d["waves_waiting_share"], "wave_issue_share": d["wave_issue_share"], "fabric_TBps": d["fabric_TBps"]},
}
json.dump(t, open(tpath, "w"), indent=1)
print(json.dumps(out["derived"], indent=1))
