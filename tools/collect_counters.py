#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes of tools/profile_round.sh into the two records bench.py reads:
profiles/sq_counters.json[workload] (VALU work of the render kernel per segment) and
profiles/hbm_traffic.json[workload] (HBM bytes per launch).

usage: collect_counters.py <workload> <dir with pmc_sq_a.csv pmc_sq_b.csv pmc_fetch_size.csv
                                          pmc_write_size.csv bench_sq_a.json> [tag]

Units (calibrated with tools/ub_counters.hip, profiles/calibration_*.csv): SQ_INSTS_VALU counts one
per wave instruction; SQ_ACTIVE_INST_VALU one issue unit per wave instruction (two for v_rcp/v_sqrt
and the like); SQ_THREAD_CYCLES_VALU that unit times the active lanes; SQ_INSTS_VALU_FMA_F32 one per
v_fma_f32 / v_fmac_f32 / v_pk_fma_f32.  FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of 16-byte-per-lane reads, so it is doubled; WRITE_SIZE is taken as it is."""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_launch(path):
    sums = defaultdict(float)
    launches = set()
    kernel = ""
    with open(path) as f:
        for r in csv.DictReader(f):
            if "render_tiles" in r["Kernel_Name"]:
                sums[r["Counter_Name"]] += float(r["Counter_Value"])
                launches.add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
                kernel = r["Kernel_Name"]
    n = max(1, len(launches))
    return {k: v / n for k, v in sorted(sums.items())}, len(launches), kernel


def merge(path, key, rec):
    data = {}
    if os.path.exists(path):
        with open(path) as f:
            data = json.load(f)
    data[key] = rec
    with open(path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
        f.write("\n")


def main():
    workload, d = sys.argv[1], sys.argv[2]
    tag = sys.argv[3] if len(sys.argv) > 3 else ""
    with open(os.path.join(d, "bench_sq_a.json")) as f:
        bench = json.loads([l for l in f if l.startswith("{")][-1])
    seg = bench["config"]["segments_per_step"]
    sps = bench["config"]["spp_per_step"]
    a, n_a, kernel = per_launch(os.path.join(d, "pmc_sq_a.csv"))
    lanes = a["SQ_THREAD_CYCLES_VALU"] / a["SQ_ACTIVE_INST_VALU"]
    sps = -(-sps // bench["config"].get("launches_per_step", 1))
    seg = seg / bench["config"].get("launches_per_step", 1)
    build = bench["roofline"].get("library_build")
    rec = {
        "kernel": kernel, "library_build": build, "launches_averaged": n_a, "spp_per_launch": sps, "segments_per_launch": seg,
        "per_launch": dict(a),
        "active_lanes_per_valu_inst": round(lanes, 2),
        "valu_lane_ops_per_segment": round(a["SQ_INSTS_VALU"] * lanes / seg, 1),
        "valu_wave_insts_per_wave_segment": round(a["SQ_INSTS_VALU"] * 64.0 / seg, 1),
        "salu_insts_per_wave_segment": round(a.get("SQ_INSTS_SALU", 0.0) * 64.0 / seg, 1),
        "valu_issue_units_per_wave_cycle": round(a["SQ_ACTIVE_INST_VALU"] / a["SQ_WAVE_CYCLES"], 4),
        "source": "profiles/%s_%s_pmc_sq_a.csv, _pmc_sq_b.csv (rocprofv3 --pmc, `python3 bench.py --workload %s "
                  "--steps 2 --warmup 1 --no-cpu-baseline --no-flat-probe --others ''`)" % (tag, workload, workload),
    }
    pb = os.path.join(d, "pmc_sq_b.csv")
    if os.path.exists(pb):
        b, _, _ = per_launch(pb)
        rec["per_launch"].update({k: v for k, v in b.items() if k not in rec["per_launch"]})
        if "SQ_INSTS_VALU_FMA_F32" in b:
            rec["fma_share_of_valu_insts"] = round(b["SQ_INSTS_VALU_FMA_F32"] / b["SQ_INSTS_VALU"], 4)
            rec["fma_lane_ops_per_segment"] = round(b["SQ_INSTS_VALU_FMA_F32"] * lanes / seg, 1)
    merge(os.path.join(ROOT, "profiles", "sq_counters.json"), workload, rec)
    print(json.dumps({k: v for k, v in rec.items() if k != "per_launch"}))
    pf, pw = os.path.join(d, "pmc_fetch_size.csv"), os.path.join(d, "pmc_write_size.csv")
    if os.path.exists(pf) and os.path.exists(pw):
        f_, n_f, _ = per_launch(pf)
        w_, _, _ = per_launch(pw)
        hbm = {
            "library_build": build,
            "bytes_per_launch": int(2 * f_["FETCH_SIZE"] * 1024 + w_["WRITE_SIZE"] * 1024),
            "fetch_size_kib_raw": f_["FETCH_SIZE"], "write_size_kib_raw": w_["WRITE_SIZE"],
            "launches_averaged": n_f, "spp_per_launch": sps,
            "correction": "FETCH_SIZE x2 (gfx950 counts 16-byte-per-lane reads at half), WRITE_SIZE x1",
            "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, one pass each, same command as sq_counters.json",
        }
        merge(os.path.join(ROOT, "profiles", "hbm_traffic.json"), workload, hbm)
        print(json.dumps(hbm))


if __name__ == "__main__":
    main()
