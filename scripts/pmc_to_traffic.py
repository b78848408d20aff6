#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes of scripts/pmc_bench.sh into the entry of profiles/traffic.json that bench.py
prices its ceilings with: per-launch counters of the traversal kernel on the workload's bounce batch, tagged with
the hash of the kernel source they were taken with.

    scripts/pmc_to_traffic.py gpurun_out/pmc_<tag> <workload> [summary.txt]

summary.txt, if named, receives every counter's mean per averaged launch as text (the files profiles/*_pmc_summary.txt).

The closest_hit launches of a bench run are: one over the camera rays (first), one counting launch, then the
bounce-batch launches that are averaged here. bytes beyond L2 = (FETCH_SIZE x 1.20 + WRITE_SIZE) x 1024: the
counters are in KiB, and FETCH_SIZE under-reports random 64-byte gathers by the factor calibrated in
profiles/r01b_fetch_size_calibration.txt."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402


def per_launch(pmc_dir):
    out = {}
    for path in glob.glob(os.path.join(pmc_dir, "*", "*counter_collection.csv")) + glob.glob(os.path.join(pmc_dir, "*", "*", "*counter_collection.csv")):
        rows = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"]
                if "trace_kernel<0, false, false" not in name:  # closest_hit, no counting, LDS-only stack
                    continue
                key = (int(row["Dispatch_Id"]), row["Counter_Name"])
                rows[key] = rows.get(key, 0.0) + float(row["Counter_Value"])
                rows[(int(row["Dispatch_Id"]), "_ns")] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
        dispatches = sorted({d for d, _ in rows})
        for d in dispatches[1:]:  # the first one traces the camera rays
            for (dd, counter), v in rows.items():
                if dd == d:
                    out.setdefault(counter if counter != "_ns" else "_ns:" + os.path.basename(os.path.dirname(path)), []).append(v)
    return {k: sum(v) / len(v) for k, v in out.items()}


def main():
    pmc_dir, workload = sys.argv[1], sys.argv[2]
    c = per_launch(pmc_dir)
    ns = [v for k, v in c.items() if k.startswith("_ns:")]
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as f:
            f.write("rtgi_dev::trace_kernel<0, false, false>, mean per launch over the bounce-batch launches of each pass\n")
            for k in sorted(c):
                if k.startswith("_ns:"):
                    f.write(" %-44s %.6g\n" % ("launch duration in pass %s [ns]" % k[4:], c[k]))
            for k in sorted(c):
                if not k.startswith("_ns:"):
                    f.write(" %-44s %.6g\n" % (k, c[k]))
    cus, simds, xcds = 256, 1024, 8

    def per(name, unit_count, elapsed_pass):
        """Counter summed over its units / (units x cycles of the launch in the pass that collected it)."""
        if name not in c:
            return None
        cycles = c["_ns:" + elapsed_pass] * clock_ghz  # ns x GHz
        return round(c[name] / unit_count / cycles, 4)

    clock_ghz = c["GRBM_GUI_ACTIVE"] / xcds / c["_ns:tcp"]  # GRBM_GUI_ACTIVE sums the 8 XCDs; timestamps are in ns
    entry = {
        "kernel_source": kernel_source_hash(),
        "bytes_beyond_l2": int((c["FETCH_SIZE"] * 1.20 + c["WRITE_SIZE"]) * 1024),
        "l2_misses": int(c["TCC_MISS_sum"]),
        "l2_hits": int(c["TCC_HIT_sum"]),
        "tcp_accesses": int(c["TCP_TOTAL_CACHE_ACCESSES_sum"]),
        "tcp_to_l2_read_requests": int(c["TCP_TCC_READ_REQ_sum"]),
        "valu_instructions": int(c["SQ_INSTS_VALU"]),
        "salu_instructions": int(c["SQ_INSTS_SALU"]),
        "vmem_read_instructions": int(c["SQ_INSTS_VMEM_RD"]),
        "lds_instructions": int(c["SQ_INSTS_LDS"]),
        "clock_ghz": round(clock_ghz, 3),
        "kernel_ms_under_profiler": round(sum(ns) / len(ns) * 1e-6, 4),
        # busy and stall shares: cycles counted by a unit, summed over the units of that kind, over units x elapsed
        # cycles of the launch in the same pass (256 TA / TCP / TD, one each per CU; 1024 SIMDs; SQ_* activity
        # counters tick once per 4 cycles)
        "busy": {k: v for k, v in {
            "ta_busy": per("TA_TA_BUSY_sum", cus, "ta1"),
            "ta_addr_stalled_by_tcp": per("TA_ADDR_STALLED_BY_TC_CYCLES_sum", cus, "ta1"),
            "ta_data_stalled_by_tcp": per("TA_DATA_STALLED_BY_TC_CYCLES_sum", cus, "ta2"),
            "ta_buffer_instruction_cycles": per("TA_BUFFER_TOTAL_CYCLES_sum", cus, "ta3"),
            "td_busy": per("TD_TD_BUSY_sum", cus, "td1"),
            "td_stalled_by_tcp": per("TD_TC_STALL_sum", cus, "td1"),
            "tcp_clock_on": per("TCP_GATE_EN1_sum", cus, "tcp1"),
            "tcp_pending_stall": per("TCP_PENDING_STALL_CYCLES_sum", cus, "tcp1"),
            "tcp_read_tagconflict_stall": per("TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", cus, "tcp2"),
            "tcp_tag_lookups_per_cycle": per("TCP_TOTAL_CACHE_ACCESSES_sum", cus, "tcp"),
            "valu_busy": (lambda v: None if v is None else round(4 * v, 4))(per("SQ_ACTIVE_INST_VALU", simds, "sq1")),
            "salu_busy": (lambda v: None if v is None else round(4 * v, 4))(per("SQ_ACTIVE_INST_SCA", simds, "sq3")),
        }.items() if v is not None},
        "ta_buffer_wavefronts": int(c.get("TA_BUFFER_WAVEFRONTS_sum", 0)),
        "wave_cycles_waiting_for_anything": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4) if "SQ_WAIT_ANY" in c else None,
        "wave_cycles_waiting_for_issue": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4) if "SQ_WAIT_INST_ANY" in c else None,
        "wave_cycles_executing": round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4) if "SQ_ACTIVE_INST_ANY" in c else None,
        "l2_request_latency_cycles": round(c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"], 1) if "TCP_TCC_READ_REQ_LATENCY_sum" in c else None,
        "source": pmc_dir,
    }
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            data = json.load(f)
    except (OSError, ValueError):
        data = {}
    data = {k: v for k, v in data.items() if isinstance(v, dict)}
    data[workload] = entry
    data["_note"] = ("per launch of trace_kernel<closest_hit> on the workload's bounce batch, separate rocprofv3 --pmc passes "
                     "(scripts/pmc_bench.sh, scripts/pmc_to_traffic.py); bytes_beyond_l2 = (FETCH_SIZE x 1.20 + WRITE_SIZE) x 1024, "
                     "1.20 = calibration of FETCH_SIZE on random 64-byte gathers (profiles/r01b_fetch_size_calibration.txt); "
                     "Infinity Cache hits are included in it; busy.*: counter / (units x elapsed cycles of the launch in the "
                     "pass that collected it), elapsed cycles = launch duration x clock_ghz")
    data = {k: data[k] for k in sorted(data, key=lambda k: (k.startswith("_"), k))}
    with open(path, "w") as f:
        json.dump(data, f, indent=1)
        f.write("\n")
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()
