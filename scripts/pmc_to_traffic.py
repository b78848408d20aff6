#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes of scripts/pmc_bench.sh into the entry of profiles/traffic.json that bench.py
prices its ceilings with: per-launch counters of the traversal kernel on the workload's bounce batch, tagged with
the hash of the kernel source they were taken with.

    scripts/pmc_to_traffic.py gpurun_out/pmc_<tag> <workload>

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
        # GRBM_GUI_ACTIVE sums the 8 XCDs; timestamps are in ns
        "clock_ghz": round(c["GRBM_GUI_ACTIVE"] / 8.0 / c["_ns:tcp"], 3),
        "kernel_ms_under_profiler": round(sum(ns) / len(ns) * 1e-6, 4),
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
                     "Infinity Cache hits are included in it")
    data = {k: data[k] for k in sorted(data, key=lambda k: (k.startswith("_"), k))}
    with open(path, "w") as f:
        json.dump(data, f, indent=1)
        f.write("\n")
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()
