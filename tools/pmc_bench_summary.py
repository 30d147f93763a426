"""Builds profiles/r03_bench_small_bf16_pmc.json, the per-kernel figures bench.py quotes next to its own event
timings, from rocprofv3 passes over `python bench.py --no-cpu-baseline --no-extras`:

    python tools/pmc_bench_summary.py OUT.json FETCH_PASS_DIR WRITE_PASS_DIR KERNEL_STATS_DIR [MFMA_PASS_DIR]

MFMA_PASS_DIR: a `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass; per kernel, MFMA busy = busy cycles /
(GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) and the clock the chip held = GRBM_GUI_ACTIVE / 8 / duration.

FETCH_PASS_DIR / WRITE_PASS_DIR hold one `*_counter_collection.csv` each (`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`,
separate runs); KERNEL_STATS_DIR holds the `*_kernel_trace.csv` of a `--kernel-trace --stats` run (durations from a run
without counters).  HBM bytes per launch = FETCH_SIZE KiB x 1024 x 2 (gfx950 tallies its 128-byte read requests
as 64 bytes) + WRITE_SIZE KiB x 1024."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KEYS = [  # (key, substring of the kernel name, grid size or None)
    ("block", "convmod_fused_kernel<true, 4, true>", None),
    ("conv12", "conv12_kernel", None),
    ("conv2", "conv256_kernel", 1280000),
    ("conv3", "conv3p_kernel", None),
    ("conv3_gemm", "conv256_kernel", 320000),
    ("conv1", "conv1_mfma_kernel", None),
    ("stats", "input_stats_kernel", None),
    ("attention", "relpos_attn_kernel", None),
    ("linear", "Li3ELb0ELi8ELi4", None),
]


def counter_means(d, counter):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] == counter:
                    acc[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def duration_means(d):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
                acc[(r["Kernel_Name"], grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def pick(table, sub, grid):
    for (name, g), v in table.items():
        if sub in name and (grid is None or g == grid):
            return v
    return None


def main():
    out, fetch_dir, write_dir, stats_dir = sys.argv[1:5]
    fetch, write, dur = counter_means(fetch_dir, "FETCH_SIZE"), counter_means(write_dir, "WRITE_SIZE"), duration_means(stats_dir)
    mfma_dir = sys.argv[5] if len(sys.argv) > 5 else None
    busy = counter_means(mfma_dir, "SQ_VALU_MFMA_BUSY_CYCLES") if mfma_dir else {}
    active = counter_means(mfma_dir, "GRBM_GUI_ACTIVE") if mfma_dir else {}
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) and --kernel-trace over "
                     "`python bench.py --no-cpu-baseline --no-extras`; HBM bytes = FETCH_SIZE KiB x 2048 + WRITE_SIZE KiB x 1024"}
    for key, sub, grid in KEYS:
        f, w, d = pick(fetch, sub, grid), pick(write, sub, grid), pick(dur, sub, grid)
        if d is not None:
            res[key + "_avg_us"] = round(d[0], 2)
            res[key + "_launches_profiled"] = d[1]
        if f is not None and w is not None:
            res[key + "_hbm_read_bytes_per_launch"] = round(f * 2048)
            res[key + "_hbm_write_bytes_per_launch"] = round(w * 1024)
            res[key + "_hbm_bytes_per_launch"] = round(f * 2048 + w * 1024)
        bz, ac = pick(busy, sub, grid), pick(active, sub, grid)
        if bz is not None and ac:
            res[key + "_mfma_busy"] = round(bz / (ac / 8 * 1024), 4)
            if d is not None:
                res[key + "_clock_ghz"] = round(ac / 8 / (d[0] * 1e3), 3)
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
