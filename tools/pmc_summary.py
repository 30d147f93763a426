"""Per-kernel means of rocprofv3 counter passes (`--pmc … --kernel-trace --output-format csv`).

    python tools/pmc_summary.py OUT.json PASS_DIR [PASS_DIR …]

Each PASS_DIR holds one `*_counter_collection.csv`.  Kernels are grouped by (name, grid size); per group the
mean of every counter and the mean duration.  Derived, where the counters are present:
  mfma_util   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)
  clock_ghz   = GRBM_GUI_ACTIVE / 8 / duration
  hbm_read_gb = FETCH_SIZE KiB x 2 (gfx950 tallies its 128-byte read requests as 64 bytes), hbm_write_gb = WRITE_SIZE KiB
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    out_path, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = {}
            with open(path) as f:
                for r in csv.DictReader(f):
                    key = (r["Kernel_Name"], int(r["Grid_Size"]))
                    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    per_dispatch[(key, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            for (key, _), us in per_dispatch.items():
                acc[key]["_us"].append(us)
    rows = []
    for (name, grid), c in acc.items():
        m = {k: sum(v) / len(v) for k, v in c.items()}
        row = {"kernel": name, "grid": grid, "launches": len(c["_us"]), "avg_us_profiled": round(m["_us"], 2)}
        for k, v in m.items():
            if k != "_us":
                row[k] = round(v, 1)
        if "GRBM_GUI_ACTIVE" in m:
            cyc = m["GRBM_GUI_ACTIVE"] / 8.0
            row["clock_ghz"] = round(cyc / (m["_us"] * 1e3), 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                row["mfma_util"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4), 4)
        if "FETCH_SIZE" in m:
            row["hbm_read_gb"] = round(m["FETCH_SIZE"] * 1024 * 2 / 1e9, 4)
        if "WRITE_SIZE" in m:
            row["hbm_write_gb"] = round(m["WRITE_SIZE"] * 1024 / 1e9, 4)
        if "hbm_read_gb" in row and "hbm_write_gb" in row:
            row["hbm_tb_per_s"] = round((row["hbm_read_gb"] + row["hbm_write_gb"]) / m["_us"] * 1e3, 3)
        rows.append(row)
    rows = [r for r in rows if "fcenc" in r["kernel"]]          # the library's kernels; torch's fills and copies are set-up
    rows.sort(key=lambda r: -r["avg_us_profiled"] * r["launches"])
    with open(out_path, "w") as f:
        json.dump(rows, f, indent=1)
    for r in rows:
        print(r["kernel"][:70], {k: v for k, v in r.items() if k in ("grid", "avg_us_profiled", "mfma_util", "clock_ghz",
                                                                      "hbm_read_gb", "hbm_write_gb", "hbm_tb_per_s")})


if __name__ == "__main__":
    main()
