"""Aggregate rocprofv3 counter_collection CSVs (one per --pmc pass) into a per-kernel table: mean counter value
per dispatch.  FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; on gfx950 FETCH_SIZE counts 128-B requests at
64 B, i.e. reads exactly half of a wide coalesced stream (MI355X_MICROARCH.md, HBM section), so the summary adds
hbm_read_bytes = 2 * FETCH_SIZE * 1024 (the guide's gfx950 correction) and hbm_write_bytes = WRITE_SIZE * 1024
(uncalibrated, per the guide)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    name = name.replace("void slr::", "").replace("slr::", "")
    return name.strip()


def main(out_dir, summary_csv):
    acc = defaultdict(lambda: defaultdict(list))
    meta = {}
    for path in sorted(glob.glob(os.path.join(out_dir, "pmc*_counters.csv"))):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row.get("Kernel_Name", "?"))
                if "slr" not in row.get("Kernel_Name", "") and not k.startswith(("mf_", "gray_", "ge_", "remap", "ray_", "pc_")):
                    continue
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                meta[k] = (row.get("VGPR_Count", ""), row.get("LDS_Block_Size", ""), row.get("Grid_Size", ""))
    counters = sorted({c for k in acc for c in acc[k]})
    with open(summary_csv, "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches", "vgpr", "lds", "grid"] + counters + ["hbm_read_bytes(2xFETCH)", "hbm_write_bytes"])
        for k in sorted(acc):
            n = max(len(v) for v in acc[k].values())
            means = {c: (sum(acc[k][c]) / len(acc[k][c]) if acc[k][c] else float("nan")) for c in counters}
            rd = 2 * means.get("FETCH_SIZE", float("nan")) * 1024
            wr = means.get("WRITE_SIZE", float("nan")) * 1024
            w.writerow([k, n] + list(meta[k]) + ["%.6g" % means[c] for c in counters] + ["%.6g" % rd, "%.6g" % wr])
            print(k, "dispatches=%d" % n, "vgpr/lds/grid=%s" % (meta[k],))
            for c in counters:
                print("    %-32s %.6g" % (c, means[c]))
            print("    %-32s %.6g" % ("hbm_read_bytes (2 x FETCH_SIZE KiB)", rd))
            print("    %-32s %.6g" % ("hbm_write_bytes (WRITE_SIZE KiB)", wr))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
