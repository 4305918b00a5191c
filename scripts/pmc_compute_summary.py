"""Summarise rocprofv3 --pmc passes of SQ / GRBM counters (one directory per pass) into profiles/rNN_pmc_compute.json: per kernel, the
average of every counter per launch plus a few ratios that say what a kernel is bound by (MI355X_MICROARCH.md, rocprofv3 PMC slots:
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over the waves; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES)."""
import csv, glob, json, os, re, sys


def read_dir(d):
    agg = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "").strip()
                a = agg.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, 0])
                a[0] += float(row["Counter_Value"]); a[1] += 1
    return agg


def main():
    out, command = sys.argv[1], sys.argv[2]
    kernels = {}
    for d in sys.argv[3:]:
        for k, cs in read_dir(d).items():
            row = kernels.setdefault(k, {})
            for c, (tot, n) in cs.items():
                row[c] = round(tot / max(1, n), 1); row["launches"] = n
    for k, r in kernels.items():
        wc = r.get("SQ_WAVE_CYCLES", 0.0)
        if wc > 0:
            r["frac_wave_cycles_waiting_on_memory_or_barrier"] = round(r.get("SQ_WAIT_ANY", 0.0) / wc, 4)
            r["frac_wave_cycles_issue_stalled"] = round(r.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4)
            r["frac_wave_cycles_issuing"] = round(r.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4)
        if r.get("SQ_BUSY_CYCLES", 0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in r:
            r["mfma_busy_over_sq_busy"] = round(r["SQ_VALU_MFMA_BUSY_CYCLES"] / r["SQ_BUSY_CYCLES"], 6)
        if r.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
            r["lds_bank_conflict_frac"] = round(r.get("SQ_LDS_BANK_CONFLICT", 0.0) / r["SQ_LDS_IDX_ACTIVE"], 4)
        if r.get("SQ_WAVES", 0) > 0 and "SQ_INSTS_VALU" in r:
            r["valu_insts_per_wave"] = round(r["SQ_INSTS_VALU"] / r["SQ_WAVES"], 1)
    json.dump({"command": command, "note": "averages per launch; SQ_* cycle counters are quad-cycles summed over waves / SIMDs as rocprofv3 reports them; "
               "GRBM_GUI_ACTIVE is summed over the 8 XCDs", "kernels": kernels}, open(out, "w"), indent=1)
    print("wrote", out, len(kernels), "kernels")


if __name__ == "__main__":
    main()
