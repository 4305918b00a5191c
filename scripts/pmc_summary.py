"""Summarise rocprofv3 --pmc passes (one directory per counter) into profiles/rNN_pmc_hbm_traffic.json: per kernel, average
FETCH_SIZE / WRITE_SIZE per launch and the HBM bytes per launch corrected as MI355X_MICROARCH.md prescribes for gfx950
(FETCH_SIZE tallies 128-byte requests at 64 bytes: hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024)."""
import time, csv, glob, json, os, re, sys


def read_counter(d, counter):
    agg = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                k = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "").strip()
                a = agg.setdefault(k, [0.0, 0])
                a[0] += float(row["Counter_Value"]); a[1] += 1
    return agg


def main():
    fetch_dir, write_dir, out, command = sys.argv[1:5]
    fe, wr = read_counter(fetch_dir, "FETCH_SIZE"), read_counter(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        f = fe.get(k, [0.0, 0]); w = wr.get(k, [0.0, 0])
        fk = f[0] / max(1, f[1]); wk = w[0] / max(1, w[1])
        kernels[k] = {"FETCH_SIZE_KB_per_launch": round(fk, 2), "launches_FETCH_SIZE": f[1], "WRITE_SIZE_KB_per_launch": round(wk, 2), "launches_WRITE_SIZE": w[1],
                      "hbm_bytes_per_launch_corrected": int((2 * fk + wk) * 1024)}
    json.dump({"command": command, "date": time.strftime("%Y-%m-%d"), "units": "KB per launch as reported by rocprofv3 (TCC_EA request counters x 64 B / 1024)",
               "gfx950_note": "MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of a wide coalesced read on gfx950 -> hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; "
                              "Infinity-Cache hits are counted too (the counters sit on the L2's fabric side)", "kernels": kernels}, open(out, "w"), indent=1)
    print("wrote", out, len(kernels), "kernels")


if __name__ == "__main__":
    main()
