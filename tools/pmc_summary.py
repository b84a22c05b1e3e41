#!/usr/bin/env python3
"""Summarise two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/pmc_traffic.json.

usage: pmc_summary.py <fetch_dir> <write_dir> <workload: c2|c3> [out.json]

Each directory holds the `*_counter_collection.csv` of one pass of `python bench.py --steps 1 --warmup 1
--no-cpu-baseline --no-seek`.  Values are KiB per dispatch (MI355X_MICROARCH.md, HBM section); the LAST dispatch of
every kernel is used (the serialised profiling step of the decode / the encode of the archive).  FETCH_SIZE is doubled only for kernels
whose reads are known to be 16-byte wide streams (the gfx950 "half of wide reads" behaviour, calibrated on
zk_k_xxh64, which reads exactly the archive's decompressed bytes).
"""
import csv
import glob
import json
import os
import sys

WIDE = {"zk_k_xxh64": 2.0, "zk_k_xxh64_wide": 2.0, "zk_k_xxh64_fed": 2.0}     # all read exactly the decompressed bytes, 8 / 16 bytes per lane and load


def short(name):
    n = name.split("(")[0]
    if n.startswith("void "):
        n = n[5:]
    n = n.split("<")[0]
    return {"zk_k_enc_match2": "zk_k_enc_match"}.get(n, n)      # (r5) the fast setting's match kernel reports under the engine's timer name


def last_per_kernel(d, counter):
    out = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                k = short(row["Kernel_Name"])
                if not k.startswith("zk_"):
                    continue
                # several rows per dispatch (one per XCD / SE): sum them per dispatch id
                did = int(row["Dispatch_Id"])
                slot = out.setdefault(k, {})
                slot[did] = slot.get(did, 0.0) + float(row["Counter_Value"])
    return {k: v[max(v)] for k, v in out.items()}, {k: max(v.values()) for k, v in out.items()}


def main():
    fdir, wdir, workload = sys.argv[1:4]
    if workload not in ("c2", "c3"):
        sys.exit("workload must be the bench.py --workload key (c2 / c3): bench.py matches it against its own flag")
    outp = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(__file__), "..", "profiles",
                                                              "pmc_traffic.json")
    # LAST dispatch: bench.py ends with its per-kernel-timing steps, which run every kernel alone on the device; in the
    # earlier steps huf and fse overlap and the (device-wide) L2 counters of one dispatch include the other's traffic
    fetch, _ = last_per_kernel(fdir, "FETCH_SIZE")
    write, _ = last_per_kernel(wdir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        corr = WIDE.get(k, 1.0)
        fk, wk = fetch.get(k, 0.0), write.get(k, 0.0)
        kernels[k] = {"fetch_kib": fk, "write_kib": wk, "fetch_correction": corr,
                      "hbm_bytes": int((fk * corr + wk) * 1024)}
    import datetime
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from bench import kernel_source_hashes         # what the counters were taken ON: bench.py nulls `traffic` once a kernel's sources differ
    doc = {"workload": workload, "source_hashes": kernel_source_hashes(), "collected": datetime.date.today().isoformat() + " from " + os.path.basename(os.path.normpath(fdir)) + " / " + os.path.basename(os.path.normpath(wdir)),
           "note": "per launch (last dispatch of each kernel = the serialised per-kernel-timing step of bench.py); FETCH_SIZE x2 only where the access pattern was "
                   "calibrated as wide (zk_k_xxh64, zk_k_xxh64_wide); others uncorrected lower bounds",
           "kernels": kernels}
    # (r5) a fifth argument names a SECTION: the kernels go under that key of an existing file instead of replacing it (the reference-made legs:
    # reference_made_level_1 / reference_made_level_3, what bench.py's roofline objects of those legs look up)
    if len(sys.argv) > 5:
        with open(outp) as f:
            base = json.load(f)
        base[sys.argv[5]] = kernels
        base.setdefault("section_source_hashes", {})[sys.argv[5]] = kernel_source_hashes()
        base["collected"] = base.get("collected", "") + "; " + sys.argv[5] + ": " + doc["collected"]
        doc = base
    with open(outp, "w") as f:
        json.dump(doc, f, indent=1)
    for k, v in kernels.items():
        print(f"{k:24s} fetch {v['fetch_kib'] / 1048576:8.3f} GiB  write {v['write_kib'] / 1048576:8.3f} GiB"
              f"  hbm {v['hbm_bytes'] / 1e9:8.3f} GB")


if __name__ == "__main__":
    main()
