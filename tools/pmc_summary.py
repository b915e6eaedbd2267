#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (counter_collection + kernel_trace CSVs) per conv launch of the last
step: effective clock (GRBM_GUI_ACTIVE/8/duration), MFMA pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES over
1024 SIMDs x clock x duration), HBM bytes (2*FETCH_SIZE + WRITE_SIZE, KB -> bytes; the x2 is the gfx950
FETCH_SIZE correction of MI355X_MICROARCH.md), L2 hit rate.  usage: pmc_summary.py <dir> [prefixes...]"""
import collections
import csv
import sys


def load(d, tag):
    rows = collections.OrderedDict()
    for x in csv.DictReader(open("%s/%s_counter_collection.csv" % (d, tag))):
        k = int(x["Dispatch_Id"])
        rows.setdefault(k, {"name": x["Kernel_Name"]})[x["Counter_Name"]] = float(x["Counter_Value"])
    for x in csv.DictReader(open("%s/%s_kernel_trace.csv" % (d, tag))):
        k = int(x["Dispatch_Id"])
        if k in rows:
            rows[k]["us"] = (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3
    ids = [k for k in rows if rows[k]["name"].startswith("nchw_to_nhwc")]
    return [rows[k] for k in rows if k >= ids[-1] and "conv" in rows[k]["name"]]


def main():
    d = sys.argv[1]
    p1, p2, p3 = load(d, "p1"), load(d, "p2"), load(d, "p3")
    print("%-36s %8s %6s %6s %9s %9s %6s" % ("kernel", "us", "GHz", "mfma%", "fetchMB", "writeMB", "L2hit"))
    for a, b, c in zip(p1, p2, p3):
        t = a["us"] * 1e-6
        ghz = a.get("GRBM_GUI_ACTIVE", 0) / 8 / t / 1e9
        util = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (ghz * 1e9 * t) if ghz else 0
        fetch = 2 * b.get("FETCH_SIZE", 0) * 1024 / 1e6
        write = c.get("WRITE_SIZE", 0) * 1024 / 1e6
        hit, miss = b.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
        name = a["name"].replace("void ", "").replace("conv_mfma_kernel", "mfma").replace("(ConvK)", "")
        print("%-36s %8.1f %6.2f %6.1f %9.1f %9.1f %6.2f" % (name[:36], a["us"], ghz, 100 * util, fetch, write,
                                                            hit / (hit + miss) if hit + miss else 0))


if __name__ == "__main__":
    main()
