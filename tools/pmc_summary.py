#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (counter_collection + kernel_trace CSVs) per conv launch of the last
step: effective clock (GRBM_GUI_ACTIVE/8/duration), MFMA pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES over
1024 SIMDs x clock x duration), HBM bytes (2*FETCH_SIZE + WRITE_SIZE, KB -> bytes; the x2 is the gfx950
FETCH_SIZE correction of MI355X_MICROARCH.md), L2 hit rate.

usage: pmc_summary.py <dir> [--names layer_kernels.json] [--json out.json] [--workload text]
  <dir> holds p1_* (GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES), p2_* (FETCH_SIZE TCC_HIT_sum), p3_* (WRITE_SIZE
  TCC_MISS_sum), each from its own `rocprofv3 --kernel-trace --pmc ... -- python3 bench.py ...` run.
  --names: the ordered engine kernel names of one step's conv launches (bench.py with Y2_BENCH_DUMP_KERNELS=file);
  with it the rows are labelled with the names bench.py reports and --json writes the per-kernel HBM bytes per
  launch that bench.py's roofline.traffic reads (profiles/r*_pmc_traffic.json)."""
import collections
import csv
import json
import re
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
    # first launch of a step: the input transform where the plan has one, else the first-layer kernel (which reads the
    # NCHW input itself)
    ids = [k for k in rows if "nchw_to_nhwc" in rows[k]["name"]] or [k for k in rows if "conv_first" in rows[k]["name"]]
    if not ids:
        sys.exit("no step marker (input transform / first-layer kernel) among the profiled kernels")
    want_conv = not OTHER
    return [rows[k] for k in rows if k >= ids[-1] and (("conv" in rows[k]["name"]) == want_conv)]


OTHER = "--other" in sys.argv      # --other: the step's non-convolution kernels (layout, pools, region head, decode, NMS)


def main():
    args = sys.argv[1:]
    d = args[0]
    names = json.load(open(args[args.index("--names") + 1])) if "--names" in args else None
    out_json = args[args.index("--json") + 1] if "--json" in args else None
    workload = args[args.index("--workload") + 1] if "--workload" in args else ""
    wl_name = args[args.index("--wl-name") + 1] if "--wl-name" in args else "yolo608_b32"     # bench.py --workload key
    p1, p2, p3 = load(d, "p1"), load(d, "p2"), load(d, "p3")
    if names is not None and len(names) != len(p1) and not OTHER:
        # r3: a layer on the fp16 256x256 kernel may take a second launch -- the stream-K fix-up, or a small register-staged
        # tile over the tail rows (same filter size, directly behind it).  Those dispatches are folded into their layer's row.
        def ks_of(n):
            m = re.search(r"conv_mfma_f16_kernel<\d+, \d+, \d+, (\d+)", n) or re.search(r"conv_p8_f16_kernel<(\d+)>", n)
            return m.group(1) if m else None
        helper = []
        for i, r in enumerate(p1):
            n = r["name"]
            prev = p1[i - 1]["name"] if i else ""
            helper.append("conv_p8_fixup_kernel" in n or
                          ("conv_mfma_f16_kernel" in n and "conv_p8_f16_kernel" in prev and ks_of(n) == ks_of(prev)))
        def fold(rows):
            out = []
            for r, h in zip(rows, helper):
                if h and out:
                    for k, v in r.items():
                        if k != "name":
                            out[-1][k] = out[-1].get(k, 0) + v
                else:
                    out.append(dict(r))
            return out
        if len(p1) - sum(helper) == len(names) and len(p2) == len(p1) == len(p3):
            p1, p2, p3 = fold(p1), fold(p2), fold(p3)
    if names is not None and len(names) != len(p1):
        sys.exit("--names lists %d conv launches, the profile has %d" % (len(names), len(p1)))
    print("%-44s %8s %6s %6s %9s %9s %8s %6s" % ("kernel", "us", "GHz", "mfma%", "fetchMB", "writeMB", "HBM GB/s", "L2hit"))
    agg = collections.OrderedDict()
    for i, (a, b, c) in enumerate(zip(p1, p2, p3)):
        t = a["us"] * 1e-6
        ghz = a.get("GRBM_GUI_ACTIVE", 0) / 8 / t / 1e9
        util = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (ghz * 1e9 * t) if ghz else 0
        fetch = 2 * b.get("FETCH_SIZE", 0) * 1024
        write = c.get("WRITE_SIZE", 0) * 1024
        hit, miss = b.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
        name = names[i] if names else a["name"].replace("void ", "").replace("conv_mfma_kernel", "mfma").replace("(ConvK)", "")
        print("%-44s %8.1f %6.2f %6.1f %9.1f %9.1f %8.0f %6.2f" % (name[:44], a["us"], ghz, 100 * util, fetch / 1e6, write / 1e6,
                                                                  (fetch + write) / 1e9 / t, hit / (hit + miss) if hit + miss else 0))
        k = agg.setdefault(name, dict(launches=0, fetch=0.0, write=0.0, us=0.0, mfma=0.0))
        k["launches"] += 1; k["fetch"] += fetch; k["write"] += write; k["us"] += a["us"]; k["mfma"] += util
    if out_json:
        doc = {"workload": wl_name, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, summarised by tools/pmc_summary.py; "
                         "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); KB -> bytes; "
                         + workload + ", last step of the run",
               "kernels": {n: {"launches": v["launches"],
                               "fetch_bytes_per_launch": v["fetch"] / v["launches"],
                               "write_bytes_per_launch": v["write"] / v["launches"],
                               "hbm_bytes_per_launch": (v["fetch"] + v["write"]) / v["launches"],
                               "avg_launch_us": v["us"] / v["launches"],
                               "mfma_busy": v["mfma"] / v["launches"]} for n, v in agg.items()}}
        json.dump(doc, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
