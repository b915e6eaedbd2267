#!/usr/bin/env python3
"""Headline benchmark: images/sec of the YOLOv2 (yolo.cfg) forward path at 608x608, fp32, on MI355X.

One "step" = one pass of the hot path over one batch of synthetic frames that are
already resident in HBM: NCHW->NHWC, 23 convolutions (+BN/bias/leaky), 5 maxpools,
route/reorg, region head, box decode, per-class NMS and compaction of the detections,
ending with the small D2H copy of the compact detection records
(y2_forward_device + y2_detect_resident of libsr_yolo2.so).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: one process per GPU; rank 0 loads the weights, packs them into the
kernel-layout arena and the arena is replicated with ONE RCCL broadcast
(torch.distributed, backend "nccl"); every rank then runs its own frame batch
(weak scaling, no per-step collective).  Timing = K steps between barriers, MAX over ranks.

Rank 0 prints one JSON line (see DESIGN.md section "Measurement"): value, roofline
of the dominant kernel from HIP-event timings taken inside the timed region, and the
CPU baseline (the reference's own CPU path compiled into oracle/_ref when present,
otherwise the oracle port) on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from sr_object_detection_amd import darknet, synth, zoo  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[2]: the configuration the metric (608x608 fp32) is quoted on
    "yolo608_b32": dict(net="yolo", size=608, batch=32),
    # configs[1]
    "yolo416_b8": dict(net="yolo", size=416, batch=8),
    "tiny416_b1": dict(net="tiny-yolo-voc", size=416, batch=1),
    # configs[3] per-GPU share (64 frames over 8 GPUs), synthetic 9418-node tree (cfg/9k.tree is corrupt)
    "yolo9000_544_b8": dict(net="yolo9000", size=544, batch=8),
    # configs[4]: darknet19_448 classifier, fp16 storage / fp32 accumulate on the fp16 matrix cores; and in fp32
    "darknet19_448_b128_f16": dict(net="darknet19", size=448, batch=128, half=True),
    "darknet19_448_b32": dict(net="darknet19", size=448, batch=32),
    "yolo608_b32_f16": dict(net="yolo", size=608, batch=32, half=True),
}


def write_cfg(tmp: str, name: str, size: int, batch: int, fname: str = "net.cfg") -> str:
    """cfg text (+ synthetic tree for yolo9000) into tmp; returns the cfg path."""
    tree = None
    if name == "yolo9000":
        tree = os.path.join(tmp, "syn9k.tree")
        if not os.path.exists(tree):
            synth.write_tree(tree, 9418)
    cfg = os.path.join(tmp, fname)
    open(cfg, "w").write(zoo.cfg_text(name, size, size, batch, tree_path=tree))
    return cfg
PEAK_FP16_MFMA_TFLOPS = 2516.6     # v_mfma_f32_32x32x16_f16: 32 cycles per 32768 FLOP per SIMD -> 16x the fp32 rate
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz x 256 FLOP/clk
THRESH, NMS = 0.2, 0.4             # Detector defaults (yolo_v2_class.hpp:45,50)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="yolo608_b32", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-iters", type=int, default=2, help="timed CPU-baseline forwards (0 disables the leg)")
    ap.add_argument("--seed", type=int, default=31)
    ap.add_argument("--host-input", action="store_true",
                    help="frames start in (pageable) host memory: the PCIe-inclusive rate noted in DESIGN.md, never `value`")
    return ap.parse_args()


def conv_layer_flops(net):
    """2*M*N*K per conv layer and image (src_yolo2/darknet.c:115-131), indexed by layer."""
    out = {}
    for i in range(net.n):
        l = net.layer(i)
        if darknet.LAYER_TYPES[l.type] == "CONVOLUTIONAL":
            out[i] = 2.0 * l.n * l.size * l.size * l.c * l.out_h * l.out_w
    return out


def pmc_traffic(kernel: str, workload: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (counters cannot be read
    live): profiles/r*_pmc_traffic.json, produced by tools/pmc_summary.py from separate --pmc runs of this
    same command (FETCH_SIZE doubled for gfx950, KB -> bytes).  None when no profile covers the kernel."""
    if workload != "yolo608_b32":
        return None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            k = json.load(open(path))["kernels"].get(kernel)
            if k:
                return int(k["hbm_bytes_per_launch"])
        except (OSError, ValueError, KeyError):
            continue
    return None


def usable_cores() -> int:
    """CPU threads this process may really use: the cgroup quota when there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(cfg_b1: str, wts: str, size: int, iters: int, tmp: str):
    """The reference CPU path on this host's cores, batch 1 (the reference's own mode), wall clock."""
    if iters <= 0:
        return None
    cores = usable_cores()
    env = dict(os.environ, OMP_NUM_THREADS=str(cores))
    ref_driver = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    t0 = time.time()
    if os.path.exists(ref_driver):
        try:
            out = subprocess.run([ref_driver, "time", cfg_b1, wts, str(iters)], env=env, capture_output=True,
                                 text=True, timeout=600, check=True).stdout.strip().splitlines()[-1]
            r = json.loads(out)
            return dict(value=round(1.0 / r["mean_s"], 4), unit="images/sec", cores=cores, kind="reference",
                        sample="%d forwards of yolo.cfg %dx%d batch 1 (network_predict only, after 1 warm-up) by the "
                               "reference's own C sources built -O2 -fopenmp; %.1f s of CPU work" % (
                                   iters, size, size, time.time() - t0))
        except Exception as e:      # fall through to the port
            sys.stderr.write("cpu_baseline: reference driver failed (%s); using the oracle port\n" % e)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    from oracle import oracle_capi
    on = oracle_capi.OracleNet(cfg_b1, wts)
    x = synth.image_batch(1, 3, size, size)
    secs = on.time_predict(x, iters)
    on.close()
    return dict(value=round(iters / secs, 4), unit="images/sec", cores=cores, kind="port",
                sample="%d forwards of yolo.cfg %dx%d batch 1 (predict only, after 1 warm-up) by oracle/y2_oracle.c; "
                       "%.1f s of CPU work" % (iters, size, size, time.time() - t0))


def map_equiv_vs_cpu(cfg_b1: str, wts: str, x0: np.ndarray, size: int, gpu_dets0: np.ndarray, tmp: str):
    """BASELINE.json's "mAP-equiv vs CPU ref": the detections of ONE frame from the CPU reference path (decode
    thresh/NMS as in the timed step) serve as ground truth for the GPU engine's detections of the same frame;
    VOC AP at IoU 0.5 averaged over the classes present (sr_object_detection_amd/voc_eval.py).  1.0 = interchangeable."""
    from sr_object_detection_amd import voc_eval
    ref_driver = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    kind = "reference"
    try:
        if not os.path.exists(ref_driver):
            raise FileNotFoundError(ref_driver)
        inp = os.path.join(tmp, "frame0.bin")
        np.ascontiguousarray(x0, dtype=np.float32).tofile(inp)
        out_dir = os.path.join(tmp, "ref0")
        os.makedirs(out_dir, exist_ok=True)
        env = dict(os.environ, OMP_NUM_THREADS=str(usable_cores()))
        subprocess.run([ref_driver, "net", cfg_b1, wts, inp, out_dir, repr(THRESH), repr(NMS), "0"], env=env,
                       capture_output=True, timeout=600, check=True)
        boxes = np.fromfile(os.path.join(out_dir, "boxes_0.bin"), dtype=np.float32).reshape(-1, 4)
        post = np.fromfile(os.path.join(out_dir, "probs_post_0.bin"), dtype=np.float32).reshape(len(boxes), -1)
    except Exception as e:
        sys.stderr.write("map_equiv: reference driver unavailable (%s); using the oracle port\n" % e)
        kind = "port"
        from oracle import oracle_capi
        on = oracle_capi.OracleNet(cfg_b1, wts)
        on.predict(x0[None])
        boxes, probs = on.region_boxes(0, THRESH)
        post = oracle_capi.do_nms_sort(boxes, probs, NMS)
        on.close()
    ref_rows = voc_eval.detections_from_dense(boxes, post, THRESH, size, size)
    cand = [(int(d["obj_id"]), float(d["prob"]), (d["x"] - d["w"] / 2) * size, (d["y"] - d["h"] / 2) * size,
             (d["x"] + d["w"] / 2) * size, (d["y"] + d["h"] / 2) * size) for d in gpu_dets0]
    m, n_ref = voc_eval.map_equiv_rows({"frame0": cand}, {"frame0": ref_rows})
    return dict(value=None if np.isnan(m) else round(m, 4), iou=0.5, frames=1, cpu_detections=n_ref, gpu_detections=len(cand), kind=kind)


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        sys.stderr.write("bench: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE\n" % (args.gpus, world))
    import torch
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    # Y2_BENCH_BACKEND=gloo + Y2_BENCH_SHARE_GPU=1 rehearse the N>1 path on a box with ONE GPU (all ranks on
    # cuda:0, broadcast staged through the host); the real multi-GPU run uses nccl (= RCCL), one GPU per rank.
    backend = os.environ.get("Y2_BENCH_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count() if os.environ.get("Y2_BENCH_SHARE_GPU") else local_rank
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    wl = WORKLOADS[args.workload]
    name, size, batch = wl["net"], wl["size"], wl["batch"]
    half = bool(wl.get("half"))
    peak = PEAK_FP16_MFMA_TFLOPS if half else PEAK_FP32_MFMA_TFLOPS
    tmp = tempfile.mkdtemp(prefix="y2bench_r%d_" % rank)
    cfg = write_cfg(tmp, name, size, batch)
    layers = zoo.resolve(name, size)

    L = darknet.lib()
    net = darknet.Network.parse_network_cfg(cfg, gpu=device_index)
    net.set_half(half)
    wts = os.path.join(tmp, "net.weights")
    if rank == 0:
        synth.write_weights(wts, layers, args.seed)
        net.load_weights(wts)
        net.prepare()
    arena_ptr, arena_bytes = net.weights_arena()
    if world > 1:
        # replicate the packed weights: ONE broadcast of the kernel-layout arena over RCCL/xGMI
        on_gpu = backend == "nccl"
        buf = torch.empty(arena_bytes, dtype=torch.uint8, device="cuda" if on_gpu else "cpu")
        to_buf = L.y2h_memcpy_d2d if on_gpu else L.y2h_memcpy_d2h
        from_buf = L.y2h_memcpy_d2d if on_gpu else L.y2h_memcpy_h2d
        if rank == 0:
            assert to_buf(buf.data_ptr(), arena_ptr, arena_bytes, None) == 0
            L.y2h_device_sync()
        dist.broadcast(buf, src=0)
        torch.cuda.synchronize()
        if rank != 0:
            assert from_buf(arena_ptr, buf.data_ptr(), arena_bytes, None) == 0
            L.y2h_device_sync()
            net.weights_resident()
        del buf

    # this rank's frames: global image index = rank*batch + i, resident in HBM before timing starts
    x = synth.image_batch(batch, 3, size, size, seed=0xC0FFEE + rank * batch)
    d_x = torch.from_numpy(x).cuda()
    torch.cuda.synchronize()

    is_detector = darknet.LAYER_TYPES[net.last.type] == "REGION"

    def step():
        if not is_detector:                 # classifier: forward + host copy of the class scores
            return net.predict_device(d_x.data_ptr()), np.zeros(1)
        if args.host_input:
            return net.detect(x, THRESH, NMS)
        net.forward_device(d_x.data_ptr())
        return net.detect_resident(THRESH, NMS)

    for _ in range(args.warmup):
        dets, counts = step()
    net.set_timing(True)
    flops = conv_layer_flops(net)
    kernels = {i: net.layer_kernel(i) for i in range(net.n)}
    if rank == 0 and os.environ.get("Y2_BENCH_DUMP_KERNELS"):
        # conv launches of one step in order, for tools/pmc_summary.py to label rocprofv3 rows with
        with open(os.environ["Y2_BENCH_DUMP_KERNELS"], "w") as f:
            json.dump([kernels[i] for i in sorted(flops)], f)
    per_kernel_ms = {}
    per_kernel_flops = {}
    per_kernel_launches = {}

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def account_layer_times():
        ms = net.layer_times_ms()          # HIP events recorded on the engine's stream around every layer
        for i, k in kernels.items():
            if i in flops and i < len(ms):
                per_kernel_ms[k] = per_kernel_ms.get(k, 0.0) + float(ms[i])
                per_kernel_flops[k] = per_kernel_flops.get(k, 0.0) + flops[i] * batch
                per_kernel_launches[k] = per_kernel_launches.get(k, 0) + 1

    pipelined = not args.host_input and not os.environ.get("Y2_BENCH_NO_PIPELINE")
    barrier()
    t0 = time.perf_counter()
    if pipelined:
        # Every step is still one forward + one decode/NMS/collect + one fetch of the detections; the host side of step
        # i (waiting for its records, unpacking them) overlaps the device side of step i+1: y2_detect_fetch waits for an
        # event behind step i's D2H copies only, so the next forward is already queued when the host blocks.
        # (a classifier's step is forward + host copy of the class scores: y2_output_enqueue / y2_output_fetch.)
        def enqueue_results():
            if is_detector:
                net.detect_enqueue(THRESH, NMS)
            else:
                net.output_enqueue()

        def fetch_results():
            return net.detect_fetch() if is_detector else (net.output_fetch(), np.zeros(1))

        net.forward_device(d_x.data_ptr())
        enqueue_results()
        for _ in range(args.steps - 1):
            account_layer_times()                       # needs the previous forward's events, not its results
            net.forward_device(d_x.data_ptr())
            dets, counts = fetch_results()
            enqueue_results()
        account_layer_times()
        dets, counts = fetch_results()
    else:
        for _ in range(args.steps):
            dets, counts = step()
            account_layer_times()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_images = world * batch * args.steps
        dom = max(per_kernel_ms, key=lambda k: per_kernel_ms[k]) if per_kernel_ms else None
        roof = None
        if dom:
            ach = per_kernel_flops[dom] / (per_kernel_ms[dom] * 1e-3) / 1e12
            roof = dict(bound="mfma", kernel=dom, achieved=round(ach, 2), peak=peak, unit="TFLOP/s",
                        frac=round(ach / peak, 4), traffic=pmc_traffic(dom, args.workload),
                        launches_per_step=per_kernel_launches[dom] // max(args.steps, 1),
                        avg_launch_ms=round(per_kernel_ms[dom] / per_kernel_launches[dom], 4),
                        avg_launch_gflop=round(per_kernel_flops[dom] / per_kernel_launches[dom] / 1e9, 3))
        conv_ms = sum(per_kernel_ms.values()) / max(args.steps, 1)
        cfg_b1 = write_cfg(tmp, name, size, 1, "net_b1.cfg")
        # the CPU leg is timed at N=1 only (rank 0 would otherwise hold the other ranks at the final barrier)
        cpu = cpu_baseline(cfg_b1, wts, size, args.cpu_iters, tmp) if world == 1 else None
        mapeq = None
        if world == 1 and args.cpu_iters > 0 and is_detector and not args.host_input:
            mapeq = map_equiv_vs_cpu(cfg_b1, wts, x[0], size, dets[0], tmp)
        line = {
            "metric": "images/sec YOLOv2 608x608 fp32" if (size == 608 and not half) else
                      "images/sec %s %dx%d %s" % (name, size, size, "fp16" if half else "fp32"),
            "value": round(total_images / elapsed, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16" if half else "f32", "data": "synthetic",
            "config": {"workload": "%s %dx%d batch %d per GPU: forward + region decode + NMS(%.1f) + collect, "
                                   "%s" % (name + ".cfg", size, size, batch, NMS,
                                                   "inputs in host memory (PCIe-inclusive)" if args.host_input
                                                   else "inputs resident in HBM"),
                       "global_batch": batch * world, "parallelism": "frame-sharded x%d (RCCL weight broadcast)" % world,
                       "gflop_per_image": round(zoo.conv_flops(layers) / 1e9, 3),
                       "conv_ms_per_step": round(conv_ms, 3),
                       "host_fetch_overlaps_next_forward": bool(pipelined),
                       "detections_in_last_batch": int(np.sum(counts))},
            "roofline": roof, "cpu_baseline": cpu, "map_equiv_vs_cpu": mapeq,
            "kernels_ms_per_step": {k: round(v / max(args.steps, 1), 3) for k, v in sorted(per_kernel_ms.items())},
            "device": darknet.device_name(),
        }
        print(json.dumps(line))
        sys.stdout.flush()
    net.free()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
