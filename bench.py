#!/usr/bin/env python3
"""Headline benchmark: images/sec of the YOLOv2 (yolo.cfg) forward path at 608x608, fp32, on MI355X.

One "step" = one pass of the hot path over one batch of synthetic frames that are
already resident in HBM: NCHW->NHWC, 23 convolutions (+BN/bias/leaky), 5 maxpools,
route/reorg, region head, box decode, per-class NMS and compaction of the detections,
ending with the small D2H copy of the compact detection records
(y2_forward_device + y2_detect_enqueue/fetch of libsr_yolo2.so).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: one process per GPU.  Under torchrun the ranks exist already; a bare
`python bench.py --gpus N` starts them itself (sr_object_detection_amd/launch.py: the parent
never touches the GPU, it spawns N fresh children with RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*
and relays rank 0's line).  Rank 0 loads the weights and packs them into the kernel-layout
arena; ONE broadcast of that arena replicates the model -- in place on the arena pointer,
through torch.distributed (backend "nccl" = RCCL) by default or through the library's own
C-ABI (`--bcast c-abi`: y2_comm_init_rank + y2_broadcast_weights, RCCL called directly);
every rank then runs its own frame batches (weak scaling, no per-step collective).
Timing = K steps between barriers, MAX over ranks.

Rank 0 prints one JSON line: value, roofline of the dominant kernel
from HIP-event timings taken inside the timed region, the same loop fed from pinned host
memory (PCIe-inclusive, never `value`), the CPU baseline (the reference's own CPU path
compiled into oracle/_ref when present, otherwise the oracle port) and the
detection-equivalence of the GPU and CPU paths on frames of the timed batch.
"""
from __future__ import annotations

import gc
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[2]: the configuration the metric (608x608 fp32) is quoted on
    "yolo608_b32": dict(net="yolo", size=608, batch=32),
    # configs[1]
    "yolo416_b8": dict(net="yolo", size=416, batch=8),
    "tiny416_b1": dict(net="tiny-yolo-voc", size=416, batch=1),
    # the robot's own call pattern (one frame per test_detector_img call) on the large nets
    "yolo608_b1": dict(net="yolo", size=608, batch=1),
    "yolo416_b1": dict(net="yolo", size=416, batch=1),
    # configs[3] per-GPU share (64 frames over 8 GPUs), synthetic 9418-node tree (cfg/9k.tree is corrupt)
    "yolo9000_544_b8": dict(net="yolo9000", size=544, batch=8),
    # configs[4]: darknet19_448 classifier, fp16 storage / fp32 accumulate on the fp16 matrix cores; and in fp32
    "darknet19_448_b128_f16": dict(net="darknet19", size=448, batch=128, half=True),
    "darknet19_448_b32": dict(net="darknet19", size=448, batch=32),
    "yolo608_b32_f16": dict(net="yolo", size=608, batch=32, half=True),
}

PEAK_FP16_MFMA_TFLOPS = 2516.6     # v_mfma_f32_32x32x16_f16: 32 cycles per 32768 FLOP per SIMD -> 16x the fp32 rate
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz x 256 FLOP/clk
THRESH, NMS = 0.2, 0.4             # Detector defaults (yolo_v2_class.hpp:45,50)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # >= 50 timed batches (the driver passes its own K / W)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="yolo608_b32", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-iters", type=int, default=3,
                    help="timed CPU-baseline forwards per leg, median reported (0 disables the CPU legs)")
    ap.add_argument("--cpu-legs", default="all", choices=["all", "headline"],
                    help="all: tiny@416 / yolo@416 / yolo@608 at 1 thread and all cores (SURVEY 8d); headline: the workload's net, all cores")
    ap.add_argument("--equiv-frames", type=int, default=8, help="frames of the timed batch compared with the CPU reference")
    ap.add_argument("--seed", type=int, default=31)
    ap.add_argument("--host-input", default="also", choices=["also", "only", "off"],
                    help="also: after the resident loop, time the same steps fed from pinned host memory (reported as "
                         "host_input, never as value); only: time just that; off: skip it")
    ap.add_argument("--bcast", default=os.environ.get("Y2_BENCH_BCAST", "torch"), choices=["torch", "c-abi"],
                    help="transport of the one weight broadcast at N>1")
    ap.add_argument("--detect-overlap", type=int, default=-1,
                    help="1: decode/NMS of step i on their own stream beside the forward of step i+1 (y2_set_detect_overlap); "
                         "-1: only for wide heads (>= 1000 classes: yolo9000 +5-7 %%; neutral to -1 %% on the 80-class heads, whose "
                         "decode kernels just queue between the persistent conv grids)")
    ap.add_argument("--autotune", type=int, default=0,
                    help="1: the plan measures the conv tile shapes once instead of modelling them (y2_set_autotune)")
    ap.add_argument("--latency-iters", type=int, default=200,
                    help="calls per entry of the batch-1 latency block (test_detector_img / Detector::detect on tiny-yolo-voc and "
                         "yolo at 416; N=1 only; 0 disables)")
    ap.add_argument("--clock-probe", type=int, default=1,
                    help="0: skip the 0.6 s clock probe behind the timed region (profiling runs: its launches would fill the kernel trace)")
    ap.add_argument("--dump-dets", default=None, help="write each rank's last-batch detections to PATH.rank<r>.npz")
    return ap.parse_args()


def write_cfg(tmp: str, name: str, size: int, batch: int, fname: str = "net.cfg") -> str:
    """cfg text (+ synthetic tree for yolo9000) into tmp; returns the cfg path."""
    from sr_object_detection_amd import synth, zoo
    tree = None
    if name == "yolo9000":
        tree = os.path.join(tmp, "syn9k.tree")
        if not os.path.exists(tree):
            synth.write_tree(tree, 9418)
    cfg = os.path.join(tmp, fname)
    open(cfg, "w").write(zoo.cfg_text(name, size, size, batch, tree_path=tree))
    return cfg


def conv_layer_flops(net):
    """2*M*N*K per conv layer and image (src_yolo2/darknet.c:115-131), indexed by layer."""
    from sr_object_detection_amd import darknet
    out = {}
    for i in range(net.n):
        l = net.layer(i)
        if darknet.LAYER_TYPES[l.type] == "CONVOLUTIONAL":
            out[i] = 2.0 * l.n * l.size * l.size * l.c * l.out_h * l.out_w
    return out


def pmc_traffic(kernel: str, workload: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (counters cannot be read
    live): profiles/r*_pmc_traffic*.json, produced by tools/pmc_summary.py from separate --pmc runs of this
    same command (FETCH_SIZE doubled for gfx950, KB -> bytes).  None when no profile covers the kernel."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True):
        try:
            j = json.load(open(path))
            if j.get("workload", "yolo608_b32") != workload:
                continue
            k = j["kernels"].get(kernel)
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


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


REF_DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")


def _cpu_leg_start(cfg_b1: str, wts: str, threads: int, iters: int):
    """start one timing run of the reference's network_predict (1 warm-up + iters timed forwards, batch 1)"""
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="false")
    return subprocess.Popen([REF_DRIVER, "time", cfg_b1, wts, str(iters)], env=env, stdout=subprocess.PIPE,
                            stderr=subprocess.DEVNULL, text=True)


def _cpu_leg_port(cfg_b1: str, wts: str, size: int, threads: int, iters: int) -> dict:
    os.environ["OMP_NUM_THREADS"] = str(threads)
    from oracle import oracle_capi
    from sr_object_detection_amd import synth
    on = oracle_capi.OracleNet(cfg_b1, wts)
    x = synth.image_batch(1, 3, size, size)
    times = sorted(on.time_predict(x, 1) for _ in range(iters))
    on.close()
    return {"median_s": times[len(times) // 2], "iters": iters}


def cpu_baseline(tmp: str, headline: dict, wts_headline: str, iters: int, legs: str, seed: int):
    """the reference's CPU path (src_yolo2/*.c built -O2 -fopenmp by oracle/build_ref.sh; gemm.c:156 is
    row-parallel OpenMP as is) on this host, batch 1 (the reference's own mode), network_predict only, 1 warm-up then the
    median of `iters` forwards -- tiny-yolo-voc@416, yolo@416 and yolo@608, at 1 thread and on all usable cores.  The
    single-thread legs run side by side (one core each) after the GPU part has finished; the all-core legs run alone.
    The long single-thread legs take fewer iterations so that the whole CPU part stays within about a minute."""
    if iters <= 0:
        return None
    from sr_object_detection_amd import synth, zoo
    cores = usable_cores()
    t_start = time.time()
    kind = "reference" if os.path.exists(REF_DRIVER) else "port"
    nets = [("tiny-yolo-voc", 416), ("yolo", 416), ("yolo", 608)] if legs == "all" else []
    hl = (headline["net"], headline["size"])
    if hl not in nets:
        nets.append(hl)
    files = {}
    for name, size in nets:
        cfg = write_cfg(tmp, name, size, 1, "cpu_%s_%d.cfg" % (name.replace("-", "_"), size))
        if name == headline["net"]:
            w = wts_headline                       # same architecture: the weight file fits every input size
        else:
            w = os.path.join(tmp, "cpu_%s.weights" % name.replace("-", "_"))
            if not os.path.exists(w):
                synth.write_weights(w, zoo.resolve(name, size), seed)
        files[(name, size)] = (cfg, w)
    gflop = {k: zoo.conv_flops(zoo.resolve(k[0], k[1])) / 1e9 for k in nets}
    rows = []

    def finish(p):
        out = p.communicate(timeout=900)[0].strip().splitlines()[-1]
        return json.loads(out)

    if kind == "reference":
        try:
            single = []
            if legs == "all":
                for k in nets:          # 1 thread: about 0.25 GFLOP/s-per-GFLOP ... budget the long ones
                    it = iters if gflop[k] < 40 else 1
                    single.append((k, it, _cpu_leg_start(*files[k], threads=1, iters=it)))
            for k, it, p in single:
                r = finish(p)
                rows.append(dict(config="%s.cfg %dx%d b1" % (k[0], k[1], k[1]), threads=1, iters=it,
                                 median_s=round(r["median_s"], 4), images_per_sec=round(1.0 / r["median_s"], 4)))
            for k in nets:
                r = finish(_cpu_leg_start(*files[k], threads=cores, iters=iters))
                rows.append(dict(config="%s.cfg %dx%d b1" % (k[0], k[1], k[1]), threads=cores, iters=iters,
                                 median_s=round(r["median_s"], 4), images_per_sec=round(1.0 / r["median_s"], 4)))
        except Exception as e:      # fall through to the port
            sys.stderr.write("cpu_baseline: reference driver failed (%s); using the oracle port\n" % e)
            kind, rows = "port", []
    if kind == "port":
        for k in nets:
            for threads in ([1, cores] if legs == "all" else [cores]):
                it = iters if (threads > 1 or gflop[k] < 40) else 1
                r = _cpu_leg_port(*files[k], size=k[1], threads=threads, iters=it)
                rows.append(dict(config="%s.cfg %dx%d b1" % (k[0], k[1], k[1]), threads=threads, iters=it,
                                 median_s=round(r["median_s"], 4), images_per_sec=round(1.0 / r["median_s"], 4)))
    head = next(r for r in rows if r["config"].startswith("%s.cfg %dx" % hl) and r["threads"] == cores)
    return dict(value=head["images_per_sec"], unit="images/sec", cores=cores, kind=kind,
                sample="median of %d forwards (after 1 warm-up) of %s batch 1, network_predict only, by %s; all legs "
                       "together %.0f s of wall time" % (
                           head["iters"], head["config"],
                           "the reference's own C sources built -O2 -fopenmp" if kind == "reference" else "oracle/y2_oracle.c",
                           time.time() - t_start),
                cpu_model=cpu_model(), legs=rows)


def equivalence_vs_cpu(tmp: str, name: str, size: int, wts: str, x, gpu_dets, nframes: int):
    """BASELINE.json's "mAP-equiv vs CPU ref" over `nframes` frames of the timed batch: the CPU reference path's
    detections (same decode thresh / NMS as the timed step) against the GPU engine's for the same frames --
    identical post-NMS counts, identical classes, box / probability differences (the 1e-4 bar of north_star), and
    VOC AP at IoU 0.5 with the CPU detections as ground truth (sr_object_detection_amd/voc_eval.py; 1.0 =
    interchangeable).  Random frames and weights carry no decision margin, so a probability within ~2e-5 of the
    threshold may legitimately flip; flips are reported, not hidden."""
    import numpy as np
    from sr_object_detection_amd import voc_eval
    nframes = max(1, min(nframes, len(gpu_dets)))
    kind = "reference"
    boxes_all, post_all = [], []
    try:
        if not os.path.exists(REF_DRIVER):
            raise FileNotFoundError(REF_DRIVER)
        cfg_n = write_cfg(tmp, name, size, nframes, "equiv_b%d.cfg" % nframes)
        inp = os.path.join(tmp, "equiv_frames.bin")
        np.ascontiguousarray(x[:nframes], dtype=np.float32).tofile(inp)
        out_dir = os.path.join(tmp, "equiv_ref")
        os.makedirs(out_dir, exist_ok=True)
        env = dict(os.environ, OMP_NUM_THREADS=str(usable_cores()))
        subprocess.run([REF_DRIVER, "net", cfg_n, wts, inp, out_dir, repr(THRESH), repr(NMS), "0"], env=env,
                       capture_output=True, timeout=900, check=True)
        for b in range(nframes):
            boxes = np.fromfile(os.path.join(out_dir, "boxes_%d.bin" % b), dtype=np.float32).reshape(-1, 4)
            post = np.fromfile(os.path.join(out_dir, "probs_post_%d.bin" % b), dtype=np.float32).reshape(len(boxes), -1)
            boxes_all.append(boxes)
            post_all.append(post)
    except Exception as e:
        sys.stderr.write("equivalence: reference driver unavailable (%s); using the oracle port\n" % e)
        kind = "port"
        from oracle import oracle_capi
        cfg_b1 = write_cfg(tmp, name, size, 1, "equiv_b1.cfg")
        on = oracle_capi.OracleNet(cfg_b1, wts)
        boxes_all, post_all = [], []
        for b in range(nframes):
            on.predict(x[b][None])
            boxes, probs = on.region_boxes(0, THRESH)
            boxes_all.append(boxes)
            post_all.append(oracle_capi.do_nms_sort(boxes, probs, NMS))
        on.close()
    ref_rows, cand_rows = {}, {}
    n_ref = n_gpu = 0
    counts_equal = True
    max_box = max_prob = 0.0
    mismatched = []
    for b in range(nframes):
        boxes, post = boxes_all[b], post_all[b]
        key = "frame%d" % b
        ref_rows[key] = voc_eval.detections_from_dense(boxes, post, THRESH, size, size)
        d = gpu_dets[b]
        cand_rows[key] = [(int(r["obj_id"]), float(r["prob"]), (r["x"] - r["w"] / 2) * size, (r["y"] - r["h"] / 2) * size,
                           (r["x"] + r["w"] / 2) * size, (r["y"] + r["h"] / 2) * size) for r in d]
        keep = np.nonzero(post.max(axis=1) > THRESH)[0]          # ascending box order, as y2_detect collects
        n_ref += keep.size
        n_gpu += len(d)
        if keep.size != len(d) or not np.array_equal(post[keep].argmax(axis=1), d["obj_id"]):
            counts_equal = counts_equal and keep.size == len(d)
            mismatched.append(b)
            continue
        if keep.size:
            got = np.stack([d["x"], d["y"], d["w"], d["h"]], 1)
            want = boxes[keep]
            max_box = max(max_box, float((np.abs(got - want) / np.maximum(1.0, np.abs(want))).max()))
            max_prob = max(max_prob, float(np.abs(d["prob"] - post[keep].max(axis=1)).max()))
    m, _ = voc_eval.map_equiv_rows(cand_rows, ref_rows)
    return dict(value=None if np.isnan(m) else round(m, 4), iou=0.5, frames=nframes, cpu_detections=n_ref, gpu_detections=n_gpu,
                post_nms_counts_equal=bool(counts_equal), frames_with_a_different_set=mismatched,
                max_abs_box_err=max_box, max_abs_prob_err=max_prob, tolerance=1e-4, kind=kind)


def latency_block(tmp, seed, iters=200):
    """Batch-1 latency of the reference's REAL callers, frame in host memory -> objects in host memory, per call:
      * test_detector_img (detector.c:558-598, what KinectUtil.cpp:403 calls per camera frame): resize_image to the network
        size, network_predict, get_region_boxes, do_nms_sort(0.1), object[] -- with a network-sized frame and with a 640x480
        camera frame (device-side resize);
      * Detector::detect(image_t) (yolo_v2_class.cpp:173-249), through the class's C face;
      * the device part alone (frame already in HBM: forward + decode + NMS + fetch), for the gap between the two.
    tiny-yolo-voc 416 (the robot's net) and yolo 416, p50 / p95 of `iters` back-to-back calls after 20 warm-up calls."""
    import numpy as np
    import torch
    from sr_object_detection_amd import darknet, synth, zoo

    def pcts(fn):
        for _ in range(20):
            fn()
        ts = []
        for _ in range(iters):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        ts = np.sort(np.asarray(ts)) * 1e3
        return dict(p50_ms=round(float(ts[len(ts) // 2]), 4), p95_ms=round(float(ts[int(len(ts) * 0.95)]), 4))

    out = dict(iters=iters, batch=1, frame="416x416x3 float planes in host memory (and 640x480x3 for the camera-sized call)",
               thresh=0.24, nms=0.1, nets={})
    cam = synth.image_batch(1, 3, 480, 640, seed=3)[0]
    for name in ("tiny-yolo-voc", "yolo"):
        cfg = write_cfg(tmp, name, 416, 1, "lat_%s.cfg" % name.replace("-", "_"))
        wts = os.path.join(tmp, "lat_%s.weights" % name.replace("-", "_"))
        synth.write_weights(wts, zoo.resolve(name, 416), seed)
        net = darknet.Network.parse_network_cfg(cfg)
        net.load_weights(wts)
        x = synth.image_batch(1, 3, 416, 416)
        d_x = torch.from_numpy(x).cuda()
        det = darknet.Detector(cfg, wts, 0)
        r = {
            "test_detector_img": pcts(lambda: net.test_detector_img(x[0], 0.24)),
            "test_detector_img_640x480": pcts(lambda: net.test_detector_img(cam, 0.24)),
            "Detector_detect": pcts(lambda: det.detect(x[0], 0.24, False, 0.1)),
            "device_only": pcts(lambda: (net.forward_device(d_x.data_ptr()), net.detect_resident(0.24, 0.1))),
        }
        net.set_timing(True)
        net.forward_device(d_x.data_ptr())
        net.detect_resident(0.24, 0.1)
        r["forward_kernels_ms"] = round(float(np.sum(net.layer_times_ms())), 4)      # HIP events around every layer of one forward
        r["objects"] = len(net.test_detector_img(x[0], 0.24))
        out["nets"][name + " 416x416"] = r
        det.free()
        net.free()
    return out


def main():
    args = parse_args()
    # Become the launcher before anything can touch the GPU (no torch, no libsr_yolo2.so in this process yet).
    from sr_object_detection_amd import launch
    launch.self_launch_if_needed(args.gpus)

    import numpy as np
    from sr_object_detection_amd import darknet, synth, zoo
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE\n" % (args.gpus, world))
    import torch
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    # Y2_BENCH_BACKEND=gloo + Y2_BENCH_SHARE_GPU=1 rehearse the N>1 path on a box with ONE GPU (all ranks on
    # cuda:0, broadcast staged through the host); the real multi-GPU run uses nccl (= RCCL), one GPU per rank.
    backend = os.environ.get("Y2_BENCH_BACKEND", "nccl")
    share = bool(os.environ.get("Y2_BENCH_SHARE_GPU"))
    ndev = torch.cuda.device_count()
    if world > ndev and not share:
        sys.exit("bench: %d ranks but only %d GPU(s) visible (one process per GPU; Y2_BENCH_SHARE_GPU=1 rehearses on one)" % (world, ndev))
    device_index = local_rank % ndev if share else local_rank
    torch.cuda.set_device(device_index)
    # Host side of rank r next to GPU r: pin this process to the cores of the NUMA node the GPU's PCI function hangs off
    # (the feed of 142 MB per batch and the fetch of the detections are the only host work; SURVEY section 7 names the host
    # feed as the scaling risk).  Skipped silently where /sys does not say (containers without the topology).
    numa_node = None
    if os.environ.get("Y2_BENCH_NUMA", "1") != "0":
        numa_node, cpus = darknet.numa_cpus_of_device(device_index)
        if cpus:
            try:
                allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
                if allowed:
                    os.sched_setaffinity(0, allowed)
                else:
                    numa_node = None
            except OSError:
                numa_node = None
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
    net.set_autotune(bool(args.autotune))
    detect_overlap = bool(args.detect_overlap) if args.detect_overlap >= 0 else (
        darknet.LAYER_TYPES[net.last.type] == "REGION" and net.last.classes >= 1000)
    net.set_detect_overlap(detect_overlap)
    wts = os.path.join(tmp, "net.weights")
    if rank == 0:
        synth.write_weights(wts, layers, args.seed)
        net.load_weights(wts)
        net.prepare()
    bcast_how, bcast_ms = "none (1 rank)", None
    comm_ranks = 1
    if world > 1:
        comm_ranks = dist.get_world_size()                 # (the C-ABI transport replaces it with ncclCommCount below)
        # replicate the packed weights: ONE broadcast of the kernel-layout arena, in place on the arena pointer
        arena_ptr, arena_bytes = net.weights_arena()
        # every rank must have laid its arena out as rank 0 did (same cfg, same modes): neither RCCL nor torch compares
        # counts across ranks.  y2_broadcast_weights does this handshake itself; for the torch transports it is done here.
        layouts = [None] * world
        dist.all_gather_object(layouts, net.weights_layout())
        if any(l != layouts[0] for l in layouts):
            sys.exit("bench: rank %d: weight arena layouts differ across ranks: %s" % (rank, layouts))
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        tb = time.perf_counter()
        if args.bcast == "c-abi" and backend == "nccl":
            # RCCL through the library's own C-ABI (include/sr_yolo2.h): the 128-byte unique id travels over the
            # process group's store, the collective runs on the arena pointer on the engine's stream
            uid = [darknet.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            comm = darknet.comm_init_rank(world, uid[0], rank, device_index)
            comm_ranks = darknet.comm_count(comm)[0]       # ncclCommCount of the communicator the arena travels on
            net.broadcast_weights(comm, 0)
            darknet.comm_destroy(comm)
            bcast_how = "RCCL ncclBroadcast via the C-ABI (y2_broadcast_weights, %s), in place on the arena" % darknet.comm_library()
        elif backend == "nccl":
            class _Arena:          # zero-copy view of the arena for torch (no D2D bounce)
                __cuda_array_interface__ = {"shape": (arena_bytes,), "typestr": "|u1", "data": (arena_ptr, False), "version": 2}
            try:
                view = torch.as_tensor(_Arena(), device=torch.device("cuda", device_index))
                assert view.data_ptr() == arena_ptr and view.numel() == arena_bytes
                dist.broadcast(view, src=0)
                torch.cuda.synchronize()
                del view
                bcast_how = "torch.distributed broadcast (nccl = RCCL), in place on the arena pointer"
            except (AssertionError, TypeError, RuntimeError) as e:
                sys.stderr.write("bench: zero-copy arena view unavailable (%s); staging through a torch tensor\n" % e)
                buf = torch.empty(arena_bytes, dtype=torch.uint8, device="cuda")
                if rank == 0:
                    assert L.y2h_memcpy_d2d(buf.data_ptr(), arena_ptr, arena_bytes, None) == 0
                    L.y2h_device_sync()
                dist.broadcast(buf, src=0)
                torch.cuda.synchronize()
                if rank != 0:
                    assert L.y2h_memcpy_d2d(arena_ptr, buf.data_ptr(), arena_bytes, None) == 0
                    L.y2h_device_sync()
                del buf
                bcast_how = "torch.distributed broadcast (nccl = RCCL) through a staging tensor"
            if rank != 0:
                net.weights_resident()
        else:
            buf = torch.empty(arena_bytes, dtype=torch.uint8, device="cpu")
            if rank == 0:
                assert L.y2h_memcpy_d2h(buf.data_ptr(), arena_ptr, arena_bytes, None) == 0
                L.y2h_device_sync()
            dist.broadcast(buf, src=0)
            if rank != 0:
                assert L.y2h_memcpy_h2d(arena_ptr, buf.data_ptr(), arena_bytes, None) == 0
                L.y2h_device_sync()
                net.weights_resident()
            del buf
            bcast_how = "%s broadcast staged through host memory (one-GPU rehearsal)" % backend
        torch.cuda.synchronize()
        bcast_ms = round((time.perf_counter() - tb) * 1e3, 2)

    # this rank's frames: global image index = rank*batch + i, resident in HBM before timing starts
    x = synth.image_batch(batch, 3, size, size, seed=0xC0FFEE + rank * batch)
    d_x = torch.from_numpy(x).cuda()
    torch.cuda.synchronize()

    is_detector = darknet.LAYER_TYPES[net.last.type] == "REGION"

    def enqueue_results():
        if is_detector:
            net.detect_enqueue(THRESH, NMS)
        else:
            net.output_enqueue()

    def fetch_results():
        return net.detect_fetch() if is_detector else (net.output_fetch(), np.zeros(1))

    def step():
        net.forward_device(d_x.data_ptr())
        enqueue_results()
        return fetch_results()

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

    def timed_loop(forward, account):
        """K steps between barriers.  Every step is one forward + one decode/NMS/collect + one fetch of the detections;
        the host side of step i (waiting for its records, unpacking them) overlaps the device side of step i+1:
        y2_detect_fetch waits for an event behind step i's D2H copies only, so the next forward is already queued when the
        host blocks.  (A classifier's step is forward + host copy of the class scores: y2_output_enqueue / _fetch.)"""
        res = (None, None)
        # The harness's own garbage collector stays out of the timed region: the interpreter's first full collection after
        # `import torch` is a 35-40 ms pause (tools/probes/steps_b1.py: once, ~450 calls into a loop; 0.2 ms per frame of a
        # 200-step batch-1 run), and what is timed here is the C library, not this script.
        gc.collect()
        gc.disable()
        barrier()
        t0 = time.perf_counter()
        forward(0)
        enqueue_results()
        for i in range(1, args.steps):
            if account:
                account_layer_times()                   # needs the previous forward's events, not its results
            forward(i)
            res = fetch_results()
            enqueue_results()
        if account:
            account_layer_times()
        res = fetch_results()
        barrier()
        el = time.perf_counter() - t0
        gc.enable()
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, res

    elapsed = None
    if args.host_input != "only":
        elapsed, (dets, counts) = timed_loop(lambda i: net.forward_device(d_x.data_ptr()), True)

    host = None
    if args.host_input != "off":
        # The same K steps with every batch starting in HOST memory: two pinned slots, the upload of batch i+1 on the copy
        # stream while batch i computes (y2_feed_*).  Two producers are timed: frames that already sit in the pinned slots
        # (a capture library / decoder that writes straight into them) and frames in pageable memory that a producer thread
        # copies into the slot first (141 MB per fp32 batch of 32 608x608 frames), as a camera thread would.
        import queue
        import threading
        net.set_timing(False)
        net.feed_open(2)
        slots = [net.feed_host(s) for s in range(2)]
        flat = x.reshape(-1)

        def run_fed(pageable: bool):
            if not pageable:
                for s_ in range(2):
                    slots[s_][:flat.size] = flat

                def forward(i):
                    if i == 0:
                        net.feed_submit(0)
                    net.feed_forward(i & 1)
                    if i + 1 < args.steps:
                        net.feed_submit((i + 1) & 1)    # waits ON THE DEVICE for the forward that last read that slot
                return timed_loop(forward, False)
            ready = queue.Queue()
            submitted = [threading.Semaphore(1), threading.Semaphore(1)]

            def producer():
                for k in range(args.steps):
                    s_ = k & 1
                    submitted[s_].acquire()             # batch k-2's upload has been submitted ...
                    net.feed_wait_host(s_)              # ... and has left the pinned buffer
                    slots[s_][:flat.size] = flat        # numpy releases the GIL for this copy
                    ready.put(k)

            th = threading.Thread(target=producer, daemon=True)

            def forward(i):
                if i == 0:
                    th.start()
                assert ready.get(timeout=120) == i
                net.feed_submit(i & 1)
                submitted[i & 1].release()
                net.feed_forward(i & 1)
            out = timed_loop(forward, False)
            th.join(timeout=120)
            return out

        keep_steps = args.steps
        args.steps = 2                                   # warm the path (copy stream, first-touch of the pinned pages)
        run_fed(False)
        args.steps = keep_steps
        p_el, p_res = run_fed(False)
        g_el, g_res = run_fed(True)
        net.feed_close()
        img = world * batch * args.steps
        host = dict(value=round(img / p_el, 2), unit="images/sec", ms_per_step=round(p_el / args.steps * 1e3, 3),
                    source="pinned host slots (2), async H2D of batch i+1 on a copy stream under batch i's forward; %.0f MB "
                           "cross PCIe per batch" % (flat.nbytes / 1e6),
                    pageable_source=dict(value=round(img / g_el, 2), ms_per_step=round(g_el / args.steps * 1e3, 3),
                                         note="a producer thread first copies each batch from pageable memory into the pinned slot"))
        if elapsed is None:
            elapsed, (dets, counts) = p_el, p_res
        else:
            host["fraction_of_resident"] = round(host["value"] / (img / elapsed), 4)
            host["pageable_source"]["fraction_of_resident"] = round(host["pageable_source"]["value"] / (img / elapsed), 4)

    if args.dump_dets and is_detector:
        np.savez(args.dump_dets + ".rank%d.npz" % rank, counts=np.asarray(counts),
                 **{"dets_%d" % b: dets[b] for b in range(len(dets))})

    # N > 1, outside the timed region: every rank's last batch is recomputed on rank 0 (same frames: seed 0xC0FFEE + r * batch,
    # same arena bytes, same kernels) and must come out bit for bit -- a rank whose replica, frames or device misbehaved
    # shows here, not in a throughput number.  Also what each rank ran on.
    shards_verified, rank_devices, shards_mismatched = None, None, []
    if world > 1:
        def digest(d_, c_):
            if is_detector:
                return [np.asarray(c_).tobytes()] + [np.ascontiguousarray(d_[b]).tobytes() for b in range(len(d_))]
            return [np.ascontiguousarray(d_).tobytes()]
        mine = digest(dets, counts)
        gathered = [None] * world
        dist.all_gather_object(gathered, (darknet.device_name(), darknet.device_pci_bus_id(), numa_node, mine))
        if rank == 0:
            rank_devices = ["%s @%s numa %s" % (g[0], g[1] or "?", "?" if g[2] is None else g[2]) for g in gathered]
            shards_verified = 1
            for r in range(1, world):
                xr = torch.from_numpy(synth.image_batch(batch, 3, size, size, seed=0xC0FFEE + r * batch)).cuda()
                net.forward_device(xr.data_ptr())
                enqueue_results()
                dr, cr = fetch_results()
                if digest(dr, cr) != gathered[r][3]:
                    # reported, not fatal: the scaling numbers of the other ranks stay usable, the line says which shard differed
                    sys.stderr.write("bench: rank %d's last batch differs from its recomputation on rank 0\n" % r)
                    shards_mismatched.append(r)
                else:
                    shards_verified += 1
                del xr
        dist.barrier()

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
        conv_flops_step = sum(per_kernel_flops.values()) / max(args.steps, 1)
        # the CPU legs are timed at N=1 only (rank 0 would otherwise hold the other ranks at the final barrier)
        cpu = cpu_baseline(tmp, wl, wts, args.cpu_iters, args.cpu_legs, args.seed) if world == 1 else None
        equiv = None
        if world == 1 and args.cpu_iters > 0 and is_detector and args.equiv_frames > 0:
            equiv = equivalence_vs_cpu(tmp, name, size, wts, x, dets, args.equiv_frames)
        latency = latency_block(tmp, args.seed, args.latency_iters) if (world == 1 and args.latency_iters > 0) else None
        what = ("forward + region decode + NMS(%.1f) + collect + fetch of the detections" % NMS) if is_detector else \
               "forward (conv trunk, avgpool, softmax) + fetch of the class scores"
        line = {
            "metric": "images/sec YOLOv2 608x608 fp32" if (name == "yolo" and size == 608 and not half) else
                      "images/sec %s %dx%d %s" % (name, size, size, "fp16" if half else "fp32"),
            "value": round(total_images / elapsed, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16" if half else "f32", "data": "synthetic",
            "ranks": comm_ranks, "shards_verified": shards_verified, "shards_mismatched": shards_mismatched, "rank_devices": rank_devices,
            "config": {"workload": "%s %dx%d batch %d per GPU: %s, %s" % (
                           name + ".cfg", size, size, batch, what,
                           "inputs in pinned host memory (PCIe-inclusive)" if args.host_input == "only" else "inputs resident in HBM"),
                       "global_batch": batch * world, "parallelism": "frame-sharded x%d (one weight broadcast)" % world,
                       "tile_choice": "measured at plan time (y2_set_autotune)" if args.autotune else "host cost model",
                       "detect_overlap": detect_overlap,
                       "weight_broadcast": bcast_how, "weight_broadcast_ms": bcast_ms,
                       "gflop_per_image": round(zoo.conv_flops(layers) / 1e9, 3),
                       "conv_ms_per_step": round(conv_ms, 3),
                       "conv_tflops": round(conv_flops_step / (conv_ms * 1e-3) / 1e12, 2) if conv_ms else None,
                       "host_fetch_overlaps_next_forward": True,
                       "detections_in_last_batch": int(np.sum(counts))},
            "roofline": roof, "cpu_baseline": cpu, "host_input": host, "map_equiv_vs_cpu": equiv, "latency": latency,
            "kernels_ms_per_step": {k: round(v / max(args.steps, 1), 3) for k, v in sorted(per_kernel_ms.items())},
            "device": darknet.device_name(),
            # what this box's silicon does: the clock it holds under a full-chip fp32 matrix load (in-kernel stamps, measured
            # right after the timed region); lines of different boxes differ by this much before any code does
            "mfma_f32_clock_ghz": round(darknet.clock_probe(), 3) if args.clock_probe else None,
        }
        print(json.dumps(line))
        sys.stdout.flush()
    net.free()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
