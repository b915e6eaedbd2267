"""Multi-GPU path on the one-GPU box: weight replication through the arena, the RCCL C-ABI, bench.py's own launcher
with two ranks sharing the GPU, and the pinned host feed.

Replaces / extends the reference's host-staged distribute_weights (src_yolo2/network_kernels.cu:240-250) and the
per-call cudaMalloc + H2D + cudaFree of network_predict_gpu (:392-405)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from sr_object_detection_amd import darknet, synth
from tests.helpers import materialize

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _replica(cfg, src, half):
    """what a rank other than 0 does: parse the cfg, NO load_weights, take the arena, receive the bytes, declare it resident"""
    L = darknet.lib()
    net = darknet.Network.parse_network_cfg(cfg)
    net.set_half(half)
    ptr, nbytes = net.weights_arena()
    sptr, sbytes = src.weights_arena()
    assert nbytes == sbytes and ptr != sptr
    assert L.y2h_memcpy_d2d(ptr, sptr, nbytes, None) == 0
    assert L.y2h_device_sync() == 0
    net.weights_resident()
    return net


@pytest.mark.parametrize("half", [False, True], ids=["fp32", "fp16"])
def test_replicated_arena_gives_bitwise_equal_outputs(workdir, monkeypatch, half):
    monkeypatch.setenv("Y2_CONV_KSPLIT", "1")          # same kernels at batch 2 and batch 1: the comparison is bitwise
    monkeypatch.setenv("Y2_CONV_TILE", "64x64")
    cfg, wts, x = materialize(workdir, "mini-mfma", 64, 2, 7)
    a = darknet.Network.parse_network_cfg(cfg)
    a.load_weights(wts)
    a.set_half(half)
    a.prepare()
    want = a.network_predict(x).copy()
    b = _replica(cfg, a, half)
    got = b.network_predict(x)
    assert np.array_equal(got, want)
    assert [b.layer_kernel(i) for i in range(b.n)] == [a.layer_kernel(i) for i in range(a.n)]
    # batch changes keep the layout: the replica stays valid
    b.set_batch_network(1)
    assert np.array_equal(b.network_predict(x[1]), want.reshape(2, -1)[1])
    a.free()
    b.free()


def test_mode_switch_on_a_replica_fails_loudly_then_recovers(workdir):
    """a replica holds no host weights: a plan whose arena layout differs (strict -> reference-layout weights, fp16 ->
    half weights) must not silently run on the old bytes"""
    cfg, wts, x = materialize(workdir, "mini-mfma", 64, 2, 7)
    a = darknet.Network.parse_network_cfg(cfg)
    a.load_weights(wts)
    a.prepare()
    b = _replica(cfg, a, False)
    b.network_predict(x)
    b.set_half(True)
    with pytest.raises(darknet.Y2Error, match="arena was filled from outside"):
        b.network_predict(x)
    # the documented recovery: request the arena again for the new plan and replicate again
    a.set_half(True)
    a.prepare()
    want = a.network_predict(x).copy()
    L = darknet.lib()
    ptr, nbytes = b.weights_arena()
    sptr, sbytes = a.weights_arena()
    assert nbytes == sbytes
    assert L.y2h_memcpy_d2d(ptr, sptr, nbytes, None) == 0 and L.y2h_device_sync() == 0
    b.weights_resident()
    assert np.array_equal(b.network_predict(x), want)
    a.free()
    b.free()


def test_rccl_c_abi_broadcast_single_rank(workdir):
    """y2_comm_unique_id / y2_comm_init_rank / y2_broadcast_weights / y2_comm_destroy against the real RCCL with a
    one-rank communicator (the box has one GPU and RCCL refuses two ranks on one device): symbol binding, communicator
    creation, the in-place collective on the arena pointer on the engine's stream.  The N-rank case runs in
    bench.py --bcast c-abi on a multi-GPU node."""
    cfg, wts, x = materialize(workdir, "mini-mfma", 64, 2, 7)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    want = net.network_predict(x).copy()
    assert "rccl" in darknet.comm_library()
    uid = darknet.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    comm = darknet.comm_init_rank(1, uid, 0, 0)
    assert darknet.comm_count(comm) == (1, 0)                  # ncclCommCount / ncclCommUserRank
    net.broadcast_weights(comm, 0)                             # layout handshake (broadcast + all-reduce) + the arena
    assert np.array_equal(net.network_predict(x), want)
    sig, nbytes = net.weights_layout()
    assert sig != 0 and nbytes == net.weights_arena()[1]
    with pytest.raises(darknet.Y2Error, match="root"):
        net.broadcast_weights(comm, 3)
    darknet.comm_destroy(comm)
    net.free()


def test_bench_launches_two_ranks_itself_and_rank1_matches_a_single_process(workdir, tmp_path):
    """`python bench.py --gpus 2` with WORLD_SIZE unset: the parent spawns the ranks (both on this box's one GPU, gloo
    + host-staged broadcast as the rehearsal transport), rank 1 gets its weights only through the arena broadcast and
    processes frames 8..15; its detections must equal a single-process run of those frames."""
    env = dict(os.environ, Y2_BENCH_BACKEND="gloo", Y2_BENCH_SHARE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    dump = str(tmp_path / "dets")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "yolo416_b8", "--steps", "3",
                        "--warmup", "1", "--cpu-iters", "0", "--host-input", "off", "--autotune", "0", "--dump-dets", dump],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 16 and line["value"] > 0
    assert "broadcast" in line["config"]["weight_broadcast"]
    assert line["ranks"] == 2 and line["shards_verified"] == 2 and len(line["rank_devices"]) == 2
    # single process, frames 8..15 (global image index = rank * batch + i, seed 0xC0FFEE + index)
    import bench
    cfg = bench.write_cfg(str(tmp_path), "yolo", 416, 8)
    from sr_object_detection_amd import zoo
    wts = str(tmp_path / "w.weights")
    synth.write_weights(wts, zoo.resolve("yolo", 416), 31)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    x = synth.image_batch(8, 3, 416, 416, seed=0xC0FFEE + 8)
    dets, counts = net.detect(x, bench.THRESH, bench.NMS)
    got = np.load(dump + ".rank1.npz")
    assert np.array_equal(got["counts"], counts) and int(counts.sum()) > 0
    for b in range(8):
        assert np.array_equal(got["dets_%d" % b], dets[b]), "frame %d" % (8 + b)
    r0 = np.load(dump + ".rank0.npz")
    assert not np.array_equal(r0["counts"], counts) or not np.array_equal(r0["dets_0"], dets[0])     # other frames
    net.free()


def test_bench_five_rank_rehearsal_on_one_gpu(tmp_path):
    """the launcher, the ports, the rank-0 relay, the arena broadcast, the per-rank frame shards and the teardown at the
    largest rank count one box allows (the pool's process guard admits six processes on a card and this test process is
    one of them; the driver's real run is eight ranks on eight GPUs over RCCL): every rank's last batch must equal its
    recomputation on rank 0, bit for bit (shards_verified), and the line must report the five ranks"""
    env = dict(os.environ, Y2_BENCH_BACKEND="gloo", Y2_BENCH_SHARE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "5", "--workload", "yolo416_b8", "--steps", "3",
                        "--warmup", "1", "--cpu-iters", "0", "--host-input", "off", "--autotune", "0"],
                       env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 5 and line["ranks"] == 5 and line["config"]["global_batch"] == 40
    assert line["shards_verified"] == 5 and len(line["rank_devices"]) == 5
    assert line["value"] > 0 and line["scaling"] == "weak"


def test_pinned_feed_matches_resident_and_overlaps(workdir):
    """y2_feed_*: batches uploaded from pinned slots on the copy stream give the same detections as network_predict,
    slot after slot, for float and for u8 camera frames"""
    cfg, wts, _ = materialize(workdir, "tiny-yolo-voc", 416, 2, 21)
    net = darknet.Network.parse_network_cfg(cfg)
    net.load_weights(wts)
    batches = [synth.image_batch(2, 3, 416, 416, seed=500 + 10 * k) for k in range(5)]
    want = [net.detect(b, 0.2, 0.4) for b in batches]
    net.feed_open(2)
    slots = [net.feed_host(s) for s in range(2)]
    for k, b in enumerate(batches):
        s = k & 1
        net.feed_wait_host(s)
        slots[s][:b.size] = b.reshape(-1)
        net.feed_submit(s)
        net.feed_forward(s)
        if k + 1 < len(batches):                 # the next batch goes up while this one computes
            o = (k + 1) & 1
            net.feed_wait_host(o)
            slots[o][:b.size] = batches[k + 1].reshape(-1)
        dets, counts = net.detect_resident(0.2, 0.4)
        assert np.array_equal(counts, want[k][1]) and int(counts.sum()) > 0
        for i in range(2):
            assert np.array_equal(dets[i], want[k][0][i])
    net.feed_close()
    # u8 frames: [batch][h][w][3] bytes in the slot, conversion + forward on the device
    frames = (synth.splitmix64(77, 2 * 416 * 416 * 3) & np.uint64(255)).astype(np.uint8).reshape(2, 416, 416, 3)
    wd, wc = net.detect_u8(frames, 0.2, 0.4)
    net.feed_open(2, frames.nbytes)
    net.feed_host(1, np.uint8)[:frames.size] = frames.reshape(-1)
    net.feed_submit(1, frames.nbytes)
    net.feed_forward_u8(1, 416, 416, 3)
    dets, counts = net.detect_resident(0.2, 0.4)
    assert np.array_equal(counts, wc)
    for i in range(2):
        assert np.array_equal(dets[i], wd[i])
    with pytest.raises(darknet.Y2Error, match="float batch needs"):
        net.feed_forward(0)                      # the u8 slots are too small for float frames
    net.feed_close()
    net.free()
