"""The CPU oracle must reproduce, bit for bit, the vectors the compiled
reference produced (tests/golden/*.npz, made by tests/golden/gen_golden.py).
This is what pins the oracle: the reference itself holds no
known-answer vectors for this path."""
import numpy as np
import pytest

from sr_object_detection_amd import synth
from tests.helpers import dense_from_sparse, load_golden, materialize

FAST = ["mini_32_b2", "mini_64_b3", "mini_mfma_64_b2", "mini_res_32_b2", "mini_v1_32_b2", "mini_v1_local_40_b2", "mini_acts_32_b2", "mini_xnor_32_b2", "mini_cls_75_b2", "tiny_yolo_v1_448_b1", "tiny_yolo_voc_416_b1", "tiny_yolo_voc_416_b1_kinect", "darknet19_224_b1"]
SLOW = ["yolo_416_b1", "yolo_608_b1", "yolo9000_96_b1", "yolo9000_96_b1_map",
        # the BASELINE configurations at full size and the dense benchmark frames (the GPU tests of tests/test_gpu_configs.py
        # compare with these fixtures directly; here the oracle port is held to them too)
        "yolo_416_b4", "yolo_608_b4", "yolo9000_544_b2", "darknet19_448_b8", "yolo_608_dense_b2"]


def check_case(oracle, workdir, name):
    g = load_golden(name)
    net, size, batch, seed = str(g["net"]), int(g["size"]), int(g["batch"]), int(g["seed"])
    thresh, nms, gain, use_map = float(g["thresh"]), float(g["nms"]), float(g["head_gain"]), int(g["use_map"])
    cfg, wts, x = materialize(workdir, net, size, batch, seed, gain, bool(use_map))
    if "image_seed" in g:
        x = synth.image_batch(batch, 3, size, size, seed=int(g["image_seed"]))
    assert np.float64(x.astype(np.float64).sum()) == g["input_checksum"]
    on = oracle.OracleNet(cfg, wts)
    out = on.predict(x)
    if use_map:
        assert np.float64(out.astype(np.float64).sum()) == g["out_sum"]
    elif "out_stride" in g:                    # yolo9000 at 544: the fixture keeps every out_stride-th value and the sum
        assert np.array_equal(out.reshape(-1)[::int(g["out_stride"])], g["out"])
        assert np.float64(out.astype(np.float64).sum()) == g["out_sum"]
    else:
        assert out.shape == g["out"].shape
        assert np.array_equal(out, g["out"]), "oracle final tensor differs from the reference's"
    # per-layer statistics written by the reference driver: sum / sum2 / min / max of every layer output
    stats = g["layer_stats"]
    assert stats.shape[0] == on.n
    for i in range(on.n):
        info = on.layer_info(i)
        assert int(stats[i, 8]) == info["outputs"]
        if info["type"] == "cost":
            continue
        o = on.layer_output(i).astype(np.float64)
        np.testing.assert_allclose(o.sum(), stats[i, 12], rtol=1e-8, atol=1e-6)
        assert float(o.min()) == pytest.approx(stats[i, 14], rel=1e-7, abs=1e-30)
        assert float(o.max()) == pytest.approx(stats[i, 15], rel=1e-7, abs=1e-30)
    head = on.layer_info(on.last)["type"]
    if head in ("region", "detection"):
        for b in range(batch):
            boxes, probs = on.region_boxes(b, thresh, use_map=use_map) if head == "region" else on.detection_boxes(b, thresh)
            assert np.array_equal(boxes, g["boxes_%d" % b])
            total, classes = probs.shape
            pre = dense_from_sparse(g["pre_idx_%d" % b], g["pre_val_%d" % b], total, classes)
            assert np.array_equal(probs, pre)
            ncls = 200 if use_map else classes
            post = probs.copy()
            post[:, :ncls] = oracle.do_nms_sort(boxes, np.ascontiguousarray(probs[:, :ncls]), nms)
            gpost = dense_from_sparse(g["post_idx_%d" % b], g["post_val_%d" % b], total, classes)
            assert np.array_equal(post, gpost)
            assert int((post > 0).sum()) == len(g["post_val_%d" % b])
    on.close()


@pytest.mark.parametrize("name", FAST)
def test_oracle_matches_reference_fast(oracle, workdir, name):
    check_case(oracle, workdir, name)


@pytest.mark.slow
@pytest.mark.parametrize("name", SLOW)
def test_oracle_matches_reference_slow(oracle, workdir, name):
    check_case(oracle, workdir, name)
