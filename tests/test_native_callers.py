"""Source-level drop-in check: a C program written like the reference's Kinect caller and a C++ program
written like yolo_console_dll.cpp are compiled against include/ (with the reference's header names) and
linked to libsr_yolo2.so; on the GPU their output is compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

from tests.helpers import dense_from_sparse, load_golden, materialize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "sr_object_detection_amd")
INC = os.path.join(ROOT, "include")
SRC = os.path.join(ROOT, "tests", "native")


def build(workdir, name, compiler, src, extra=()):
    exe = os.path.join(workdir, name)
    cmd = [compiler, "-O1", "-I", INC, os.path.join(SRC, src), "-o", exe, "-L", LIBDIR, "-lsr_yolo2",
           "-Wl,-rpath," + LIBDIR, "-lm"] + list(extra)
    subprocess.check_call(cmd)
    return exe


def test_native_callers_compile_and_link(workdir):
    """No GPU needed: the reference-style callers build against our headers and resolve every symbol."""
    build(workdir, "kinect_like", "gcc", "kinect_like.c")
    build(workdir, "detector_cpp", "g++", "detector_cpp.cpp", ["-std=c++11"])
    # the OPENCV surface of yolo_v2_class.hpp (detect(cv::Mat), detect_resized, mat_to_image_resize, mat_to_image) against
    # a stand-in <opencv2/opencv.hpp>: the calls of yolo_console_dll.cpp:137,148 compile and link
    build(workdir, "console_dll_like", "g++", "console_dll_like.cpp", OPENCV_FLAGS)


OPENCV_FLAGS = ["-std=c++11", "-DOPENCV", "-I", os.path.join(SRC, "opencv_stub")]


def write_frame(path, chw):
    with open(path, "wb") as f:
        np.array(chw.shape, dtype=np.int32).tofile(f)
        np.ascontiguousarray(chw, dtype=np.float32).tofile(f)


@pytest.mark.gpu
def test_kinect_style_c_caller_matches_oracle(oracle, workdir):
    g = load_golden("tiny_yolo_voc_416_b1_kinect")
    cfg, wts, x = materialize(workdir, "tiny-yolo-voc", 416, 1, int(g["seed"]), float(g["head_gain"]))
    frame = os.path.join(workdir, "frame416.bin")
    write_frame(frame, x[0])
    exe = build(workdir, "kinect_like", "gcc", "kinect_like.c")
    thresh = float(g["thresh"])
    out = subprocess.run([exe, cfg, wts, frame, repr(thresh)], capture_output=True, text=True, timeout=300, check=True).stdout
    lines = out.strip().splitlines()
    objs = [l.split() for l in lines if l.startswith("OBJ ")]
    total, classes = 845, 20
    gpost = dense_from_sparse(g["post_idx_0"], g["post_val_0"], total, classes)
    want = oracle.test_detector_objects(g["boxes_0"], gpost, thresh)
    assert len(objs) == len(want) > 0
    for o, w in zip(objs, want):
        assert int(o[1]) == int(w[5]) and o[7] == "class%d" % int(w[5])
        got = np.array([float(v) for v in o[2:7]])
        assert np.abs(got[:2] - w[:2]).max() < 1e-4 and np.abs(got[4] - w[4]) < 1e-4
        assert np.allclose([float(v) for v in o[8:11]], w[6:9])
    legacy = [l for l in lines if l.startswith("LEGACY")][0].split()
    assert int(legacy[1]) == len(want) and int(legacy[3]) == 845 * 25


@pytest.mark.gpu
def test_cpp_detector_matches_oracle(oracle, workdir):
    g = load_golden("yolo_416_b1")
    cfg, wts, x = materialize(workdir, "yolo", 416, 1, int(g["seed"]), float(g["head_gain"]))
    frame = os.path.join(workdir, "frame416y.bin")
    write_frame(frame, x[0])
    exe = build(workdir, "detector_cpp", "g++", "detector_cpp.cpp", ["-std=c++11"])
    thresh = float(g["thresh"])
    out = subprocess.run([exe, cfg, wts, frame, repr(thresh)], capture_output=True, text=True, timeout=300, check=True).stdout
    lines = out.strip().splitlines()
    boxes = [l.split()[1:] for l in lines if l.startswith("BOX ")]
    total, classes = 845, 80
    gpost = dense_from_sparse(g["post_idx_0"], g["post_val_0"], total, classes)
    want = oracle.detector_bboxes(g["boxes_0"], gpost, thresh, 416, 416)
    assert len(boxes) == len(want) > 0
    for b, w in zip(boxes, want):
        # unsigned pixel coordinates: a 1e-6 difference can move a truncation by one pixel
        assert all(abs(int(b[i]) - int(w[k])) <= 1 for i, k in enumerate(("x", "y", "w", "h")))
        assert abs(float(b[4]) - float(w["prob"])) < 1e-4 and int(b[5]) == int(w["obj_id"]) and int(b[6]) == 0
    assert "TRACK 1" in lines
    frame_line = [l for l in lines if l.startswith("FRAME")][0].split()
    assert frame_line[1] == "1" and int(frame_line[2]) > 0      # detect_frame(u8) == host conversion + detect(image_t)
    assert any(l.startswith("THROW file not found") for l in lines)
    mean = [l for l in lines if l.startswith("MEAN")][0].split()
    assert int(mean[3]) == len(want)         # third use_mean call: the average of three identical frames


@pytest.mark.gpu
def test_console_dll_style_opencv_calls(workdir):
    """yolo_console_dll.cpp:137,148: mat_to_image_resize (8-bit BGR frame -> planar RGB floats, v / 255.) followed by
    detect_resized equals detect(image_t) on the same pixels; detect(cv::Mat) on a frame of twice the size returns the same
    objects with boxes in the frame's pixels; an empty Mat throws as the reference does (yolo_v2_class.hpp:59-92)"""
    g = load_golden("yolo_416_b1")
    cfg, wts, x = materialize(workdir, "yolo", 416, 1, int(g["seed"]), float(g["head_gain"]))
    frame = os.path.join(workdir, "frame416c.bin")
    write_frame(frame, x[0])
    exe = build(workdir, "console_dll_like", "g++", "console_dll_like.cpp", OPENCV_FLAGS)
    out = subprocess.run([exe, cfg, wts, frame, repr(float(g["thresh"]))], capture_output=True, text=True, timeout=300, check=True).stdout
    lines = out.strip().splitlines()
    assert "IMAGE 1" in lines
    resized = [l for l in lines if l.startswith("RESIZED")][0].split()
    assert resized[1] == "1" and int(resized[2]) > 0
    mat = [l for l in lines if l.startswith("MAT")][0].split()
    assert mat[1] == "1" and int(mat[2]) == int(resized[2])
    assert any(l.startswith("TRACKED %s" % resized[2]) for l in lines)
    assert "THROW Image is empty" in lines
