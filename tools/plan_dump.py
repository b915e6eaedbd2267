#!/usr/bin/env python3
"""The tile plan of the fp32 matrix-core convolutions WITHOUT a GPU: for every conv layer of a zoo network at a given size and
batch, what pick_variant (y2_conv.hip, host arithmetic only) would launch -- kernel name and the candidate list.  Used to
check that a change of the cost model leaves the plan of the headline configuration alone.
    tools/plan_dump.py yolo 608 32 [yolo 416 8 ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sr_object_detection_amd import zoo  # noqa: E402


class Conv(C.Structure):          # include/y2_hip.h: y2h_conv
    _fields_ = [("batch", C.c_int), ("h", C.c_int), ("w", C.c_int), ("c", C.c_int), ("ldx", C.c_int), ("x_halo", C.c_int),
                ("n", C.c_int), ("size", C.c_int), ("stride", C.c_int), ("pad", C.c_int), ("out_h", C.c_int), ("out_w", C.c_int),
                ("ldy", C.c_int), ("fuse_maxpool2", C.c_int), ("batch_normalize", C.c_int), ("activation", C.c_int),
                ("x", C.c_void_p), ("w_packed", C.c_void_p), ("w_ref", C.c_void_p), ("mean", C.c_void_p), ("rinv", C.c_void_p),
                ("scale", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p), ("ws", C.c_void_p), ("ws_bytes", C.c_size_t),
                ("x_f16", C.c_int), ("y_f16", C.c_int), ("alpha", C.c_void_p), ("beta", C.c_void_p),
                ("tile_bm", C.c_int), ("tile_bn", C.c_int), ("ksplit", C.c_int), ("x_nchw", C.c_int)]


def main():
    lib = C.CDLL(os.environ.get("Y2_LIB") or os.path.join(ROOT, "sr_object_detection_amd", "libsr_yolo2.so"))
    lib.y2h_conv_variant.restype = C.c_char_p
    lib.y2h_conv_variant.argtypes = [C.POINTER(Conv), C.c_int]
    lib.y2h_conv_workspace_bytes.restype = C.c_size_t
    args = sys.argv[1:]
    for k in range(0, len(args), 3):
        net, size, batch = args[k], int(args[k + 1]), int(args[k + 2])
        layers = zoo.resolve(net, size)
        print("%s %d b%d" % (net, size, batch))
        for i, l in enumerate(layers):
            if l["type"] != "convolutional" or l["c"] % 16:
                continue
            pool = i + 1 < len(layers) and layers[i + 1]["type"] == "maxpool" and layers[i + 1]["size"] == 2 and layers[i + 1]["stride"] == 2
            d = Conv(batch=batch, h=l["h"], w=l["w"], c=l["c"], ldx=l["c"], n=l["filters"], size=l["size"], stride=l["stride"],
                     pad=l["size"] // 2 if l["pad"] else 0, out_h=l["out_h"], out_w=l["out_w"], ldy=l["filters"],
                     fuse_maxpool2=1 if pool else 0, batch_normalize=l.get("batch_normalize", 0), activation=1,
                     x=0x10000, w_packed=0x20000, y=0x30000, bias=0x40000)
            name = lib.y2h_conv_variant(C.byref(d), 0).decode()
            ws = lib.y2h_conv_workspace_bytes(C.byref(d))
            print("  L%-2d %3dx%-3d c%-4d n%-5d k%d%s  %-34s ws %d" % (i, l["h"], l["w"], l["c"], l["filters"], l["size"],
                                                                  "+p" if pool else "  ", name, ws))


if __name__ == "__main__":
    main()
