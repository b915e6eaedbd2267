"""MI355X-native YOLOv2 / Darknet forward engine (HIP, gfx950) behind the
reference's C API.  The compute lives in csrc/ (libsr_yolo2.so, a C-ABI
shared library); this package is the thin Python mirror used by tests and
bench.py."""
__all__ = ["zoo", "synth"]
