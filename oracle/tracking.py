"""TEST INFRASTRUCTURE (oracle): restatement of Detector::tracking, the nearest-centre track-id hand-over of
src_yolo2/yolo_v2_class.cpp:251-303, in plain Python for small sequences.  Only tests/ may import this.

State (cpp:33,75-76, hpp:53): per-class id counters starting at 1, a deque of the last `frames_story` result vectors,
newest first.  bbox fields are unsigned ints (hpp:27-33), so centres use integer halving and the distance is the
float square root truncated to unsigned (cpp:275-277).

Parity: the reference function is C++ behind OpenCV-free code but lives in a translation unit that needs the whole
GPU build (cuda_runtime.h, cpp:13-21), which this image cannot compile -- parity unpinned; this restatement was
written from the source text and is what tests/test_tracking.py holds the product's Detector::tracking to."""
from __future__ import annotations

import math
from collections import deque

import numpy as np

UINT_MAX = 0xFFFFFFFF


class Tracker:
    def __init__(self, classes: int):
        self.next_id = [1] * classes                 # cpp:75-76
        self.history = deque()                       # hpp:53 prev_bbox_vec_deque

    def _remember(self, cur, frames_story):
        self.history.appendleft([dict(b) for b in cur])      # cpp:263 / :300 push_front
        if len(self.history) > frames_story:                 # cpp:264 / :301
            self.history.pop()

    def tracking(self, boxes, frames_story: int = 4):
        """boxes: list of dicts x,y,w,h (unsigned), prob, obj_id, track_id (0 on entry).  Returns the list with ids."""
        cur = [dict(b) for b in boxes]
        if not any(len(f) > 0 for f in self.history):        # cpp:255-259
            for b in cur:
                b["track_id"] = self.next_id[b["obj_id"]]
                self.next_id[b["obj_id"]] += 1
            self._remember(cur, frames_story)
            return cur
        dist_vec = [UINT_MAX] * len(cur)                     # cpp:267
        for frame in self.history:                           # newest first
            for old in frame:
                cur_index = -1
                for m, k in enumerate(cur):
                    if old["obj_id"] != k["obj_id"]:
                        continue
                    dx = np.float32(old["x"] + old["w"] // 2) - np.float32(k["x"] + k["w"] // 2)      # cpp:275
                    dy = np.float32(old["y"] + old["h"] // 2) - np.float32(k["y"] + k["h"] // 2)      # cpp:276
                    cur_dist = int(math.sqrt(float(np.float32(dx * dx) + np.float32(dy * dy))))     # cpp:277
                    if cur_dist < 100 and (k["track_id"] == 0 or dist_vec[m] > cur_dist):            # cpp:278
                        dist_vec[m] = cur_dist
                        cur_index = m
                absent = not any(b["track_id"] == old["track_id"] and b["obj_id"] == old["obj_id"] for b in cur)   # cpp:285
                if cur_index >= 0 and absent:                # cpp:288-292
                    c = cur[cur_index]
                    c["track_id"] = old["track_id"]
                    c["w"] = (c["w"] + old["w"]) // 2
                    c["h"] = (c["h"] + old["h"]) // 2
        for b in cur:                                        # cpp:296-298
            if b["track_id"] == 0:
                b["track_id"] = self.next_id[b["obj_id"]]
                self.next_id[b["obj_id"]] += 1
        self._remember(cur, frames_story)
        return cur
