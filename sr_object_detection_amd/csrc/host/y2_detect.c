/*
 * Region-head hand-off on the host side: the reference's CPU functions
 * get_region_boxes (src_yolo2/region_layer.c:328-379), do_nms_sort / do_nms
 * (src_yolo2/box.c:249-298), the Kinect entry test_detector_img
 * (src_yolo2/detector.c:558-598 + draw_detections_test image.c:662-738) and
 * the small helpers around them -- all backed by the device kernels of
 * y2_detect.hip.  The legacy signatures take caller-owned HOST arrays
 * (probs is float*[total], boxes is box[total]); they are honoured by staging
 * through HBM, so there is no CPU implementation of decode or NMS in this
 * library.  y2_detect*() is the fused, HBM-resident form the Detector class
 * and bench.py use: forward -> decode -> NMS -> compaction with only the
 * compact detection records crossing PCIe.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "y2_internal.h"

#define HIPCALL(expr) do { int rc_ = (expr); if (rc_ != 0) { y2_fail("%s failed (%d): %s", #expr, rc_, y2h_last_error()); return; } } while (0)
/* inside functions that own scratch buffers: report, then release them at `cleanup:` */
#define HIPCALL_C(expr) do { int rc_ = (expr); if (rc_ != 0) { y2_fail("%s failed (%d): %s", #expr, rc_, y2h_last_error()); goto cleanup; } } while (0)
#define HIPCALL_I(expr) do { int rc_ = (expr); if (rc_ != 0) { y2_fail("%s failed (%d): %s", #expr, rc_, y2h_last_error()); return -1; } } while (0)

static y2_ldev *ld_of(const layer *l) { return (y2_ldev *)l->dev; }

/* get_region_boxes: l.output is a HOST pointer the caller may have replaced
 * (yolo_v2_class.cpp:211 substitutes the 3-frame mean).  When it still is the
 * engine's own output buffer the tensor is already in HBM and is not re-sent. */
void get_region_boxes(layer l, int w, int h, float thresh, float **probs, box *boxes, int only_objectness, int *map)
{
    y2_ldev *d = ld_of(&l);
    y2_engine *e;
    y2h_decode q;
    int total = l.w * l.h * l.n, i;
    size_t pred_floats = (size_t)total * (l.classes + 5);
    float *d_pred, *d_tmp_pred = NULL, *h_probs = NULL;
    int *d_tmp_map = NULL;
    if (l.type != REGION || !d || !d->eng || !d->eng->built) { y2_fail("get_region_boxes: layer is not a prepared region layer"); return; }
    if (!l.output) { y2_fail("get_region_boxes: l.output is NULL"); return; }
    e = d->eng;
    HIPCALL(y2h_set_device(e->device));
    if (e->det_pending == 1 && e->det_overlap) HIPCALL(y2h_event_sync(e->ev_det));   /* the shared decode scratch is in use on det_stream */
    memset(&q, 0, sizeof q);
    q.batch = 1; q.w = l.w; q.h = l.h; q.num = l.n; q.classes = l.classes;
    q.img_w = w; q.img_h = h; q.thresh = thresh; q.only_objectness = only_objectness; q.classfix = l.classfix;
    q.anchors = d->d_anchors;
    q.tree_parent = l.softmax_tree ? d->d_tree_parent : NULL;
    q.tree_order = d->d_tree_order; q.tree_level_off = d->d_tree_loff; q.tree_levels = d->tree_levels;
    if (l.output >= e->h_out && l.output < e->h_out + e->out_floats && (size_t)(l.output - e->h_out) % l.outputs == 0) {
        d_pred = d->d_region + (l.output - e->h_out);           /* batch item (l.output - h_out)/outputs */
    } else {
        HIPCALL_C(y2h_malloc((void **)&d_tmp_pred, pred_floats * sizeof(float)));
        HIPCALL_C(y2h_memcpy_h2d(d_tmp_pred, l.output, pred_floats * sizeof(float), e->stream));
        d_pred = d_tmp_pred;
    }
    if (map && l.softmax_tree) {
        if (map == l.map && d->d_map) q.map = d->d_map;
        else {
            HIPCALL_C(y2h_malloc((void **)&d_tmp_map, 200 * sizeof(int)));
            HIPCALL_C(y2h_memcpy_h2d(d_tmp_map, map, 200 * sizeof(int), e->stream));
            q.map = d_tmp_map;
        }
    }
    q.pred = d_pred; q.boxes = e->d_boxes; q.probs = e->d_probs;
    if (q.map) HIPCALL_C(y2h_memset(e->d_probs, 0, (size_t)total * l.classes * sizeof(float), e->stream));
    HIPCALL_C(y2h_region_boxes(&q, e->stream));
    h_probs = malloc((size_t)total * l.classes * sizeof(float));
    HIPCALL_C(y2h_memcpy_d2h(boxes, e->d_boxes, (size_t)total * sizeof(box), e->stream));
    HIPCALL_C(y2h_memcpy_d2h(h_probs, e->d_probs, (size_t)total * l.classes * sizeof(float), e->stream));
    if (l.softmax_tree)      /* the reference rewrites the class scores in l.output in place (region_layer.c:350) */
        HIPCALL_C(y2h_memcpy_d2h(l.output, d_pred, pred_floats * sizeof(float), e->stream));
    HIPCALL_C(y2h_stream_sync(e->stream));
    {
        int ncopy = (q.map) ? 200 : l.classes;               /* with a map only 200 entries per row are written */
        for (i = 0; i < total; ++i) memcpy(probs[i], h_probs + (size_t)i * l.classes, ncopy * sizeof(float));
        if (q.map && only_objectness) for (i = 0; i < total; ++i) probs[i][0] = h_probs[(size_t)i * l.classes];
    }
cleanup:
    free(h_probs);
    y2h_free(d_tmp_pred);
    y2h_free(d_tmp_map);
}

/* detection_layer.c:222-251 get_detection_boxes with the reference's host-array signature, decoded on the device.
 * l.output may be the engine's own output buffer (already in HBM) or a caller-provided tensor (uploaded). */
void get_detection_boxes(layer l, int w, int h, float thresh, float **probs, box *boxes, int only_objectness)
{
    y2_ldev *d = ld_of(&l);
    y2_engine *e;
    int total = l.side * l.side * l.n, i;
    float *d_pred, *d_tmp = NULL, *h_probs = NULL;
    if (l.type != DETECTION || !d || !d->eng || !d->eng->built) { y2_fail("get_detection_boxes: layer is not a prepared detection layer"); return; }
    if (!l.output) { y2_fail("get_detection_boxes: l.output is NULL"); return; }
    e = d->eng;
    if (!e->d_boxes || e->det_total < total) { y2_fail("get_detection_boxes: the detection layer is not the network's output layer"); return; }
    HIPCALL(y2h_set_device(e->device));
    if (l.output >= e->h_out && l.output < e->h_out + e->out_floats && (size_t)(l.output - e->h_out) % l.outputs == 0)
        d_pred = d->d_flat + (l.output - e->h_out);
    else {
        HIPCALL_C(y2h_malloc((void **)&d_tmp, (size_t)l.outputs * sizeof(float)));
        HIPCALL_C(y2h_memcpy_h2d(d_tmp, l.output, (size_t)l.outputs * sizeof(float), e->stream));
        d_pred = d_tmp;
    }
    HIPCALL_C(y2h_detection_boxes(d_pred, (long)l.outputs, 1, l.side, l.n, l.classes, l.sqrt, w, h, thresh, only_objectness,
                                e->d_boxes, e->d_probs, e->stream));
    h_probs = malloc((size_t)total * l.classes * sizeof(float));
    HIPCALL_C(y2h_memcpy_d2h(boxes, e->d_boxes, (size_t)total * sizeof(box), e->stream));
    HIPCALL_C(y2h_memcpy_d2h(h_probs, e->d_probs, (size_t)total * l.classes * sizeof(float), e->stream));
    HIPCALL_C(y2h_stream_sync(e->stream));
    for (i = 0; i < total; ++i) memcpy(probs[i], h_probs + (size_t)i * l.classes, l.classes * sizeof(float));
cleanup:
    free(h_probs);
    y2h_free(d_tmp);
}

/* scratch for the array-in/array-out NMS entry points (not thread-safe, like the reference) */
static struct { float *d_boxes, *d_probs, *d_probs_in; int *d_counts; size_t boxes_cap, probs_cap, counts_cap; y2h_stream stream; int device; } g_nms = {0, 0, 0, 0, 0, 0, 0, 0, -1};

static int nms_scratch(size_t nboxes, size_t nprobs, size_t nclasses)
{
    int dev = 0;
    if (y2h_device_count() <= 0) { y2_fail("do_nms: no HIP device visible and this library has no CPU path"); return -1; }
    HIPCALL_I(y2h_get_device(&dev));
    if (g_nms.device != dev) {
        g_nms.d_boxes = g_nms.d_probs = g_nms.d_probs_in = NULL; g_nms.d_counts = NULL;
        g_nms.boxes_cap = g_nms.probs_cap = g_nms.counts_cap = 0; g_nms.stream = NULL;   /* per-device scratch */
        g_nms.device = dev;
    }
    if (!g_nms.stream) HIPCALL_I(y2h_stream_create(&g_nms.stream));
    if (nboxes > g_nms.boxes_cap) { y2h_free(g_nms.d_boxes); HIPCALL_I(y2h_malloc((void **)&g_nms.d_boxes, nboxes * sizeof(float))); g_nms.boxes_cap = nboxes; }
    if (nprobs > g_nms.probs_cap) {
        y2h_free(g_nms.d_probs); y2h_free(g_nms.d_probs_in);
        HIPCALL_I(y2h_malloc((void **)&g_nms.d_probs, nprobs * sizeof(float)));
        HIPCALL_I(y2h_malloc((void **)&g_nms.d_probs_in, nprobs * sizeof(float)));
        g_nms.probs_cap = nprobs;
    }
    if (nclasses > g_nms.counts_cap) {
        y2h_free(g_nms.d_counts);
        HIPCALL_I(y2h_malloc((void **)&g_nms.d_counts, nclasses * sizeof(int)));
        g_nms.counts_cap = nclasses;
    }
    return 0;
}

static void nms_host(box *boxes, float **probs, int total, int classes, float thresh, int sorted)
{
    float *flat;
    int i;
    if (total <= 0 || classes <= 0) return;
    if (nms_scratch((size_t)total * 4, (size_t)total * classes, (size_t)classes) != 0) return;
    flat = malloc((size_t)total * classes * sizeof(float));
    for (i = 0; i < total; ++i) memcpy(flat + (size_t)i * classes, probs[i], classes * sizeof(float));
    HIPCALL(y2h_memcpy_h2d(g_nms.d_boxes, boxes, (size_t)total * sizeof(box), g_nms.stream));
    HIPCALL(y2h_memcpy_h2d(g_nms.d_probs, flat, (size_t)total * classes * sizeof(float), g_nms.stream));
    if (sorted) {
        HIPCALL(y2h_memcpy_d2d(g_nms.d_probs_in, g_nms.d_probs, (size_t)total * classes * sizeof(float), g_nms.stream));
        HIPCALL(y2h_nms_sort(g_nms.d_boxes, g_nms.d_probs_in, g_nms.d_probs, 1, total, classes, classes, thresh, g_nms.d_counts, g_nms.stream));
    }
    else HIPCALL(y2h_nms(g_nms.d_boxes, g_nms.d_probs, 1, total, classes, classes, thresh, g_nms.stream));
    HIPCALL(y2h_memcpy_d2h(flat, g_nms.d_probs, (size_t)total * classes * sizeof(float), g_nms.stream));
    HIPCALL(y2h_stream_sync(g_nms.stream));
    for (i = 0; i < total; ++i) memcpy(probs[i], flat + (size_t)i * classes, classes * sizeof(float));
    free(flat);
}

void do_nms_sort(box *boxes, float **probs, int total, int classes, float thresh) { nms_host(boxes, probs, total, classes, thresh, 1); }
void do_nms(box *boxes, float **probs, int total, int classes, float thresh) { nms_host(boxes, probs, total, classes, thresh, 0); }

/* box.c:67-97.  A scalar helper on host values (two boxes in, one float out); kept on the
 * host because a kernel launch per call would be absurd.  Same expression order as the reference. */
static float overlap1(float x1, float w1, float x2, float w2)
{
    float l1 = x1 - w1 / 2, l2 = x2 - w2 / 2;
    float left = l1 > l2 ? l1 : l2;
    float r1 = x1 + w1 / 2, r2 = x2 + w2 / 2;
    float right = r1 < r2 ? r1 : r2;
    return right - left;
}
float box_iou(box a, box b)
{
    float w = overlap1(a.x, a.w, b.x, b.w), h = overlap1(a.y, a.h, b.y, b.h);
    float inter = (w < 0 || h < 0) ? 0 : w * h;
    float uni = a.w * a.h + b.w * b.h - inter;
    return inter / uni;
}
box float_to_box(float *f) { box b; b.x = f[0]; b.y = f[1]; b.w = f[2]; b.h = f[3]; return b; }

int max_index(float *a, int n)               /* utils.c:533-545 */
{
    int i, mi = 0;
    float m;
    if (n <= 0) return -1;
    m = a[0];
    for (i = 1; i < n; ++i) if (a[i] > m) { m = a[i]; mi = i; }
    return mi;
}

void top_k(float *a, int n, int k, int *index)   /* utils.c:179-193 */
{
    int i, j;
    for (j = 0; j < k; ++j) index[j] = -1;
    for (i = 0; i < n; ++i) {
        int curr = i;
        for (j = 0; j < k && curr >= 0; ++j)
            if (index[j] < 0 || a[curr] > a[index[j]]) { int s = curr; curr = index[j]; index[j] = s; }
    }
}

void mean_arrays(float **a, int n, int els, float *avg)   /* utils.c:420-432 */
{
    int i, j;
    memset(avg, 0, (size_t)els * sizeof(float));
    for (j = 0; j < n; ++j) for (i = 0; i < els; ++i) avg[i] += a[j][i];
    for (i = 0; i < els; ++i) avg[i] /= n;
}

float get_color(int c, int x, int max)       /* image.c:33-42 */
{
    static const float colors[6][3] = { {1,0,1}, {0,0,1}, {0,1,1}, {0,1,0}, {1,1,0}, {1,0,0} };
    float ratio = ((float)x / max) * 5;
    int i = (int)floor(ratio), j = (int)ceil(ratio);
    ratio -= i;
    return (1 - ratio) * colors[i][c] + ratio * colors[j][c];
}

image make_image(int w, int h, int c)        /* image.c:1436 */
{
    image im;
    im.w = w; im.h = h; im.c = c;
    im.data = calloc((size_t)w * h * c > 0 ? (size_t)w * h * c : 1, sizeof(float));
    return im;
}
void free_image(image m) { free(m.data); }

/* scratch of the image-in/image-out helpers (resize_image, letterbox_image): one stream and three grow-only
 * device buffers per device, so that a per-frame caller pays no hipMalloc/hipFree (not thread-safe, like the
 * reference's own static buffers) */
static struct { float *d_src, *d_tmp, *d_dst; size_t src_cap, tmp_cap, dst_cap; y2h_stream stream; int device; } g_img = {0, 0, 0, 0, 0, 0, 0, -1};

static int img_scratch(const char *who, size_t ns, size_t nt, size_t nd)
{
    int dev = 0;
    if (y2h_device_count() <= 0) { y2_fail("%s: no HIP device visible and this library has no CPU path", who); return -1; }
    if (gpu_index >= 0) y2h_set_device(gpu_index);
    HIPCALL_I(y2h_get_device(&dev));
    if (g_img.device != dev) {
        g_img.d_src = g_img.d_tmp = g_img.d_dst = NULL;
        g_img.src_cap = g_img.tmp_cap = g_img.dst_cap = 0; g_img.stream = NULL;      /* per-device scratch */
        g_img.device = dev;
    }
    if (!g_img.stream) HIPCALL_I(y2h_stream_create(&g_img.stream));
    if (ns > g_img.src_cap) { y2h_free(g_img.d_src); g_img.d_src = NULL; g_img.src_cap = 0; HIPCALL_I(y2h_malloc((void **)&g_img.d_src, ns * 4)); g_img.src_cap = ns; }
    if (nt > g_img.tmp_cap) { y2h_free(g_img.d_tmp); g_img.d_tmp = NULL; g_img.tmp_cap = 0; HIPCALL_I(y2h_malloc((void **)&g_img.d_tmp, nt * 4)); g_img.tmp_cap = nt; }
    if (nd > g_img.dst_cap) { y2h_free(g_img.d_dst); g_img.d_dst = NULL; g_img.dst_cap = 0; HIPCALL_I(y2h_malloc((void **)&g_img.d_dst, nd * 4)); g_img.dst_cap = nd; }
    return 0;
}

image resize_image(image im, int w, int h)   /* image.c:1950-1992, on the device */
{
    image out = make_image(w, h, im.c);
    size_t ns = (size_t)im.w * im.h * im.c, nt = (size_t)w * im.h * im.c, nd = (size_t)w * h * im.c;
    if (!im.data || ns == 0 || nd == 0) { y2_fail("resize_image: empty image"); return out; }
    if (img_scratch("resize_image", ns, nt, nd) != 0) return out;
    if (y2h_memcpy_h2d(g_img.d_src, im.data, ns * 4, g_img.stream) ||
        y2h_resize_chw(g_img.d_src, im.c, im.h, im.w, g_img.d_tmp, g_img.d_dst, h, w, g_img.stream) ||
        y2h_memcpy_d2h(out.data, g_img.d_dst, nd * 4, g_img.stream) || y2h_stream_sync(g_img.stream))
        y2_fail("resize_image: %s", y2h_last_error());
    return out;
}

/* ------------------------------------------------------------------ */
/* fused, HBM-resident detection                                       */
/* ------------------------------------------------------------------ */
/* Decode + NMS + compaction are enqueued by detect_enqueue (with the D2H copies of the counts and, when they are
 * small, of the record blocks, followed by an event); detect_fetch waits for that event only -- not for whatever the
 * caller has enqueued behind it, e.g. the next batch's forward pass -- and unpacks.  y2_detect_resident = both. */
static int detect_enqueue(network net, float *d_pred, float thresh, float nms, int img_w, int img_h)
{
    y2_engine *e = y2_engine_of(&net);
    layer *l;
    y2_ldev *d;
    y2h_decode q;
    const float *final_probs;
    int b, keep;
    y2h_stream ds;
    if (!e || !e->built) { y2_fail("y2_detect_resident: run a forward first"); return -1; }
    l = &net.layers[e->out_layer];
    d = ld_of(l);
    if (l->type != REGION && l->type != DETECTION) { y2_fail("y2_detect_resident: the network does not end in a region or detection layer"); return -1; }
    HIPCALL_I(y2h_set_device(e->device));
    if (e->det_pending == 1) HIPCALL_I(y2h_event_sync(e->ev_det));      /* an enqueue that was never fetched: its scratch is ours again */
    e->det_pending = 0;
    if (!e->ev_det) HIPCALL_I(y2h_event_create(&e->ev_det));
    /* y2_set_detect_overlap (region heads, resident output): the chain runs on its own stream behind an event of the
     * forward pass that produced d_region; the next forward pass waits for ev_det before its region layer writes
     * d_region again (enqueue_forward), and nothing else touches the chain's scratch until y2_detect_fetch. */
    ds = e->stream;
    if (e->det_overlap && !d_pred && l->type == REGION) {
        if (!e->det_stream) HIPCALL_I(y2h_stream_create(&e->det_stream));
        if (!e->ev_fwd) HIPCALL_I(y2h_event_create(&e->ev_fwd));
        HIPCALL_I(y2h_event_record(e->ev_fwd, e->stream));
        HIPCALL_I(y2h_stream_wait_event(e->det_stream, e->ev_fwd));
        ds = e->det_stream;
    }
    if (!d_pred) d_pred = l->type == REGION ? d->d_region : d->d_flat;
    memset(&q, 0, sizeof q);
    if (l->type == DETECTION) {           /* YOLOv1 head: detection_layer.c:222 decode, then the same NMS / compaction */
        HIPCALL_I(y2h_detection_boxes(d_pred, (long)l->outputs, net.batch, l->side, l->n, l->classes, l->sqrt, img_w, img_h,
                                      thresh, 0, e->d_boxes, e->d_probs, ds));
    } else {
    q.batch = net.batch; q.w = l->w; q.h = l->h; q.num = l->n; q.classes = l->classes;
    q.img_w = img_w; q.img_h = img_h; q.thresh = thresh; q.classfix = l->classfix;
    q.anchors = d->d_anchors;
    q.tree_parent = l->softmax_tree ? d->d_tree_parent : NULL;
    q.tree_order = d->d_tree_order; q.tree_level_off = d->d_tree_loff; q.tree_levels = d->tree_levels;
    q.pred = d_pred; q.boxes = e->d_boxes; q.probs = e->d_probs;
    if (y2h_detect_chain_ok(&q)) {
        /* plain head: decode + NMS + compaction in three launches (the class counts stay zero between calls) */
        if (!e->class_counts_zeroed) {
            HIPCALL_I(y2h_memset(e->d_class_counts, 0, (size_t)net.batch * l->classes * sizeof(int), ds));
            e->class_counts_zeroed = 1;
        }
        HIPCALL_I(y2h_detect_chain(&q, nms, e->d_probs_nms, e->d_class_counts, e->d_records, e->d_counts, e->det_cap, e->d_best, ds));
        goto fetch;
    }
    if (y2h_detect_tree_chain_ok(&q)) {
        /* tree head without a map (yolo9000): one (class, score) pair per box instead of the dense score arrays, two launches */
        HIPCALL_I(y2h_detect_tree_chain(&q, nms, e->d_records, e->d_counts, e->det_cap, e->d_best,
                                        (d_pred == d->d_region && d->tree_best_valid) ? d->d_tree_best : NULL, ds));
        goto fetch;
    }
    HIPCALL_I(y2h_region_boxes(&q, ds));
    }
    final_probs = e->d_probs;
    e->class_counts_zeroed = 0;               /* y2h_nms_sort leaves its counts behind */
    if (nms > 0) {
        HIPCALL_I(y2h_memcpy_d2d(e->d_probs_nms, e->d_probs, (size_t)net.batch * e->det_total * l->classes * sizeof(float), ds));
        HIPCALL_I(y2h_nms_sort(e->d_boxes, e->d_probs, e->d_probs_nms, net.batch, e->det_total, l->classes, l->classes, nms, e->d_class_counts, ds));
        final_probs = e->d_probs_nms;
    }
    HIPCALL_I(y2h_collect(e->d_boxes, final_probs, net.batch, e->det_total, l->classes, l->classes, thresh,
                          e->d_records, e->d_counts, e->det_cap, e->d_best, ds));
fetch:
    HIPCALL_I(y2h_memcpy_d2h(e->h_counts, e->d_counts, (size_t)net.batch * sizeof(int), ds));
    keep = 0;
    if ((size_t)net.batch * e->det_cap * 6 * sizeof(float) <= ((size_t)8 << 20)) {
        /* small enough: fetch every record block with the counts, one copy and one sync per batch */
        HIPCALL_I(y2h_memcpy_d2h(e->h_records, e->d_records, (size_t)net.batch * e->det_cap * 6 * sizeof(float), ds));
        HIPCALL_I(y2h_event_record(e->ev_det, ds));
        e->det_pending = 1;                   /* detect_fetch waits for the event */
        return 0;
    } else {
        HIPCALL_I(y2h_stream_sync(ds));
        for (b = 0; b < net.batch; ++b) if (e->h_counts[b] > keep) keep = e->h_counts[b];
        if (keep > e->det_cap) keep = e->det_cap;
    }
    if (keep > 0) {
        /* one strided copy of the used prefix of every image's record block */
        for (b = 0; b < net.batch; ++b) {
            int nb = e->h_counts[b] < e->det_cap ? e->h_counts[b] : e->det_cap;
            if (nb > 0)
                HIPCALL_I(y2h_memcpy_d2h(e->h_records + (size_t)b * e->det_cap * 6, e->d_records + (size_t)b * e->det_cap * 6,
                                         (size_t)nb * 6 * sizeof(float), ds));
        }
        HIPCALL_I(y2h_stream_sync(ds));
    }
    e->det_pending = 2;                       /* wide heads (yolo9000): fetched synchronously above */
    return 0;
}

static int detect_fetch(network net, y2_det *dets, int *counts, int max_per_image)
{
    y2_engine *e = y2_engine_of(&net);
    int b, i;
    if (!e || !e->det_pending) { y2_fail("y2_detect_fetch: nothing was enqueued (call y2_detect_enqueue after a forward)"); return -1; }
    if (e->det_pending == 1) HIPCALL_I(y2h_event_sync(e->ev_det));
    e->det_pending = 0;
    for (b = 0; b < net.batch; ++b) {
        int nb = e->h_counts[b];
        counts[b] = nb;
        if (nb > e->det_cap) nb = e->det_cap;
        if (nb > max_per_image) nb = max_per_image;
        for (i = 0; i < nb; ++i) {
            const float *r = e->h_records + ((size_t)b * e->det_cap + i) * 6;
            y2_det *o = &dets[(size_t)b * max_per_image + i];
            o->x = r[0]; o->y = r[1]; o->w = r[2]; o->h = r[3]; o->prob = r[4]; o->obj_id = (int)r[5];
        }
    }
    return 0;
}

static int detect_from(network net, float *d_pred, float thresh, float nms, int img_w, int img_h,
                       y2_det *dets, int *counts, int max_per_image)
{
    if (detect_enqueue(net, d_pred, thresh, nms, img_w, img_h) != 0) return -1;
    return detect_fetch(net, dets, counts, max_per_image);
}

int y2_detect_resident(network net, float thresh, float nms, int img_w, int img_h,
                       y2_det *dets, int *counts, int max_per_image)
{
    return detect_from(net, NULL, thresh, nms, img_w, img_h, dets, counts, max_per_image);
}

/* The two halves of y2_detect_resident for callers that keep the GPU busy across batches: enqueue batch i's decode /
 * NMS / compaction, enqueue batch i+1's forward pass, then fetch batch i's detections -- the host-side wait and
 * unpacking overlap the next forward instead of idling the device. */
int y2_detect_enqueue(network net, float thresh, float nms, int img_w, int img_h)
{
    return detect_enqueue(net, NULL, thresh, nms, img_w, img_h);
}

int y2_detect_fetch(network net, y2_det *dets, int *counts, int max_per_image)
{
    return detect_fetch(net, dets, counts, max_per_image);
}

/* The 3-frame smoothing of Detector::detect(..., use_mean = true) (yolo_v2_class.cpp:208-213: memcpy into
 * predictions[demo_index], mean_arrays over the FRAMES = 3 slots, l.output = avg, demo_index++), kept in HBM: the
 * region output of the last forward goes into a three-slot ring that starts zeroed (the reference callocs its
 * slots), the slots are summed in slot order and divided by 3 exactly as utils.c:420-432 does, and decode + NMS run
 * on the average.  Batch-1 networks only, like the reference. */
int y2_detect_mean(network net, float thresh, float nms, int img_w, int img_h, y2_det *dets, int *counts, int max_per_image)
{
    y2_engine *e = y2_engine_of(&net);
    layer *l;
    y2_ldev *d;
    size_t els;
    if (!e || !e->built) { y2_fail("y2_detect_mean: run a forward first"); return -1; }
    if (e->det_pending == 1 && e->det_overlap) HIPCALL_I(y2h_event_sync(e->ev_det));
    l = &net.layers[e->out_layer];
    d = ld_of(l);
    if (l->type != REGION || net.batch != 1) { y2_fail("y2_detect_mean: needs a batch-1 network ending in a region layer"); return -1; }
    HIPCALL_I(y2h_set_device(e->device));
    els = (size_t)l->outputs;
    if (!e->d_mean_ring || e->mean_els != els) {
        y2h_free(e->d_mean_ring); e->d_mean_ring = NULL;
        HIPCALL_I(y2h_malloc((void **)&e->d_mean_ring, 4 * els * sizeof(float)));      /* 3 slots + the average */
        HIPCALL_I(y2h_memset(e->d_mean_ring, 0, 4 * els * sizeof(float), e->stream));
        e->mean_els = els; e->mean_index = 0;
    }
    HIPCALL_I(y2h_memcpy_d2d(e->d_mean_ring + (size_t)e->mean_index * els, d->d_region, els * sizeof(float), e->stream));
    e->mean_index = (e->mean_index + 1) % 3;
    HIPCALL_I(y2h_mean_frames(e->d_mean_ring, 3, (long)els, e->d_mean_ring + 3 * els, e->stream));
    return detect_from(net, e->d_mean_ring + 3 * els, thresh, nms, img_w, img_h, dets, counts, max_per_image);
}

int y2_detect(network net, float *input, float thresh, float nms, int img_w, int img_h,
              y2_det *dets, int *counts, int max_per_image)
{
    y2_engine *e;
    if (y2_prepare(&net) != 0) return -1;
    e = y2_engine_of(&net);
    HIPCALL_I(y2h_memcpy_h2d(e->d_in_nchw, input, e->in_floats * sizeof(float), e->stream));
    if (y2_forward_device(net, e->d_in_nchw) != 0) return -1;
    return y2_detect_resident(net, thresh, nms, img_w, img_h, dets, counts, max_per_image);
}

/* letterbox_image_into / letterbox_image (image.c:1607-1645) on the device: aspect-preserving resize,
 * embedded centred; letterbox_image fills the box with .5 first, _into keeps what `boxed` holds. */
static int letterbox_device(image im, int w, int h, image boxed, int fill)
{
    float *d_src, *d_tmp, *d_dst;
    y2h_stream s;
    int nw, nh, rc;
    size_t ns = (size_t)im.w * im.h * im.c, nd = (size_t)w * h * im.c, nt;
    if (!im.data || !boxed.data || im.w <= 0 || im.h <= 0 || w <= 0 || h <= 0) { y2_fail("letterbox_image: empty image"); return -1; }
    y2h_letterbox_dims(im.w, im.h, w, h, &nw, &nh);
    if (nw <= 0 || nh <= 0) { y2_fail("letterbox_image: degenerate size %d x %d", nw, nh); return -1; }
    nt = (size_t)im.c * im.h * nw + (size_t)im.c * nh * nw;
    if (img_scratch("letterbox_image", ns, nt, nd) != 0) return -1;
    d_src = g_img.d_src; d_tmp = g_img.d_tmp; d_dst = g_img.d_dst; s = g_img.stream;
    rc = y2h_memcpy_h2d(d_src, im.data, ns * 4, s);
    if (!rc) {
        if (fill) rc = y2h_letterbox_chw(d_src, im.c, im.h, im.w, d_tmp, d_dst, h, w, s);
        else {
            float *d_res = d_tmp + (size_t)im.c * im.h * nw;
            rc = y2h_memcpy_h2d(d_dst, boxed.data, nd * 4, s) || y2h_resize_chw(d_src, im.c, im.h, im.w, d_tmp, d_res, nh, nw, s) ||
                 y2h_embed_chw(d_res, im.c, nh, nw, d_dst, h, w, (w - nw) / 2, (h - nh) / 2, s);
        }
    }
    rc = rc || y2h_memcpy_d2h(boxed.data, d_dst, nd * 4, s) || y2h_stream_sync(s);
    if (rc) y2_fail("letterbox_image: %s", y2h_last_error());
    return rc ? -1 : 0;
}

image letterbox_image(image im, int w, int h)
{
    image boxed = make_image(w, h, im.c);
    letterbox_device(im, w, h, boxed, 1);
    return boxed;
}

void letterbox_image_into(image im, int w, int h, image boxed) { letterbox_device(im, w, h, boxed, 0); }

/* Frames as a camera hands them over: `batch` 8-bit interleaved images (h x w x c, BGR/BGRA/RGB, row
 * pitch `step` bytes) of any size.  One H2D of the bytes (a quarter of the float image), then on the
 * device: u8 -> [0,1] planes (+ BGR->RGB), resize_image or letterbox_image to the network's input,
 * forward, decode, NMS, compaction.  Equivalent to the reference's per-frame host sequence
 * ipl_to_image + rgbgr_image + resize_image + network_predict + get_region_boxes + do_nms_sort
 * (yolo_v2_class.cpp:173-249 with the hpp:59-76 glue). */
static int grow(void **p, size_t *cap, size_t need)
{
    if (need <= *cap) return 0;
    y2h_free(*p); *p = NULL; *cap = 0;
    if (y2h_malloc(p, need) != 0) return -1;
    *cap = need;
    return 0;
}

/* the device half of y2_ingest_u8: `d_frames` already sits in HBM */
int y2_ingest_u8_device(network net, const unsigned char *d_frames, int h, int w, int c, int step, int swap_rb, int letterbox)
{
    y2_engine *e;
    size_t frame_bytes, nplanes, ntmp = 0;
    int planes, nw = net.w, nh = net.h;
    if (!d_frames || h <= 0 || w <= 0 || c <= 0 || step < w * c) { y2_fail("y2_ingest_u8: bad frame geometry"); return -1; }
    if (c < net.c) { y2_fail("y2_ingest_u8: frames have %d channels, the network reads %d", c, net.c); return -1; }
    if (y2_prepare(&net) != 0) return -1;
    e = y2_engine_of(&net);
    HIPCALL_I(y2h_set_device(e->device));
    planes = net.c;                       /* a 4th (alpha) plane is never read by the network (detector.c:567) */
    frame_bytes = (size_t)step * h;
    nplanes = (size_t)net.batch * planes * h * w;
    if (w == net.w && h == net.h) {
        HIPCALL_I(y2h_u8_to_planes(d_frames, net.batch, h, w, c, step, (long)frame_bytes, planes, swap_rb, e->d_in_nchw, e->stream));
        return 0;
    }
    if (letterbox) y2h_letterbox_dims(w, h, net.w, net.h, &nw, &nh);
    if (nw <= 0 || nh <= 0) { y2_fail("y2_ingest_u8: degenerate letterbox %d x %d", nw, nh); return -1; }
    ntmp = (size_t)net.batch * planes * h * nw + (letterbox ? (size_t)net.batch * planes * nh * nw : 0);
    if (grow((void **)&e->d_planes, &e->planes_cap, nplanes * 4) || grow((void **)&e->d_rtmp, &e->rtmp_cap, ntmp * 4)) {
        y2_fail("y2_ingest_u8: %s", y2h_last_error()); return -1;
    }
    HIPCALL_I(y2h_u8_to_planes(d_frames, net.batch, h, w, c, step, (long)frame_bytes, planes, swap_rb, e->d_planes, e->stream));
    /* planes are independent in resize/embed, so the whole batch goes through as batch*planes planes */
    if (letterbox) HIPCALL_I(y2h_letterbox_chw(e->d_planes, net.batch * planes, h, w, e->d_rtmp, e->d_in_nchw, net.h, net.w, e->stream));
    else HIPCALL_I(y2h_resize_chw(e->d_planes, net.batch * planes, h, w, e->d_rtmp, e->d_in_nchw, net.h, net.w, e->stream));
    return 0;
}

int y2_ingest_u8(network net, const unsigned char *frames, int h, int w, int c, int step, int swap_rb, int letterbox)
{
    y2_engine *e;
    size_t frame_bytes;
    if (!frames || h <= 0 || w <= 0 || c <= 0 || step < w * c) { y2_fail("y2_ingest_u8: bad frame geometry"); return -1; }
    if (y2_prepare(&net) != 0) return -1;
    e = y2_engine_of(&net);
    HIPCALL_I(y2h_set_device(e->device));
    frame_bytes = (size_t)step * h;
    if (grow((void **)&e->d_u8, &e->u8_cap, frame_bytes * net.batch)) { y2_fail("y2_ingest_u8: %s", y2h_last_error()); return -1; }
    HIPCALL_I(y2h_memcpy_h2d(e->d_u8, frames, frame_bytes * net.batch, e->stream));
    return y2_ingest_u8_device(net, e->d_u8, h, w, c, step, swap_rb, letterbox);
}

/* A float CHW frame of any size (the `image` the reference's callers hold): its first net.c planes are uploaded
 * and resized (image.c:1950) straight into the network's device input -- what resize_image + the input copy of
 * network_predict do in the reference (detector.c:567-573), without the host round trip or any allocation. */
int y2_ingest_image(network net, image im)
{
    y2_engine *e;
    size_t plane = (size_t)im.w * im.h, ntmp;
    if (!im.data || im.w <= 0 || im.h <= 0) { y2_fail("y2_ingest_image: empty image"); return -1; }
    if (im.c < net.c) { y2_fail("y2_ingest_image: the image has %d planes, the network reads %d", im.c, net.c); return -1; }
    if (net.batch != 1) { y2_fail("y2_ingest_image: set_batch_network(&net, 1) first"); return -1; }
    if (y2_prepare(&net) != 0) return -1;
    e = y2_engine_of(&net);
    HIPCALL_I(y2h_set_device(e->device));
    if (im.w == net.w && im.h == net.h) {
        HIPCALL_I(y2h_memcpy_h2d(e->d_in_nchw, im.data, plane * net.c * sizeof(float), e->stream));
        return 0;
    }
    ntmp = (size_t)net.c * im.h * net.w;
    if (grow((void **)&e->d_planes, &e->planes_cap, plane * net.c * 4) || grow((void **)&e->d_rtmp, &e->rtmp_cap, ntmp * 4)) {
        y2_fail("y2_ingest_image: %s", y2h_last_error()); return -1;
    }
    HIPCALL_I(y2h_memcpy_h2d(e->d_planes, im.data, plane * net.c * sizeof(float), e->stream));
    HIPCALL_I(y2h_resize_chw(e->d_planes, net.c, im.h, im.w, e->d_rtmp, e->d_in_nchw, net.h, net.w, e->stream));
    return 0;
}

int y2_detect_u8(network net, const unsigned char *frames, int h, int w, int c, int step, int swap_rb, int letterbox,
                 float thresh, float nms, int img_w, int img_h, y2_det *dets, int *counts, int max_per_image)
{
    y2_engine *e;
    if (y2_ingest_u8(net, frames, h, w, c, step, swap_rb, letterbox) != 0) return -1;
    e = y2_engine_of(&net);
    if (y2_forward_device(net, e->d_in_nchw) != 0) return -1;
    return y2_detect_resident(net, thresh, nms, img_w, img_h, dets, counts, max_per_image);
}

/* detector.c:558-598 test_detector_img: resize -> predict -> get_region_boxes(1,1,thresh) ->
 * do_nms_sort(nms=0.1) -> draw_detections_test, which fills RecObects (image.c:662-738).
 * Reads exactly net.inputs floats of the resized image (planes 0..2; a 4th plane is ignored). */
void test_detector_img(char **names, image **alphabet, network net, image im, float thresh,
                       object *RecObects, int *objectNumPerFrame)
{
    const float nms = 0.1f;
    layer l = net.layers[net.n - 1];
    int total = l.w * l.h * l.n, n, i;
    y2_det *dets;
    int count = 0;
    (void)alphabet;
    if (net.batch != 1) { y2_fail("test_detector_img: set_batch_network(&net, 1) first"); return; }
    /* resize_image + network_predict's input copy, fused on the device (planes are resized independently, so
     * resizing only the planes the network reads gives the same input as the reference's 4-plane resize) */
    if (y2_ingest_image(net, im) != 0) return;
    dets = calloc(total > 0 ? total : 1, sizeof(y2_det));
    if (y2_forward_device(net, NULL) != 0 || y2_detect_resident(net, thresh, nms, 1, 1, dets, &count, total) != 0) { free(dets); return; }
    n = count < total ? count : total;
    for (i = 0; i < n; ++i) {
        object *o = &RecObects[*objectNumPerFrame];
        int cls = dets[i].obj_id;
        int offset = cls * 123457 % l.classes;
        if (names) printf("%s: %.0f%%\n", names[cls], dets[i].prob * 100);
        o->x = dets[i].x; o->y = dets[i].y; o->w = dets[i].w; o->h = dets[i].h;
        o->prob = dets[i].prob;
        o->objClass = cls;
        if (names && names[cls]) { strncpy(o->name, names[cls], sizeof o->name - 1); o->name[sizeof o->name - 1] = 0; }
        else o->name[0] = 0;
        o->boxRGB[0] = get_color(2, offset, l.classes);
        o->boxRGB[1] = get_color(1, offset, l.classes);
        o->boxRGB[2] = get_color(0, offset, l.classes);
        (*objectNumPerFrame)++;
    }
    free(dets);
}
