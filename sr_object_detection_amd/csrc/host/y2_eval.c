/* y2_eval.c -- evaluation writers of the detector.
 *
 * Restates, on top of this library's own network_predict / get_region_boxes / do_nms_sort (which run
 * on the GPU), the reference's
 *   print_cocos                 detector.c:175-199   (static there; exported here)
 *   print_detector_detections   detector.c:201-220
 *   print_imagenet_detections   detector.c:222-243
 *   validate_detector           detector.c:245-368   -> y2_validate_detector_frames
 *   validate_detector_recall    detector.c:371-450   -> y2_validate_recall_frames
 * The two validate_* loops of the reference read an image list from disk with stb_image in loader
 * threads; file decoding is outside this engine, so the loops here take frames that are already in
 * memory (network-sized CHW floats) and otherwise do exactly what the reference does per image:
 * same thresholds, same call sequence, same text written to the same file names.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "y2_internal.h"

/* detector.c:23 -- the 80 COCO category ids in class order */
static const int coco_ids[] = {1,2,3,4,5,6,7,8,9,10,11,13,14,15,16,17,18,19,20,21,22,23,24,25,27,28,31,32,33,34,35,36,37,
                               38,39,40,41,42,43,44,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63,64,65,67,70,72,
                               73,74,75,76,77,78,79,80,81,82,84,85,86,87,88,89,90};

/* the box of detection i in corner form, clipped to the image (detector.c:180-188) */
static void corners(box b, int w, int h, float *xmin, float *ymin, float *xmax, float *ymax)
{
    *xmin = b.x - b.w / 2.;
    *xmax = b.x + b.w / 2.;
    *ymin = b.y - b.h / 2.;
    *ymax = b.y + b.h / 2.;
    if (*xmin < 0) *xmin = 0;
    if (*ymin < 0) *ymin = 0;
    if (*xmax > w) *xmax = w;
    if (*ymax > h) *ymax = h;
}

int get_coco_image_id(char *filename)            /* detector.c:169 */
{
    char *p = filename ? strrchr(filename, '_') : NULL;
    return p ? atoi(p + 1) : 0;                  /* the reference dereferences NULL when there is no '_' */
}

void print_cocos(FILE *fp, char *image_path, box *boxes, float **probs, int num_boxes, int classes, int w, int h)
{
    int i, j;
    const int image_id = get_coco_image_id(image_path);
    for (i = 0; i < num_boxes; ++i) {
        float xmin, ymin, xmax, ymax;
        corners(boxes[i], w, h, &xmin, &ymin, &xmax, &ymax);
        for (j = 0; j < classes && j < (int)(sizeof coco_ids / sizeof coco_ids[0]); ++j)
            if (probs[i][j])
                fprintf(fp, "{\"image_id\":%d, \"category_id\":%d, \"bbox\":[%f, %f, %f, %f], \"score\":%f},\n", image_id,
                        coco_ids[j], xmin, ymin, xmax - xmin, ymax - ymin, probs[i][j]);
    }
}

void print_detector_detections(FILE **fps, char *id, box *boxes, float **probs, int total, int classes, int w, int h)
{
    int i, j;
    for (i = 0; i < total; ++i) {
        float xmin, ymin, xmax, ymax;
        corners(boxes[i], w, h, &xmin, &ymin, &xmax, &ymax);
        for (j = 0; j < classes; ++j)
            if (probs[i][j]) fprintf(fps[j], "%s %f %f %f %f %f\n", id, probs[i][j], xmin, ymin, xmax, ymax);
    }
}

void print_imagenet_detections(FILE *fp, int id, box *boxes, float **probs, int total, int classes, int w, int h)
{
    int i, j;
    for (i = 0; i < total; ++i) {
        float xmin, ymin, xmax, ymax;
        corners(boxes[i], w, h, &xmin, &ymin, &xmax, &ymax);
        for (j = 0; j < classes; ++j)
            if (probs[i][j]) fprintf(fp, "%d %d %f %f %f %f %f\n", id, j + 1, probs[i][j], xmin, ymin, xmax, ymax);
    }
}

char *basecfg(char *cfgfile)                     /* utils.c:121: file name without directory and extension */
{
    char *c = cfgfile, *next, *out;
    while ((next = strchr(c, '/'))) c = next + 1;
    out = malloc(strlen(c) + 1);
    strcpy(out, c);
    next = strchr(out, '.');
    if (next) *next = 0;
    return out;
}

typedef struct { box *boxes; float **probs; float *store; int total, classes; } det_arrays;

static int det_arrays_make(det_arrays *d, int total, int classes)
{
    int j;
    d->total = total; d->classes = classes;
    d->boxes = calloc(total, sizeof(box));
    d->probs = calloc(total, sizeof(float *));
    d->store = calloc((size_t)total * classes, sizeof(float));
    if (!d->boxes || !d->probs || !d->store) return -1;
    for (j = 0; j < total; ++j) d->probs[j] = d->store + (size_t)j * classes;
    return 0;
}

static void det_arrays_free(det_arrays *d) { free(d->boxes); free(d->probs); free(d->store); }

/* One forward of up to net.batch frames (the tail of the last batch is zero padded). */
static float *predict_chunk(network net, float *frames, int first, int n, float *staging)
{
    const size_t per = (size_t)net.inputs;
    int cnt = n - first < net.batch ? n - first : net.batch;
    if (cnt == net.batch) return network_predict(net, frames + per * first);
    memset(staging, 0, per * net.batch * sizeof(float));
    memcpy(staging, frames + per * first, per * cnt * sizeof(float));
    return network_predict(net, staging);
}

/* yolo.c:95-114: the YOLOv1 writer -- corner form clipped to [0,w] x [0,h] (no +1 offset), one file per class */
void print_yolo_detections(FILE **fps, char *id, box *boxes, float **probs, int total, int classes, int w, int h)
{
    int i, j;
    for (i = 0; i < total; ++i) {
        float xmin = boxes[i].x - boxes[i].w / 2.;
        float xmax = boxes[i].x + boxes[i].w / 2.;
        float ymin = boxes[i].y - boxes[i].h / 2.;
        float ymax = boxes[i].y + boxes[i].h / 2.;
        if (xmin < 0) xmin = 0;
        if (ymin < 0) ymin = 0;
        if (xmax > w) xmax = w;
        if (ymax > h) ymax = h;
        for (j = 0; j < classes; ++j)
            if (probs[i][j]) fprintf(fps[j], "%s %f %f %f %f %f\n", id, probs[i][j], xmin, ymin, xmax, ymax);
    }
}

int y2_validate_detector_frames(network net, float *frames, int n, char **paths, int *orig_w, int *orig_h,
                                char *eval, char *prefix, char **names, int *map)
{
    layer l;
    int classes, total, j, i, b, coco = 0, imagenet = 0, rc = -1;
    char buff[1024];
    FILE *fp = NULL, **fps = NULL;
    det_arrays d = {0};
    float *staging = NULL;
    /* a network ending in [detection] (YOLOv1) follows validate_yolo instead (yolo.c:116-200): thresh .001, NMS .5,
     * get_detection_boxes, print_yolo_detections */
    const int v1 = net.n > 0 && net.layers[net.n - 1].type == DETECTION;
    const float thresh = v1 ? .001f : .005f, nms = v1 ? .5f : .45f;       /* detector.c:305-306 / yolo.c:150-152 */
    const char *base = "comp4_det_test_";
    if (!frames || n <= 0 || !paths || !orig_w || !orig_h || !prefix) { y2_fail("y2_validate_detector_frames: missing argument"); return -1; }
    if (net.n <= 0 || (net.layers[net.n - 1].type != REGION && !v1)) { y2_fail("y2_validate_detector_frames: the network does not end in a region or detection layer"); return -1; }
    if (v1 && eval && (0 == strcmp(eval, "coco") || 0 == strcmp(eval, "imagenet"))) { y2_fail("y2_validate_detector_frames: a [detection] head writes the voc format only"); return -1; }
    l = net.layers[net.n - 1];
    classes = l.classes;
    total = v1 ? l.side * l.side * l.n : l.w * l.h * l.n;
    if (eval && 0 == strcmp(eval, "coco")) {
        snprintf(buff, sizeof buff, "%s/coco_results.json", prefix);
        fp = fopen(buff, "w");
        if (!fp) { y2_fail("cannot write %s", buff); return -1; }
        fprintf(fp, "[\n");
        coco = 1;
    } else if (eval && 0 == strcmp(eval, "imagenet")) {
        snprintf(buff, sizeof buff, "%s/imagenet-detection.txt", prefix);
        fp = fopen(buff, "w");
        if (!fp) { y2_fail("cannot write %s", buff); return -1; }
        imagenet = 1;
        classes = 200;                            /* detector.c:286 */
    } else {
        if (!names) { y2_fail("y2_validate_detector_frames: the voc writer needs class names"); return -1; }
        fps = calloc(classes, sizeof(FILE *));
        for (j = 0; j < classes; ++j) {
            snprintf(buff, sizeof buff, "%s/%s%s.txt", prefix, base, names[j]);
            fps[j] = fopen(buff, "w");
            if (!fps[j]) { y2_fail("cannot write %s", buff); goto done; }
        }
    }
    if (det_arrays_make(&d, total, l.classes > classes ? l.classes : classes)) { y2_fail("out of memory"); goto done; }
    d.classes = classes;
    staging = calloc((size_t)net.inputs * net.batch, sizeof(float));
    for (i = 0; i < n; i += net.batch) {
        float *out = predict_chunk(net, frames, i, n, staging);
        if (!out) goto done;
        for (b = 0; b < net.batch && i + b < n; ++b) {
            layer lb = l;
            const int w = orig_w[i + b], h = orig_h[i + b];
            lb.output = out + (size_t)b * l.outputs;                 /* this frame's slice of the region output */
            if (v1) get_detection_boxes(lb, w, h, thresh, d.probs, d.boxes, 0);
            else get_region_boxes(lb, w, h, thresh, d.probs, d.boxes, 0, map);
            if (y2_failed()) goto done;
            if (nms) do_nms_sort(d.boxes, d.probs, total, classes, nms);
            if (y2_failed()) goto done;
            if (v1) { char *id = basecfg(paths[i + b]); print_yolo_detections(fps, id, d.boxes, d.probs, total, classes, w, h); free(id); }
            else if (coco) print_cocos(fp, paths[i + b], d.boxes, d.probs, total, classes, w, h);
            else if (imagenet) print_imagenet_detections(fp, i + b + 1, d.boxes, d.probs, total, classes, w, h);
            else { char *id = basecfg(paths[i + b]); print_detector_detections(fps, id, d.boxes, d.probs, total, classes, w, h); free(id); }
        }
    }
    rc = 0;
done:
    if (fps) { for (j = 0; j < l.classes; ++j) if (fps[j]) fclose(fps[j]); free(fps); }
    if (fp) {
        if (coco) { fseek(fp, -2, SEEK_CUR); fprintf(fp, "\n]\n"); }   /* detector.c:362-364: drop the trailing ",\n" */
        fclose(fp);
    }
    det_arrays_free(&d);
    free(staging);
    return rc;
}

int y2_validate_recall_frames(network net, float *frames, int n, const box *truth, const int *truth_first, y2_recall *res)
{
    layer l;
    int total_boxes, i, b, j, k, rc = -1;
    det_arrays d = {0};
    float *staging = NULL;
    const float thresh = .2f, iou_thresh = .5f, nms = .4f;    /* detector.c:399-401 */
    int total = 0, correct = 0, proposals = 0;
    float avg_iou = 0;
    if (!frames || n <= 0 || !truth_first || !res) { y2_fail("y2_validate_recall_frames: missing argument"); return -1; }
    if (net.n <= 0 || net.layers[net.n - 1].type != REGION) { y2_fail("y2_validate_recall_frames: the network does not end in a region layer"); return -1; }
    l = net.layers[net.n - 1];
    total_boxes = l.w * l.h * l.n;
    if (det_arrays_make(&d, total_boxes, l.classes)) { y2_fail("out of memory"); goto done; }
    staging = calloc((size_t)net.inputs * net.batch, sizeof(float));
    for (i = 0; i < n; i += net.batch) {
        float *out = predict_chunk(net, frames, i, n, staging);
        if (!out) goto done;
        for (b = 0; b < net.batch && i + b < n; ++b) {
            layer lb = l;
            const int f = i + b;
            lb.output = out + (size_t)b * l.outputs;
            get_region_boxes(lb, 1, 1, thresh, d.probs, d.boxes, 1, 0);      /* only_objectness = 1 */
            if (y2_failed()) goto done;
            if (nms) do_nms(d.boxes, d.probs, total_boxes, 1, nms);
            if (y2_failed()) goto done;
            for (k = 0; k < total_boxes; ++k) if (d.probs[k][0] > thresh) ++proposals;
            for (j = truth_first[f]; j < truth_first[f + 1]; ++j) {
                float best_iou = 0;
                ++total;
                for (k = 0; k < total_boxes; ++k) {
                    float iou = box_iou(d.boxes[k], truth[j]);
                    if (d.probs[k][0] > thresh && iou > best_iou) best_iou = iou;
                }
                avg_iou += best_iou;
                if (best_iou > iou_thresh) ++correct;
            }
            fprintf(stderr, "%5d %5d %5d\tRPs/Img: %.2f\tIOU: %.2f%%\tRecall:%.2f%%\n", f, correct, total,
                    (float)proposals / (f + 1), avg_iou * 100 / total, 100. * correct / total);
        }
    }
    res->total = total; res->correct = correct; res->proposals = proposals; res->avg_iou = avg_iou;
    rc = 0;
done:
    det_arrays_free(&d);
    free(staging);
    return rc;
}

/* validate_classifier_single (classifier.c:469-529) with the image list replaced by `n` network-sized CHW frames in
 * memory and the label-from-path lookup (:502-507) by `truth[f]` (-1 = no label, as when no label string matches):
 * per frame network_predict -> top_k(pred, classes, topk, indexes) -> running top-1 / top-k accuracy, the reference's
 * progress line per frame on stdout.  Frames are processed net.batch at a time.  A hierarchical classifier
 * (net.hierarchy, softmax tree=) is refused: that head is not implemented on the device. */
int y2_validate_classifier_frames(network net, float *frames, int n, const int *truth, int classes, int topk,
                                  float *top1_out, float *topk_out)
{
    float *staging = NULL, avg_acc = 0, avg_topk = 0;
    int *indexes = NULL, i, b, j, rc = -1, outputs;
    if (!frames || n <= 0 || !truth || classes <= 0 || topk <= 0) { y2_fail("y2_validate_classifier_frames: missing argument"); return -1; }
    if (net.hierarchy) { y2_fail("y2_validate_classifier_frames: hierarchical classifiers (softmax tree=) are not implemented on the device"); return -1; }
    outputs = get_network_output_size(net);
    if (classes > outputs || topk > classes) { y2_fail("y2_validate_classifier_frames: classes %d / top %d against %d network outputs", classes, topk, outputs); return -1; }
    indexes = calloc(topk, sizeof(int));
    staging = calloc((size_t)net.inputs * net.batch, sizeof(float));
    if (!indexes || !staging) { y2_fail("out of memory"); goto done; }
    for (i = 0; i < n; i += net.batch) {
        float *out = predict_chunk(net, frames, i, n, staging);
        if (!out) goto done;
        for (b = 0; b < net.batch && i + b < n; ++b) {
            const int f = i + b;
            top_k(out + (size_t)b * outputs, classes, topk, indexes);
            if (indexes[0] == truth[f]) avg_acc += 1;
            for (j = 0; j < topk; ++j) if (indexes[j] == truth[f]) avg_topk += 1;
            printf("%d: top 1: %f, top %d: %f\n", f, avg_acc / (f + 1), topk, avg_topk / (f + 1));
        }
    }
    if (top1_out) *top1_out = avg_acc / n;
    if (topk_out) *topk_out = avg_topk / n;
    rc = 0;
done:
    free(indexes);
    free(staging);
    return rc;
}
