/*
 * ref_driver.c -- TEST INFRASTRUCTURE ONLY (never linked by the product).
 *
 * A small command-line driver, written for this repo, that calls the
 * *reference's own* compiled CPU path (oracle/_ref/libdarknet_ref.so, built
 * by oracle/build_ref.sh from /root/reference/src_yolo2/*.c) through the
 * reference's public C API:
 *
 *   parse_network_cfg      src_yolo2/parser.h:5
 *   load_weights           src_yolo2/parser.h:10
 *   network_predict        src_yolo2/network.h:109
 *   get_region_boxes       src_yolo2/region_layer.h:12
 *   do_nms_sort / box_iou  src_yolo2/box.h:13-17
 *   reorg_cpu/flatten/softmax   src_yolo2/blas.h
 *
 * It is used (a) to generate the golden vectors under tests/golden/ and
 * (b) as the "reference" flavour of bench.py's cpu_baseline.
 *
 * Raw little-endian float32 / int32 files are used for all tensors.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>

#include "network.h"
#include "parser.h"
#include "region_layer.h"
#include "detection_layer.h"
#include "box.h"
#include "blas.h"
#include "utils.h"
#include "tree.h"

int gpu_index = -1; /* cuda.h:8 -- defined by darknet.c in the reference's CLI; CPU path */

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static float *read_floats(const char *path, size_t *n_out)
{
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "ref_driver: cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    float *x = malloc(bytes > 0 ? bytes : 4);
    if (fread(x, 1, bytes, f) != (size_t)bytes) { fprintf(stderr, "short read %s\n", path); exit(2); }
    fclose(f);
    if (n_out) *n_out = bytes / sizeof(float);
    return x;
}

static void write_raw(const char *dir, const char *name, const void *p, size_t bytes)
{
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE *f = fopen(path, "wb");
    if (!f) { fprintf(stderr, "ref_driver: cannot write %s\n", path); exit(2); }
    fwrite(p, 1, bytes, f);
    fclose(f);
}

static layer last_real_layer(network net, int *idx)
{
    int i;
    for (i = net.n - 1; i > 0; --i) if (net.layers[i].type != COST) break;
    if (idx) *idx = i;
    return net.layers[i];
}

/* net <cfg> <weights|-> <input.bin> <outdir> <thresh> <nms> <dump_layers 0/1> */
static int cmd_net(int argc, char **argv)
{
    if (argc < 9) { fprintf(stderr, "usage: net cfg weights input outdir thresh nms dump\n"); return 2; }
    char *cfg = argv[2], *weights = argv[3], *input = argv[4], *outdir = argv[5];
    float thresh = atof(argv[6]);
    float nms = atof(argv[7]);
    int dump = atoi(argv[8]);

    network net = parse_network_cfg(cfg);
    if (strcmp(weights, "-") != 0) load_weights(&net, weights);
    size_t nin = 0;
    float *X = read_floats(input, &nin);
    if (nin != (size_t)net.inputs * net.batch) {
        fprintf(stderr, "ref_driver: input has %zu floats, net wants %d x %d\n", nin, net.batch, net.inputs);
        return 2;
    }
    double t0 = now_s();
    float *out = network_predict(net, X);
    double t1 = now_s();

    int li;
    layer l = last_real_layer(net, &li);
    write_raw(outdir, "out.bin", out, (size_t)l.outputs * net.batch * sizeof(float));

    char path[1024];
    snprintf(path, sizeof path, "%s/layers.txt", outdir);
    FILE *lf = fopen(path, "w");
    int i;
    for (i = 0; i < net.n; ++i) {
        layer q = net.layers[i];
        double s = 0, s2 = 0; float mn = INFINITY, mx = -INFINITY;
        size_t n = (size_t)q.outputs * q.batch, j;
        if (q.output) for (j = 0; j < n; ++j) { float v = q.output[j]; s += v; s2 += (double)v * v; if (v < mn) mn = v; if (v > mx) mx = v; }
        fprintf(lf, "%d %d %d %d %d %d %d %d %d %d %d %d %.9g %.9g %.9g %.9g\n", i, (int)q.type, q.w, q.h, q.c,
                q.out_w, q.out_h, q.out_c, q.outputs, q.n, q.size, q.stride, s, s2, (double)mn, (double)mx);
        if (dump && q.output) {
            char nm[64];
            snprintf(nm, sizeof nm, "layer_%02d.bin", i);
            write_raw(outdir, nm, q.output, n * sizeof(float));
        }
    }
    fclose(lf);

    if (l.type == REGION) {
        int total = l.w * l.h * l.n, b, j;
        int ncls = l.classes;
        int out_cls = (l.softmax_tree && l.map) ? 200 : ncls; /* region_layer.c:351 */
        box *boxes = calloc(total, sizeof(box));
        float **probs = calloc(total, sizeof(float *));
        float *flat = calloc((size_t)total * ncls, sizeof(float));
        for (j = 0; j < total; ++j) probs[j] = flat + (size_t)j * ncls;
        for (b = 0; b < net.batch; ++b) {
            layer lb = l;
            lb.output = l.output + (size_t)b * l.outputs; /* batched decode: SURVEY 8b extension */
            memset(flat, 0, (size_t)total * ncls * sizeof(float));
            get_region_boxes(lb, 1, 1, thresh, probs, boxes, 0, l.map);
            char nm[64];
            snprintf(nm, sizeof nm, "boxes_%d.bin", b);
            write_raw(outdir, nm, boxes, total * sizeof(box));
            snprintf(nm, sizeof nm, "probs_pre_%d.bin", b);
            write_raw(outdir, nm, flat, (size_t)total * ncls * sizeof(float));
            if (nms > 0) do_nms_sort(boxes, probs, total, out_cls, nms);
            snprintf(nm, sizeof nm, "probs_post_%d.bin", b);
            write_raw(outdir, nm, flat, (size_t)total * ncls * sizeof(float));
        }
        free(boxes); free(probs); free(flat);
    }
    if (l.type == DETECTION) {      /* YOLOv1 head: yolo.c:324-325 get_detection_boxes(l,1,1,thresh,..,0) + do_nms_sort */
        int total = l.side * l.side * l.n, b, j;
        int ncls = l.classes;
        box *boxes = calloc(total, sizeof(box));
        float **probs = calloc(total, sizeof(float *));
        float *flat = calloc((size_t)total * ncls, sizeof(float));
        for (j = 0; j < total; ++j) probs[j] = flat + (size_t)j * ncls;
        for (b = 0; b < net.batch; ++b) {
            layer lb = l;
            char nm[64];
            lb.output = l.output + (size_t)b * l.outputs;
            memset(flat, 0, (size_t)total * ncls * sizeof(float));
            get_detection_boxes(lb, 1, 1, thresh, probs, boxes, 0);
            snprintf(nm, sizeof nm, "boxes_%d.bin", b);
            write_raw(outdir, nm, boxes, total * sizeof(box));
            snprintf(nm, sizeof nm, "probs_pre_%d.bin", b);
            write_raw(outdir, nm, flat, (size_t)total * ncls * sizeof(float));
            if (nms > 0) do_nms_sort(boxes, probs, total, ncls, nms);
            snprintf(nm, sizeof nm, "probs_post_%d.bin", b);
            write_raw(outdir, nm, flat, (size_t)total * ncls * sizeof(float));
        }
        free(boxes); free(probs); free(flat);
    }
    snprintf(path, sizeof path, "%s/meta.txt", outdir);
    FILE *mf = fopen(path, "w");
    fprintf(mf, "n %d\nbatch %d\nw %d\nh %d\nc %d\ninputs %d\noutputs %d\nlast %d\nlast_type %d\nlw %d\nlh %d\nln %d\nclasses %d\npredict_s %.6f\n",
            net.n, net.batch, net.w, net.h, net.c, net.inputs, l.outputs, li, (int)l.type, l.w, l.h, l.n, l.classes, t1 - t0);
    fclose(mf);
    return 0;
}

/* time <cfg> <weights|-> <iters>: wall-clock of network_predict on a ramp image */
static int cmd_time(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: time cfg weights iters\n"); return 2; }
    network net = parse_network_cfg(argv[2]);
    if (strcmp(argv[3], "-") != 0) load_weights(&net, argv[3]);
    int iters = atoi(argv[4]), i;
    size_t n = (size_t)net.inputs * net.batch, j;
    float *X = malloc(n * sizeof(float));
    for (j = 0; j < n; ++j) X[j] = (float)((j * 2654435761u) % 1000) / 1000.f;
    network_predict(net, X); /* warm-up */
    double best = 1e30, sum = 0, med;
    double *ts = calloc(iters > 0 ? iters : 1, sizeof(double));
    int k;
    for (i = 0; i < iters; ++i) {
        double t0 = now_s();
        network_predict(net, X);
        double dt = now_s() - t0;
        if (dt < best) best = dt;
        sum += dt;
        for (k = i; k > 0 && ts[k - 1] > dt; --k) ts[k] = ts[k - 1];     /* keep sorted */
        ts[k] = dt;
    }
    med = iters > 0 ? ((iters & 1) ? ts[iters / 2] : 0.5 * (ts[iters / 2 - 1] + ts[iters / 2])) : 0;
    printf("{\"batch\": %d, \"iters\": %d, \"mean_s\": %.6f, \"best_s\": %.6f, \"median_s\": %.6f}\n", net.batch, iters,
           sum / iters, best, med);
    free(ts);
    return 0;
}

/* prim <name> <in.bin> <out.bin> ints... : single reference primitives */
static int cmd_prim(int argc, char **argv)
{
    if (argc < 5) return 2;
    char *name = argv[2];
    size_t n = 0;
    float *x = read_floats(argv[3], &n);
    float *y = calloc(n + 16, sizeof(float));
    size_t nout = n;
    int a[8] = {0}, i;
    for (i = 0; i < 8 && 5 + i < argc; ++i) a[i] = atoi(argv[5 + i]);
    if (!strcmp(name, "reorg")) {            /* w h c batch stride forward  (blas.c:8) */
        reorg_cpu(x, a[0], a[1], a[2], a[3], a[4], a[5], y);
    } else if (!strcmp(name, "flatten")) {   /* size layers batch forward   (blas.c:31) */
        flatten(x, a[0], a[1], a[2], a[3]);
        memcpy(y, x, n * sizeof(float));
    } else if (!strcmp(name, "softmax")) {   /* n, temp = argv[6] as float  (blas.c:205) */
        softmax(x, a[0], atof(argv[6]), y);
        nout = a[0];
    } else if (!strcmp(name, "iou")) {       /* pairs of boxes -> iou       (box.c:94) */
        size_t p, np = n / 8;
        for (p = 0; p < np; ++p) {
            box ba = {x[8*p], x[8*p+1], x[8*p+2], x[8*p+3]};
            box bb = {x[8*p+4], x[8*p+5], x[8*p+6], x[8*p+7]};
            y[p] = box_iou(ba, bb);
        }
        nout = np;
    } else if (!strcmp(name, "nms")) {       /* total classes, thresh=argv[7]; input = boxes[total*4] ++ probs[total*classes] (box.c:249) */
        int total = a[0], classes = a[1];
        float th = atof(argv[7]);
        box *boxes = (box *)x;
        float *pf = x + (size_t)total * 4;
        float **probs = calloc(total, sizeof(float *));
        for (i = 0; i < total; ++i) probs[i] = pf + (size_t)i * classes;
        do_nms_sort(boxes, probs, total, classes, th);
        memcpy(y, pf, (size_t)total * classes * sizeof(float));
        nout = (size_t)total * classes;
    } else {
        fprintf(stderr, "unknown prim %s\n", name);
        return 2;
    }
    FILE *f = fopen(argv[4], "wb");
    fwrite(y, sizeof(float), nout, f);
    fclose(f);
    return 0;
}

/* evalw <in.bin> <total> <classes> <w> <h> <id> <out_prefix> : the reference's evaluation writers
 * (detector.c:201 print_detector_detections -> <out_prefix>_c<j>.txt per class, detector.c:222
 * print_imagenet_detections with id 7 -> <out_prefix>_imagenet.txt); input = boxes[total*4] ++ probs[total*classes] */
void print_detector_detections(FILE **fps, char *id, box *boxes, float **probs, int total, int classes, int w, int h);
void print_imagenet_detections(FILE *fp, int id, box *boxes, float **probs, int total, int classes, int w, int h);
static int cmd_evalw(int argc, char **argv)
{
    if (argc < 9) return 2;
    size_t n = 0;
    float *x = read_floats(argv[2], &n);
    int total = atoi(argv[3]), classes = atoi(argv[4]), w = atoi(argv[5]), h = atoi(argv[6]), i;
    char path[1024];
    box *boxes = (box *)x;
    float **probs = calloc(total, sizeof(float *));
    FILE **fps = calloc(classes, sizeof(FILE *));
    for (i = 0; i < total; ++i) probs[i] = x + (size_t)total * 4 + (size_t)i * classes;
    for (i = 0; i < classes; ++i) { snprintf(path, sizeof path, "%s_c%d.txt", argv[8], i); fps[i] = fopen(path, "w"); }
    print_detector_detections(fps, argv[7], boxes, probs, total, classes, w, h);
    for (i = 0; i < classes; ++i) fclose(fps[i]);
    snprintf(path, sizeof path, "%s_imagenet.txt", argv[8]);
    FILE *fp = fopen(path, "w");
    print_imagenet_detections(fp, 7, boxes, probs, total, classes, w, h);
    fclose(fp);
    return 0;
}

/* denorm <cfg> <weights> <out.weights>: the reference's weight surgery, denormalize_net (darknet.c:309-345) for the
 * convolutional layers: denormalize_convolutional_layer (convolutional_layer.c:321) + batch_normalize = 0 + save_weights */
void denormalize_convolutional_layer(layer l);
static int cmd_denorm(int argc, char **argv)
{
    if (argc < 5) return 2;
    network net = parse_network_cfg(argv[2]);
    load_weights(&net, argv[3]);
    int i;
    for (i = 0; i < net.n; ++i) {
        layer l = net.layers[i];
        if (l.type == CONVOLUTIONAL && l.batch_normalize) {
            denormalize_convolutional_layer(l);
            net.layers[i].batch_normalize = 0;
        }
    }
    save_weights(net, argv[4]);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: ref_driver net|time|prim ...\n"); return 2; }
    if (!strcmp(argv[1], "net")) return cmd_net(argc, argv);
    if (!strcmp(argv[1], "time")) return cmd_time(argc, argv);
    if (!strcmp(argv[1], "prim")) return cmd_prim(argc, argv);
    if (!strcmp(argv[1], "evalw")) return cmd_evalw(argc, argv);
    if (!strcmp(argv[1], "denorm")) return cmd_denorm(argc, argv);
    return 2;
}
