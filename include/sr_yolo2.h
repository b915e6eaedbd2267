/*
 * sr_yolo2.h -- drop-in C API of the MI355X-native YOLOv2 / Darknet forward engine.
 *
 * This header declares, with the reference's own names, argument meaning and
 * error behaviour, the entry points a caller of the reference's forward path
 * binds.  Each declaration cites the reference
 * interface it replaces (paths relative to src_yolo2/ of
 * NidhiMishra/SR_object_detection).  The thin headers network.h, parser.h,
 * layer.h, box.h, region_layer.h, cuda.h, image.h, utils.h, option_list.h,
 * tree.h and test_detector.h next to this file simply include it, so the
 * Kinect / CLI callers keep their #include lines.
 *
 * Differences a caller can observe (all deliberate):
 *   - `layer` and `network` keep every field the forward-path callers read
 *     (yolo_v2_class.cpp:64-76,202-216; KinectUtil.cpp:85; detector.c:568-576)
 *     but drop the training-only members, and their layout no longer depends
 *     on -DGPU / -DCUDNN (layer.h:205-263, network.h:63-66): device state
 *     hangs off one opaque pointer.
 *   - there is no CPU compute path in this library: network_predict always
 *     runs on the GPU selected by net.gpu_index (>= 0).  gpu_index < 0, which
 *     selects the CPU path in the reference (network.c:461), is an error here.
 *   - only the last non-[cost] layer has a host `output` buffer; other layers'
 *     activations stay in HBM (use y2_pull_layer_output to inspect them).
 *   - set_batch_network may grow the batch (the reference overflows its
 *     parse-time buffers); buffers are re-planned lazily.
 *   - net.seen is 8 bytes, so a version >= 0.2 .weights header no longer
 *     overruns it (parser.c:1027-1029 vs network.c:137).
 */
#ifndef SR_YOLO2_H
#define SR_YOLO2_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums: names and order are part of the ABI callers compile against ---- */
/* activations.h:7-9 */
typedef enum {
    LOGISTIC, RELU, RELIE, LINEAR, RAMP, TANH, PLSE, LEAKY, ELU, LOGGY, STAIR, HARDTAN, LHTAN
} ACTIVATION;

/* layer.h:13-38 */
typedef enum {
    CONVOLUTIONAL, DECONVOLUTIONAL, CONNECTED, MAXPOOL, SOFTMAX, DETECTION, DROPOUT, CROP, ROUTE, COST,
    NORMALIZATION, AVGPOOL, LOCAL, SHORTCUT, ACTIVE, RNN, GRU, CRNN, BATCHNORM, NETWORK, XNOR, REGION, REORG, BLANK
} LAYER_TYPE;

/* layer.h:40-42 */
typedef enum { SSE, MASKED, SMOOTH } COST_TYPE;

/* tree.h:4-14 */
typedef struct {
    int *leaf;
    int n;
    int *parent;
    int *group;
    char **name;
    int groups;
    int *group_size;
    int *group_offset;
} tree;

/* box.h:4-6 */
typedef struct { float x, y, w, h; } box;

/* image.h:12-17: CHW planes, values in [0,1] */
typedef struct { int h; int w; int c; float *data; } image;

/* utils.h:14-28: what test_detector_img hands to the Kinect pipeline */
typedef struct {
    float x, y;
    float w, h;
    char name[20];
    float prob;
    int objClass;
    float CameraX, CameraY, CameraZ;
    float CameraWidth, CameraHeight;
    unsigned char flagBelong2Person;
    float boxRGB[3];
    int bodyId;
} object;

/* list.h / option_list.h: key=value lists returned by read_data_cfg */
typedef struct node { void *val; struct node *next; struct node *prev; } node;
typedef struct list { int size; node *front; node *back; } list;
typedef struct { char *key; char *val; int used; } kvp;

struct layer;
typedef struct layer layer;
struct network;

/* layer.h:44-264, forward-path subset.  Host-visible description of one layer. */
struct layer {
    LAYER_TYPE type;
    ACTIVATION activation;
    COST_TYPE cost_type;
    int batch_normalize;
    int batch;
    int flipped;
    int inputs, outputs;
    int h, w, c;
    int out_h, out_w, out_c;
    int n;                     /* filters | anchors | number of route inputs */
    int groups;
    int size, stride, pad;
    int reverse;
    int index;
    int binary, xnor;
    /* [region] */
    int softmax, classes, coords, classfix, log, sqrt, max_boxes, rescore, bias_match, random, absolute, truths;
    float jitter, thresh, coord_scale, object_scale, noobject_scale, class_scale;
    /* [softmax] / [cost] */
    float temperature, scale;
    int dontload, dontloadscales;
    tree *softmax_tree;
    int *map;
    int *input_layers, *input_sizes;      /* [route] */
    float *biases, *scales, *weights;     /* host copies, reference layouts ([n][c][k][k]) */
    float *rolling_mean, *rolling_variance;
    float *output;                        /* host; non-NULL only for the network's output layer */
    float *delta;                         /* always NULL (no training) */
    float *cost;
    size_t workspace_size;                /* the reference's im2col bytes (convolutional_layer.c:135); informational */
    void *dev;                            /* opaque device-side state */
    /* YOLOv1 family: [detection] grid side and `forced`, [dropout] probability */
    int side, forced;
    float probability;
    /* [crop] (crop_layer.c:16-46; `scale` above holds crop_height / h for resize_crop_layer) */
    int flip, noadjust;
    float angle, saturation, exposure, shift;
};

/* network.h:19-67, forward-path subset */
typedef struct network {
    float *workspace;          /* always NULL: there is no im2col workspace */
    int n;
    int batch;
    int *seen;                 /* 8 bytes are allocated behind this pointer */
    int subdivisions;
    float learning_rate, momentum, decay;
    int max_batches, time_steps;
    layer *layers;
    int outputs;
    float *output;
    int inputs;
    int h, w, c;
    int gpu_index;
    tree *hierarchy;
    void *engine;              /* opaque: plan, HBM buffers, stream */
} network;

/* network.h:69-77 kept for source compatibility of callers that mention the type */
typedef struct network_state {
    float *truth;
    float *input;
    float *delta;
    float *workspace;
    int train;
    int index;
    network net;
} network_state;

/* ---- device selection (cuda.h:8,28; cuda.c:1,12) ---- */
extern int gpu_index;                          /* default device for parse_network_cfg; 0 at start-up */
void cuda_set_device(int n);

/* ---- cfg + weights (parser.h:5-11) ---- */
network parse_network_cfg(char *filename);                       /* parser.c:585 */
void load_weights(network *net, char *filename);                 /* parser.c:1084 */
void load_weights_upto(network *net, char *filename, int cutoff);/* parser.c:1009 */
void denormalize_convolutional_layer(layer l);                   /* convolutional_layer.c:321 */
void y2_denormalize_network(network *net);                       /* darknet.c:309-345 denormalize_net, conv layers */
void save_weights(network net, char *filename);                  /* parser.c:878  (version 0.1 header) */
void save_weights_upto(network net, char *filename, int cutoff); /* parser.c:822 */

/* ---- network runtime (network.h:83-127) ---- */
network make_network(int n);                                     /* network.c:132 */
void free_network(network net);                                  /* network.c:592 */
void set_batch_network(network *net, int b);                     /* network.c:308 */
int resize_network(network *net, int w, int h);                  /* network.c:322 */
float *network_predict(network net, float *input);               /* network.c:458 */
float *network_predict_gpu(network net, float *input);           /* network_kernels.cu:392 */
float *get_network_output(network net);                          /* network.c:173 */
float *get_network_output_gpu(network net);                      /* network_kernels.cu:385 */
int get_network_output_size(network net);                        /* network.c:390 */
int get_network_input_size(network net);                         /* network.c:397 */
void top_predictions(network net, int k, int *index);            /* network.c:449 */
char *get_layer_string(LAYER_TYPE a);                            /* network.c:73 */

/* ---- region head hand-off (region_layer.h:12, box.h:13-17) ---- */
void get_region_boxes(layer l, int w, int h, float thresh, float **probs, box *boxes, int only_objectness, int *map);
void get_detection_boxes(layer l, int w, int h, float thresh, float **probs, box *boxes, int only_objectness);   /* detection_layer.c:222 (YOLOv1 head) */
void do_nms_sort(box *boxes, float **probs, int total, int classes, float thresh);   /* box.c:249 */
void do_nms(box *boxes, float **probs, int total, int classes, float thresh);        /* box.c:279 */
float box_iou(box a, box b);                                                         /* box.c:94  */
box float_to_box(float *f);                                                          /* box.c:5   */

/* ---- Kinect-pipeline entry (test_detector.h:2, detector.c:558) ---- */
void test_detector_img(char **names, image **alphabet, network net, image im, float thresh,
                       object *RecObects, int *objectNumPerFrame);

/* ---- evaluation writers (detector.c:169-243) and the validate loops over in-memory frames ---- */
int get_coco_image_id(char *filename);                                                    /* detector.c:169 */
void print_cocos(FILE *fp, char *image_path, box *boxes, float **probs, int num_boxes, int classes, int w, int h);
void print_detector_detections(FILE **fps, char *id, box *boxes, float **probs, int total, int classes, int w, int h);
void print_imagenet_detections(FILE *fp, int id, box *boxes, float **probs, int total, int classes, int w, int h);
void print_yolo_detections(FILE **fps, char *id, box *boxes, float **probs, int total, int classes, int w, int h);   /* yolo.c:95 */
char *basecfg(char *cfgfile);                                                             /* utils.c:121 */
/* validate_detector (detector.c:245-368) with the image list replaced by `n` network-sized CHW frames in
 * memory: per frame network_predict -> get_region_boxes(l, orig_w, orig_h, .005, .., 0, map) ->
 * do_nms_sort(.45) -> the writer `eval` selects: "voc" (default; <prefix>/comp4_det_test_<name>.txt per
 * class, id = basecfg(path)), "coco" (<prefix>/coco_results.json), "imagenet" (200 classes,
 * <prefix>/imagenet-detection.txt).  A network ending in [detection] (YOLOv1) follows validate_yolo (yolo.c:116-200)
 * instead: get_detection_boxes at .001, do_nms_sort(.5), print_yolo_detections (voc files only).
 * Frames are processed net.batch at a time.  Returns 0 / -1. */
int y2_validate_detector_frames(network net, float *frames, int n, char **paths, int *orig_w, int *orig_h,
                                char *eval, char *prefix, char **names, int *map);
/* validate_detector_recall (detector.c:371-450): thresh .2, objectness-only decode, do_nms(.., 1, .4), IoU .5
 * against truth[truth_first[f] .. truth_first[f+1]) (relative centre-form boxes); prints the reference's
 * progress line per frame on stderr and returns the running totals. */
typedef struct { int total, correct, proposals; float avg_iou; } y2_recall;
int y2_validate_recall_frames(network net, float *frames, int n, const box *truth, const int *truth_first, y2_recall *res);
/* validate_classifier_single (classifier.c:469-529) over `n` network-sized CHW frames in memory: truth[f] is the class
 * the reference derives from the file path (-1 = none); per frame network_predict -> top_k -> running top-1 / top-k
 * accuracy, the reference's progress line on stdout; the final averages are returned.  Returns 0 / -1. */
int y2_validate_classifier_frames(network net, float *frames, int n, const int *truth, int classes, int topk,
                                  float *top1_out, float *topk_out);

/* ---- small helpers the callers use (option_list.h:12-19, data.c:474, utils.c, tree.c, image.c) ---- */
list *read_data_cfg(char *filename);
char *option_find(list *l, char *key);
char *option_find_str(list *l, char *key, char *def);
int option_find_int(list *l, char *key, int def);
int option_find_int_quiet(list *l, char *key, int def);
float option_find_float(list *l, char *key, float def);
float option_find_float_quiet(list *l, char *key, float def);
void free_list(list *l);
char **get_labels(char *filename);
image **load_alphabet(void);               /* glyph PNGs are UI (image.c:212): returns NULL here */
tree *read_tree(char *filename);           /* tree.c:53 */
int *read_map(char *filename);             /* utils.c:17 */
int max_index(float *a, int n);            /* utils.c:533 */
void top_k(float *a, int n, int k, int *index);   /* utils.c:179 */
void mean_arrays(float **a, int n, int els, float *avg);   /* utils.c:420 */
image make_image(int w, int h, int c);     /* image.c:1436 */
void free_image(image m);                  /* image.c:2245 */
image resize_image(image im, int w, int h);/* image.c:1950; runs on the GPU */
image letterbox_image(image im, int w, int h);                   /* image.c:1624; runs on the GPU */
void letterbox_image_into(image im, int w, int h, image boxed);  /* image.c:1607; runs on the GPU */
float get_color(int c, int x, int max);    /* image.c:33 */
void error(const char *s);                 /* utils.c:195: perror + exit(-1) */
void file_error(char *s);                  /* utils.c:208: message + exit(0) */

/* ------------------------------------------------------------------------- */
/* Extensions (not in the reference): batched, HBM-resident operation        */
/* ------------------------------------------------------------------------- */

/* one detection of y2_detect*: centre-form box scaled by (img_w,img_h) of the call */
typedef struct { float x, y, w, h, prob; int obj_id; } y2_det;

/* Build plan, allocate HBM and upload weights now instead of at the first predict.  Returns 0 or <0. */
int y2_prepare(network *net);
/* Force every convolution through the reference-order VALU kernel (bit-identical to the CPU path). */
void y2_set_strict(network *net, int strict);
/* conv -> 2x2/2 maxpool pairs are fused by default (the conv pools in its epilogue and the
 * full-resolution activation is never stored); 0 turns that off, e.g. to inspect every layer. */
void y2_set_fusion(network *net, int on);
/* Measure instead of model: when the plan is (re)built, every convolution that runs on the fp32 matrix cores times
 * each instantiated tile shape (and a few K-splits) inside whole forward passes and keeps the fastest (y2h_conv_candidates); the result
 * is remembered per layer shape for the life of the process.  Costs about a second per network at plan time.  A tile
 * shape never changes a result bit; a different K-split changes the last bits of that layer (fixed-order partial sums).
 * Off by default; env Y2_AUTOTUNE=1 turns it on for every network.  Ignored in strict mode. */
void y2_set_autotune(network *net, int on);
/* Throughput mode for callers that pipeline batches with y2_detect_enqueue / y2_detect_fetch: decode, NMS and
 * compaction of batch i run on a stream of their own, so their small grids (one workgroup per image and class group)
 * share the GPU with the forward pass of batch i+1 instead of idling it; the forward pass waits for them only before
 * its region layer overwrites the tensor they read.  Results are identical.  Off by default (a lone
 * y2_detect_resident call gains nothing and pays two more events); region heads only. */
void y2_set_detect_overlap(network *net, int on);
/* Record the forward pass's kernel launches into a hipGraph at the next call and replay it afterwards (one
 * hipGraphLaunch instead of 20-60 launches; for batch-1 callers such as test_detector_img).  The graph is tied to
 * the plan and to the input pointer: a resize / set_batch / mode switch or a different device input re-records.
 * Off by default; env Y2_GRAPH=1 turns it on for every network.  Ignored in strict mode and while layer timing is on. */
void y2_set_graph(network *net, int on);
/* The packed, kernel-layout weight arena (one allocation; what a multi-GPU launcher broadcasts). */
int y2_weights_arena(network *net, void **dev_ptr, size_t *bytes);
/* fp16 storage mode (no reference counterpart; BASELINE configs[4]): activations and packed weights are
 * IEEE half in HBM, convolutions accumulate in fp32 on the fp16 matrix cores, batch-norm is folded into one
 * fp32 fma, the region / avgpool / softmax heads stay fp32.  Takes effect at the next forward (the plan is
 * rebuilt).  Ignored while strict mode is on.  Env Y2_FP16=1 turns it on for every network. */
void y2_set_half(network *net, int on);
/* Declare the arena contents valid although load_weights was not called on this process
 * (e.g. it was filled by an RCCL broadcast from rank 0). */
void y2_weights_resident(network *net);
/* Multi-GPU (one process per GPU, frames sharded): replicate root's packed weights with ONE in-place RCCL broadcast of
 * the arena over xGMI -- replaces the host-staged distribute_weights of network_kernels.cu:240-250.  `comm` is an
 * ncclComm_t of the RCCL this process uses (y2_comm_library() names it); the three y2_comm_* helpers create one
 * without any other dependency: rank 0 calls y2_comm_unique_id and hands the 128 bytes to the other ranks by whatever
 * channel the launcher has (a file, a socket, an env variable), then every rank calls y2_comm_init_rank.  Root must
 * have called load_weights; the other ranks only parse the cfg.  Returns 0 or <0 (y2_last_error()). */
#define Y2_COMM_ID_BYTES 128
const char *y2_comm_library(void);
int y2_comm_unique_id(void *id_out);
int y2_comm_init_rank(void **comm, int nranks, const void *id, int rank, int device);
int y2_comm_destroy(void *comm);
int y2_broadcast_weights(network *net, void *comm, int root);
/* ranks of `comm` and this process's rank in it, as RCCL reports them (ncclCommCount / ncclCommUserRank) */
int y2_comm_count(void *comm, int *nranks, int *rank);
/* (layout signature, bytes) of the weight arena under the current plan.  y2_broadcast_weights compares them across the
 * ranks before it moves a byte (RCCL itself checks neither counts nor types); a launcher that replicates the arena by
 * other means must do the same. */
int y2_weights_layout(network *net, unsigned long long *signature, size_t *bytes);
/* C face of the C++ Detector class (include/yolo_v2_class.hpp) for FFI callers that cannot bind a C++ class (ctypes, cgo,
 * JNI).  y2_detector_detect = Detector::detect(image_t) [+ Detector::tracking when track != 0]: `chw` is a planar float
 * image in [0,1], up to `max` results are written to `out` in bbox_t's own layout (4 unsigned, float, 2 unsigned = 28
 * bytes each); returns the number of boxes (may exceed max) or < 0 with the message in y2_last_error().  nms < 0 keeps
 * the detector's current Detector::nms. */
void *y2_detector_create(const char *cfg, const char *weights, int gpu_id);
void  y2_detector_destroy(void *det);
int   y2_detector_net_size(void *det, int *w, int *h);
int   y2_detector_detect(void *det, const float *chw, int c, int h, int w, float thresh, int use_mean, float nms, int track,
                         void *out, int max);
/* Pinned, multi-buffered host feed (replaces the per-call cudaMalloc + pageable H2D + cudaFree of
 * network_kernels.cu:392-405): `slots` pairs of (pinned host buffer, HBM buffer) of slot_bytes each (0 = one batch of
 * float NCHW frames) and a copy stream.  The producer writes a batch into y2_feed_host(net, s), y2_feed_submit starts
 * its upload, y2_feed_forward / y2_feed_forward_u8 (camera frames [batch][h][step] bytes, as y2_ingest_u8) make the
 * forward wait ON THE DEVICE for that upload -- so batch i+1 crosses PCIe while batch i computes.  Follow with
 * y2_detect_enqueue / y2_detect_fetch.  y2_feed_wait_host blocks until a slot's pinned buffer may be overwritten. */
int y2_feed_open(network *net, int slots, size_t slot_bytes);
void y2_feed_close(network *net);
void *y2_feed_host(network net, int slot);
void *y2_feed_device(network net, int slot);
size_t y2_feed_slot_bytes(network net);
int y2_feed_submit(network net, int slot, size_t bytes);
int y2_feed_wait_host(network net, int slot);
int y2_feed_forward(network net, int slot);
int y2_feed_forward_u8(network net, int slot, int h, int w, int c, int step, int swap_rb, int letterbox);
/* network_predict with the input already in HBM (NCHW, batch*inputs floats).  The returned
 * pointer is the same host buffer network_predict returns. */
float *y2_network_predict_device(network net, const float *d_input);
/* The host copy of the output as two halves, for callers that keep the device busy across batches (classifiers):
 * y2_forward_device(i); y2_output_enqueue(); y2_forward_device(i+1); p = y2_output_fetch();  -- the fetch waits for
 * batch i's copy only and returns the same host buffer network_predict returns (valid until the next enqueue). */
int y2_output_enqueue(network net);
float *y2_output_fetch(network net);
/* Forward only (no host copy of the output); then decode+NMS+collect with y2_detect_resident. */
int y2_forward_device(network net, const float *d_input);
/* Decode + per-class NMS + compaction of the last forward, all on device; up to max_per_image
 * detections per image are written to dets[b*max_per_image ...], counts[b] = number found. */
int y2_detect_resident(network net, float thresh, float nms, int img_w, int img_h,
                       y2_det *dets, int *counts, int max_per_image);
/* Its two halves, for callers that keep the device busy across batches: y2_detect_enqueue(i); y2_forward_device(i+1);
 * y2_detect_fetch(i) -- the fetch waits for batch i's results only (an event behind their D2H copies), so the host-side
 * wait and unpacking overlap the next forward pass.  One enqueue may be outstanding per network. */
int y2_detect_enqueue(network net, float thresh, float nms, int img_w, int img_h);
int y2_detect_fetch(network net, y2_det *dets, int *counts, int max_per_image);
/* Same on the average of the last three forwards' region outputs (Detector::detect use_mean, yolo_v2_class.cpp:
 * 208-213): the three-slot ring and the average live in HBM; slots start zeroed like the reference's calloc. */
int y2_detect_mean(network net, float thresh, float nms, int img_w, int img_h,
                   y2_det *dets, int *counts, int max_per_image);
/* Host-input convenience: H2D + forward + y2_detect_resident. */
int y2_detect(network net, float *input, float thresh, float nms, int img_w, int img_h,
              y2_det *dets, int *counts, int max_per_image);
/* Camera-frame entry: `batch` 8-bit interleaved frames (h x w x c, row pitch `step` bytes; swap_rb
 * exchanges channels 0 and 2 = BGR->RGB) are uploaded as bytes, converted to [0,1] planes, resized
 * (letterbox != 0: letterboxed) to the network input on the device, then forward + y2_detect_resident.
 * Same result as ipl_to_image + rgbgr_image + resize_image/letterbox_image + the float path
 * (yolo_v2_class.hpp:94-141, yolo_v2_class.cpp:173-249).  y2_ingest_u8 stops after filling the
 * network's device input (follow with y2_forward_device(net, NULL)). */
/* Float CHW frame of any size (batch-1 networks): its first net.c planes go up and are resized on the device
 * straight into the network input -- resize_image + the input copy of network_predict (detector.c:567-573)
 * without a host round trip.  Follow with y2_forward_device(net, NULL) / y2_network_predict_device(net, NULL). */
int y2_ingest_image(network net, image im);
int y2_ingest_u8(network net, const unsigned char *frames, int h, int w, int c, int step, int swap_rb, int letterbox);
int y2_detect_u8(network net, const unsigned char *frames, int h, int w, int c, int step, int swap_rb, int letterbox,
                 float thresh, float nms, int img_w, int img_h, y2_det *dets, int *counts, int max_per_image);
/* Copy layer i's activations to host as NCHW [batch][out_c][out_h][out_w] (or [batch][outputs]). */
int y2_pull_layer_output(network net, int i, float *dst);
/* Per-layer device time of the last forward in ms (needs y2_set_timing(net,1)); returns layers written. */
void y2_set_timing(network *net, int on);
int y2_layer_times_ms(network net, float *ms, int max_layers);
/* Name of the kernel a layer runs ("conv_mfma_f32_128x128x32_k3", "maxpool", ...). */
const char *y2_layer_kernel(network net, int i);
/* Stream the engine launches on (hipStream_t) and a whole-device sync. */
void *y2_stream(network net);
void y2_sync(network net);
/* Last error text of the library (never NULL). */
const char *y2_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SR_YOLO2_H */
