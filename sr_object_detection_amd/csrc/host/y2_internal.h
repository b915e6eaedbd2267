/* Internal declarations shared by the host C sources of libsr_yolo2.so. */
#ifndef Y2_INTERNAL_H
#define Y2_INTERNAL_H

#include <stddef.h>
#include <stdint.h>
#include "sr_yolo2.h"
#include "y2_hip.h"

/* per-layer device-side state (layer.dev) */
typedef struct y2_ldev {
    struct y2_engine *eng;
    int index;
    /* where this layer's activations live: NHWC, `ld` floats between pixels */
    float *out;
    int out_ld;
    float *out_alloc;          /* allocation owned by this layer, NULL when the output sits in another buffer */
    size_t out_floats;
    int placed_in;             /* index of the [route] layer whose buffer holds this output, or -1 */
    int alias_of;              /* [route] with one input / [cost]: index of the layer whose output is reused, or -1 */
    unsigned copy_mask;        /* [route]: bit k set -> input k needs a copy kernel (could not be placed) */
    /* convolution parameters inside the weight arena */
    size_t off_w_packed, off_w_ref, off_bias, off_mean, off_scale, off_rinv;
    int has_w_ref;
    int uses_mfma;
    int fused_pool;            /* conv: the following 2x2/2 maxpool runs in this conv's epilogue */
    int fused_into;            /* maxpool: index of the conv that computes it, or -1 */
    int out_half;              /* this layer's activations are IEEE half (fp16 mode), out_ld counts halves */
    size_t off_alpha, off_beta; /* fp16 mode: folded batch-norm, y = act(acc*alpha + beta) */
    char kname[80];
    int tile_bm, tile_bn, ksplit;   /* measured tile choice (y2_set_autotune), 0 = the host's cost model */
    /* region */
    float *d_anchors;
    int *d_tree_parent, *d_tree_gsize, *d_tree_goff, *d_map;
    int *d_tree_order, *d_tree_loff;   /* nodes by depth level (only when parents precede children) */
    int tree_levels;
    float *d_tree_best;                /* [2 * boxes]: the region layer's (score | class) per box for y2h_detect_tree_chain, or NULL */
    int tree_best_valid;               /* the last forward filled d_tree_best (it does unless the detection chain overlaps the next forward) */
    float *d_region;           /* [batch][outputs] flattened region output */
    /* classifier tail */
    float *d_flat;             /* avgpool / softmax output [batch][outputs] */
    /* [crop] in front of a few-channel convolution: a second copy of the window with a zero border of halo_px
     * pixels ([batch][out_h+2p][out_w+2p][out_c]), which is what the first-layer / stem kernels read */
    float *d_halo;
    int halo_px;
    float *d_bin;              /* xnor=1 convolution: its input binarized to +-1 ([batch][h][w][c], contiguous) */
    const char *kernel;        /* name for profiles */
} y2_ldev;

typedef struct y2_engine {
    int device;
    y2h_stream stream;
    int strict;
    int timing;
    int fusion, built_fusion;  /* conv+maxpool fusion enabled / state of the current plan */
    int half, built_half;      /* fp16 storage requested (y2_set_half) / state of the current plan */
    int autotune, built_autotune; /* measure the conv tile shapes at plan time (y2_set_autotune) */
    int in_halo;               /* the NHWC copy of the input carries a zero border (2: the half NHWC4 form; 3: no copy at all,
                                  the fp16 first layer reads the fp32 NCHW input) */
    const float *cur_input;    /* the NCHW input of the forward pass being enqueued (in_halo 3) */
    int in_halo_px;            /* its width in pixels: 1 for the 3x3 first-layer kernels, the padding for the stem kernel */
    /* hipGraph replay of the forward launch sequence (y2_set_graph): recorded for one input pointer, dropped with the plan */
    y2h_event ev_out;          /* recorded behind the output copy of y2_output_enqueue */
    int out_pending;
    y2h_event ev_det;          /* recorded behind the D2H copies of y2_detect_enqueue */
    /* y2_set_detect_overlap: decode / NMS / compaction of batch i on their own stream beside the forward pass of batch i+1 */
    int det_overlap;
    y2h_stream det_stream;
    y2h_event ev_fwd;          /* recorded on `stream` behind the forward pass whose region output the detect chain reads */
    int det_pending;           /* 1: wait for ev_det in y2_detect_fetch, 2: already fetched synchronously */
    int graph_on;
    y2h_graph graph;           /* the graph in use (one of graphs[]) */
    const float *graph_src;
    y2h_graph graphs[4];       /* recorded forward passes by input pointer (a double-buffered feed alternates between two) */
    const float *graph_srcs[4];
    int graph_next;            /* slot the next recording replaces */
    /* plan state */
    int built;
    int built_batch, built_w, built_h, built_strict;
    int weights_dirty;         /* host weights changed since the last upload */
    int weights_external;      /* arena filled from outside (broadcast) */
    /* weight arena */
    unsigned char *arena;
    size_t arena_bytes;
    int arena_pending;         /* the arena was laid out for a fill from outside (y2_weights_arena) that has not happened yet */
    int class_counts_zeroed;   /* d_class_counts is all zero (the three-launch detect chain keeps it so) */
    int capturing;             /* inside the hipGraph capture of a forward pass (no cross-stream waits may be recorded) */
    uint64_t arena_sig;        /* hash of the per-layer offsets / forms the arena was laid out with (0: none yet) */
    /* io buffers */
    float *d_in_nchw, *d_in_nhwc;
    size_t in_floats;
    float *d_out_nchw;         /* staging when the output layer is image-like */
    unsigned char *d_u8;       /* y2_detect_u8: raw frames, float planes, resize scratch (grow-only) */
    float *d_planes, *d_rtmp;
    size_t u8_cap, planes_cap, rtmp_cap;
    float *d_ws;               /* split-K scratch shared by all conv layers */
    size_t ws_bytes;
    float *h_out;              /* what network_predict returns; allocated at parse time like the reference's
                                  l.output (callers copy `layer` structs early), pinned once a GPU is in use */
    size_t out_floats;
    size_t h_out_cap;          /* floats allocated behind h_out */
    int h_out_pinned;          /* h_out came from hipHostMalloc (a GPU was present at parse time): predict copies straight into it */
    float *h_out_stage;        /* pinned (hipHostMalloc) landing buffer of the PIPELINED output copy (y2_output_enqueue / _fetch: the
                                  caller may still read h_out while the next copy flies) and of hosts without pinned h_out;
                                  heap memory is never handed to the GPU: registering h_out in place
                                  (hipHostRegister) made the runtime treat a pageable buffer that starts in the page behind
                                  it as part of the registration -- the next weight upload from such a buffer faulted */
    size_t h_out_stage_cap;
    int out_layer;
    /* decode / nms buffers for the output region layer */
    float *d_boxes, *d_probs, *d_probs_nms, *d_records;
    int *d_counts;
    int *d_class_counts;       /* [batch][classes] non-zero scores (NMS skips empty classes) */
    float *d_best;             /* [2][batch][total] best score / class per box */
    float *d_mean_ring;        /* y2_detect_mean: three region-output slots + their average (batch 1) */
    size_t mean_els;
    int mean_index;
    float *h_records;
    int *h_counts;
    int det_cap;               /* records per image */
    int det_batch, det_total, det_classes;
    /* pinned, multi-buffered host feed (y2_feed.c) */
    int feed_slots;
    size_t feed_bytes;
    void **feed_host, **feed_dev;
    y2h_event *feed_up, *feed_done;    /* per slot: H2D finished / the forward that read the device copy was enqueued and ran */
    int *feed_used;                    /* per slot: feed_done has been recorded at least once */
    y2h_stream feed_stream;            /* copies run here, next to the engine stream's kernels */
    /* timing */
    y2h_event *ev;             /* n+1 events */
    int n_ev;
    int n_layers;
} y2_engine;

/* error handling: mode 0 = the reference's contract (message + exit), 1 = record and return */
void y2_fail(const char *fmt, ...);
int y2_error_mode(void);
int y2_failed(void);           /* and clear */

/* engine */
y2_engine *y2_engine_of(const network *net);
int y2_engine_create(network *net);
void y2_engine_destroy(network *net);
void y2_engine_invalidate(network *net);
void y2_engine_host_output(network *net);
int y2_engine_build(network *net);
int y2_engine_forward(network *net, const float *d_input_nchw);
int y2_engine_fetch_output(network *net);
int y2_ingest_u8_device(network net, const unsigned char *d_frames, int h, int w, int c, int step, int swap_rb, int letterbox);

/* cfg helpers shared with other files */
char *y2_fgetl(FILE *fp);
void y2_strip(char *s);
int y2_out_layer(const network *net);

#endif
