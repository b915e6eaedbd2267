/*
 * y2_hip.h -- the thin C-ABI device layer of the MI355X (gfx950) YOLOv2 engine.
 *
 * Everything the C host code (csrc/host, C sources) needs from the GPU goes through
 * these entry points: plain pointers and sizes, no C++ or torch types.  It
 * replaces the reference's device wrapper and its per-layer *_gpu functions:
 *
 *   device/memory      src_yolo2/cuda.h:24-33  (cuda_set_device, cuda_make_array,
 *                      cuda_push_array, cuda_pull_array, cuda_free, check_error)
 *   conv+BN+bias+act   src_yolo2/convolutional_kernels.cu:77-131
 *                      (fill + im2col_ongpu + gemm_ongpu + normalize_gpu +
 *                       scale_bias_gpu + add_bias_gpu + activate_array_ongpu,
 *                       >= 6 launches per layer there; ONE launch here)
 *   maxpool            src_yolo2/maxpool_layer_kernels.cu:87-97
 *   reorg              src_yolo2/reorg_layer.c:97-104 (reorg_ongpu, blas_kernels.cu:332)
 *   route              src_yolo2/route_layer.c:104-117 (copy_ongpu per input per item)
 *   region head        src_yolo2/region_layer.c:383-417 (flatten_ongpu + softmax_gpu,
 *                      then D2H and logistic on the CPU there; all on device here)
 *   decode / NMS       src_yolo2/region_layer.c:328-379, src_yolo2/box.c:249-277
 *                      (CPU-only in the reference, both builds)
 *   avgpool / softmax  src_yolo2/avgpool_layer_kernels.cu:44, src_yolo2/softmax_layer.c:73
 *
 * Data layout: activations are NHWC fp32 with an explicit channel stride
 * (`ld`, in floats) so that [route] concatenation is a channel-offset write
 * instead of a copy.  Convolution weights are pre-packed per layer as
 * [Cout][kh][kw][Cin] (K index = (kh*size + kw)*Cin + ci).
 *
 * All functions return 0 on success or a negative Y2H_E* code; none aborts.
 * The legacy abort-on-error contract (cuda.c:27-49 check_error) is applied one
 * level up, in the host C layer.
 */
#ifndef Y2_HIP_H
#define Y2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Y2H_OK            0
#define Y2H_EHIP         -1   /* a HIP runtime call failed (y2h_last_error() has the text) */
#define Y2H_EINVAL       -2   /* argument outside what the kernels support */
#define Y2H_ENODEV       -3   /* no gfx950 device visible */

typedef void *y2h_stream;   /* hipStream_t */
typedef void *y2h_event;    /* hipEvent_t */
typedef void *y2h_graph;    /* hipGraphExec_t */

/* activation codes (src_yolo2/activations.h:7).  0-3 are the ones the target cfgs use and every kernel's epilogue
 * applies in place; the others only exist as the separate pass y2h_activate_array (the engine then runs the
 * producing kernel with Y2H_ACT_LINEAR, which is exactly the reference's own order: activate_array is a pass of its
 * own over the stored fp32 output, activations.c:95-101) */
enum { Y2H_ACT_LINEAR = 0, Y2H_ACT_LEAKY = 1, Y2H_ACT_LOGISTIC = 2, Y2H_ACT_RELU = 3,
       Y2H_ACT_RELIE = 4, Y2H_ACT_RAMP = 5, Y2H_ACT_TANH = 6, Y2H_ACT_PLSE = 7, Y2H_ACT_ELU = 8, Y2H_ACT_LOGGY = 9,
       Y2H_ACT_STAIR = 10, Y2H_ACT_HARDTAN = 11, Y2H_ACT_LHTAN = 12 };

/* ---- device / memory / streams (cuda.h:24-33) ---- */
int         y2h_device_count(void);
int         y2h_set_device(int dev);
int         y2h_get_device(int *dev);
const char *y2h_last_error(void);
const char *y2h_device_name(void);
/* PCI address ("0000:c1:00.0") of device `dev` (< 0: the current one), "" if unknown: lets a launcher pin a rank's host
 * threads to the NUMA node of its GPU (/sys/bus/pci/devices/<address>/numa_node) */
const char *y2h_device_pci_bus_id(int dev);
int         y2h_malloc(void **ptr, size_t bytes);
int         y2h_free(void *ptr);
int         y2h_host_alloc(void **ptr, size_t bytes);           /* pinned host memory */
int         y2h_host_free(void *ptr);
/* pin an existing host allocation.  HAZARD (profiles/r02_notes.md): only register buffers that own their pages -- mmap'ed
 * or page-aligned AND page-padded.  A registered range that ends inside a heap page makes the runtime treat a pageable
 * buffer that malloc placed in the rest of that page as pinned: an asynchronous copy from it then faults on the GPU.
 * The engine itself never registers memory (its staging buffers come from y2h_host_alloc). */
int         y2h_host_register(void *ptr, size_t bytes);
int         y2h_host_unregister(void *ptr);
int         y2h_memcpy_h2d(void *dst, const void *src, size_t bytes, y2h_stream s);
int         y2h_memcpy_d2h(void *dst, const void *src, size_t bytes, y2h_stream s);
int         y2h_memcpy_d2d(void *dst, const void *src, size_t bytes, y2h_stream s);
int         y2h_memset(void *dst, int value, size_t bytes, y2h_stream s);
int         y2h_stream_create(y2h_stream *s);
int         y2h_stream_destroy(y2h_stream s);
int         y2h_stream_sync(y2h_stream s);
int         y2h_device_sync(void);
/* the clock (GHz) this device holds under a full-chip fp32 matrix load: `iters` x 2 v_mfma_f32_32x32x2_f32 per wave on
 * every CU, in-kernel s_memtime / s_memrealtime stamps, median over waves (diagnostic: benchmark lines quote it) */
int         y2h_clock_probe(int iters, float *ghz, y2h_stream s);
int         y2h_event_create(y2h_event *e);
int         y2h_event_destroy(y2h_event e);
int         y2h_event_record(y2h_event e, y2h_stream s);
int         y2h_event_sync(y2h_event e);                                        /* wait for the event only */
int         y2h_stream_wait_event(y2h_stream s, y2h_event e);                    /* later work on s runs after e (device-side wait) */
int         y2h_event_elapsed_ms(y2h_event start, y2h_event stop, float *ms);   /* syncs on stop */

/* ---- hipGraph capture of a kernel sequence on a stream (replaces nothing in the reference: its GPU path launches
 * every kernel from the host each call) ---- */
int  y2h_graph_begin(y2h_stream s);                   /* start recording what is enqueued on s (nothing executes) */
int  y2h_graph_end(y2h_stream s, y2h_graph *out);      /* stop recording, instantiate */
void y2h_graph_abort(y2h_stream s);                   /* leave capture mode after an error, keep nothing */
int  y2h_graph_launch(y2h_graph g, y2h_stream s);
int  y2h_graph_destroy(y2h_graph g);

/* ---- layout ---- */
/* [n][c][h][w] -> [n][h][w][ld] (channels 0..c-1 of each pixel row) and back */
int y2h_nchw_to_nhwc(const float *src, float *dst, int n, int c, int h, int w, int ld, y2h_stream s);
int y2h_nhwc_to_nchw(const float *src, int ld, float *dst, int n, int c, int h, int w, y2h_stream s);
/* [n][c][h][w] -> interior of [n][h+2*halo][w+2*halo][ld]; the border is not written
 * (the caller zeroes the buffer once) */
int y2h_nchw_to_nhwc_halo(const float *src, float *dst, int n, int c, int h, int w, int ld, int halo, y2h_stream s);
/* copy `c` channels of `npix` pixels between two NHWC buffers ([route] fallback) */
int y2h_copy_channels(const float *src, int ld_src, float *dst, int ld_dst, int c, long npix, y2h_stream s);

/* ---- convolution + fused epilogue ---- */
typedef struct y2h_conv {
    int batch, h, w, c;          /* input  NHWC dims                                   */
    int ldx;                     /* input  channel stride (floats)                     */
    int x_halo;                  /* 0, or p > 0: x is [batch][h+2p][w+2p][ldx] with a zero border of p pixels
                                    (layout of y2h_nchw_to_nhwc_halo; first-layer / stem kernels only) */
    int n;                       /* filters (output channels)                          */
    int size, stride, pad;       /* square kernel                                      */
    int out_h, out_w;
    int ldy;                     /* output channel stride (floats)                     */
    int fuse_maxpool2;           /* 1: a 2x2 stride-2 pad-0 maxpool follows the activation inside
                                    the kernel; y is then [batch][out_h/2][out_w/2][ldy]
                                    (matrix-core kernels only, out_h and out_w even)   */
    int batch_normalize;         /* 1: (x-mean)*rinv*scale + bias ; 0: x + bias        */
    int activation;              /* Y2H_ACT_*                                          */
    const float  *x;             /* device, NHWC                                       */
    const float  *w_packed;      /* device, [n][size][size][c]                         */
    const float  *w_ref;         /* device, [n][c][size][size] (strict path only) or 0 */
    const float  *mean;          /* device [n]  rolling_mean        (BN only)          */
    const double *rinv;          /* device [n]  1/(sqrt((double)var)+1e-6f) (BN only)  */
    const float  *scale;         /* device [n]  scales              (BN only)          */
    const float  *bias;          /* device [n]                                         */
    float        *y;             /* device, NHWC (already offset to the first channel) */
    float        *ws;            /* device scratch for split-K partial sums, or 0      */
    size_t        ws_bytes;      /* size of ws; see y2h_conv_workspace_bytes           */
    /* fp16 storage (engine extension, the reference is fp32 only): */
    int           x_f16;         /* 1: x and w_packed hold IEEE half (ldx counts halves); fp32 accumulate on
                                    v_mfma_f32_32x32x16_f16; the epilogue is y = act(acc*alpha + beta)   */
    int           y_f16;         /* 1: y is stored as half (ldy counts halves)         */
    const float  *alpha;         /* device [n]  scale/(sqrt(var)+1e-6)  (1 without BN)  */
    const float  *beta;          /* device [n]  bias - mean*alpha                       */
    /* tile choice of the matrix-core kernel: 0 = the host's cost model decides (see y2h_conv_candidates) */
    int           tile_bm, tile_bn;   /* GEMM tile (output pixels x filters), one of the instantiated shapes */
    int           ksplit;             /* K ranges per output tile (1 = no split-K) */
    /* first layer only: x is the network input itself, fp32 planes [batch][c][h][w] (no NHWC copy) */
    int           x_nchw;
} y2h_conv;

/* Measure instead of model: the (tile_bm, tile_bn, ksplit) combinations worth timing for this descriptor, for callers
 * that measure in their own context (the engine, y2_set_autotune, times them inside whole forward passes, where the
 * caches hold what they hold in production); entry 0 is the cost model's choice.  Every tile shape accumulates each
 * output in the same K order, so the choice of tile never changes a result bit; a K-split does (partial sums are added
 * in a fixed order).  Returns the count (0: not an fp32 matrix-core convolution). */
int y2h_conv_candidates(const y2h_conv *d, int *bm, int *bn, int *ks, int max);

/* 1 if a first-layer kernel can read the fp32 NCHW network input directly (x_nchw = 1; 3 channels, 3x3/1 pad 1,
 * <= 64 filters; y_f16 as in the descriptor): no input transform kernel at all */
int y2h_conv_first_layer_nchw_ok(const y2h_conv *d);

/* which kernel y2h_conv_forward would pick: 1 = MFMA implicit GEMM, 0 = direct VALU */
int y2h_conv_uses_mfma(const y2h_conv *d);
/* bytes of split-K scratch y2h_conv_forward needs for this descriptor (0 = none): small grids
 * (13x13 maps at small batch, batch-1 inference) are cut along K so that all 256 CUs get work */
size_t y2h_conv_workspace_bytes(const y2h_conv *d);
/* Stream-K plan of the fp16 256x256 persistent kernel (replaces nothing in the reference: its GEMM is one cuBLAS call per
 * image, convolutional_kernels.cu:108-116): of `ntiles` output tiles of `nk` K-tiles each on a persistent grid of `grid`
 * workgroups, the last *sk_tiles are cut along K into equal shares for workgroups 0 .. *sk_wgs - 1 and finished by a
 * fix-up launch.  Returns 1 when the plan splits, 0 when every tile is walked whole.  Host arithmetic only (no GPU). */
int y2h_p8_stream_k_plan(long ntiles, int nk, long grid, int *sk_tiles, int *sk_wgs);
/* number of stream-K launches (main + fix-up pairs) issued by this process so far (tests, benchmark reports) */
unsigned long y2h_stream_k_launches(void);
/* number of small-tile tail launches behind the fp16 256x256 kernel (the other way to finish a partial last round) */
unsigned long y2h_tail_launches(void);
/* number of fp32 matrix-core launches that used the XCD-grouped tile order (wide heads: yolo9000's final 1x1) */
unsigned long y2h_xcd_order_launches(void);
/* number of fp32 matrix-core launches that used stream-K work items (grids smaller than the machine) */
unsigned long y2h_f32_stream_k_launches(void);
/* number of fp32 matrix-core launches that cut their partial last round along K over all workgroups and finished the
 * pieces inside the launch (hybrid stream-K: producer pieces publish raw sums write-through and raise a flag, the
 * tile's last piece adds them and runs the epilogue; no second launch) */
unsigned long y2h_f32_hybrid_stream_k_launches(void);
/* flag waits of such launches that gave up after seconds (0 always, unless a workgroup was lost; synchronous read) */
int y2h_f32_stream_k_timeouts(void);
/* 1 when the shape fits the dedicated first-layer kernel (3 channels, 3x3/1 pad 1, <= 64
 * filters) provided the input is supplied with a halo (x_halo = 1) */
int y2h_conv_first_layer_ok(const y2h_conv *d);
/* halo width (>= 0) the few-channel stem kernel wants for this shape (any size / stride with pad as the halo, c <= 4,
 * <= 128 filters: the 7x7/2 stems of cfg/yolov1/yolo.cfg, resnet50.cfg, extraction.cfg), or -1 when the shape does
 * not fit it; pass the descriptor back with x_halo set to that value */
int y2h_conv_stem_halo(const y2h_conv *d);
/* same for the fp16 first-layer kernel, whose input is [batch][h+2][w+2][4] halves (x_f16 = 1, ldx = 4,
 * layout of y2h_nchw_to_nhwc4_halo_f16) and whose weights stay the fp32 packed [n][27] */
int y2h_conv_first_layer_f16_ok(const y2h_conv *d);
/* strict != 0 forces the direct kernel, which accumulates in the reference's exact
 * order (ci, kh, kw ascending; product and sum rounded separately: gemm.c:74-88)
 * and is therefore bit-identical to the CPU path; needs w_ref. */
int y2h_conv_forward(const y2h_conv *d, int strict, y2h_stream s);
/* name of the kernel variant last chosen for this descriptor (for profiles) */
const char *y2h_conv_variant(const y2h_conv *d, int strict);

/* ---- other layers (NHWC) ---- */
int y2h_maxpool(const float *x, int ldx, float *y, int ldy, int batch, int h, int w, int c,
                int size, int stride, int pad, int out_h, int out_w, y2h_stream s);
/* the reference's reorg quirk (blas.c:8-29 called with forward=0 for a non-reverse
 * layer, reorg_layer.c:83), re-expressed for NHWC in and out; reverse!=0 is forward=1 */
int y2h_reorg(const float *x, int ldx, float *y, int ldy, int batch, int h, int w, int c,
              int stride, int reverse, y2h_stream s);
/* [connected] in the reference's accumulation order (connected_layer.c:141-176 / gemm.c:90-106 gemm_nt): input element
 * k = c*hw + p of batch item b is x[b*x_batch_stride + p*ld + c] (an NHWC producer; hw = 1 for a flat one), weights
 * w_ref [outputs][hw*c] in the reference layout; BN / bias / activation as for a convolution */
int y2h_connected_ref(const float *x, long x_batch_stride, int ld, int hw, int c, const float *w_ref, float *y,
                      int outputs, int batch, int batch_normalize, int activation, const float *mean,
                      const double *rinv, const float *scale, const float *bias, y2h_stream s);
/* YOLOv1 head decode (detection_layer.c:222-251): pred = [batch] blocks of pred_stride floats; boxes
 * [batch][side*side*num][4], probs [batch][side*side*num][classes] */
int y2h_detection_boxes(const float *pred, long pred_stride, int batch, int side, int num, int classes, int sqrt_flag,
                        int w, int h, float thresh, int only_objectness, float *boxes, float *probs, y2h_stream s);
/* residual add (shortcut_layer.c:38-43, blas.c:57-81): out = act(in + add) where the shapes overlap; `add` is
 * w1 x h1 x c1, in/out are w2 x h2 x c2; stride = w1/w2 and sample = w2/w1 (each >= 1) as in shortcut_cpu */
int y2h_shortcut(const float *in, int ld_in, const float *add, int ld_add, float *out, int ld_out, int batch,
                 int w1, int h1, int c1, int w2, int h2, int c2, int activation, y2h_stream s);
/* [crop] at inference (crop_layer.c:69-105, !state.train): centred out_h x out_w window, x*2-1 unless noadjust.
 * halo > 0: y is [batch][out_h+2*halo][out_w+2*halo][ldy] and only its interior is written (the caller zeroes it once) */
int y2h_crop(const float *x, int ldx, float *y, int ldy, int batch, int h, int w, int c, int out_h, int out_w,
             int noadjust, int halo, y2h_stream s);
/* standalone [batchnorm] at inference (batchnorm_layer.c:122-146): ((x - mean) * rinv) * scale per channel, rinv =
 * 1 / (sqrt(var) + 1e-6f) prepared in double (blas.c:122) */
int y2h_batchnorm(const float *x, int ldx, float *y, int ldy, long pixels, int c, const float *mean, const double *rinv,
                  const float *scale, y2h_stream s);
/* [local] (local_layer.c:95-126): per-location filter banks.  w_packed [location][filter][kh][kw][c], bias_packed
 * [location][filter] (the engine re-orders the reference's [location][filter][c][kh][kw] / [filter][location]);
 * strict = 1 runs the reference's summation order (bias first, taps in c,kh,kw order, one rounding per step) */
int y2h_local(const float *x, int ldx, const float *w_packed, const float *bias_packed, float *y, int ldy, int batch,
              int h, int w, int c, int n, int size, int stride, int pad, int out_h, int out_w, int activation,
              int strict, y2h_stream s);
/* binarize_cpu (convolutional_layer.c:52-58) for xnor=1 convolutions: y[row][k] = x[row*ldx + k] > 0 ? 1 : -1, y contiguous */
int y2h_binarize(const float *x, int ldx, float *y, long rows, int c, y2h_stream s);
/* activate_array (activations.c:95-101, the formulas of activations.h:21-54) in place on channels 0..c-1 of `rows`
 * pixels with channel stride ld; any Y2H_ACT_* code */
int y2h_activate_array(float *x, int ld, long rows, int c, int activation, y2h_stream s);
/* global average pool: [batch][h*w][ld] -> [batch][c] (sequential fp32 sum, avgpool_layer.c:40) */
int y2h_avgpool(const float *x, int ldx, float *y, int batch, int h, int w, int c, y2h_stream s);
/* rows of `n` floats: softmax with temperature (blas.c:205); in/out may alias */
int y2h_softmax_rows(const float *x, float *y, long rows, int n, float temp, y2h_stream s);

/* ---- half-storage variants (engine extension: BASELINE configs[4]; void* = IEEE half, ld in halves) ---- */
int y2h_maxpool_f16(const void *x, int ldx, void *y, int ldy, int batch, int h, int w, int c,
                    int size, int stride, int pad, int out_h, int out_w, y2h_stream s);
int y2h_reorg_f16(const void *x, int ldx, void *y, int ldy, int batch, int h, int w, int c,
                  int stride, int reverse, y2h_stream s);
int y2h_copy_channels_f16(const void *src, int ld_src, void *dst, int ld_dst, int c, long npix, y2h_stream s);
int y2h_avgpool_f16(const void *x, int ldx, float *y, int batch, int h, int w, int c, y2h_stream s);  /* fp32 sum and result */
int y2h_nhwc_f16_to_nchw(const void *src, int ld, float *dst, int n, int c, int h, int w, y2h_stream s);
/* fp32 [n][3][h][w] -> interior of half [n][h+2][w+2][4] (channel 3 = 0; the border is not written) */
int y2h_nchw_to_nhwc4_halo_f16(const float *src, void *dst, int n, int c, int h, int w, y2h_stream s);
int y2h_f32_to_f16(const float *src, void *dst, long n, y2h_stream s);
int y2h_f16_to_f32(const void *src, float *dst, long n, y2h_stream s);

/* ---- region head ---- */
/* x: last conv output NHWC [batch][h*w][ldx] holding num*(coords+1+classes) channels.
 * y: [batch][h*w*num][coords+1+classes] (the reference's flattened layout).
 * logistic on objectness, softmax (softmax!=0) or per-group tree softmax
 * (group_size/group_offset device arrays, groups>0) on the class scores. */
int y2h_region_forward(const float *x, int ldx, float *y, int batch, int hw, int num, int classes, int coords,
                       int softmax, int groups, const int *group_size, const int *group_offset, y2h_stream s);
/* The same for a tree head (groups > 0), with two options.
 * best != 0: from the class row still in LDS, what get_region_boxes needs of it in detect mode (region_layer.c:351-367
 *   without a map; tree.c:37-44): best[0 .. boxes) = hierarchy probability of the deepest class above .5 (0: none),
 *   best[boxes .. 2 boxes) = that class as int.  Independent of the detection threshold; y is written exactly as without.
 *   parent / order / level_off / levels: the tree with its nodes listed by depth level (parents precede children);
 *   y2h_region_tree_best_ok: 1 when the row fits the LDS kernel.
 * flags & Y2H_REGION_FAST_EXP: the group softmax takes expf (single precision, ~1 ulp) where the reference rounds a double
 *   exp to float (blas.c softmax): outputs differ by ~1e-7 relative, sums run in the reference's order.  The double exps are
 *   what the layer costs (9418 per box in yolo9000).  Env Y2_REGION_EXP_DOUBLE=1 overrides the flag. */
#define Y2H_REGION_FAST_EXP 1
int y2h_region_tree_best_ok(int classes, int levels);
int y2h_region_forward_tree(const float *x, int ldx, float *y, int batch, int hw, int num, int classes, int coords,
                            int groups, const int *group_size, const int *group_offset, const int *parent,
                            const int *order, const int *level_off, int levels, float *best, int flags, y2h_stream s);

typedef struct y2h_decode {
    int batch, w, h, num, classes;
    int img_w, img_h;               /* the (w,h) scale arguments of get_region_boxes   */
    float thresh;
    int only_objectness;
    int classfix;
    const float *anchors;           /* device [2*num]  (l.biases)                      */
    const int   *tree_parent;       /* device [classes] or 0 (softmax_tree)            */
    const int   *tree_order;        /* device [classes]: node ids sorted by depth, or 0; with
                                       tree_level_off [levels+1] and tree_levels > 0 the tree is
                                       walked level by level (valid only when every parent index
                                       is smaller than its children's: the caller checks)       */
    const int   *tree_level_off;
    int          tree_levels;
    const int   *map;               /* device [200] or 0                               */
    float       *pred;              /* device region output [batch][w*h*num][5+classes];
                                       the tree branch updates it in place, as the
                                       reference does (region_layer.c:350)            */
    float       *boxes;             /* device out [batch][w*h*num][4]                  */
    float       *probs;             /* device out [batch][w*h*num][classes]            */
} y2h_decode;
int y2h_region_boxes(const y2h_decode *d, y2h_stream s);

/* per-class sort + greedy suppression on device arrays, box.c:249-277.
 * probs rows have `stride` floats; only the first `classes` columns take part.
 * `probs_in` holds the scores and is only read (the tie order of the reference's
 * repeated stable sort depends on the original scores of earlier classes);
 * `probs` must be a separate copy of it, in which suppressed scores are zeroed.
 * `class_counts` is device scratch of batch*classes ints (non-zero scores per image and class: empty
 * classes are skipped). */
int y2h_nms_sort(const float *boxes, const float *probs_in, float *probs, int batch, int total, int classes,
                 int stride, float thresh, int *class_counts, y2h_stream s);
/* The chain get_region_boxes -> do_nms_sort -> compaction of a plain region head (no tree, no map, <= 256 classes) in three
 * (four for more than a million scores) launches: y2h_region_boxes + y2h_nms_sort (nms > 0) + y2h_collect with the same results (region_layer.c:328-379,
 * box.c:249-277, yolo_v2_class.cpp:221-238).  q->thresh is both the decode and the collect threshold.  `probs_nms` is the
 * copy the NMS suppresses in (the final scores when nms > 0; q->probs keeps the unsuppressed ones), `class_counts` is
 * batch * classes ints that must be zero on entry and are zero again on return.  y2h_detect_chain_ok: 1 if `q` qualifies. */
int y2h_detect_chain_ok(const y2h_decode *q);
int y2h_detect_chain(const y2h_decode *q, float nms, float *probs_nms, int *class_counts, float *records, int *counts,
                     int max_per_image, float *best_scratch /* 2 * batch * w*h*num floats, as y2h_collect, or 0 */, y2h_stream s);
/* The same chain for a TREE head without a class map (yolo9000 in detect mode) in two launches.  region_layer.c:351-367
 * leaves at most one non-zero score per box (the deepest class above .5, kept when the objectness exceeds q->thresh), so
 * the dense [boxes][classes] arrays of y2h_region_boxes / y2h_nms_sort / y2h_collect (261 MB per batch of yolo9000 544 b8,
 * walked six times) shrink to one (class, score) pair per box: same records and counts, no dense scores, and the prediction
 * rows are NOT edited in place.  Needs the level-ordered tree (tree_order / tree_level_off), q->thresh >= 0, no
 * only_objectness, <= 4096 boxes per image.  best_scratch: 2 * batch * w*h*num floats.  With `tree_best` (the region layer's
 * own by-product for these predictions) the class rows are not read again at all: three small launches. */
int y2h_detect_tree_chain_ok(const y2h_decode *q);
int y2h_detect_tree_chain(const y2h_decode *q, float nms, float *records, int *counts, int max_per_image, float *best_scratch,
                          const float *tree_best /* y2h_region_forward_tree's `best` for q->pred, or 0 */, y2h_stream s);
/* class-agnostic variant, box.c:279-298 */
int y2h_nms(const float *boxes, float *probs, int batch, int total, int classes, int stride,
            float thresh, y2h_stream s);

/* compact detections per image in ascending box order (yolo_v2_class.cpp:221-238,
 * image.c:662-738): record = {x,y,w,h,prob,class} (6 floats); counts[b] = number found
 * (may exceed max_per_image; only the first max_per_image are stored). */
int y2h_collect(const float *boxes, const float *probs, int batch, int total, int classes, int stride,
                float thresh, float *records, int *counts, int max_per_image,
                float *best_scratch /* device, 2*batch*total floats */, y2h_stream s);

/* avg[i] = (0 + f[0][i] + f[1][i] + ... ) / n over n frames of `els` floats laid out back to back, summed in frame
 * order like utils.c:420-432 mean_arrays */
int y2h_mean_frames(const float *frames, int n, long els, float *avg, y2h_stream s);

/* separable align-corners bilinear resize of a CHW image (image.c:1950-1992) */
int y2h_resize_chw(const float *src, int c, int ih, int iw, float *tmp, float *dst, int h, int w, y2h_stream s);

/* ---- frame ingest: the steps in front of network_predict, on the device ---- */
/* `batch` 8-bit interleaved frames (h x w x c, row pitch `step` bytes, `frame_bytes` between frames) ->
 * float planes: dst[b][k][y][x] = (float)(src[..][k'] / 255.), k' = k with planes 0 and 2 exchanged when
 * swap_rb (BGR->RGB); only the first `planes` planes are written (a BGRA frame's alpha can be dropped).
 * yolo_v2_class.hpp:94-113,133-141; image.c:2045-2067,1181 */
int y2h_u8_to_planes(const unsigned char *src, int batch, int h, int w, int c, long step, long frame_bytes,
                     int planes, int swap_rb, float *dst, y2h_stream s);
int y2h_fill(float *dst, long n, float v, y2h_stream s);                                   /* image.c:1601 */
int y2h_embed_chw(const float *src, int c, int sh, int sw, float *dst, int dh, int dw, int dx, int dy,
                  y2h_stream s);                                                            /* image.c:1087 */
void y2h_letterbox_dims(int iw, int ih, int w, int h, int *new_w, int *new_h);             /* image.c:1607-1618 */
/* letterbox_image (image.c:1624): tmp holds c*ih*new_w + c*new_h*new_w floats */
int y2h_letterbox_chw(const float *src, int c, int ih, int iw, float *tmp, float *dst, int h, int w, y2h_stream s);

#ifdef __cplusplus
}
#endif
#endif /* Y2_HIP_H */
