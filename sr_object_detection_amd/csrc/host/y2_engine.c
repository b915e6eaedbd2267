/*
 * Network runtime: buffer plan, weight arena and the forward pass.
 *
 * Host-side replacement for src_yolo2/network.c (forward_network :145,
 * network_predict :458, set_batch_network :308, resize_network :322,
 * free_network :592) and src_yolo2/network_kernels.cu (forward_network_gpu :43,
 * network_predict_gpu :392), re-designed for one MI355X:
 *
 *  - activations are NHWC in HBM and never leave it between layers; there is
 *    no im2col workspace, no per-call cudaMalloc/H2D/cudaFree
 *    (network_kernels.cu:399,405), no per-layer fill of `delta`
 *    (network_kernels.cu:50-52);
 *  - [route] is planned away: a one-input route aliases its source, and the
 *    sources of a concatenating route write straight into the route's buffer
 *    at their channel offset (conv / maxpool / reorg kernels take an output
 *    channel stride), so the copy_ongpu launches of route_layer.c:104-117
 *    disappear; a copy kernel remains only for sources that cannot be placed;
 *  - all weights sit in ONE device allocation ("arena") in kernel layout
 *    ([n][kh][kw][c] filters + the per-filter epilogue constants), so that a
 *    multi-GPU launcher replicates the model with a single broadcast;
 *  - the plan is built lazily at the first predict after parse / resize /
 *    set_batch, which also lets set_batch_network grow the batch safely.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <pthread.h>
#include <string.h>
#include "y2_internal.h"

#define HIPCALL(expr) do { int rc_ = (expr); if (rc_ != 0) { y2_fail("%s failed (%d): %s", #expr, rc_, y2h_last_error()); return -1; } } while (0)

static y2_ldev *ld_of(const layer *l) { return (y2_ldev *)l->dev; }

static void drop_graphs(y2_engine *e)
{
    int k;
    for (k = 0; k < 4; ++k) {
        if (e->graphs[k]) y2h_graph_destroy(e->graphs[k]);
        e->graphs[k] = NULL; e->graph_srcs[k] = NULL;
    }
    e->graph = NULL; e->graph_src = NULL; e->graph_next = 0;
}

y2_engine *y2_engine_of(const network *net) { return net ? (y2_engine *)net->engine : NULL; }

int y2_engine_create(network *net)
{
    y2_engine *e = calloc(1, sizeof *e);
    int i;
    const char *st = getenv("Y2_STRICT");
    e->device = net->gpu_index;
    e->weights_dirty = 1;
    e->strict = (st && atoi(st) != 0) ? 1 : 0;
    e->fusion = getenv("Y2_NO_FUSE") ? 0 : 1;
    { const char *g = getenv("Y2_GRAPH"); e->graph_on = (g && atoi(g) != 0) ? 1 : 0; }
    { const char *hf = getenv("Y2_FP16"); e->half = (hf && atoi(hf) != 0) ? 1 : 0; }
    { const char *at = getenv("Y2_AUTOTUNE"); e->autotune = (at && atoi(at) != 0) ? 1 : 0; }
    e->n_layers = net->n;
    e->out_layer = y2_out_layer(net);
    for (i = 0; i < net->n; ++i) {
        y2_ldev *d = calloc(1, sizeof *d);
        d->eng = e;
        d->index = i;
        d->placed_in = -1;
        d->alias_of = -1;
        d->fused_into = -1;
        net->layers[i].dev = d;
    }
    net->engine = e;
    y2_engine_host_output(net);
    return 0;
}

/* (Re)allocate the host output buffer for the current batch/size and publish it as l.output of the output
 * layer.  Done at parse time and on set_batch / resize, not at the first predict: the reference allocates
 * l.output in make_*_layer, and callers copy `layer l = net.layers[n-1]` long before they predict. */
void y2_engine_host_output(network *net)
{
    y2_engine *e = y2_engine_of(net);
    layer *ol;
    size_t need;
    if (!e || !net->layers) return;
    e->out_layer = y2_out_layer(net);
    ol = &net->layers[e->out_layer];
    need = (size_t)net->batch * ol->outputs;
    if (need > e->h_out_cap || !e->h_out) {
        if (e->h_out_pinned) y2h_host_free(e->h_out); else free(e->h_out);
        e->h_out = NULL; e->h_out_pinned = 0;
        /* With a GPU in the machine the buffer is pinned memory of its own (hipHostMalloc, never a registered heap block:
         * profiles/r02_notes.md), so that network_predict's device-to-host copy lands in the buffer the caller reads --
         * one copy, as the reference's cuda_pull_array into l.output (network_kernels.cu:412-416).  Without one (cfg tools,
         * the CPU test suite) it is plain page-aligned memory; nothing can be predicted then anyway. */
        if (y2h_device_count() > 0 && y2h_host_alloc((void **)&e->h_out, (need ? need : 1) * sizeof(float)) == 0) e->h_out_pinned = 1;
        else if (posix_memalign((void **)&e->h_out, 4096, (need ? need : 1) * sizeof(float)) != 0) e->h_out = NULL;
        if (e->h_out) memset(e->h_out, 0, (need ? need : 1) * sizeof(float));
        e->h_out_cap = need;
    }
    ol->output = e->h_out;
}

static void free_plan(network *net)
{
    y2_engine *e = y2_engine_of(net);
    int i;
    if (!e) return;
    for (i = 0; i < net->n; ++i) {
        y2_ldev *d = ld_of(&net->layers[i]);
        if (!d) continue;
        y2h_free(d->out_alloc); d->out_alloc = NULL; d->out = NULL;
        y2h_free(d->d_region); d->d_region = NULL;
        y2h_free(d->d_flat); d->d_flat = NULL;
        y2h_free(d->d_halo); d->d_halo = NULL; d->halo_px = 0;
        y2h_free(d->d_bin); d->d_bin = NULL;
        d->placed_in = -1; d->alias_of = -1; d->copy_mask = 0;
        d->fused_pool = 0; d->fused_into = -1;
        d->out_half = 0;
        d->tile_bm = d->tile_bn = d->ksplit = 0;
    }
    drop_graphs(e);
    y2h_free(e->d_in_nchw); e->d_in_nchw = NULL;
    y2h_free(e->d_in_nhwc); e->d_in_nhwc = NULL;
    y2h_free(e->d_out_nchw); e->d_out_nchw = NULL;
    y2h_free(e->d_ws); e->d_ws = NULL; e->ws_bytes = 0;
    y2h_free(e->d_u8); e->d_u8 = NULL; e->u8_cap = 0;
    y2h_free(e->d_planes); e->d_planes = NULL; e->planes_cap = 0;
    y2h_free(e->d_rtmp); e->d_rtmp = NULL; e->rtmp_cap = 0;
    y2h_free(e->d_boxes); e->d_boxes = NULL;
    y2h_free(e->d_probs); e->d_probs = NULL;
    y2h_free(e->d_probs_nms); e->d_probs_nms = NULL;
    y2h_free(e->d_records); e->d_records = NULL;
    y2h_free(e->d_counts); e->d_counts = NULL;
    y2h_free(e->d_class_counts); e->d_class_counts = NULL;
    y2h_free(e->d_best); e->d_best = NULL;
    y2h_free(e->d_mean_ring); e->d_mean_ring = NULL; e->mean_els = 0; e->mean_index = 0;
    y2h_host_free(e->h_records); e->h_records = NULL;
    y2h_host_free(e->h_counts); e->h_counts = NULL;
    e->built = 0;
}

void y2_engine_invalidate(network *net)
{
    y2_engine *e = y2_engine_of(net);
    if (e) e->built = 0;
}

void y2_engine_destroy(network *net)
{
    y2_engine *e = y2_engine_of(net);
    int i;
    if (!e) return;
    if (e->stream || e->arena || e->built) y2h_set_device(e->device);
    free_plan(net);
    y2_feed_close(net);
    for (i = 0; i < net->n; ++i) {
        y2_ldev *d = ld_of(&net->layers[i]);
        if (!d) continue;
        y2h_free(d->d_anchors); y2h_free(d->d_tree_parent); y2h_free(d->d_tree_gsize); y2h_free(d->d_tree_goff); y2h_free(d->d_map);
        y2h_free(d->d_tree_order); y2h_free(d->d_tree_loff); y2h_free(d->d_tree_best);
        free(d);
        net->layers[i].dev = NULL;
    }
    y2h_free(e->arena);
    y2h_host_free(e->h_out_stage);
    if (e->h_out_pinned) y2h_host_free(e->h_out); else free(e->h_out);
    if (e->ev) { for (i = 0; i < e->n_ev; ++i) y2h_event_destroy(e->ev[i]); free(e->ev); }
    if (e->ev_det) y2h_event_destroy(e->ev_det);
    if (e->ev_out) y2h_event_destroy(e->ev_out);
    drop_graphs(e);
    if (e->det_stream) y2h_stream_destroy(e->det_stream);
    if (e->ev_fwd) y2h_event_destroy(e->ev_fwd);
    y2h_stream_destroy(e->stream);
    free(e);
    net->engine = NULL;
}

/* ------------------------------------------------------------------ */
/* plan                                                                */
/* ------------------------------------------------------------------ */
static int producer_can_place(const layer *l)
{
    return l->type == CONVOLUTIONAL || l->type == MAXPOOL || l->type == REORG;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

/* device code of a reference ACTIVATION (activations.h:7) */
static int act_code(ACTIVATION a)
{
    switch (a) {
    case LINEAR: return Y2H_ACT_LINEAR;
    case LEAKY: return Y2H_ACT_LEAKY;
    case LOGISTIC: return Y2H_ACT_LOGISTIC;
    case RELU: return Y2H_ACT_RELU;
    case RELIE: return Y2H_ACT_RELIE;
    case RAMP: return Y2H_ACT_RAMP;
    case TANH: return Y2H_ACT_TANH;
    case PLSE: return Y2H_ACT_PLSE;
    case ELU: return Y2H_ACT_ELU;
    case LOGGY: return Y2H_ACT_LOGGY;
    case STAIR: return Y2H_ACT_STAIR;
    case HARDTAN: return Y2H_ACT_HARDTAN;
    case LHTAN: return Y2H_ACT_LHTAN;
    }
    return -1;
}
/* the four activations of the target cfgs are applied in the producing kernel's epilogue; the others run as the
 * reference runs every activation -- a pass of their own over the stored output (activations.c:95) */
static int act_in_kernel(ACTIVATION a) { const int c = act_code(a); return c >= 0 && c <= Y2H_ACT_RELU; }
static int act_for_kernel(ACTIVATION a) { return act_in_kernel(a) ? act_code(a) : Y2H_ACT_LINEAR; }

static void conv_desc(const network *net, int i, y2h_conv *c, const float *x, int ldx)
{
    const layer *l = &net->layers[i];
    const y2_ldev *d = ld_of(l);
    const y2_engine *e = d->eng;
    memset(c, 0, sizeof *c);
    c->batch = l->batch; c->h = l->h; c->w = l->w; c->c = l->c; c->ldx = ldx;
    c->n = l->n; c->size = l->size; c->stride = l->stride; c->pad = l->pad;
    c->out_h = l->out_h; c->out_w = l->out_w; c->ldy = d->out_ld;
    c->batch_normalize = l->batch_normalize;
    c->activation = act_for_kernel(l->activation);
    c->x = x;
    c->x_halo = (i == 0 && e->in_halo && e->in_halo != 3) ? e->in_halo_px : 0;   /* in_halo 2: half [b][h+2][w+2][4] for the fp16 first-layer kernel;
                                                                                    3: that kernel reads the fp32 NCHW input itself */
    if (i > 0 && ld_of(&net->layers[i - 1])->d_halo) c->x_halo = ld_of(&net->layers[i - 1])->halo_px;
    if (l->xnor) c->x_halo = 0;
    c->fuse_maxpool2 = d->fused_pool;
    c->ws = e->d_ws;
    c->ws_bytes = e->ws_bytes;
    c->y = d->out;
    c->y_f16 = d->out_half;
    c->x_f16 = (i > 0) ? ld_of(&net->layers[i - 1])->out_half : (e->in_halo == 2);
    c->x_nchw = (i == 0 && e->in_halo == 3);
    c->tile_bm = d->tile_bm; c->tile_bn = d->tile_bn; c->ksplit = d->ksplit;
    if (e->arena) {
        c->w_packed = (const float *)(e->arena + d->off_w_packed);
        c->w_ref = d->has_w_ref ? (const float *)(e->arena + d->off_w_ref) : NULL;
        c->bias = (const float *)(e->arena + d->off_bias);
        if (c->x_f16 && i > 0) {
            c->alpha = (const float *)(e->arena + d->off_alpha);
            c->beta = (const float *)(e->arena + d->off_beta);
        }
        if (l->batch_normalize) {
            c->mean = (const float *)(e->arena + d->off_mean);
            c->scale = (const float *)(e->arena + d->off_scale);
            c->rinv = (const double *)(e->arena + d->off_rinv);
        }
    }
}

static void input_view(const network *net, int i, const float **x, int *ldx)
{
    const y2_engine *e = y2_engine_of(net);
    if (i == 0) { *x = (e->in_halo == 3) ? e->cur_input : e->d_in_nhwc; *ldx = (e->in_halo == 2) ? 4 : net->c; }
    else {
        const y2_ldev *p = ld_of(&net->layers[i - 1]);
        *x = p->out; *ldx = p->out_ld;
        if (p->d_halo && net->layers[i].type == CONVOLUTIONAL) { *x = p->d_halo; *ldx = net->layers[i - 1].out_c; }
    }
    /* an xnor convolution reads the binarized copy of its input */
    if (net->layers[i].type == CONVOLUTIONAL && net->layers[i].xnor && ld_of(&net->layers[i])->d_bin) {
        *x = ld_of(&net->layers[i])->d_bin; *ldx = net->layers[i].c;
    }
    /* a [connected] layer is run as a 1x1 convolution over a 1x1 image whose channels are the whole input vector */
    if (net->layers[i].type == CONNECTED) *ldx = net->layers[i].inputs;
}

/* the layer whose activations a layer reads, looking through the inference no-ops ([dropout], [cost]) */
static int producer_of(const network *net, int i)
{
    int p = i - 1;
    while (p > 0 && (net->layers[p].type == DROPOUT || net->layers[p].type == COST)) --p;
    return p;
}

/* 1: the layer's activations are a flat [batch][outputs] fp32 vector, not an NHWC image */
static int is_flat(const network *net, int i)
{
    switch (net->layers[i].type) {
    case REGION: case AVGPOOL: case SOFTMAX: case CONNECTED: case DETECTION: return 1;
    case DROPOUT: case COST: return i > 0 ? is_flat(net, i - 1) : 0;
    default: return 0;
    }
}

static int upload_small(void **dst, const void *src, size_t bytes, y2h_stream s)
{
    if (*dst) { y2h_free(*dst); *dst = NULL; }
    if (y2h_malloc(dst, bytes) != 0) return -1;
    if (y2h_memcpy_h2d(*dst, src, bytes, s) != 0) return -1;
    return y2h_stream_sync(s);
}

/* fp32 -> IEEE half, round to nearest even (what the device's v_cvt_f16_f32 does) */
static unsigned short f32_to_f16_rne(float f)
{
    unsigned int x, sign, mant;
    int exp;
    memcpy(&x, &f, sizeof x);
    sign = (x >> 16) & 0x8000u;
    exp = (int)((x >> 23) & 0xff) - 127 + 15;
    mant = x & 0x7fffffu;
    if (((x >> 23) & 0xff) == 0xff) return (unsigned short)(sign | 0x7c00u | (mant ? 0x200u : 0));   /* inf / nan */
    if (exp >= 31) return (unsigned short)(sign | 0x7c00u);                                          /* overflow */
    if (exp <= 0) {                                                                                  /* subnormal / zero */
        unsigned int shift, half, rem;
        if (exp < -10) return (unsigned short)sign;
        mant |= 0x800000u;
        shift = (unsigned int)(14 - exp);
        half = mant >> shift;
        rem = mant & ((1u << shift) - 1);
        if (rem > (1u << (shift - 1)) || (rem == (1u << (shift - 1)) && (half & 1))) ++half;
        return (unsigned short)(sign | half);
    }
    {
        unsigned int half = ((unsigned int)exp << 10) | (mant >> 13), rem = mant & 0x1fffu;
        if (rem > 0x1000u || (rem == 0x1000u && (half & 1))) ++half;      /* may carry into the exponent: correct */
        return (unsigned short)(sign | half);
    }
}

static int upload_weights(network *net)
{
    y2_engine *e = y2_engine_of(net);
    /* pinned staging: no pageable buffer of ours is ever handed to an asynchronous copy */
    unsigned char *host = NULL;
    if (y2h_host_alloc((void **)&host, e->arena_bytes ? e->arena_bytes : 16) != 0) host = NULL;
    int i;
    if (!host) { y2_fail("weight upload: no pinned host memory for %zu bytes: %s", e->arena_bytes, y2h_last_error()); return -1; }
    memset(host, 0, e->arena_bytes ? e->arena_bytes : 16);
    for (i = 0; i < net->n; ++i) {
        const layer *l = &net->layers[i];
        const y2_ldev *d = ld_of(l);
        float *wp, *b;
        int K, co, ci, kh, kw, f;
        const int w_half = (i > 0) && ld_of(&net->layers[i - 1])->out_half;   /* half input -> half weights */
        if (l->type == BATCHNORM) {
            double *r = (double *)(host + d->off_rinv);
            memcpy(host + d->off_mean, l->rolling_mean, l->c * sizeof(float));
            memcpy(host + d->off_scale, l->scales, l->c * sizeof(float));
            for (f = 0; f < l->c; ++f) r[f] = 1.0 / (sqrt((double)l->rolling_variance[f]) + (double).000001f);
            continue;
        }
        if (l->type == LOCAL) {
            /* reference: weights [location][filter][c][kh][kw], biases [filter][location] (local_layer.c:100,111-121);
             * kernel: weights [location][filter][kh][kw][c], biases [location][filter] */
            const int locations = l->out_h * l->out_w, kk = l->size * l->size;
            int loc, t;
            K = kk * l->c;
            wp = (float *)(host + d->off_w_packed);
            b = (float *)(host + d->off_bias);
            for (loc = 0; loc < locations; ++loc)
                for (co = 0; co < l->n; ++co) {
                    const float *src = l->weights + ((size_t)loc * l->n + co) * K;
                    float *dst = wp + ((size_t)loc * l->n + co) * K;
                    for (ci = 0; ci < l->c; ++ci)
                        for (t = 0; t < kk; ++t) dst[(size_t)t * l->c + ci] = src[(size_t)ci * kk + t];
                    b[(size_t)loc * l->n + co] = l->biases[(size_t)co * locations + loc];
                }
            continue;
        }
        if (l->type != CONVOLUTIONAL && l->type != CONNECTED) continue;
        K = l->size * l->size * l->c;
        wp = (float *)(host + d->off_w_packed);
        if (l->type == CONNECTED) {
            /* [outputs][inputs]: the reference flattens an image producer as [c][y][x], our activations are
             * [y][x][c], so input k = c*HW + p moves to p*C + c; a flat producer keeps its order */
            const layer *pl = i > 0 ? &net->layers[producer_of(net, i)] : NULL;
            const int hw = (pl && !is_flat(net, producer_of(net, i))) ? pl->out_h * pl->out_w : 1;
            const int C = K / (hw > 0 ? hw : 1);
            int pix;
            for (co = 0; co < l->n; ++co)
                for (ci = 0; ci < C; ++ci)
                    for (pix = 0; pix < hw; ++pix)
                        wp[(size_t)co * K + (size_t)pix * C + ci] = l->weights[(size_t)co * K + (size_t)ci * hw + pix];
        } else
        /* reference layout [n][c][kh][kw] (im2col.c:24-27) -> kernel layout [n][kh][kw][c] */
        for (co = 0; co < l->n; ++co) {
            float bmean = 0;
            if (l->xnor) {          /* binarize_weights, convolutional_layer.c:37-50: +-mean|w| per filter, sequential fp32 sum */
                int q;
                for (q = 0; q < K; ++q) bmean += fabs(l->weights[(size_t)co * K + q]);
                bmean = bmean / K;
            }
            for (ci = 0; ci < l->c; ++ci)
                for (kh = 0; kh < l->size; ++kh)
                    for (kw = 0; kw < l->size; ++kw) {
                        const size_t dst = (size_t)co * K + (size_t)(kh * l->size + kw) * l->c + ci;
                        float v = l->weights[(((size_t)co * l->c + ci) * l->size + kh) * l->size + kw];
                        if (l->xnor) v = (v > 0) ? bmean : -bmean;
                        if (w_half) ((unsigned short *)wp)[dst] = f32_to_f16_rne(v);
                        else wp[dst] = v;
                    }
        }
        if (w_half) {
            /* folded batch-norm for the fp16 kernels: y = act(acc*alpha + beta), constants evaluated in double */
            float *al = (float *)(host + d->off_alpha), *be = (float *)(host + d->off_beta);
            for (f = 0; f < l->n; ++f) {
                double a = 1.0, bb = l->biases[f];
                if (l->batch_normalize) {
                    a = (double)l->scales[f] / (sqrt((double)l->rolling_variance[f]) + (double).000001f);
                    bb = (double)l->biases[f] - (double)l->rolling_mean[f] * a;
                }
                al[f] = (float)a; be[f] = (float)bb;
            }
        }
        if (d->has_w_ref) memcpy(host + d->off_w_ref, l->weights, (size_t)l->n * K * sizeof(float));
        if (d->has_w_ref && l->type == CONVOLUTIONAL && l->xnor) {
            float *wr = (float *)(host + d->off_w_ref);
            for (co = 0; co < l->n; ++co) {
                float bmean = 0;
                int q;
                for (q = 0; q < K; ++q) bmean += fabs(l->weights[(size_t)co * K + q]);
                bmean = bmean / K;
                for (q = 0; q < K; ++q) wr[(size_t)co * K + q] = (l->weights[(size_t)co * K + q] > 0) ? bmean : -bmean;
            }
        }
        b = (float *)(host + d->off_bias);
        memcpy(b, l->biases, l->n * sizeof(float));
        if (l->batch_normalize) {
            double *r = (double *)(host + d->off_rinv);
            memcpy(host + d->off_mean, l->rolling_mean, l->n * sizeof(float));
            memcpy(host + d->off_scale, l->scales, l->n * sizeof(float));
            /* blas.c:122: x / (sqrt(variance) + .000001f), the divisor evaluated in double */
            for (f = 0; f < l->n; ++f) r[f] = 1.0 / (sqrt((double)l->rolling_variance[f]) + (double).000001f);
        }
    }
    if (y2h_memcpy_h2d(e->arena, host, e->arena_bytes, e->stream) != 0 || y2h_stream_sync(e->stream) != 0) {
        y2h_host_free(host);
        y2_fail("weight upload failed: %s", y2h_last_error());
        return -1;
    }
    y2h_host_free(host);
    e->weights_dirty = 0;
    return 0;
}

/* split-K scratch: the largest request of any conv layer under the current tile choices */
static int size_workspace(network *net)
{
    y2_engine *e = y2_engine_of(net);
    size_t need = 0;
    int i;
    for (i = 0; i < net->n; ++i) {
        y2h_conv c;
        const float *x; int ldx;
        size_t b;
        if ((net->layers[i].type != CONVOLUTIONAL && net->layers[i].type != CONNECTED) || e->strict) continue;
        input_view(net, i, &x, &ldx);
        conv_desc(net, i, &c, x, ldx);
        c.w_packed = (const float *)(uintptr_t)256;
        b = y2h_conv_workspace_bytes(&c);
        if (b > need) need = b;
    }
    if (need > e->ws_bytes) {
        y2h_free(e->d_ws); e->d_ws = NULL; e->ws_bytes = 0;
        HIPCALL(y2h_malloc((void **)&e->d_ws, need));
        e->ws_bytes = need;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* tile autotuning (y2_set_autotune)                                   */
/* ------------------------------------------------------------------ */
/* Measured choices are remembered per layer shape for the life of the process, so that re-plans (set_batch_network,
 * resize_network back and forth, several networks of one family) do not measure again. */
typedef struct { int batch, h, w, c, n, size, stride, pool, bm, bn, ks; } tune_entry;
static tune_entry g_tuned[256];
static int g_ntuned = 0;
static pthread_mutex_t g_tuned_mu = PTHREAD_MUTEX_INITIALIZER;    /* networks / Detectors may be built from several threads */

static int enqueue_forward(network *net, const float *d_input_nchw);

/* Candidates are timed INSIDE whole forward passes (per-layer HIP events, as y2_layer_times_ms reads them): timed in
 * isolation, back to back, a layer finds its own weights in the Infinity Cache and small tiles look better than they are
 * in the real sequence, where the 204 MB of yolo.cfg weights stream from HBM once per forward. */
static int autotune_layers(network *net)
{
    y2_engine *e = y2_engine_of(net);
    const int keep_timing = e->timing;
    int i, k, a;
    size_t need = 0;
    for (i = 0; i < net->n; ++i) {          /* scratch for the largest K-split any candidate may ask for */
        const layer *l = &net->layers[i];
        y2_ldev *d = ld_of(l);
        y2h_conv c;
        const float *x; int ldx, n, bm[64], bn[64], ks[64];
        if (l->type != CONVOLUTIONAL || !d->uses_mfma || l->xnor) continue;
        input_view(net, i, &x, &ldx);
        conv_desc(net, i, &c, x, ldx);
        n = y2h_conv_candidates(&c, bm, bn, ks, 64);
        for (a = 0; a < n; ++a) {
            size_t b = (size_t)ks[a] * l->batch * l->out_h * l->out_w * l->out_c * sizeof(float);
            if (ks[a] > 1 && b > need) need = b;
        }
    }
    if (need > e->ws_bytes) {
        y2h_free(e->d_ws); e->d_ws = NULL; e->ws_bytes = 0;
        HIPCALL(y2h_malloc((void **)&e->d_ws, need));
        e->ws_bytes = need;
    }
    if (y2h_memset(e->d_in_nchw, 0, e->in_floats * sizeof(float), e->stream) != 0) { y2_fail("autotune: %s", y2h_last_error()); return -1; }
    e->timing = 1;
    for (i = 0; i < net->n; ++i) {
        layer *l = &net->layers[i];
        y2_ldev *d = ld_of(l);
        y2h_conv c;
        const float *x; int ldx, n, hit = 0, bm[64], bn[64], ks[64], best = 0;
        float best_ms = 0.f;
        if (l->type != CONVOLUTIONAL || !d->uses_mfma || l->xnor) continue;
        input_view(net, i, &x, &ldx);
        conv_desc(net, i, &c, x, ldx);
        pthread_mutex_lock(&g_tuned_mu);
        for (k = 0; k < g_ntuned && !hit; ++k) {
            const tune_entry *t = &g_tuned[k];
            if (t->batch == c.batch && t->h == c.h && t->w == c.w && t->c == c.c && t->n == c.n && t->size == c.size &&
                t->stride == c.stride && t->pool == c.fuse_maxpool2) { d->tile_bm = t->bm; d->tile_bn = t->bn; d->ksplit = t->ks; hit = 1; }
        }
        pthread_mutex_unlock(&g_tuned_mu);
        /* big grids (hundreds of tiles per CU round) are where the cost model is reliable and a measurement costly */
        if (!hit && 2.0 * l->batch * l->out_h * l->out_w * (double)l->n * l->size * l->size * l->c > 40e9) continue;
        if (!hit) {
            c.tile_bm = c.tile_bn = c.ksplit = 0;
            n = y2h_conv_candidates(&c, bm, bn, ks, 64);
            if (n < 0) { e->timing = keep_timing; y2_fail("autotune of layer %d failed (%d): %s", i, n, y2h_last_error()); return -1; }
            if (n == 0) continue;
            for (a = 0; a < n; ++a) {
                float ms = 0.f, m2 = 0.f;
                int rep;
                d->tile_bm = bm[a]; d->tile_bn = bn[a]; d->ksplit = ks[a];
                for (rep = 0; rep < 2; ++rep) {            /* the first pass also sets the kernel's LDS attribute */
                    if (enqueue_forward(net, e->d_in_nchw) != 0) { e->timing = keep_timing; return -1; }
                    if (y2h_event_elapsed_ms(e->ev[i], e->ev[i + 1], &m2) != 0) { e->timing = keep_timing; y2_fail("autotune: %s", y2h_last_error()); return -1; }
                    ms = (rep == 0 || m2 < ms) ? m2 : ms;
                }
                if (getenv("Y2_AUTOTUNE_LOG"))
                    fprintf(stderr, "autotune layer %2d %3dx%-3d c%-4d n%-5d k%d%s: %3dx%-3d ks%-2d %.4f ms%s\n", i, l->h, l->w, l->c, l->n, l->size,
                            c.fuse_maxpool2 ? "+pool" : "", bm[a], bn[a], ks[a], ms, a == 0 ? "  (model)" : "");
                if (a == 0) ms *= 0.98f;                     /* the model's choice stays unless another wins by 2 % */
                if (a == 0 || ms < best_ms) { best_ms = ms; best = a; }
            }
            d->tile_bm = bm[best]; d->tile_bn = bn[best]; d->ksplit = ks[best];
            pthread_mutex_lock(&g_tuned_mu);
            if (g_ntuned < (int)(sizeof g_tuned / sizeof g_tuned[0])) {
                tune_entry *t = &g_tuned[g_ntuned++];
                t->batch = c.batch; t->h = c.h; t->w = c.w; t->c = c.c; t->n = c.n; t->size = c.size; t->stride = c.stride;
                t->pool = c.fuse_maxpool2; t->bm = d->tile_bm; t->bn = d->tile_bn; t->ks = d->ksplit;
            }
            pthread_mutex_unlock(&g_tuned_mu);
        }
        conv_desc(net, i, &c, x, ldx);
        d->kernel = y2h_conv_variant(&c, 0);
        if (d->fused_pool) { snprintf(d->kname, sizeof d->kname, "%s+maxpool2", d->kernel); d->kernel = d->kname; }
    }
    e->timing = keep_timing;
    HIPCALL(y2h_stream_sync(e->stream));
    return 0;
}

/* A replica (y2_weights_arena on a rank that never loads weights) is planned while its arena is still uninitialised HBM:
 * timing candidates on garbage / NaN data would let every rank keep a different K-split, and ranks that are supposed to be
 * bit-identical replicas would differ in the last bits.  Such a build keeps the cost model's choices; the measurement runs
 * at the first build AFTER the arena became resident (y2_weights_resident drops the plan when autotuning is on). */
static int autotune_allowed(const y2_engine *e) { return e->autotune && !e->strict && !e->arena_pending; }

int y2_engine_build(network *net)
{
    y2_engine *e = y2_engine_of(net);
    int i, k;
    size_t off;
    if (!e) { y2_fail("network has no engine (was it built by parse_network_cfg?)"); return -1; }
    if (net->gpu_index < 0) {
        y2_fail("gpu_index %d: this library has no CPU compute path; select a GPU (>= 0)", net->gpu_index);
        return -1;
    }
    if (y2h_device_count() <= 0) { y2_fail("no HIP device visible: the MI355X engine cannot run"); return -1; }
    e->device = net->gpu_index;
    HIPCALL(y2h_set_device(e->device));
    if (!e->stream) HIPCALL(y2h_stream_create(&e->stream));
    free_plan(net);

    /* every layer follows the network batch (set_batch_network only rewrites the field) */
    for (i = 0; i < net->n; ++i) net->layers[i].batch = net->batch;

    /* pass 1: let the sources of concatenating routes write into the route buffer */
    for (i = 0; i < net->n; ++i) {
        layer *l = &net->layers[i];
        y2_ldev *d = ld_of(l);
        if (l->type == ROUTE && l->n == 1) d->alias_of = l->input_layers[0];
        if (l->type == COST) d->alias_of = i - 1;
        if (l->type != ROUTE || l->n < 2) continue;
        if (l->n > 32) { y2_fail("route layer %d has %d inputs (max 32)", i, l->n); return -1; }
        if (!l->out_c) { y2_fail("route layer %d concatenates layers of different spatial size", i); return -1; }
        for (k = 0; k < l->n; ++k) {
            layer *src = &net->layers[l->input_layers[k]];
            y2_ldev *sd = ld_of(src);
            int dup = 0, m;
            for (m = 0; m < k; ++m) if (l->input_layers[m] == l->input_layers[k]) dup = 1;
            if (!dup && producer_can_place(src) && sd->placed_in < 0 && l->input_layers[k] != e->out_layer)
                sd->placed_in = i;
            else
                d->copy_mask |= 1u << k;
        }
    }
    /* pass 1b: conv -> 2x2/2 maxpool pairs whose full-resolution activation nobody else reads are fused:
     * the conv kernel pools in its epilogue and writes straight into the maxpool layer's buffer */
    if (e->fusion && !e->strict) {
        for (i = 0; i + 1 < net->n; ++i) {
            layer *l = &net->layers[i], *m = &net->layers[i + 1];
            int used = 0, j;
            if (l->type != CONVOLUTIONAL || m->type != MAXPOOL) continue;
            if (m->size != 2 || m->stride != 2 || m->pad != 0 || (l->out_h & 1) || (l->out_w & 1)) continue;
            if (l->stride != 1 || l->pad != l->size / 2 || !(l->size == 1 || l->size == 3)) continue;
            if (!act_in_kernel(l->activation)) continue;       /* the separate activation pass must see every pixel */
            if (!((l->c % 16 == 0) || (i == 0 && l->c == 3 && l->size == 3 && l->n <= 64))) continue;
            if (i == e->out_layer || ld_of(l)->placed_in >= 0) continue;
            for (j = 0; j < net->n; ++j) {
                if (net->layers[j].type == ROUTE)
                    for (k = 0; k < net->layers[j].n; ++k) if (net->layers[j].input_layers[k] == i) used = 1;
                if (net->layers[j].type == SHORTCUT && net->layers[j].index == i) used = 1;
            }
            if (used) continue;
            ld_of(l)->fused_pool = 1;
            ld_of(m)->fused_into = i;
        }
    }
    /* pass 1c: fp16 storage (y2_set_half): image-like activations are half, heads stay fp32 */
    if (e->half && !e->strict) {
        for (i = 0; i < net->n; ++i) {
            layer *l = &net->layers[i];
            y2_ldev *d = ld_of(l), *pd = i > 0 ? ld_of(&net->layers[i - 1]) : NULL;
            switch (l->type) {
            case CONVOLUTIONAL:
                if (l->xnor) { y2_fail("fp16 mode: layer %d: xnor convolutions have no half-precision form", i); return -1; }
                if (!act_in_kernel(l->activation)) { y2_fail("fp16 mode: layer %d: activation %d has no half-precision form", i, (int)l->activation); return -1; }
                /* the conv feeding a region head writes fp32: the head's logistic/softmax/exp run in fp32 */
                d->out_half = !(i + 1 < net->n && net->layers[i + 1].type == REGION);
                break;
            case MAXPOOL: case REORG:
                if (!pd || !pd->out_half) { y2_fail("fp16 mode: layer %d (%s) needs a half-precision producer", i, get_layer_string(l->type)); return -1; }
                d->out_half = 1;
                break;
            case ROUTE:
                if (l->n == 1) d->out_half = ld_of(&net->layers[l->input_layers[0]])->out_half;
                else {
                    for (k = 0; k < l->n; ++k)
                        if (!ld_of(&net->layers[l->input_layers[k]])->out_half) { y2_fail("fp16 mode: route layer %d mixes fp32 and half inputs", i); return -1; }
                    d->out_half = 1;
                }
                break;
            case REGION: case SOFTMAX:
                if (pd && pd->out_half) { y2_fail("fp16 mode: layer %d (%s) needs an fp32 producer (a convolutional or avgpool layer)", i, get_layer_string(l->type)); return -1; }
                break;
            case COST: d->out_half = pd ? pd->out_half : 0; break;
            case SHORTCUT: case CONNECTED: case DETECTION: case DROPOUT: case CROP: case LOCAL: case BATCHNORM:
                y2_fail("fp16 mode: layer %d (%s) has no half-precision kernel", i, get_layer_string(l->type)); return -1;
            default: break;
            }
        }
        if (net->n > 0 && net->layers[0].type != CONVOLUTIONAL) { y2_fail("fp16 mode: the first layer must be convolutional"); return -1; }
    }
    /* pass 2: allocate.  Routes first (their sources point into them). */
    for (i = 0; i < net->n; ++i) {
        layer *l = &net->layers[i];
        y2_ldev *d = ld_of(l);
        if (l->type == ROUTE && l->n >= 2) {
            d->out_floats = (size_t)l->batch * l->out_h * l->out_w * l->out_c;
            HIPCALL(y2h_malloc((void **)&d->out_alloc, d->out_floats * (d->out_half ? 2 : 4)));
            d->out = d->out_alloc;
            d->out_ld = l->out_c;
        }
    }
    for (i = 0; i < net->n; ++i) {
        layer *l = &net->layers[i];
        y2_ldev *d = ld_of(l);
        switch (l->type) {
        case CONVOLUTIONAL: case MAXPOOL: case REORG:
            d->kernel = l->type == MAXPOOL ? "maxpool_nhwc" : (l->type == REORG ? "reorg_nhwc" : "conv");
            if (d->fused_pool) break;            /* writes into the maxpool layer's buffer (set below) */
            if (d->placed_in >= 0) {
                layer *r = &net->layers[d->placed_in];
                y2_ldev *rd = ld_of(r);
                int choff = 0;
                for (k = 0; k < r->n && r->input_layers[k] != i; ++k) choff += net->layers[r->input_layers[k]].out_c;
                d->out = d->out_half ? (float *)((unsigned short *)rd->out + choff) : rd->out + choff;
                d->out_ld = r->out_c;
            } else {
                d->out_floats = (size_t)l->batch * l->out_h * l->out_w * l->out_c;
                HIPCALL(y2h_malloc((void **)&d->out_alloc, d->out_floats * (d->out_half ? 2 : 4)));
                d->out = d->out_alloc;
                d->out_ld = l->out_c;
            }
            break;
        case SHORTCUT: case CROP: case LOCAL: case BATCHNORM:
            d->out_floats = (size_t)l->batch * l->out_h * l->out_w * l->out_c;
            HIPCALL(y2h_malloc((void **)&d->out_alloc, d->out_floats * sizeof(float)));
            d->out = d->out_alloc;
            d->out_ld = l->out_c;
            d->kernel = l->type == SHORTCUT ? "shortcut" : l->type == CROP ? "crop" : l->type == LOCAL ? (e->strict ? "local_ref" : "local") : "batchnorm";
            break;
        case ROUTE:
            d->kernel = "route(zero-copy)";
            if (l->n == 1) { y2_ldev *sd = ld_of(&net->layers[d->alias_of]); d->out = sd->out; d->out_ld = sd->out_ld; }
            else if (d->copy_mask) d->kernel = "route(copy_channels)";
            break;
        case COST: {
            y2_ldev *sd = ld_of(&net->layers[i - 1]);
            d->out = sd->out; d->out_ld = sd->out_ld; d->kernel = "none";
        } break;
        case REGION:
            HIPCALL(y2h_malloc((void **)&d->d_region, (size_t)l->batch * l->outputs * sizeof(float)));
            d->out = d->d_region; d->out_ld = l->outputs / (l->h * l->w);
            d->kernel = l->softmax_tree ? "region+tree_softmax" : "region";
            break;
        case AVGPOOL: case SOFTMAX:
            HIPCALL(y2h_malloc((void **)&d->d_flat, (size_t)l->batch * l->outputs * sizeof(float)));
            d->out = d->d_flat; d->out_ld = l->outputs;
            d->kernel = l->type == AVGPOOL ? "avgpool" : "softmax_rows";
            break;
        case CONNECTED: case DETECTION: {
            /* YOLOv1 family: flat fp32 vectors.  The producer of a dense layer must be contiguous (an image
             * producer is read as [y][x][c] with re-ordered weights, see upload_weights) */
            const int pi = producer_of(net, i);
            const y2_ldev *pd = i > 0 ? ld_of(&net->layers[pi]) : NULL;
            if (i == 0) { y2_fail("layer %d (%s) cannot be the first layer", i, get_layer_string(l->type)); return -1; }
            if (!is_flat(net, pi) && (pd->out_ld != net->layers[pi].out_c || pd->fused_pool)) {
                y2_fail("layer %d (%s): its input (layer %d) is not stored contiguously", i, get_layer_string(l->type), pi);
                return -1;
            }
            if (l->type == DETECTION && !is_flat(net, pi)) { y2_fail("detection layer %d must follow a flat layer ([connected])", i); return -1; }
            HIPCALL(y2h_malloc((void **)&d->d_flat, (size_t)l->batch * l->outputs * sizeof(float)));
            d->out = d->d_flat; d->out_ld = l->outputs;
            d->kernel = l->type == DETECTION ? (l->softmax ? "detection(copy+softmax)" : "detection(copy)") : "connected";
        } break;
        case DROPOUT: {
            y2_ldev *sd = i > 0 ? ld_of(&net->layers[i - 1]) : NULL;
            if (!sd) { y2_fail("dropout layer %d has no input layer", i); return -1; }
            d->alias_of = i - 1;
            d->out = sd->out; d->out_ld = sd->out_ld; d->kernel = "none (inference)";
        } break;
        default:
            y2_fail("layer %d: type %d has no device implementation", i, (int)l->type);
            return -1;
        }
    }
    for (i = 0; i + 1 < net->n; ++i) {
        y2_ldev *d = ld_of(&net->layers[i]);
        if (d->fused_pool) {
            d->out = ld_of(&net->layers[i + 1])->out;
            d->out_ld = ld_of(&net->layers[i + 1])->out_ld;
            ld_of(&net->layers[i + 1])->kernel = "(fused into the conv before)";
        }
    }
    /* a [crop] in front of a few-channel convolution (vgg-16.cfg, strided.cfg, yolov1/yolo-small.cfg) also writes its
     * window with the zero border the first-layer / stem kernels want */
    if (!e->strict && !e->half) {
        for (i = 0; i + 1 < net->n; ++i) {
            const layer *l = &net->layers[i], *nl = &net->layers[i + 1];
            y2_ldev *d = ld_of(l);
            y2h_conv c0;
            int px;
            if (l->type != CROP || nl->type != CONVOLUTIONAL || nl->c > 4 || ld_of(nl)->fused_pool || nl->xnor) continue;
            memset(&c0, 0, sizeof c0);
            c0.batch = nl->batch; c0.h = nl->h; c0.w = nl->w; c0.c = nl->c; c0.ldx = nl->c; c0.n = nl->n;
            c0.size = nl->size; c0.stride = nl->stride; c0.pad = nl->pad; c0.out_h = nl->out_h; c0.out_w = nl->out_w;
            c0.w_packed = (const float *)(uintptr_t)256;
            px = y2h_conv_first_layer_ok(&c0) ? 1 : y2h_conv_stem_halo(&c0);
            if (px <= 0) continue;
            {
                const size_t fl = (size_t)l->batch * (l->out_h + 2 * px) * (l->out_w + 2 * px) * l->out_c;
                HIPCALL(y2h_malloc((void **)&d->d_halo, fl * sizeof(float)));
                HIPCALL(y2h_memset(d->d_halo, 0, fl * sizeof(float), e->stream));
                d->halo_px = px;
            }
        }
    }
    /* xnor=1 convolutions (convolutional_layer.c:443-447) read a +-1 copy of their input */
    for (i = 0; i < net->n; ++i) {
        layer *l = &net->layers[i];
        if (l->type == CONVOLUTIONAL && l->xnor)
            HIPCALL(y2h_malloc((void **)&ld_of(l)->d_bin, (size_t)l->batch * l->h * l->w * l->c * sizeof(float)));
    }
    /* io */
    e->in_floats = (size_t)net->batch * net->inputs;
    e->in_halo = 0;
    if (!e->strict && net->n > 0 && net->layers[0].type == CONVOLUTIONAL && !net->layers[0].xnor) {
        /* a 3-channel 3x3 first layer reads its input with a one-pixel zero halo (no tap bounds tests) */
        const layer *l0 = &net->layers[0];
        y2h_conv c0;
        memset(&c0, 0, sizeof c0);
        c0.batch = l0->batch; c0.h = l0->h; c0.w = l0->w; c0.c = l0->c; c0.ldx = net->c; c0.n = l0->n;
        c0.size = l0->size; c0.stride = l0->stride; c0.pad = l0->pad; c0.out_h = l0->out_h; c0.out_w = l0->out_w;
        c0.w_packed = (const float *)(uintptr_t)256;
        e->in_halo = y2h_conv_first_layer_ok(&c0);
        e->in_halo_px = 1;
        if (!e->in_halo && !(e->half && ld_of(l0)->out_half)) {
            /* other few-channel stems (7x7/2, 11x11/4, ...): the stem kernel reads a halo as wide as the padding */
            const int px = y2h_conv_stem_halo(&c0);
            if (px > 0) { e->in_halo = 1; e->in_halo_px = px; }
        }
        /* fp16 mode: the first layer reads a half [b][h+2][w+2][4] copy of the input on the fp16 matrix cores */
        if (e->half && ld_of(l0)->out_half && net->c <= 4 && y2h_conv_first_layer_f16_ok(&c0)) e->in_halo = 2;
        /* ... or, where the shape allows, reads the fp32 planes of the network input directly: no transform kernel */
        if (e->in_halo == 2 || (e->in_halo == 1 && e->in_halo_px == 1 && y2h_conv_first_layer_ok(&c0))) {
            c0.fuse_maxpool2 = ld_of(l0)->fused_pool;
            c0.y_f16 = ld_of(l0)->out_half;
            c0.x = (const float *)(uintptr_t)256;
            if (y2h_conv_first_layer_nchw_ok(&c0)) e->in_halo = 3;
        }
    }
    HIPCALL(y2h_malloc((void **)&e->d_in_nchw, e->in_floats * sizeof(float)));
    {
        size_t nhwc = e->in_halo ? (size_t)net->batch * (net->h + 2 * e->in_halo_px) * (net->w + 2 * e->in_halo_px) * net->c : e->in_floats;
        if (e->in_halo == 2) nhwc = (size_t)net->batch * (net->h + 2) * (net->w + 2) * 2;   /* 4 halves = 2 floats per pixel */
        if (e->in_halo == 3) nhwc = 64;                                                      /* not used */
        HIPCALL(y2h_malloc((void **)&e->d_in_nhwc, nhwc * sizeof(float)));
        HIPCALL(y2h_memset(e->d_in_nhwc, 0, nhwc * sizeof(float), e->stream));     /* the halo stays zero */
    }
    {
        layer *ol = &net->layers[e->out_layer];
        e->out_floats = (size_t)net->batch * ol->outputs;
        y2_engine_host_output(net);
        if (!e->h_out) { y2_fail("out of host memory for the network output"); return -1; }
        if (e->out_floats > e->h_out_stage_cap) {
            y2h_host_free(e->h_out_stage); e->h_out_stage = NULL; e->h_out_stage_cap = 0;
            HIPCALL(y2h_host_alloc((void **)&e->h_out_stage, e->out_floats * sizeof(float)));
            e->h_out_stage_cap = e->out_floats;
        }
        HIPCALL(y2h_malloc((void **)&e->d_out_nchw, e->out_floats * sizeof(float)));
        if (ol->type == REGION || ol->type == DETECTION) {
            e->det_total = ol->w * ol->h * ol->n;         /* a [detection] layer has w = h = side */
            e->det_classes = ol->classes;
            e->det_batch = net->batch;
            e->det_cap = e->det_total;
            HIPCALL(y2h_malloc((void **)&e->d_boxes, (size_t)net->batch * e->det_total * 4 * sizeof(float)));
            HIPCALL(y2h_malloc((void **)&e->d_probs, (size_t)net->batch * e->det_total * ol->classes * sizeof(float)));
            HIPCALL(y2h_malloc((void **)&e->d_probs_nms, (size_t)net->batch * e->det_total * ol->classes * sizeof(float)));
            HIPCALL(y2h_malloc((void **)&e->d_records, (size_t)net->batch * e->det_cap * 6 * sizeof(float)));
            HIPCALL(y2h_malloc((void **)&e->d_counts, (size_t)net->batch * sizeof(int)));
            HIPCALL(y2h_malloc((void **)&e->d_class_counts, (size_t)net->batch * ol->classes * sizeof(int)));
            e->class_counts_zeroed = 0;
            HIPCALL(y2h_malloc((void **)&e->d_best, (size_t)2 * net->batch * e->det_total * sizeof(float)));
            HIPCALL(y2h_host_alloc((void **)&e->h_records, (size_t)net->batch * e->det_cap * 6 * sizeof(float)));
            HIPCALL(y2h_host_alloc((void **)&e->h_counts, (size_t)net->batch * sizeof(int)));
        }
    }
    /* region constants */
    for (i = 0; i < net->n; ++i) {
        layer *l = &net->layers[i];
        y2_ldev *d = ld_of(l);
        if (l->type != REGION) continue;
        if (upload_small((void **)&d->d_anchors, l->biases, 2 * l->n * sizeof(float), e->stream)) { y2_fail("anchor upload: %s", y2h_last_error()); return -1; }
        if (l->softmax_tree) {
            tree *t = l->softmax_tree;
            if (upload_small((void **)&d->d_tree_parent, t->parent, t->n * sizeof(int), e->stream) ||
                upload_small((void **)&d->d_tree_gsize, t->group_size, t->groups * sizeof(int), e->stream) ||
                upload_small((void **)&d->d_tree_goff, t->group_offset, t->groups * sizeof(int), e->stream)) {
                y2_fail("tree upload: %s", y2h_last_error());
                return -1;
            }
            {   /* depth levels for the level-parallel hierarchy walk; only valid when parents come first */
                int *depth = calloc(t->n, sizeof(int)), *order = calloc(t->n, sizeof(int)), *loff, j, ok = 1, maxd = 0, lv;
                for (j = 0; j < t->n && ok; ++j) {
                    int par = t->parent[j];
                    if (par >= j) ok = 0;
                    else depth[j] = par < 0 ? 0 : depth[par] + 1;
                    if (ok && depth[j] > maxd) maxd = depth[j];
                }
                d->tree_levels = 0;
                if (ok) {
                    int pos = 0;
                    loff = calloc(maxd + 2, sizeof(int));
                    for (lv = 0; lv <= maxd; ++lv) {
                        loff[lv] = pos;
                        for (j = 0; j < t->n; ++j) if (depth[j] == lv) order[pos++] = j;
                    }
                    loff[maxd + 1] = pos;
                    if (upload_small((void **)&d->d_tree_order, order, t->n * sizeof(int), e->stream) ||
                        upload_small((void **)&d->d_tree_loff, loff, (maxd + 2) * sizeof(int), e->stream)) {
                        free(depth); free(order); free(loff);
                        y2_fail("tree upload: %s", y2h_last_error());
                        return -1;
                    }
                    d->tree_levels = maxd + 1;
                    free(loff);
                    /* detect mode's (score, class) per box as a by-product of the region layer (y2h_region_forward_tree) */
                    y2h_free(d->d_tree_best); d->d_tree_best = NULL;
                    if (!l->map && l->coords == 4 && y2h_region_tree_best_ok(l->classes, d->tree_levels) &&
                        y2h_malloc((void **)&d->d_tree_best, (size_t)2 * l->batch * l->h * l->w * l->n * sizeof(float))) {
                        free(depth); free(order);
                        y2_fail("tree scratch: %s", y2h_last_error());
                        return -1;
                    }
                }
                free(depth); free(order);
            }
        }
        if (l->map && upload_small((void **)&d->d_map, l->map, 200 * sizeof(int), e->stream)) { y2_fail("map upload: %s", y2h_last_error()); return -1; }
    }
    /* weight arena: decide per conv whether it runs on the matrix cores, then lay the arena out */
    off = 0;
    for (i = 0; i < net->n; ++i) {
        layer *l = &net->layers[i];
        y2_ldev *d = ld_of(l);
        y2h_conv c;
        const float *x; int ldx;
        size_t wbytes;
        int w_half;
        if (l->type == BATCHNORM) {
            d->off_mean = off; off = align_up(off + l->c * sizeof(float), 64);
            d->off_scale = off; off = align_up(off + l->c * sizeof(float), 64);
            d->off_rinv = off; off = align_up(off + l->c * sizeof(double), 64);
            continue;
        }
        if (l->type == LOCAL) {
            d->off_w_packed = off; off = align_up(off + (size_t)l->out_h * l->out_w * l->n * l->size * l->size * l->c * sizeof(float), 256);
            d->off_bias = off; off = align_up(off + (size_t)l->outputs * sizeof(float), 64);
            continue;
        }
        if (l->type != CONVOLUTIONAL && l->type != CONNECTED) continue;
        if (act_code(l->activation) < 0) { y2_fail("layer %d: unknown activation %d", i, (int)l->activation); return -1; }
        wbytes = (size_t)l->n * l->size * l->size * l->c * sizeof(float);
        w_half = (i > 0) && ld_of(&net->layers[i - 1])->out_half;
        input_view(net, i, &x, &ldx);
        conv_desc(net, i, &c, x, ldx);
        c.w_packed = (const float *)(uintptr_t)256;       /* alignment stand-in for the query */
        d->uses_mfma = !e->strict && y2h_conv_uses_mfma(&c);
        if (d->fused_pool && !d->uses_mfma) {
            /* the shape test above was optimistic (e.g. misaligned input view): give the conv its own buffer back */
            y2_ldev *md = ld_of(&net->layers[i + 1]);
            d->fused_pool = 0; md->fused_into = -1; md->kernel = "maxpool_nhwc";
            d->out_floats = (size_t)l->batch * l->out_h * l->out_w * l->out_c;
            HIPCALL(y2h_malloc((void **)&d->out_alloc, d->out_floats * (d->out_half ? 2 : 4)));
            d->out = d->out_alloc; d->out_ld = l->out_c;
            conv_desc(net, i, &c, x, ldx);
            c.w_packed = (const float *)(uintptr_t)256;
        }
        d->has_w_ref = !d->uses_mfma;
        d->off_w_packed = off; off = align_up(off + (w_half ? wbytes / 2 : wbytes), 256);
        if (d->has_w_ref) { d->off_w_ref = off; off = align_up(off + wbytes, 256); }
        d->off_bias = off; off = align_up(off + l->n * sizeof(float), 64);
        if (w_half) {
            d->off_alpha = off; off = align_up(off + l->n * sizeof(float), 64);
            d->off_beta = off; off = align_up(off + l->n * sizeof(float), 64);
        }
        if (l->batch_normalize) {
            d->off_mean = off; off = align_up(off + l->n * sizeof(float), 64);
            d->off_scale = off; off = align_up(off + l->n * sizeof(float), 64);
            d->off_rinv = off; off = align_up(off + l->n * sizeof(double), 64);
        }
        d->kernel = y2h_conv_variant(&c, e->strict);
        if (l->type == CONNECTED && !d->uses_mfma) d->kernel = "connected_ref";
        if (d->fused_pool) { snprintf(d->kname, sizeof d->kname, "%s+maxpool2", d->kernel); d->kernel = d->kname; }
    }
    if (size_workspace(net) != 0) return -1;
    {   /* The packed arena is only valid for the layout it was filled for: a re-plan may move a layer between the
         * matrix-core and the reference-layout form, or switch the weights to half, without changing the total size.
         * Signature = FNV-1a over every per-layer offset and form flag. */
        uint64_t sig = 1469598103934665603ull;
#define SIG_MIX(v) do { uint64_t v_ = (uint64_t)(v); int b_; for (b_ = 0; b_ < 8; ++b_) { sig ^= (v_ >> (8 * b_)) & 0xff; sig *= 1099511628211ull; } } while (0)
        for (i = 0; i < net->n; ++i) {
            const layer *l = &net->layers[i];
            const y2_ldev *d = ld_of(l);
            if (l->type != CONVOLUTIONAL && l->type != CONNECTED && l->type != LOCAL && l->type != BATCHNORM) continue;
            SIG_MIX(i); SIG_MIX(d->off_w_packed); SIG_MIX(d->has_w_ref ? d->off_w_ref + 1 : 0); SIG_MIX(d->off_bias);
            SIG_MIX(d->uses_mfma); SIG_MIX((i > 0) && ld_of(&net->layers[i - 1])->out_half);
            SIG_MIX(l->batch_normalize ? d->off_rinv + 1 : 0);
        }
        SIG_MIX(e->strict); SIG_MIX(e->half); SIG_MIX(off);
#undef SIG_MIX
        const int had_layout = e->arena_sig != 0;
        if (off != e->arena_bytes || !e->arena) {
            if (e->arena) y2h_free(e->arena);
            e->arena = NULL;
            e->arena_bytes = off;
            HIPCALL(y2h_malloc((void **)&e->arena, off));
            e->arena_sig = 0;
        }
        if (sig != e->arena_sig) {
            if (e->weights_external && had_layout) {
                /* a replicated rank holds no host weights to re-pack from: silently keeping (or re-uploading zeros
                 * over) an arena of another layout would compute garbage */
                e->weights_external = 0;
                y2_fail("the weight arena was filled from outside (y2_weights_resident) for another plan "
                        "(strict / fp16 / fusion / size changed its layout): call y2_weights_arena() again and replicate the "
                        "weights for the new plan");
                return -1;
            }
            if (!e->weights_external) e->weights_dirty = 1;
            e->arena_sig = sig;
        }
    }
    /* timing events */
    if (e->n_ev != net->n + 1) {
        if (e->ev) { for (i = 0; i < e->n_ev; ++i) y2h_event_destroy(e->ev[i]); free(e->ev); }
        e->n_ev = net->n + 1;
        e->ev = calloc(e->n_ev, sizeof(y2h_event));
        for (i = 0; i < e->n_ev; ++i) HIPCALL(y2h_event_create(&e->ev[i]));
    }
    e->built = 1;
    e->built_batch = net->batch; e->built_w = net->w; e->built_h = net->h; e->built_strict = e->strict;
    e->built_fusion = e->fusion;
    e->built_half = e->half;
    if (!e->weights_external) e->arena_pending = 0;              /* an ordinary build uploads the host weights below */
    e->built_autotune = e->arena_pending ? 0 : e->autotune;      /* a skipped measurement is made up for at the next forward */
    if (e->weights_dirty && !e->weights_external && upload_weights(net) != 0) return -1;
    if (autotune_allowed(e)) {
        /* measured tile shapes: needs the buffers and the arena, so it runs last; the scratch is sized again afterwards */
        if (autotune_layers(net) != 0 || size_workspace(net) != 0) { e->built = 0; return -1; }
    }
    return 0;
}

static int ensure_built(network *net)
{
    y2_engine *e = y2_engine_of(net);
    if (!e) { y2_fail("network has no engine (was it built by parse_network_cfg?)"); return -1; }
    if (!e->built || e->built_batch != net->batch || e->built_w != net->w || e->built_h != net->h ||
        e->built_strict != e->strict || e->built_fusion != e->fusion || e->built_half != e->half ||
        e->built_autotune != e->autotune) {
        if (y2_engine_build(net) != 0) return -1;
    } else {
        HIPCALL(y2h_set_device(e->device));
        if (e->weights_dirty && !e->weights_external && upload_weights(net) != 0) return -1;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* forward                                                             */
/* ------------------------------------------------------------------ */
static int enqueue_forward(network *net, const float *d_input_nchw);

/* One forward pass = a fixed sequence of 20-60 kernel launches with fixed arguments as long as the plan and the input
 * pointer stay the same.  At batch 1 (the Kinect application's mode) most of them run for 5-30 us, the same order as
 * the host-side cost of a launch; with y2_set_graph the sequence is captured into a hipGraph at the first call and
 * replayed with one hipGraphLaunch afterwards.  A new input pointer re-records; a new plan drops the graph. */
int y2_engine_forward(network *net, const float *d_input_nchw)
{
    y2_engine *e;
    if (ensure_built(net) != 0) return -1;
    e = y2_engine_of(net);
    if (net->c <= 0 || net->h <= 0 || net->w <= 0) { y2_fail("network input must be an image (h,w,c > 0)"); return -1; }
    if (!d_input_nchw) d_input_nchw = e->d_in_nchw;     /* filled by y2_ingest_u8 */
    if (!e->graph_on || e->timing || e->strict) return enqueue_forward(net, d_input_nchw);
    /* y2_set_detect_overlap together with graph replay: the wait that keeps this forward's region layer from overwriting
     * d_region while the previous batch's decode / NMS still read it on det_stream cannot live inside the graph (it would
     * be captured once, against whatever det_pending was then, on an event recorded outside the capture).  It is issued
     * here, in front of the capture and of every replay: the whole forward waits, slightly more than the eager path's
     * wait in front of the region layer, and the captured sequence itself carries no wait (e->capturing). */
    if (e->det_overlap && e->det_pending == 1 && e->ev_det) HIPCALL(y2h_stream_wait_event(e->stream, e->ev_det));
    if (!e->graph || e->graph_src != d_input_nchw) {
        /* one recording per input pointer, up to four (y2_feed_forward alternates between its HBM slots: with a single
         * recording every step of a double-buffered feed would capture and instantiate again) */
        int k, slot = -1;
        for (k = 0; k < 4; ++k) if (e->graphs[k] && e->graph_srcs[k] == d_input_nchw) slot = k;
        if (slot < 0) {
            y2h_graph g = NULL;
            slot = e->graph_next;
            e->graph_next = (e->graph_next + 1) & 3;
            if (e->graphs[slot]) { y2h_graph_destroy(e->graphs[slot]); e->graphs[slot] = NULL; e->graph_srcs[slot] = NULL; }
            e->graph = NULL; e->graph_src = NULL;
            HIPCALL(y2h_graph_begin(e->stream));
            e->capturing = 1;
            if (enqueue_forward(net, d_input_nchw) != 0) { e->capturing = 0; y2h_graph_abort(e->stream); return -1; }
            e->capturing = 0;
            if (y2h_graph_end(e->stream, &g) != 0) { y2_fail("hipGraph capture of the forward pass failed: %s", y2h_last_error()); return -1; }
            e->graphs[slot] = g; e->graph_srcs[slot] = d_input_nchw;
        }
        e->graph = e->graphs[slot];
        e->graph_src = d_input_nchw;
    }
    e->cur_input = d_input_nchw;
    HIPCALL(y2h_graph_launch(e->graph, e->stream));
    return 0;
}

static int enqueue_forward(network *net, const float *d_input_nchw)
{
    y2_engine *e = y2_engine_of(net);
    int i, k;
    e->cur_input = d_input_nchw;
    if (e->in_halo == 3 && (!e->half || ((uintptr_t)d_input_nchw % 16) == 0))
        ;                                                       /* the first layer reads d_input_nchw (fp32 kernel: dword loads, any float pointer) */
    else if (e->in_halo == 3) {
        /* the fp16 first-layer kernel reads the planes with 16-byte loads: a caller's pointer that is not 16-byte aligned
         * (a frame slice of an odd-sized batch) goes through the engine's own input slot */
        if (d_input_nchw != e->d_in_nchw) HIPCALL(y2h_memcpy_d2d(e->d_in_nchw, d_input_nchw, e->in_floats * sizeof(float), e->stream));
        e->cur_input = e->d_in_nchw;
    }
    else if (e->in_halo == 2)
        HIPCALL(y2h_nchw_to_nhwc4_halo_f16(d_input_nchw, e->d_in_nhwc, net->batch, net->c, net->h, net->w, e->stream));
    else if (e->in_halo)
        HIPCALL(y2h_nchw_to_nhwc_halo(d_input_nchw, e->d_in_nhwc, net->batch, net->c, net->h, net->w, net->c, e->in_halo_px, e->stream));
    else
        HIPCALL(y2h_nchw_to_nhwc(d_input_nchw, e->d_in_nhwc, net->batch, net->c, net->h, net->w, net->c, e->stream));
    if (e->timing) HIPCALL(y2h_event_record(e->ev[0], e->stream));
    for (i = 0; i < net->n; ++i) {
        layer *l = &net->layers[i];
        y2_ldev *d = ld_of(l);
        const float *x; int ldx;
        input_view(net, i, &x, &ldx);
        switch (l->type) {
        case CONVOLUTIONAL: {
            y2h_conv c;
            if (l->xnor) {
                /* input_view already points at d_bin: binarize the producer's activations into it first */
                const float *px; int pld;
                if (i == 0) { px = e->d_in_nhwc; pld = net->c; }
                else { const y2_ldev *p = ld_of(&net->layers[i - 1]); px = p->out; pld = p->out_ld; }
                HIPCALL(y2h_binarize(px, pld, d->d_bin, (long)l->batch * l->h * l->w, l->c, e->stream));
            }
            conv_desc(net, i, &c, x, ldx);
            HIPCALL(y2h_conv_forward(&c, e->strict, e->stream));
            if (!act_in_kernel(l->activation))
                HIPCALL(y2h_activate_array(d->out, d->out_ld, (long)l->batch * l->out_h * l->out_w, l->out_c, act_code(l->activation), e->stream));
        } break;
        case MAXPOOL:
            if (d->fused_into >= 0) break;       /* already produced by the conv before it */
            if (d->out_half)
                HIPCALL(y2h_maxpool_f16(x, ldx, d->out, d->out_ld, l->batch, l->h, l->w, l->c, l->size, l->stride, l->pad,
                                        l->out_h, l->out_w, e->stream));
            else
                HIPCALL(y2h_maxpool(x, ldx, d->out, d->out_ld, l->batch, l->h, l->w, l->c, l->size, l->stride, l->pad,
                                    l->out_h, l->out_w, e->stream));
            break;
        case REORG:
            if (d->out_half)
                HIPCALL(y2h_reorg_f16(x, ldx, d->out, d->out_ld, l->batch, l->h, l->w, l->c, l->stride, l->reverse, e->stream));
            else
                HIPCALL(y2h_reorg(x, ldx, d->out, d->out_ld, l->batch, l->h, l->w, l->c, l->stride, l->reverse, e->stream));
            break;
        case ROUTE:
            if (l->n >= 2 && d->copy_mask) {
                int choff = 0;
                for (k = 0; k < l->n; ++k) {
                    layer *src = &net->layers[l->input_layers[k]];
                    y2_ldev *sd = ld_of(src);
                    if ((d->copy_mask & (1u << k)) && d->out_half)
                        HIPCALL(y2h_copy_channels_f16(sd->out, sd->out_ld, (unsigned short *)d->out + choff, d->out_ld, src->out_c,
                                                      (long)l->batch * l->out_h * l->out_w, e->stream));
                    else if (d->copy_mask & (1u << k))
                        HIPCALL(y2h_copy_channels(sd->out, sd->out_ld, d->out + choff, d->out_ld, src->out_c,
                                                  (long)l->batch * l->out_h * l->out_w, e->stream));
                    choff += src->out_c;
                }
            }
            break;
        case REGION: {
            tree *t = l->softmax_tree;
            /* y2_set_detect_overlap: the previous batch's decode / NMS may still be reading d_region on det_stream */
            if (e->det_overlap && e->det_pending == 1 && i == e->out_layer && !e->capturing) HIPCALL(y2h_stream_wait_event(e->stream, e->ev_det));
            /* The (score, class) pair per box that detect mode needs, as a by-product of this layer: it saves the detect call
             * a sweep over the class rows (batch-1 latency), but it lengthens the forward; with y2_set_detect_overlap that
             * sweep runs beside the NEXT forward on the detection stream, where it is the cheaper place (yolo9000 544 b8:
             * 1999 against 1977 images/s, profiles/r03_notes.md section 10) */
            d->tree_best_valid = 0;
            if (t && d->d_tree_best && !e->det_overlap) d->tree_best_valid = 1;
            if (t)
                /* (strict mode keeps the reference's double exp in the group softmax; otherwise expf: the 9418 double exps per
                 * box are what this layer costs in yolo9000) */
                HIPCALL(y2h_region_forward_tree(x, ldx, d->d_region, l->batch, l->h * l->w, l->n, l->classes, l->coords, t->groups,
                                                d->d_tree_gsize, d->d_tree_goff, d->d_tree_parent, d->d_tree_order, d->d_tree_loff,
                                                d->tree_levels, d->tree_best_valid ? d->d_tree_best : NULL,
                                                e->strict ? 0 : Y2H_REGION_FAST_EXP, e->stream));
            else
            HIPCALL(y2h_region_forward(x, ldx, d->d_region, l->batch, l->h * l->w, l->n, l->classes, l->coords, l->softmax,
                                       0, d->d_tree_gsize, d->d_tree_goff, e->stream));
        } break;
        case AVGPOOL:
            if (i > 0 && ld_of(&net->layers[i - 1])->out_half)
                HIPCALL(y2h_avgpool_f16(x, ldx, d->d_flat, l->batch, l->h, l->w, l->c, e->stream));
            else
                HIPCALL(y2h_avgpool(x, ldx, d->d_flat, l->batch, l->h, l->w, l->c, e->stream));
            break;
        case SOFTMAX: {
            /* the input of a softmax layer is a flat [batch][inputs] vector; an image-like producer
             * (1x1 spatial, as after avgpool) is contiguous when its stride equals its channel count */
            layer *pl = &net->layers[i - 1];
            if (i == 0 || (pl->out_h * pl->out_w > 1 && pl->type != AVGPOOL && pl->type != SOFTMAX)) {
                y2_fail("softmax layer %d: input must be a flat vector (e.g. after avgpool)", i);
                return -1;
            }
            if (l->softmax_tree) { y2_fail("softmax layer with tree= is not implemented on the device"); return -1; }
            HIPCALL(y2h_softmax_rows(x, d->d_flat, (long)l->batch * l->groups, l->inputs / l->groups, l->temperature, e->stream));
        } break;
        case CONNECTED: {
            /* connected_layer.c:141-176.  Fast path: a 1x1 convolution over a 1x1 image on the matrix cores (weights
             * re-ordered for an NHWC producer at upload); otherwise, and in strict mode, the reference-order kernel */
            y2h_conv c;
            conv_desc(net, i, &c, x, ldx);
            if (d->uses_mfma && !e->strict) HIPCALL(y2h_conv_forward(&c, 0, e->stream));
            else {
                const int pi = producer_of(net, i);
                const layer *pl = &net->layers[pi];
                const int flat = is_flat(net, pi);
                const int hw = flat ? 1 : pl->out_h * pl->out_w, cc = l->inputs / hw;
                HIPCALL(y2h_connected_ref(x, (long)l->inputs, flat ? cc : ld_of(pl)->out_ld, hw, cc, c.w_ref, d->d_flat, l->outputs,
                                          l->batch, l->batch_normalize, c.activation, c.mean, c.rinv, c.scale, c.bias, e->stream));
            }
            if (!act_in_kernel(l->activation))
                HIPCALL(y2h_activate_array(d->d_flat, l->outputs, (long)l->batch, l->outputs, act_code(l->activation), e->stream));
        } break;
        case DROPOUT:
            break;                    /* dropout_layer.c:34: nothing happens at inference; the output is the input */
        case DETECTION: {
            /* detection_layer.c:49-66 at inference: copy, then a softmax over every cell's class scores */
            int b;
            HIPCALL(y2h_memcpy_d2d(d->d_flat, x, (size_t)l->batch * l->outputs * sizeof(float), e->stream));
            if (l->softmax)
                for (b = 0; b < l->batch; ++b)
                    HIPCALL(y2h_softmax_rows(d->d_flat + (size_t)b * l->outputs, d->d_flat + (size_t)b * l->outputs,
                                             (long)l->side * l->side, l->classes, 1.f, e->stream));
        } break;
        case SHORTCUT: {
            const y2_ldev *fd = ld_of(&net->layers[l->index]);
            if (act_code(l->activation) < 0) { y2_fail("shortcut layer %d: unknown activation %d", i, (int)l->activation); return -1; }
            if (i == 0) { y2_fail("shortcut layer %d has no input layer", i); return -1; }
            HIPCALL(y2h_shortcut(x, ldx, fd->out, fd->out_ld, d->out, d->out_ld, l->batch, l->w, l->h, l->c,
                                 l->out_w, l->out_h, l->out_c, act_for_kernel(l->activation), e->stream));
            if (!act_in_kernel(l->activation))
                HIPCALL(y2h_activate_array(d->out, d->out_ld, (long)l->batch * l->out_h * l->out_w, l->out_c, act_code(l->activation), e->stream));
        } break;
        case CROP:
            HIPCALL(y2h_crop(x, ldx, d->out, d->out_ld, l->batch, l->h, l->w, l->c, l->out_h, l->out_w, l->noadjust, 0, e->stream));
            if (d->d_halo)
                HIPCALL(y2h_crop(x, ldx, d->d_halo, l->out_c, l->batch, l->h, l->w, l->c, l->out_h, l->out_w, l->noadjust, d->halo_px, e->stream));
            break;
        case BATCHNORM:
            HIPCALL(y2h_batchnorm(x, ldx, d->out, d->out_ld, (long)l->batch * l->h * l->w, l->c, (const float *)(e->arena + d->off_mean),
                                  (const double *)(e->arena + d->off_rinv), (const float *)(e->arena + d->off_scale), e->stream));
            break;
        case LOCAL: {
            if (act_code(l->activation) < 0) { y2_fail("local layer %d: unknown activation %d", i, (int)l->activation); return -1; }
            HIPCALL(y2h_local(x, ldx, (const float *)(e->arena + d->off_w_packed), (const float *)(e->arena + d->off_bias), d->out,
                              d->out_ld, l->batch, l->h, l->w, l->c, l->n, l->size, l->stride, l->pad, l->out_h, l->out_w,
                              act_for_kernel(l->activation), e->strict, e->stream));
            if (!act_in_kernel(l->activation))
                HIPCALL(y2h_activate_array(d->out, d->out_ld, (long)l->batch * l->out_h * l->out_w, l->out_c, act_code(l->activation), e->stream));
        } break;
        case COST:
            break;                    /* cost_layer.c:75: nothing happens without truth */
        default:
            y2_fail("layer %d: unsupported type", i);
            return -1;
        }
        if (e->timing) HIPCALL(y2h_event_record(e->ev[i + 1], e->stream));
    }
    return 0;
}

/* copy the output layer to the pinned host buffer in the reference's layout */
int y2_engine_fetch_output(network *net)
{
    y2_engine *e = y2_engine_of(net);
    layer *l = &net->layers[e->out_layer];
    y2_ldev *d = ld_of(l);
    const float *src;
    if (is_flat(net, e->out_layer)) src = d->out;
    else {
        if (d->out_half)
            HIPCALL(y2h_nhwc_f16_to_nchw(d->out, d->out_ld, e->d_out_nchw, l->batch, l->out_c, l->out_h, l->out_w, e->stream));
        else
            HIPCALL(y2h_nhwc_to_nchw(d->out, d->out_ld, e->d_out_nchw, l->batch, l->out_c, l->out_h, l->out_w, e->stream));
        src = e->d_out_nchw;
    }
    if (e->h_out_pinned) {          /* the synchronous call: straight into the caller's buffer */
        HIPCALL(y2h_memcpy_d2h(e->h_out, src, e->out_floats * sizeof(float), e->stream));
        HIPCALL(y2h_stream_sync(e->stream));
        return 0;
    }
    HIPCALL(y2h_memcpy_d2h(e->h_out_stage, src, e->out_floats * sizeof(float), e->stream));
    HIPCALL(y2h_stream_sync(e->stream));
    memcpy(e->h_out, e->h_out_stage, e->out_floats * sizeof(float));
    return 0;
}

/* the same copy without the wait: an event is recorded behind it for y2_output_fetch */
int y2_output_enqueue(network net)
{
    y2_engine *e = y2_engine_of(&net);
    layer *l;
    y2_ldev *d;
    const float *src;
    if (!e || !e->built) { y2_fail("y2_output_enqueue: run a forward first"); return -1; }
    l = &net.layers[e->out_layer];
    d = ld_of(l);
    HIPCALL(y2h_set_device(e->device));
    if (!e->ev_out) HIPCALL(y2h_event_create(&e->ev_out));
    if (is_flat(&net, e->out_layer)) src = d->out;
    else {
        if (d->out_half)
            HIPCALL(y2h_nhwc_f16_to_nchw(d->out, d->out_ld, e->d_out_nchw, l->batch, l->out_c, l->out_h, l->out_w, e->stream));
        else
            HIPCALL(y2h_nhwc_to_nchw(d->out, d->out_ld, e->d_out_nchw, l->batch, l->out_c, l->out_h, l->out_w, e->stream));
        src = e->d_out_nchw;
    }
    HIPCALL(y2h_memcpy_d2h(e->h_out_stage, src, e->out_floats * sizeof(float), e->stream));
    HIPCALL(y2h_event_record(e->ev_out, e->stream));
    e->out_pending = 1;
    return 0;
}

float *y2_output_fetch(network net)
{
    y2_engine *e = y2_engine_of(&net);
    if (!e || !e->out_pending) { y2_fail("y2_output_fetch: nothing was enqueued (call y2_output_enqueue after a forward)"); return NULL; }
    if (y2h_event_sync(e->ev_out) != 0) { y2_fail("y2_output_fetch: %s", y2h_last_error()); return NULL; }
    e->out_pending = 0;
    memcpy(e->h_out, e->h_out_stage, e->out_floats * sizeof(float));
    return e->h_out;
}

/* ------------------------------------------------------------------ */
/* public runtime API                                                  */
/* ------------------------------------------------------------------ */
void cuda_set_device(int n)                  /* cuda.c:12-17 */
{
    gpu_index = n;
    if (n >= 0 && y2h_set_device(n) != 0) y2_fail("cuda_set_device(%d): %s", n, y2h_last_error());
}

float *network_predict(network net, float *input)
{
    y2_engine *e;
    if (ensure_built(&net) != 0) return NULL;
    e = y2_engine_of(&net);
    if (y2h_memcpy_h2d(e->d_in_nchw, input, e->in_floats * sizeof(float), e->stream) != 0) {
        y2_fail("input upload: %s", y2h_last_error());
        return NULL;
    }
    if (y2_engine_forward(&net, e->d_in_nchw) != 0) return NULL;
    if (y2_engine_fetch_output(&net) != 0) return NULL;
    return e->h_out;
}

float *network_predict_gpu(network net, float *input) { return network_predict(net, input); }

float *y2_network_predict_device(network net, const float *d_input)
{
    if (y2_engine_forward(&net, d_input) != 0) return NULL;
    if (y2_engine_fetch_output(&net) != 0) return NULL;
    return y2_engine_of(&net)->h_out;
}

int y2_forward_device(network net, const float *d_input) { return y2_engine_forward(&net, d_input); }

int y2_prepare(network *net) { return ensure_built(net); }

void y2_set_strict(network *net, int strict)
{
    y2_engine *e = y2_engine_of(net);
    if (e) e->strict = strict ? 1 : 0;
}

void y2_set_half(network *net, int on)
{
    y2_engine *e = y2_engine_of(net);
    if (e) e->half = on ? 1 : 0;
}

void y2_set_autotune(network *net, int on)
{
    y2_engine *e = y2_engine_of(net);
    if (e) e->autotune = on ? 1 : 0;
}

void y2_set_detect_overlap(network *net, int on)
{
    y2_engine *e = y2_engine_of(net);
    if (!e) return;
    if (e->det_pending == 1 && e->ev_det) y2h_event_sync(e->ev_det);     /* nothing of the other mode left in flight */
    e->det_overlap = on ? 1 : 0;
}

void y2_set_fusion(network *net, int on)
{
    y2_engine *e = y2_engine_of(net);
    if (e) e->fusion = on ? 1 : 0;
}

void y2_set_graph(network *net, int on)
{
    y2_engine *e = y2_engine_of(net);
    if (!e) return;
    e->graph_on = on ? 1 : 0;
    if (!on) drop_graphs(e);
}

void y2_set_timing(network *net, int on)
{
    y2_engine *e = y2_engine_of(net);
    if (e) e->timing = on ? 1 : 0;
}

int y2_layer_times_ms(network net, float *ms, int max_layers)
{
    y2_engine *e = y2_engine_of(&net);
    int i, n;
    if (!e || !e->built || !e->timing) return 0;
    n = net.n < max_layers ? net.n : max_layers;
    for (i = 0; i < n; ++i) if (y2h_event_elapsed_ms(e->ev[i], e->ev[i + 1], &ms[i]) != 0) return i;
    return n;
}

const char *y2_layer_kernel(network net, int i)
{
    if (i < 0 || i >= net.n || !net.layers[i].dev) return "";
    return ld_of(&net.layers[i])->kernel ? ld_of(&net.layers[i])->kernel : "";
}

int y2_weights_arena(network *net, void **dev_ptr, size_t *bytes)
{
    y2_engine *e;
    int keep;
    if (!y2_engine_of(net)) return -1;
    e = y2_engine_of(net);
    /* building must not try to upload host weights that were never loaded */
    keep = e->weights_dirty;
    if (!e->built || e->built_strict != e->strict || e->built_half != e->half || e->built_fusion != e->fusion ||
        e->built_batch != net->batch || e->built_w != net->w || e->built_h != net->h) {
        /* (re-)requesting the arena: whatever layout it had before no longer binds */
        e->weights_external = 1; e->arena_sig = 0;
        e->arena_pending = 1;                /* this build uploads nothing: the arena is uninitialised HBM until it is filled from outside */
        if (y2_engine_build(net) != 0) { e->weights_external = 0; e->arena_pending = 0; return -1; }
        e->weights_external = 0; e->weights_dirty = keep;
    }
    if (dev_ptr) *dev_ptr = e->arena;
    if (bytes) *bytes = e->arena_bytes;
    return 0;
}

void y2_weights_resident(network *net)
{
    y2_engine *e = y2_engine_of(net);
    if (e) { e->weights_external = 1; e->weights_dirty = 0; e->arena_pending = 0; }
}

void *y2_stream(network net) { y2_engine *e = y2_engine_of(&net); return e ? e->stream : NULL; }
void y2_sync(network net) { y2_engine *e = y2_engine_of(&net); if (e && e->stream) y2h_stream_sync(e->stream); }

int y2_pull_layer_output(network net, int i, float *dst)
{
    y2_engine *e = y2_engine_of(&net);
    layer *l;
    y2_ldev *d;
    float *tmp = NULL;
    size_t n;
    if (!e || !e->built || i < 0 || i >= net.n) { y2_fail("y2_pull_layer_output: no forward has run"); return -1; }
    l = &net.layers[i];
    d = ld_of(l);
    n = (size_t)l->batch * l->outputs;
    if (d->fused_pool) {
        y2_fail("layer %d is fused with the maxpool behind it and its full-resolution output is never stored; "
                "call y2_set_fusion(&net, 0) (or set Y2_NO_FUSE=1) to inspect it", i);
        return -1;
    }
    HIPCALL(y2h_set_device(e->device));
    if (is_flat(&net, i)) {
        HIPCALL(y2h_memcpy_d2h(dst, d->out, n * sizeof(float), e->stream));
        HIPCALL(y2h_stream_sync(e->stream));
        return 0;
    }
    HIPCALL(y2h_malloc((void **)&tmp, n * sizeof(float)));
    if ((d->out_half ? y2h_nhwc_f16_to_nchw(d->out, d->out_ld, tmp, l->batch, l->out_c, l->out_h, l->out_w, e->stream)
                     : y2h_nhwc_to_nchw(d->out, d->out_ld, tmp, l->batch, l->out_c, l->out_h, l->out_w, e->stream)) != 0 ||
        y2h_memcpy_d2h(dst, tmp, n * sizeof(float), e->stream) != 0 || y2h_stream_sync(e->stream) != 0) {
        y2h_free(tmp);
        y2_fail("y2_pull_layer_output: %s", y2h_last_error());
        return -1;
    }
    y2h_free(tmp);
    return 0;
}

float *get_network_output(network net)       /* network.c:173-181 */
{
    int i = y2_out_layer(&net);
    return net.layers[i].output;
}
float *get_network_output_gpu(network net) { return get_network_output(net); }
int get_network_output_size(network net) { return net.layers[y2_out_layer(&net)].outputs; }
int get_network_input_size(network net) { return net.layers[0].inputs; }

void set_batch_network(network *net, int b)  /* network.c:308-320 */
{
    int i;
    if (b <= 0) { y2_fail("set_batch_network: batch %d", b); return; }
    net->batch = b;
    for (i = 0; i < net->n; ++i) net->layers[i].batch = b;
    y2_engine_host_output(net);     /* HBM buffers are re-planned at the next predict if the batch changed */
}

int resize_network(network *net, int w, int h)   /* network.c:322-388 */
{
    int i, inputs = 0, k;
    net->w = w; net->h = h;
    net->inputs = w * h * net->c;
    for (i = 0; i < net->n; ++i) {
        layer *l = &net->layers[i];
        switch (l->type) {
        case CONVOLUTIONAL:                  /* convolutional_layer.c:360-398 */
            l->w = w; l->h = h;
            l->out_w = (l->w + 2 * l->pad - l->size) / l->stride + 1;
            l->out_h = (l->h + 2 * l->pad - l->size) / l->stride + 1;
            l->outputs = l->out_h * l->out_w * l->out_c;
            l->inputs = l->w * l->h * l->c;
            l->workspace_size = (size_t)l->out_h * l->out_w * l->size * l->size * l->c * sizeof(float);
            break;
        case MAXPOOL:                        /* maxpool_layer.c:54-77 */
            l->w = w; l->h = h;
            l->inputs = h * w * l->c;
            l->out_w = (w + 2 * l->pad) / l->stride;
            l->out_h = (h + 2 * l->pad) / l->stride;
            l->outputs = l->out_w * l->out_h * l->c;
            break;
        case CROP:                           /* crop_layer.c:48-66 */
            l->w = w; l->h = h;
            l->out_w = l->scale * w;
            l->out_h = l->scale * h;
            l->inputs = l->w * l->h * l->c;
            l->outputs = l->out_h * l->out_w * l->out_c;
            break;
        case REGION:                         /* region_layer.c:53-71 */
            l->w = w; l->h = h;
            l->outputs = h * w * l->n * (l->classes + l->coords + 1);
            l->inputs = l->outputs;
            break;
        case ROUTE: {                        /* route_layer.c:39-71 */
            layer *first = &net->layers[l->input_layers[0]];
            l->out_w = first->out_w; l->out_h = first->out_h; l->out_c = first->out_c;
            l->outputs = first->outputs;
            l->input_sizes[0] = first->outputs;
            for (k = 1; k < l->n; ++k) {
                layer *nx = &net->layers[l->input_layers[k]];
                l->outputs += nx->outputs;
                l->input_sizes[k] = nx->outputs;
                if (nx->out_w == first->out_w && nx->out_h == first->out_h) l->out_c += nx->out_c;
                else l->out_h = l->out_w = l->out_c = 0;
            }
            l->inputs = l->outputs;
            l->h = l->out_h; l->w = l->out_w; l->c = l->out_c;
        } break;
        case REORG:                          /* reorg_layer.c:45-76 */
            l->w = w; l->h = h;
            if (l->reverse) { l->out_w = w * l->stride; l->out_h = h * l->stride; l->out_c = l->c / (l->stride * l->stride); }
            else { l->out_w = w / l->stride; l->out_h = h / l->stride; l->out_c = l->c * (l->stride * l->stride); }
            l->outputs = l->out_h * l->out_w * l->out_c;
            l->inputs = l->outputs;
            break;
        case AVGPOOL:                        /* avgpool_layer.c:33-38 */
            l->w = w; l->h = h;
            l->inputs = h * w * l->c;
            break;
        case COST: case SOFTMAX:
            l->inputs = inputs; l->outputs = inputs;
            break;
        default:
            fprintf(stderr, "Resizing type %d \n", (int)l->type);
            y2_fail("Cannot resize this type of layer");
            return -1;
        }
        inputs = l->outputs;
        w = l->out_w; h = l->out_h;
        if (l->type == AVGPOOL) break;       /* network.c:366: layers after avgpool keep their size */
    }
    net->outputs = net->layers[y2_out_layer(net)].outputs;
    y2_engine_invalidate(net);
    y2_engine_host_output(net);
    return 0;
}

static void free_tree(tree *t)
{
    int i;
    if (!t) return;
    if (t->name) for (i = 0; i < t->n; ++i) free(t->name[i]);
    free(t->name); free(t->leaf); free(t->parent); free(t->group); free(t->group_size); free(t->group_offset);
    free(t);
}

void free_network(network net)               /* network.c:592-609 */
{
    int i;
    if (!net.layers) return;
    y2_engine_destroy(&net);
    for (i = 0; i < net.n; ++i) {
        layer *l = &net.layers[i];
        free(l->weights); free(l->biases); free(l->scales); free(l->rolling_mean); free(l->rolling_variance);
        free(l->input_layers); free(l->input_sizes); free(l->map); free(l->cost);
        free_tree(l->softmax_tree);
    }
    free(net.layers);
    free(net.seen);
}

void top_predictions(network net, int k, int *index)   /* network.c:449-454 */
{
    top_k(get_network_output(net), get_network_output_size(net), k, index);
}

char *get_layer_string(LAYER_TYPE a)         /* network.c:73-130 */
{
    switch (a) {
    case CONVOLUTIONAL: return "convolutional";
    case MAXPOOL: return "maxpool";
    case ROUTE: return "route";
    case REORG: return "reorg";
    case REGION: return "region";
    case AVGPOOL: return "avgpool";
    case SOFTMAX: return "softmax";
    case COST: return "cost";
    case SHORTCUT: return "shortcut";
    case CROP: return "crop";
    case LOCAL: return "local";
    case BATCHNORM: return "batchnorm";
    case CONNECTED: return "connected";
    case DROPOUT: return "dropout";
    case DETECTION: return "detection";
    default: return "none";
    }
}
