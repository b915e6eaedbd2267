// Shared between the fp32 (y2_conv.hip) and fp16 (y2_conv_f16.hip) convolution kernels.
#pragma once
#include "y2_common.hpp"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

struct ConvK {
    const float *x;    // fp16 kernels: const _Float16 *
    const float *w;
    float *y;          // y_f16: _Float16 *
    const float *mean;
    const double *rinv;
    const float *scale;
    const float *bias;
    const float *alpha, *beta;   // fp16 path: y = act(acc * alpha[co] + beta[co])
    int y_f16;         // 1: the output is stored as IEEE half
    int vec_store;     // fp16 kernels: half outputs leave as 16-byte stores (ldy, y and Cout aligned to 8 halves)
    int nchw;          // first-layer kernel: x is the fp32 NCHW network input (no halo, no NHWC copy)
    int dbg;           // ablation switches for tuning runs (env Y2_DBG; 0 in normal use): see y2_conv_f16.hip
    int H, W, Cin, ldx, Cout, ldy, K;
    int npix;          // GEMM rows = output pixels: batch * out_h * out_w (== batch * H * W at stride 1)
    int pool;          // 1: a 2x2 stride-2 maxpool is fused behind the activation (see pool_pixel)
    int bn, act;
    unsigned xbytes, wbytes;
    unsigned ybytes;   // fp32 8-wave tiles with vec_store: byte size of the output buffer from a.y on (buffer stores)
    int tiles_n;
    int ntiles;        // tiles_m * tiles_n * ksplit work items; workgroups walk them with stride gridDim.x
    int ksplit;        // >= 1: number of K ranges each output tile is cut into (split-K)
    float *ws;         // split-K partial sums [ksplit][npix][Cout]; stream-K: piece slots (see sk_tiles)
    // Stream-K (persistent kernels): the LAST sk_tiles output tiles are not walked whole.  Their sk_tiles * nk K-tile
    // iterations are dealt in equal contiguous shares to workgroups 0 .. sk_wgs-1 (share <= one tile: sk_tiles <= sk_wgs),
    // each share is one or two pieces (tile, K range) whose raw fp32 sums go to slot 2*wg + piece of `ws`, and a fix-up
    // launch adds a tile's pieces in ascending K order and applies the epilogue.  0 = every tile is walked whole.
    int sk_tiles, sk_wgs;
    int *sk_flags;     // hybrid stream-K of the fp32 kernel (conv_mfma_kernel<..., 2>): one flag per piece slot (2 * sk_wgs ints), zeroed before the launch
    // fp32 matrix-core kernel, wide heads (yolo9000's 28 269-filter 1x1: 116 MB of weights against 9.5 MB of input): tiles are
    // dealt per XCD -- workgroups b with the same b % 8 share an L2 -- so that XCD x owns filter tiles x, x + 8, ... (every
    // weight byte enters ONE L2) and walks them in blocks of `pblk` pixel tiles whose input rows stay L2-resident meanwhile.
    // 0 = plain order (filter tile fastest).  Needs gridDim.x % 8 == 0 and ksplit == 1; placement is speed only.
    int xcd_order, tiles_m, pblk;
    int row0;          // fp16 matrix-core kernels: first GEMM row of this launch (tail launch behind the 256x256 kernel); tile t starts at row0 + (t / tiles_n) * BM
    unsigned long long *stamps;   // diagnostic builds (-DY2_F32_STAMPS): per-wave cycle counters
    // stride / out_h / out_w: fp32 matrix-core and direct kernels; size / pad / batch: direct kernel only
    int size, stride, pad, out_h, out_w, batch;
};

// The reference's epilogue, step by step (blas.c:122, convolutional_layer.c:407-419, activations.h:35-41)
__device__ __forceinline__ float epilogue(float v, bool bn, float mean, double rinv, float scale, float bias, int act)
{
    if (bn) {
        float d = v - mean;                 // blas.c:122 numerator, fp32
        v = (float)((double)d * rinv);      // divide by (sqrt(var)+1e-6f) evaluated in double
        v = v * scale;                      // convolutional_layer.c:419
    }
    v = v + bias;                           // convolutional_layer.c:407
    if (act == Y2H_ACT_LEAKY) v = (v > 0) ? v : (float)(.1 * (double)v);              // activations.h:41
    else if (act == Y2H_ACT_LOGISTIC) v = (float)(1. / (1. + exp(-(double)v)));       // activations.h:35
    else if (act == Y2H_ACT_RELU) v = v * (float)(v > 0);                             // activations.h:37
    return v;
}

// Build option -DY2_FAST_EPILOGUE: the same steps in fp32 arithmetic for the matrix-core kernels (at most 2 ulp from
// epilogue()).  Measured on yolo.cfg 608 b32: +0.5-0.9 % images/s (A/B on one box) -- not worth leaving the reference's
// arithmetic, so the default keeps the exact form everywhere.
__device__ __forceinline__ float epilogue_f32(float v, bool bn, float mean, double rinv, float scale, float bias, int act)
{
#ifndef Y2_FAST_EPILOGUE
    return epilogue(v, bn, mean, rinv, scale, bias, act);
#else
    if (bn) {
        float d = v - mean;
        v = d * (float)rinv;
        v = v * scale;
    }
    v = v + bias;
    if (act == Y2H_ACT_LEAKY) v = (v > 0) ? v : .1f * v;
    else if (act == Y2H_ACT_LOGISTIC) v = (float)(1. / (1. + exp(-(double)v)));
    else if (act == Y2H_ACT_RELU) v = v * (float)(v > 0);
    return v;
#endif
}

// Fused conv + maxpool: the epilogue is a MONOTONE function of the accumulator -- every step (x - mean, * rinv > 0 in
// double, * scale, + bias, leaky / linear / logistic / relu) is monotone and every rounding is monotone; the whole is
// non-decreasing when scale >= 0 (or without batch-norm) and non-increasing otherwise.  So
//     max_t epilogue(acc_t) == epilogue(max_t acc_t)      (scale >= 0),      == epilogue(min_t acc_t)     (scale < 0)
// EXACTLY, and a pooling window costs one epilogue evaluation instead of four (the first layer was bound by this
// arithmetic: 378 M double-precision evaluations per 32 frames of 608x608).
__device__ __forceinline__ float pool_pick(float a0, float a1, float a2, float a3, bool increasing)
{
    const float mx = __builtin_fmaxf(__builtin_fmaxf(a0, a1), __builtin_fmaxf(a2, a3));
    const float mn = __builtin_fminf(__builtin_fminf(a0, a1), __builtin_fminf(a2, a3));
    return increasing ? mx : mn;
}

// fp16 path: BN folded into one fma on the fp32 accumulator (alpha = scale/(sqrt(var)+1e-6), beta = bias - mean*alpha),
// fp32 activation.  There is no reference arithmetic to mirror here: the reference has no half path.
__device__ __forceinline__ float epilogue_fast(float v, float alpha, float beta, int act)
{
    v = __builtin_fmaf(v, alpha, beta);
    if (act == Y2H_ACT_LEAKY) v = __builtin_fmaxf(v, 0.1f * v);       // slope < 1: max(v, .1v) is the leaky select
    else if (act == Y2H_ACT_LOGISTIC) v = 1.f / (1.f + __expf(-v));
    else if (act == Y2H_ACT_RELU) v = (v > 0.f) ? v : 0.f;
    return v;
}

// Fused conv + 2x2/2 maxpool (maxpool_layer.c:79-114 with size 2, stride 2, pad 0).
// The GEMM rows are enumerated in POOL-MAJOR order: row r = 4*q + t is pixel
// (2*yo + t/2, 2*xo + t%2) of pooling window q = (n, yo, xo).  An MFMA accumulator lane holds
// rows (reg&3) + 8*(reg>>2) + 4*half, i.e. registers 4g..4g+3 are the four pixels of ONE window,
// so the pool is a max over four registers of the same lane -- no cross-lane traffic -- and the
// full-resolution activation is never written.  Values are identical to conv followed by maxpool.
__device__ __forceinline__ int pool_pixel(int r, int H, int W)
{
    const int q = r >> 2, t = r & 3;
    const int Wp = W >> 1, HWp = (H >> 1) * Wp;
    const int n = q / HWp, rem = q - n * HWp;
    const int yo = rem / Wp, xo = rem - yo * Wp;
    return (n * H + 2 * yo + (t >> 1)) * W + 2 * xo + (t & 1);
}

// fp16 side (y2_conv_f16.hip)
bool y2_f16_conv_ok(const y2h_conv *d);
const char *y2_f16_conv_variant(const y2h_conv *d);
int y2_f16_conv_launch(const y2h_conv *d, ConvK &a, y2h_stream s);
size_t y2_f16_conv_workspace_bytes(const y2h_conv *d);
bool y2_f16_first_ok(const y2h_conv *d);
int y2_f16_first_launch(const y2h_conv *d, ConvK &a, y2h_stream s);
bool y2_f16_first_nchw_ok(const y2h_conv *d);
int y2_f16_first_nchw_launch(const y2h_conv *d, ConvK &a, y2h_stream s);

// Grid of a kernel that walks its tiles with a grid stride: exactly what is co-resident (256 CUs x the runtime's answer for
// this kernel), no more -- a surplus block only starts when a resident one has finished ALL its tiles, i.e. runs a second,
// nearly empty round (the first-layer kernel asked for four blocks per CU with registers for three: 1024 blocks, 768
// resident).  Cached per kernel and device.
static inline long resident_blocks(const void *fn, int threads, size_t lds, long fallback_per_cu)
{
    struct Entry { const void *fn; int dev; size_t lds; int per_cu; };
    static Entry cache[32];
    static int n = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256L * fallback_per_cu;
    for (int i = 0; i < n; ++i) if (cache[i].fn == fn && cache[i].dev == dev && cache[i].lds == lds) return 256L * cache[i].per_cu;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds) != hipSuccess || per_cu < 1) per_cu = (int)fallback_per_cu;
    if (getenv("Y2_FIRST_GRID_PER_CU")) per_cu = atoi(getenv("Y2_FIRST_GRID_PER_CU")) > 0 ? atoi(getenv("Y2_FIRST_GRID_PER_CU")) : per_cu;
    if (n < 32) { cache[n].fn = fn; cache[n].dev = dev; cache[n].lds = lds; cache[n].per_cu = per_cu; ++n; }
    return 256L * per_cu;
}

