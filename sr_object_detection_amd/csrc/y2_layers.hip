// Non-convolution layers of the forward path, NHWC fp32 (include/y2_hip.h).
// All of these are HBM-bound copies/reductions; the arithmetic follows the
// reference CPU order exactly so results are bit-identical to it.
// Compiled with -ffp-contract=off.
#include "y2_common.hpp"
#include <float.h>
#include <stdlib.h>

// the four in-kernel activations, with the reference's arithmetic (activations.h:35-41: leaky and logistic in double)
__device__ __forceinline__ float activate_ref(float v, int act)
{
    if (act == Y2H_ACT_LEAKY) v = (v > 0) ? v : (float)(.1 * (double)v);
    else if (act == Y2H_ACT_LOGISTIC) v = (float)(1. / (1. + exp(-(double)v)));
    else if (act == Y2H_ACT_RELU) v = v * (float)(v > 0);
    return v;
}

// ---------------------------------------------------------------------------
// maxpool  (src_yolo2/maxpool_layer.c:79-114; CUDA twin maxpool_layer_kernels.cu:10)
// window origin = -pad + o*stride, out-of-image taps read as -FLT_MAX, strict '>'
// ---------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(256) void maxpool_kernel(const float *__restrict__ x, int ldx, float *__restrict__ y, int ldy,
                                                      int h, int w, int c, int size, int stride, int pad,
                                                      int out_h, int out_w, long total)
{
    const int cv = c / V;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int cg = (int)(idx % cv);
        const long op = idx / cv;
        const int ox = (int)(op % out_w);
        const int oy = (int)((op / out_w) % out_h);
        const long n = op / ((long)out_w * out_h);
        float m[V];
#pragma unroll
        for (int v = 0; v < V; ++v) m[v] = -FLT_MAX;
        for (int kh = 0; kh < size; ++kh) {
            const int iy = -pad + oy * stride + kh;
            for (int kw = 0; kw < size; ++kw) {
                const int ix = -pad + ox * stride + kw;
                if (iy >= 0 && iy < h && ix >= 0 && ix < w) {
                    const float *src = x + ((n * h + iy) * (long)w + ix) * ldx + cg * V;
                    if (V == 4) {
                        const float4 q = *(const float4 *)src;
                        m[0] = (q.x > m[0]) ? q.x : m[0];
                        m[1 % V] = (q.y > m[1 % V]) ? q.y : m[1 % V];
                        m[2 % V] = (q.z > m[2 % V]) ? q.z : m[2 % V];
                        m[3 % V] = (q.w > m[3 % V]) ? q.w : m[3 % V];
                    } else {
                        const float q = *src;
                        m[0] = (q > m[0]) ? q : m[0];
                    }
                }
            }
        }
        float *dst = y + op * ldy + cg * V;
        if (V == 4) *(float4 *)dst = make_float4(m[0], m[1 % V], m[2 % V], m[3 % V]);
        else *dst = m[0];
    }
}

extern "C" int y2h_maxpool(const float *x, int ldx, float *y, int ldy, int batch, int h, int w, int c,
                           int size, int stride, int pad, int out_h, int out_w, y2h_stream s)
{
    if (batch <= 0 || h <= 0 || w <= 0 || c <= 0 || size <= 0 || stride <= 0 || ldx < c || ldy < c) return Y2H_EINVAL;
    if (out_h != (h + 2 * pad) / stride || out_w != (w + 2 * pad) / stride) return Y2H_EINVAL;
    const bool v4 = (c % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && (((uintptr_t)x | (uintptr_t)y) % 16 == 0);
    const long npix = (long)batch * out_h * out_w;
    if (v4) {
        const long total = npix * (c / 4);
        hipLaunchKernelGGL(maxpool_kernel<4>, dim3(y2h_grid(total, 256, 256 * 32)), dim3(256), 0, S(s),
                           x, ldx, y, ldy, h, w, c, size, stride, pad, out_h, out_w, total);
    } else {
        const long total = npix * c;
        hipLaunchKernelGGL(maxpool_kernel<1>, dim3(y2h_grid(total, 256, 256 * 32)), dim3(256), 0, S(s),
                           x, ldx, y, ldy, h, w, c, size, stride, pad, out_h, out_w, total);
    }
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// ---------------------------------------------------------------------------
// reorg  (src_yolo2/blas.c:8-29 reorg_cpu as called by reorg_layer.c:78-85)
//
// The reference works on flat NCHW indices.  For the ordinary (non-reverse)
// layer it calls reorg_cpu(x, w,h,c, batch, stride, forward=0, out): for every
// flat index f of the input-shaped iteration space (k,j,i) it gathers
//   out[f] = x[ w2 + w*s*(h2 + h*s*c2) ],  c2 = k % (c/s^2), off = k / (c/s^2),
//   w2 = i*s + off % s, h2 = j*s + off / s
// and the result is then *labelled* [c*s*s][h/s][w/s].  This kernel computes
// the same gather with both tensors in NHWC: flat output index f is decoded
// in the output's label geometry to find where it lives in NHWC, and the flat
// source index is decoded in the input's [c][h][w] geometry.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reorg_kernel(const float *__restrict__ x, int ldx, float *__restrict__ y, int ldy,
                                                    int h, int w, int c, int s, int reverse, long total)
{
    const int oc_small = c / (s * s);
    // label geometry of the output
    const int lo_c = reverse ? oc_small : c * s * s;
    const int lo_h = reverse ? h * s : h / s;
    const int lo_w = reverse ? w * s : w / s;
    const long per = (long)c * h * w;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        // idx enumerates NHWC output elements: (b, yo, xo, co), co fastest
        const int co = (int)(idx % lo_c);
        const long opix = idx / lo_c;
        const int xo = (int)(opix % lo_w);
        const int yo = (int)((opix / lo_w) % lo_h);
        const long b = opix / ((long)lo_w * lo_h);
        const long f = ((long)co * lo_h + yo) * lo_w + xo;     // flat NCHW index of this output element
        long q;                                                // flat NCHW index of the source element
        if (!reverse) {
            const int i = (int)(f % w);
            const int j = (int)((f / w) % h);
            const int k = (int)(f / ((long)w * h));
            const int c2 = k % oc_small, off = k / oc_small;
            const int w2 = i * s + off % s, h2 = j * s + off / s;
            q = w2 + (long)w * s * (h2 + (long)h * s * c2);
        } else {
            // forward=1: out[out_index] = x[in_index]; invert out_index = w2 + w*s*(h2 + h*s*c2)
            const int w2 = (int)(f % ((long)w * s));
            const int h2 = (int)((f / ((long)w * s)) % ((long)h * s));
            const int c2 = (int)(f / ((long)w * s * h * s));
            const int i = w2 / s, j = h2 / s;
            const int off = (h2 % s) * s + (w2 % s);
            const int k = off * oc_small + c2;
            q = i + (long)w * (j + (long)h * k);
        }
        const int xi = (int)(q % w);
        const int yi = (int)((q / w) % h);
        const int ci = (int)(q / ((long)w * h));
        (void)per;
        y[opix * ldy + co] = x[((b * h + yi) * (long)w + xi) * ldx + ci];
    }
}

extern "C" int y2h_reorg(const float *x, int ldx, float *y, int ldy, int batch, int h, int w, int c,
                         int stride, int reverse, y2h_stream s)
{
    if (batch <= 0 || h <= 0 || w <= 0 || c <= 0 || stride <= 0 || ldx < c) return Y2H_EINVAL;
    if (c % (stride * stride) != 0) return Y2H_EINVAL;
    if (!reverse && (h % stride != 0 || w % stride != 0)) return Y2H_EINVAL;
    const int oc = reverse ? c / (stride * stride) : c * stride * stride;
    if (ldy < oc) return Y2H_EINVAL;
    const long total = (long)batch * h * w * c;
    hipLaunchKernelGGL(reorg_kernel, dim3(y2h_grid(total, 256)), dim3(256), 0, S(s), x, ldx, y, ldy, h, w, c, stride,
                       reverse ? 1 : 0, total);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// ---------------------------------------------------------------------------
// global average pool (src_yolo2/avgpool_layer.c:40-54): sequential fp32 sum, one divide
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool_kernel(const float *__restrict__ x, int ldx, float *__restrict__ y,
                                                      int hw, int c, long total)
{
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int k = (int)(idx % c);
        const long b = idx / c;
        const float *src = x + b * hw * (long)ldx + k;
        float sum = 0.f;
        for (int i = 0; i < hw; ++i) sum += src[(long)i * ldx];
        y[idx] = sum / hw;
    }
}

extern "C" int y2h_avgpool(const float *x, int ldx, float *y, int batch, int h, int w, int c, y2h_stream s)
{
    if (batch <= 0 || h <= 0 || w <= 0 || c <= 0 || ldx < c) return Y2H_EINVAL;
    const long total = (long)batch * c;
    hipLaunchKernelGGL(avgpool_kernel, dim3(y2h_grid(total, 256)), dim3(256), 0, S(s), x, ldx, y, h * w, c, total);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// ---------------------------------------------------------------------------
// softmax (src_yolo2/blas.c:205-221): max-subtract, exp in double, fp32 running
// sum in index order, divide.  One thread walks one row (or one tree group) so
// the summation order is the reference's.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void softmax_seq(const float *in, int n, float temp, float *out)
{
    float sum = 0.f, largest = -FLT_MAX;
    for (int i = 0; i < n; ++i) if (in[i] > largest) largest = in[i];
    for (int i = 0; i < n; ++i) {
        const float e = (float)exp((double)(in[i] / temp - largest / temp));
        sum += e;
        out[i] = e;
    }
    for (int i = 0; i < n; ++i) out[i] /= sum;
}

// The same walk with the single-precision exponential (expf, ~1 ulp) in place of the reference's double exp rounded to float
// (blas.c:softmax): results differ by ~1e-7 relative, the order of the sums is the reference's.  9418 double exps per box are
// what region_tree_lds_kernel spends its time on (yolo9000 544 b8: 223 us per batch on the forward's critical path).
__device__ __forceinline__ void softmax_seq_fast(const float *in, int n, float temp, float *out)
{
    float sum = 0.f, largest = -FLT_MAX;
    for (int i = 0; i < n; ++i) if (in[i] > largest) largest = in[i];
    for (int i = 0; i < n; ++i) {
        const float e = expf(in[i] / temp - largest / temp);
        sum += e;
        out[i] = e;
    }
    for (int i = 0; i < n; ++i) out[i] /= sum;
}

// One workgroup per row.  The maximum (order-independent) and the double-precision exp of every
// element (the expensive part) are done by all lanes; only the fp32 running sum -- whose order the
// reference fixes -- is walked by one lane, so the result is still bit-identical to softmax_seq.
// LDS = true: the exponentials also stay in LDS (n floats of dynamic shared memory), so the one lane that walks the
// running sum reads them at LDS rather than at global-memory latency (1000 classes: 69 -> a few microseconds).
template <bool LDS>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float *x, float *y, long rows, int n, float temp)
{
    extern __shared__ float s_exp[];
    __shared__ float s_red[256];
    __shared__ float s_val;
    const float *in = x + (long)blockIdx.x * n;
    float *out = y + (long)blockIdx.x * n;
    const int t = threadIdx.x;
    float largest = -FLT_MAX;
    for (int i = t; i < n; i += 256) if (in[i] > largest) largest = in[i];
    s_red[t] = largest;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off && s_red[t + off] > s_red[t]) s_red[t] = s_red[t + off];
        __syncthreads();
    }
    largest = s_red[0];
    for (int i = t; i < n; i += 256) {
        const float e = (float)exp((double)(in[i] / temp - largest / temp));
        if (LDS) s_exp[i] = e; else out[i] = e;
    }
    __syncthreads();
    if (t == 0) {
        float sum = 0.f;
        if (LDS) for (int i = 0; i < n; ++i) sum += s_exp[i];
        else for (int i = 0; i < n; ++i) sum += out[i];
        s_val = sum;
    }
    __syncthreads();
    const float sum = s_val;
    for (int i = t; i < n; i += 256) out[i] = (LDS ? s_exp[i] : out[i]) / sum;
}

extern "C" int y2h_softmax_rows(const float *x, float *y, long rows, int n, float temp, y2h_stream s)
{
    if (rows <= 0 || n <= 0) return Y2H_EINVAL;
    if ((size_t)n * sizeof(float) <= 48 * 1024)
        hipLaunchKernelGGL(softmax_rows_kernel<true>, dim3((unsigned)rows), dim3(256), (size_t)n * sizeof(float), S(s), x, y, rows, n, temp);
    else
        hipLaunchKernelGGL(softmax_rows_kernel<false>, dim3((unsigned)rows), dim3(256), 0, S(s), x, y, rows, n, temp);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// ---------------------------------------------------------------------------
// region head (src_yolo2/region_layer.c:144-177, CPU build): the conv output in
// NHWC *is* the reference's flattened layout [cell][anchor][tx,ty,tw,th,obj,cls..]
// (blas.c:31 flatten), so this is a copy with logistic on objectness
// (activations.h:35, double) and softmax / tree softmax on the class scores.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void region_kernel(const float *__restrict__ x, int ldx, float *__restrict__ y,
                                                    long boxes, int hw, int num, int classes, int coords, int softmax)
{
    const long i = (long)blockIdx.x * 64 + threadIdx.x;   // box index over batch*hw*num
    if (i >= boxes) return;
    const int size = coords + 1 + classes;
    const int a = (int)(i % num);
    const long cell = i / num;                             // b*hw + cell
    const float *src = x + cell * ldx + (long)a * size;
    float *dst = y + i * size;
    for (int k = 0; k < coords; ++k) dst[k] = src[k];
    dst[coords] = (float)(1. / (1. + exp(-(double)src[coords])));
    if (softmax == 1) softmax_seq(src + coords + 1, classes, 1.f, dst + coords + 1);
    else if (softmax == 0) for (int k = 0; k < classes; ++k) dst[coords + 1 + k] = src[coords + 1 + k];
    // softmax == 2: tree; class scores are produced by region_tree_kernel
}

__global__ __launch_bounds__(256) void region_tree_kernel(const float *__restrict__ x, int ldx, float *__restrict__ y,
                                                          long boxes, int num, int classes, int coords, int groups,
                                                          const int *__restrict__ gsize, const int *__restrict__ goff)
{
    const long total = boxes * groups;
    const int size = coords + 1 + classes;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int g = (int)(idx % groups);
        const long i = idx / groups;
        const int a = (int)(i % num);
        const long cell = i / num;
        const float *src = x + cell * ldx + (long)a * size + coords + 1 + goff[g];
        float *dst = y + i * size + coords + 1 + goff[g];
        softmax_seq(src, gsize[g], 1.f, dst);
    }
}

// region_kernel with coalesced memory traffic: one thread per box still does the arithmetic (sequential, as the
// reference), but the rows of the 64 boxes of a workgroup travel through LDS (a box is `size` consecutive floats;
// thread-strided access to them cost 93 us per 32 frames of yolo.cfg 608).  size is odd for the cfgs of the family
// (85, 25, 30), so the per-thread LDS rows do not collide on banks.
__global__ __launch_bounds__(256) void region_lds_kernel(const float *__restrict__ x, int ldx, float *__restrict__ y,
                                                         long boxes, int num, int classes, int coords, int softmax)
{
    extern __shared__ float rows[];                        // [64][size] + [64] largest + [64] sum
    const int size = coords + 1 + classes;
    const long b0 = (long)blockIdx.x * 64;
    const int nb = (boxes - b0 < 64) ? (int)(boxes - b0) : 64;
    float *s_max = rows + 64 * size, *s_sum = s_max + 64;
    const int t = threadIdx.x;
    for (int idx = t; idx < nb * size; idx += 256) {
        const int bx = idx / size, k = idx - bx * size;
        const long i = b0 + bx;
        rows[idx] = x[(i / num) * ldx + (long)(i % num) * size + k];
    }
    __syncthreads();
    // The double-precision exponentials (all of the time: 80 per box) are independent, so every thread takes a share; only
    // the running fp32 sum, whose order the reference fixes (softmax_seq), is walked by one thread per box.  Same values,
    // same order, same bits as one thread per box doing everything (45 -> 15 us per 32 frames of yolo.cfg 608).
    if (t < nb) {
        float *r = rows + (size_t)t * size;
        r[coords] = (float)(1. / (1. + exp(-(double)r[coords])));
        float largest = -FLT_MAX;
        if (softmax == 1) for (int k = 0; k < classes; ++k) if (r[coords + 1 + k] > largest) largest = r[coords + 1 + k];
        s_max[t] = largest;
    }
    if (softmax == 1) {
        __syncthreads();
        for (int idx = t; idx < nb * classes; idx += 256) {
            const int bx = idx / classes, k = idx - bx * classes;
            float *v = rows + (size_t)bx * size + coords + 1 + k;
            *v = (float)exp((double)(*v / 1.f - s_max[bx] / 1.f));
        }
        __syncthreads();
        if (t < nb) {
            const float *r = rows + (size_t)t * size + coords + 1;
            float sum = 0.f;
            for (int k = 0; k < classes; ++k) sum += r[k];
            s_sum[t] = sum;
        }
        __syncthreads();
        for (int idx = t; idx < nb * classes; idx += 256) {
            const int bx = idx / classes, k = idx - bx * classes;
            rows[(size_t)bx * size + coords + 1 + k] /= s_sum[bx];
        }
    }
    __syncthreads();
    float *dst = y + b0 * size;
    for (int idx = t; idx < nb * size; idx += 256) dst[idx] = rows[idx];
}

// Same arithmetic, one workgroup per box: the class scores of the box are staged through LDS with coalesced
// loads, every thread then walks whole groups sequentially (the reference's order inside a group, softmax_seq),
// and the result leaves coalesced.  The thread-per-(box,group) kernel above reads and writes 4-byte pieces
// scattered over a 37 KB row (yolo9000: 590 us per 8 frames); this one moves each row once.
//
// With `tb.best_val` set the workgroup goes on, the row still in LDS, to what get_region_boxes will ask of it in detect mode
// (region_layer.c:351-367 without a map): hierarchy_predictions (tree.c:37-44: every node times its parent's final value,
// level by level -- nodes of one depth are independent when parents precede their children) and the deepest class whose
// probability exceeds .5 (the LAST such index), as one (score, class) pair per box for y2h_detect_tree_chain.  None of it
// depends on the detection threshold, and the layer's output is written before: the extra pass changes no output value
// and saves the detect call a second sweep over the [boxes][classes] scores (261 MB per batch of yolo9000 544 b8).
struct TreeBestK {
    const int *parent, *order, *level_off;
    int levels;
    float *best_val;
    int *best_cls;
};

#ifndef RT_NT
#define RT_NT 512        // threads per box: 8 wavefronts keep twice the bytes of a 37 KB row in flight (four rows per CU by LDS)
#endif
template <bool FAST>
__global__ __launch_bounds__(RT_NT) void region_tree_lds_kernel(const float *__restrict__ x, int ldx, float *__restrict__ y,
                                                              int num, int classes, int coords, int groups,
                                                              const int *__restrict__ gsize, const int *__restrict__ goff, TreeBestK tb)
{
    extern __shared__ float cls[];
    __shared__ int s_best;
    const long i = blockIdx.x;                      // box
    const int size = coords + 1 + classes;
    const int a = (int)(i % num);
    const long cell = i / num;
    const float *src = x + cell * ldx + (long)a * size + coords + 1;
    float *dst = y + i * size + coords + 1;
    for (int k = threadIdx.x; k < classes; k += RT_NT) cls[k] = src[k];
    __syncthreads();
    for (int g = threadIdx.x; g < groups; g += RT_NT) {
        if (FAST) softmax_seq_fast(cls + goff[g], gsize[g], 1.f, cls + goff[g]);
        else softmax_seq(cls + goff[g], gsize[g], 1.f, cls + goff[g]);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < classes; k += RT_NT) dst[k] = cls[k];
    if (!tb.best_val) return;
    if (threadIdx.x == 0) s_best = -1;
    __syncthreads();                                // every lane has read the row it stores before the row is edited
    for (int lv = 1; lv < tb.levels; ++lv) {        // level 0 = roots (parent < 0): unchanged
        const int b = tb.level_off[lv], e = tb.level_off[lv + 1];
        for (int q = b + threadIdx.x; q < e; q += RT_NT) {
            const int j = tb.order[q];
            cls[j] *= cls[tb.parent[j]];
        }
        __syncthreads();
    }
    int best = -1;
    for (int j = threadIdx.x; j < classes; j += RT_NT) if (cls[j] > .5) best = j;       // ascending: keeps the last
    if (best >= 0) atomicMax(&s_best, best);
    __syncthreads();
    if (threadIdx.x == 0) {
        best = s_best;
        tb.best_val[i] = best >= 0 ? cls[best] : 0.f;
        tb.best_cls[i] = best >= 0 ? best : 0;
    }
}

static bool region_tree_lds_ok(int classes, long boxes)
{
    return (size_t)classes * sizeof(float) <= 64 * 1024 && boxes < 0x7fffffffL && !getenv("Y2_REGION_TREE_SIMPLE");
}

extern "C" int y2h_region_tree_best_ok(int classes, int levels)
{
    return levels > 0 && region_tree_lds_ok(classes, 1) && !getenv("Y2_NO_TREE_BEST");
}

static int region_forward_impl(const float *x, int ldx, float *y, int batch, int hw, int num, int classes, int coords,
                               int softmax, int groups, const int *group_size, const int *group_offset, TreeBestK tb, int flags, y2h_stream s)
{
    if (batch <= 0 || hw <= 0 || num <= 0 || classes <= 0 || coords != 4) return Y2H_EINVAL;
    if (ldx < num * (coords + 1 + classes)) return Y2H_EINVAL;
    const long boxes = (long)batch * hw * num;
    const int mode = groups > 0 ? 2 : (softmax ? 1 : 0);
    const size_t row_bytes = ((size_t)64 * (coords + 1 + classes) + 128) * sizeof(float);
    if (mode != 2 && row_bytes <= 64 * 1024)
        hipLaunchKernelGGL(region_lds_kernel, dim3((unsigned)((boxes + 63) / 64)), dim3(256), row_bytes, S(s),
                           x, ldx, y, boxes, num, classes, coords, mode);
    else    // tree heads (their class scores are rewritten by the tree kernel below) and very wide rows
        hipLaunchKernelGGL(region_kernel, dim3((unsigned)((boxes + 63) / 64)), dim3(64), 0, S(s),
                           x, ldx, y, boxes, hw, num, classes, coords, mode);
    Y2H_LAUNCH_CHECK();
    if (groups > 0) {
        if (!group_size || !group_offset) return Y2H_EINVAL;
        if (region_tree_lds_ok(classes, boxes)) {
            if ((flags & Y2H_REGION_FAST_EXP) && !getenv("Y2_REGION_EXP_DOUBLE"))
                hipLaunchKernelGGL(region_tree_lds_kernel<true>, dim3((unsigned)boxes), dim3(RT_NT), (size_t)classes * sizeof(float), S(s),
                                   x, ldx, y, num, classes, coords, groups, group_size, group_offset, tb);
            else
                hipLaunchKernelGGL(region_tree_lds_kernel<false>, dim3((unsigned)boxes), dim3(RT_NT), (size_t)classes * sizeof(float), S(s),
                                   x, ldx, y, num, classes, coords, groups, group_size, group_offset, tb);
        } else {
            if (tb.best_val) return Y2H_EINVAL;          // (y2h_region_tree_best_ok said no)
            hipLaunchKernelGGL(region_tree_kernel, dim3(y2h_grid(boxes * groups, 256, 256 * 64)), dim3(256), 0, S(s),
                               x, ldx, y, boxes, num, classes, coords, groups, group_size, group_offset);
        }
        Y2H_LAUNCH_CHECK();
    } else if (tb.best_val) return Y2H_EINVAL;
    return Y2H_OK;
}

extern "C" int y2h_region_forward(const float *x, int ldx, float *y, int batch, int hw, int num, int classes, int coords,
                                  int softmax, int groups, const int *group_size, const int *group_offset, y2h_stream s)
{
    TreeBestK tb = {nullptr, nullptr, nullptr, 0, nullptr, nullptr};
    return region_forward_impl(x, ldx, y, batch, hw, num, classes, coords, softmax, groups, group_size, group_offset, tb, 0, s);
}

extern "C" int y2h_region_forward_tree(const float *x, int ldx, float *y, int batch, int hw, int num, int classes, int coords,
                                       int groups, const int *group_size, const int *group_offset, const int *parent,
                                       const int *order, const int *level_off, int levels, float *best /* val[boxes] | cls[boxes], or 0 */,
                                       int flags, y2h_stream s)
{
    if (groups <= 0) return Y2H_EINVAL;
    if (best && (!parent || !order || !level_off || !y2h_region_tree_best_ok(classes, levels))) return Y2H_EINVAL;
    const long boxes = (long)batch * hw * num;
    TreeBestK tb = {parent, order, level_off, levels, best, best ? (int *)(best + boxes) : nullptr};
    return region_forward_impl(x, ldx, y, batch, hw, num, classes, coords, 1, groups, group_size, group_offset, tb, flags, s);
}

// ---------------------------------------------------------------------------
// shortcut (residual add): shortcut_layer.c:38-43 = copy input, shortcut_cpu (blas.c:57-81), activate_array.
// `add` is the output of layer `from` (w1 x h1 x c1), the result has this layer's input shape (w2 x h2 x c2);
// with stride = w1/w2 and sample = w2/w1 (each at least 1) element (x*sample, y*sample, k) of the result receives
// add(x*stride, y*stride, k) for x < min(w1,w2), y < min(h1,h2), k < min(c1,c2).  One add per element, then the
// activation in the reference's arithmetic (leaky = .1*x in double).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void shortcut_kernel(const float *__restrict__ in, int ld_in, const float *__restrict__ add,
                                                       int ld_add, float *__restrict__ out, int ld_out, int w1, int h1, int c1,
                                                       int w2, int h2, int c2, int stride, int sample, int act, long total)
{
    const int minw = w1 < w2 ? w1 : w2, minh = h1 < h2 ? h1 : h2, minc = c1 < c2 ? c1 : c2;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int k = (int)(idx % c2);
        const long pix = idx / c2;
        const int x = (int)(pix % w2);
        const int y = (int)((pix / w2) % h2);
        const long b = pix / ((long)w2 * h2);
        float v = in[pix * ld_in + k];
        if (k < minc && x % sample == 0 && y % sample == 0 && x / sample < minw && y / sample < minh) {
            const long ap = (b * h1 + (long)(y / sample) * stride) * w1 + (long)(x / sample) * stride;
            v = v + add[ap * ld_add + k];
        }
        if (act == Y2H_ACT_LEAKY) v = (v > 0) ? v : (float)(.1 * (double)v);
        else if (act == Y2H_ACT_LOGISTIC) v = (float)(1. / (1. + exp(-(double)v)));
        else if (act == Y2H_ACT_RELU) v = v * (float)(v > 0);
        out[pix * ld_out + k] = v;
    }
}

// the common case -- both operands of the same shape (every [shortcut] of resnet50.cfg but the three that change width),
// channels a multiple of 4: 16-byte loads and stores, no coordinate decode
template <int ACT>        // a Y2H_ACT_* code compiled in, or -1: taken from `act_rt` (a per-value branch)
__global__ __launch_bounds__(256) void shortcut_same_kernel(const float *__restrict__ in, int ld_in, const float *__restrict__ add,
                                                            int ld_add, float *__restrict__ out, int ld_out, int c4, int act_rt, long total)
{
    const int act = ACT >= 0 ? ACT : act_rt;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long pix = idx / c4;
        const int k = (int)(idx - pix * c4) * 4;
        const float4 a = *(const float4 *)(in + pix * ld_in + k), b = *(const float4 *)(add + pix * ld_add + k);
        float4 r;
        r.x = activate_ref(a.x + b.x, act); r.y = activate_ref(a.y + b.y, act);
        r.z = activate_ref(a.z + b.z, act); r.w = activate_ref(a.w + b.w, act);
        *(float4 *)(out + pix * ld_out + k) = r;
    }
}

extern "C" int y2h_shortcut(const float *in, int ld_in, const float *add, int ld_add, float *out, int ld_out, int batch,
                            int w1, int h1, int c1, int w2, int h2, int c2, int activation, y2h_stream s)
{
    if (!in || !add || !out || batch <= 0 || w1 <= 0 || h1 <= 0 || c1 <= 0 || w2 <= 0 || h2 <= 0 || c2 <= 0) return Y2H_EINVAL;
    if (ld_in < c2 || ld_out < c2 || ld_add < c1) return Y2H_EINVAL;
    if (w1 == w2 && h1 == h2 && c1 == c2 && c2 % 4 == 0 && ld_in % 4 == 0 && ld_add % 4 == 0 && ld_out % 4 == 0 &&
        (((uintptr_t)in | (uintptr_t)add | (uintptr_t)out) & 15) == 0) {
        const long total4 = (long)batch * h2 * w2 * (c2 / 4);
        void (*fn)(const float *, int, const float *, int, float *, int, int, int, long) =
            activation == Y2H_ACT_LEAKY ? shortcut_same_kernel<Y2H_ACT_LEAKY> :
            activation == Y2H_ACT_LINEAR ? shortcut_same_kernel<Y2H_ACT_LINEAR> : shortcut_same_kernel<-1>;
        hipLaunchKernelGGL(fn, dim3(y2h_grid(total4, 256)), dim3(256), 0, S(s), in, ld_in, add, ld_add, out, ld_out,
                           c2 / 4, activation, total4);
        Y2H_LAUNCH_CHECK();
        return Y2H_OK;
    }
    int stride = w1 / w2, sample = w2 / w1;
    if (stride != h1 / h2 || sample != h2 / h1) return Y2H_EINVAL;      /* the reference asserts this (blas.c:61-62) */
    if (stride < 1) stride = 1;
    if (sample < 1) sample = 1;
    const long total = (long)batch * h2 * w2 * c2;
    hipLaunchKernelGGL(shortcut_kernel, dim3(y2h_grid(total, 256)), dim3(256), 0, S(s), in, ld_in, add, ld_add, out, ld_out,
                       w1, h1, c1, w2, h2, c2, stride, sample, activation, total);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// ---------------------------------------------------------------------------
// [crop] at inference (crop_layer.c:69-105 with !state.train): the centred out_h x out_w window of every image,
// each value mapped x*scale + trans (2, -1 unless noadjust).  NHWC in, NHWC out; one thread per value.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void crop_kernel(const float *__restrict__ x, int ldx, float *__restrict__ y, int ldy, int h, int w,
                                                   int c, int oh, int ow, int dh, int dw, int halo, float scale, float trans, long total)
{
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int k = (int)(idx % c);
        const long pix = idx / c;
        const int ox = (int)(pix % ow);
        const int oy = (int)((pix / ow) % oh);
        const long b = pix / ((long)ow * oh);
        const float v = x[((b * h + oy + dh) * w + ox + dw) * ldx + k];
        y[((b * (oh + 2 * halo) + oy + halo) * (long)(ow + 2 * halo) + ox + halo) * ldy + k] = v * scale + trans;
    }
}

extern "C" int y2h_crop(const float *x, int ldx, float *y, int ldy, int batch, int h, int w, int c, int out_h, int out_w,
                        int noadjust, int halo, y2h_stream s)
{
    if (!x || !y || batch <= 0 || h <= 0 || w <= 0 || c <= 0 || out_h <= 0 || out_w <= 0 || out_h > h || out_w > w || halo < 0) return Y2H_EINVAL;
    if (ldx < c || ldy < c) return Y2H_EINVAL;
    const long total = (long)batch * out_h * out_w * c;
    hipLaunchKernelGGL(crop_kernel, dim3(y2h_grid(total, 256)), dim3(256), 0, S(s), x, ldx, y, ldy, h, w, c, out_h, out_w,
                       (h - out_h) / 2, (w - out_w) / 2, halo, noadjust ? 1.f : 2.f, noadjust ? 0.f : -1.f, total);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// ---------------------------------------------------------------------------
// standalone [batchnorm] at inference (batchnorm_layer.c:122-146): y = ((x - mean) / (sqrt(var) + 1e-6f)) * scale per
// channel, the divide evaluated in double as in blas.c:122 (rinv = 1 / (sqrt(var) + 1e-6f) prepared in double).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void batchnorm_kernel(const float *__restrict__ x, int ldx, float *__restrict__ y, int ldy, int c,
                                                        const float *__restrict__ mean, const double *__restrict__ rinv,
                                                        const float *__restrict__ scale, long total)
{
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int k = (int)(idx % c);
        const long pix = idx / c;
        const float d = x[pix * ldx + k] - mean[k];
        float v = (float)((double)d * rinv[k]);
        y[pix * ldy + k] = v * scale[k];
    }
}

extern "C" int y2h_batchnorm(const float *x, int ldx, float *y, int ldy, long pixels, int c, const float *mean, const double *rinv,
                             const float *scale, y2h_stream s)
{
    if (!x || !y || !mean || !rinv || !scale || pixels <= 0 || c <= 0 || ldx < c || ldy < c) return Y2H_EINVAL;
    const long total = pixels * c;
    hipLaunchKernelGGL(batchnorm_kernel, dim3(y2h_grid(total, 256)), dim3(256), 0, S(s), x, ldx, y, ldy, c, mean, rinv, scale, total);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// ---------------------------------------------------------------------------
// [local] (local_layer.c:95-126): a convolution with a separate filter bank per output location.  Every weight is
// used once per image, so the layer streams its weights from HBM (yolov1/yolo.cfg: 49 x 256 x 9216 floats = 462 MB):
// one wave owns (location, 4 filters), lanes stride over the K = size*size*c taps with 16-byte loads of the weights
// (re-ordered at upload to [location][filter][kh][kw][c], matching the NHWC activations) and of the input, up to
// eight images per pass so the weights are read once per eight images; a wave reduction finishes the dot products.
//   y = act(bias + sum_k w*x)         (the reference adds the bias first; same value to rounding)
// local_ref_kernel is the strict-mode form: one thread per output value, bias first, taps in the reference's
// k = (c, kh, kw) order, product and sum rounded separately -- the arithmetic of gemm_nn (gemm.c:74-88).
// ---------------------------------------------------------------------------

// every activation of activations.h:21-54, with the reference's own promotion rules (float x, double constants, the
// result rounded to float on return)
__device__ float activate_any(float x, int act)
{
    const double xd = (double)x;
    switch (act) {
    case Y2H_ACT_LINEAR: return x;
    case Y2H_ACT_LEAKY: return (x > 0) ? x : (float)(.1 * xd);
    case Y2H_ACT_LOGISTIC: return (float)(1. / (1. + exp(-xd)));
    case Y2H_ACT_RELU: return x * (float)(x > 0);
    case Y2H_ACT_RELIE: return (x > 0) ? x : (float)(.01 * xd);
    case Y2H_ACT_RAMP: return (float)((double)(x * (float)(x > 0)) + .1 * xd);
    case Y2H_ACT_TANH: { const float t = 2 * x; return (float)((exp((double)t) - 1) / (exp((double)t) + 1)); }
    case Y2H_ACT_PLSE:
        if (x < -4) return (float)(.01 * (double)(x + 4));
        if (x > 4) return (float)(.01 * (double)(x - 4) + 1);
        return (float)(.125 * xd + .5);
    case Y2H_ACT_ELU: return (float)((double)((float)(x >= 0) * x) + (double)(x < 0) * (exp(xd) - 1));
    case Y2H_ACT_LOGGY: return (float)(2. / (1. + exp(-xd)) - 1);
    case Y2H_ACT_STAIR: {
        const int n = (int)floor(xd);
        if (n % 2 == 0) return (float)floor(xd / 2.);
        return (float)((double)(x - (float)n) + floor(xd / 2.));
    }
    case Y2H_ACT_HARDTAN: return x < -1 ? -1.f : (x > 1 ? 1.f : x);
    case Y2H_ACT_LHTAN:
        if (x < 0) return (float)(.001 * xd);
        if (x > 1) return (float)(.001 * (double)(x - 1) + 1);
        return x;
    }
    return x;
}

// binarize_cpu (convolutional_layer.c:52-58): +1 where the value is positive, -1 elsewhere; NHWC in, contiguous NHWC out
__global__ __launch_bounds__(256) void binarize_kernel(const float *__restrict__ x, int ldx, float *__restrict__ y, int c, long total)
{
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long row = idx / c;
        y[idx] = (x[row * ldx + (idx - row * c)] > 0) ? 1.f : -1.f;
    }
}

extern "C" int y2h_binarize(const float *x, int ldx, float *y, long rows, int c, y2h_stream s)
{
    if (!x || !y || rows <= 0 || c <= 0 || ldx < c) return Y2H_EINVAL;
    const long total = rows * c;
    hipLaunchKernelGGL(binarize_kernel, dim3(y2h_grid(total, 256)), dim3(256), 0, S(s), x, ldx, y, c, total);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

__global__ __launch_bounds__(256) void activate_kernel(float *__restrict__ x, int ld, int c, int act, long total)
{
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long row = idx / c;
        const int k = (int)(idx - row * c);
        x[row * ld + k] = activate_any(x[row * ld + k], act);
    }
}

extern "C" int y2h_activate_array(float *x, int ld, long rows, int c, int activation, y2h_stream s)
{
    if (!x || rows <= 0 || c <= 0 || ld < c || activation < 0 || activation > Y2H_ACT_LHTAN) return Y2H_EINVAL;
    const long total = rows * c;
    hipLaunchKernelGGL(activate_kernel, dim3(y2h_grid(total, 256)), dim3(256), 0, S(s), x, ld, c, activation, total);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

struct LocalK {
    const float *x; int ldx;
    const float *wp;            // [loc][n][kh][kw][c]
    const float *bias;          // [loc][n]
    float *y; int ldy;
    int batch, h, w, c, n, size, stride, pad, oh, ow, act;
};

template <int NB, bool VEC>
__global__ __launch_bounds__(256) void local_kernel(LocalK a)
{
    const int lane = threadIdx.x & 63;
    const int groups = (a.n + 3) >> 2;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const long nwork = (long)a.oh * a.ow * groups;
    if (wave >= nwork) return;
    const int loc = (int)(wave / groups), m0 = (int)(wave - (long)loc * groups) * 4;
    const int oy = loc / a.ow, ox = loc - oy * a.ow;
    const int K = a.size * a.size * a.c;
    const float *w0 = a.wp + ((size_t)loc * a.n + m0) * K;
    const int nf = (a.n - m0 < 4) ? a.n - m0 : 4;
    for (int b0 = 0; b0 < a.batch; b0 += NB) {
        float acc[NB][4];
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
        constexpr int STEP = VEC ? 4 : 1;
        for (int k = lane * STEP; k < K; k += 64 * STEP) {
            const int tap = k / a.c, ci = k - tap * a.c;
            const int kh = tap / a.size, kw = tap - kh * a.size;
            const int iy = oy * a.stride + kh - a.pad, ix = ox * a.stride + kw - a.pad;
            const bool in = iy >= 0 && iy < a.h && ix >= 0 && ix < a.w;
            float wv[4][STEP];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < nf) {
                    if constexpr (VEC) { const float4 t = *(const float4 *)(w0 + (size_t)j * K + k); wv[j][0] = t.x; wv[j][1] = t.y; wv[j][2] = t.z; wv[j][3] = t.w; }
                    else wv[j][0] = w0[(size_t)j * K + k];
                } else {
#pragma unroll
                    for (int q = 0; q < STEP; ++q) wv[j][q] = 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                float xv[STEP];
#pragma unroll
                for (int q = 0; q < STEP; ++q) xv[q] = 0.f;
                if (in && b0 + i < a.batch) {
                    const float *xp = a.x + (((size_t)(b0 + i) * a.h + iy) * a.w + ix) * a.ldx + ci;
                    if constexpr (VEC) { const float4 t = *(const float4 *)xp; xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w; }
                    else xv[0] = *xp;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < STEP; ++q) acc[i][j] = __builtin_fmaf(wv[j][q], xv[q], acc[i][j]);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[i][j];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                acc[i][j] = v;
            }
        if (lane < 4 && lane < nf) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (b0 + i >= a.batch) break;
                float v = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) v = (lane == j) ? acc[i][j] : v;
                v = a.bias[(size_t)loc * a.n + m0 + lane] + v;
                a.y[((size_t)(b0 + i) * a.oh * a.ow + loc) * a.ldy + m0 + lane] = activate_ref(v, a.act);
            }
        }
    }
}

__global__ __launch_bounds__(256) void local_ref_kernel(LocalK a)
{
    const long total = (long)a.batch * a.oh * a.ow * a.n;
    const int K = a.size * a.size * a.c;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int m = (int)(idx % a.n);
        const long r = idx / a.n;
        const int loc = (int)(r % ((long)a.oh * a.ow));
        const long b = r / ((long)a.oh * a.ow);
        const int oy = loc / a.ow, ox = loc - oy * a.ow;
        const float *wr = a.wp + ((size_t)loc * a.n + m) * K;
        float v = a.bias[(size_t)loc * a.n + m];
        for (int ci = 0; ci < a.c; ++ci)
            for (int kh = 0; kh < a.size; ++kh)
                for (int kw = 0; kw < a.size; ++kw) {
                    const int iy = oy * a.stride + kh - a.pad, ix = ox * a.stride + kw - a.pad;
                    const float xv = (iy >= 0 && iy < a.h && ix >= 0 && ix < a.w) ? a.x[((b * a.h + iy) * a.w + ix) * a.ldx + ci] : 0.f;
                    const float p = wr[(kh * a.size + kw) * a.c + ci] * xv;
                    v = v + p;
                }
        a.y[(b * a.oh * a.ow + loc) * a.ldy + m] = activate_ref(v, a.act);
    }
}

extern "C" int y2h_local(const float *x, int ldx, const float *w_packed, const float *bias_packed, float *y, int ldy, int batch,
                         int h, int w, int c, int n, int size, int stride, int pad, int out_h, int out_w, int activation,
                         int strict, y2h_stream s)
{
    if (!x || !w_packed || !bias_packed || !y || batch <= 0 || h <= 0 || w <= 0 || c <= 0 || n <= 0 || size <= 0 || stride <= 0 ||
        pad < 0 || out_h <= 0 || out_w <= 0 || ldx < c || ldy < n) return Y2H_EINVAL;
    if ((out_h - 1) * stride + size - pad > h + pad || (out_w - 1) * stride + size - pad > w + pad) return Y2H_EINVAL;
    LocalK a;
    a.x = x; a.ldx = ldx; a.wp = w_packed; a.bias = bias_packed; a.y = y; a.ldy = ldy; a.batch = batch; a.h = h; a.w = w; a.c = c;
    a.n = n; a.size = size; a.stride = stride; a.pad = pad; a.oh = out_h; a.ow = out_w; a.act = activation;
    if (strict) {
        hipLaunchKernelGGL(local_ref_kernel, dim3(y2h_grid((long)batch * out_h * out_w * n, 256)), dim3(256), 0, S(s), a);
        Y2H_LAUNCH_CHECK();
        return Y2H_OK;
    }
    const long waves = (long)out_h * out_w * ((n + 3) / 4);
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    const bool vec = (c % 4 == 0) && (ldx % 4 == 0) && (((uintptr_t)x | (uintptr_t)w_packed) % 16 == 0);
    if (batch == 1) {
        if (vec) hipLaunchKernelGGL((local_kernel<1, true>), dim3(blocks), dim3(256), 0, S(s), a);
        else hipLaunchKernelGGL((local_kernel<1, false>), dim3(blocks), dim3(256), 0, S(s), a);
    } else if (batch <= 4) {
        if (vec) hipLaunchKernelGGL((local_kernel<4, true>), dim3(blocks), dim3(256), 0, S(s), a);
        else hipLaunchKernelGGL((local_kernel<4, false>), dim3(blocks), dim3(256), 0, S(s), a);
    } else {        // eight images per pass over the weights
        if (vec) hipLaunchKernelGGL((local_kernel<8, true>), dim3(blocks), dim3(256), 0, S(s), a);
        else hipLaunchKernelGGL((local_kernel<8, false>), dim3(blocks), dim3(256), 0, S(s), a);
    }
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}
