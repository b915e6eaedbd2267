// Half-storage versions of the non-convolution layers (engine extension for BASELINE configs[4]; the
// reference has no fp16 path).  Same index arithmetic as y2_layers.hip / y2_runtime.hip, elements are
// IEEE half, `ld` arguments count halves.  max / copy / permutation are exact in half; the average pool
// accumulates in fp32 in the reference's sequential order and returns fp32.
#include "y2_common.hpp"
#include <stdint.h>
#include <stdlib.h>
#include <float.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// maxpool_layer.c:79-114 (window origin -pad + o*stride, out-of-image taps = -inf, strict '>')
template <int V>
__global__ __launch_bounds__(256) void maxpool_f16_kernel(const _Float16 *__restrict__ x, int ldx, _Float16 *__restrict__ y,
                                                          int ldy, int h, int w, int c, int size, int stride, int pad,
                                                          int out_h, int out_w, long total)
{
    const int cv = c / V;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int cg = (int)(idx % cv);
        const long op = idx / cv;
        const int ox = (int)(op % out_w);
        const int oy = (int)((op / out_w) % out_h);
        const long n = op / ((long)out_w * out_h);
        _Float16 m[V];
#pragma unroll
        for (int v = 0; v < V; ++v) m[v] = (_Float16)(-65504.f);
        for (int kh = 0; kh < size; ++kh) {
            const int iy = -pad + oy * stride + kh;
            for (int kw = 0; kw < size; ++kw) {
                const int ix = -pad + ox * stride + kw;
                if (iy >= 0 && iy < h && ix >= 0 && ix < w) {
                    const _Float16 *src = x + ((n * h + iy) * (long)w + ix) * ldx + cg * V;
                    if (V == 8) {
                        const h8 q = *(const h8 *)src;
#pragma unroll
                        for (int v = 0; v < V; ++v) m[v] = (q[v] > m[v]) ? q[v] : m[v];
                    } else {
                        const _Float16 q = *src;
                        m[0] = (q > m[0]) ? q : m[0];
                    }
                }
            }
        }
        _Float16 *dst = y + op * ldy + cg * V;
        if (V == 8) {
            h8 o;
#pragma unroll
            for (int v = 0; v < V; ++v) o[v] = m[v];
            *(h8 *)dst = o;
        } else *dst = m[0];
    }
}

extern "C" int y2h_maxpool_f16(const void *x, int ldx, void *y, int ldy, int batch, int h, int w, int c,
                               int size, int stride, int pad, int out_h, int out_w, y2h_stream s)
{
    if (!x || !y || batch <= 0 || h <= 0 || w <= 0 || c <= 0 || size <= 0 || stride <= 0 || ldx < c || ldy < c) return Y2H_EINVAL;
    if (out_h != (h + 2 * pad) / stride || out_w != (w + 2 * pad) / stride) return Y2H_EINVAL;
    const bool v8 = (c % 8 == 0) && (ldx % 8 == 0) && (ldy % 8 == 0) && (((uintptr_t)x | (uintptr_t)y) % 16 == 0);
    const long npix = (long)batch * out_h * out_w;
    if (v8) {
        const long total = npix * (c / 8);
        hipLaunchKernelGGL(maxpool_f16_kernel<8>, dim3(y2h_grid(total, 256, 256 * 32)), dim3(256), 0, S(s),
                           (const _Float16 *)x, ldx, (_Float16 *)y, ldy, h, w, c, size, stride, pad, out_h, out_w, total);
    } else {
        const long total = npix * c;
        hipLaunchKernelGGL(maxpool_f16_kernel<1>, dim3(y2h_grid(total, 256, 256 * 32)), dim3(256), 0, S(s),
                           (const _Float16 *)x, ldx, (_Float16 *)y, ldy, h, w, c, size, stride, pad, out_h, out_w, total);
    }
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// reorg with the reference's forward=0 quirk (blas.c:8-29 via reorg_layer.c:78-85); see y2_layers.hip reorg_kernel
__global__ __launch_bounds__(256) void reorg_f16_kernel(const _Float16 *__restrict__ x, int ldx, _Float16 *__restrict__ y,
                                                        int ldy, int h, int w, int c, int s, int reverse, long total)
{
    const int oc_small = c / (s * s);
    const int lo_c = reverse ? oc_small : c * s * s;
    const int lo_h = reverse ? h * s : h / s;
    const int lo_w = reverse ? w * s : w / s;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int co = (int)(idx % lo_c);
        const long opix = idx / lo_c;
        const int xo = (int)(opix % lo_w);
        const int yo = (int)((opix / lo_w) % lo_h);
        const long b = opix / ((long)lo_w * lo_h);
        const long f = ((long)co * lo_h + yo) * lo_w + xo;
        long q;
        if (!reverse) {
            const int i = (int)(f % w);
            const int j = (int)((f / w) % h);
            const int k = (int)(f / ((long)w * h));
            const int c2 = k % oc_small, off = k / oc_small;
            const int w2 = i * s + off % s, h2 = j * s + off / s;
            q = w2 + (long)w * s * (h2 + (long)h * s * c2);
        } else {
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
        y[opix * ldy + co] = x[((b * h + yi) * (long)w + xi) * ldx + ci];
    }
}

extern "C" int y2h_reorg_f16(const void *x, int ldx, void *y, int ldy, int batch, int h, int w, int c,
                             int stride, int reverse, y2h_stream s)
{
    if (!x || !y || batch <= 0 || h <= 0 || w <= 0 || c <= 0 || stride <= 0 || ldx < c) return Y2H_EINVAL;
    if (c % (stride * stride) != 0) return Y2H_EINVAL;
    if (!reverse && (h % stride != 0 || w % stride != 0)) return Y2H_EINVAL;
    const int oc = reverse ? c / (stride * stride) : c * stride * stride;
    if (ldy < oc) return Y2H_EINVAL;
    const long total = (long)batch * h * w * c;
    hipLaunchKernelGGL(reorg_f16_kernel, dim3(y2h_grid(total, 256)), dim3(256), 0, S(s), (const _Float16 *)x, ldx,
                       (_Float16 *)y, ldy, h, w, c, stride, reverse ? 1 : 0, total);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// [route] fallback copy, V halves per thread
template <int V>
__global__ __launch_bounds__(256) void copy_channels_f16_kernel(const _Float16 *__restrict__ src, int ld_src,
                                                                _Float16 *__restrict__ dst, int ld_dst, int c, long total)
{
    const int cv = c / V;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / cv;
        const int ch = (int)(i - p * cv) * V;
        if (V == 8) *(h8 *)&dst[p * ld_dst + ch] = *(const h8 *)&src[p * ld_src + ch];
        else dst[p * ld_dst + ch] = src[p * ld_src + ch];
    }
}

extern "C" int y2h_copy_channels_f16(const void *src, int ld_src, void *dst, int ld_dst, int c, long npix, y2h_stream s)
{
    if (c <= 0 || npix <= 0) return Y2H_OK;
    if (!src || !dst || ld_src < c || ld_dst < c) return Y2H_EINVAL;
    const bool v8 = (c % 8 == 0) && (ld_src % 8 == 0) && (ld_dst % 8 == 0) && (((uintptr_t)src | (uintptr_t)dst) % 16 == 0);
    if (v8) hipLaunchKernelGGL(copy_channels_f16_kernel<8>, dim3(y2h_grid(npix * (c / 8), 256)), dim3(256), 0, S(s),
                               (const _Float16 *)src, ld_src, (_Float16 *)dst, ld_dst, c, npix * (c / 8));
    else hipLaunchKernelGGL(copy_channels_f16_kernel<1>, dim3(y2h_grid(npix * c, 256)), dim3(256), 0, S(s),
                            (const _Float16 *)src, ld_src, (_Float16 *)dst, ld_dst, c, npix * c);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// avgpool_layer.c:40-54 on half input: fp32 sequential sum in pixel order, one divide, fp32 result
__global__ __launch_bounds__(256) void avgpool_f16_kernel(const _Float16 *__restrict__ x, int ldx, float *__restrict__ y,
                                                          int hw, int c, long total)
{
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int k = (int)(idx % c);
        const long b = idx / c;
        const _Float16 *src = x + b * hw * (long)ldx + k;
        float sum = 0.f;
        for (int i = 0; i < hw; ++i) sum += (float)src[(long)i * ldx];
        y[idx] = sum / hw;
    }
}

// The same with 16-byte loads: block = (image, 64 groups of 8 channels), a lane owns 8 channels, the block's four waves each
// sum a contiguous quarter of the pixels in pixel order and wave 0 adds the four partial sums in order.  (The scalar form
// above issues one 2-byte load per lane and pixel: 0.9 TB/s on the 50 MB of darknet19_448 b128, 1.6 % of that step.)  The
// fp16 mode has no reference arithmetic to mirror; on integer data both forms are exact.
typedef _Float16 avg_f16x8 __attribute__((ext_vector_type(8)));
typedef float avg_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void avgpool_f16_v8_kernel(const _Float16 *__restrict__ x, int ldx, float *__restrict__ y, int hw, int c)
{
    __shared__ float part[4][64][9];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c0 = (blockIdx.x * 64 + lane) * 8;
    const long b = blockIdx.y;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < c) {
        const int p0 = (int)((long)q * hw / 4), p1 = (int)((long)(q + 1) * hw / 4);
        const _Float16 *src = x + (b * hw + p0) * (long)ldx + c0;
#pragma unroll 7
        for (int p = p0; p < p1; ++p, src += ldx) {
            const avg_f16x8 v = *(const avg_f16x8 *)src;
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += (float)v[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) part[q][lane][k] = s[k];
    __syncthreads();
    if (q == 0 && c0 < c) {
        float r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = (((part[0][lane][k] + part[1][lane][k]) + part[2][lane][k]) + part[3][lane][k]) / hw;
        avg_f32x4 *dst = (avg_f32x4 *)(y + b * c + c0);
        dst[0] = avg_f32x4{r[0], r[1], r[2], r[3]};
        dst[1] = avg_f32x4{r[4], r[5], r[6], r[7]};
    }
}

extern "C" int y2h_avgpool_f16(const void *x, int ldx, float *y, int batch, int h, int w, int c, y2h_stream s)
{
    if (!x || !y || batch <= 0 || h <= 0 || w <= 0 || c <= 0 || ldx < c) return Y2H_EINVAL;
    const long total = (long)batch * c;
    if (c % 8 == 0 && ldx % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && h * w >= 4 && !getenv("Y2_AVGPOOL_SCALAR")) {
        hipLaunchKernelGGL(avgpool_f16_v8_kernel, dim3((unsigned)((c / 8 + 63) / 64), (unsigned)batch), dim3(256), 0, S(s),
                           (const _Float16 *)x, ldx, y, h * w, c);
        Y2H_LAUNCH_CHECK();
        return Y2H_OK;
    }
    hipLaunchKernelGGL(avgpool_f16_kernel, dim3(y2h_grid(total, 256)), dim3(256), 0, S(s), (const _Float16 *)x, ldx, y,
                       h * w, c, total);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// half NHWC [n][hw][ld] -> fp32 NCHW [n][c][hw] (what network_predict / y2_pull_layer_output hand to the host)
__global__ __launch_bounds__(256) void nhwc_f16_to_nchw_kernel(const _Float16 *__restrict__ src, int ld, float *__restrict__ dst,
                                                               int c, long hw)
{
    __shared__ float tile[32][65];
    const long p0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 32;
    const int n = blockIdx.z;
    const _Float16 *s = src + (long)n * hw * ld;
    float *d = dst + (long)n * c * hw;
    const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;
    for (int pp = py; pp < 64; pp += 8) {
        const long p = p0 + pp;
        const int ch = c0 + cx;
        tile[cx][pp] = (ch < c && p < hw) ? (float)s[p * ld + ch] : 0.f;
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int cc = ty; cc < 32; cc += 4) {
        const int ch = c0 + cc;
        const long p = p0 + tx;
        if (ch < c && p < hw) d[(long)ch * hw + p] = tile[cc][tx];
    }
}

extern "C" int y2h_nhwc_f16_to_nchw(const void *src, int ld, float *dst, int n, int c, int h, int w, y2h_stream s)
{
    if (!src || !dst || n <= 0 || c <= 0 || h <= 0 || w <= 0 || ld < c) return Y2H_EINVAL;
    const long hw = (long)h * w;
    dim3 grid((unsigned)((hw + 63) / 64), (unsigned)((c + 31) / 32), (unsigned)n);
    hipLaunchKernelGGL(nhwc_f16_to_nchw_kernel, grid, dim3(256), 0, S(s), (const _Float16 *)src, ld, dst, c, hw);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// network input for the fp16 first-layer kernel: fp32 NCHW (<= 4 planes) -> half [n][h+2][w+2][4], interior only;
// one thread per pixel reads its planes (coalesced along x) and writes one 8-byte pixel
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void nchw_to_nhwc4_halo_f16_kernel(const float *__restrict__ src, _Float16 *__restrict__ dst,
                                                                     int c, int h, int w, long total)
{
    const long hw = (long)h * w;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int x = (int)(idx % w);
        const int y = (int)((idx / w) % h);
        const long n = idx / hw;
        const float *s = src + n * c * hw + (long)y * w + x;
        h4 px;
#pragma unroll
        for (int k = 0; k < 4; ++k) px[k] = (k < c) ? (_Float16)s[k * hw] : (_Float16)0.f;
        *(h4 *)&dst[((n * (h + 2) + (y + 1)) * (long)(w + 2) + (x + 1)) * 4] = px;
    }
}

extern "C" int y2h_nchw_to_nhwc4_halo_f16(const float *src, void *dst, int n, int c, int h, int w, y2h_stream s)
{
    if (!src || !dst || n <= 0 || c <= 0 || c > 4 || h <= 0 || w <= 0 || ((uintptr_t)dst % 8) != 0) return Y2H_EINVAL;
    const long total = (long)n * h * w;
    hipLaunchKernelGGL(nchw_to_nhwc4_halo_f16_kernel, dim3(y2h_grid(total, 256)), dim3(256), 0, S(s), src, (_Float16 *)dst,
                       c, h, w, total);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// array conversions (weights are converted on the host; these serve the kernel-level tests)
__global__ __launch_bounds__(256) void f32_to_f16_kernel(const float *__restrict__ src, _Float16 *__restrict__ dst, long n)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = (_Float16)src[i];
}
__global__ __launch_bounds__(256) void f16_to_f32_kernel(const _Float16 *__restrict__ src, float *__restrict__ dst, long n)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = (float)src[i];
}
extern "C" int y2h_f32_to_f16(const float *src, void *dst, long n, y2h_stream s)
{
    if (!src || !dst || n <= 0) return Y2H_EINVAL;
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3(y2h_grid(n, 256)), dim3(256), 0, S(s), src, (_Float16 *)dst, n);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}
extern "C" int y2h_f16_to_f32(const void *src, float *dst, long n, y2h_stream s)
{
    if (!src || !dst || n <= 0) return Y2H_EINVAL;
    hipLaunchKernelGGL(f16_to_f32_kernel, dim3(y2h_grid(n, 256)), dim3(256), 0, S(s), (const _Float16 *)src, dst, n);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}
