// y2_image.hip -- frame ingest on the device: what every caller does
// immediately before network_predict.  Reference behaviour restated:
//   * 8-bit interleaved frame -> float planes in [0,1]       yolo_v2_class.hpp:94-113 (ipl_to_image),
//                                                            image.c:2045-2067 (load_image_stb: (float)v/255.)
//   * swap planes 0 and 2 (BGR -> RGB)                        yolo_v2_class.hpp:133-141, image.c:1181
//   * fill_image / embed_image / letterbox_image              image.c:1601, :1087, :1607-1645
// All of it is HBM-bound byte shuffling: one read + one write per element, 128-bit stores where
// the layout allows.  The arithmetic (a single double division rounded to fp32, copies) is
// bit-identical to the C code.
#include "y2_common.hpp"

// dst[k][y][x] = (float)( src[y*step + x*c + sk] / 255. ), sk = k with planes 0 and 2 exchanged when swap_rb.
// One thread per output pixel; the c source bytes of a pixel are read once and fanned out to the planes.
__global__ __launch_bounds__(256) void u8_to_planes_kernel(const unsigned char *__restrict__ src, int h, int w, int c,
                                                           long step, long frame_bytes, int planes, int swap_rb,
                                                           float *__restrict__ dst)
{
    const long hw = (long)h * w;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= hw) return;
    const int b = blockIdx.y;
    const int y = (int)(idx / w), x = (int)(idx - (long)y * w);
    const unsigned char *p = src + (size_t)b * frame_bytes + (size_t)y * step + (size_t)x * c;
    float *o = dst + (size_t)b * planes * hw + idx;
    for (int k = 0; k < planes; ++k) {
        int sk = k;
        if (swap_rb && c >= 3) sk = (k == 0) ? 2 : (k == 2 ? 0 : k);
        o[(size_t)k * hw] = (float)((double)p[sk] / 255.);
    }
}

extern "C" int y2h_u8_to_planes(const unsigned char *src, int batch, int h, int w, int c, long step, long frame_bytes,
                                int planes, int swap_rb, float *dst, y2h_stream s)
{
    if (!src || !dst || batch <= 0 || h <= 0 || w <= 0 || c <= 0 || planes <= 0 || planes > c || step < (long)w * c ||
        frame_bytes < step * h) return Y2H_EINVAL;
    const long hw = (long)h * w;
    hipLaunchKernelGGL(u8_to_planes_kernel, dim3((unsigned)((hw + 255) / 256), (unsigned)batch), dim3(256), 0, S(s),
                       src, h, w, c, step, frame_bytes, planes, swap_rb, dst);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

__global__ __launch_bounds__(256) void fill_kernel(float *__restrict__ dst, long n, float v)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = v;
}

extern "C" int y2h_fill(float *dst, long n, float v, y2h_stream s)
{
    if (!dst || n <= 0) return Y2H_EINVAL;
    hipLaunchKernelGGL(fill_kernel, dim3(y2h_grid(n, 256)), dim3(256), 0, S(s), dst, n, v);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// embed_image (image.c:1087) with set_pixel's bounds test (image.c:2121-2123: writes outside dest are dropped)
__global__ __launch_bounds__(256) void embed_kernel(const float *__restrict__ src, int c, int sh, int sw,
                                                    float *__restrict__ dst, int dh, int dw, int dx, int dy)
{
    const long total = (long)c * sh * sw;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % sw);
        const int y = (int)((i / sw) % sh);
        const int k = (int)(i / ((long)sw * sh));
        const int X = dx + x, Y = dy + y;
        if (X < 0 || Y < 0 || X >= dw || Y >= dh) continue;
        dst[((size_t)k * dh + Y) * dw + X] = src[i];
    }
}

extern "C" int y2h_embed_chw(const float *src, int c, int sh, int sw, float *dst, int dh, int dw, int dx, int dy,
                             y2h_stream s)
{
    if (!src || !dst || c <= 0 || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0) return Y2H_EINVAL;
    hipLaunchKernelGGL(embed_kernel, dim3(y2h_grid((long)c * sh * sw, 256)), dim3(256), 0, S(s), src, c, sh, sw, dst, dh,
                       dw, dx, dy);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// letterbox geometry (image.c:1607-1622): keep the aspect ratio, integer arithmetic as in the reference
extern "C" void y2h_letterbox_dims(int iw, int ih, int w, int h, int *new_w, int *new_h)
{
    int nw = iw, nh = ih;
    if (((float)w / iw) < ((float)h / ih)) { nw = w; nh = (ih * w) / iw; }
    else { nh = h; nw = (iw * h) / ih; }
    *new_w = nw; *new_h = nh;
}

// letterbox_image (image.c:1624): resize to (new_w,new_h), fill the box with .5, embed centred.
// tmp: c*ih*new_w floats (resize pass 1) + c*new_h*new_w floats (the resized image).
extern "C" int y2h_letterbox_chw(const float *src, int c, int ih, int iw, float *tmp, float *dst, int h, int w, y2h_stream s)
{
    if (!src || !tmp || !dst || c <= 0 || ih <= 0 || iw <= 0 || h <= 0 || w <= 0) return Y2H_EINVAL;
    int nw, nh;
    y2h_letterbox_dims(iw, ih, w, h, &nw, &nh);
    if (nw <= 0 || nh <= 0) return Y2H_EINVAL;
    float *resized = tmp + (size_t)c * ih * nw;
    int rc = y2h_resize_chw(src, c, ih, iw, tmp, resized, nh, nw, s);
    if (rc) return rc;
    rc = y2h_fill(dst, (long)c * h * w, .5f, s);
    if (rc) return rc;
    return y2h_embed_chw(resized, c, nh, nw, dst, h, w, (w - nw) / 2, (h - nh) / 2, s);
}

// utils.c:420-432 mean_arrays on device buffers: avg = 0; for j: avg += frame j; avg /= n  (fp32, frame order)
__global__ __launch_bounds__(256) void mean_frames_kernel(const float *__restrict__ frames, int n, long els, float *__restrict__ avg)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < els; i += (long)gridDim.x * 256) {
        float a = 0.f;
        for (int j = 0; j < n; ++j) a += frames[(size_t)j * els + i];
        avg[i] = a / n;
    }
}

extern "C" int y2h_mean_frames(const float *frames, int n, long els, float *avg, y2h_stream s)
{
    if (!frames || !avg || n <= 0 || els <= 0) return Y2H_EINVAL;
    hipLaunchKernelGGL(mean_frames_kernel, dim3(y2h_grid(els, 256)), dim3(256), 0, S(s), frames, n, els, avg);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}
