// Device, memory, stream and layout entry points of the C-ABI HIP layer
// (include/y2_hip.h).  Replaces the reference's cuda.c wrapper
// (src_yolo2/cuda.c:12-160); unlike it, nothing here aborts -- errors come
// back as codes and the text is kept for y2h_last_error().
#include "y2_common.hpp"
#include <mutex>
#include <unordered_map>
#include <vector>

static thread_local char g_err[512] = "";
static char g_name[256] = "";

extern "C" void y2h_set_error_(const char *what, const char *detail)
{
    snprintf(g_err, sizeof g_err, "%s: %s", what, detail ? detail : "");
}

extern "C" const char *y2h_last_error(void) { return g_err; }

extern "C" int y2h_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int y2h_set_device(int dev) { Y2H_CHECK(hipSetDevice(dev)); return Y2H_OK; }
extern "C" int y2h_get_device(int *dev) { Y2H_CHECK(hipGetDevice(dev)); return Y2H_OK; }

extern "C" const char *y2h_device_name(void)
{
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return "";
    snprintf(g_name, sizeof g_name, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return g_name;
}

// "0000:c1:00.0" of device `dev` (dev < 0: the current device); "" when the runtime does not report it
extern "C" const char *y2h_device_pci_bus_id(int dev)
{
    static thread_local char bdf[32];
    bdf[0] = 0;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return bdf;
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, dev) != hipSuccess) bdf[0] = 0;
    return bdf;
}

// Debug aid (env Y2_GUARD=1, e.g. for a whole `pytest -m gpu` run): every device buffer gets a 4 KB canary in front of and
// behind it, filled with 0xA5 at allocation and checked when the buffer is freed (hipFree has waited for the device by
// then); a kernel that wrote outside its buffer is reported with the buffer's size and the first damaged offset.  Off, the
// two functions are plain hipMalloc / hipFree.
namespace {
constexpr size_t GUARD = 4096;
std::mutex g_guard_mu;
std::unordered_map<void *, size_t> g_guarded;         // user pointer -> user bytes
bool guard_on()
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("Y2_GUARD"); on = (e && atoi(e) != 0) ? 1 : 0; }
    return on == 1;
}
}

extern "C" int y2h_malloc(void **ptr, size_t bytes)
{
    if (!guard_on()) { Y2H_CHECK(hipMalloc(ptr, bytes ? bytes : 16)); return Y2H_OK; }
    const size_t user = ((bytes ? bytes : 16) + 255) & ~(size_t)255;
    unsigned char *base = nullptr;
    Y2H_CHECK(hipMalloc((void **)&base, user + 2 * GUARD));
    Y2H_CHECK(hipMemset(base, 0xA5, GUARD));
    Y2H_CHECK(hipMemset(base + GUARD + (bytes ? bytes : 16), 0xA5, user - (bytes ? bytes : 16) + GUARD));
    *ptr = base + GUARD;
    std::lock_guard<std::mutex> lk(g_guard_mu);
    g_guarded[*ptr] = bytes ? bytes : 16;
    return Y2H_OK;
}

extern "C" int y2h_free(void *ptr)
{
    if (!ptr) return Y2H_OK;
    if (!guard_on()) { Y2H_CHECK(hipFree(ptr)); return Y2H_OK; }
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> lk(g_guard_mu);
        auto it = g_guarded.find(ptr);
        if (it == g_guarded.end()) { Y2H_CHECK(hipFree(ptr)); return Y2H_OK; }       // allocated before the switch was read
        bytes = it->second;
        g_guarded.erase(it);
    }
    const size_t user = (bytes + 255) & ~(size_t)255;
    unsigned char *base = (unsigned char *)ptr - GUARD;
    std::vector<unsigned char> h(user + 2 * GUARD);
    Y2H_CHECK(hipDeviceSynchronize());
    Y2H_CHECK(hipMemcpy(h.data(), base, GUARD, hipMemcpyDeviceToHost));
    Y2H_CHECK(hipMemcpy(h.data() + GUARD + bytes, base + GUARD + bytes, user - bytes + GUARD, hipMemcpyDeviceToHost));
    long bad_front = -1, bad_back = -1;
    for (size_t i = 0; i < GUARD; ++i) if (h[GUARD - 1 - i] != 0xA5) { bad_front = (long)i + 1; break; }
    for (size_t i = GUARD + bytes; i < user + 2 * GUARD; ++i) if (h[i] != 0xA5) { bad_back = (long)(i - GUARD - bytes); break; }
    Y2H_CHECK(hipFree(base));
    if (bad_front >= 0 || bad_back >= 0) {
        fprintf(stderr, "Y2_GUARD: a kernel wrote outside a device buffer of %zu bytes: %ld bytes in front of it / %ld bytes past its end (-1 = intact)\n",
                bytes, bad_front, bad_back);
        y2h_set_error_("Y2_GUARD", "write outside a device buffer");
        return Y2H_EHIP;
    }
    return Y2H_OK;
}
extern "C" int y2h_host_alloc(void **ptr, size_t bytes) { Y2H_CHECK(hipHostMalloc(ptr, bytes ? bytes : 16, hipHostMallocDefault)); return Y2H_OK; }
extern "C" int y2h_host_free(void *ptr) { if (ptr) Y2H_CHECK(hipHostFree(ptr)); return Y2H_OK; }
extern "C" int y2h_host_register(void *ptr, size_t bytes) { Y2H_CHECK(hipHostRegister(ptr, bytes, hipHostRegisterDefault)); return Y2H_OK; }
extern "C" int y2h_host_unregister(void *ptr) { if (ptr) Y2H_CHECK(hipHostUnregister(ptr)); return Y2H_OK; }

extern "C" int y2h_memcpy_h2d(void *dst, const void *src, size_t bytes, y2h_stream s)
{
    if (bytes) Y2H_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, S(s)));
    return Y2H_OK;
}
extern "C" int y2h_memcpy_d2h(void *dst, const void *src, size_t bytes, y2h_stream s)
{
    if (bytes) Y2H_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, S(s)));
    return Y2H_OK;
}
extern "C" int y2h_memcpy_d2d(void *dst, const void *src, size_t bytes, y2h_stream s)
{
    if (bytes) Y2H_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, S(s)));
    return Y2H_OK;
}
extern "C" int y2h_memset(void *dst, int value, size_t bytes, y2h_stream s)
{
    if (bytes) Y2H_CHECK(hipMemsetAsync(dst, value, bytes, S(s)));
    return Y2H_OK;
}

extern "C" int y2h_stream_create(y2h_stream *s)
{
    hipStream_t st;
    Y2H_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *s = (y2h_stream)st;
    return Y2H_OK;
}
extern "C" int y2h_stream_destroy(y2h_stream s) { if (s) Y2H_CHECK(hipStreamDestroy(S(s))); return Y2H_OK; }
extern "C" int y2h_stream_sync(y2h_stream s) { Y2H_CHECK(hipStreamSynchronize(S(s))); return Y2H_OK; }
extern "C" int y2h_device_sync(void) { Y2H_CHECK(hipDeviceSynchronize()); return Y2H_OK; }

// ---- hipGraph: a launch-bound kernel sequence (batch-1 inference: ~30 launches of 5-30 us) recorded once, replayed
// with one call ----
extern "C" int y2h_graph_begin(y2h_stream s) { Y2H_CHECK(hipStreamBeginCapture(S(s), hipStreamCaptureModeThreadLocal)); return Y2H_OK; }
extern "C" int y2h_graph_end(y2h_stream s, y2h_graph *out)
{
    hipGraph_t g = nullptr;
    hipGraphExec_t ex = nullptr;
    Y2H_CHECK(hipStreamEndCapture(S(s), &g));
    hipError_t err = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    Y2H_CHECK(err);
    *out = (y2h_graph)ex;
    return Y2H_OK;
}
// abandon a capture after an error inside it (the stream leaves capture mode; nothing is kept)
extern "C" void y2h_graph_abort(y2h_stream s)
{
    hipGraph_t g = nullptr;
    if (hipStreamEndCapture(S(s), &g) == hipSuccess && g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
}
extern "C" int y2h_graph_launch(y2h_graph g, y2h_stream s) { Y2H_CHECK(hipGraphLaunch((hipGraphExec_t)g, S(s))); return Y2H_OK; }
extern "C" int y2h_graph_destroy(y2h_graph g) { if (g) Y2H_CHECK(hipGraphExecDestroy((hipGraphExec_t)g)); return Y2H_OK; }

extern "C" int y2h_event_create(y2h_event *e)
{
    hipEvent_t ev;
    Y2H_CHECK(hipEventCreate(&ev));
    *e = (y2h_event)ev;
    return Y2H_OK;
}
extern "C" int y2h_event_destroy(y2h_event e) { if (e) Y2H_CHECK(hipEventDestroy((hipEvent_t)e)); return Y2H_OK; }
extern "C" int y2h_event_record(y2h_event e, y2h_stream s) { Y2H_CHECK(hipEventRecord((hipEvent_t)e, S(s))); return Y2H_OK; }
extern "C" int y2h_event_sync(y2h_event e) { Y2H_CHECK(hipEventSynchronize((hipEvent_t)e)); return Y2H_OK; }
extern "C" int y2h_stream_wait_event(y2h_stream s, y2h_event e) { Y2H_CHECK(hipStreamWaitEvent(S(s), (hipEvent_t)e, 0)); return Y2H_OK; }
extern "C" int y2h_event_elapsed_ms(y2h_event start, y2h_event stop, float *ms)
{
    Y2H_CHECK(hipEventSynchronize((hipEvent_t)stop));
    Y2H_CHECK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return Y2H_OK;
}

// ---------------------------------------------------------------------------
// layout kernels
// ---------------------------------------------------------------------------

// NCHW -> NHWC through an LDS tile so both sides stay coalesced:
// a block moves 64 pixels x 32 channels.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                           int c, long hw, int ld)
{
    __shared__ float tile[32][65];
    const long p0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 32;
    const int n = blockIdx.z;
    const float *s = src + (long)n * c * hw;
    float *d = dst + (long)n * hw * ld;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
    for (int cc = ty; cc < 32; cc += 4) {
        const int ch = c0 + cc;
        const long p = p0 + tx;
        tile[cc][tx] = (ch < c && p < hw) ? s[(long)ch * hw + p] : 0.f;
    }
    __syncthreads();
    const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;   // 32 x 8
    for (int pp = py; pp < 64; pp += 8) {
        const long p = p0 + pp;
        const int ch = c0 + cx;
        if (ch < c && p < hw) d[p * ld + ch] = tile[cx][pp];
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float *__restrict__ src, int ld, float *__restrict__ dst,
                                                           int c, long hw)
{
    __shared__ float tile[32][65];
    const long p0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 32;
    const int n = blockIdx.z;
    const float *s = src + (long)n * hw * ld;
    float *d = dst + (long)n * c * hw;
    const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;
    for (int pp = py; pp < 64; pp += 8) {
        const long p = p0 + pp;
        const int ch = c0 + cx;
        tile[cx][pp] = (ch < c && p < hw) ? s[p * ld + ch] : 0.f;
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int cc = ty; cc < 32; cc += 4) {
        const int ch = c0 + cc;
        const long p = p0 + tx;
        if (ch < c && p < hw) d[(long)ch * hw + p] = tile[cc][tx];
    }
}

extern "C" int y2h_nchw_to_nhwc(const float *src, float *dst, int n, int c, int h, int w, int ld, y2h_stream s)
{
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0 || ld < c) return Y2H_EINVAL;
    const long hw = (long)h * w;
    dim3 grid((unsigned)((hw + 63) / 64), (unsigned)((c + 31) / 32), (unsigned)n);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, S(s), src, dst, c, hw, ld);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

extern "C" int y2h_nhwc_to_nchw(const float *src, int ld, float *dst, int n, int c, int h, int w, y2h_stream s)
{
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0 || ld < c) return Y2H_EINVAL;
    const long hw = (long)h * w;
    dim3 grid((unsigned)((hw + 63) / 64), (unsigned)((c + 31) / 32), (unsigned)n);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, S(s), src, ld, dst, c, hw);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// few-channel NCHW -> NHWC with a `halo`-pixel border: one thread per pixel reads its c planes
// (coalesced along x) and writes c contiguous floats into the interior of the padded image
__global__ __launch_bounds__(256) void nchw_to_nhwc_halo_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                                int c, int h, int w, int ld, int halo, long total)
{
    const long hw = (long)h * w;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int x = (int)(idx % w);
        const int y = (int)((idx / w) % h);
        const long n = idx / hw;
        const float *s = src + n * c * hw + (long)y * w + x;
        float *d = dst + ((n * (h + 2 * halo) + (y + halo)) * (long)(w + 2 * halo) + (x + halo)) * ld;
        for (int k = 0; k < c; ++k) d[k] = s[k * hw];
    }
}

extern "C" int y2h_nchw_to_nhwc_halo(const float *src, float *dst, int n, int c, int h, int w, int ld, int halo, y2h_stream s)
{
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0 || ld < c || halo < 0) return Y2H_EINVAL;
    const long total = (long)n * h * w;
    hipLaunchKernelGGL(nchw_to_nhwc_halo_kernel, dim3(y2h_grid(total, 256)), dim3(256), 0, S(s), src, dst, c, h, w, ld, halo, total);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

__global__ __launch_bounds__(256) void copy_channels_kernel(const float *__restrict__ src, int ld_src,
                                                            float *__restrict__ dst, int ld_dst, int c, long total)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / c;
        const int ch = (int)(i - p * c);
        dst[p * ld_dst + ch] = src[p * ld_src + ch];
    }
}

__global__ __launch_bounds__(256) void copy_channels4_kernel(const float4 *__restrict__ src, int ld_src4,
                                                             float4 *__restrict__ dst, int ld_dst4, int c4, long total)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / c4;
        const int ch = (int)(i - p * c4);
        dst[p * ld_dst4 + ch] = src[p * ld_src4 + ch];
    }
}

extern "C" int y2h_copy_channels(const float *src, int ld_src, float *dst, int ld_dst, int c, long npix, y2h_stream s)
{
    if (c <= 0 || npix <= 0) return Y2H_OK;
    if (ld_src < c || ld_dst < c) return Y2H_EINVAL;
    const bool v4 = (c % 4 == 0) && (ld_src % 4 == 0) && (ld_dst % 4 == 0) &&
                    (((uintptr_t)src | (uintptr_t)dst) % 16 == 0);
    if (v4) {
        const long total = npix * (c / 4);
        hipLaunchKernelGGL(copy_channels4_kernel, dim3(y2h_grid(total, 256)), dim3(256), 0, S(s),
                           (const float4 *)src, ld_src / 4, (float4 *)dst, ld_dst / 4, c / 4, total);
    } else {
        const long total = npix * c;
        hipLaunchKernelGGL(copy_channels_kernel, dim3(y2h_grid(total, 256)), dim3(256), 0, S(s),
                           src, ld_src, dst, ld_dst, c, total);
    }
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// ---------------------------------------------------------------------------
// Clock probe: what clock does THIS device hold under a full-chip fp32 matrix load?  Boxes of the pool differ by several
// per cent in every number (power cap, silicon); the benchmark line carries this figure so that two lines can be compared.
// One workgroup per CU slot, four waves each issuing `iters` dependent-free v_mfma_f32_32x32x2_f32 back to back (the
// instruction of the dominant kernel); every wave stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its
// loop.  Returns the median over waves of cycles / (ticks * 10 ns), in GHz.  Diagnostic only: nothing reads the values
// on the device, no output of the engine depends on them.
// ---------------------------------------------------------------------------
typedef float probe_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void clock_probe_kernel(unsigned long long *out, int iters, float seed)
{
    probe_f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = seed * r; acc1[r] = seed + r; }
    const float a = seed + threadIdx.x * 1e-3f, b = 1.0f - 1e-6f * threadIdx.x;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sink = 0.f;
    for (int r = 0; r < 16; ++r) sink += acc0[r] + acc1[r];
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) {
        out[2 * wave] = c1 - c0;
        out[2 * wave + 1] = (r1 - r0) | (sink == 12345.678f ? 1ull << 62 : 0ull);      // keeps the loop alive
    }
}

extern "C" int y2h_clock_probe(int iters, float *ghz, y2h_stream s)
{
    if (!ghz || iters <= 0) return Y2H_EINVAL;
    const int blocks = 256, waves = blocks * 4;
    unsigned long long *d = nullptr;
    Y2H_CHECK(hipMalloc((void **)&d, (size_t)waves * 2 * sizeof(unsigned long long)));
    // the clock governor needs a sustained load to settle (a single 0.5 ms launch reads 2.15 GHz from idle and 2.41 GHz right
    // behind a benchmark loop on the same box): ~0.6 s of back-to-back launches, the LAST one is the measurement
    const double launch_ms = (double)iters * 2.0 * 64.0 / 2.2e6;             // two MFMAs of 64 cycles per iteration at ~2.2 GHz
    int reps = (int)(600.0 / (launch_ms > 0.01 ? launch_ms : 0.01));
    if (reps < 2) reps = 2;
    if (reps > 5000) reps = 5000;
    for (int rep = 0; rep < reps; ++rep) {
        hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(256), 0, S(s), d, iters, 0.5f);
        if (hipGetLastError() != hipSuccess) { (void)hipFree(d); return Y2H_EHIP; }
    }
    std::vector<unsigned long long> h((size_t)waves * 2);
    hipError_t e = hipStreamSynchronize(S(s));
    if (e == hipSuccess) e = hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) { y2h_set_error_("clock probe", hipGetErrorString(e)); return Y2H_EHIP; }
    std::vector<double> g;
    for (int w = 0; w < waves; ++w) {
        const double ticks = (double)(h[2 * w + 1] & ((1ull << 62) - 1));
        if (ticks > 0) g.push_back((double)h[2 * w] / (ticks * 10.0));      // cycles per ns = GHz
    }
    if (g.empty()) return Y2H_EHIP;
    std::sort(g.begin(), g.end());
    *ghz = (float)g[g.size() / 2];
    return Y2H_OK;
}
