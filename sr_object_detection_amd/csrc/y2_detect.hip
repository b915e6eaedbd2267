// Box decode, non-maximum suppression and detection hand-off on the device
// (include/y2_hip.h).  In the reference these run on the CPU in both builds
// (src_yolo2/region_layer.c:328-379, src_yolo2/box.c:239-298,
// src_yolo2/yolo_v2_class.cpp:221-238); here they are kernels so a frame batch
// never leaves HBM until the compact detection records do.
//
// Every float expression mirrors the reference's C expression (same operand
// types, same order) and the file is compiled with -ffp-contract=off, so given
// the same region-layer tensor the boxes, probabilities and NMS decisions are
// bit-identical to the CPU path.
#include "y2_common.hpp"
#include <float.h>
#include <stdlib.h>

// ---------------------------------------------------------------------------
// get_region_boxes (region_layer.c:328-379) + get_region_box (:73-85, DOABS=1)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float logistic_f(float x) { return (float)(1. / (1. + exp(-(double)x))); }

struct DecodeK {
    int w, h, num, classes, img_w, img_h;
    float thresh;
    int only_objectness, classfix;
    const float *anchors;
    const int *parent;
    const int *map;
    float *pred;
    float *boxes;
    float *probs;
    long nboxes;       // batch * w * h * num
    int tree_seq;      // 1: decode_boxes_kernel also walks the tree sequentially (fallback)
};

__global__ __launch_bounds__(64) void decode_boxes_kernel(DecodeK d)
{
    const long gi = (long)blockIdx.x * 64 + threadIdx.x;
    if (gi >= d.nboxes) return;
    const int total = d.w * d.h * d.num;
    const int index = (int)(gi % total);             // box index inside its image
    const int n = index % d.num;
    const int cell = index / d.num;
    const int row = cell / d.w, col = cell % d.w;
    const int size = d.classes + 5;
    float *x = d.pred + gi * size;
    float scale = x[4];
    if (d.classfix == -1 && scale < .5) scale = 0;
    float bx = (col + logistic_f(x[0])) / d.w;
    float by = (row + logistic_f(x[1])) / d.h;
    float bw = (float)(exp((double)x[2]) * d.anchors[2 * n] / d.w);
    float bh = (float)(exp((double)x[3]) * d.anchors[2 * n + 1] / d.h);
    bx *= d.img_w; by *= d.img_h; bw *= d.img_w; bh *= d.img_h;
    float *bo = d.boxes + gi * 4;
    bo[0] = bx; bo[1] = by; bo[2] = bw; bo[3] = bh;
    float *pr = d.probs + gi * d.classes;
    if (d.parent && d.tree_seq) {
        // tree.c:37-44 hierarchy_predictions, sequential form (any node order), in place
        float *p = x + 5;
        for (int j = 0; j < d.classes; ++j) {
            const int par = d.parent[j];
            if (par >= 0) p[j] *= p[par];
        }
        if (d.map) {
            for (int j = 0; j < 200; ++j) {
                const float prob = scale * p[d.map[j]];
                pr[j] = (prob > d.thresh) ? prob : 0;
            }
        } else {
            int found = 0;
            for (int j = d.classes - 1; j >= 0; --j) {
                if (!found && p[j] > .5) found = 1;
                else p[j] = 0;
                const float prob = p[j];
                pr[j] = (scale > d.thresh) ? prob : 0;
            }
        }
        if (d.only_objectness) pr[0] = scale;
    }
}

__global__ __launch_bounds__(256) void decode_probs_kernel(DecodeK d)
{
    const long total = d.nboxes * d.classes;
    const int size = d.classes + 5;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int j = (int)(idx % d.classes);
        const long gi = idx / d.classes;
        const float *x = d.pred + gi * size;
        float scale = x[4];
        if (d.classfix == -1 && scale < .5) scale = 0;
        const float prob = scale * x[5 + j];
        float v = (prob > d.thresh) ? prob : 0;
        if (d.only_objectness && j == 0) v = scale;
        d.probs[idx] = v;
    }
}

// Tree head of one box per workgroup (yolo9000: 9418 classes).  hierarchy_predictions (tree.c:37-44)
// multiplies every node by its parent's FINAL value when parents precede their children in the file
// (checked on the host; otherwise decode_boxes_kernel's sequential walk is used), so nodes of one
// depth level are independent: the class row is staged in LDS and walked level by level.  The
// products are the same fp32 multiplications, hence bit-identical to the sequential loop.  Then, as
// region_layer.c:351-367: with a map, probs[j] = obj * p[map[j]] thresholded for j < 200; without,
// only the deepest class whose probability exceeds .5 (the LAST such index) keeps its value and
// every other class score is zeroed, in the probs row and in the prediction row itself.
struct TreeK {
    const int *order;        // node ids sorted by depth
    const int *level_off;    // [levels + 1] offsets into order
    int levels;
};

__global__ __launch_bounds__(256) void decode_tree_kernel(DecodeK d, TreeK tk)
{
    extern __shared__ __attribute__((aligned(16))) float row[];      // [classes]
    __shared__ int s_best;
    const long gi = blockIdx.x;
    const int t = threadIdx.x;
    const int size = d.classes + 5;
    float *x = d.pred + gi * size;
    float *p = x + 5;
    float *pr = d.probs + gi * d.classes;
    float scale = x[4];
    if (d.classfix == -1 && scale < .5) scale = 0;
    for (int j = t; j < d.classes; j += 256) row[j] = p[j];
    if (t == 0) s_best = -1;
    __syncthreads();
    for (int lv = 1; lv < tk.levels; ++lv) {             // level 0 = roots (parent < 0): unchanged
        const int b = tk.level_off[lv], e = tk.level_off[lv + 1];
        for (int i = b + t; i < e; i += 256) {
            const int j = tk.order[i];
            row[j] *= row[d.parent[j]];
        }
        __syncthreads();
    }
    if (d.map) {
        for (int j = t; j < d.classes; j += 256) p[j] = row[j];
        for (int j = t; j < 200; j += 256) {
            const float prob = scale * row[d.map[j]];
            pr[j] = (prob > d.thresh) ? prob : 0;
        }
    } else {
        int best = -1;
        for (int j = t; j < d.classes; j += 256) if (row[j] > .5) best = j;   // ascending: keeps the last
        if (best >= 0) atomicMax(&s_best, best);
        __syncthreads();
        best = s_best;
        for (int j = t; j < d.classes; j += 256) {
            const float v = (j == best) ? row[j] : 0.f;
            p[j] = v;
            pr[j] = (scale > d.thresh) ? v : 0;
        }
    }
    if (d.only_objectness && t == 0) { __threadfence_block(); pr[0] = scale; }
}

extern "C" int y2h_region_boxes(const y2h_decode *q, y2h_stream s)
{
    if (!q || !q->pred || !q->boxes || !q->probs || !q->anchors) return Y2H_EINVAL;
    if (q->batch <= 0 || q->w <= 0 || q->h <= 0 || q->num <= 0 || q->classes <= 0) return Y2H_EINVAL;
    if (q->map && (!q->tree_parent || q->classes < 200)) return Y2H_EINVAL;
    DecodeK d;
    d.w = q->w; d.h = q->h; d.num = q->num; d.classes = q->classes; d.img_w = q->img_w; d.img_h = q->img_h;
    d.thresh = q->thresh; d.only_objectness = q->only_objectness; d.classfix = q->classfix;
    d.anchors = q->anchors; d.parent = q->tree_parent; d.map = q->map;
    d.pred = q->pred; d.boxes = q->boxes; d.probs = q->probs;
    d.nboxes = (long)q->batch * q->w * q->h * q->num;
    const bool level_tree = d.parent && q->tree_order && q->tree_level_off && q->tree_levels > 0 &&
                            (size_t)q->classes * sizeof(float) <= 150 * 1024;
    d.tree_seq = (d.parent && !level_tree) ? 1 : 0;
    hipLaunchKernelGGL(decode_boxes_kernel, dim3((unsigned)((d.nboxes + 63) / 64)), dim3(64), 0, S(s), d);
    Y2H_LAUNCH_CHECK();
    if (level_tree) {
        TreeK tk;
        tk.order = q->tree_order; tk.level_off = q->tree_level_off; tk.levels = q->tree_levels;
        const size_t lds = (size_t)q->classes * sizeof(float);
        static bool attr_set[16] = {false};
        int dev = 0;
        Y2H_CHECK(hipGetDevice(&dev));
        if (dev < 0 || dev >= 16 || !attr_set[dev]) {
            Y2H_CHECK(hipFuncSetAttribute((const void *)decode_tree_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            if (dev >= 0 && dev < 16) attr_set[dev] = true;
        }
        hipLaunchKernelGGL(decode_tree_kernel, dim3((unsigned)d.nboxes), dim3(256), lds, S(s), d, tk);
        Y2H_LAUNCH_CHECK();
    }
    if (!d.parent) {
        hipLaunchKernelGGL(decode_probs_kernel, dim3(y2h_grid(d.nboxes * d.classes, 256)), dim3(256), 0, S(s), d);
        Y2H_LAUNCH_CHECK();
    }
    return Y2H_OK;
}

// ---------------------------------------------------------------------------
// box_iou (box.c:67-97): no guard for 0/0, exactly as the reference
// ---------------------------------------------------------------------------
__device__ __forceinline__ float overlap1(float x1, float w1, float x2, float w2)
{
    const float l1 = x1 - w1 / 2, l2 = x2 - w2 / 2;
    const float left = l1 > l2 ? l1 : l2;
    const float r1 = x1 + w1 / 2, r2 = x2 + w2 / 2;
    const float right = r1 < r2 ? r1 : r2;
    return right - left;
}
__device__ __forceinline__ float box_iou_f(float4 a, float4 b)
{
    const float w = overlap1(a.x, a.z, b.x, b.z);
    const float h = overlap1(a.y, a.w, b.y, b.w);
    const float inter = (w < 0 || h < 0) ? 0 : w * h;
    const float uni = a.z * a.w + b.z * b.w - inter;
    return inter / uni;
}

// ---------------------------------------------------------------------------
// do_nms_sort (box.c:249-277): per class, sort by score descending, then greedy
// suppression.  One workgroup per (image, class):
//   1. gather the non-zero scores of the class into LDS as 64-bit keys
//      (score bits << 32 | ~box index): descending key order = descending score,
//      ties by ascending box index (the order a stable sort gives from the
//      reference's initial ascending array);
//   2. bitonic sort the keys in LDS;
//   3. walk the sorted list: a live entry kills every later entry whose IoU with
//      it exceeds thresh (all lanes test later entries in parallel);
//   4. zero the killed scores in the probs array.
// Zero scores never act and zeroing them again is a no-op, so restricting the
// sort to non-zero scores is exactly the reference's result.
// ---------------------------------------------------------------------------
//
// Ties: the reference sorts ONE array again and again (box.c:252-264), so boxes
// with equal score in class k keep the order the class k-1 sort left, i.e. the
// order is lexicographic over (p_k, p_{k-1}, ..., p_0) descending, then box
// index ascending (with a stable qsort, as glibc's is).  Each class's scores
// are only changed by that class's own suppression, after its sort, so the
// tie-break reads the ORIGINAL scores: `pin` is never written here and the
// survivors go to `pout` (a copy of pin made by the caller).  Runs of equal
// keys are rare; each is re-ordered by one lane after the bitonic sort.
__device__ __forceinline__ bool tie_before(const float *pin, int stride, int k, unsigned ia, unsigned ib)
{
    for (int c = k - 1; c >= 0; --c) {
        const float pa = pin[(size_t)ia * stride + c], pb = pin[(size_t)ib * stride + c];
        if (pa > pb) return true;
        if (pa < pb) return false;
    }
    return ia < ib;
}

// non-zero scores per (image, class): classes with fewer than two candidates have nothing to suppress, and
// with thousands of classes (yolo9000) almost all of them are empty -- their workgroups leave at once
__global__ __launch_bounds__(256) void class_count_kernel(const float *__restrict__ probs, int *__restrict__ counts,
                                                          int total, int classes, int stride, long n)
{
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const int k = (int)(idx % classes);
        const long row = idx / classes;                  // b * total + i
        if (probs[row * stride + k] != 0) atomicAdd(&counts[(row / total) * classes + k], 1);
    }
}

#define NMS_LDS_BOXES 1024
__global__ __launch_bounds__(256) void nms_sort_kernel(const float *__restrict__ boxes, const float *__restrict__ pin_all,
                                                       float *__restrict__ pout_all, const int *__restrict__ class_counts,
                                                       int total, int classes, int stride, float thresh, int cap, int lds_boxes,
                                                       int *__restrict__ reset_counts)
{
    // (the three-launch chain, y2h_detect_chain: this workgroup is the only reader of its count; it leaves the word zero for the
    // next frame's decode kernel to add to -- every lane reads before lane 0 writes)
    const int my_count = class_counts[blockIdx.x];
    if (reset_counts) {
        __syncthreads();
        if (threadIdx.x == 0) reset_counts[blockIdx.x] = 0;
    }
    if (my_count < 2) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char nms_smem[];
    unsigned long long *keys = (unsigned long long *)nms_smem;          // [cap]
    unsigned char *dead = (unsigned char *)(keys + cap);                // [cap]
    __shared__ int s_count;
    float4 *s_box = (float4 *)(nms_smem + (((size_t)cap * 9 + 15) & ~(size_t)15));      // [lds_boxes], behind keys and flags

    const int k = blockIdx.x % classes;
    const int b = blockIdx.x / classes;
    const float *bx = boxes + (size_t)b * total * 4;
    const float *pr = pin_all + (size_t)b * total * stride;
    float *pw = pout_all + (size_t)b * total * stride;
    const int t = threadIdx.x;

    if (t == 0) s_count = 0;
    __syncthreads();
    for (int i = t; i < total; i += 256) {
        const float p = pr[(size_t)i * stride + k];
        if (p != 0) {
            const int slot = atomicAdd(&s_count, 1);
            keys[slot] = ((unsigned long long)__float_as_uint(p) << 32) | (unsigned)(~(unsigned)i);
        }
    }
    __syncthreads();
    const int n = s_count;
    if (n < 2) return;                       // nothing can be suppressed
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int i = n + t; i < np2; i += 256) keys[i] = 0ull;
    for (int i = t; i < np2; i += 256) dead[i] = 0;
    __syncthreads();
    // bitonic sort, descending
    for (int kk = 2; kk <= np2; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = t; i < np2; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], c = keys[ixj];
                    const bool desc = ((i & kk) == 0);
                    if (desc ? (a < c) : (a > c)) { keys[i] = c; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    // re-order runs of equal scores the way the reference's repeated stable sort does
    for (int i = t; i < n - 1; i += 256) {
        const unsigned pi = (unsigned)(keys[i] >> 32);
        const bool starts = (i == 0 || (unsigned)(keys[i - 1] >> 32) != pi) && (unsigned)(keys[i + 1] >> 32) == pi;
        if (starts && k > 0) {
            int e = i + 1;
            while (e < n && (unsigned)(keys[e] >> 32) == pi) ++e;
            for (int a = i + 1; a < e; ++a) {                 // insertion sort of the run [i, e)
                const unsigned long long ka = keys[a];
                const unsigned ia = ~(unsigned)(ka & 0xffffffffull);
                int j = a - 1;
                while (j >= i && tie_before(pr, stride, k, ia, ~(unsigned)(keys[j] & 0xffffffffull))) {
                    keys[j + 1] = keys[j];
                    --j;
                }
                keys[j + 1] = ka;
            }
        }
    }
    __syncthreads();
    // greedy suppression in sorted order.  The walk is n - 1 dependent rounds; with the boxes fetched from global memory
    // inside each round a round costs a memory latency (62 us per 32 frames of yolo.cfg 608 for ~30 candidates of the
    // busiest class), so up to NMS_LDS_BOXES candidates are gathered into LDS once, in sorted order.
    if (n <= lds_boxes) {
        for (int j = t; j < n; j += 256) s_box[j] = *(const float4 *)(bx + (size_t)(~(unsigned)(keys[j] & 0xffffffffull)) * 4);
        __syncthreads();
        for (int i = 0; i < n - 1; ++i) {
            if (!dead[i]) {
                const float4 a = s_box[i];
                for (int j = i + 1 + t; j < n; j += 256)
                    if (box_iou_f(a, s_box[j]) > thresh) dead[j] = 1;
            }
            __syncthreads();
        }
    } else
    for (int i = 0; i < n - 1; ++i) {
        if (!dead[i]) {                      // uniform: every lane reads the same LDS byte
            const unsigned ia = ~(unsigned)(keys[i] & 0xffffffffull);
            const float4 a = *(const float4 *)(bx + (size_t)ia * 4);
            for (int j = i + 1 + t; j < n; j += 256) {
                const unsigned ib = ~(unsigned)(keys[j] & 0xffffffffull);
                const float4 c = *(const float4 *)(bx + (size_t)ib * 4);
                if (box_iou_f(a, c) > thresh) dead[j] = 1;
            }
        }
        __syncthreads();
    }
    for (int j = t; j < n; j += 256)
        if (dead[j]) {
            const unsigned ib = ~(unsigned)(keys[j] & 0xffffffffull);
            pw[(size_t)ib * stride + k] = 0;
        }
}

extern "C" int y2h_nms_sort(const float *boxes, const float *probs_in, float *probs, int batch, int total, int classes,
                            int stride, float thresh, int *class_counts, y2h_stream s)
{
    if (!boxes || !probs || !probs_in || !class_counts || probs == probs_in || batch <= 0 || total <= 0 || classes <= 0 ||
        stride < classes)
        return Y2H_EINVAL;
    if (total > 16384) return Y2H_EINVAL;            // LDS holds every candidate of one class
    int cap = 1;
    while (cap < total) cap <<= 1;
    const int lds_boxes = cap <= 8192 ? NMS_LDS_BOXES : 0;           // 16384 candidates fill the LDS by themselves
    const size_t lds = (((size_t)cap * 9 + 15) & ~(size_t)15) + (size_t)lds_boxes * 16;
    static bool attr_set[16] = {false};
    int dev = 0;
    Y2H_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
        Y2H_CHECK(hipFuncSetAttribute((const void *)nms_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 9));
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    Y2H_CHECK(hipMemsetAsync(class_counts, 0, (size_t)batch * classes * sizeof(int), S(s)));
    const long nel = (long)batch * total * classes;
    hipLaunchKernelGGL(class_count_kernel, dim3(y2h_grid(nel, 256, 256 * 32)), dim3(256), 0, S(s),
                       probs_in, class_counts, total, classes, stride, nel);
    Y2H_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_sort_kernel, dim3((unsigned)(batch * classes)), dim3(256), lds, S(s),
                       boxes, probs_in, probs, class_counts, total, classes, stride, thresh, cap, lds_boxes, (int *)nullptr);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// ---------------------------------------------------------------------------
// do_nms (box.c:279-298), the class-agnostic variant demo.c uses.  Sequential
// in i by construction; one workgroup per image.  For a fixed i the inner j
// loop is independent per class, so lanes take classes and walk j in order.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nms_plain_kernel(const float *__restrict__ boxes, float *__restrict__ probs,
                                                        int total, int classes, int stride, float thresh)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ov[];   // [total] overlap flags for the current i
    __shared__ int s_any;
    const int b = blockIdx.x, t = threadIdx.x;
    const float *bx = boxes + (size_t)b * total * 4;
    float *pr = probs + (size_t)b * total * stride;
    for (int i = 0; i < total; ++i) {
        if (t == 0) s_any = 0;
        __syncthreads();
        int any = 0;
        for (int k = t; k < classes; k += 256) any |= (pr[(size_t)i * stride + k] > 0);
        if (any) s_any = 1;
        __syncthreads();
        if (!s_any) continue;                // uniform
        const float4 a = *(const float4 *)(bx + (size_t)i * 4);
        for (int j = i + 1 + t; j < total; j += 256)
            ov[j] = box_iou_f(a, *(const float4 *)(bx + (size_t)j * 4)) > thresh;
        __syncthreads();
        for (int k = t; k < classes; k += 256) {
            float pi = pr[(size_t)i * stride + k];
            for (int j = i + 1; j < total; ++j) {
                if (!ov[j]) continue;
                float *pj = &pr[(size_t)j * stride + k];
                if (pi < *pj) pi = 0;
                else *pj = 0;
            }
            pr[(size_t)i * stride + k] = pi;
        }
        __syncthreads();
    }
}

extern "C" int y2h_nms(const float *boxes, float *probs, int batch, int total, int classes, int stride,
                       float thresh, y2h_stream s)
{
    if (!boxes || !probs || batch <= 0 || total <= 0 || classes <= 0 || stride < classes) return Y2H_EINVAL;
    if (total > 65536) return Y2H_EINVAL;
    hipLaunchKernelGGL(nms_plain_kernel, dim3((unsigned)batch), dim3(256), (size_t)((total + 15) / 16 * 16), S(s),
                       boxes, probs, total, classes, stride, thresh);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// ---------------------------------------------------------------------------
// collect: per image, in ascending box order, keep boxes whose best class
// (utils.c:533 max_index: first maximum) has prob > thresh
// (yolo_v2_class.cpp:221-238, image.c:662-672).  Ordered compaction by a
// workgroup-wide prefix sum so the record order equals the reference's loop order.
// ---------------------------------------------------------------------------
// best class per box (utils.c:533 max_index: the FIRST maximum): one wavefront per box, lanes stride the
// classes in ascending order, then a butterfly reduction that prefers the larger value and, on equal
// values, the smaller index
__global__ __launch_bounds__(256) void best_class_kernel(const float *__restrict__ probs, long nboxes, int classes, int stride,
                                                         float *__restrict__ best_val, int *__restrict__ best_cls)
{
    const long box = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (box >= nboxes) return;
    const float *p = probs + box * stride;
    float best = p[0];
    int cls = 0;
    bool have = (lane == 0);
    if (!have && lane < classes) { best = p[lane]; cls = lane; have = true; }
    for (int k = lane + 64; k < classes; k += 64) {
        const float v = p[k];
        if (v > best) { best = v; cls = k; }
    }
    if (!have) { best = p[0]; cls = 0; }                 // lanes beyond the class count restate element 0
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off);
        const int oc = __shfl_xor(cls, off);
        if (ov > best || (ov == best && oc < cls)) { best = ov; cls = oc; }
    }
    if (lane == 0) { best_val[box] = best; best_cls[box] = cls; }
}

__global__ __launch_bounds__(256) void collect_kernel(const float *__restrict__ boxes,
                                                      const float *__restrict__ best_val, const int *__restrict__ best_cls,
                                                      int total, float thresh,
                                                      float *__restrict__ records, int *__restrict__ counts, int max_per)
{
    __shared__ int s_scan[256];
    __shared__ int s_base;
    const int b = blockIdx.x, t = threadIdx.x;
    const float *bx = boxes + (size_t)b * total * 4;
    float *rec = records + (size_t)b * max_per * 6;
    if (t == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < total; i0 += 256) {
        const int i = i0 + t;
        int cls = 0;
        float best = 0;
        int keep = 0;
        if (i < total) {
            best = best_val[(size_t)b * total + i];
            cls = best_cls[(size_t)b * total + i];
            keep = best > thresh;
        }
        s_scan[t] = keep;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {       // inclusive Hillis-Steele scan
            const int v = (t >= off) ? s_scan[t - off] : 0;
            __syncthreads();
            s_scan[t] += v;
            __syncthreads();
        }
        const int pos = s_base + s_scan[t] - keep;
        if (keep && pos < max_per) {
            float *r = rec + (size_t)pos * 6;
            r[0] = bx[(size_t)i * 4 + 0]; r[1] = bx[(size_t)i * 4 + 1];
            r[2] = bx[(size_t)i * 4 + 2]; r[3] = bx[(size_t)i * 4 + 3];
            r[4] = best; r[5] = (float)cls;
        }
        __syncthreads();
        if (t == 255) s_base += s_scan[255];
        __syncthreads();
    }
    if (t == 0) counts[b] = s_base;
}

extern "C" int y2h_collect(const float *boxes, const float *probs, int batch, int total, int classes, int stride,
                           float thresh, float *records, int *counts, int max_per_image, float *best_scratch, y2h_stream s)
{
    if (!boxes || !probs || !records || !counts || !best_scratch || batch <= 0 || total <= 0 || classes <= 0 ||
        stride < classes || max_per_image <= 0) return Y2H_EINVAL;
    const long nboxes = (long)batch * total;
    float *best_val = best_scratch;
    int *best_cls = (int *)(best_scratch + nboxes);
    hipLaunchKernelGGL(best_class_kernel, dim3((unsigned)((nboxes * 64 + 255) / 256)), dim3(256), 0, S(s),
                       probs, nboxes, classes, stride, best_val, best_cls);
    Y2H_LAUNCH_CHECK();
    hipLaunchKernelGGL(collect_kernel, dim3((unsigned)batch), dim3(256), 0, S(s),
                       boxes, best_val, best_cls, total, thresh, records, counts, max_per_image);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// ---------------------------------------------------------------------------
// The detection chain of a plain region head in THREE launches instead of eight (batch 1, the robot's mode: every one of
// decode_boxes / decode_probs / copy / memset / class_count / nms_sort / best_class / collect runs 4-6 us for 845 boxes x 20
// classes, 60 us of a 250 us frame; profiles/r03_notes.md):
//   decode_all_kernel   = decode_boxes_kernel + decode_probs_kernel + the copy of the scores NMS suppresses in + the
//                         per-class candidate counts (which the NMS kernel zeroes again after reading them)
//   nms_sort_kernel     unchanged
//   best_collect_kernel = best_class_kernel + collect_kernel
// Same expressions, same order, same results as the separate kernels (tests/test_gpu_kernels.py compares them).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decode_all_kernel(DecodeK d, float *__restrict__ probs2, int *__restrict__ class_counts)
{
    const long total = d.nboxes * d.classes;
    const int size = d.classes + 5, per = d.w * d.h * d.num;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int j = (int)(idx % d.classes);
        const long gi = idx / d.classes;
        const float *x = d.pred + gi * size;
        float scale = x[4];
        if (d.classfix == -1 && scale < .5) scale = 0;
        const float prob = scale * x[5 + j];
        float v = (prob > d.thresh) ? prob : 0;
        if (d.only_objectness && j == 0) v = scale;
        d.probs[idx] = v;
        if (probs2) {
            probs2[idx] = v;
            if (v != 0) atomicAdd(&class_counts[(gi / per) * d.classes + j], 1);
        }
        if (j == 0) {                                   // region_layer.c:73-85 get_region_box, as decode_boxes_kernel
            const int index = (int)(gi % per);
            const int n = index % d.num, cell = index / d.num;
            const int row = cell / d.w, col = cell % d.w;
            float bx = (col + logistic_f(x[0])) / d.w;
            float by = (row + logistic_f(x[1])) / d.h;
            float bw = (float)(exp((double)x[2]) * d.anchors[2 * n] / d.w);
            float bh = (float)(exp((double)x[3]) * d.anchors[2 * n + 1] / d.h);
            bx *= d.img_w; by *= d.img_h; bw *= d.img_w; bh *= d.img_h;
            float *bo = d.boxes + gi * 4;
            bo[0] = bx; bo[1] = by; bo[2] = bw; bo[3] = bh;
        }
    }
}

// one workgroup per image: each lane finds the best class of its box (utils.c:533 max_index: the FIRST maximum), then the
// ordered compaction of collect_kernel
__global__ __launch_bounds__(256) void best_collect_kernel(const float *__restrict__ boxes, const float *__restrict__ probs,
                                                           int total, int classes, int stride, float thresh,
                                                           float *__restrict__ records, int *__restrict__ counts, int max_per)
{
    __shared__ int s_scan[256];
    __shared__ int s_base;
    const int b = blockIdx.x, t = threadIdx.x;
    const float *bx = boxes + (size_t)b * total * 4;
    const float *pr = probs + (size_t)b * total * stride;
    float *rec = records + (size_t)b * max_per * 6;
    const bool vec4 = (classes & 3) == 0 && (stride & 3) == 0 && ((size_t)pr & 15) == 0;
    if (t == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < total; i0 += 256) {
        const int i = i0 + t;
        int cls = 0, keep = 0;
        float best = 0;
        if (i < total) {
            const float *p = pr + (size_t)i * stride;
            best = p[0];
            if (vec4) {                                  // rows start on 16 bytes: four classes per load, scanned in order
                for (int k4 = 0; k4 < classes; k4 += 4) {
                    const float4 q = *(const float4 *)(p + k4);
                    if (q.x > best) { best = q.x; cls = k4; }
                    if (q.y > best) { best = q.y; cls = k4 + 1; }
                    if (q.z > best) { best = q.z; cls = k4 + 2; }
                    if (q.w > best) { best = q.w; cls = k4 + 3; }
                }
            } else {
                for (int k = 1; k < classes; ++k) {
                    const float v = p[k];
                    if (v > best) { best = v; cls = k; }
                }
            }
            keep = best > thresh;
        }
        s_scan[t] = keep;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int v = (t >= off) ? s_scan[t - off] : 0;
            __syncthreads();
            s_scan[t] += v;
            __syncthreads();
        }
        const int pos = s_base + s_scan[t] - keep;
        if (keep && pos < max_per) {
            float *r = rec + (size_t)pos * 6;
            r[0] = bx[(size_t)i * 4 + 0]; r[1] = bx[(size_t)i * 4 + 1];
            r[2] = bx[(size_t)i * 4 + 2]; r[3] = bx[(size_t)i * 4 + 3];
            r[4] = best; r[5] = (float)cls;
        }
        __syncthreads();
        if (t == 255) s_base += s_scan[255];
        __syncthreads();
    }
    if (t == 0) counts[b] = s_base;
}

extern "C" int y2h_detect_chain_ok(const y2h_decode *q)
{
    return q && !q->tree_parent && !q->map && q->classes <= 256 && (long)q->w * q->h * q->num <= 16384 && !getenv("Y2_DETECT_SEPARATE");
}

// decode + NMS (nms > 0) + compaction for a plain region head; `class_counts` (batch * classes ints) must be ZERO on entry
// and is zero again on return (the NMS workgroups reset their own words)
extern "C" int y2h_detect_chain(const y2h_decode *q, float nms, float *probs_nms, int *class_counts, float *records,
                                int *counts, int max_per_image, float *best_scratch, y2h_stream s)
{
    if (!y2h_detect_chain_ok(q) || !q->pred || !q->boxes || !q->probs || !q->anchors || !records || !counts || max_per_image <= 0)
        return Y2H_EINVAL;
    if (nms > 0 && (!probs_nms || !class_counts)) return Y2H_EINVAL;
    DecodeK d;
    d.w = q->w; d.h = q->h; d.num = q->num; d.classes = q->classes; d.img_w = q->img_w; d.img_h = q->img_h;
    d.thresh = q->thresh; d.only_objectness = q->only_objectness; d.classfix = q->classfix;
    d.anchors = q->anchors; d.parent = nullptr; d.map = nullptr;
    d.pred = q->pred; d.boxes = q->boxes; d.probs = q->probs;
    d.nboxes = (long)q->batch * q->w * q->h * q->num;
    d.tree_seq = 0;
    const int total = q->w * q->h * q->num;
    hipLaunchKernelGGL(decode_all_kernel, dim3(y2h_grid(d.nboxes * d.classes, 256)), dim3(256), 0, S(s), d,
                       nms > 0 ? probs_nms : (float *)nullptr, class_counts);
    Y2H_LAUNCH_CHECK();
    const float *final_probs = q->probs;
    if (nms > 0) {
        int cap = 1;
        while (cap < total) cap <<= 1;
        const int lds_boxes = cap <= 8192 ? NMS_LDS_BOXES : 0;
        const size_t lds = (((size_t)cap * 9 + 15) & ~(size_t)15) + (size_t)lds_boxes * 16;
        static bool attr_set[16] = {false};
        int dev = 0;
        Y2H_CHECK(hipGetDevice(&dev));
        if (dev < 0 || dev >= 16 || !attr_set[dev]) {
            Y2H_CHECK(hipFuncSetAttribute((const void *)nms_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 9));
            if (dev >= 0 && dev < 16) attr_set[dev] = true;
        }
        hipLaunchKernelGGL(nms_sort_kernel, dim3((unsigned)(q->batch * q->classes)), dim3(256), lds, S(s),
                           q->boxes, q->probs, probs_nms, class_counts, total, q->classes, q->classes, nms, cap, lds_boxes, class_counts);
        Y2H_LAUNCH_CHECK();
        final_probs = probs_nms;
    }
    // best class + ordered compaction: one workgroup per image scans its boxes' class rows itself up to ~1 M scores; beyond
    // that (yolo.cfg 608 b32: 4.6 M) the scan wants more than `batch` workgroups: the wave-per-box kernel, then the compaction
    // (measured at 608 b32: 105 us fused against 14 + 13 us; at batch 1 the fused form saves a launch)
    if ((long)d.nboxes * d.classes > (1L << 20) && best_scratch) {
        float *best_val = best_scratch;
        int *best_cls = (int *)(best_scratch + d.nboxes);
        hipLaunchKernelGGL(best_class_kernel, dim3((unsigned)((d.nboxes * 64 + 255) / 256)), dim3(256), 0, S(s),
                           final_probs, d.nboxes, q->classes, q->classes, best_val, best_cls);
        Y2H_LAUNCH_CHECK();
        hipLaunchKernelGGL(collect_kernel, dim3((unsigned)q->batch), dim3(256), 0, S(s),
                           q->boxes, best_val, best_cls, total, q->thresh, records, counts, max_per_image);
        Y2H_LAUNCH_CHECK();
        return Y2H_OK;
    }
    hipLaunchKernelGGL(best_collect_kernel, dim3((unsigned)q->batch), dim3(256), 0, S(s), q->boxes, final_probs, total, q->classes,
                       q->classes, q->thresh, records, counts, max_per_image);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// ---------------------------------------------------------------------------
// The detection chain of a TREE head without a class map (yolo9000 in detect mode: 9418 classes) in TWO launches.
// region_layer.c:351-367 leaves at most ONE non-zero score per box -- the deepest class whose hierarchy probability
// exceeds .5 (the last such index), kept only when the box's objectness exceeds the threshold -- so the dense
// [boxes][classes] score array the reference hands to do_nms_sort / max_index (8 x 867 x 9418 floats = 261 MB per batch of
// yolo9000 544, written by the decode, copied for the NMS, scanned by the class counts, by the NMS gather and by the
// best-class search) is one (class, value) pair per box:
//   decode_tree_sparse_kernel  one workgroup per box: the box (get_region_box), then -- only for boxes whose objectness
//                              passes, the others cannot produce a score -- the class row staged in LDS, hierarchy products
//                              level by level (as decode_tree_kernel: same fp32 multiplications), the last class above .5
//   nms_collect_sparse_kernel  one workgroup per image: candidates sorted by (class, score descending, box index ascending)
//                              -- do_nms_sort's order: with one non-zero class per box every tie of box.c:252-264's
//                              repeated stable sort falls through to the box index --, greedy suppression inside each class
//                              (one wavefront per class with two or more candidates), then max_index + the ordered
//                              compaction of collect_kernel (a box whose score was suppressed has an all-zero row: best 0).
// Same records and counts as y2h_region_boxes + y2h_nms_sort + y2h_collect (tests/test_gpu_ingest.py compares them); the
// dense arrays and the in-place edit of the prediction rows (nobody reads them behind a detect call) are not produced.
// ---------------------------------------------------------------------------
#define DTS_NT 512       // threads per box (as region_tree_lds_kernel: twice the bytes of a row in flight)
__global__ __launch_bounds__(DTS_NT) void decode_tree_sparse_kernel(DecodeK d, TreeK tk, float *__restrict__ cand_val, int *__restrict__ cand_cls)
{
    extern __shared__ __attribute__((aligned(16))) float row[];      // [classes]
    __shared__ int s_best;
    const long gi = blockIdx.x;
    const int t = threadIdx.x;
    const int size = d.classes + 5;
    const float *x = d.pred + gi * size;
    float scale = x[4];
    if (d.classfix == -1 && scale < .5) scale = 0;
    if (t == 0) {                                       // region_layer.c:73-85 get_region_box, as decode_boxes_kernel
        const int total = d.w * d.h * d.num;
        const int index = (int)(gi % total);
        const int n = index % d.num, cell = index / d.num;
        const int r = cell / d.w, col = cell % d.w;
        float bx = (col + logistic_f(x[0])) / d.w;
        float by = (r + logistic_f(x[1])) / d.h;
        float bw = (float)(exp((double)x[2]) * d.anchors[2 * n] / d.w);
        float bh = (float)(exp((double)x[3]) * d.anchors[2 * n + 1] / d.h);
        bx *= d.img_w; by *= d.img_h; bw *= d.img_w; bh *= d.img_h;
        float *bo = d.boxes + gi * 4;
        bo[0] = bx; bo[1] = by; bo[2] = bw; bo[3] = bh;
    }
    if (!(scale > d.thresh)) {                          // uniform: every score of this box is zero (region_layer.c:364)
        if (t == 0) { cand_val[gi] = 0.f; cand_cls[gi] = 0; }
        return;
    }
    const float *p = x + 5;
    for (int j = t; j < d.classes; j += (int)blockDim.x) row[j] = p[j];
    if (t == 0) s_best = -1;
    __syncthreads();
    for (int lv = 1; lv < tk.levels; ++lv) {            // tree.c:37-44, level by level (see decode_tree_kernel)
        const int b = tk.level_off[lv], e = tk.level_off[lv + 1];
        for (int i = b + t; i < e; i += (int)blockDim.x) {
            const int j = tk.order[i];
            row[j] *= row[d.parent[j]];
        }
        __syncthreads();
    }
    int best = -1;
    for (int j = t; j < d.classes; j += (int)blockDim.x) if (row[j] > .5) best = j;       // ascending: keeps the last
    if (best >= 0) atomicMax(&s_best, best);
    __syncthreads();
    if (t == 0) {
        best = s_best;
        cand_val[gi] = best >= 0 ? row[best] : 0.f;
        cand_cls[gi] = best >= 0 ? best : 0;
    }
}

__global__ __launch_bounds__(256) void nms_collect_sparse_kernel(const float *__restrict__ boxes, const float *__restrict__ cand_val,
                                                                 const int *__restrict__ cand_cls, int total, int cap, float nms,
                                                                 float thresh, float *__restrict__ records, int *__restrict__ counts,
                                                                 int max_per)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
    unsigned long long *keys = (unsigned long long *)sp_smem;                   // [cap]  class << 48 | score bits << 16 | ~box index
    float4 *s_box = (float4 *)(keys + cap);                                     // [cap]  boxes in sorted order
    float *bval = (float *)(s_box + cap);                                       // [cap]  score per box (0: none / suppressed)
    int *bcls = (int *)(bval + cap);                                            // [cap]
    int *runs = bcls + cap;                                                     // [cap]  start << 16 | end of a class with >= 2 candidates
    volatile unsigned char *dead = (volatile unsigned char *)(runs + cap);      // [cap]
    __shared__ int s_count, s_runs, s_base;
    __shared__ int s_scan[256];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const float *bx = boxes + (size_t)b * total * 4;
    if (t == 0) { s_count = 0; s_runs = 0; s_base = 0; }
    __syncthreads();
    for (int i = t; i < total; i += 256) {
        const float v = cand_val[(size_t)b * total + i];
        const int c = cand_cls[(size_t)b * total + i];
        bval[i] = v; bcls[i] = c;
        if (v != 0 && nms > 0) {
            const int slot = atomicAdd(&s_count, 1);
            keys[slot] = ((unsigned long long)(unsigned)c << 48) | ((unsigned long long)__float_as_uint(v) << 16) | (unsigned long long)((~(unsigned)i) & 0xffffu);
        }
    }
    __syncthreads();
    const int n = s_count;
    if (n >= 2) {
        int np2 = 1;
        while (np2 < n) np2 <<= 1;
        for (int i = n + t; i < np2; i += 256) keys[i] = 0ull;
        for (int i = t; i < np2; i += 256) dead[i] = 0;
        __syncthreads();
        for (int kk = 2; kk <= np2; kk <<= 1) {              // bitonic sort, descending
            for (int j = kk >> 1; j > 0; j >>= 1) {
                for (int i = t; i < np2; i += 256) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const unsigned long long a = keys[i], c = keys[ixj];
                        const bool desc = ((i & kk) == 0);
                        if (desc ? (a < c) : (a > c)) { keys[i] = c; keys[ixj] = a; }
                    }
                }
                __syncthreads();
            }
        }
        for (int j = t; j < n; j += 256) {
            const unsigned idx = (~(unsigned)(keys[j] & 0xffffull)) & 0xffffu;
            s_box[j] = *(const float4 *)(bx + (size_t)idx * 4);
            const unsigned c = (unsigned)(keys[j] >> 48);
            const bool starts = (j == 0 || (unsigned)(keys[j - 1] >> 48) != c) && (j + 1 < n && (unsigned)(keys[j + 1] >> 48) == c);
            if (starts) {
                int e = j + 1;
                while (e < n && (unsigned)(keys[e] >> 48) == c) ++e;
                runs[atomicAdd(&s_runs, 1)] = (j << 16) | e;
            }
        }
        __syncthreads();
        // greedy suppression (box.c:266-275), one wavefront per class: LDS operations of a wavefront execute in order, and
        // `dead` is volatile, so a flag set by the inner loop is seen by the next round's test without a barrier
        const int nr = s_runs;
        for (int r = wv; r < nr; r += 4) {
            const int rs = runs[r] >> 16, re = runs[r] & 0xffff;
            for (int i = rs; i < re - 1; ++i) {
                if (!dead[i]) {
                    const float4 a = s_box[i];
                    for (int j = i + 1 + lane; j < re; j += 64)
                        if (box_iou_f(a, s_box[j]) > nms) dead[j] = 1;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        for (int j = t; j < n; j += 256)
            if (dead[j]) bval[(~(unsigned)(keys[j] & 0xffffull)) & 0xffffu] = 0.f;
        __syncthreads();
    }
    // max_index over a row with at most one non-zero score, then collect_kernel's ordered compaction
    float *rec = records + (size_t)b * max_per * 6;
    for (int i0 = 0; i0 < total; i0 += 256) {
        const int i = i0 + t;
        float best = 0;
        int cls = 0, keep = 0;
        if (i < total) {
            best = bval[i];
            cls = best > 0 ? bcls[i] : 0;
            keep = best > thresh;
        }
        s_scan[t] = keep;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int v = (t >= off) ? s_scan[t - off] : 0;
            __syncthreads();
            s_scan[t] += v;
            __syncthreads();
        }
        const int pos = s_base + s_scan[t] - keep;
        if (keep && pos < max_per) {
            float *r = rec + (size_t)pos * 6;
            r[0] = bx[(size_t)i * 4 + 0]; r[1] = bx[(size_t)i * 4 + 1];
            r[2] = bx[(size_t)i * 4 + 2]; r[3] = bx[(size_t)i * 4 + 3];
            r[4] = best; r[5] = (float)cls;
        }
        __syncthreads();
        if (t == 255) s_base += s_scan[255];
        __syncthreads();
    }
    if (t == 0) counts[b] = s_base;
}

// the first launch when the region layer has already produced the (score, class) pair of every box
// (y2h_region_forward_tree): the box and the objectness test, one lane per box
__global__ __launch_bounds__(256) void decode_tree_cand_kernel(DecodeK d, const float *__restrict__ pre_val, const int *__restrict__ pre_cls,
                                                               float *__restrict__ cand_val, int *__restrict__ cand_cls)
{
    const long gi = (long)blockIdx.x * 256 + threadIdx.x;
    if (gi >= d.nboxes) return;
    const int size = d.classes + 5;
    const float *x = d.pred + gi * size;
    float scale = x[4];
    if (d.classfix == -1 && scale < .5) scale = 0;
    const int total = d.w * d.h * d.num;
    const int index = (int)(gi % total);
    const int n = index % d.num, cell = index / d.num;
    const int r = cell / d.w, col = cell % d.w;
    float bx = (col + logistic_f(x[0])) / d.w;
    float by = (r + logistic_f(x[1])) / d.h;
    float bw = (float)(exp((double)x[2]) * d.anchors[2 * n] / d.w);
    float bh = (float)(exp((double)x[3]) * d.anchors[2 * n + 1] / d.h);
    bx *= d.img_w; by *= d.img_h; bw *= d.img_w; bh *= d.img_h;
    float *bo = d.boxes + gi * 4;
    bo[0] = bx; bo[1] = by; bo[2] = bw; bo[3] = bh;
    const bool on = scale > d.thresh;
    cand_val[gi] = on ? pre_val[gi] : 0.f;
    cand_cls[gi] = on ? pre_cls[gi] : 0;
}

extern "C" int y2h_detect_tree_chain_ok(const y2h_decode *q)
{
    return q && q->tree_parent && !q->map && !q->only_objectness && q->thresh >= 0 && q->tree_order && q->tree_level_off &&
           q->tree_levels > 0 && q->classes < 65536 && (size_t)q->classes * sizeof(float) <= 150 * 1024 &&
           (long)q->w * q->h * q->num <= 4096 && !getenv("Y2_DETECT_SEPARATE");
}

// decode + NMS (nms > 0) + compaction of a tree head without a map; best_scratch: 2 * boxes floats (as y2h_collect)
extern "C" int y2h_detect_tree_chain(const y2h_decode *q, float nms, float *records, int *counts, int max_per_image,
                                     float *best_scratch, const float *tree_best, y2h_stream s)
{
    if (!y2h_detect_tree_chain_ok(q) || !q->pred || !q->boxes || !q->anchors || !records || !counts || !best_scratch || max_per_image <= 0 ||
        q->batch <= 0 || q->w <= 0 || q->h <= 0 || q->num <= 0 || q->classes <= 0)
        return Y2H_EINVAL;
    DecodeK d;
    d.w = q->w; d.h = q->h; d.num = q->num; d.classes = q->classes; d.img_w = q->img_w; d.img_h = q->img_h;
    d.thresh = q->thresh; d.only_objectness = 0; d.classfix = q->classfix;
    d.anchors = q->anchors; d.parent = q->tree_parent; d.map = nullptr;
    d.pred = q->pred; d.boxes = q->boxes; d.probs = nullptr;
    d.nboxes = (long)q->batch * q->w * q->h * q->num;
    d.tree_seq = 0;
    TreeK tk;
    tk.order = q->tree_order; tk.level_off = q->tree_level_off; tk.levels = q->tree_levels;
    const int total = q->w * q->h * q->num;
    float *cand_val = best_scratch;
    int *cand_cls = (int *)(best_scratch + d.nboxes);
    int cap = 16;
    while (cap < total) cap <<= 1;
    const size_t lds2 = (size_t)cap * (8 + 16 + 4 + 4 + 4 + 1);
    static bool attr_set[16] = {false};
    int dev = 0;
    Y2H_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
        Y2H_CHECK(hipFuncSetAttribute((const void *)decode_tree_sparse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        Y2H_CHECK(hipFuncSetAttribute((const void *)nms_collect_sparse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4096 * 37));
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    if (tree_best)
        hipLaunchKernelGGL(decode_tree_cand_kernel, dim3((unsigned)((d.nboxes + 255) / 256)), dim3(256), 0, S(s), d, tree_best,
                           (const int *)(tree_best + d.nboxes), cand_val, cand_cls);
    else
        hipLaunchKernelGGL(decode_tree_sparse_kernel, dim3((unsigned)d.nboxes), dim3(getenv("Y2_DTS_THREADS") && atoi(getenv("Y2_DTS_THREADS")) == 256 ? 256 : DTS_NT), (size_t)q->classes * sizeof(float), S(s), d, tk,
                           cand_val, cand_cls);
    Y2H_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_collect_sparse_kernel, dim3((unsigned)q->batch), dim3(256), lds2, S(s), q->boxes, cand_val, cand_cls, total, cap,
                       nms, q->thresh, records, counts, max_per_image);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// ---------------------------------------------------------------------------
// resize_image (image.c:1950-1992): separable align-corners bilinear, two fp32
// passes (columns first into `tmp` [c][ih][w], then rows), same rounding order
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_cols_kernel(const float *__restrict__ src, float *__restrict__ part,
                                                          int c, int ih, int iw, int w, float w_scale)
{
    const long total = (long)c * ih * w;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int col = (int)(idx % w);
        const long kr = idx / w;                        // k*ih + r
        const float *row = src + kr * iw;
        float val;
        if (col == w - 1 || iw == 1) val = row[iw - 1];
        else {
            const float sx = col * w_scale;
            const int ix = (int)sx;
            const float dx = sx - ix;
            val = (1 - dx) * row[ix] + dx * row[ix + 1];
        }
        part[idx] = val;
    }
}

__global__ __launch_bounds__(256) void resize_rows_kernel(const float *__restrict__ part, float *__restrict__ dst,
                                                          int c, int ih, int w, int h, float h_scale)
{
    const long total = (long)c * h * w;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int col = (int)(idx % w);
        const int r = (int)((idx / w) % h);
        const int k = (int)(idx / ((long)w * h));
        const float sy = r * h_scale;
        const int iy = (int)sy;
        const float dy = sy - iy;
        const float *p = part + ((long)k * ih + iy) * w + col;
        float val = (1 - dy) * p[0];
        if (!(r == h - 1 || ih == 1)) val = val + dy * p[w];
        dst[idx] = val;
    }
}

extern "C" int y2h_resize_chw(const float *src, int c, int ih, int iw, float *tmp, float *dst, int h, int w, y2h_stream s)
{
    if (!src || !tmp || !dst || c <= 0 || ih <= 0 || iw <= 0 || h <= 0 || w <= 0) return Y2H_EINVAL;
    const float w_scale = (float)(iw - 1) / (w - 1);
    const float h_scale = (float)(ih - 1) / (h - 1);
    hipLaunchKernelGGL(resize_cols_kernel, dim3(y2h_grid((long)c * ih * w, 256)), dim3(256), 0, S(s), src, tmp, c, ih, iw, w, w_scale);
    Y2H_LAUNCH_CHECK();
    hipLaunchKernelGGL(resize_rows_kernel, dim3(y2h_grid((long)c * h * w, 256)), dim3(256), 0, S(s), tmp, dst, c, ih, w, h, h_scale);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}

// ---------------------------------------------------------------------------
// YOLOv1 head decode: detection_layer.c:222-251 get_detection_boxes.  pred = one [detection] output per batch
// item: side*side*classes class scores, side*side*num box confidences, side*side*num*4 box terms.
// One thread per (image, cell, box); the C expressions are mirrored operand for operand: (p + col) / side * w in
// fp32 with the int operands converted, pow(p, sqrt ? 2 : 1) * w in double (p*p is exact in double, which is what
// glibc's pow returns for the exponent 2), prob = scale * class score, kept when > thresh.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void detection_boxes_kernel(const float *__restrict__ pred, long pred_stride, int batch, int side,
                                                              int num, int classes, int sq, int w, int h, float thresh,
                                                              int only_objectness, float *__restrict__ boxes,
                                                              float *__restrict__ probs)
{
    const long per = (long)side * side * num;
    const long total = per * batch;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long b = idx / per;
        const int index = (int)(idx - b * per);          // i*num + n
        const int i = index / num;
        const int row = i / side, col = i % side;
        const float *p = pred + b * pred_stride;
        const float scale = p[(long)side * side * classes + index];
        const float *bx = p + (long)side * side * (classes + num) + (long)index * 4;
        float *ob = boxes + idx * 4;
        ob[0] = (bx[0] + col) / side * w;
        ob[1] = (bx[1] + row) / side * h;
        const double pw = sq ? (double)bx[2] * (double)bx[2] : (double)bx[2];
        const double ph = sq ? (double)bx[3] * (double)bx[3] : (double)bx[3];
        ob[2] = (float)(pw * w);
        ob[3] = (float)(ph * h);
        float *op = probs + idx * classes;
        const float *cls = p + (long)i * classes;
        for (int j = 0; j < classes; ++j) {
            const float prob = scale * cls[j];
            op[j] = (prob > thresh) ? prob : 0.f;
        }
        if (only_objectness) op[0] = scale;
    }
}

extern "C" int y2h_detection_boxes(const float *pred, long pred_stride, int batch, int side, int num, int classes, int sqrt_flag,
                                   int w, int h, float thresh, int only_objectness, float *boxes, float *probs, y2h_stream s)
{
    if (!pred || !boxes || !probs || batch <= 0 || side <= 0 || num <= 0 || classes <= 0) return Y2H_EINVAL;
    const long total = (long)batch * side * side * num;
    hipLaunchKernelGGL(detection_boxes_kernel, dim3(y2h_grid(total, 256)), dim3(256), 0, S(s), pred, pred_stride, batch, side, num,
                       classes, sqrt_flag, w, h, thresh, only_objectness, boxes, probs);
    Y2H_LAUNCH_CHECK();
    return Y2H_OK;
}
