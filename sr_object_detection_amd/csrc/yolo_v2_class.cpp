// Detector: the C++ face of the engine (reference: src_yolo2/yolo_v2_class.cpp).
//
//  ctor     cpp:37-81    parse_network_cfg + load_weights + set_batch_network(1)
//  detect   cpp:173-249  (resize) -> predict -> get_region_boxes(1,1,thresh) ->
//                        do_nms_sort(nms) -> bbox_t conversion with unsigned truncation
//  tracking cpp:251-303  nearest-centre (< 100 px) track-id hand-over across frames
//
// The frame goes up once (y2_ingest_image: device-side resize into the network input) and the whole
// chain runs HBM-resident: y2_forward_device + y2_detect_resident, or with use_mean y2_detect_mean (the
// 3-frame ring of raw predictions, cpp:208-213, its average, decode and NMS on the device).
#include "yolo_v2_class.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "sr_yolo2.h"
#include "y2_hip.h"

extern "C" void y2_set_error_mode(int mode);

namespace {


struct DetectorState {
    network net;
    std::vector<unsigned int> next_track_id;     // per class
    std::vector<y2_det> dets;                    // compact detections of the last call
};

DetectorState &state_of(const std::shared_ptr<void> &p) { return *static_cast<DetectorState *>(p.get()); }

bbox_t to_bbox(float bx, float by, float bw, float bh, float prob, int cls, int im_w, int im_h)
{
    // cpp:229-235: doubles, clamped at 0, then truncated to unsigned
    bbox_t r;
    r.x = (unsigned int)std::max((double)0, (bx - bw / 2.) * im_w);
    r.y = (unsigned int)std::max((double)0, (by - bh / 2.) * im_h);
    r.w = (unsigned int)(bw * im_w);
    r.h = (unsigned int)(bh * im_h);
    r.prob = prob;
    r.obj_id = (unsigned int)cls;
    r.track_id = 0;
    return r;
}

// The reference saves the caller's current device around the constructor, detect() and the destructor and
// restores it afterwards (cpp:41,48,79,99-108,178-181,245), so that several Detectors on different GPUs can be
// driven from one thread; the engine selects its own device on every call, this puts the caller's back.
struct DeviceGuard {
    int saved = -1;
    DeviceGuard() { if (y2h_device_count() > 0 && y2h_get_device(&saved) != 0) saved = -1; }
    ~DeviceGuard() { if (saved >= 0) y2h_set_device(saved); }
};

}  // namespace

Detector::Detector(std::string cfg_filename, std::string weight_filename, int gpu_id)
{
    DeviceGuard guard;
    auto *st = new DetectorState();
    detector_gpu_ptr = std::shared_ptr<void>(st, [](void *p) { delete static_cast<DetectorState *>(p); });
    const int saved = gpu_index;
    gpu_index = gpu_id;
    st->net = parse_network_cfg(const_cast<char *>(cfg_filename.c_str()));
    gpu_index = saved;
    if (!st->net.layers) throw std::runtime_error(std::string("cannot build network: ") + y2_last_error());
    st->net.gpu_index = gpu_id;
    if (!weight_filename.empty()) load_weights(&st->net, const_cast<char *>(weight_filename.c_str()));
    set_batch_network(&st->net, 1);
    const layer &l = st->net.layers[st->net.n - 1];
    st->next_track_id.assign(std::max(l.classes, 1), 1u);
    st->dets.resize(std::max(l.w * l.h * l.n, 1));
}

Detector::~Detector()
{
    DeviceGuard guard;
    if (detector_gpu_ptr) free_network(state_of(detector_gpu_ptr).net);
}

int Detector::get_net_width() const { return state_of(detector_gpu_ptr).net.w; }
int Detector::get_net_height() const { return state_of(detector_gpu_ptr).net.h; }

// Image file -> planar RGB floats in [0,1], three channels whatever the file holds (the reference calls
// stbi_load(..., 3) and divides by 255., cpp:127-149).  PNG, baseline JPEG and binary PNM are decoded by
// y2_imgfile.cpp (dependency-free; progressive JPEG is refused with a message); a missing file throws
// "file not found" as the reference does.
void y2_decode_image_file(const std::string &path, int &w, int &h, std::vector<unsigned char> &rgb);

image_t Detector::load_image(std::string image_filename)
{
    int w = 0, h = 0;
    std::vector<unsigned char> raw;
    y2_decode_image_file(image_filename, w, h, raw);
    image_t im;
    im.w = w; im.h = h; im.c = 3;
    im.data = (float *)std::calloc((size_t)w * h * 3, sizeof(float));
    if (!im.data) throw std::runtime_error("load_image: out of memory");
    for (int k = 0; k < 3; ++k)
        for (int j = 0; j < h; ++j)
            for (int i = 0; i < w; ++i)
                im.data[i + (size_t)w * j + (size_t)w * h * k] = (float)raw[k + 3 * i + 3 * (size_t)w * j] / 255.;
    return im;
}

void Detector::free_image(image_t m) { if (m.data) std::free(m.data); }

std::vector<bbox_t> Detector::detect(std::string image_filename, float thresh, bool use_mean)
{
    image_t im = load_image(image_filename);
    std::vector<bbox_t> r;
    try { r = detect(im, thresh, use_mean); } catch (...) { free_image(im); throw; }
    free_image(im);
    return r;
}

std::vector<bbox_t> Detector::detect(image_t img, float thresh, bool use_mean)
{
    DeviceGuard guard;
    DetectorState &st = state_of(detector_gpu_ptr);
    network &net = st.net;
    if (!img.data) throw std::runtime_error("Image is empty");
    image im; im.w = img.w; im.h = img.h; im.c = img.c; im.data = img.data;
    // cpp:195-200 resize_image when the sizes differ, then the input copy of network_predict: one upload and
    // a device-side resize straight into the network input
    if (y2_ingest_image(net, im) != 0) throw std::runtime_error(y2_last_error());
    const layer &last = net.layers[net.n - 1];
    const int total = last.w * last.h * last.n;
    std::vector<bbox_t> out;
    // use_mean (cpp:208-213): the three-frame ring, its average, decode and NMS all stay in HBM (y2_detect_mean)
    int count = 0;
    int rc = y2_forward_device(net, NULL);
    if (rc == 0)
        rc = use_mean ? y2_detect_mean(net, thresh, nms, 1, 1, st.dets.data(), &count, total)
                      : y2_detect_resident(net, thresh, nms, 1, 1, st.dets.data(), &count, total);
    if (rc != 0) throw std::runtime_error(y2_last_error());
    for (int i = 0; i < std::min(count, total); ++i) {
        const y2_det &d = st.dets[i];
        out.push_back(to_bbox(d.x, d.y, d.w, d.h, d.prob, d.obj_id, im.w, im.h));
    }
    return out;
}

std::vector<bbox_t> Detector::detect_frame(const unsigned char *data, int w, int h, int c, int step, float thresh, bool bgr)
{
    DeviceGuard guard;
    DetectorState &st = state_of(detector_gpu_ptr);
    network &net = st.net;
    if (!data) throw std::runtime_error("Image is empty");
    const layer &last = net.layers[net.n - 1];
    const int total = last.w * last.h * last.n;
    int count = 0;
    if (y2_detect_u8(net, data, h, w, c, step, bgr ? 1 : 0, 0, thresh, nms, 1, 1, st.dets.data(), &count, total) != 0)
        throw std::runtime_error(y2_last_error());
    std::vector<bbox_t> out;
    for (int i = 0; i < std::min(count, total); ++i) {
        const y2_det &d = st.dets[i];
        out.push_back(to_bbox(d.x, d.y, d.w, d.h, d.prob, d.obj_id, w, h));
    }
    return out;
}

std::vector<bbox_t> Detector::tracking(std::vector<bbox_t> cur, int const frames_story)
{
    DetectorState &st = state_of(detector_gpu_ptr);
    auto remember = [&]() {
        prev_bbox_vec_deque.push_front(cur);
        if ((int)prev_bbox_vec_deque.size() > frames_story) prev_bbox_vec_deque.pop_back();
    };
    auto fresh_id = [&](bbox_t &b) {
        if (b.obj_id >= st.next_track_id.size()) st.next_track_id.resize(b.obj_id + 1, 1u);
        b.track_id = st.next_track_id[b.obj_id]++;
    };
    bool history = false;
    for (auto &v : prev_bbox_vec_deque) if (!v.empty()) history = true;
    if (!history) {
        for (auto &b : cur) fresh_id(b);
        remember();
        return cur;
    }
    std::vector<unsigned int> best(cur.size(), std::numeric_limits<unsigned int>::max());
    for (auto &frame : prev_bbox_vec_deque) {
        for (auto &old : frame) {
            int pick = -1;
            for (size_t m = 0; m < cur.size(); ++m) {
                const bbox_t &k = cur[m];
                if (old.obj_id != k.obj_id) continue;
                const float dx = (float)(old.x + old.w / 2) - (float)(k.x + k.w / 2);
                const float dy = (float)(old.y + old.h / 2) - (float)(k.y + k.h / 2);
                const unsigned int dist = (unsigned int)std::sqrt(dx * dx + dy * dy);
                if (dist < 100 && (k.track_id == 0 || best[m] > dist)) { best[m] = dist; pick = (int)m; }
            }
            const bool taken = std::any_of(cur.begin(), cur.end(), [&](const bbox_t &b) {
                return b.track_id == old.track_id && b.obj_id == old.obj_id; });
            if (pick >= 0 && !taken) {
                cur[pick].track_id = old.track_id;
                cur[pick].w = (cur[pick].w + old.w) / 2;
                cur[pick].h = (cur[pick].h + old.h) / 2;
            }
        }
    }
    for (auto &b : cur) if (b.track_id == 0) fresh_id(b);
    remember();
    return cur;
}

// ---------------------------------------------------------------------------
// C face of the Detector class for FFI callers (ctypes, cgo, JNI: languages that cannot bind a C++ class).  Plain
// pointers and sizes; an exception becomes a negative return and a message in y2_last_error().  Declared in
// include/sr_yolo2.h.  (The reference exports only the C++ class from its DLL, yolo_v2_class.hpp:42-57.)
// ---------------------------------------------------------------------------
extern "C" void y2_set_error_(const char *msg);

extern "C" void *y2_detector_create(const char *cfg, const char *weights, int gpu_id)
{
    try { return new Detector(cfg ? cfg : "", weights ? weights : "", gpu_id); }
    catch (const std::exception &e) { y2_set_error_(e.what()); return nullptr; }
}

extern "C" void y2_detector_destroy(void *det) { delete static_cast<Detector *>(det); }

extern "C" int y2_detector_net_size(void *det, int *w, int *h)
{
    if (!det) return -1;
    if (w) *w = static_cast<Detector *>(det)->get_net_width();
    if (h) *h = static_cast<Detector *>(det)->get_net_height();
    return 0;
}

// detect(image_t) (+ tracking when track != 0): up to `max` boxes are written to out[] as 7 unsigned / float words each in
// bbox_t's own layout; returns the number of boxes found (which may exceed max) or < 0
extern "C" int y2_detector_detect(void *det, const float *chw, int c, int h, int w, float thresh, int use_mean, float nms,
                                  int track, void *out, int max)
{
    if (!det || !chw) { y2_set_error_("y2_detector_detect: NULL detector or image"); return -1; }
    try {
        Detector *d = static_cast<Detector *>(det);
        if (nms >= 0.f) d->nms = nms;
        image_t im; im.c = c; im.h = h; im.w = w; im.data = const_cast<float *>(chw);
        std::vector<bbox_t> r = d->detect(im, thresh, use_mean != 0);
        if (track) r = d->tracking(r);
        const int n = (int)r.size();
        if (out && max > 0) std::memcpy(out, r.data(), sizeof(bbox_t) * (size_t)std::min(n, max));
        return n;
    } catch (const std::exception &e) { y2_set_error_(e.what()); return -1; }
}
