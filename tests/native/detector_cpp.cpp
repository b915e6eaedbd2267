// A yolo_console_dll.cpp-style caller of the C++ Detector class (yolo_v2_class.hpp:42-146).
//   detector_cpp <cfg> <weights> <frame.bin> <thresh>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "yolo_v2_class.hpp"

int main(int argc, char **argv)
{
    if (argc < 5) return 2;
    Detector det(argv[1], argv[2], 0);
    FILE *f = std::fopen(argv[3], "rb");
    int hdr[3];
    if (!f || std::fread(hdr, sizeof(int), 3, f) != 3) return 2;
    image_t im;
    im.c = hdr[0]; im.h = hdr[1]; im.w = hdr[2];
    std::vector<float> data((size_t)im.c * im.h * im.w);
    if (std::fread(data.data(), sizeof(float), data.size(), f) != data.size()) return 2;
    std::fclose(f);
    im.data = data.data();
    const float thresh = (float)std::atof(argv[4]);
    std::vector<bbox_t> boxes = det.detect(im, thresh);
    std::printf("NET %d %d BOXES %zu\n", det.get_net_width(), det.get_net_height(), boxes.size());
    for (const bbox_t &b : boxes)
        std::printf("BOX %u %u %u %u %.9g %u %u\n", b.x, b.y, b.w, b.h, b.prob, b.obj_id, b.track_id);
    // use_mean: the first call averages one real frame with two zero frames (cpp:208-213)
    std::vector<bbox_t> m1 = det.detect(im, thresh, true);
    std::vector<bbox_t> m2 = det.detect(im, thresh, true);
    std::vector<bbox_t> m3 = det.detect(im, thresh, true);
    std::printf("MEAN %zu %zu %zu\n", m1.size(), m2.size(), m3.size());
    for (const bbox_t &b : m3)
        std::printf("MBOX %u %u %u %u %.9g %u\n", b.x, b.y, b.w, b.h, b.prob, b.obj_id);
    // tracking: ids are assigned on the first call and kept on the second (same boxes)
    std::vector<bbox_t> t1 = det.tracking(boxes), t2 = det.tracking(boxes);
    bool ok = t1.size() == t2.size();
    for (size_t i = 0; ok && i < t1.size(); ++i) ok = t1[i].track_id > 0 && t2[i].track_id > 0;
    std::printf("TRACK %d\n", ok ? 1 : 0);
    // camera-frame entry: the same bytes through detect_frame (device ingest) and through the host conversion
    // the reference's OpenCV glue does (hpp:94-141: v/255. into planes, BGR -> RGB) + detect(image_t)
    {
        const int w = im.w, h = im.h;
        std::vector<unsigned char> bgr((size_t)w * h * 3);
        std::vector<float> planes((size_t)w * h * 3);
        for (int k = 0; k < 3; ++k) for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
            const unsigned char v = (unsigned char)(data[((size_t)k * h + y) * w + x] * 255.f);
            bgr[((size_t)y * w + x) * 3 + (2 - k)] = v;
            planes[((size_t)k * h + y) * w + x] = (float)(v / 255.);
        }
        image_t q = im; q.data = planes.data();
        std::vector<bbox_t> a = det.detect(q, thresh), b = det.detect_frame(bgr.data(), w, h, 3, w * 3, thresh, true);
        bool same = a.size() == b.size();
        for (size_t i = 0; same && i < a.size(); ++i)
            same = a[i].x == b[i].x && a[i].y == b[i].y && a[i].w == b[i].w && a[i].h == b[i].h && a[i].prob == b[i].prob &&
                   a[i].obj_id == b[i].obj_id;
        std::printf("FRAME %d %zu\n", same ? 1 : 0, b.size());
    }
    try { Detector::load_image("/nonexistent.ppm"); std::printf("NOTHROW\n"); }
    catch (const std::exception &e) { std::printf("THROW %s\n", e.what()); }
    return 0;
}
