// The calls yolo_console_dll.cpp makes on a video frame (reference src_yolo2/yolo_console_dll.cpp:137,148 and the
// cv::Mat overload of detect): det_image = detector.mat_to_image_resize(frame); detector.detect_resized(*det_image,
// frame_size, thresh, use_mean); detector.tracking(...).  Built with -DOPENCV against the stand-in cv::Mat of
// tests/native/opencv_stub (this container has no OpenCV).
//   console_dll_like <cfg> <weights> <frame.bin (int c,h,w + floats, CHW RGB in [0,1])> <thresh>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "yolo_v2_class.hpp"

int main(int argc, char **argv)
{
    if (argc < 5) return 2;
    Detector detector(argv[1], argv[2], 0);
    FILE *f = std::fopen(argv[3], "rb");
    int hdr[3];
    if (!f || std::fread(hdr, sizeof(int), 3, f) != 3) return 2;
    const int c = hdr[0], h = hdr[1], w = hdr[2];
    std::vector<float> chw((size_t)c * h * w);
    if (std::fread(chw.data(), sizeof(float), chw.size(), f) != chw.size()) return 2;
    std::fclose(f);
    const float thresh = (float)std::atof(argv[4]);
    // the camera frame: 8-bit BGR at the network's own size (so the stand-in resize is the identity)
    cv::Mat frame(h, w, 3);
    std::vector<float> planes((size_t)3 * h * w);
    for (int k = 0; k < 3; ++k) for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
        const unsigned char v = (unsigned char)(chw[((size_t)k * h + y) * w + x] * 255.f);
        frame.ptr<unsigned char>(y)[x * 3 + (2 - k)] = v;
        planes[((size_t)k * h + y) * w + x] = (float)(v / 255.);
    }
    cv::Size const frame_size = frame.size();
    std::shared_ptr<image_t> det_image = detector.mat_to_image_resize(frame);
    bool same_pixels = det_image && det_image->w == detector.get_net_width() && det_image->h == detector.get_net_height() && det_image->c == 3;
    if (same_pixels && det_image->w == w && det_image->h == h)
        for (size_t i = 0; same_pixels && i < planes.size(); ++i) same_pixels = det_image->data[i] == planes[i];
    std::printf("IMAGE %d\n", same_pixels ? 1 : 0);
    std::vector<bbox_t> result_vec = detector.detect_resized(*det_image, frame_size, thresh, false);
    image_t direct; direct.c = 3; direct.h = h; direct.w = w; direct.data = planes.data();
    std::vector<bbox_t> want = detector.detect(direct, thresh);
    bool same = result_vec.size() == want.size();
    for (size_t i = 0; same && i < want.size(); ++i)
        same = result_vec[i].x == want[i].x && result_vec[i].y == want[i].y && result_vec[i].w == want[i].w && result_vec[i].h == want[i].h &&
               result_vec[i].prob == want[i].prob && result_vec[i].obj_id == want[i].obj_id;
    std::printf("RESIZED %d %zu\n", same ? 1 : 0, result_vec.size());
    // a frame of twice the size: boxes come back in the frame's pixels (x2)
    cv::Mat big;
    cv::resize(frame, big, cv::Size(2 * w, 2 * h));
    std::vector<bbox_t> big_boxes = detector.detect(big, thresh);
    bool scaled = big_boxes.size() == want.size();
    for (size_t i = 0; scaled && i < want.size(); ++i) scaled = big_boxes[i].obj_id == want[i].obj_id;
    std::printf("MAT %d %zu\n", scaled ? 1 : 0, big_boxes.size());
    result_vec = detector.tracking(result_vec);
    std::printf("TRACKED %zu\n", result_vec.size());
    try { detector.detect(cv::Mat(), thresh); std::printf("NOTHROW\n"); }
    catch (const std::exception &e) { std::printf("THROW %s\n", e.what()); }
    return 0;
}
