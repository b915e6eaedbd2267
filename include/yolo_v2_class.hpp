// yolo_v2_class.hpp -- C++ Detector API of the MI355X-native YOLOv2 engine.
//
// Same public surface as the reference's src_yolo2/yolo_v2_class.hpp:27-146
// (struct bbox_t, struct image_t, class Detector with detect / load_image /
// free_image / get_net_width / get_net_height / tracking and the public `nms`
// member), so yolo_console_dll.cpp-style callers compile unchanged.  The
// OpenCV convenience overloads of the reference (hpp:59-143) are header-only
// glue around detect(image_t) and are provided under the same OPENCV guard.
#pragma once
#include <deque>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#ifdef OPENCV
#include <opencv2/opencv.hpp>
#endif

#if defined(_MSC_VER)
#define YOLODLL_API __declspec(dllexport)
#else
#define YOLODLL_API __attribute__((visibility("default")))
#endif

struct bbox_t {
    unsigned int x, y, w, h;     // top-left corner and size, in pixels of the image passed to detect()
    float prob;                  // confidence of the best class
    unsigned int obj_id;         // class index in [0, classes)
    unsigned int track_id;       // 0 = untracked; tracking() assigns 1, 2, ...
};

struct image_t {
    int h, w, c;                 // CHW planes
    float *data;                 // values in [0,1]
};

class Detector {
    std::shared_ptr<void> detector_gpu_ptr;
public:
    float nms = .4f;

    YOLODLL_API Detector(std::string cfg_filename, std::string weight_filename, int gpu_id = 0);
    YOLODLL_API ~Detector();

    YOLODLL_API std::vector<bbox_t> detect(std::string image_filename, float thresh = 0.2f, bool use_mean = false);
    YOLODLL_API std::vector<bbox_t> detect(image_t img, float thresh = 0.2f, bool use_mean = false);
    static YOLODLL_API image_t load_image(std::string image_filename);
    static YOLODLL_API void free_image(image_t m);
    YOLODLL_API int get_net_width() const;
    YOLODLL_API int get_net_height() const;
    YOLODLL_API std::vector<bbox_t> tracking(std::vector<bbox_t> cur_bbox_vec, int const frames_story = 6);

    // Extension (no reference counterpart): a raw 8-bit interleaved camera frame (w x h x c, row pitch
    // `step` bytes, BGR(A) when bgr) goes to the GPU as bytes; conversion to [0,1] RGB planes and the
    // resize to the network size (hpp:94-141 + cpp:195-200 of the reference, done there on the host)
    // run on the device.  Boxes are in pixels of the frame, exactly as detect(image_t) returns them.
    YOLODLL_API std::vector<bbox_t> detect_frame(const unsigned char *data, int w, int h, int c, int step,
                                                 float thresh = 0.2f, bool bgr = true);

#ifdef OPENCV
    // BGR 8-bit cv::Mat -> resized planar RGB float image -> detect -> boxes scaled back to mat's size
    std::vector<bbox_t> detect(cv::Mat mat, float thresh = 0.2f, bool use_mean = false)
    {
        if (mat.data == NULL) throw std::runtime_error("Image is empty");
        cv::Mat small;
        cv::resize(mat, small, cv::Size(get_net_width(), get_net_height()));
        image_t im;
        im.h = small.rows; im.w = small.cols; im.c = small.channels();
        std::vector<float> planes((size_t)im.h * im.w * im.c);
        for (int k = 0; k < im.c; ++k)
            for (int y = 0; y < im.h; ++y)
                for (int x = 0; x < im.w; ++x)
                    planes[((size_t)(im.c - 1 - k) * im.h + y) * im.w + x] = (float)(small.ptr<unsigned char>(y)[x * im.c + k] / 255.);
        im.data = planes.data();
        std::vector<bbox_t> out = detect(im, thresh, use_mean);
        const float wk = (float)mat.cols / im.w, hk = (float)mat.rows / im.h;
        for (auto &b : out) { b.x *= wk; b.w *= wk; b.y *= hk; b.h *= hk; }
        return out;
    }
#endif

private:
    std::deque<std::vector<bbox_t>> prev_bbox_vec_deque;
};
