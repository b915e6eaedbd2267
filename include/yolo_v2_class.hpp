// yolo_v2_class.hpp -- C++ Detector API of the MI355X-native YOLOv2 engine.
//
// Same public surface as the reference's src_yolo2/yolo_v2_class.hpp:27-146
// (struct bbox_t, struct image_t, class Detector with detect / load_image /
// free_image / get_net_width / get_net_height / tracking and the public `nms`
// member), so yolo_console_dll.cpp-style callers compile unchanged.  The
// OpenCV convenience overloads of the reference (hpp:59-143) are header-only
// glue around detect(image_t) and are provided under the same OPENCV guard.
#pragma once
#include <cstdlib>
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
    // The OpenCV convenience surface of the reference (yolo_v2_class.hpp:59-92), same names, signatures and results, so
    // that yolo_console_dll.cpp:137,148 (det_image = detector.mat_to_image_resize(frame); detector.detect_resized(
    // *det_image, frame_size, 0.24, true)) compiles unchanged.  Written against cv::Mat only (the reference goes through
    // the C-API IplImage, which OpenCV 4 no longer ships).
    std::vector<bbox_t> detect(cv::Mat mat, float thresh = 0.2f, bool use_mean = false)
    {
        if (mat.data == NULL) throw std::runtime_error("Image is empty");
        auto image_ptr = mat_to_image_resize(mat);
        return detect_resized(*image_ptr, mat.size(), thresh, use_mean);
    }

    // detect on an image already at network size; boxes scaled back to the frame it was resized from (hpp:67-75)
    std::vector<bbox_t> detect_resized(image_t img, cv::Size init_size, float thresh = 0.2f, bool use_mean = false)
    {
        if (img.data == NULL) throw std::runtime_error("Image is empty");
        std::vector<bbox_t> boxes = detect(img, thresh, use_mean);
        const float wk = (float)init_size.width / img.w, hk = (float)init_size.height / img.h;
        for (auto &b : boxes) { b.x *= wk; b.w *= wk; b.y *= hk; b.h *= hk; }
        return boxes;
    }

    // cv::resize to the network size, then mat_to_image (hpp:77-83)
    std::shared_ptr<image_t> mat_to_image_resize(cv::Mat mat) const
    {
        if (mat.data == NULL) return std::shared_ptr<image_t>(NULL);
        cv::Mat det_mat;
        cv::resize(mat, det_mat, cv::Size(get_net_width(), get_net_height()));
        return mat_to_image(det_mat);
    }

    // 8-bit interleaved BGR(A) -> planar float, value / 255., channels 0 and 2 exchanged (hpp:85-92,96-141); the image
    // frees itself with the last shared_ptr
    static std::shared_ptr<image_t> mat_to_image(cv::Mat img)
    {
        std::shared_ptr<image_t> image_ptr(new image_t, [](image_t *im) { free_image(*im); delete im; });
        const int h = img.rows, w = img.cols, c = img.channels();
        image_ptr->h = h; image_ptr->w = w; image_ptr->c = c;
        image_ptr->data = (float *)calloc((size_t)h * w * c, sizeof(float));
        if (!image_ptr->data) throw std::runtime_error("mat_to_image: out of memory");
        for (int k = 0; k < c; ++k) {
            const int plane = (c >= 3 && k < 3) ? 2 - k : k;       // BGR -> RGB
            for (int y = 0; y < h; ++y) {
                const unsigned char *row = img.ptr<unsigned char>(y);
                float *dst = image_ptr->data + ((size_t)plane * h + y) * w;
                for (int x = 0; x < w; ++x) dst[x] = (float)(row[x * c + k] / 255.);
            }
        }
        return image_ptr;
    }
#endif

private:
    std::deque<std::vector<bbox_t>> prev_bbox_vec_deque;
};
