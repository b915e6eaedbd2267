// A 60-line stand-in for <opencv2/opencv.hpp> so that the OPENCV part of include/yolo_v2_class.hpp can be compiled and
// exercised in a container without OpenCV: cv::Size, an 8-bit interleaved cv::Mat and a nearest-neighbour cv::resize.
// TEST INFRASTRUCTURE ONLY (tests/native/console_dll_like.cpp); nothing in the product includes it.
#pragma once
#include <cstddef>
#include <memory>
#include <vector>

namespace cv {

struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
};

class Mat {
    std::shared_ptr<std::vector<unsigned char>> store;
    int ch;
public:
    int rows, cols;
    unsigned char *data;
    Mat() : ch(0), rows(0), cols(0), data(NULL) {}
    Mat(int r, int c, int channels) : store(new std::vector<unsigned char>((size_t)r * c * channels)), ch(channels), rows(r), cols(c)
    {
        data = store->data();
    }
    int channels() const { return ch; }
    Size size() const { return Size(cols, rows); }
    bool empty() const { return data == NULL; }
    template <typename T> T *ptr(int y) { return (T *)(data + (size_t)y * cols * ch); }
    template <typename T> const T *ptr(int y) const { return (const T *)(data + (size_t)y * cols * ch); }
};

inline void resize(const Mat &src, Mat &dst, Size sz)
{
    dst = Mat(sz.height, sz.width, src.channels());
    for (int y = 0; y < sz.height; ++y)
        for (int x = 0; x < sz.width; ++x) {
            const int sy = (int)((long)y * src.rows / sz.height), sx = (int)((long)x * src.cols / sz.width);
            for (int k = 0; k < src.channels(); ++k)
                dst.ptr<unsigned char>(y)[x * src.channels() + k] = src.ptr<unsigned char>(sy)[sx * src.channels() + k];
        }
}

}  // namespace cv
