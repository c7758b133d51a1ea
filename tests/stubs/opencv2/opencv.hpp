// SYNTAX PROBE ONLY -- not a build of the reference and not an OpenCV replacement.
// tests/test_facade_compile.py runs `g++ -fsyntax-only` over the reference's homo/fhe_resize.h and
// homo/fhe_decode.h (which include <opencv2/opencv.hpp> for debug/compare helpers unrelated to the
// ciphertext path) to show that seal/seal.h declares every SEAL name those headers use.  Nothing
// is linked or executed; the declarations below only name what the reference's helper code mentions.
#pragma once
#include <string>
namespace cv {
typedef std::string String;
struct Size { Size(int, int) {} };
struct Vec3b { unsigned char val[3]; unsigned char &operator[](int i) { return val[i]; } };
struct Mat {
    int rows, cols;
    unsigned char *data;
    Mat() : rows(0), cols(0), data(0) {}
    Mat(int, int, int) : rows(0), cols(0), data(0) {}
    Mat(int, int, int, void *p) : rows(0), cols(0), data((unsigned char *)p) {}
    template <class T> T &at(int, int) { static T t; return t; }
    Mat clone() const { return *this; }
};
enum { IMREAD_COLOR = 1, INTER_LINEAR = 1, INTER_CUBIC = 2, WINDOW_AUTOSIZE = 1, CV_8UC3 = 16, CV_IMWRITE_PNG_COMPRESSION = 16 };
inline Mat imread(const String &, int = 1) { return Mat(); }
template <class V> inline bool imwrite(const String &, const Mat &, const V &) { return true; }
inline bool imwrite(const String &, const Mat &) { return true; }
inline void resize(const Mat &, Mat &, Size, double = 0, double = 0, int = 1) {}
inline void namedWindow(const String &, int = 1) {}
inline void imshow(const String &, const Mat &) {}
inline int waitKey(int = 0) { return 0; }
}  // namespace cv
