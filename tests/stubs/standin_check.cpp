// TEST INFRASTRUCTURE: dumps what tests/stubs/opencv2/opencv.hpp's imread / resize produce, for
// tests/test_opencv_standin.py to compare with Pillow's libjpeg-turbo decode and the published 113.692.
// usage: standin_check in.jpg out.raw [w h interp]   -> out.raw = u32 W, u32 H, BGR bytes (of the resized image if w h given)
#include <cstdio>
#include <cstdlib>
#include <opencv2/opencv.hpp>
int main(int argc, char **argv) {
    if (argc < 3) return 2;
    cv::Mat m = cv::imread(argv[1], cv::IMREAD_COLOR);
    if (m.empty()) { std::fprintf(stderr, "unsupported or unreadable JPEG\n"); return 3; }
    if (argc >= 5) {
        cv::Mat r;
        cv::resize(m, r, cv::Size(std::atoi(argv[3]), std::atoi(argv[4])), 0, 0, argc >= 6 ? std::atoi(argv[5]) : cv::INTER_LINEAR);
        m = r;
    }
    FILE *f = std::fopen(argv[2], "wb");
    if (!f) return 2;
    unsigned hdr[2] = {(unsigned)m.cols, (unsigned)m.rows};
    std::fwrite(hdr, 4, 2, f);
    std::fwrite(m.data, 1, (size_t)m.rows * m.cols * 3, f);
    std::fclose(f);
    return 0;
}
