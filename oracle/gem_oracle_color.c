/*
 * gem_oracle_color.c -- CPU ORACLE (test infrastructure only; see gem_oracle.h) for the step IN FRONT of the hot path: the
 * input colourisation loop of ElevationMapping::Callback (elevation_mapping/src/ElevationMapping.cpp:349-381; "EMg.cpp").
 *
 *   EMg.cpp:342-345  P_lidar2img = Tcamera (3x4) * TLidar (4x4), Eigen::MatrixXd (doubles)          -> gemo_lidar_to_image
 *   EMg.cpp:350-355  P_img = P_lidar2img * (x, y, z, 1), doubles
 *   EMg.cpp:357-358  P_x = P_img.x / P_img.z, P_y = P_img.y / P_img.z, stored in FLOATS
 *   EMg.cpp:360-363  cv::Point midPoint; midPoint.x = P_x: int members, the float is truncated
 *   EMg.cpp:366      sampled iff 0 < x < width && 0 < y < height && P_img.z > 0
 *   EMg.cpp:367-369  b, g, r = img.at<cv::Vec3b>(y, x)
 *   EMg.cpp:370      cv::circle(img, midPoint, 1, cv::Scalar(b, g, r)) -- drawn INTO the image the later points sample
 *   EMg.cpp:371-373  the point takes b, g, r
 *   EMg.cpp:375-380  otherwise b = g = r = 0 and intensity = 0
 *
 * OpenCV is a third-party dependency (package.xml: cv_bridge; version unpinned, not vendored under /root/reference and not
 * installed in this image).  cv::circle with thickness 1, LINE_8, shift 0 runs the integer midpoint rasteriser of
 * modules/imgproc/src/drawing.cpp (Circle(), unchanged from 2.4 to 4.x), restated below for any radius: for radius 1 its single
 * round (dx = 1, dy = 0) stores the four edge neighbours of the centre, each one only if it lies inside the image; the centre is
 * not drawn.  The Eigen products are restated with the sums running k = 0, 1, 2, 3 (Eigen's own order depends on its version and
 * on alignment).  Nothing in the reference pins these outputs: PARITY UNPINNED beyond this restatement and the hand-derived
 * checks in tests/test_colorize.py.
 */
#include "gem_oracle.h"

#include <limits.h>
#include <string.h>

/* EMg.cpp:343: out (3x4) = Tcamera (3x4) * TLidar (4x4), all row-major here */
void gemo_lidar_to_image(const double tcamera[12], const double tlidar[16], double out[12])
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
            double acc = tcamera[4 * r] * tlidar[c];
            for (int k = 1; k < 4; ++k) acc = acc + tcamera[4 * r + k] * tlidar[4 * k + c];
            out[4 * r + c] = acc;
        }
}

static void put_clipped(unsigned char* img, int width, int height, size_t stride, int x, int y, const unsigned char bgr[3])
{
    if (x < 0 || x >= width || y < 0 || y >= height) return;
    memcpy(img + (size_t)y * stride + (size_t)x * 3, bgr, 3);
}

/* outline of the midpoint circle: per round the eight symmetric points (+-dx, +-dy), (+-dy, +-dx), each clipped to the image */
static void circle_outline(unsigned char* img, int width, int height, size_t stride, int cx, int cy, int radius, const unsigned char bgr[3])
{
    int err = 0, dx = radius, dy = 0, plus = 1, minus = 2 * radius - 1;
    while (dx >= dy) {
        put_clipped(img, width, height, stride, cx - dx, cy - dy, bgr); put_clipped(img, width, height, stride, cx - dx, cy + dy, bgr);
        put_clipped(img, width, height, stride, cx + dx, cy - dy, bgr); put_clipped(img, width, height, stride, cx + dx, cy + dy, bgr);
        put_clipped(img, width, height, stride, cx - dy, cy - dx, bgr); put_clipped(img, width, height, stride, cx - dy, cy + dx, bgr);
        put_clipped(img, width, height, stride, cx + dy, cy - dx, bgr); put_clipped(img, width, height, stride, cx + dy, cy + dx, bgr);
        ++dy; err += plus; plus += 2;
        if (err > 0) { err -= minus; --dx; minus -= 2; }
    }
}

/* float -> int like the x86 conversion the reference compiles to (cvttss2si): out-of-range and NaN give INT_MIN */
static int trunc_to_int(float v)
{
    if (!(v > -2147483648.0f && v < 2147483648.0f)) return INT_MIN;
    return (int)v;
}

/* The loop of EMg.cpp:349-381.  image_bgr (height rows of `stride` bytes, BGR8) IS DRAWN ON, like the reference's copy of the
 * camera image; xyzi (n x 4 floats) gets intensity 0 for the points outside; rgb[i] = 0x00RRGGBB.  Returns the number coloured. */
int gemo_colorize(const double P[12], int width, int height, unsigned char* image_bgr, size_t stride, int n, float* xyzi, uint32_t* rgb)
{
    int coloured = 0;
    for (int i = 0; i < n; ++i) {
        const double v[3] = {(double)xyzi[4 * i], (double)xyzi[4 * i + 1], (double)xyzi[4 * i + 2]};
        double r[3];
        for (int k = 0; k < 3; ++k) r[k] = ((P[4 * k] * v[0] + P[4 * k + 1] * v[1]) + P[4 * k + 2] * v[2]) + P[4 * k + 3];
        const float px = (float)(r[0] / r[2]), py = (float)(r[1] / r[2]);
        const int x = trunc_to_int(px), y = trunc_to_int(py);
        if (x > 0 && x < width && y > 0 && y < height && r[2] > 0.0) {
            unsigned char bgr[3];
            memcpy(bgr, image_bgr + (size_t)y * stride + (size_t)x * 3, 3);
            circle_outline(image_bgr, width, height, stride, x, y, 1, bgr);
            rgb[i] = ((uint32_t)bgr[2] << 16) | ((uint32_t)bgr[1] << 8) | (uint32_t)bgr[0];
            ++coloured;
        } else {
            rgb[i] = 0u;
            xyzi[4 * i + 3] = 0.0f;
        }
    }
    return coloured;
}
