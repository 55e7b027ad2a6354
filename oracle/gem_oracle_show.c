/*
 * gem_oracle_show.c -- CPU ORACLE (test infrastructure only; see gem_oracle.h) for the step AFTER the hot path:
 * ElevationMap::show (elevation_mapping/src/ElevationMap.cpp:85-149; "EM.cpp"), the cell loop that turns the nine flat
 * [storage_x * L + storage_y] arrays into the visualMap_ grid_map layers, the point cloud and the orthomosaic.
 *
 *   EM.cpp:89      visualMap_.clearAll()                      -> every layer NaN
 *   EM.cpp:97      GridMapIterator: linear index over the Eigen (column-major) matrix, *iterator = BUFFER index
 *                  (grid_map_core GridMapIterator.cpp: getIndexFromLinearIndex -> (lin % rows, lin / rows))
 *   EM.cpp:98-100  index = index_x * length + index_y with the buffer index
 *   EM.cpp:101     kept iff elevation != -10 && traver != -10 && !isnan(traver)
 *   EM.cpp:103-111 nine layers copied (colours int -> float)
 *   EM.cpp:113-114 getPosition(): grid_map_core GridMapMath.cpp getPositionFromIndex
 *                      position = mapPosition + (0.5 * mapLength - 0.5 * resolution) - resolution * unwrapped_index   (doubles)
 *                      unwrapped = (buffer - start) wrapped into [0, size)
 *   EM.cpp:116-121 point {float x, y, z; uint8 r, g, b} (colours: float -> uint8_t), pushed in iteration order
 *   EM.cpp:124-126 image(unwrapped row, unwrapped col) = {b, g, r}
 *
 * grid_map is a third-party dependency (package.xml: grid_map_core / grid_map_ros, version unpinned, not vendored under
 * /root/reference); the two functions used are restated from its published source (ANYbotics grid_map 1.6.x), which has not
 * changed them since 1.4.  Nothing in the reference pins these outputs: PARITY UNPINNED beyond this restatement and the
 * hand-derived checks in tests/test_show.py.
 */
#include "gem_oracle.h"

#include <math.h>
#include <string.h>

/* visual: 9 layers x L*L floats, Eigen column-major ([col * L + row]), order: elevation, variance, rough, slope, traver, color_r,
 * color_g, color_b, intensity (EM.cpp:44 + :103-111).  points_xyz: up to L*L x 3 floats, points_rgb: x 3 bytes.  image_bgr: L*L*3 bytes
 * (row-major [row][col][3], zero-initialised here like cv::Mat(..., Scalar(0,0,0))).  Any output may be NULL.  Returns the number of points. */
int gemo_show(const gemo_map* m, const float* rough, const float* slope, double map_length, double resolution, const double map_position[2],
              float* visual, float* points_xyz, unsigned char* points_rgb, unsigned char* image_bgr)
{
    const int L = m->L;
    const size_t cells = (size_t)L * (size_t)L;
    if (visual) for (size_t i = 0; i < 9 * cells; ++i) visual[i] = NAN;                         /* EM.cpp:89 */
    if (image_bgr) memset(image_bgr, 0, cells * 3);
    const double off = 0.5 * map_length - 0.5 * resolution;                                     /* getVectorToFirstCell */
    int n = 0;
    for (size_t lin = 0; lin < cells; ++lin) {                                                  /* EM.cpp:97 */
        const int ix = (int)(lin % (size_t)L), iy = (int)(lin / (size_t)L);                     /* buffer index of the column-major linear index */
        const size_t index = (size_t)ix * L + iy;                                               /* EM.cpp:100 */
        const float tr = m->traver[index];
        if (!(m->elevation[index] != -10.0f && tr != -10.0f && !isnan(tr))) continue;           /* EM.cpp:101 */
        const float cr = (float)m->colorR[index], cg = (float)m->colorG[index], cb = (float)m->colorB[index];
        if (visual) {
            const float vals[9] = {m->elevation[index], m->variance[index], rough ? rough[index] : 0.0f, slope ? slope[index] : 0.0f, tr,
                                   cr, cg, cb, m->intensity[index]};
            for (int l = 0; l < 9; ++l) visual[(size_t)l * cells + lin] = vals[l];
        }
        int ux = ix - m->start[0], uy = iy - m->start[1];                                        /* getIndexFromBufferIndex */
        if (ux < 0) ux += L;
        if (uy < 0) uy += L;
        if (points_xyz) {
            const double px = (map_position[0] + off) + resolution * (double)(-ux);             /* getPositionFromIndex */
            const double py = (map_position[1] + off) + resolution * (double)(-uy);
            points_xyz[3 * n + 0] = (float)px; points_xyz[3 * n + 1] = (float)py; points_xyz[3 * n + 2] = m->elevation[index];
        }
        /* float -> uint8_t (EM.cpp:118-120) is only defined for 0..255, which is what colours are; outside it this restatement (and the
         * device kernel) goes through int and keeps the low byte */
        if (points_rgb) { points_rgb[3 * n + 0] = (unsigned char)(int)cr; points_rgb[3 * n + 1] = (unsigned char)(int)cg; points_rgb[3 * n + 2] = (unsigned char)(int)cb; }
        if (image_bgr) {
            unsigned char* px = image_bgr + ((size_t)ux * L + uy) * 3;                          /* EM.cpp:124-126 */
            px[0] = (unsigned char)(int)cb; px[1] = (unsigned char)(int)cg; px[2] = (unsigned char)(int)cr;
        }
        ++n;
    }
    return n;
}
