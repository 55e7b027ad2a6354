/*
 * gem_oracle_raytrace.c -- CPU ORACLE (test infrastructure only; see gem_oracle.h): the visibility clean-up.
 *
 * Restates Raytracing (GPU:1304-1318) = G_Raytracing (GPU:708-891, helpers GPU:672-706) followed by
 * G_Clear_maplowest (GPU:232-239).  GPU = elevation_mapping/elevation_mapping/cuda/gpu_process.cu.
 * Pinned against the reference's own code compiled for the CPU (tests/test_reference_compiled.py).
 *
 * What the kernel does, per cell that holds an elevation and whose traversability is below obstacle_threshold: walk
 * from the cell AWAY from the map centre along the centre->cell ray (a DDA over cell borders), and for every crossed
 * cell that has a lowest scan point this frame bound the obstacle's height by the line of sight from the sensor
 * (sensorZatLowestScan above the map centre) over that point; if elevation - 3 sigma is above the tightest bound,
 * the cell is deleted (elevation = -10; the variance stays).  Quirks kept: map_lowest is indexed by the GEOGRAPHIC
 * cell; robot_index is declared int, so the even-L centre 299.5 becomes 299 (GPU:720,733); cells on the centre row or
 * column compute their bound and return without applying it (GPU:760-791); both line-of-sight abscissae use the x
 * index only (GPU:693-694); "no scan point" is lowest == 10 (after G_Clear_maplowest), so right after Init (lowest =
 * 100) every cell counts as scanned.
 * A thread writes only its own cell's elevation and reads only its own cell's elevation / variance, so the kernel's
 * result does not depend on the schedule.
 */
#include "gem_oracle.h"

#include <math.h>

void gemo_set_obstacle_threshold(gemo_map* m, float t) { m->obstacle_threshold = t; }

/* GPU:681-689 */
static int p_is_valid(const gemo_map* m, int gx, int gy) { return m->lowest[gx * m->L + gy] != 10.0f; }

/* GPU:691-706 */
static float d_min_elevation(const gemo_map* m, int gx, int gy, int obstacle_x, float robot_index_x)
{
    float x1 = (float)(gx - obstacle_x);
    float x2 = (float)gx - robot_index_x;
    float low = m->lowest[gx * m->L + gy];
    float h2 = m->sensor_z - low;
    return low + h2 / x2 * x1;
}

void gemo_raytracing(gemo_map* m)
{
    const int L = m->L;
    for (int i = 0; i < L * L; ++i) {
        if (!(m->traver[i] < m->obstacle_threshold && m->elevation[i] != -10.0f)) continue;      /* GPU:712 */
        const int cell_x = i / L, cell_y = i % L;
        int robot_index;
        int ob[2];
        ob[0] = (cell_x + L - m->start[0]) % L;                                                    /* GPU:672-675 */
        ob[1] = (cell_y + L - m->start[1]) % L;
        const float obstacle_ele = m->elevation[i];
        int cur[2] = { ob[0], ob[1] };
        float inc[2];
        int inc_x, inc_y;
        if (L % 2 == 0) robot_index = (int)(float)(L / 2 - 0.5);                                   /* GPU:733: float -> int */
        else            robot_index = (int)(float)(L / 2);                                         /* GPU:739 */
        inc[0] = (float)(ob[0] - robot_index);
        inc[1] = (float)(ob[1] - robot_index);
        inc_x = inc[0] > 0 ? 1 : (inc[0] == 0 ? 0 : -1);                                           /* GPU:744-756 */
        inc_y = inc[1] > 0 ? 1 : (inc[1] == 0 ? 0 : -1);

        float restrict_ele = obstacle_ele;
        /* GPU:760-791: on the centre row / column the bound is computed and thrown away (every path returns) */
        if (inc_x == 0 || inc_y == 0) continue;

        float dis = sqrtf(inc[0] * inc[0] + inc[1] * inc[1]);                                      /* GPU:793 */
        float dir[2] = { inc[0] / dis, inc[1] / dis };
        float threshold;                                                                           /* GPU:798-802: double arithmetic */
        if (fabsf(inc[0]) > fabsf(inc[1])) threshold = (float)sqrt(0.5 * 0.5 + pow(0.5 / inc[0] * inc[1], 2));
        else                               threshold = (float)sqrt(0.5 * 0.5 + pow(0.5 / inc[1] * inc[0], 2));

        float bound_x = (float)inc_x / 2, bound_y = (float)inc_y / 2;                             /* GPU:808-809 */
        float dir_num_x = bound_x / dir[0], dir_num_y = bound_y / dir[1];
        float later = 0;
        while (cur[0] >= 0 && cur[0] < L && cur[1] >= 0 && cur[1] < L) {                           /* GPU:819-880 */
            const int crossed = cur[0] != ob[0] && cur[1] != ob[1];
            if (dir_num_x > dir_num_y) {
                if (dir_num_y - later > threshold && crossed && p_is_valid(m, cur[0], cur[1])) {
                    float e = d_min_elevation(m, cur[0], cur[1], ob[0], (float)robot_index);
                    if (e < restrict_ele) restrict_ele = e;
                }
                cur[1] += inc_y; bound_y += (float)inc_y; later = dir_num_y; dir_num_y = bound_y / dir[1];
            } else if (dir_num_x < dir_num_y) {
                if (dir_num_x - later > threshold && crossed && p_is_valid(m, cur[0], cur[1])) {
                    float e = d_min_elevation(m, cur[0], cur[1], ob[0], (float)robot_index);
                    if (e < restrict_ele) restrict_ele = e;
                }
                cur[0] += inc_x; bound_x += (float)inc_x; later = dir_num_x; dir_num_x = bound_x / dir[0];
            } else {
                if (dir_num_x - later > threshold && crossed && p_is_valid(m, cur[0], cur[1])) {
                    float e = d_min_elevation(m, cur[0], cur[1], ob[0], (float)robot_index);
                    if (e < restrict_ele) restrict_ele = e;
                }
                cur[0] += inc_x; cur[1] += inc_y; bound_x += (float)inc_x; bound_y += (float)inc_y;
                later = dir_num_x; dir_num_x = bound_x / dir[0]; dir_num_y = bound_y / dir[1];
            }
        }
        if (obstacle_ele - 3 * sqrtf(m->variance[i]) > restrict_ele) m->elevation[i] = -10.0f;      /* GPU:884-885 */
    }
    for (int i = 0; i < L * L; ++i) m->lowest[i] = 10.0f;                                          /* GPU:232-239 */
}
