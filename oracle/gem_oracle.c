/*
 * gem_oracle.c -- CPU ORACLE (test infrastructure only; see gem_oracle.h).
 *
 * Plain C restatement of the reference's CUDA hot path.  Every function cites the reference
 * lines it follows (GPU = elevation_mapping/elevation_mapping/cuda/gpu_process.cu).
 * Pinned against the reference's own gpu_process.cu compiled for the CPU (oracle/ref_build, tests/test_reference_compiled.py)
 * and by our own KATs; the reference itself has no tests / golden vectors.  See gem_oracle.h.
 *
 * Third-party arithmetic restated here: Eigen fixed-size products / norm used inside
 * G_pointsprocess (GPU:403-425).  Eigen is not vendored and its version is unpinned
 * (README.md:82 mentions 3.2.9 / 3.3.4).  Eigen >= 3.2.92 evaluates a fixed-size 3-term
 * coefficient product as  sum() -> redux_novec_unroller<.,0,3>  =  c0 + (c1 + c2)
 * (Eigen/src/Core/Redux.h, HalfLength = Length/2), which is what dot3() below does.
 * Explicit scalar expressions in the reference (GPU:389,399-400,297) are left-to-right.
 *
 * Compile: gcc -O2 -ffp-contract=off -fno-fast-math -std=c11
 */
#include "gem_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define EMPTY_ELEV (-10.0f)

/* ---------------------------------------------------------------- state ---------- */

/* GPU:198-214 (G_Init_map) + GPU:972-973 (centre/start = 0) */
gemo_map* gemo_create(int length, float resolution, float mahalanobis, float var_floor)
{
    gemo_map* m = (gemo_map*)calloc(1, sizeof(gemo_map));
    if (!m) return NULL;
    size_t n = (size_t)length * (size_t)length;
    m->L = length; m->res = resolution; m->mahal = mahalanobis; m->var_floor = var_floor;
    m->elevation = (float*)malloc(n * sizeof(float));
    m->variance  = (float*)malloc(n * sizeof(float));
    m->intensity = (float*)malloc(n * sizeof(float));
    m->traver    = (float*)malloc(n * sizeof(float));
    m->lowest    = (float*)malloc(n * sizeof(float));
    m->colorR = (int*)malloc(n * sizeof(int));
    m->colorG = (int*)malloc(n * sizeof(int));
    m->colorB = (int*)malloc(n * sizeof(int));
    for (size_t i = 0; i < n; ++i) {
        m->intensity[i] = 0.0f;
        m->elevation[i] = EMPTY_ELEV;
        m->variance[i]  = -10.0f;
        m->lowest[i]    = 100.0f;
        m->traver[i]    = -10.0f;
        m->colorR[i] = m->colorG[i] = m->colorB[i] = 0;
    }
    m->center[0] = m->center[1] = 0.0f;
    m->start[0] = m->start[1] = 0;
    m->sensor_z = 0.0f;
    m->obstacle_threshold = 0.7f;
    return m;
}

void gemo_destroy(gemo_map* m)
{
    if (!m) return;
    free(m->elevation); free(m->variance); free(m->intensity); free(m->traver); free(m->lowest);
    free(m->colorR); free(m->colorG); free(m->colorB);
    free(m);
}

/* ---------------------------------------------------------------- move ----------- */

/* GPU:914-919 */
static int index_to_range(int index, int L)
{
    if (index < 0) index += ((-index / L) + 1) * L;
    return index % L;
}

/* GPU:255-276 (G_Clear_map): rows -> [start*L, start*L + shift*L); cols -> L rows x shift cols.
 * Clears intensity, elevation, variance, colours; NOT traver / lowest. */
static void clear_rows(gemo_map* m, int start, int shift)
{
    int L = m->L;
    for (int i = 0; i < L * shift; ++i) {
        int c = start * L + i;
        m->intensity[c] = 0.0f; m->elevation[c] = EMPTY_ELEV; m->variance[c] = -10.0f;
        m->colorR[c] = m->colorG[c] = m->colorB[c] = 0;
    }
}
static void clear_cols(gemo_map* m, int start, int shift)
{
    int L = m->L;
    for (int i = 0; i < L * shift; ++i) {
        int c = i / shift * L + i % shift + start;
        m->intensity[c] = 0.0f; m->elevation[c] = EMPTY_ELEV; m->variance[c] = -10.0f;
        m->colorR[c] = m->colorG[c] = m->colorB[c] = 0;
    }
}
/* GPU:216-230 (G_Clear_allmap): also resets traver, not lowest */
static void clear_all(gemo_map* m)
{
    size_t n = (size_t)m->L * m->L;
    for (size_t c = 0; c < n; ++c) {
        m->intensity[c] = 0.0f; m->elevation[c] = EMPTY_ELEV; m->variance[c] = -10.0f;
        m->traver[c] = -10.0f;
        m->colorR[c] = m->colorG[c] = m->colorB[c] = 0;
    }
}

/* GPU:996-1002 (PositionToRange): int = round(float/float), result int*float */
static float position_to_range(float p, float shift, float res)
{
    int p_index = (int)roundf(p / res);
    int shift_index = (int)roundf(shift / res);
    int current = p_index + shift_index;
    return (float)current * res;
}

/* GPU:1004-1083 (Move) with helpers GPU:893-912.
 * Deviation (documented): for |indexShift| >= L in the NEGATIVE direction the reference runs
 * G_Clear_map past the end of the arrays (GPU:1053-1066 with nCells > L); we clear the whole
 * map instead, as the reference does for the positive direction (GPU:1034-1038). */
int gemo_move(gemo_map* m, const float pos[3], float out_center[2], int out_start[2], float out_shift[2])
{
    int L = m->L; float res = m->res;
    int launches = 0;
    float position_shift[2]; int index_shift[2]; float aligned[2];
    m->sensor_z = pos[2];
    for (int i = 0; i < 2; ++i) {
        position_shift[i] = pos[i] - m->center[i];
        /* GPU:897: float/float, then + (double)0.5*sign in double, truncation */
        index_shift[i] = (int)((double)(position_shift[i] / res) + 0.5 * (position_shift[i] > 0 ? 1 : -1));
        aligned[i] = (float)index_shift[i] * res;             /* GPU:909 */
    }
    for (int i = 0; i < 2; ++i) {
        if (index_shift[i] != 0) {
            if (index_shift[i] >= L || index_shift[i] <= -L) {
                clear_all(m); ++launches;
            } else {
                int sign = index_shift[i] > 0 ? 1 : -1;
                int start_index = m->start[i] - (sign > 0 ? 1 : 0);
                int end_index = start_index + sign - index_shift[i];
                int n_cells = abs(index_shift[i]);
                int index = sign < 0 ? start_index : end_index;
                index = index_to_range(index, L);
                if (index + n_cells <= L) {
                    if (i == 0) clear_rows(m, index, n_cells); else clear_cols(m, index, n_cells);
                    ++launches;
                } else {
                    int first_n = L - index;
                    if (i == 0) clear_rows(m, index, first_n); else clear_cols(m, index, first_n);
                    int second_n = n_cells - first_n;
                    if (i == 0) clear_rows(m, 0, second_n); else clear_cols(m, 0, second_n);
                    launches += 2;
                }
            }
        }
        m->start[i] -= index_shift[i];
        m->start[i] = index_to_range(m->start[i], L);
        m->center[i] = position_to_range(m->center[i], aligned[i], res);
    }
    out_center[0] = m->center[0]; out_center[1] = m->center[1];
    out_start[0] = m->start[0];   out_start[1] = m->start[1];
    out_shift[0] = aligned[0];    out_shift[1] = aligned[1];
    return launches;
}

/* ---------------------------------------------------------------- binning -------- */

/* One axis of GPU:315-323.  Returns INT_MIN-like sentinel (-1 is enough: caller range-checks)
 * for values the C cast cannot represent (map to "outside" before the cast). */
static int axis_index(int L, float res, float shift)
{
    if (L % 2 == 0) {
        float v = (float)(L / 2) - shift / res;                 /* GPU:316 float arithmetic */
        if (!(v > -2147483648.0f && v < 2147483648.0f)) return -1;
        return (int)v;                                          /* truncation toward zero */
    } else {
        double v = (double)(shift / res) + 0.5 * (shift > 0 ? 1 : -1);   /* GPU:321 double sum */
        if (!(v > -2147483648.0 && v < 2147483648.0)) return -1;
        return L / 2 - (int)v;
    }
}

/* GPU:309-330 */
int gemo_points_to_index(const gemo_map* m, float px, float py)
{
    int L = m->L;
    float sx = px - m->center[0], sy = py - m->center[1];
    int ix = axis_index(L, m->res, sx), iy = axis_index(L, m->res, sy);
    if (ix >= 0 && ix < L && iy >= 0 && iy < L) return ix * L + iy;
    return -1;
}

/* GPU:332-358 */
int gemo_points_to_map_index(const gemo_map* m, float px, float py)
{
    int L = m->L;
    float sx = px - m->center[0], sy = py - m->center[1];
    int ix = axis_index(L, m->res, sx), iy = axis_index(L, m->res, sy);
    if (ix >= 0 && ix < L && iy >= 0 && iy < L) {
        int storage_x = (ix + m->start[0]) % L;
        int storage_y = (iy + m->start[1]) % L;
        return storage_x * L + storage_y;
    }
    return -1;
}

/* ---------------------------------------------------------------- variance ------- */

/* Eigen redux order for a 3-term sum: c0 + (c1 + c2) (see file header). */
static float dot3(float a0, float b0, float a1, float b1, float a2, float b2)
{
    float c0 = a0 * b0, c1 = a1 * b1, c2 = a2 * b2;
    return c0 + (c1 + c2);
}

/* sensor-model part: (varianceNormal, varianceLateral) */
static void sensor_variances(const gemo_frame* f, float x, float y, float z, int orig, float* vn, float* vl)
{
    const double* sp = f->sp;
    switch (f->sensor_model) {
    default:
    case GEMO_MODEL_LASER: {
        /* GPU:404-408: distance = Eigen norm() = sqrt(x^2 + (y^2 + z^2)) */
        float d = sqrtf(dot3(x, x, y, y, z, z));
        float min_r = (float)sp[0], beam_a = (float)sp[1], beam_c = (float)sp[2];   /* SPB.cpp:286-288 */
        *vn = min_r * min_r;                               /* pow(C_min_r, 2) */
        float t = beam_c + beam_a * d;
        *vl = t * t;
        break; }
    case GEMO_MODEL_STRUCTURED_LIGHT: {
        /* SL.cpp:128-139: distance = z; deviationNormal = a + b (z-c)(z-c) + d * pow(z, e).
         * sensorParameters_ is std::map<string,double>: the expression is evaluated in DOUBLE and
         * rounded to float once. */
        double zd = (double)z;
        double a = sp[0], b = sp[1], c = sp[2], dd = sp[3], e = sp[4], lat = sp[5];
        float dev_n = (float)(a + b * (zd - c) * (zd - c) + dd * pow(zd, e));
        *vn = dev_n * dev_n;
        float dev_l = (float)(lat * zd);
        *vl = dev_l * dev_l;
        break; }
    case GEMO_MODEL_STEREO: {
        /* Stereo.cpp:78-92 (all double, rounded to float on assignment) */
        double p1 = sp[0], p2 = sp[1], p3 = sp[2], p4 = sp[3], p5 = sp[4], lat = sp[5], f2d = sp[6];
        int w = f->original_width > 0 ? f->original_width : 1;
        int I = orig / w, J = orig % w;                    /* Stereo.cpp:108-116 */
        double disparity = f2d / (double)z;
        float d = sqrtf(dot3(x, x, y, y, z, z));
        double t = p3 * disparity + p4 - (double)J;
        double u = 240.0 - (double)I;
        double g = f2d / (disparity * disparity);
        *vn = (float)(g * g * ((p5 * disparity + p2) * sqrt(t * t + u * u) + p1));
        double l = lat * (double)d;
        *vl = (float)(l * l);
        break; }
    case GEMO_MODEL_PERFECT:
        *vn = 0.0f; *vl = 0.0f;                            /* Perfect.cpp:86-88 */
        break;
    }
}

/* GPU:403-425: heightVariance = Jq Sigma_q Jq^T + Js diag(vl,vl,vn) Js^T */
static float height_variance(const gemo_frame* f, float x, float y, float z, int orig)
{
    float vn, vl;
    sensor_variances(f, x, y, z, orig, &vn, &vl);

    const float* C = f->C_SB_T; const float* P = f->P_mul_C_BM_T; const float* Bs = f->B_r_BS_skew;
    /* q = C_SB_T * p   (3x3 * 3x1, Eigen coefficient product) */
    float q0 = dot3(C[0], x, C[1], y, C[2], z);
    float q1 = dot3(C[3], x, C[4], y, C[5], z);
    float q2 = dot3(C[6], x, C[7], y, C[8], z);
    /* S = skew(q) + B_r_BS_skew   (GPU:302-307: [[0,-q2,q1],[q2,0,-q0],[-q1,q0,0]]) */
    float S[9];
    S[0] = 0.0f + Bs[0]; S[1] = -q2 + Bs[1];  S[2] = q1 + Bs[2];
    S[3] = q2 + Bs[3];   S[4] = 0.0f + Bs[4]; S[5] = -q0 + Bs[5];
    S[6] = -q1 + Bs[6];  S[7] = q0 + Bs[7];   S[8] = 0.0f + Bs[8];
    /* Jq = P * S   (1x3 * 3x3) */
    float Jq0 = dot3(P[0], S[0], P[1], S[3], P[2], S[6]);
    float Jq1 = dot3(P[0], S[1], P[1], S[4], P[2], S[7]);
    float Jq2 = dot3(P[0], S[2], P[1], S[5], P[2], S[8]);
    /* cuda_computer(Jq, Sigma_q, Jq^T)  GPU:293-298: A1 = A*B (Eigen), then explicit left-to-right dot */
    const float* Q = f->rotation_variance;
    float a0 = dot3(Jq0, Q[0], Jq1, Q[3], Jq2, Q[6]);
    float a1 = dot3(Jq0, Q[1], Jq1, Q[4], Jq2, Q[7]);
    float a2 = dot3(Jq0, Q[2], Jq1, Q[5], Jq2, Q[8]);
    float hv = a0 * Jq0 + a1 * Jq1 + a2 * Jq2;
    /* cuda_computer(Js, diag(vl,vl,vn), Js^T) */
    const float* Js = f->sensor_jacobian;
    float b0 = dot3(Js[0], vl, Js[1], 0.0f, Js[2], 0.0f);
    float b1 = dot3(Js[0], 0.0f, Js[1], vl, Js[2], 0.0f);
    float b2 = dot3(Js[0], 0.0f, Js[1], 0.0f, Js[2], vn);
    hv += b0 * Js[0] + b1 * Js[1] + b2 * Js[2];
    return hv;
}

/* ---------------------------------------------------------------- process -------- */

/* one point of GPU:384-455.  Returns 1 if accepted. */
static int process_one(const gemo_map* m, const gemo_frame* f, float x, float y, float z, int orig,
                       int* map_index, float* var, float* xt, float* yt, float* zt)
{
    const float* T = f->T;
    float h = T[8] * x + T[9] * y + T[10] * z + T[11];                       /* GPU:389 */
    int flag = 0;
    if (f->filter_on) {                                                       /* GPU:393 */
        float bx = f->filter_box_x, by = f->filter_box_y, band = f->filter_band_y, plane = f->filter_plane_y;
        if ((x > -bx && x < bx && y > -by && y < by) || (y > -band && y < band) || y > plane) flag = 1;
    }
    if (((double)h > f->lower && (double)h < f->upper) && flag == 0) {        /* GPU:397 */
        *xt = T[0] * x + T[1] * y + T[2] * z + T[3];                          /* GPU:399 */
        *yt = T[4] * x + T[5] * y + T[6] * z + T[7];                          /* GPU:400 */
        *zt = h;
        *var = height_variance(f, x, y, z, orig);                             /* GPU:403-428 */
        *map_index = gemo_points_to_map_index(m, *xt, *yt);                   /* GPU:431 */
        return 1;
    }
    *map_index = -1; *xt = -1.0f; *yt = -1.0f; *zt = -1.0f; *var = -1.0f;     /* GPU:441-451 */
    return 0;
}

/* GPU:430-439: the lowest scan point of the (geographic) cell, with the reference's 3 * var (not 3 * sigma) margin */
static void lowest_update(gemo_map* m, float xt, float yt, float h, float var)
{
    int g = gemo_points_to_index(m, xt, yt);                                  /* GPU:430 */
    if (g == -1) return;
    m->lowest[g] = fminf(h, m->lowest[g]);                                    /* GPU:434 atomicMin (GPU:372-382) */
    if (h == m->lowest[g]) m->lowest[g] = m->lowest[g] + 3 * var;             /* GPU:435-438 */
}

int gemo_process_points(gemo_map* m, const gemo_frame* f, int n,
                        float* x, float* y, float* z, const int* orig_index,
                        int* map_index, float* var, float* x_ts, float* y_ts, float* z_ts)
{
    int accepted = 0;
    for (int i = 0; i < n; ++i) {
        int ok = process_one(m, f, x[i], y[i], z[i], orig_index ? orig_index[i] : i,
                             &map_index[i], &var[i], &x_ts[i], &y_ts[i], &z_ts[i]);
        if (!ok) { x[i] = -1.0f; y[i] = -1.0f; z[i] = -1.0f; }               /* GPU:443-446 */
        else lowest_update(m, x_ts[i], y_ts[i], z_ts[i], var[i]);
        accepted += ok;
    }
    return accepted;
}

/* ---------------------------------------------------------------- fuse ----------- */

/* body of the per-point branch of G_fuse for cell c (GPU:484-529) */
static void fuse_one(gemo_map* m, int c, float h, float v, int r, int g, int b, float inten)
{
    int colour_ok = (r != 0 && g != 0 && b != 0 && inten != 0.0f);
    if (m->elevation[c] == EMPTY_ELEV) {                                      /* GPU:484 */
        m->elevation[c] = h;
        m->variance[c] = v;
        if (colour_ok) { m->intensity[c] = inten; m->colorR[c] = r; m->colorG[c] = g; m->colorB[c] = b; }
        return;
    }
    if (m->variance[c] < m->var_floor) m->variance[c] = m->var_floor;        /* GPU:500-501 */
    float e = m->elevation[c], s = m->variance[c];
    float mahal = fabsf(h - e) / sqrtf(s);                                    /* GPU:502 */
    if (mahal > m->mahal) {                                                   /* GPU:504 */
        if (e < h) {                                                          /* GPU:505 */
            m->elevation[c] = h;
            m->variance[c] = v;
            if (colour_ok) { m->intensity[c] = inten; m->colorR[c] = r; m->colorG[c] = g; m->colorB[c] = b; }
        }
    } else {
        m->elevation[c] = (s * h + v * e) / (s + v);                          /* GPU:518 */
        m->variance[c]  = (v * s) / (v + s);                                  /* GPU:519 */
        if (colour_ok) { m->intensity[c] = inten; m->colorR[c] = r; m->colorG[c] = g; m->colorB[c] = b; }
    }
}

void gemo_fuse(gemo_map* m, int n, const int* index, const int* R, const int* G, const int* B,
               const float* intensity, const float* height, const float* var)
{
    int cells = m->L * m->L;
    for (int i = 0; i < n; ++i) {
        int c = index[i];
        if (c < 0 || c >= cells || height[i] == -1.0f) continue;              /* GPU:482 */
        fuse_one(m, c, height[i], var[i], R ? R[i] : 0, G ? G[i] : 0, B ? B[i] : 0,
                 intensity ? intensity[i] : 0.0f);
    }
    for (int c = 0; c < cells; ++c)                                           /* GPU:533-534: every cell */
        if (m->variance[c] < m->var_floor) m->variance[c] = m->var_floor;
}

void gemo_fuse_literal(gemo_map* m, int n, const int* index, const int* R, const int* G, const int* B,
                       const float* intensity, const float* height, const float* var)
{
    int cells = m->L * m->L;
    for (int c = 0; c < cells; ++c) {                 /* one reference "thread" per cell, GPU:478-479 */
        for (int i = 0; i < n; ++i) {                 /* GPU:480 */
            if (index[i] != c || height[i] == -1.0f) continue;
            fuse_one(m, c, height[i], var[i], R ? R[i] : 0, G ? G[i] : 0, B ? B[i] : 0,
                     intensity ? intensity[i] : 0.0f);
        }
        if (m->variance[c] < m->var_floor) m->variance[c] = m->var_floor;
    }
}

/* GPU:540-547 */
void gemo_mapvar_update(gemo_map* m, float var_update)
{
    int cells = m->L * m->L;
    for (int c = 0; c < cells; ++c)
        if (m->variance[c] != -10.0f) m->variance[c] += var_update;
}

/* ---------------------------------------------------------------- fused add ------ */

/* EMg.cpp:254-283 (processpoints): SensorProcessorBase::process -> Process_points, then Fuse,
 * on an interleaved XYZI cloud.  r,g,b are taken from packed 0x00RRGGBB (PointXYZRGBICT.hpp:26-48:
 * union { rgb; struct { b, g, r, a } }). */
int gemo_add(gemo_map* m, const gemo_frame* f, int n, const float* xyzi, const unsigned* rgb,
             const int* orig_index, long long counts[2])
{
    int cells = m->L * m->L;
    int accepted = 0;
    unsigned char* touched = counts ? (unsigned char*)calloc((size_t)cells, 1) : NULL;
    long long ntouched = 0;
    for (int i = 0; i < n; ++i) {
        float x = xyzi[4 * i + 0], y = xyzi[4 * i + 1], z = xyzi[4 * i + 2], inten = xyzi[4 * i + 3];
        int idx; float v, xt, yt, zt;
        int ok = process_one(m, f, x, y, z, orig_index ? orig_index[i] : i, &idx, &v, &xt, &yt, &zt);
        accepted += ok;
        if (ok) lowest_update(m, xt, yt, zt, v);
        if (idx < 0 || zt == -1.0f) continue;                                 /* GPU:482 */
        int r = 0, g = 0, b = 0;
        if (rgb) { r = (rgb[i] >> 16) & 0xff; g = (rgb[i] >> 8) & 0xff; b = rgb[i] & 0xff; }
        fuse_one(m, idx, zt, v, r, g, b, inten);
        if (touched && !touched[idx]) { touched[idx] = 1; ++ntouched; }
    }
    for (int c = 0; c < cells; ++c)
        if (m->variance[c] < m->var_floor) m->variance[c] = m->var_floor;
    if (counts) { counts[0] = accepted; counts[1] = ntouched; }
    free(touched);
    return accepted;
}


/* the single-point pieces, for the all-core form in gem_oracle_mt.c (same code, other loop structure) */
int gemo_process_one(const gemo_map* m, const gemo_frame* f, float x, float y, float z, int orig,
                     int* map_index, float* var, float* xt, float* yt, float* zt)
{
    return process_one(m, f, x, y, z, orig, map_index, var, xt, yt, zt);
}

void gemo_fuse_one(gemo_map* m, int c, float h, float v, int r, int g, int b, float inten)
{
    fuse_one(m, c, h, v, r, g, b, inten);
}

/* GPU:1195-1202 (G_update_mapheight): elevation += dz on the cells that hold one */
static void update_map_height(gemo_map* m, float dz)
{
    for (int i = 0; i < m->L * m->L; ++i) if (m->elevation[i] != -10.0f) m->elevation[i] += dz;
}

/* GPU:1215-1233 (Map_optmove, called EMg.cpp:1020 after a loop closure) with alignedPosition GPU:1203-1213:
 * the map CENTRE is relabelled to the optimised position snapped to the cell lattice of the old centre (the
 * circular buffer is not shifted and nothing is cleared) and every valid elevation is shifted by dz. */
void gemo_map_optmove(gemo_map* m, const float opt_p[2], float height_update, float out_aligned[2])
{
    for (int i = 0; i < 2; ++i) {
        const float position_shift = opt_p[i] - m->center[i];
        const int index_shift = (int)((double)(position_shift / m->res) + 0.5 * (position_shift > 0 ? 1 : -1));   /* GPU:1210 */
        out_aligned[i] = m->center[i] + m->res * (float)index_shift;                                              /* GPU:1211 */
    }
    m->center[0] = out_aligned[0]; m->center[1] = out_aligned[1];
    update_map_height(m, height_update);
}

/* GPU:1235-1254 (Map_closeloop; declared EMg.cpp:46, never called): the centre moves by the aligned shift
 * through PositionToRange like Move's, again without touching the buffer, plus the height shift. */
void gemo_map_closeloop(gemo_map* m, const float update_position[2], float height_update)
{
    for (int i = 0; i < 2; ++i) {
        const float position_shift = update_position[i] - m->center[i];
        const int index_shift = (int)((double)(position_shift / m->res) + 0.5 * (position_shift > 0 ? 1 : -1));   /* GPU:897 */
        const float aligned = (float)index_shift * m->res;                                                        /* GPU:909 */
        m->center[i] = position_to_range(m->center[i], aligned, m->res);                                          /* GPU:996-1002 */
    }
    update_map_height(m, height_update);
}
