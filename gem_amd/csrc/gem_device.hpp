// gem_device.hpp -- per-point device arithmetic of the GEM hot path for gfx950.
//
// Semantics follow the reference's G_pointsprocess / PointsToMapIndex
// (elevation_mapping/elevation_mapping/cuda/gpu_process.cu:384-455, 332-358; "GPU:" below).
// The translation unit is compiled with -ffp-contract=off: cell indices must be bit-exact, so no
// product+sum may be contracted into an FMA, and the IEEE divide/sqrt of hipcc's default mode
// are kept (no -ffast-math).  3-term sums written c0 + (c1 + c2) mirror Eigen's fixed-size
// redux order for the products the reference writes with Eigen types (GPU:403-425).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gem {

constexpr float kEmptyElevation = -10.0f;   // GPU:204 "elevation == -10" is the empty-cell sentinel
constexpr float kInitVariance   = -10.0f;   // GPU:205
constexpr int   kInvalidTile    = 0x7fffffff;

// Everything a kernel needs for one frame; passed BY VALUE as a kernel argument so it lives in
// SGPRs / the scalar cache (the reference uploads the same data with 5 cudaMemcpyToSymbol calls
// and 5 Eigen by-value kernel arguments per frame, GPU:1110-1119).
struct FrameConst {
    float  T[12];            // rows 0..2 of the sensor->map transform
    double lower, upper;     // GPU:50-51 are doubles
    float  lower_f, upper_f; // the same window as float bounds: h > lower_f <=> (double)h > lower, h < upper_f <=> (double)h < upper (fill_frame)
    double sp[8];            // sensor-model parameters
    float  Js[3];            // sensorJacobian
    float  Q[9];             // rotationVariance
    float  C[9];             // C_SB_transpose
    float  P[3];             // P_mul_C_BM_transpose
    float  Bs[9];            // B_r_BS_skew
    float  fbx, fby, fband, fplane;
    int    filter_on;
    int    model;
    int    orig_width;
    // map pose / geometry (GPU:30-31,35-36)
    float  cx, cy;
    int    sx, sy;
    int    L;
    float  res;
    // strip of storage rows owned by this device (multi-GPU tiling); whole map = [0, L)
    int    row0, row1;
    // kModelLaserFast (see height_variance): the laser model's constants as floats and the point-independent term of the variance
    float  beam_a, beam_c, t2;
    int    fast_laser;       // 1: the frame qualifies for kModelLaserFast (fill_frame decides: among other things 2^-20 <= res <= 2^20)
};

constexpr int kModelLaserFast = 4;   // template argument only: the laser model of a frame whose rotation variance is zero

// Lidar -> image projection of the input colourisation (EMg.cpp:321-345: P_lidar2img = T.camera (3x4) * T.lidar (4x4), in
// double), by value like FrameConst.
struct CameraConst {
    double P[12];            // row-major 3 x 4
    int    width, height;    // of the BGR image
};

// EMg.cpp:350-367: P_img = P_lidar2img * (x, y, z, 1) in double; P_x, P_y are FLOATS; cv::Point's members are ints, so the
// assignment truncates; the pixel is sampled when 0 < x < width, 0 < y < height and P_img.z > 0.  Returns y * width + x, or -1.
// The sums run k = 0..3 (Eigen's gemv order depends on its version and on alignment; a different order moves P_img by an ulp of
// a double, which reaches the truncated pixel only on a double-rounding tie).  A float outside the int range (or NaN: z = 0)
// converts to INT_MIN on x86 (cvttss2si) and fails the "> 0" test: outside.
__host__ __device__ inline int camera_pixel(const CameraConst& c, float x, float y, float z)
{
    const double v0 = (double)x, v1 = (double)y, v2 = (double)z;
    double r[3];
    for (int k = 0; k < 3; ++k) r[k] = ((c.P[4 * k] * v0 + c.P[4 * k + 1] * v1) + c.P[4 * k + 2] * v2) + c.P[4 * k + 3];
    const float px = (float)(r[0] / r[2]), py = (float)(r[1] / r[2]);
    if (!(px > -2147483648.0f && px < 2147483648.0f && py > -2147483648.0f && py < 2147483648.0f)) return -1;
    const int ix = (int)px, iy = (int)py;
    if (ix > 0 && ix < c.width && iy > 0 && iy < c.height && r[2] > 0.0) return iy * c.width + ix;
    return -1;
}

struct Projected {
    float h, var, xt, yt;
    int   row, col;          // storage row / column, -1 if outside the map (or the point was rejected)
    int   cell;              // row * L + col, -1 if outside
    bool  accepted;          // passed the reject filter and the height window (GPU:397)
};

__device__ __forceinline__ float dot3(float a0, float b0, float a1, float b1, float a2, float b2)
{
    const float c0 = a0 * b0, c1 = a1 * b1, c2 = a2 * b2;
    return c0 + (c1 + c2);
}

// one axis of GPU:340-348
__device__ __forceinline__ int axis_index(int L, float res, float shift)
{
    if ((L & 1) == 0) {
        const float v = (float)(L / 2) - shift / res;
        if (!(v > -2147483648.0f && v < 2147483648.0f)) return -1;   // non-finite / unrepresentable -> outside
        return (int)v;                                               // truncation toward zero
    } else {
        const double v = (double)(shift / res) + 0.5 * (shift > 0 ? 1 : -1);
        if (!(v > -2147483648.0 && v < 2147483648.0)) return -1;
        return L / 2 - (int)v;
    }
}

// GPU:332-358 (PointsToMapIndex): storage (circular-buffer) row / column of a map-frame point.
// 0 <= ix < L and 0 <= start < L, so (ix + start) % L is one conditional subtraction.
__device__ __forceinline__ bool map_row_col(const FrameConst& f, float px, float py, int& row, int& col)
{
    const float shx = px - f.cx, shy = py - f.cy;
    const int ix = axis_index(f.L, f.res, shx), iy = axis_index(f.L, f.res, shy);
    if (ix >= 0 && ix < f.L && iy >= 0 && iy < f.L) {
        const int stx = ix + f.sx, sty = iy + f.sy;
        row = stx >= f.L ? stx - f.L : stx;
        col = sty >= f.L ? sty - f.L : sty;
        return true;
    }
    row = -1; col = -1;
    return false;
}

__device__ __forceinline__ int map_index(const FrameConst& f, float px, float py)
{
    int row, col;
    return map_row_col(f, px, py, row, col) ? row * f.L + col : -1;
}

// MODEL >= 0: the sensor model is known at compile time (the laser-only instantiations of the big-pass kernels do not carry the
// double-precision pow / sqrt code of the camera models); MODEL < 0: taken from the frame
template <int MODEL = -1>
__device__ __forceinline__ void sensor_variances(const FrameConst& f, float x, float y, float z, int orig,
                                                 float& vn, float& vl)
{
    switch (MODEL >= 0 ? MODEL : f.model) {
    default:
    case 0: {   // laser, GPU:404-408
        const float d = sqrtf(dot3(x, x, y, y, z, z));
        const float min_r = (float)f.sp[0], beam_a = (float)f.sp[1], beam_c = (float)f.sp[2];
        vn = min_r * min_r;
        const float t = beam_c + beam_a * d;
        vl = t * t;
        break; }
    case 1: {   // structured light, StructuredLightSensorProcessor.cpp:128-139 (double expression)
        const double zd = (double)z;
        const float dev_n = (float)(f.sp[0] + f.sp[1] * (zd - f.sp[2]) * (zd - f.sp[2]) + f.sp[3] * pow(zd, f.sp[4]));
        vn = dev_n * dev_n;
        const float dev_l = (float)(f.sp[5] * zd);
        vl = dev_l * dev_l;
        break; }
    case 2: {   // stereo, StereoSensorProcessor.cpp:78-92
        const int w = f.orig_width > 0 ? f.orig_width : 1;
        const int I = orig / w, J = orig % w;
        const double disparity = f.sp[6] / (double)z;
        const float d = sqrtf(dot3(x, x, y, y, z, z));
        const double t = f.sp[2] * disparity + f.sp[3] - (double)J;
        const double u = 240.0 - (double)I;
        const double g = f.sp[6] / (disparity * disparity);
        vn = (float)(g * g * ((f.sp[4] * disparity + f.sp[1]) * sqrt(t * t + u * u) + f.sp[0]));
        const double l = f.sp[5] * (double)d;
        vl = (float)(l * l);
        break; }
    case 3:     // perfect, PerfectSensorProcessor.cpp:86-88
        vn = 0.0f; vl = 0.0f;
        break;
    }
}

// GPU:403-425
// MODEL == kModelLaserFast: the reference hard-zeroes rotationVariance (SPB.cpp:202-204), and with Sigma_q = 0 the first term
// Jq Sigma_q Jq^T -- q = C_SB^T p, S = skew(q) + B_r_BS_skew, Jq = P S, three dot products with Sigma_q's columns, one with Jq:
// 45 of the projection's instructions -- is a sum of products with zero: +0 or -0 whenever Jq is finite.  The second term is
// Js diag(vl, vl, vn) Js^T = ((Js0 vl) Js0 + (Js1 vl) Js1) + (Js2 vn) Js2 up to the signs of zeros, and its last addend t2 is a
// constant of the frame (vn = min_r^2).  If t2 > 0 the sum is positive, a zero of either sign added to it changes nothing, and
// the variance is exactly this expression.  fill_frame() checks what that needs: Sigma_q = 0, t2 > 0, finite constants, and a
// frame that bounds the points it accepts (orthonormal rotation, finite height window: a point inside the map and the window
// then has |p| < 1e10, so Jq is finite) -- every point that leaves a record gets the reference's variance bit for bit; a
// frame that does not qualify takes the generic instantiation.
// sqrtf(x) for x >= 0 as hipcc expands it (v_sqrt_f32, then the two neighbours' residuals pick the correctly rounded value) WITHOUT
// the expansion's rescaling of tiny arguments: the same operations on the same values -- the same bits -- for x = 0, x = inf, NaN and
// every x >= 2^-96 (below that the expansion multiplies by 2^32 first: the residual fmas would lose bits); seven instructions
// instead of fourteen.  A wave that holds a positive x below 2^-96 (a point within 3.5e-15 m of the sensor) takes sqrtf as written.
__device__ __forceinline__ float sqrt_plain(float x)
{
    const bool tiny = (__float_as_uint(x) - 1u) < 0x0f7fffffu;          // 0 < x < 2^-96 (the kernels pass sums of squares: never negative)
    if (__builtin_expect(__ballot(tiny) != 0, 0)) return sqrtf(x);      // wave-uniform
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    float r = rm <= 0.0f ? sm : s;
    r = rp > 0.0f ? sp : r;
    return r;
}

// n / d for a divisor d in [2^-20, 2^20] whose refined reciprocal r = rcp_refined(d) is at hand: hipcc expands an IEEE float division
// into v_div_scale x 2, v_rcp, a Newton step on the reciprocal, the quotient, two residual corrections (the last a v_div_fmas) and
// v_div_fixup; for 2^-60 <= |n| <= 2^60 the scale factors are 1, v_div_fmas is a plain fma and the fix-up passes the quotient
// through -- the same operations on the same values, the same bits, in five instructions instead of eleven (fuse_step<true> has
// the same argument for the Kalman quotients).  OUTSIDE that range of n the result may differ from n / d, but only where the
// binning cannot tell: |n| < 2^-60 gives |q| < 2^-40 either way, which (float)(L / 2) - q rounds away for L >= 2; |n| > 2^60
// gives |q| > 2^40, an infinity or a NaN: outside the int range either way (the callers test that before the cast).
__device__ __forceinline__ float rcp_refined(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
}
__device__ __forceinline__ float div_binning(float n, float d, float r)
{
    float q = n * r;
    float t = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(t, r, q);
    t = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(t, r, q);
}

template <int MODEL = -1>
__device__ __forceinline__ float height_variance(const FrameConst& f, float x, float y, float z, int orig)
{
    if constexpr (MODEL == kModelLaserFast) {
        const float d = sqrt_plain(dot3(x, x, y, y, z, z));             // GPU:404
        const float t = f.beam_c + f.beam_a * d;
        const float vl = t * t;                                         // GPU:407
        return ((f.Js[0] * vl) * f.Js[0] + (f.Js[1] * vl) * f.Js[1]) + f.t2;
    }
    float vn, vl;
    sensor_variances<MODEL>(f, x, y, z, orig, vn, vl);
    const float q0 = dot3(f.C[0], x, f.C[1], y, f.C[2], z);
    const float q1 = dot3(f.C[3], x, f.C[4], y, f.C[5], z);
    const float q2 = dot3(f.C[6], x, f.C[7], y, f.C[8], z);
    const float S0 = 0.0f + f.Bs[0], S1 = -q2 + f.Bs[1],  S2 = q1 + f.Bs[2];
    const float S3 = q2 + f.Bs[3],   S4 = 0.0f + f.Bs[4], S5 = -q0 + f.Bs[5];
    const float S6 = -q1 + f.Bs[6],  S7 = q0 + f.Bs[7],   S8 = 0.0f + f.Bs[8];
    const float Jq0 = dot3(f.P[0], S0, f.P[1], S3, f.P[2], S6);
    const float Jq1 = dot3(f.P[0], S1, f.P[1], S4, f.P[2], S7);
    const float Jq2 = dot3(f.P[0], S2, f.P[1], S5, f.P[2], S8);
    const float a0 = dot3(Jq0, f.Q[0], Jq1, f.Q[3], Jq2, f.Q[6]);
    const float a1 = dot3(Jq0, f.Q[1], Jq1, f.Q[4], Jq2, f.Q[7]);
    const float a2 = dot3(Jq0, f.Q[2], Jq1, f.Q[5], Jq2, f.Q[8]);
    float hv = a0 * Jq0 + a1 * Jq1 + a2 * Jq2;                      // cuda_computer, GPU:293-298
    const float b0 = dot3(f.Js[0], vl, f.Js[1], 0.0f, f.Js[2], 0.0f);
    const float b1 = dot3(f.Js[0], 0.0f, f.Js[1], vl, f.Js[2], 0.0f);
    const float b2 = dot3(f.Js[0], 0.0f, f.Js[1], 0.0f, f.Js[2], vn);
    hv += b0 * f.Js[0] + b1 * f.Js[1] + b2 * f.Js[2];
    return hv;
}

// one point of G_pointsprocess (GPU:384-455), without the racy map_lowest side effect
template <int MODEL = -1>
__device__ __forceinline__ Projected project_point(const FrameConst& f, float x, float y, float z, int orig)
{
    Projected r;
    const float h = f.T[8] * x + f.T[9] * y + f.T[10] * z + f.T[11];               // GPU:389
    bool flag = false;
    if (f.filter_on)                                                               // GPU:393
        flag = (x > -f.fbx && x < f.fbx && y > -f.fby && y < f.fby) || (y > -f.fband && y < f.fband) || (y > f.fplane);
    if ((h > f.lower_f && h < f.upper_f) && !flag) {                               // GPU:397, doubles there: see FrameConst
        r.xt = f.T[0] * x + f.T[1] * y + f.T[2] * z + f.T[3];                      // GPU:399
        r.yt = f.T[4] * x + f.T[5] * y + f.T[6] * z + f.T[7];                      // GPU:400
        r.h = h;
        r.var = height_variance<MODEL>(f, x, y, z, orig);
        r.cell = map_row_col(f, r.xt, r.yt, r.row, r.col) ? r.row * f.L + r.col : -1;   // GPU:431
        r.accepted = true;
    } else {                                                                       // GPU:441-451
        r.xt = -1.0f; r.yt = -1.0f; r.h = -1.0f; r.var = -1.0f; r.cell = -1; r.row = -1; r.col = -1; r.accepted = false;
    }
    return r;
}

// project_point<kModelLaserFast> + the binning as STRAIGHT-LINE code: the same expressions, every decision a select (the branchy form
// spends a third of its issue slots on exec-mask bookkeeping; computing the variance of a rejected point costs less).  Returns
// "accepted, inside the map, inside this device's strip, not the h == -1 sentinel (GPU:482) unless those are kept".
__device__ __forceinline__ bool project_bin_laser_fast(const FrameConst& fc, float x, float y, float z, bool in_range, bool keep_sentinel,
                                                       int& row, int& col, float& h_out, float& var_out)
{
    const float h = fc.T[8] * x + fc.T[9] * y + fc.T[10] * z + fc.T[11];           // GPU:389
    bool acc = in_range && h > fc.lower_f && h < fc.upper_f;                        // GPU:397 (doubles there: see FrameConst)
    if (fc.filter_on)                                                              // GPU:393 (frame-uniform)
        acc = acc && !((x > -fc.fbx && x < fc.fbx && y > -fc.fby && y < fc.fby) || (y > -fc.fband && y < fc.fband) || (y > fc.fplane));
    const float xt = fc.T[0] * x + fc.T[1] * y + fc.T[2] * z + fc.T[3];            // GPU:399
    const float yt = fc.T[4] * x + fc.T[5] * y + fc.T[6] * z + fc.T[7];            // GPU:400
    var_out = height_variance<kModelLaserFast>(fc, x, y, z, 0);
    h_out = h;
    const float shx = xt - fc.cx, shy = yt - fc.cy;
    int ix, iy;
    if ((fc.L & 1) == 0) {                                                         // GPU:340-348, even L (map-uniform)
        const float rr = rcp_refined(fc.res);                                      // (frame-uniform: hoisted out of the callers' loops)
        const float vx = (float)(fc.L / 2) - div_binning(shx, fc.res, rr), vy = (float)(fc.L / 2) - div_binning(shy, fc.res, rr);
        const bool okx = vx > -2147483648.0f && vx < 2147483648.0f, oky = vy > -2147483648.0f && vy < 2147483648.0f;
        ix = okx ? (int)(okx ? vx : 0.0f) : -1;                                    // truncation toward zero; non-finite / unrepresentable -> outside
        iy = oky ? (int)(oky ? vy : 0.0f) : -1;
    } else {
        ix = axis_index(fc.L, fc.res, shx); iy = axis_index(fc.L, fc.res, shy);
    }
    const bool inside = ix >= 0 && ix < fc.L && iy >= 0 && iy < fc.L;
    const int stx = ix + fc.sx, sty = iy + fc.sy;
    row = stx >= fc.L ? stx - fc.L : stx; col = sty >= fc.L ? sty - fc.L : sty;    // (% L: one conditional subtraction)
    return acc && inside && row >= fc.row0 && row < fc.row1 && (h != -1.0f || keep_sentinel);
}

// The per-cell recurrence of G_fuse (GPU:484-529) on register state (e, s).  Returns true when the
// point was "taken" (colour / intensity of this point may overwrite the cell's, GPU:487-494 etc.).
// Straight-line form: the Mahalanobis quotient and the two Kalman quotients are independent
// correctly-rounded divisions, so all three are issued together and the reference's three-way
// branch becomes a select.  Every value that is kept is computed by exactly the reference's
// expression (GPU:502, 518, 519); discarded lanes may hold inf / NaN, which is harmless.
// SHARED_RCP: the two Kalman quotients from one refined reciprocal (below).  Measured per kernel: k_fuse_block's one-step loop
// gains (C4: 99.5 -> 95.8 us per batch), k_fuse_walk's four-step groups lose (C3: 85 -> 91 us: the guard and its branch cost more
// than the divisions' slow instructions there), k_frame's straight-line chains are even -- so only k_fuse_block asks for it.
template <bool SHARED_RCP = false>
__device__ __forceinline__ bool fuse_step(float& e, float& s, float h, float v, float mahal_thr, float var_floor)
{
    const bool empty = e == kEmptyElevation;                                       // GPU:484
    const float sf = s < var_floor ? var_floor : s;                                // GPU:500-501 (written back)
    // GPU:502 computes m = |h - e| / sqrt(sf) and only ever compares it with the threshold (GPU:504).  The
    // correctly-rounded sqrt -> division chain is the longest dependency of the per-cell recurrence, so the
    // decision is first taken from |h - e| * v_rsq_f32(sf) (error < 4 ulp against the reference's m) and
    // the reference's own expression is evaluated only inside a +-1e-5 relative band around the threshold
    // (or for a subnormal sf, which v_rsq_f32 flushes): the decision is the reference's in every case.
    const float d  = fabsf(h - e);
    float m = d * __builtin_amdgcn_rsqf(sf);
    if (__builtin_expect(fabsf(m - mahal_thr) <= 1e-5f * fabsf(mahal_thr) || !(sf >= 1e-30f), 0))
        m = d / sqrtf(sf);                                                         // a float, not a flag, leaves the rare branch
    const bool outlier = m > mahal_thr;
    // GPU:518, 519: two IEEE divisions by the same denominator D = sf + v.  hipcc expands each into v_div_scale x 2, v_rcp, one
    // Newton step on the reciprocal, the quotient, two residual corrections (the last a v_div_fmas) and v_div_fixup -- eleven
    // instructions, five of them quarter-rate, the longest part of the recurrence's step.  When D and both numerators lie well
    // inside the exponent range (here: within 2^-60 .. 2^60; v_div_scale only rescales operands near its ends, v_div_fixup only
    // touches zeros, infinities, NaNs and denormals) the scale factors are 1, v_div_fmas is a plain fma and the fix-up passes
    // the quotient through: the SAME operations on the same values -- the same bits -- are one shared refined reciprocal and,
    // per quotient, a multiplication and four fmas.  Anything else (a zero or denormal numerator, NaNs, huge values: decided for
    // the whole wave) takes the division as written.  Every exactness test of the suite runs through this.
    const float D = sf + v, N1 = sf * h + v * e, N2 = v * sf;
    float en, sn;
    if constexpr (!SHARED_RCP) {
        en = N1 / D;
        sn = N2 / (v + sf);
    } else {
        const uint32_t xd = (__float_as_uint(D) >> 23) & 0xffu, x1 = (__float_as_uint(N1) >> 23) & 0xffu, x2 = (__float_as_uint(N2) >> 23) & 0xffu;
        const bool plain = min(xd, min(x1, x2)) >= 67u && max(xd, max(x1, x2)) <= 187u && D > 0.0f;
        if (__builtin_expect(__ballot(!plain) == 0, 1)) {               // wave-uniform
            const float r0 = __builtin_amdgcn_rcpf(D);
            const float r = __builtin_fmaf(__builtin_fmaf(-D, r0, 1.0f), r0, r0);
            float q = N1 * r;  float t = __builtin_fmaf(-D, q, N1);  q = __builtin_fmaf(t, r, q);  t = __builtin_fmaf(-D, q, N1);  en = __builtin_fmaf(t, r, q);
            q = N2 * r;        t = __builtin_fmaf(-D, q, N2);        q = __builtin_fmaf(t, r, q);  t = __builtin_fmaf(-D, q, N2);  sn = __builtin_fmaf(t, r, q);
        } else {
            en = N1 / D;
            sn = N2 / (v + sf);
        }
    }
    // (bitwise, not short-circuit: `||` / `&&` become selects between i1 values, which this compiler carries through VGPRs --
    //  five instructions per record; `|` / `&` stay operations on the wave's lane masks)
    const bool replace = empty | (outlier & (e < h));                              // GPU:484-486, 505-507
    // not replaced: an outlier below the cell leaves it alone (the floor is still written back), anything else is fused --
    // two selects per value
    const float e_new = replace ? h : (outlier ? e : en);
    const float s_new = replace ? v : (outlier ? sf : sn);
    const bool fuse = !outlier;
    e = e_new; s = s_new;
    return replace | fuse;
}

} // namespace gem
