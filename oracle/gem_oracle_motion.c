/*
 * gem_oracle_motion.c -- CPU ORACLE (test infrastructure only; see gem_oracle.h).
 *
 * Restates RobotMotionMapUpdater::update / computeReducedCovariance / computeRelativeCovariance
 * (RMU.cpp:42-90, 92-109, 111-145): 6x6 pose covariance -> scalar variance increment that the
 * reference hands to Mapvar_update (RMU.cpp:80-81).
 *
 * Third-party code on this path: kindr (un-vendored, version unpinned) for rotations.  Restated
 * with plain 3x3 matrices under kindr-1.x conventions: a rotation C_IB maps base coordinates to
 * inertial (map) coordinates, I_r = C_IB * B_r; rotate(v) = C v; inverseRotate(v) = C^T v;
 * A * B composes as the matrix product; EulerAnglesZyx: C = Rz(yaw) Ry(pitch) Rx(roll);
 * RotationVector = axis * angle (log map).  PARITY UNPINNED (no reference fixtures).
 */
#include "gem_oracle.h"

#include <math.h>
#include <string.h>

static void mat_mul(const double* A, const double* B, double* C, int n, int k, int m)
{   /* C[n x m] = A[n x k] * B[k x m], row-major */
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            double s = 0.0;
            for (int l = 0; l < k; ++l) s += A[i * k + l] * B[l * m + j];
            C[i * m + j] = s;
        }
}
static void mat_T(const double* A, double* At, int n, int m)
{   /* At[m x n] = A[n x m]^T */
    for (int i = 0; i < n; ++i) for (int j = 0; j < m; ++j) At[j * n + i] = A[i * m + j];
}

void gemo_motion_init(gemo_motion_state* s, double covariance_scale)
{
    memset(s, 0, sizeof(*s));                      /* RMU.cpp:27: previousReducedCovariance_.setZero() */
    s->prev_R[0] = s->prev_R[4] = s->prev_R[8] = 1.0;  /* default-constructed kindr pose = identity     */
    s->covariance_scale = covariance_scale;        /* RMU.cpp:25,38                                     */
}

double gemo_motion_update(gemo_motion_state* s, const double pos[3], const double R_IB[9],
                          const double cov6x6[36], const double map_R[9])
{
    /* RMU.cpp:46: scaled covariance */
    double cov[36];
    for (int i = 0; i < 36; ++i) cov[i] = s->covariance_scale * cov6x6[i];

    /* ---- computeReducedCovariance, RMU.cpp:92-109 ---- */
    double yaw = atan2(R_IB[3], R_IB[0]);
    double pitch = atan2(-R_IB[6], sqrt(R_IB[0] * R_IB[0] + R_IB[3] * R_IB[3]));
    double tan_pitch = tan(pitch);
    double J[24]; memset(J, 0, sizeof(J));          /* 4 x 6 */
    J[0 * 6 + 0] = J[1 * 6 + 1] = J[2 * 6 + 2] = 1.0;
    J[3 * 6 + 3] = cos(yaw) * tan_pitch; J[3 * 6 + 4] = sin(yaw) * tan_pitch; J[3 * 6 + 5] = 1.0;
    double JC[24], Jt[24], reduced[16];
    mat_mul(J, cov, JC, 4, 6, 6);
    mat_T(J, Jt, 4, 6);
    mat_mul(JC, Jt, reduced, 4, 6, 4);

    /* ---- computeRelativeCovariance, RMU.cpp:111-145 ---- */
    /* rotation vector of C_IB, keep only z (RMU.cpp:116-118) */
    double tr = R_IB[0] + R_IB[4] + R_IB[8];
    double c = 0.5 * (tr - 1.0); if (c > 1.0) c = 1.0; if (c < -1.0) c = -1.0;
    double angle = acos(c);
    double wz = 0.5 * (R_IB[3] - R_IB[1]);           /* (R21 - R12)/2 = axis_z * sin(angle) */
    double rz = (angle < 1e-12) ? wz : wz * angle / sin(angle);
    double Rt[9] = { cos(rz), -sin(rz), 0.0, sin(rz), cos(rz), 0.0, 0.0, 0.0, 1.0 };  /* R_I_tilde_B */

    /* positionInRobotFrame = prevR^T (p - p_prev)  (RMU.cpp:121-123) */
    double dp[3] = { pos[0] - s->prev_pos[0], pos[1] - s->prev_pos[1], pos[2] - s->prev_pos[2] };
    double v[3];
    for (int i = 0; i < 3; ++i)
        v[i] = s->prev_R[0 * 3 + i] * dp[0] + s->prev_R[1 * 3 + i] * dp[1] + s->prev_R[2 * 3 + i] * dp[2];

    /* F (RMU.cpp:126-129): identity, top-right 3x1 = skew(e_z) * Rt * v */
    double Rv[3];
    for (int i = 0; i < 3; ++i) Rv[i] = Rt[i * 3 + 0] * v[0] + Rt[i * 3 + 1] * v[1] + Rt[i * 3 + 2] * v[2];
    double F[16]; memset(F, 0, sizeof(F));
    F[0] = F[5] = F[10] = F[15] = 1.0;
    F[0 * 4 + 3] = -Rv[1];                          /* skew((0,0,1)) = [[0,-1,0],[1,0,0],[0,0,0]] */
    F[1 * 4 + 3] = Rv[0];
    F[2 * 4 + 3] = 0.0;

    /* inv(G) dt and its transpose (RMU.cpp:132-137) */
    double G[16], Gt[16]; memset(G, 0, sizeof(G)); memset(Gt, 0, sizeof(Gt));
    G[15] = 1.0; Gt[15] = 1.0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { G[i * 4 + j] = Rt[j * 3 + i]; Gt[i * 4 + j] = Rt[i * 3 + j]; }

    /* relative = G (reduced - F prev F^T) Gt  (RMU.cpp:140-142) */
    double FP[16], Ft[16], FPF[16], D[16], GD[16], rel[16];
    mat_mul(F, s->prev_reduced_cov, FP, 4, 4, 4);
    mat_T(F, Ft, 4, 4);
    mat_mul(FP, Ft, FPF, 4, 4, 4);
    for (int i = 0; i < 16; ++i) D[i] = reduced[i] - FPF[i];
    mat_mul(G, D, GD, 4, 4, 4);
    mat_mul(GD, Gt, rel, 4, 4, 4);

    /* ---- update(), RMU.cpp:58-80 ---- */
    double Sigma[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Sigma[i * 3 + j] = rel[i * 4 + j];
    double RIBt[9], RBM[9], Jr[9], Jrt[9], JS[9], out[9];
    mat_T(R_IB, RIBt, 3, 3);
    mat_mul(RIBt, map_R, RBM, 3, 3, 3);            /* R_B_M = R_I_B^T R_I_M (RMU.cpp:62-63) */
    mat_T(RBM, Jr, 3, 3);
    for (int i = 0; i < 9; ++i) Jr[i] = -Jr[i];    /* J_r = -R_B_M^T (RMU.cpp:66) */
    mat_T(Jr, Jrt, 3, 3);
    mat_mul(Jr, Sigma, JS, 3, 3, 3);
    mat_mul(JS, Jrt, out, 3, 3, 3);
    float var_update = (float)out[8];              /* diagonal().cast<float>().z()  (RMU.cpp:69,80) */

    memcpy(s->prev_reduced_cov, reduced, sizeof(reduced));   /* RMU.cpp:85-86 */
    memcpy(s->prev_pos, pos, 3 * sizeof(double));
    memcpy(s->prev_R, R_IB, 9 * sizeof(double));
    return (double)var_update;
}
