/*
 * gem_oracle_feature.c -- CPU ORACLE (test infrastructure, see gem_oracle.h) for the traversability
 * stage that runs right after the fusion: G_Mapfeature (GPU:549-670) with its Jacobi eigen-solver
 * computerEigenvalue (GPU:66-187), reached through Map_feature (GPU:1256-1302, EMg.cpp:410).
 *
 * Literal restatement, arrays and loops as in the reference.  Arithmetic types follow C++ overload
 * resolution in CUDA device code: fabs / acos / atan2 / sin / cos on float arguments are the float
 * overloads; double literals (0.5, 0.6, 0.2, 1.0) promote the expressions they appear in.
 * The reference's trigonometry is CUDA's libm (2-3 ulp, not reproducible outside CUDA); here every
 * float trig call is evaluated in double and rounded to float, which the HIP kernel does too, so that
 * oracle and kernel agree except when a double result falls within ~1e-16 of a float rounding boundary.
 * Build with -ffp-contract=off.
 */
#include "gem_oracle.h"

#include <math.h>

static float f_sin(float x)            { return (float)sin((double)x); }
static float f_cos(float x)            { return (float)cos((double)x); }
static float f_atan2(float y, float x) { return (float)atan2((double)y, (double)x); }
static float f_acos(float x)           { return (float)acos((double)x); }

/* GPU:66-187 */
static void computer_eigenvalue(float* pMatrix, int nDim, float* maxvector, float dbEps, int nJt)
{
    float pdblVects[9];
    float pdbEigenValues[3];
    for (int i = 0; i < nDim; i++) {                                   /* GPU:71-79 */
        pdblVects[i * nDim + i] = 1.0f;
        for (int j = 0; j < nDim; j++) if (i != j) pdblVects[i * nDim + j] = 0.0f;
    }
    int nCount = 0;
    while (1) {
        float dbMax = pMatrix[1];                                      /* GPU:85: signed, not fabs */
        int nRow = 0, nCol = 1;
        for (int i = 0; i < nDim; i++)
            for (int j = 0; j < nDim; j++) {
                float d = fabsf(pMatrix[i * nDim + j]);
                if ((i != j) && (d > dbMax)) { dbMax = d; nRow = i; nCol = j; }
            }
        if (dbMax < dbEps) break;                                      /* GPU:103 */
        if (nCount > nJt) break;                                       /* GPU:106 */
        nCount++;
        float dbApp = pMatrix[nRow * nDim + nRow];
        float dbApq = pMatrix[nRow * nDim + nCol];
        float dbAqq = pMatrix[nCol * nDim + nCol];
        float dbAngle = (float)(0.5 * (double)f_atan2(-2 * dbApq, dbAqq - dbApp));     /* GPU:116 */
        float dbSinTheta = f_sin(dbAngle);
        float dbCosTheta = f_cos(dbAngle);
        float dbSin2Theta = f_sin(2 * dbAngle);
        float dbCos2Theta = f_cos(2 * dbAngle);
        pMatrix[nRow * nDim + nRow] = dbApp * dbCosTheta * dbCosTheta +
            dbAqq * dbSinTheta * dbSinTheta + 2 * dbApq * dbCosTheta * dbSinTheta;     /* GPU:122-123 */
        pMatrix[nCol * nDim + nCol] = dbApp * dbSinTheta * dbSinTheta +
            dbAqq * dbCosTheta * dbCosTheta - 2 * dbApq * dbCosTheta * dbSinTheta;     /* GPU:124-125 */
        pMatrix[nRow * nDim + nCol] = (float)(0.5 * (double)(dbAqq - dbApp) * (double)dbSin2Theta + (double)(dbApq * dbCos2Theta));   /* GPU:126 */
        pMatrix[nCol * nDim + nRow] = pMatrix[nRow * nDim + nCol];
        for (int i = 0; i < nDim; i++)                                 /* GPU:129-139 */
            if ((i != nCol) && (i != nRow)) {
                int u = i * nDim + nRow, w = i * nDim + nCol;
                dbMax = pMatrix[u];
                pMatrix[u] = pMatrix[w] * dbSinTheta + dbMax * dbCosTheta;
                pMatrix[w] = pMatrix[w] * dbCosTheta - dbMax * dbSinTheta;
            }
        for (int j = 0; j < nDim; j++)                                 /* GPU:141-151 */
            if ((j != nCol) && (j != nRow)) {
                int u = nRow * nDim + j, w = nCol * nDim + j;
                dbMax = pMatrix[u];
                pMatrix[u] = pMatrix[w] * dbSinTheta + dbMax * dbCosTheta;
                pMatrix[w] = pMatrix[w] * dbCosTheta - dbMax * dbSinTheta;
            }
        for (int i = 0; i < nDim; i++) {                               /* GPU:154-161 */
            int u = i * nDim + nRow, w = i * nDim + nCol;
            dbMax = pdblVects[u];
            pdblVects[u] = pdblVects[w] * dbSinTheta + dbMax * dbCosTheta;
            pdblVects[w] = pdblVects[w] * dbCosTheta - dbMax * dbSinTheta;
        }
    }
    int min_id = 0;
    float minEigenvalue = 0.0f;
    for (int i = 0; i < nDim; i++) {                                   /* GPU:168-181 */
        pdbEigenValues[i] = pMatrix[i * nDim + i];
        if (i == 0) minEigenvalue = pdbEigenValues[i];
        else if (minEigenvalue > pdbEigenValues[i]) { minEigenvalue = pdbEigenValues[i]; min_id = i; }
    }
    for (int i = 0; i < nDim; i++) maxvector[i] = pdblVects[min_id + nDim * i];   /* GPU:183-186 */
}

/* GPU:549-670.  rough / slope / traver_out: L*L arrays or NULL.  m->traver is updated like map_traver.
 * Cells with elevation == -10 are not written by the reference (its output arrays stay uninitialised and
 * map_traver keeps its value); here they report rough = slope = 0 and the stored traversability. */
void gemo_map_feature(gemo_map* m, float* rough, float* slope, float* traver_out)
{
    const int Length = m->L;
    const float Resolution = m->res;
    for (int idx = 0; idx < Length * Length; ++idx) {
        float r_out = 0.0f, s_out = 0.0f;
        if (m->elevation[idx] != -10.0f) {
            float px[25], py[25], pz[25];
            float px_mean = 0, py_mean = 0, pz_mean = 0;
            const int cell_x = idx / Length, cell_y = idx % Length;
            int p_n = 0;
            for (int i = -2; i < 3; i++)
                for (int j = -2; j < 3; j++) {
                    int Ele_x = (cell_x + Length - m->start[0]) % Length;      /* unrolled (geographic) index, GPU:587-588 */
                    int Ele_y = (cell_y + Length - m->start[1]) % Length;
                    Ele_x = Ele_x + i; Ele_y = Ele_y + j;
                    if (Ele_x >= 0 && Ele_x < Length && Ele_y >= 0 && Ele_y < Length) {
                        const int point_x = (cell_x + i + Length) % Length;      /* storage neighbour, wraps (GPU:596-600) */
                        const int point_y = (cell_y + j + Length) % Length;
                        const float s_z = m->elevation[point_x * Length + point_y];
                        if (s_z != -10.0f) {
                            px[p_n] = point_x * Resolution;                      /* STORAGE coordinates, GPU:604-605 */
                            py[p_n] = point_y * Resolution;
                            pz[p_n] = s_z;
                            px_mean = px_mean + px[p_n]; py_mean = py_mean + py[p_n]; pz_mean = pz_mean + pz[p_n];
                            p_n++;
                        }
                    }
                }
            if (p_n > 7) {
                px_mean = px_mean / p_n; py_mean = py_mean / p_n; pz_mean = pz_mean / p_n;
                float pMatrix[9] = {0};
                for (int i = 0; i < p_n; i++) {                                  /* GPU:624-635 */
                    pMatrix[0] = pMatrix[0] + (px[i] - px_mean) * (px[i] - px_mean);
                    pMatrix[4] = pMatrix[4] + (py[i] - py_mean) * (py[i] - py_mean);
                    pMatrix[8] = pMatrix[8] + (pz[i] - pz_mean) * (pz[i] - pz_mean);
                    pMatrix[1] = pMatrix[1] + (px[i] - px_mean) * (py[i] - py_mean);
                    pMatrix[2] = pMatrix[2] + (px[i] - px_mean) * (pz[i] - pz_mean);
                    pMatrix[5] = pMatrix[5] + (py[i] - py_mean) * (pz[i] - pz_mean);
                    pMatrix[3] = pMatrix[1]; pMatrix[6] = pMatrix[2]; pMatrix[7] = pMatrix[5];
                }
                float normal_vec[3];
                computer_eigenvalue(pMatrix, 3, normal_vec, (float)0.01, 30);     /* GPU:637-642 */
                const float height = m->elevation[idx], smooth_height = pz_mean;
                float Slope = normal_vec[2] > 0 ? f_acos(normal_vec[2]) : f_acos(-normal_vec[2]);   /* GPU:647-650 */
                float Rough = fabsf(height - smooth_height);
                float Traver = (float)(0.5 * (1.0 - (double)Slope / 0.6) + 0.5 * (1.0 - ((double)Rough / 0.2)));   /* GPU:653 */
                s_out = Slope; r_out = Rough;
                m->traver[idx] = Traver;
            } else {                                                             /* GPU:660-666 */
                m->traver[idx] = -10.0f;
            }
        }
        if (rough) rough[idx] = r_out;
        if (slope) slope[idx] = s_out;
        if (traver_out) traver_out[idx] = m->traver[idx];
    }
}
