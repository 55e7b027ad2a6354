/*
 * gem_oracle_mt.c -- the ALL-CORE form of the CPU oracle's fused path (test infrastructure only; see gem_oracle.h).
 *
 * "all host cores with row-strip tiling (each thread owns a strip of cells and scans the index array,
 * preserving order)".  This is how the reference's algorithm parallelises on a CPU without changing a single result:
 * G_fuse is one thread per CELL scanning the points in input order (GPU:477-537), so any partition of the CELLS is exact.
 *
 *   for every sweep s of the batch:
 *     phase A  (points split evenly over the threads)  process_one of gem_oracle.c: GPU:384-455 -> idx / h / var / xt / yt
 *     barrier
 *     phase B  (storage rows split evenly over the threads)  thread t: Mapvar_update on its rows (GPU:540-547), then ONE scan
 *              of idx[0..n) in input order fusing the points whose cell lies in its rows (GPU:480-531) and keeping the
 *              lowest scan point of those cells (GPU:430-439), then the variance floor on its rows (GPU:533-534)
 *     barrier
 *
 * The result is bit-identical to calling gemo_mapvar_update + gemo_add per sweep (tests/test_oracle_kat.py).
 * It exists to time the CPU path on all host cores (bench.py cpu_baseline) and to replay long bench sequences quickly.
 */
#define _GNU_SOURCE
#include "gem_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* the single-point pieces of gem_oracle.c (same translation-unit flags, same arithmetic) */
int  gemo_process_one(const gemo_map* m, const gemo_frame* f, float x, float y, float z, int orig,
                      int* map_index, float* var, float* xt, float* yt, float* zt);
void gemo_fuse_one(gemo_map* m, int c, float h, float v, int r, int g, int b, float inten);

typedef struct {
    gemo_map* m;
    int n_sweeps, nthreads;
    const gemo_frame* frames;
    const float* xyzi;
    const long long* offsets;
    const float* var_updates;          /* may be NULL */
    int* idx; float *h, *v, *xt, *yt;  /* per-point scratch of the current sweep */
    pthread_barrier_t bar;
    long long accepted;                /* summed by thread 0 from acc[] */
    long long* acc;
} mt_job;

typedef struct { mt_job* job; int t; } mt_arg;

static void* mt_worker(void* p)
{
    mt_arg* a = (mt_arg*)p;
    mt_job* j = a->job;
    gemo_map* m = j->m;
    const int t = a->t, NT = j->nthreads, L = m->L;
    const int r0 = (int)((long long)L * t / NT), r1 = (int)((long long)L * (t + 1) / NT);
    long long acc = 0;
    for (int s = 0; s < j->n_sweeps; ++s) {
        const long long base = j->offsets[s];
        const int n = (int)(j->offsets[s + 1] - base);
        const gemo_frame* f = &j->frames[s];
        /* phase A */
        const int i0 = (int)((long long)n * t / NT), i1 = (int)((long long)n * (t + 1) / NT);
        for (int i = i0; i < i1; ++i) {
            const float* q = j->xyzi + 4 * (base + i);
            acc += gemo_process_one(m, f, q[0], q[1], q[2], i, &j->idx[i], &j->v[i], &j->xt[i], &j->yt[i], &j->h[i]);
        }
        pthread_barrier_wait(&j->bar);
        /* phase B */
        if (j->var_updates) {
            const float u = j->var_updates[s];
            for (int c = r0 * L; c < r1 * L; ++c) if (m->variance[c] != -10.0f) m->variance[c] += u;      /* GPU:540-547 */
        }
        const int c_lo = r0 * L, c_hi = r1 * L;
        for (int i = 0; i < n; ++i) {
            const int c = j->idx[i];
            if (c < c_lo || c >= c_hi) continue;
            {   /* GPU:430-439, by geographic cell: the owner of the storage cell owns the geographic one too (1:1) */
                const int g = gemo_points_to_index(m, j->xt[i], j->yt[i]);
                if (g != -1) {
                    m->lowest[g] = fminf(j->h[i], m->lowest[g]);
                    if (j->h[i] == m->lowest[g]) m->lowest[g] = m->lowest[g] + 3 * j->v[i];
                }
            }
            if (j->h[i] == -1.0f) continue;                                                               /* GPU:482 */
            gemo_fuse_one(m, c, j->h[i], j->v[i], 0, 0, 0, j->xyzi[4 * (base + i) + 3]);
        }
        for (int c = c_lo; c < c_hi; ++c) if (m->variance[c] < m->var_floor) m->variance[c] = m->var_floor;   /* GPU:533-534 */
        pthread_barrier_wait(&j->bar);
    }
    j->acc[t] = acc;
    return NULL;
}

/* for s in 0..n_sweeps-1: Mapvar_update(var_updates[s]) (if given); add(frames[s], cloud s) -- on `nthreads` threads.
 * Colours are not handled (rgb == NULL form of gemo_add).  Returns the number of accepted points, -1 on failure. */
long long gemo_add_batch_mt(gemo_map* m, int n_sweeps, const gemo_frame* frames, const float* xyzi, const long long* offsets,
                            const float* var_updates, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > m->L) nthreads = m->L;
    long long nmax = 0;
    for (int s = 0; s < n_sweeps; ++s) if (offsets[s + 1] - offsets[s] > nmax) nmax = offsets[s + 1] - offsets[s];
    mt_job j;
    memset(&j, 0, sizeof(j));
    j.m = m; j.n_sweeps = n_sweeps; j.nthreads = nthreads; j.frames = frames; j.xyzi = xyzi; j.offsets = offsets; j.var_updates = var_updates;
    j.idx = (int*)malloc((size_t)(nmax + 1) * sizeof(int));
    j.h = (float*)malloc((size_t)(nmax + 1) * sizeof(float)); j.v = (float*)malloc((size_t)(nmax + 1) * sizeof(float));
    j.xt = (float*)malloc((size_t)(nmax + 1) * sizeof(float)); j.yt = (float*)malloc((size_t)(nmax + 1) * sizeof(float));
    j.acc = (long long*)calloc((size_t)nthreads, sizeof(long long));
    pthread_t* th = (pthread_t*)malloc((size_t)nthreads * sizeof(pthread_t));
    mt_arg* args = (mt_arg*)malloc((size_t)nthreads * sizeof(mt_arg));
    long long total = -1;
    if (j.idx && j.h && j.v && j.xt && j.yt && j.acc && th && args && pthread_barrier_init(&j.bar, NULL, (unsigned)nthreads) == 0) {
        int started = 0;
        for (int t = 0; t < nthreads; ++t) {
            args[t].job = &j; args[t].t = t;
            if (pthread_create(&th[t], NULL, mt_worker, &args[t]) != 0) break;
            ++started;
        }
        if (started == nthreads) {
            total = 0;
            for (int t = 0; t < nthreads; ++t) { pthread_join(th[t], NULL); total += j.acc[t]; }
        } else {
            /* cannot run a barrier with fewer threads than it was initialised for: give up loudly */
            for (int t = 0; t < started; ++t) pthread_cancel(th[t]);
            for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
        }
        pthread_barrier_destroy(&j.bar);
    }
    free(j.idx); free(j.h); free(j.v); free(j.xt); free(j.yt); free(j.acc); free(th); free(args);
    return total;
}
