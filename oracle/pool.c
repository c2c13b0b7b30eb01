/*
 * pool.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A persistent worker pool standing in for cv::parallel_for_ (OpenCV keeps its worker threads alive between
 * calls; a pthread_create / join per call, as the first version of this oracle did, charges the CPU baseline
 * ~50 us per parallel region that the real OpenCV never pays).  Used by the CLAHE, pyramid and LK loops the
 * reference parallelises through OpenCV: cv::CLAHE::apply (CLAHE_CalcLut_Body / CLAHE_Interpolation_Body),
 * cv::pyrDown, calcSharrDeriv rows, LKTrackerInvoker over points.  Every loop body is independent of the
 * split, so results do not depend on the thread count (tests run with 1 and with several threads).
 */
#include "ov2_oracle.h"
#include <pthread.h>
#include <stdlib.h>

#define ORC_MAX_THREADS 256

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;       /* protects the job description */
static pthread_cond_t g_cv_work = PTHREAD_COND_INITIALIZER, g_cv_done = PTHREAD_COND_INITIALIZER;
static pthread_mutex_t g_call = PTHREAD_MUTEX_INITIALIZER;     /* one parallel region at a time */
static pthread_t g_thr[ORC_MAX_THREADS];
static int g_nworkers = 0;             /* threads created so far (workers 1..g_nworkers) */
static int g_nthreads = 1;             /* threads a parallel region uses (incl. the caller) */
static unsigned long g_epoch = 0;
static struct { orc_range_fn fn; void *ctx; int n, parts, next, pending; } g_job;

static void run_parts(void)
{
    for (;;) {
        pthread_mutex_lock(&g_mu);
        const int p = g_job.next < g_job.parts ? g_job.next++ : -1;
        pthread_mutex_unlock(&g_mu);
        if (p < 0) return;
        const int per = (g_job.n + g_job.parts - 1) / g_job.parts;
        int b = p * per, e = b + per;
        if (e > g_job.n) e = g_job.n;
        if (b < e) g_job.fn(b, e, g_job.ctx);
        pthread_mutex_lock(&g_mu);
        if (--g_job.pending == 0) pthread_cond_broadcast(&g_cv_done);
        pthread_mutex_unlock(&g_mu);
    }
}

static void *worker(void *arg)
{
    (void)arg;
    unsigned long seen = 0;
    for (;;) {
        pthread_mutex_lock(&g_mu);
        while (g_epoch == seen) pthread_cond_wait(&g_cv_work, &g_mu);
        seen = g_epoch;
        pthread_mutex_unlock(&g_mu);
        run_parts();
    }
    return NULL;
}

void orc_set_num_threads(int n)
{
    if (n < 1) n = 1;
    if (n > ORC_MAX_THREADS) n = ORC_MAX_THREADS;
    pthread_mutex_lock(&g_call);
    while (g_nworkers < n - 1) {
        if (pthread_create(&g_thr[g_nworkers], NULL, worker, NULL) != 0) break;
        pthread_detach(g_thr[g_nworkers]);
        g_nworkers++;
    }
    g_nthreads = g_nworkers + 1 < n ? g_nworkers + 1 : n;
    pthread_mutex_unlock(&g_call);
}

int orc_get_num_threads(void) { return g_nthreads; }

void orc_parallel_for(int n, orc_range_fn fn, void *ctx, int min_grain)
{
    if (n <= 0) return;
    if (min_grain < 1) min_grain = 1;
    int parts = g_nthreads;
    if (parts > (n + min_grain - 1) / min_grain) parts = (n + min_grain - 1) / min_grain;
    if (parts <= 1) { fn(0, n, ctx); return; }
    pthread_mutex_lock(&g_call);
    pthread_mutex_lock(&g_mu);
    g_job.fn = fn; g_job.ctx = ctx; g_job.n = n; g_job.parts = parts; g_job.next = 0; g_job.pending = parts;
    g_epoch++;
    pthread_cond_broadcast(&g_cv_work);
    pthread_mutex_unlock(&g_mu);
    run_parts();                                   /* the caller works too */
    pthread_mutex_lock(&g_mu);
    while (g_job.pending > 0) pthread_cond_wait(&g_cv_done, &g_mu);
    pthread_mutex_unlock(&g_mu);
    pthread_mutex_unlock(&g_call);
}
