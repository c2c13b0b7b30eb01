/*
 * pool.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A persistent worker pool standing in for cv::parallel_for_ (OpenCV keeps its worker threads alive between
 * calls; a pthread_create / join per call, as the first version of this oracle did, charges the CPU baseline
 * ~50 us per parallel region that the real OpenCV never pays).  Used by the CLAHE, pyramid and LK loops the
 * reference parallelises through OpenCV: cv::CLAHE::apply (CLAHE_CalcLut_Body / CLAHE_Interpolation_Body),
 * cv::pyrDown, calcSharrDeriv rows, LKTrackerInvoker over points.  Every loop body is independent of the
 * split, so results do not depend on the thread count (tests run with 1 and with several threads).
 *
 * Static partition, one semaphore per worker: a region wakes exactly the workers it uses (a broadcast on a
 * 256-core host woke every idle worker and spent more time in futex calls than in the loop bodies).
 */
#include "ov2_oracle.h"
#include <pthread.h>
#include <semaphore.h>
#include <stdlib.h>

#define ORC_MAX_THREADS 256

typedef struct { sem_t go; int index; } orc_worker;

static pthread_mutex_t g_call = PTHREAD_MUTEX_INITIALIZER;     /* one parallel region at a time */
static orc_worker g_w[ORC_MAX_THREADS];
static sem_t g_done;
static int g_done_init = 0;
static int g_nworkers = 0;             /* threads created so far */
static int g_nthreads = 1;             /* threads a parallel region uses (incl. the caller) */
static struct { orc_range_fn fn; void *ctx; int n, parts; } g_job;

static void run_part(int p)
{
    const int per = (g_job.n + g_job.parts - 1) / g_job.parts;
    int b = p * per, e = b + per;
    if (e > g_job.n) e = g_job.n;
    if (b < e) g_job.fn(b, e, g_job.ctx);
}

static void *worker(void *arg)
{
    orc_worker *w = (orc_worker *)arg;
    for (;;) {
        sem_wait(&w->go);
        run_part(w->index + 1);          /* part 0 belongs to the caller */
        sem_post(&g_done);
    }
    return NULL;
}

void orc_set_num_threads(int n)
{
    if (n < 1) n = 1;
    if (n > ORC_MAX_THREADS) n = ORC_MAX_THREADS;
    pthread_mutex_lock(&g_call);
    if (!g_done_init) { sem_init(&g_done, 0, 0); g_done_init = 1; }
    while (g_nworkers < n - 1) {
        orc_worker *w = &g_w[g_nworkers];
        pthread_t th;
        w->index = g_nworkers;
        sem_init(&w->go, 0, 0);
        if (pthread_create(&th, NULL, worker, w) != 0) break;
        pthread_detach(th);
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
    if (parts > g_nworkers + 1) parts = g_nworkers + 1;
    g_job.fn = fn; g_job.ctx = ctx; g_job.n = n; g_job.parts = parts;
    for (int i = 0; i < parts - 1; i++) sem_post(&g_w[i].go);      /* sem_post is a release: the job description is visible */
    run_part(0);                                                   /* the caller works too */
    for (int i = 0; i < parts - 1; i++) sem_wait(&g_done);
    pthread_mutex_unlock(&g_call);
}
