/*
 * time_mt.c -- multi-thread timing harness of the CPU baseline (bench.py's `cpu_baseline` leg).
 *
 * TEST INFRASTRUCTURE ONLY, like everything under oracle/.  It times a q32-shaped plane function --
 * the REAL reference's tier function through oracle/_ref's ref_call_tier, or this directory's
 * restatement orc_q32_avx -- the way SURVEY.md 8(d) asks: N persistent threads, each pinned to its own
 * CPU (pthread_setaffinity_np, as the reference's harness pins itself, main.cpp:252-257), over disjoint
 * block-row ranges through the reference's own startY/endY hook (simd_dct.cpp:2245-2255), the whole plane
 * per run via the sizeY = 2H call form, `warmups` untimed runs then `runs` timed ones.  One clock pair per
 * run around two barriers, read by worker 0 (which owns a CPU; a coordinating thread without one would read
 * its start clock late when every CPU is busy); no Python, no thread creation inside a timed run.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <time.h>

typedef int (*tier_fn)(int which, const uint8_t *, uint8_t *, const float *, size_t, size_t, size_t, size_t);
typedef int (*plane_fn)(const uint8_t *, uint8_t *, const float *, size_t, size_t, size_t, size_t);

struct shared
{
  void *fn;
  int which; /* >= 0: fn is a tier_fn called with this id; < 0: fn is a plane_fn */
  const uint8_t *from;
  uint8_t *to;
  const float *lut;
  size_t W, H;
  int iterations, warmups;
  double *seconds;
  int nthreads;          /* the threads that really started (set before `go` is released) */
  pthread_mutex_t go;    /* held by the caller while it creates the threads */
  pthread_barrier_t bar; /* initialised for nthreads before `go` is released */
};

struct worker
{
  struct shared *s;
  int index;
  int cpu; /* -1: not pinned */
  int pinned;
};

static void *work(void *p)
{
  struct worker *w = (struct worker *)p;
  struct shared *s = w->s;
  if (w->cpu >= 0)
  {
    cpu_set_t set;
    CPU_ZERO(&set);
    CPU_SET(w->cpu, &set);
    w->pinned = pthread_setaffinity_np(pthread_self(), sizeof(set), &set) == 0;
  }
  pthread_mutex_lock(&s->go);
  pthread_mutex_unlock(&s->go);
  const size_t rows = s->H / 8, row0 = rows * (size_t)w->index / (size_t)s->nthreads, row1 = rows * (size_t)(w->index + 1) / (size_t)s->nthreads;
  /* block row r is processed iff startY <= 2 * (8 r) <= endY (simd_dct.cpp:2247), sizeY = 2H so that y < sizeY / 2 covers the plane */
  const size_t y0 = 16 * row0, y1 = 16 * row1 - 16;
  for (int it = 0; it < s->iterations; it++)
  {
    struct timespec a, b;
    pthread_barrier_wait(&s->bar);
    if (w->index == 0)
      clock_gettime(CLOCK_MONOTONIC, &a);
    if (row1 > row0)
    {
      if (s->which >= 0)
        ((tier_fn)s->fn)(s->which, s->from, s->to, s->lut, s->W, 2 * s->H, y0, y1);
      else
        ((plane_fn)s->fn)(s->from, s->to, s->lut, s->W, 2 * s->H, y0, y1);
    }
    pthread_barrier_wait(&s->bar);
    if (w->index == 0 && it >= s->warmups)
    {
      clock_gettime(CLOCK_MONOTONIC, &b);
      s->seconds[it - s->warmups] = (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
    }
  }
  return NULL;
}

/* seconds[runs] <- wall time of every timed run.  cpus[ncpus]: the CPUs to pin to (thread i -> cpus[i % ncpus]), ncpus == 0: no pinning.
 * Returns the number of threads that were really pinned, or < 0 on error. */
int orc_time_q32_mt(void *fn, int which, const uint8_t *from, uint8_t *to, const float *lut, size_t W, size_t H,
                    int nthreads, const int *cpus, int ncpus, int warmups, int runs, double *seconds)
{
  if (!fn || !from || !to || !lut || nthreads < 1 || runs < 1 || warmups < 0 || (H % 8) || (W % 64))
    return -1;
  struct shared s;
  s.fn = fn, s.which = which, s.from = from, s.to = to, s.lut = lut, s.W = W, s.H = H, s.iterations = warmups + runs, s.warmups = warmups, s.seconds = seconds;
  pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
  struct worker *ws = (struct worker *)calloc((size_t)nthreads, sizeof(struct worker));
  if (!th || !ws)
  {
    free(th);
    free(ws);
    return -2;
  }
  pthread_mutex_init(&s.go, NULL);
  pthread_mutex_lock(&s.go);
  int started = 0;
  for (int i = 0; i < nthreads; i++)
  {
    ws[i].s = &s;
    ws[i].index = i;
    ws[i].cpu = ncpus > 0 ? cpus[i % ncpus] : -1;
    if (pthread_create(&th[i], NULL, work, &ws[i]))
      break;
    started++;
  }
  /* a host that refuses some threads still gets a run: the plane is cut over the threads that exist */
  s.nthreads = started;
  int rc = 0;
  if (!started || pthread_barrier_init(&s.bar, NULL, (unsigned)started))
  {
    s.iterations = 0; /* the threads fall straight through */
    pthread_mutex_unlock(&s.go);
    for (int i = 0; i < started; i++)
      pthread_join(th[i], NULL);
    rc = -3;
  }
  else
  {
    pthread_mutex_unlock(&s.go);
    for (int i = 0; i < started; i++)
    {
      pthread_join(th[i], NULL);
      rc += ws[i].pinned;
    }
    pthread_barrier_destroy(&s.bar);
  }
  pthread_mutex_destroy(&s.go);
  free(th);
  free(ws);
  return rc;
}
