/* How many cores does this box really give a process?  N threads each run the same fixed register-only loop; the
 * aggregate rate relative to one thread is the number of cores' worth of CPU the N threads received.  (The GPU boxes
 * report 256 hardware threads; the drop-in scaling figures only make sense next to this number.)
 * Build: gcc -O2 scripts/cpu_scaling.c -o scripts/cpu_scaling -lpthread */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

static pthread_barrier_t gate;
static volatile unsigned long sink;

static void *worker(void *arg) {
  const unsigned long iters = *(unsigned long *)arg;
  pthread_barrier_wait(&gate);
  unsigned long x = 88172645463325252ul;
  for (unsigned long i = 0; i < iters; i++)
    x ^= x << 13, x ^= x >> 7, x ^= x << 17;
  sink += x;
  pthread_barrier_wait(&gate);
  return NULL;
}

static double now(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

int main(int argc, char **argv) {
  unsigned long iters = argc > 1 ? strtoul(argv[1], NULL, 0) : 300000000ul;
  double base = 0;
  for (int n = 1; n <= 256; n *= 2) {
    pthread_t tid[256];
    pthread_barrier_init(&gate, NULL, (unsigned)n + 1);
    for (int t = 0; t < n; t++)
      pthread_create(&tid[t], NULL, worker, &iters);
    pthread_barrier_wait(&gate);
    const double t0 = now();
    pthread_barrier_wait(&gate);
    const double dt = now() - t0;
    for (int t = 0; t < n; t++)
      pthread_join(tid[t], NULL);
    pthread_barrier_destroy(&gate);
    if (n == 1)
      base = dt;
    printf("%3d threads: %.3f s for the same work per thread -> %.1f cores' worth\n", n, dt, n * base / dt);
  }
  return 0;
}
