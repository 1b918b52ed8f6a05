/*
 * TEST INFRASTRUCTURE.  The product's host-side C (rectdetect_amd/csrc/rd_post.c, rd_helper.c, rd_synth.c) under the sanitizers - SURVEY.md section 5, row 2.
 * Built by `make -C tests/native tsan` / `asan` together with the oracle (which only produces the inputs: segments, boundary plane, vote table of
 * two small synthetic frames) into build/host_stress_tsan / host_stress_asan and run by tests/test_cpu_sanitizers.py.
 *
 *   1. the post-process with helper threads returns the bytes of the caller's thread alone (armed, not armed, 0..7 helpers);
 *   2. arm / run / shut down / configure again, 1000 cycles, 0-7 helpers, TWO caller threads contending for the pool (one finds it taken and runs alone),
 *      every job of every batch run exactly once;
 *   3. the helper pool survives rd_post_helpers_shutdown() while armed helpers spin, and a fork() (the child starts with an empty pool).
 * Exit code 0 and "host_stress: ok" on success; any sanitizer report makes the process fail (halt_on_error).
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#include "rd_post.h"
#include "rectdetect_hip.h"
#include "rd_oracle.h"

static void fail(const char *what) { fprintf(stderr, "host_stress: FAILED: %s\n", what); exit(1); }

typedef struct { int iw, ih; void *segs; int *boundary, *table; void *want; size_t want_bytes; } frame_case;

static void *rects(const frame_case *c, size_t *bytes) {
  void *r = rd_postprocess_planes(c->segs, c->boundary, c->table, c->iw, c->ih, 0.7);
  *bytes = (size_t)((int *)r)[0] * 176;
  return r;
}

static void make_case(frame_case *c, int iw, int ih, uint64_t seed) {
  const size_t N = (size_t)iw * ih;
  uint8_t *img = (uint8_t *)malloc(N * 3);
  rd_synth_frame(img, iw, ih, iw * 3, seed, 0, 1);
  rdo_rect_t *o = rdo_rect_new(iw, ih);
  rdo_rect_frame(o, img, iw * 3);
  const int n = ((int *)o->lslist)[0];
  c->iw = iw; c->ih = ih;
  c->segs = malloc((size_t)(n + 1) * 56); memcpy(c->segs, o->lslist, (size_t)(n + 1) * 56);
  c->boundary = (int *)malloc(N * 4); memcpy(c->boundary, o->boundary, N * 4);
  c->table = (int *)malloc(N * 16); memcpy(c->table, o->table, N * 16);
  rdo_rect_free(o);
  free(img);
  c->want = rects(c, &c->want_bytes);
}

static frame_case cases[2];

static void *same_bytes_worker(void *arg) {
  const frame_case *c = (const frame_case *)arg;
  for (int rep = 0; rep < 12; rep++) {
    if (rep & 1) rd_post_helpers_arm();
    size_t b; void *r = rects(c, &b);
    if (b != c->want_bytes || memcmp(r, c->want, b)) fail("helper threads changed the rectangle list");
    free(r);
  }
  return NULL;
}

/* ---- the pool itself: batches of jobs that count how often each ran */
typedef struct { int n; int *ran; } batch;
static void job(void *ctx, int i) { batch *b = (batch *)ctx; __atomic_add_fetch(&b->ran[i], 1, __ATOMIC_RELAXED); for (volatile int k = 0; k < 200; k++) { } }
static int stop_callers;
static void *caller(void *arg) {
  unsigned r = (unsigned)(uintptr_t)arg * 2654435761u + 12345u;
  long batches = 0;
  while (!__atomic_load_n(&stop_callers, __ATOMIC_ACQUIRE)) {
    r = r * 1664525u + 1013904223u;
    batch b; b.n = 1 + (int)((r >> 16) % 24);
    b.ran = (int *)calloc((size_t)b.n, sizeof(int));
    if ((r >> 8) & 1) rd_post_helpers_arm();
    rd_helpers_run(job, &b, b.n, NULL);
    for (int i = 0; i < b.n; i++) if (__atomic_load_n(&b.ran[i], __ATOMIC_RELAXED) != 1) fail("a job of a batch ran zero times or twice");
    free(b.ran);
    batches++;
  }
  return (void *)(uintptr_t)batches;
}

int main(int argc, char **argv) {
  const int cycles = argc > 1 ? atoi(argv[1]) : 1000;
  make_case(&cases[0], 333, 217, 0x5EED0002ull);
  make_case(&cases[1], 640, 480, 0x5EED0005ull);
  if (cases[1].want_bytes < 2 * 176) fail("the frames should have candidates");

  /* 1. same bytes, 0..7 helpers, one and two callers */
  for (int helpers = 0; helpers <= RD_POST_MAX_HELPERS; helpers += helpers < 2 ? 1 : 2) {
    rd_post_helpers_configure(helpers);
    if (rd_post_helpers() < helpers) fail("helpers did not start");
    same_bytes_worker(&cases[1]);
    pthread_t th[2];
    for (int i = 0; i < 2; i++) pthread_create(&th[i], NULL, same_bytes_worker, &cases[i]);
    for (int i = 0; i < 2; i++) pthread_join(th[i], NULL);
  }

  /* 2. arm / run / shut down / configure again while two callers keep the pool busy */
  pthread_t cth[2];
  for (int i = 0; i < 2; i++) pthread_create(&cth[i], NULL, caller, (void *)(uintptr_t)(i + 1));
  unsigned r = 99;
  for (int c = 0; c < cycles; c++) {
    r = r * 1664525u + 1013904223u;
    rd_post_helpers_configure((int)((r >> 20) % (RD_POST_MAX_HELPERS + 1)));
    rd_post_helpers_arm();
    if ((r >> 12) % 3 == 0) usleep(50);
    if (c % 4 == 3) rd_post_helpers_shutdown();
  }
  __atomic_store_n(&stop_callers, 1, __ATOMIC_RELEASE);
  long total = 0;
  for (int i = 0; i < 2; i++) { void *ret; pthread_join(cth[i], &ret); total += (long)(uintptr_t)ret; }
  if (total < 10) fail("the callers ran no batches");

  /* 3. fork with helpers alive and armed: the child has none and runs everything itself */
  rd_post_helpers_configure(3);
  rd_post_helpers_arm();
  const pid_t pid = fork();
  if (pid == 0) {
    if (rd_post_helpers() != 0) _exit(2);
    size_t b; void *rr = rects(&cases[0], &b);
    const int ok = b == cases[0].want_bytes && !memcmp(rr, cases[0].want, b);
    free(rr);
#if !defined(__SANITIZE_THREAD__)      /* (ThreadSanitizer's runtime cannot start threads in the child of a multi-threaded fork; the other builds do) */
    rd_post_helpers_configure(2);
#endif
    size_t b2; void *r2 = rects(&cases[1], &b2);
    const int ok2 = b2 == cases[1].want_bytes && !memcmp(r2, cases[1].want, b2);
    free(r2);
    rd_post_helpers_shutdown();
    _exit(ok && ok2 ? 0 : 3);
  }
  int status = 0;
  waitpid(pid, &status, 0);
  if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) fail("the child of a fork did not get through its frames");
  rd_post_helpers_shutdown();
  for (int i = 0; i < 2; i++) { free(cases[i].segs); free(cases[i].boundary); free(cases[i].table); free(cases[i].want); }
  printf("host_stress: ok (%d cycles, %ld batches by two callers)\n", cycles, total);
  return 0;
}
