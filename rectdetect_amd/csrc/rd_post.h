/* rectdetect-mi355x: host post-process entry (rd_post.c) */
#ifndef RD_POST_H
#define RD_POST_H
#include <stddef.h>
#if defined(__cplusplus)
extern "C" {
#endif
/* segs: linesegment_t array with header record (n = first int), at most max_records records readable;
 * probes: for segment i and probe k (0..14) six ints at probes[(i*15+k)*6]: {boundary id, slot owner, 4 box values}.
 * Returns a malloc'd rect_t array (element 0: nItems). */
void *rd_post_run(const void *segs, int max_records, const int *probes, int iw, int ih, double tanAOV);
/* Helper threads for the pose estimations of a frame's candidates (rd_post.c: "side by side").  configure(n): at least n helpers exist from now
 * on (process-wide, never fewer again; 0 = none, the default: rd_post_run then runs everything on its caller's thread).  arm(): the caller is
 * about to wait for a frame whose post-process it will run itself - the helpers wake and spin for at most RD_POST_SPIN_US microseconds. */
#define RD_POST_MAX_HELPERS 7
#define RD_POST_SPIN_US 2000.0
#define RD_POST_HOT_US 300.0      /* of which a tight spin; then sched_yield between two looks */
void rd_post_helpers_configure(int n);
void rd_post_helpers_arm(void);
int rd_post_helpers(void);
/* asks the helpers to end and joins them (also run when the library is unloaded); configure() may start new ones afterwards */
void rd_post_helpers_shutdown(void);
/* fn(ctx, 0) .. fn(ctx, n - 1), each once, claimed in index order by the caller's thread and by whichever helpers are awake; returns when all have run.
 * progress (may be NULL) is called on the CALLER's thread after each of its own jobs and while it waits for the helpers' last ones. */
typedef void (*rd_job_fn)(void *ctx, int idx);
typedef void (*rd_progress_fn)(void *ctx);
void rd_helpers_run(rd_job_fn fn, void *ctx, int n, rd_progress_fn progress);
/* memcpy into a staging buffer nobody on the host reads again (the DMA engine does): non-temporal stores - no read-for-ownership of the destination, the
 * caller's cache keeps its contents.  Falls back to memcpy without AVX2. */
void rd_copy_to_staging(void *dst, const void *src, size_t n);
#if defined(__cplusplus)
}
#endif
#endif
