/*
 * rectdetect-mi355x: entry points beyond the reference's own headers.  Plain C ABI (pointers and sizes only).
 *
 * The reference API (oclrect.h) takes HOST frames, so each call pays a PCIe upload.  The rd_* functions below let a
 * caller keep frames resident in HBM, run many frames through a multi-stream pipeline with the host post-process
 * on worker threads, and look at intermediate planes (tests).  Semantics of the detector are unchanged.
 */
#ifndef RECTDETECT_HIP_H
#define RECTDETECT_HIP_H
#include <stddef.h>
#include <stdint.h>
#if defined(__cplusplus)
extern "C" {
#endif

const char *rd_version(void);
int rd_device_count(void);                  /* number of HIP devices visible to this process */
void rd_select_device(int ordinal);         /* device used by simpleGetDevice(0) and new detectors (default 0) */
int rd_device_pci_bus_id(int ordinal, char *buf, int len);   /* "0000:c1:00.0" into buf (len >= 16); 0 on success: for pinning a per-GPU host process to the GPU's NUMA cores */

/* device memory helpers for callers without a HIP binding of their own (tests, bench) */
void *rd_device_alloc(size_t bytes);
void rd_device_free(void *dptr);
void *rd_host_alloc(size_t bytes);           /* pinned (page-locked) host memory: see RD_FRAME_HOST_PINNED */
void rd_host_free(void *p);
void rd_upload(void *dptr, const void *host, size_t bytes);
void rd_download(void *host, const void *dptr, size_t bytes);

/* ---- one detector instance = one stream of frames (state carried between frames) */
typedef struct rd_detector rd_detector;

/* nslots frames in flight (>= 1); nworkers != 0: the host post-process runs on worker threads, ONE PER SLOT (the value itself is
 * not a thread count), 0: on the polling thread.
 * 1-2 frames in flight: a frame spreads over two HIP streams (shortest latency); from 3 on: one stream per frame, and frames
 * beyond the fourth queue up on the same four streams (the device runs four hardware queues side by side).  From 6 slots on the
 * frames of two consecutive slots, from 12 on of four, from 32 on of eight, run as ONE set of launches (group launches: a frame waits until its group is
 * full or a poll asks for it - then it is launched on its own); 64 is what bench.py runs ("Data layout in HBM, execution"). */
rd_detector *rd_detector_create(int device, int iw, int ih, int nslots, int nworkers);
void rd_detector_destroy(rd_detector *d);

/* Detectors with one or two frames in flight take their HIP streams from a process-wide cache (hardware queues of their own: DESIGN.md) and hand them back when they
 * are destroyed, so that the next detector runs on the same queues.  This destroys the cached streams of `device` (-1: all devices); live detectors are not affected. */
void rd_release_cached_streams(int device);

/* Enqueue one BGR frame (row stride ws bytes).  Where the frame lies (`on_device`):
 *   RD_FRAME_HOST (0)         host memory of any kind; copied before the call returns (the reference's oclrect_enqueueTask does the same, oclrect.c:1256), the caller may
 *                             reuse the buffer at once
 *   RD_FRAME_DEVICE (1)       a device pointer, read in place: must stay valid and unchanged until the matching rd_detector_poll returned
 *   RD_FRAME_HOST_PINNED (2)  PINNED host memory (rd_host_alloc, allocatePinnedMemory of oclhelper.h, hipHostMalloc, or registered with hipHostRegister): the copy engine
 *                             reads it in place - no copy by the caller's thread (6 MB per 1920x1080 frame) - so it must stay unchanged until the matching poll returned,
 *                             like a device pointer; memory that is not pinned is refused (fatal)
 * Returns the frame's sequence number. */
#define RD_FRAME_HOST 0
#define RD_FRAME_DEVICE 1
#define RD_FRAME_HOST_PINNED 2
long rd_detector_enqueue(rd_detector *d, const void *frame, int ws, int on_device);

/* Result of the oldest frame not yet polled: malloc'd array of rect_t-compatible records (176 bytes each, element 0
 * holds nItems), owned by the caller.  Blocks until that frame is done. */
void *rd_detector_poll(rd_detector *d, double tanAOV);

/* optional: the aperture (tan of half the horizontal angle of view) the polls to come will pass.  The reference's API hands it over with
 * the poll, after the frame; work that runs ahead of the poll (worker threads, RD_DEVICE_POST=1) uses the last one seen, so without this
 * call the first frames of a stream are post-processed at poll time on the polling thread. */
void rd_detector_set_aperture(rd_detector *d, double tanAOV);

/* device-side work only (no host post-process): waits until every enqueued frame has left the GPU */
void rd_detector_drain(rd_detector *d);

/* copy of the line-segment list of the most recently POLLED frame: returns n (records 1..n), writes up to max records */
int rd_detector_last_segments(rd_detector *d, void *dst, int max_records);

/* counters since creation: which 0 = frames whose polyline stage did not fit the single-launch kernel's on-chip tables and
 * was repeated with the multi-launch path (same results, slower); 1 = device microseconds summed over the polled frames
 * (HIP events on the frame's stream: first kernel start to last copy end, so concurrent frames overlap); 2 = frames in that sum; 3 = host microseconds spent inside rd_detector_enqueue;
 * 4 = frames whose region merge had not settled within their launch budget and were repeated with more launches (32, then 64, then 128); 6 = frames still changing after 128 (none so far);
 * 5 = the current launch budget of the region merge (8, 10, .. 20); 20..26 = frames launched with a budget of 8 / 10 / .. / 20; 40 + k = frames whose merge needed k launches;
 * 10 = frames with more line segments than a slot's probe buffer holds (65535), whose probes were taken again into a larger buffer (the list keeps the
 * reference's capacity of 16N / 56 records; nothing is dropped); 11 / 12 = frames whose rectangles came from the device post-process (RD_DEVICE_POST=1:
 * candidate funnel + pose estimation in rd_k_post.hip) / from the host post-process; 13 = microseconds the worker threads spent in the
 * host post-process (sum over frames); 14 = frames whose small-region absorption (oclrect.cl:348-371) was finished by the slow path -
 * rounds over work lists until nothing changes - because they left more undecided pixels than the single-block tail holds (frames made of small regions);
 * 15 = frames per group launch; 16 / 17 = groups whose strong masks took one launch / one launch per frame; 18 = frames that travelled straight from the caller's pinned
 * memory (RD_FRAME_HOST_PINNED), 19 = host frames copied into the detector's own pinned staging first (RD_FRAME_HOST) */
long rd_detector_counter(rd_detector *d, int which);

/* ---- Environment.  Everything is read when a detector is created (rd_detector_create / init_oclrect) and never again; an empty value counts as unset.
 *   for users:   RD_DEVICE_POST=0|1      candidate funnel + pose estimation on the device (rd_k_post.hip) instead of the host threads (default: host, unless the process may
 *                                        use no more than two cores)
 *                RD_POST_HELPERS=n       helper threads that share a frame's pose estimations with the polling thread, reference-API shape only (default by core count, 0 = none)
 *                RD_NO_GRAPH=1           plain launches instead of captured hipGraphs
 *   test hooks:  RD_ZBATCH=k / RD_BATCH=k (frames per group launch / per set of sparse-stage launches), RD_REGION_ROUNDS_FIXED=8..20 and RD_BUDGET_CYCLE=n (launch budget of
 *                the region merge pinned / cycling), RD_POLY_MULTILAUNCH, RD_POLY_FORCE_REDO, RD_ABSORB_FORCE_SLOW (the fallback paths for every frame), RD_MAXREC_DEV=n (small
 *                probe buffers), RD_TEST_THRESHOLDS=a,b (strength thresholds), RD_STRONG_BY_FRAME (a group's strong masks frame by frame), RD_IIR_FORCE_FIX (every blur column
 *                through the full-length path; read per call).  Each is driven by a test in tests/test_gpu_parity.py.
 * Switches of experiments that were measured and not kept exist only in tuning builds (-DRD_TUNING, tools/variants.sh). */

/* Test hook: copy an internal plane of the most recently completed frame to host memory.  Returns bytes written,
 * 0 for an unknown name.  Names: plab0 plab1 lblur vxy strength nms mask0 tidy label1 strsum edge500 smooth quant
 * strong junction mergemask region0 (merged regions) rsize region (after absorbing small ones) boundarysrc boundary lsid table */
size_t rd_detector_debug_plane(rd_detector *d, const char *name, void *dst, size_t max_bytes);

/* ---- host post-process alone (oclrect.c:1049-1226 restated): segments + samples -> rectangles.  Used by tests to
 * check the post-process against the reference with identical inputs.  `boundary` is the boundary-label plane,
 * `table` the reduceLS table (iw*ih*4/5 entries of 5 ints), `segs` the linesegment_t list with header. */
void *rd_postprocess_planes(const void *segs, const int32_t *boundary, const int32_t *table, int iw, int ih, double tanAOV);
/* test tap: the 15 probe pixels of a segment as the sampling kernel computes them (x, y pairs; (-1, -1): outside the frame) - compared with the oracle's
 * independent restatement of oclrect.c:1066-1083 */
void rd_probe_pixels(float x0, float y0, float x1, float y1, int iw, int ih, int32_t *out);

/* ---- synthetic frames (csrc/rd_synth.c) */
int rd_synth_num_quads(int iw, int ih);
void rd_synth_frame(uint8_t *bgr, int iw, int ih, int ws, uint64_t seed, int t, int noise);

#if defined(__cplusplus)
}
#endif
#endif
