/* rectdetect-mi355x: host post-process entry (rd_post.c) */
#ifndef RD_POST_H
#define RD_POST_H
#if defined(__cplusplus)
extern "C" {
#endif
/* segs: linesegment_t array with header record (n = first int), at most max_records records readable;
 * probes: for segment i and probe k (0..14) six ints at probes[(i*15+k)*6]: {boundary id, slot owner, 4 box values}.
 * Returns a malloc'd rect_t array (element 0: nItems). */
void *rd_post_run(const void *segs, int max_records, const int *probes, int iw, int ih, double tanAOV);
#if defined(__cplusplus)
}
#endif
#endif
