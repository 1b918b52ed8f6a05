// rectdetect-mi355x: host-callable launchers of the gfx950 kernels (implemented in rd_k_*.hip).
// All pointers are device pointers; every launcher enqueues on `s` and returns immediately.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

// Group launches (small frames, rd_api.hip): launchers of the frame path take `nz` frames per launch and `zs`, the byte pitch between the planes
// of consecutive frame slots (rd_device.h: RD_ZSHIFT); the defaults launch one frame.
#define RD_ZB_MAX 8

namespace rdk {

// Kernels with more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised once PER DEVICE (a process may drive
// several GPUs from several threads): `done` holds one bit per device ordinal; a race only repeats an idempotent call.
inline void set_max_lds_once(const void *func, int bytes, std::atomic<unsigned> &done) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned bit = 1u << (dev & 31);
  if (done.load(std::memory_order_acquire) & bit) return;
  (void)hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  done.fetch_or(bit, std::memory_order_release);
}

// ---- rd_k_front.hip: colour, blur, gradient, non-max suppression, element-wise ops
void bgr2plab(hipStream_t s, uint32_t *out, const uint8_t *bgr, int iw, int ih, int ws);
// colour conversion that also leaves the unpacked L, a, b planes transposed (ih wide, iw tall) for the first blur sweep
void bgr2plab_transposed(hipStream_t s, uint32_t *out, float *const dst[3], const uint8_t *bgr, int iw, int ih, int ws);
void bgr2plab_transposed(hipStream_t s, uint32_t *out, float *const dst[3], const uint8_t *const *bgr, int iw, int ih, int ws, int nz, size_t zs);   // bgr: nz source frames   // dst: the three planes transposed, as 16-bit integer fields (iir_blur_pass src16)
void unpack_plab(hipStream_t s, float *L, float *a, float *b, const uint32_t *in, int n);
void pack_plab(hipStream_t s, uint32_t *out, const float *L, const float *a, const float *b, int n);
// transposes of `np` float planes (src planes W x H row-major -> dst planes H x W); src may be packed Lab (np = 3)
void transpose_f(hipStream_t s, float *const dst[3], const float *const src[3], int np, int W, int H);
// one complete blur pass (causal + anti-causal sweep + combination) in a single launch, see rd_k_front.hip
size_t iir_pass_scratch_floats(int np, int W, int H);
void iir_blur_pass(hipStream_t s, float *const dst[3], const float *const src[3], float *const fwd[3], float *const bwd[3], int np, int W, int H,
                   int transpose_out, float *tails, int *bad, int src16 = 0, int nz = 1, size_t zs = 0);
void iir_blur_lines(hipStream_t s, float *dst, const float *src, float *fw, float *bw, int W, int H, int r);   // any radius 0..31, full-length sweeps along y; dst may be fw or bw
#define RD_IIR_MAX_R 31
void edgevec(hipStream_t s, float *vxy, const float *in, int iw, int ih, uint32_t *pack_out = nullptr, const float *a = nullptr, const float *b = nullptr, int nz = 1, size_t zs = 0);   // pack_out (optional): pack_plab(in, a, b) on the way
// visualisers / operators no application calls (oclimgutil.h:84-98)
void convert_bgr_lumaf(hipStream_t s, uint8_t *out, const float *in, float f, int iw, int ih, int ws);
void convert_bgr_labeli(hipStream_t s, uint8_t *out, const int *in, int bgc, int iw, int ih, int ws);
void plab2bgr(hipStream_t s, uint8_t *out, const uint32_t *in, int iw, int ih, int ws);
void edge_f(hipStream_t s, float *out, const float *in, int iw, int ih);
void edgevec_plab(hipStream_t s, float *vxy, const uint32_t *in, int iw, int ih);
void thincubic(hipStream_t s, float *out, const float *in, const float *vxy, int iw, int ih);
void edge_plab(hipStream_t s, float *out, const uint32_t *in, int iw, int ih, int nz = 1, size_t zs = 0);
void thinthres(hipStream_t s, float *out, const float *in, const float *vxy, int iw, int ih, int nz = 1, size_t zs = 0);
// gradient direction + re-packing + strength + non-max suppression of the frame path in one launch (bl: the three blurred planes); taps (all or none):
// the intermediate planes plab1 / vxy / strength are written as well (tests, debug planes)
int grad_nms_fits(int iw, int ih);
void grad_nms(hipStream_t s, float *nms, float *const bl[3], int iw, int ih, int nz = 1, size_t zs = 0, uint32_t *tap_plab1 = nullptr, float *tap_vxy = nullptr, float *tap_strength = nullptr);
void threshold_f(hipStream_t s, float *out, const float *in, float lo, float thr, float hi, int n);
void threshold_i(hipStream_t s, int *out, const int *in, int lo, int thr, int hi, int n);
void cast_i_f(hipStream_t s, int *out, const float *in, float scale, int n);
void cast_c_i(hipStream_t s, int8_t *out, const int *in, int n);
void clear_i(hipStream_t s, int *out, int n);
void copy_i(hipStream_t s, int *out, const int *in, int n);
void rand_i(hipStream_t s, int *out, uint64_t seed, int n);

// ---- rd_k_label.hip: connected components and per-label reductions
// 8-connected components of equal `pix` value, pixels equal to bgc -> -1, label = smallest pixel index
void label8(hipStream_t s, int *label, const int *pix, int bgc, int iw, int ih, int skip_flatten = 0);   // label = smallest index of the 8-connected component of equal value, -1 for bgc; skip_flatten: the final walk to the roots is left to calc_strength(flatten = 1)
void label8_tidy(hipStream_t s, int *label, int *mask0, int *tidy, const float *nms, int *zero_plane, int iw, int ih, int skip_flatten = 0, int nz = 1, size_t zs = 0);   // rect_tidy + label8(tidy, background -1) in the same tile kernel
// flatten = 0: the component plane is left as the union-find forest of the tile and border kernels and its readers - votes, probes - walk to the roots themselves
// (-DRD_BOUNDARY_FLATTEN=0, tuning builds).  Built in round 6, lists identical, and NOT the default: k_label_flatten 5.6 -> 0.9 us per frame, but the 7x7 windows of
// k_reduce_claim then walk (7.6 -> 14.4 us) - 2883-2899 against 2879-2892 frames/s, nothing.  profiles/NOTES_r06.md.
#ifndef RD_BOUNDARY_FLATTEN
#define RD_BOUNDARY_FLATTEN 1
#endif
void label8_boundary(hipStream_t s, int *label, int *marks, const int *region, int iw, int ih, int *vt_table = nullptr, int *vt_claim = nullptr, int *vt_list = nullptr, int nz = 1, size_t zs = 0, int flatten = 1);
void label8_flatten(hipStream_t s, int *label, int n);      // phase 3 alone, later (debug plane)   // mark_boundary + label8(marks, background -1) with the marking fused into the tile kernel
// add (optional): a plane whose non-zero elements are added to out element by element in the same launch (out = zeros + add + sums)
void calc_strength(hipStream_t s, int *out, const float *edge, int *label, int iw, int ih, const int8_t *add = nullptr, int flatten = 0, int nz = 1, size_t zs = 0);   // add (optional): a 0/1 byte plane added to the sums (H1)
void filter_strength(hipStream_t s, int *label, const int *str, int thre, int iw, int ih);
// strong mask at t_strong (two copies) + edge mask at t_edge (int, int8), both from the unfiltered labels, + filter_strength at t_strong (label in place), one pass; t_edge <= t_strong
// prev (optional): a 0/1 byte plane added to the sums element by element (sum of label l = str[l] + prev[l]: quirk H1 without a pass of its own); strong2 != prev
// the strong masks of the nz frames of a group launch (frame z = sequence number t0 + z, planes zs bytes apart, masks in ring planes (t0 + z + 1) mod nring) in one launch
int strength_masks_group_fits(int iw, const void *label, const void *ring, const void *edge8, size_t zs);
void strength_masks_group(hipStream_t s, int8_t *ring, int8_t *edge8, int *label, const int *str, int t_edge, int t_strong, int iw, int ih, unsigned long long *bits, long t0, int nring, int nz, size_t zs);
void strength_masks(hipStream_t s, int *strong, int8_t *strong2, int *edge /* may be NULL */, int8_t *edge8, int *label, const int *str, int t_edge, int t_strong, int iw, int ih, const int8_t *prev = nullptr, unsigned long long *bits = nullptr);   // bits (optional): the strong mask as a bit plane too (ceil(iw / 64) words per row); strong may then be null

// ---- rd_k_rect.hip: rect-path stages
// "counted" / "curve end" bit rows of the strong mask's junction counts (oclrect.cl:74-95), 2 words per 64 pixels: what merge_mask() reads
void junction_bits(hipStream_t s, unsigned long long *bits, const unsigned long long *strong, int iw, int ih, int nz = 1, size_t zs = 0);
// run extents of the edge-stopped blur (depend on the edge mask only): ext[p] = nl_h | nr_h<<3 | nl_v<<6 | nr_v<<9
void blblur_extents(hipStream_t s, uint16_t *ext, const int8_t *edge, int iw, int ih, int nz = 1, size_t zs = 0);
// one horizontal + vertical pass pair; out must not alias in
void blblur_pair(hipStream_t s, uint32_t *out, const uint16_t *ext, const uint32_t *in, int iw, int ih, int nz = 1, size_t zs = 0);
void quant_lut_init(hipStream_t s);   // once per device before the first despeckle(quantize24 = 1): builds the 24-level quantisation tables on the device
void despeckle(hipStream_t s, uint32_t *out, const uint32_t *in, const float *edge, int iw, int ih, int quantize24, int nz = 1, size_t zs = 0);   // quantize24: `in` is quantised to 24 levels per field on the fly
void merge_mask(hipStream_t s, unsigned long long *out, const unsigned long long *bits, int iw, int ih, int nz = 1, size_t zs = 0);   // out: bit plane, ceil(iw / 64) words per row
// oclrect.cl:289-334 with the work-items of a launch concurrent (rd_k_rect.hip: k_region_init = the links and launch 0, k_region_round = the others).
// mask / edge: the merge mask and the strong mask as bit planes.
// launches: even, 2..64 (launches after one that changed nothing return at once; flag r of scratch[N + r] = launch r changed something);
// size_out (optional) <- the junction counts of the strong mask (what the reference's size plane holds when the counting starts: quirk H2), for region_size;
// scratch: 3N + 256 ints; *marked <- 1: `label` is left as label << 3 | mark words, which region_size(…, marked) turns into plain labels
void region_merge(hipStream_t s, int *label, int *scratch, const int *pix, const unsigned long long *mask, const unsigned long long *edge, int iw, int ih, int launches,
                  int *size_out, int *marked, int nz = 1, size_t zs = 0);
void region_size(hipStream_t s, int *out, int *label, int n, int *zero_me, int marked = 0, int nz = 1, size_t zs = 0);   // accumulates into out; zero_me (optional): an int to clear on the way
#define RD_D2_SCRATCH_INTS(N) (5 * (size_t)(N) + 64)
// The region merge is launched until a launch changes nothing - at most RD_REGION_MAX_LAUNCHES times (the definition's limit: DESIGN.md, "Region stages").  Its
// flag words (one per launch) are followed by the status words of the absorption at RD_REGION_STATUS_AT.  (Round 6: 128 - one 3840x2160 frame of the held-out stream needs 86,
// two 1920x1080 frames 49; the limit had been 64.)
#define RD_REGION_MAX_LAUNCHES 128
#define RD_REGION_STATUS_AT 128
// absorption of small regions (oclrect.cl:348-371) exactly as the reference's serial raster order gives it.  out != in; scratch: RD_D2_SCRATCH_INTS(N) ints
// (scratch[N] = 0 already if count_is_zero); status: 3 device ints, [0] != 0 <=> not finished (out holds negative words): run despeckle2_slow()
void despeckle2(hipStream_t s, int *out, const int *in, int *scratch, const int *size, int thre, int iw, int ih, int count_is_zero, int *status, int nz = 1, size_t zs = 0);
void despeckle2_slow(hipStream_t s, int *out, const int *in, int *scratch, const int *size, int thre, int iw, int ih);   // the same result for any input; synchronises s (not for captured streams)
struct PolyScratch;
struct PolyFrame;      // rd_poly_scratch.h: per-frame descriptor of the sparse stages; launches cover nb <= RD_MAXB frames (frame = blockIdx.z), descriptors in HOST memory
void reduce_ls_init(hipStream_t s, int *table, int *claim, int *tlist, int nentry);   // once per allocation
// votes of the chain pixels left in each frame's scratch by the last polyline() call (their final segment ids) into frames[z].table
void reduce_ls(hipStream_t s, const PolyFrame *frames, int nb, int iw, int ih, int nentry, int tables_are_clean = 0);   // tables_are_clean: label8_boundary(vt_*) has undone the previous use already
// per segment, 15 probe points: {boundary id, table slot owner, 4 box values} -> out[(seg*15 + k)*6 ..]
// pack (may be null): the block for the host in one piece - 64 ints of counters / flags, pack_records records of 14 ints, then their probes
void sample_segments(hipStream_t s, const PolyFrame *frames, int nb, int max_records, int iw, int ih, int nentry, int pack_records);

// ---- rd_k_post.hip: segments + probes -> rectangles on the device (candidate funnel + pose estimation, one wave per candidate)
size_t post_scratch_ints();      // ints of PolyFrame::post_scratch
size_t post_out_ints();          // ints of PolyFrame::post_out
void post_device(hipStream_t s, const PolyFrame *frames, int nb, int max_records, int iw, int ih, double tanAOV);

// ---- rd_k_poly.hip: polyline stage on compacted chain pixels
PolyScratch *poly_scratch_create(int iw, int ih);
void poly_scratch_destroy(PolyScratch *ps);
const int *poly_scratch_counters(const PolyScratch *ps);   // device pointer: [0] chain pixels, [1] chains, [2+r] split candidates of round r
// frames: nb descriptors (host memory); frames[z].ring_src: plane whose 2-px frame ring supplies the stale ring values (null -> ring_const)
void polyline(hipStream_t s, const PolyFrame *frames, int nb, int lslist_bytes, int ring_const, float minerror, int sizeThre, int iw, int ih, int mode);
// materialises the dense id planes frames[z].ids from the compact state
void polyline_ids(hipStream_t s, const PolyFrame *frames, int nb, int n);

}  // namespace rdk
