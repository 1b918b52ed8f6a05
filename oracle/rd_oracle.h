/* TEST INFRASTRUCTURE ONLY - see rd_oracle.c.  Interface of the CPU restatement. */
#ifndef RD_ORACLE_H
#define RD_ORACLE_H
#include <stdint.h>

/* same layout as the reference's linesegment_t (oclpolyline.h:74-83) / LS_t (oclpolyline.cl:29-39) */
typedef struct {
  float x0, y0, x1, y1;
  int32_t startIndex, endIndex, leftPtr, rightPtr, startCount, endCount, maxDist, polyid, npix, level;
} rdo_ls_t;

/* optional taps on the polyline stage's intermediate planes (each N ints, may be NULL) */
typedef struct {
  int *connect;     /* after the gap-bridging step (oclpolyline.c:225) */
  int *chain;       /* chain mask after removeBranch + breakLoops (tmp1 at oclpolyline.c:247) */
  int *chain_label; /* chain labels after breakLoops (lsIdOut at oclpolyline.c:247) */
  int *num;         /* distance-from-end numbering (tmp2 at oclpolyline.c:275) */
  int *sub_label;   /* labels after the numbering-continuity split (tmpBig at oclpolyline.c:280) */
  int *ids0;        /* compact ids before subdivision (lsIdOut at oclpolyline.c:295) */
  int converged;    /* out: always 1 here (union-find); kept for symmetry with fixtures */
} rdo_poly_dbg_t;

const uint16_t *rdo_lut(int which, int *n);
void rdo_bgr2plab(uint32_t *out, const uint8_t *bgr, int iw, int ih, int ws);
void rdo_unpack_plab(float *L, float *a, float *b, const uint32_t *in, int n);
void rdo_pack_plab(uint32_t *out, const float *L, const float *a, const float *b, int n);
void rdo_iirblur(float *out, const float *in, int iw, int ih);                 /* r = 2 (sigma 1): what every caller in the reference passes */
int rdo_iirblur_r(float *out, const float *in, int iw, int ih, int r);         /* any radius 0..31; -1 if outside the reference's defined domain */
const float *rdo_iircoef(int r);                                              /* the 15 coefficients of radius r (-1: the sigma = 1 path's own constants) */
void rdo_edgevec(float *vxy, const float *in, int iw, int ih);
void rdo_edge_plab(float *out, const uint32_t *in, int iw, int ih);
void rdo_thinthres(float *out, const float *in, const float *vxy, int iw, int ih);
void rdo_positive_mask(int *out, const float *in, int n);
/* visualisers and operators no application calls (oclimgutil.h:84-98) */
void rdo_convert_bgr_lumaf(uint8_t *out, const float *in, float f, int iw, int ih, int ws);
void rdo_convert_bgr_labeli(uint8_t *out, const int *in, int bgc, int iw, int ih, int ws);
void rdo_plab2bgr(uint8_t *out, const uint32_t *in, int iw, int ih, int ws);
void rdo_edge_f_f(float *out, const float *in, int iw, int ih);
void rdo_edgevec_plab(float *vxy, const uint32_t *in, int iw, int ih);
void rdo_thincubic(float *out, const float *in, const float *vxy, int iw, int ih);
void rdo_label8(int *label, const int *pix, int bgc, int iw, int ih);
void rdo_calc_strength(int *out, const float *edge, const int *label, int iw, int ih);
void rdo_filter_strength(int *label, const int *str, int thre, int iw, int ih);
void rdo_threshold_i(int *out, const int *in, int lo, int thr, int hi, int n);
void rdo_junction(int *out, const int *in, int nonzero_variant, int iw, int ih);
void rdo_connect_rect(int *out, const int *in, int iw, int ih);
void rdo_stringify(int *out, const int *in, int mod2, int iw, int ih);
void rdo_blblur(uint32_t *out, const int8_t *edge, const uint32_t *in, int npairs, int iw, int ih);
void rdo_quantize(uint32_t *out, const uint32_t *in, int n0, int n1, int n2, int n);
void rdo_despeckle(uint32_t *out, const uint32_t *in, const float *edge, int iw, int ih);
void rdo_merge_mask(int *out, const int *junction, int iw, int ih);
void rdo_region_label_init(int *label, const int *pix, int iw, int ih);
void rdo_region_merge_pass(int *label, const int *pix, const int *mask, const int *edge, int iw, int ih);
void rdo_region_size(int *out, const int *label, int n);
void rdo_despeckle2(int *label, const int *size, int thre, int iw, int ih);
void rdo_mark_boundary(int *out, const int *in, int iw, int ih);
/* the reference's region merge with concurrent work-items: what the HIP path reproduces (SPEC, see rd_oracle.c) */
#define RDO_REGION_MAX_LAUNCHES 128      /* (round 6: a 3840x2160 frame of the held-out stream needs 86) */
int rdo_region_concurrent(int *label, const int *pix, const int *mask, const int *edge, int iw, int ih, int launches);
int rdo_despeckle2_jacobi_k(int *label, const int *size, int thre, int iw, int ih, int *nsmall, int max_rounds);
void rdo_reduce_ls(int *table, const int *boundary, const int *lsid, int iw, int ih, int nentry);

/* oclpolyline_execute (oclpolyline.c:218-309).  lslist: lslist_bytes bytes, record 0 = header.
 * ring_nonzero: value model of the 2-px frame ring of the gap-bridging output. */
void rdo_polyline(void *lslist, int lslist_bytes, int *ids, const int *in, int ring_nonzero, float minerror, int sizeThre,
                  int iw, int ih, rdo_poly_dbg_t *dbg);

/* The whole device part of one rect-path frame (oclrect.c:235-381) with every plane kept for inspection. */
typedef struct {
  int iw, ih;
  uint32_t *plab0, *plab1;
  float *Lblur, *vxy, *strength, *nms;
  int *mask0, *tidy, *label1, *str_sum, *prev_strong, *edge500;
  uint32_t *smooth, *quant;
  int *strong, *junction, *mergemask, *region, *rsize, *boundary_src, *boundary, *lsid;
  void *lslist;
  int *table;
  int region_mode;      /* 0: the reference's in-place merge kernel in serial raster order, 8 launches; 1 (SPEC): with concurrent work-items, launched until nothing changes; 2: concurrent, 8 launches; the absorption always in serial raster order */
  int region_rounds, absorb_rounds;   /* launches of the merge kernel the last frame evaluated (absorb_rounds: always 0) */
} rdo_rect_t;

rdo_rect_t *rdo_rect_new(int iw, int ih);
void rdo_rect_free(rdo_rect_t *c);
void *rdo_rect_plane(rdo_rect_t *c, const char *name);
void rdo_rect_frame(rdo_rect_t *c, const uint8_t *bgr, int ws);
void rdo_rect_set_region_mode(rdo_rect_t *c, int mode);
int rdo_rect_info(const rdo_rect_t *c, int which);   /* 0: region_mode, 1: region_rounds, 2: absorb_rounds */

/* oclrect.c:1066-1083: where the host looks for the boundary component next to a segment - 3 points along the segment (its end points rounded to
 * integers first) x 5 offsets of -2..2 pixels along its normal.  out: 15 (x, y) pairs in the reference's loop order (point j outer, offset inner);
 * a probe outside the frame is (-1, -1).  An independent restatement: the product's sampling kernel and its test tap share one helper of their own. */
void rdo_probe_pixels(float x0, float y0, float x1, float y1, int iw, int ih, int *out);

/* poly.cpp:104-123 device part: ids (N ints) and lslist (16N bytes) */
void rdo_poly_frame(void *lslist, int *ids, const uint8_t *bgr, int iw, int ih, int ws, int strengthThre, float minerror, int sizeThre);

#endif
