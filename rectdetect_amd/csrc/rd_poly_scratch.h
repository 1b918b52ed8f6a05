// rectdetect-mi355x: device scratch of the polyline stage (compact chain-pixel arrays); shared with the voting kernels.
#pragma once

namespace rdk {

struct PolyScratch {
  int cap;            // = iw*ih, capacity of every per-pixel array
  unsigned long long *sb, *rb;     // bit planes (ceil(iw / 64) words per row) of a caller's int planes: the mask, the plane that supplies the frame ring (k_mask_bits)
  unsigned long long *tb;          // bit plane of the chain pixels (k_tidy_bits)
  int *pw, *rowsum, *rowbase;      // chain pixels before each word in its row, per row, in the rows above (k_row_prefix, k_chain_scatter): pixel -> compact index = px_rank()
  unsigned long long *cstate;   // compaction: one state word per block and call site (generation | status | count or running total)
  int *csync;         // compaction: [0] generation, advanced once per frame
  int *pos;           // compact: pixel index, ascending
  int *nbr;           // compact: 8 neighbour compact indices (E,NE,N,NW,W,SW,S,SE), -1 = none
  int *lab, *alive, *ends;
  int *nx[2], *pv[2], *flag, *flag2;
  int *num[2], *link[2];
  int *lab2, *size, *rootid, *id, *dist;
  int *cand;          // candidate records of one split round (8 ints each)
  int *ctr;           // counters: [0]=cnt, [1]=nchains, [2..18]=candidates per round, [20]=refine done count ...
  void *lsx;          // per-segment moment sums
  int *segaux;        // per-segment: startPix, endPix
  int *live;          // compact indices of the chain pixels that survived the size filter (ascending); count in ctr[24]
};

// Everything the sparse stages (polylines, votes, probes) of ONE frame work on.  A launch of these stages covers gridDim.z frames:
// the kernels take the descriptors BY VALUE (PolyFrames, in the kernel-argument segment: uniform scalar loads, indexable) and work on
// element blockIdx.z - frames of one stream, batched, so that the 20-odd latency-bound launches of these stages are paid once per
// batch instead of once per frame.
struct PolyFrame {
  PolyScratch ps;
  const int *in;           // the mask whose curves are traced: a dense int plane ...
  const unsigned long long *in_bits;   // ... or (if not null) a bit plane, ceil(iw / 64) words per row, bit b of word wx = pixel wx * 64 + b
  const int *ring_src;     // plane supplying the stale 2-px ring of the bridging step; null: a constant is used
  void *lslist;            // linesegment_t list, record 0 = header
  int *ids;                // dense per-pixel segment ids (only written by polyline_ids)
  // votes and probes (rect path only)
  const int *boundary;     // boundary-component labels
  int *table, *claim, *tlist;
  int *probes;             // 15 probes x 6 ints per segment
  int *pack;               // the block that travels to the host (pinned host memory, device address); may be null
  const int *rflags;       // round flags of the region merge (travel with the block)
  // rectangles on the device (rd_k_post.hip; optional): scratch in device memory, result block in pinned host memory (device address)
  int *post_scratch;
  int *post_out;           // [0] candidates, [1] overflow, [2..3] tanAOV (double), [8 + c] validity of candidate c, then RD_POST_MAXC rect_t records
};
#define RD_POST_MAXC 1024  // candidates per frame the device post-process holds (more: the host path takes the frame)

#define RD_MAXB 8        // frames per launch of the sparse stages (8 descriptors = 2.8 KB of kernel arguments; the limit is 4 KB)
struct PolyFrames { PolyFrame f[RD_MAXB]; };
inline PolyFrames pack_frames(const PolyFrame *frames, int nb) {
  PolyFrames r;
  for (int z = 0; z < RD_MAXB; z++) r.f[z] = frames[z < nb ? z : 0];
  return r;
}

}  // namespace rdk
