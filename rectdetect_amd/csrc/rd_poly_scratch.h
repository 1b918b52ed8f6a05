// rectdetect-mi355x: device scratch of the polyline stage (compact chain-pixel arrays); shared with the voting kernels.
#pragma once

namespace rdk {

struct PolyScratch {
  int cap;            // = iw*ih, capacity of every per-pixel array
  int *planeA, *planeB, *planeC;   // dense int planes
  int *cidx;          // dense: pixel -> compact index or -1
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

}  // namespace rdk
