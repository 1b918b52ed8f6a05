"""Study (not a test; profiles/NOTES_r04.md): do tiles of the edge-stopped blur reach a fixed point within its ten pairs of passes?  (No: 60 % of the pixels still change in the last pair.)
Runs the CPU oracle's blur pair by pair on a bench-stream frame.  python tests/studies/blur_tile_convergence.py"""
import os
import sys, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import helpers
from rectdetect_amd import synth
iw, ih = 1920, 1080
N = iw*ih
o = helpers.OracleRect(iw, ih, helpers.REGION_SPEC)
for t in range(2):
    o.frame(synth.frame(synth.SEED0, iw, ih, t))
plab0 = o.plane("plab0").view(np.uint32).copy()
edge500 = o.plane("edge500").view(np.int32)
e8 = (edge500 != 0).astype(np.int8) if edge500.dtype != np.int8 else edge500
L = helpers.oracle()
L.rdo_blblur.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
cur = plab0
TW = TH = 64
gx, gy = (iw+TW-1)//TW, (ih+TH-1)//TH
prev_changed = np.ones((gy, gx), bool)
tot_run = 0
for k in range(10):
    out = np.empty_like(cur)
    L.rdo_blblur(out.ctypes.data, e8.ctypes.data, cur.ctypes.data, 1, iw, ih)
    d = (out != cur).reshape(ih, iw)
    ch = np.zeros((gy, gx), bool)
    for ty in range(gy):
        for tx in range(gx):
            ch[ty, tx] = d[ty*TH:(ty+1)*TH, tx*TW:(tx+1)*TW].any()
    # tiles that had to run this pair: any of the 9 around changed in the previous pair
    need = np.zeros_like(prev_changed)
    pc = np.pad(prev_changed, 1)
    for dy in range(3):
        for dx in range(3):
            need |= pc[dy:dy+gy, dx:dx+gx]
    tot_run += need.sum()
    print("pair %d: pixels changed %8d (%.1f %%), tiles changed %4d of %d, tiles that had to run %4d" % (k, d.sum(), 100.0*d.sum()/N, ch.sum(), gx*gy, need.sum()), flush=True)
    prev_changed = ch
    cur = out
print("tile launches", tot_run, "of", 10*gx*gy)
