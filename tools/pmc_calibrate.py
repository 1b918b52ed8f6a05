#!/usr/bin/env python3
"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for this code base's dominant access pattern (4 bytes per lane,
coalesced, grid-stride): the library's copy operator moves a buffer far larger than L2 + Infinity Cache, so every byte
has to cross the memory-side counters.  Run under `rocprofv3 --pmc FETCH_SIZE` (and WRITE_SIZE in a second run) and
compare the reported value of k_copy_i with the printed byte count."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rectdetect_amd as ra

NBYTES = 1 << 30
ctx = ra.Context(0)
L = ra.lib()
iu = L.init_oclimgutil(ctx.device, ctx.context)
a, b = ctx.buffer(NBYTES), ctx.buffer(NBYTES)
L.oclimgutil_clear(iu, a, NBYTES, ctx.queue, None)
for _ in range(3):
    L.oclimgutil_copy(iu, b, a, NBYTES, ctx.queue, None)
L.clFinish(ctx.queue)
print("k_copy_i: 3 calls, each reads %d bytes and writes %d bytes (= %d KiB)" % (NBYTES, NBYTES, NBYTES // 1024))
ctx.release(a, b)
L.dispose_oclimgutil(iu)
ctx.close()
