#!/usr/bin/env python3
"""Sustained HBM write bandwidth reference: hipMemsetAsync of an 800 MB / 8 GB / 80 GB buffer, back to back."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd                                                  # noqa: E402

eng = hsd_amd.Engine(0)
for nbytes in (800_000_000, 8_000_000_000, 80_000_000_000):
    p = eng.malloc(nbytes)
    for _ in range(3):
        eng._call("hsd_memset", C.c_void_p(p), 0, C.c_size_t(nbytes))
    eng.sync()
    reps = 20 if nbytes < 10**10 else 5
    eng.timer_start()
    for _ in range(reps):
        eng._call("hsd_memset", C.c_void_p(p), 0, C.c_size_t(nbytes))
    ms = eng.timer_stop() / reps
    print(f"memset {nbytes/1e6:.0f} MB: {ms:.4f} ms -> {nbytes/ms/1e6:.1f} GB/s")
    eng.free(p)
eng.close()
