#!/usr/bin/env python3
"""Is hipMemcpyAsync / hipMemcpy2DAsync from PAGEABLE host memory finished with the host buffer when the call returns?
Overwrites the source right after the call and checks what arrived on the device."""
import ctypes as C
import numpy as np

hip = C.CDLL("libamdhip64.so")
st = C.c_void_p()
assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0
for name, nbytes in (("2 MB", 2 << 20), ("466 KB", 1242 * 375), ("64 KB", 64 << 10)):
    for mode in ("1D", "2D"):
        bad = 0
        for trial in range(20):
            src = np.full(nbytes, 7, dtype=np.uint8)
            d = C.c_void_p()
            assert hip.hipMalloc(C.byref(d), C.c_size_t(nbytes)) == 0
            if mode == "1D":
                rc = hip.hipMemcpyAsync(d, src.ctypes.data_as(C.c_void_p), C.c_size_t(nbytes), 1, st)
            else:
                w = 1242 if nbytes == 1242 * 375 else 1024
                rc = hip.hipMemcpy2DAsync(d, C.c_size_t(w), src.ctypes.data_as(C.c_void_p), C.c_size_t(w), C.c_size_t(w), C.c_size_t(nbytes // w), 1, st)
            assert rc == 0
            src[:] = 9          # the caller reuses its buffer immediately
            hip.hipStreamSynchronize(st)
            out = np.zeros(nbytes, dtype=np.uint8)
            hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), d, C.c_size_t(nbytes), 2)
            bad += int((out != 7).any())
            hip.hipFree(d)
        print("%-7s %s: %d of 20 copies saw the overwritten source" % (name, mode, bad))
