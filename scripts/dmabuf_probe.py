#!/usr/bin/env python3
"""What does a dma-buf export of a hipMalloc'ed range cover?  Two allocations of each size, filled with different bytes, exported
(hipMemGetHandleForAddressRange) and imported again (through ow_import_buffer): which bytes does each import see at offset 0?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from godotoceanwaves_amd import _lib
L = _lib.load()
h = C.CDLL("libamdhip64.so")
h.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
h.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
h.hipMemGetAddressRange.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p]
h.hipMemGetHandleForAddressRange.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_ulonglong]
for size in (256 << 10, 1 << 20, 3 << 19, 2 << 20, 3 << 20, 16 << 20):
    ptrs = []
    for k in range(3):
        p = C.c_void_p()
        assert h.hipMalloc(C.byref(p), size) == 0
        assert h.hipMemset(p, 0x11 * (k + 1), size) == 0
        ptrs.append(p)
    h.hipDeviceSynchronize()
    row = []
    for k, p in enumerate(ptrs):
        base, rng = C.c_void_p(), C.c_size_t()
        h.hipMemGetAddressRange(C.byref(base), C.byref(rng), p)
        fd = C.c_int(-1)
        e = h.hipMemGetHandleForAddressRange(C.byref(fd), p, size, 1, 0)
        if e != 0:
            row.append(f"[{k}] ptr {p.value:#x} export failed {e}")
            continue
        im, q = C.c_void_p(), C.c_void_p()
        st = L.ow_import_buffer(0, fd.value, 0, size, C.byref(im), C.byref(q))
        if st != 0:
            row.append(f"[{k}] ptr {p.value:#x} base {base.value:#x}+{rng.value:#x} import failed: {L.ow_last_error().decode()}")
            os.close(fd.value)
            continue
        b = (C.c_ubyte * 4)()
        h.hipMemcpy(b, q, 4, 2)
        st_size = os.fstat(fd.value).st_size
        row.append(f"[{k}] ptr {p.value:#x} base {base.value:#x}+{rng.value:#x} fd size {st_size:#x} imported at {q.value:#x} sees {b[0]:#x} (want {0x11 * (k + 1):#x})")
        L.ow_release_buffer(im)
        os.close(fd.value)
    print(f"size {size:#x}:\n  " + "\n  ".join(row), flush=True)
