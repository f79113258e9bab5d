"""Zero-copy hand-off (include/ocean_waves.h "the maps as dma-buf file descriptors"): the reference's consumers sample the two array
textures IN PLACE on the engine's device (wave_generator.gd:19,34-35; README.md:85 names the PCIe copy as what killed the async
experiment).  No Vulkan exists in this image, so the half that is HIP's is exercised end to end between HIP endpoints:
  * export: a context's maps as dma-buf fds -> imported again (same process, and a SECOND PROCESS that only inherits the fds) ->
    the importer reads the very bytes ow_get_maps returns, and sees later ticks without any copy;
  * import: memory allocated elsewhere (standing in for the engine's exported VkDeviceMemory) -> imported -> handed to ow_create
    as the context's output arrays -> the kernels' results appear in the foreign allocation."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, _lib
from godotoceanwaves_amd.presets import UPDATE_DELTA, cascade_preset

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def hip():
    h = C.CDLL("libamdhip64.so")
    h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    h.hipDeviceSynchronize.argtypes = []
    return h


def read_device(ptr, nbytes):
    out = np.empty(nbytes, np.uint8)
    assert hip().hipMemcpy(out.ctypes.data, ptr, nbytes, 2) == 0  # hipMemcpyDeviceToHost
    return out


def make(n, ids, **kw):
    gen = WaveGenerator()
    gen.map_size = n
    for k, v in kw.items():
        setattr(gen, k, v)
    gen.init_gpu(max(2, len(ids)))
    return gen, [WaveCascadeParameters(**cascade_preset(ci)) for ci in ids]


def export(gen):
    L = _lib.load()
    d, m, nb = C.c_int32(-1), C.c_int32(-1), C.c_size_t()
    _lib.check(L.ow_export_maps(gen.context, C.byref(d), C.byref(m), C.byref(nb)))
    return d.value, m.value, nb.value


def test_exported_maps_are_the_live_arrays_in_this_and_in_a_second_process():
    L = _lib.load()
    n, ids = 256, [0, 1, 2]
    gen, params = make(n, ids)
    gen.run(UPDATE_DELTA, params, 3)
    gen.sync()
    dfd, nfd, nbytes = export(gen)
    assert dfd >= 0 and nfd >= 0 and dfd != nfd and nbytes == 3 * n * n * 8
    try:
        handles = []
        for fd, which in ((dfd, 0), (nfd, 1)):
            im, ptr = C.c_void_p(), C.c_void_p()
            _lib.check(L.ow_import_buffer(-1, fd, 0, nbytes, C.byref(im), C.byref(ptr)))
            handles.append((im, ptr, which))
        for tick in range(2):  # the mapping is the live array: later ticks show up without any copy
            for im, ptr, which in handles:
                got = read_device(ptr, nbytes).view(np.uint16).reshape(3, n, n, 4)
                for i in range(3):
                    assert np.array_equal(got[i], gen.get_maps(i)[which].view(np.uint16)), (tick, which, i)
            gen.update_all(UPDATE_DELTA, params)
            gen.sync()
        # a second process that has nothing but the two descriptors
        child = ("import sys, ctypes as C, hashlib, numpy as np\n"
                 f"sys.path.insert(0, {ROOT!r})\n"
                 "from godotoceanwaves_amd import _lib\n"
                 "L = _lib.load()\n"
                 "h = C.CDLL('libamdhip64.so'); h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]\n"
                 "for fd in (int(sys.argv[1]), int(sys.argv[2])):\n"
                 "    im, ptr = C.c_void_p(), C.c_void_p()\n"
                 "    _lib.check(L.ow_import_buffer(0, fd, 0, int(sys.argv[3]), C.byref(im), C.byref(ptr)))\n"
                 "    out = np.empty(int(sys.argv[3]), np.uint8)\n"
                 "    assert h.hipMemcpy(out.ctypes.data, ptr, out.size, 2) == 0\n"
                 "    print(hashlib.sha1(out.tobytes()).hexdigest())\n"
                 "    L.ow_release_buffer(im)\n")
        r = subprocess.run([sys.executable, "-c", child, str(dfd), str(nfd), str(nbytes)], pass_fds=(dfd, nfd), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        import hashlib
        want = [hashlib.sha1(np.stack([gen.get_maps(i)[w] for i in range(3)]).tobytes()).hexdigest() for w in (0, 1)]
        assert r.stdout.split() == want
        for im, _, _ in handles:
            L.ow_release_buffer(im)
    finally:
        os.close(dfd)
        os.close(nfd)
    gen.update_all(UPDATE_DELTA, params)  # the context is none the worse for it
    gen.sync()


def test_kernels_write_into_imported_foreign_memory():
    """the engine's side of the hand-off: a foreign allocation (here: another context's arrays, exported) is imported and becomes
    THIS context's output arrays; what its kernels compute lands in the foreign memory, bit for bit what a self-contained context
    computes"""
    L = _lib.load()
    n, ids = 256, [2, 0]
    owner, _ = make(n, ids)                  # stands in for the engine: it owns the memory, never runs a tick
    dfd, nfd, nbytes = export(owner)
    try:
        im_d, p_d, im_n, p_n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(L.ow_import_buffer(0, dfd, 0, nbytes, C.byref(im_d), C.byref(p_d)))
        _lib.check(L.ow_import_buffer(0, nfd, 0, nbytes, C.byref(im_n), C.byref(p_n)))
        gen, params = make(n, ids, external_maps=(p_d.value, p_n.value))
        ref, rparams = make(n, ids)
        for g, p in ((gen, params), (ref, rparams)):
            g.run(UPDATE_DELTA, p, 4)
            g.sync()
        for i in range(2):
            want = ref.get_maps(i)
            seen_by_owner = owner.get_maps(i)   # the owner's own view of its memory
            for w in (0, 1):
                assert np.array_equal(seen_by_owner[w].view(np.uint16), want[w].view(np.uint16)), (i, w)
        gen.free()
        L.ow_release_buffer(im_d)
        L.ow_release_buffer(im_n)
    finally:
        os.close(dfd)
        os.close(nfd)


def test_argument_errors():
    L = _lib.load()
    im, ptr = C.c_void_p(), C.c_void_p()
    assert L.ow_import_buffer(0, -1, 0, 4096, C.byref(im), C.byref(ptr)) == _lib.OW_ERR_INVALID
    assert L.ow_import_buffer(0, 0, 0, 0, C.byref(im), C.byref(ptr)) == _lib.OW_ERR_INVALID
    assert L.ow_import_buffer(99, 0, 0, 4096, C.byref(im), C.byref(ptr)) == _lib.OW_ERR_INVALID
    r, w = os.pipe()  # a descriptor that is no dma-buf: refused by the runtime, reported, nothing leaked
    try:
        assert L.ow_import_buffer(0, r, 0, 4096, C.byref(im), C.byref(ptr)) == _lib.OW_ERR_HIP and not im
    finally:
        os.close(r)
        os.close(w)
    assert L.ow_export_maps(None, None, None, None) == _lib.OW_ERR_INVALID
    L.ow_release_buffer(None)


def test_offset_import_and_refusal_of_packed_caller_memory():
    """an import may map a window of the exported object (one layer here); caller-owned arrays that are carved out of a larger
    allocation cannot be exported unambiguously (a dma-buf covers the whole buffer object) and are refused"""
    import torch
    L = _lib.load()
    n = 256
    gen, params = make(n, [0, 1, 2])
    gen.run(UPDATE_DELTA, params, 2)
    gen.sync()
    dfd, nfd, nbytes = export(gen)
    try:
        layer = n * n * 8
        im, ptr = C.c_void_p(), C.c_void_p()
        _lib.check(L.ow_import_buffer(0, nfd, 2 * layer, layer, C.byref(im), C.byref(ptr)))
        got = read_device(ptr, layer).view(np.uint16).reshape(n, n, 4)
        assert np.array_equal(got, gen.get_maps(2)[1].view(np.uint16))
        L.ow_release_buffer(im)
    finally:
        os.close(dfd)
        os.close(nfd)
    arena = torch.zeros(6 << 20, dtype=torch.uint8, device="cuda")   # the caller's own arena: the maps sit 1 MiB into it
    base = arena.data_ptr() + (1 << 20)
    packed, _ = make(n, [0, 1], external_maps=(base, base + 2 * n * n * 8))
    d, m, nb = C.c_int32(-1), C.c_int32(-1), C.c_size_t()
    assert L.ow_export_maps(packed.context, C.byref(d), C.byref(m), C.byref(nb)) == _lib.OW_ERR_STATE and b"whole buffer" in L.ow_last_error()
    assert d.value == -1 and m.value == -1
    packed.free()


def test_import_release_cycles_leak_no_descriptors_and_no_memory():
    """every ow_import_buffer duplicates the caller's descriptor for the runtime; after ow_release_buffer nothing of it may be left"""
    import torch
    L = _lib.load()
    gen, params = make(256, [0, 1])
    gen.update_all(UPDATE_DELTA, params)
    gen.sync()
    dfd, nfd, nbytes = export(gen)
    try:
        def cycle():
            im, ptr = C.c_void_p(), C.c_void_p()
            _lib.check(L.ow_import_buffer(0, dfd, 0, nbytes, C.byref(im), C.byref(ptr)))
            first = read_device(ptr, 64)
            L.ow_release_buffer(im)
            return first
        want = cycle()
        fds0, free0 = len(os.listdir("/proc/self/fd")), torch.cuda.mem_get_info()[0]
        for _ in range(40):
            assert np.array_equal(cycle(), want)
        assert len(os.listdir("/proc/self/fd")) <= fds0 + 1, (fds0, len(os.listdir("/proc/self/fd")))
        assert torch.cuda.mem_get_info()[0] >= free0 - (8 << 20)
    finally:
        os.close(dfd)
        os.close(nfd)


def test_contexts_and_groups_come_and_go_without_leaking_device_memory():
    import torch
    from godotoceanwaves_amd import WaveGeneratorGroup
    def once():
        gen, params = make(512, [0, 1, 2, 3])
        gen.run(UPDATE_DELTA, params, 6)           # grows the scratch (tick groups)
        gen.readback_begin([0, 3])
        gen.readback_wait(3)
        gen.free()
        grp = WaveGeneratorGroup()
        grp.map_size = 256
        grp.force_peer_path = True
        grp.init_gpu([0, 0, 0], 2)
        p = [WaveCascadeParameters(**cascade_preset(i)) for i in range(6)]
        grp.run(UPDATE_DELTA, p, 5)
        grp.gather_begin()
        grp.gather_wait()
        grp.free()
    once()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(8):
        once()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free0 - (16 << 20), (free0, torch.cuda.mem_get_info()[0])
