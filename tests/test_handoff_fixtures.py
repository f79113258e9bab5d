"""Every hand-off path of SURVEY 8f N2 against COMMITTED FIXTURES (tests/golden/ref_n1024_c*_f3.npz = bytes the reference's own shaders
produced, see tests/golden/make_golden.py) instead of against ow_get_maps or the Python mirror: a layer-index or stride bug that is
consistent on both sides of a self-comparison cannot hide here, because every layer is held to a DIFFERENT fixture.
  * ow_readback_begin / ow_readback_wait (page-locked staging, the bytes for RenderingDevice.texture_update);
  * ow_export_maps -> a second process that inherits nothing but the two dma-buf descriptors -> ow_import_buffer;
  * examples/c_consumer.c (C99) and examples/wave_generator_host.cpp (C++17 host class): what their sinks are handed.
Tolerance as in test_golden.py: RGBA16F within one FP16 ulp + 1e-5 of the channel maximum, foam within one FP16 step."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers as H
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, _lib
from godotoceanwaves_amd.presets import UPDATE_DELTA, cascade_preset

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
N, FRAMES = 1024, 3


def fixture(ci):
    z = np.load(os.path.join(GOLDEN, f"ref_n{N}_c{ci}_f{FRAMES}.npz"))
    assert int(z["map_size"]) == N and int(z["cascade"]) == ci and int(z["frames"]) == FRAMES and float(z["delta"]) == UPDATE_DELTA
    return z


def hold_to_fixture(disp_bits, norm_bits, ci, what):
    """disp / norm: [N][N][4] uint16 of ONE layer, against the reference-shader bytes of cascade preset `ci` after three updates"""
    z = fixture(ci)
    stride = int(z["row_stride"])
    d = np.asarray(disp_bits).view(np.uint16).reshape(N, N, 4)[::stride]
    m = np.asarray(norm_bits).view(np.uint16).reshape(N, N, 4)[::stride]
    assert H.fp16_close(d, z["displacement"]) <= 1.0, (what, ci, "displacement")
    assert H.fp16_close(m[..., :3], z["normal"][..., :3]) <= 1.0, (what, ci, "normal")
    foam, foam_ref = m[..., 3].view(np.float16).astype(np.float64), z["normal"][..., 3].view(np.float16).astype(np.float64)
    assert np.abs(foam - foam_ref).max() <= H.TOL_FOAM_ABS, (what, ci, "foam")
    # ... and the layers are really distinguishable: the same bytes fail against another cascade's fixture
    other = fixture((ci + 1) % 3)
    assert H.fp16_close(d, other["displacement"]) > 4.0


def run_three_updates(order):
    """cascade presets `order` in layers 0, 1, 2 (NOT the identity: the layer index must be carried, not assumed), three updates"""
    gen = WaveGenerator()
    gen.map_size = N
    gen.init_gpu(len(order))
    params = [WaveCascadeParameters(**cascade_preset(ci)) for ci in order]
    for _ in range(FRAMES):
        gen.update_all(UPDATE_DELTA, params)
    return gen, params


def test_readback_staging_holds_the_reference_bytes_layer_by_layer():
    order = [1, 2, 0]
    gen, params = run_three_updates(order)
    gen.readback_begin(range(3))
    gen.run(UPDATE_DELTA, params, 2)                       # later ticks overwrite the live maps while the copy is in flight
    for layer in (2, 0, 1):
        d, m = gen.readback_wait(layer)
        hold_to_fixture(d, m, order[layer], "readback")


def test_dma_buf_import_in_a_second_process_sees_the_reference_bytes(tmp_path):
    L = _lib.load()
    order = [2, 0, 1]
    gen, params = run_three_updates(order)
    gen.sync()
    dfd, nfd, nb = C.c_int32(-1), C.c_int32(-1), C.c_size_t()
    _lib.check(L.ow_export_maps(gen.context, C.byref(dfd), C.byref(nfd), C.byref(nb)))
    assert nb.value == 3 * N * N * 8
    try:
        child = ("import sys, ctypes as C, numpy as np\n"
                 f"sys.path.insert(0, {ROOT!r})\n"
                 "from godotoceanwaves_amd import _lib\n"
                 "L = _lib.load()\n"
                 "h = C.CDLL('libamdhip64.so'); h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]\n"
                 "for fd, path in ((int(sys.argv[1]), sys.argv[4]), (int(sys.argv[2]), sys.argv[5])):\n"
                 "    im, ptr = C.c_void_p(), C.c_void_p()\n"
                 "    _lib.check(L.ow_import_buffer(0, fd, 0, int(sys.argv[3]), C.byref(im), C.byref(ptr)))\n"
                 "    out = np.empty(int(sys.argv[3]), np.uint8)\n"
                 "    assert h.hipMemcpy(out.ctypes.data, ptr, out.size, 2) == 0\n"
                 "    out.tofile(path)\n"
                 "    L.ow_release_buffer(im)\n")
        dpath, npath = str(tmp_path / "disp.bin"), str(tmp_path / "norm.bin")
        r = subprocess.run([sys.executable, "-c", child, str(dfd.value), str(nfd.value), str(nb.value), dpath, npath],
                           pass_fds=(dfd.value, nfd.value), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
    finally:
        os.close(dfd.value)
        os.close(nfd.value)
    disp = np.fromfile(dpath, np.uint16).reshape(3, N, N, 4)
    norm = np.fromfile(npath, np.uint16).reshape(3, N, N, 4)
    for layer in range(3):
        hold_to_fixture(disp[layer], norm[layer], order[layer], "dma-buf import")


@pytest.mark.parametrize("host", ["c_consumer", "wave_generator_host"])
def test_compiled_hosts_hand_the_reference_bytes_to_their_sinks(tmp_path, host):
    """the reference's schedule (update + one cascade per rendered frame, wave_generator.gd:56-63,90-109): nine frames = three updates of
    the three cascades of main.tscn (presets 0, 1, 2 in layers 0, 1, 2); what the sink got LAST for each layer is the fixture's state"""
    if host == "c_consumer":
        from test_c_consumer import build
    else:
        from test_cpp_host import build
    prefix = str(tmp_path / "layer")
    r = subprocess.run([build(tmp_path), str(N), str(3 * FRAMES), prefix], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    assert f"layers_handed_off={3 * FRAMES}" in r.stdout
    for layer in range(3):
        raw = np.fromfile(f"{prefix}{layer}.bin", np.uint16)
        assert raw.size == 2 * N * N * 4
        hold_to_fixture(raw[:N * N * 4], raw[N * N * 4:], layer, host)
