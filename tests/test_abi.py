"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol
include/ocean_waves.h declares; error behaviour without a device; no product code touches the oracle."""
import ctypes as C
import os
import re
import subprocess

import pytest

from godotoceanwaves_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build_library()
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ocean_waves.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ow_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 18
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (ow_[a-z0-9_]+)", out))
    assert set(syms) <= exported, sorted(set(syms) - exported)
    assert set(syms) == set(_lib.SIGNATURES), "ctypes table and header disagree"
    # ... and nothing else: the library is built with -fvisibility=hidden, its C++ internals (launchers, kernels' host stubs) stay inside
    other = [l.split()[-1] for l in out.splitlines() if " T " in l and not l.split()[-1].startswith("ow_")]
    assert other == [], other


def test_header_is_plain_c():
    src = '#include "ocean_waves.h"\nint main(void){ow_cascade_params p; ow_cascade_params_default(0); (void)p; return sizeof(ow_config) > 0 ? 0 : 1;}\n'
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                        "-x", "c", "-"], input=src, text=True, capture_output=True)
    assert r.returncode == 0, r.stderr


def test_struct_layout_matches_ctypes(lib):
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "ocean_waves.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %d\\n",'
           'sizeof(ow_cascade_params), offsetof(ow_cascade_params, spectrum_seed), offsetof(ow_cascade_params, time),'
           'sizeof(ow_config), offsetof(ow_config, stream), offsetof(ow_config, flags), offsetof(ow_cascade_params, displacement_scale),'
           'offsetof(ow_cascade_params, wind_speed), offsetof(ow_cascade_params, foam_amount), OW_ABI_VERSION);return 0;}\n')
    exe = "/tmp/ow_layout_check"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-x", "c", "-", "-o", exe], input=src, text=True, check=True)
    got = [int(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    P, Cfg = _lib.ow_cascade_params, _lib.ow_config
    assert got == [C.sizeof(P), P.spectrum_seed.offset, P.time.offset, C.sizeof(Cfg), Cfg.stream.offset, Cfg.flags.offset,
                   P.displacement_scale.offset, P.wind_speed.offset, P.foam_amount.offset, lib.ow_abi_version()]
    # ABI 4: the scalar parameters are FP64 (a GDScript float), the record is 128 bytes; tile_length stays a pair of FP32 (Vector2)
    assert got[0] == 128 and got[-1] == 4 and P.tile_length.size == 8 and P.wind_speed.size == 8 and P.foam_amount.size == 8
    # ... and the oracle's record carries the same types (the checker narrows where the reference does, not earlier)
    from oracle import oracle as O
    for name in ("wind_speed", "wind_direction", "fetch_length", "swell", "spread", "detail", "whitecap", "foam_amount"):
        assert getattr(O.CascadeParams, name).size == 8, name
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "ocean_waves.h"\nint main(void){printf("%zu %zu %zu %zu %zu %d\\n",'
           'sizeof(ow_group_config), offsetof(ow_group_config, device_ids), offsetof(ow_group_config, root),'
           'offsetof(ow_group_config, flags), offsetof(ow_group_config, normal_map), OW_MAX_DEVICES);return 0;}\n')
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-x", "c", "-", "-o", exe], input=src, text=True, check=True)
    got = [int(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    G = _lib.ow_group_config
    assert got == [C.sizeof(G), G.device_ids.offset, G.root.offset, G.flags.offset, G.normal_map.offset, _lib.OW_MAX_DEVICES]


def test_defaults_match_reference(lib):
    p = _lib.ow_cascade_params()
    lib.ow_cascade_params_default(C.byref(p))
    # wave_cascade_parameters.gd:7-38
    assert tuple(p.tile_length) == (50.0, 50.0) and p.wind_speed == 20.0 and p.fetch_length == 550.0
    assert p.swell == 0.8 and p.spread == 0.2 and p.detail == 1.0   # the FP64 literals of the .gd file, not their FP32 roundings
    assert p.displacement_scale == 1.0 and p.normal_scale == 1.0 and p.wind_direction == 0.0
    assert p.whitecap == 0.5 and p.foam_amount == 5.0 and p.should_generate_spectrum == 1 and p.time == 0.0


def test_jonswap_host_math(lib):
    # wave_generator.gd:116-121 evaluated by hand in FP64
    U, F, g = 20.0, 550e3, 9.81
    assert lib.ow_jonswap_alpha(U, F) == pytest.approx(0.076 * (U ** 2 / (F * g)) ** 0.22, rel=1e-15)
    assert lib.ow_jonswap_peak_angular_frequency(U, F) == pytest.approx(22.0 * (g * g / (U * F)) ** (1.0 / 3.0), rel=1e-15)


def test_create_argument_errors(lib):
    ctx = C.c_void_p()
    bad = _lib.ow_config(map_size=300, num_cascades=2, device_id=-1, depth=20.0)
    assert lib.ow_create(C.byref(bad), C.byref(ctx)) == _lib.OW_ERR_INVALID and b"map_size" in lib.ow_last_error()
    bad = _lib.ow_config(map_size=256, num_cascades=9, device_id=-1, depth=20.0)
    assert lib.ow_create(C.byref(bad), C.byref(ctx)) == _lib.OW_ERR_INVALID
    assert lib.ow_create(None, C.byref(ctx)) == _lib.OW_ERR_INVALID
    lib.ow_destroy(None)  # allowed


def test_no_device_is_a_loud_error(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    ctx = C.c_void_p()
    cfg = _lib.ow_config(map_size=256, num_cascades=2, device_id=-1, depth=20.0)
    assert lib.ow_create(C.byref(cfg), C.byref(ctx)) == _lib.OW_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.ow_last_error() and not ctx.value


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "godotoceanwaves_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in text and "ow_oracle" not in text and "import oracle" not in text \
                    and "from oracle" not in text and "tests.emul" not in text, f
