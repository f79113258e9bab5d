"""examples/wave_generator.hpp: the compiled host class in the shape of the reference's WaveGenerator (wave_generator.gd:8-17,56,90,
water.gd:93-100) -- what a GDExtension node would wrap.  CPU: it builds with -Wall -Wextra -Werror as C++17 against the C header
and fails loudly without a device.  GPU: examples/wave_generator_host.cpp (update / _process once per frame / layers delivered to a
texture sink one frame later) produces exactly what the Python mirror produces on the same schedule."""
import os
import subprocess

import pytest

from test_c_consumer import PKG, ROOT, check_against_mirror


def build(tmp_path):
    exe = str(tmp_path / "wave_generator_host")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "wave_generator_host.cpp"), "-o", exe, "-L", PKG, "-locean_waves",
                    f"-Wl,-rpath,{PKG}", "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    return exe


def test_builds_and_fails_loudly_without_a_device(tmp_path):
    import torch
    exe = build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr and "no CPU fallback" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("n,frames", [(256, 12), (512, 7)])
def test_cpp_host_and_python_mirror_agree(tmp_path, n, frames):
    r = subprocess.run([build(tmp_path), str(n), str(frames)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    check_against_mirror(r.stdout, n, frames)
