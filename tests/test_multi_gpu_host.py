"""examples/multi_gpu_host.c: the device group of the C-ABI from plain C99 (-pedantic -Werror) -- BASELINE config C4's shape driven by a
single-process host.  CPU: builds and fails loudly without a device.  GPU (one MI355X: two / four shards on device 0, every shard
through the peer path): its checksum of the gathered arrays equals the Python mirror's on the same schedule, whatever the device
list (cascades are independent units)."""
import os
import subprocess

import numpy as np
import pytest

from test_c_consumer import PKG, ROOT


def build(tmp_path):
    exe = str(tmp_path / "multi_gpu_host")
    subprocess.run(["gcc", "-O2", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "multi_gpu_host.c"), "-o", exe, "-L", PKG, "-locean_waves",
                    f"-Wl,-rpath,{PKG}", "-Wl,-rpath-link,/opt/rocm/lib", "-lm"], check=True)
    return exe


def test_builds_as_pedantic_c99_and_fails_loudly_without_a_device(tmp_path):
    import torch
    exe = build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr and "no CPU fallback" in r.stderr


def fnv1a(b, h):
    for x in np.frombuffer(b, np.uint8).tolist():
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.gpu
@pytest.mark.parametrize("devices,per", [("0,0", 1), ("0,0,0,0", 1), ("0,0", 2)])
def test_c_program_and_python_group_agree(tmp_path, devices, per):
    from godotoceanwaves_amd import WaveCascadeParameters, WaveGeneratorGroup
    from godotoceanwaves_amd.presets import UPDATE_DELTA, cascade_preset
    n, ticks, every = 256, 40, 8
    r = subprocess.run([build(tmp_path), str(n), str(per), str(ticks), str(every), devices], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    # round 6: one line per shard first -- how its layers reach the root device (ow_group_link_info: peer access, link type, hops, which path)
    links = [l for l in lines if l.startswith("shard ")]
    assert len(links) == len(devices.split(",")) and all("device 0 -> root device 0: same device, peer_access=1" in l for l in links)
    lines = [l for l in lines if not l.startswith("shard ")]
    assert lines[0].startswith("model: per-device tick") and "153 GB/s per link" in lines[0]   # the model the first real multi-GPU run is read against
    out = dict(kv.split("=") for kv in lines[-1].split())
    shards = len(devices.split(","))
    assert int(out["devices"]) == shards and int(out["cascades"]) == shards * per and int(out["bytes_per_shard"]) == per * n * n * 16
    assert float(out["maps_per_s_no_gather"]) > 0 and float(out[f"maps_per_s_gather_every_{every}"]) > 0 and float(out["gather_copy_ms"]) > 0
    # the same schedule through the Python mirror: 50 warm-up ticks, `ticks` plain, `ticks` in chunks with a gather after each
    grp = WaveGeneratorGroup()
    grp.map_size = n
    grp.init_gpu([0] * shards, per)
    params = [WaveCascadeParameters(**cascade_preset(i)) for i in range(shards * per)]
    grp.run(UPDATE_DELTA, params, 50)
    grp.run(UPDATE_DELTA, params, ticks)
    for _ in range(ticks // every):
        grp.run(UPDATE_DELTA, params, every)
        grp.gather_begin()
    grp.gather_wait()
    h = 1469598103934665603
    for c in range(shards * per):
        d, m = grp.get_maps(c)
        h = fnv1a(m.tobytes(), fnv1a(d.tobytes(), h))
    assert int(out["checksum"], 16) == h
    assert float(out["time"]) == params[-1].time
    grp.free()
