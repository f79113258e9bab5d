"""Host-side mirror of the C-ABI's device group (include/ocean_waves.h "several devices"): the shape of WaveGenerator
(assets/water/wave_generator.gd) over cascades sharded across the GPUs of one node INSIDE one process -- the form a C# / GDExtension
host uses.  (bench.py's multi-GPU runs are the other form: one process per GPU, gather over RCCL, godotoceanwaves_amd/sharding.py.)
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (OW_FLAG_DEBUG_F32, OW_FLAG_NO_TICK_GROUPS, OW_GROUP_FLAG_FORCE_PEER_PATH, OW_MAX_DEVICES, ow_cascade_params,
                   ow_group_config)
from .presets import DEPTH
from .wave_generator import WaveGenerator


class WaveGeneratorGroup:
    """gen = WaveGeneratorGroup(); gen.map_size = 1024; gen.init_gpu(device_ids=[0..7], cascades_per_device=1)
    gen.update(delta, parameters)  /  gen._process()  /  gen.update_all(...)  /  gen.run(..., frames)
    gen.gather_begin(); ...; gen.gather_wait(); gen.get_maps(g)   # layer g of the consumer's arrays = global cascade g"""

    def __init__(self):
        self.map_size = 0
        self.depth = DEPTH
        self.debug_f32 = False
        self.tick_groups = True
        self.force_peer_path = False  # test hook: every shard gathers through snapshot + side stream + hipMemcpyPeerAsync
        self.group = None
        self.pass_parameters = []
        self._lib = None
        self.num_devices = self.cascades_per_device = 0

    def init_gpu(self, device_ids, cascades_per_device, root=0):
        self._lib = _lib.load()
        if self.group:
            self.free()
        if not (1 <= len(device_ids) <= OW_MAX_DEVICES):
            raise ValueError(f"1..{OW_MAX_DEVICES} devices")
        cfg = ow_group_config(map_size=int(self.map_size), num_devices=len(device_ids), cascades_per_device=int(cascades_per_device),
                              root=int(root), depth=float(self.depth),
                              flags=(OW_FLAG_DEBUG_F32 if self.debug_f32 else 0) | (0 if self.tick_groups else OW_FLAG_NO_TICK_GROUPS) |
                                    (OW_GROUP_FLAG_FORCE_PEER_PATH if self.force_peer_path else 0))
        for i, d in enumerate(device_ids):
            cfg.device_ids[i] = int(d)
        g = C.c_void_p()
        _lib.check(self._lib.ow_group_create(C.byref(cfg), C.byref(g)))
        self.group = g
        self.num_devices, self.cascades_per_device = len(device_ids), int(cascades_per_device)

    @property
    def num_cascades(self):
        return int(self._lib.ow_group_num_cascades(self.group)) if self.group else 0

    @property
    def pass_num_cascades_remaining(self):
        return int(self._lib.ow_group_cascades_remaining(self.group)) if self.group else 0

    def shard(self, i):
        """a WaveGenerator VIEW of shard i's context (borrowed: per-shard queries such as get_maps / last_kernel_family)"""
        ctx = self._lib.ow_group_context(self.group, int(i))
        if not ctx:
            raise IndexError(i)
        v = WaveGenerator()
        v.map_size, v._lib, v.context, v.num_cascades = self.map_size, self._lib, C.c_void_p(ctx), self.cascades_per_device
        v.free = lambda: None  # the group owns the context
        return v

    def _call(self, fn, delta, parameters, drains_all):
        arr = (ow_cascade_params * len(parameters))()
        for p, c in zip(parameters, arr):
            p._pack(c)
        _lib.check(fn(self.group, float(delta), arr, len(parameters)))
        for p, c in zip(parameters, arr):  # time and the foam rates advance inside the objects (wave_generator.gd:103-106)
            p.time, p.foam_grow_rate, p.foam_decay_rate = c.time, c.foam_grow_rate, c.foam_decay_rate
            if drains_all:
                p.should_generate_spectrum = False
        self.pass_parameters = list(parameters)

    def update(self, delta, parameters):
        self._call(self._lib.ow_group_update, delta, parameters, False)

    def _process(self, delta=0.0):
        left = self.pass_num_cascades_remaining
        if left == 0:
            return
        # which record goes next: the highest armed global index (shards drain from the top, each from its own top)
        per = self.cascades_per_device
        for s in reversed(range(self.num_devices)):
            r = int(self._lib.ow_cascades_remaining(self._lib.ow_group_context(self.group, s)))
            if r:
                idx = s * per + r - 1
                break
        _lib.check(self._lib.ow_group_process(self.group))
        self.pass_parameters[idx].should_generate_spectrum = False

    def update_all(self, delta, parameters):
        self._call(self._lib.ow_group_update_all, delta, parameters, True)

    def run(self, delta, parameters, frames):
        fn = self._lib.ow_group_run
        self._call(lambda g, d, arr, cnt: fn(g, d, arr, cnt, int(frames)), delta, parameters, True)

    def sync(self):
        _lib.check(self._lib.ow_group_sync(self.group))

    def gather_begin(self):
        _lib.check(self._lib.ow_group_gather_begin(self.group))

    def gather_wait(self):
        _lib.check(self._lib.ow_group_gather_wait(self.group))

    def gather_stats(self):
        """(max copy ms over shards, bytes per shard) of the most recent completed gather"""
        ms, nb = C.c_float(), C.c_size_t()
        _lib.check(self._lib.ow_group_gather_stats(self.group, C.byref(ms), C.byref(nb)))
        return ms.value, nb.value

    def link_info(self):
        """per shard: how its layers reach the root device (peer access, link type / hops as the HIP runtime reports them; ow_group_link_info)"""
        out = []
        for i in range(self.num_devices):
            l = _lib.ow_group_link()
            _lib.check(self._lib.ow_group_link_info(self.group, i, C.byref(l)))
            out.append(l.as_dict())
        return out

    def device_ptrs(self):
        d, n, stride = C.c_void_p(), C.c_void_p(), C.c_size_t()
        _lib.check(self._lib.ow_group_get_device_ptrs(self.group, C.byref(d), C.byref(n), C.byref(stride)))
        return d.value, n.value, stride.value

    def get_maps(self, cascade):
        n = self.map_size
        disp, norm = np.empty((n, n, 4), np.float16), np.empty((n, n, 4), np.float16)
        _lib.check(self._lib.ow_group_get_maps(self.group, int(cascade), disp.ctypes.data, norm.ctypes.data))
        return disp, norm

    def sample_surface(self, world_xz, map_scales):
        xz = np.ascontiguousarray(world_xz, np.float32).reshape(-1, 2)
        sc = np.ascontiguousarray(map_scales, np.float32).reshape(-1, 4)
        out = np.zeros(len(xz), WaveGenerator.SURFACE_SAMPLE)
        _lib.check(self._lib.ow_group_sample_surface(self.group, xz.ctypes.data, len(xz), sc.ctypes.data, len(sc), out.ctypes.data))
        return out

    def free(self):
        if self.group:
            self._lib.ow_group_destroy(self.group)
            self.group = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
