"""Host-side mirror of the reference's generator interface on top of the C-ABI.

`WaveCascadeParameters` mirrors assets/water/wave_cascade_parameters.gd (same field names, defaults,
clamps and dirty-flag setters); `WaveGenerator` mirrors assets/water/wave_generator.gd (map_size,
init_gpu, update, _process, descriptors, JONSWAP statics).  All compute happens in
libocean_waves.so (HIP, gfx950); this file holds no numerics beyond packing the parameter record.
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import (OW_FLAG_ALWAYS_REGENERATE_SPECTRUM, OW_FLAG_LAZY_SCRATCH, OW_FLAG_SINGLE_STREAM, OW_FLAG_DEBUG_F32, OW_FLAG_GROUP_P1_COMPACT, OW_FLAG_GROUP_P1_LP, OW_FLAG_GROUP_P2_PIPE, OW_FLAG_GROUP_P2_PLAIN, OW_FLAG_KERNELS_COMPACT,
                   OW_FLAG_KERNELS_LAYER_PARALLEL, OW_FLAG_KERNELS_STANDARD, OW_FLAG_NO_TICK_GROUPS, OW_FLAG_RUN_AS_CALLS, OW_FLAG_RUN_AS_REFERENCE_SCHEDULE,
                   ow_cascade_params, ow_config)

G = 9.81       # wave_generator.gd:5
DEPTH = 20.0   # wave_generator.gd:6


def _dirty(name, clamp=None):
    """@export var with `set(value): ...; should_generate_spectrum = true` (wave_cascade_parameters.gd:7-35)"""
    attr = "_" + name

    def getter(self):
        return getattr(self, attr)

    def setter(self, value):
        setattr(self, attr, clamp(value) if clamp else value)
        self.should_generate_spectrum = True

    return property(getter, setter)


class WaveCascadeParameters:
    """wave_cascade_parameters.gd:1-56 (the imgui mirror fields :44-56 are UI-only and omitted)."""
    tile_length = _dirty("tile_length", lambda v: (float(v[0]), float(v[1])))                 # :7
    wind_speed = _dirty("wind_speed", lambda v: max(0.0001, float(v)))                        # :15
    wind_direction = _dirty("wind_direction", float)                                          # :17
    fetch_length = _dirty("fetch_length", lambda v: max(0.0001, float(v)))                    # :20
    swell = _dirty("swell", float)                                                            # :22
    spread = _dirty("spread", float)                                                          # :25
    detail = _dirty("detail", float)                                                          # :28
    whitecap = _dirty("whitecap", float)      # yes, the reference re-generates on these too    :32-35
    foam_amount = _dirty("foam_amount", float)

    def __init__(self, **kw):
        self.displacement_scale = 1.0   # :9  (consumer-side only; no dirty flag)
        self.normal_scale = 1.0         # :11
        self.tile_length = (50.0, 50.0)
        self.wind_speed = 20.0
        self.wind_direction = 0.0
        self.fetch_length = 550.0
        self.swell = 0.8
        self.spread = 0.2
        self.detail = 1.0
        self.whitecap = 0.5
        self.foam_amount = 5.0
        self.spectrum_seed = (0, 0)            # :37 Vector2i.ZERO
        self.should_generate_spectrum = True   # :38
        self.time = 0.0                        # :40
        self.foam_grow_rate = 0.0              # :41
        self.foam_decay_rate = 0.0             # :42
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(f"WaveCascadeParameters has no field {k!r}")
            setattr(self, k, v)

    def _pack(self, c):
        c.tile_length[0], c.tile_length[1] = self.tile_length
        c.displacement_scale, c.normal_scale = self.displacement_scale, self.normal_scale
        c.wind_speed, c.wind_direction, c.fetch_length = self.wind_speed, self.wind_direction, self.fetch_length
        c.swell, c.spread, c.detail = self.swell, self.spread, self.detail
        c.whitecap, c.foam_amount = self.whitecap, self.foam_amount
        c.spectrum_seed[0], c.spectrum_seed[1] = int(self.spectrum_seed[0]), int(self.spectrum_seed[1])
        c.should_generate_spectrum = 1 if self.should_generate_spectrum else 0
        c.time, c.foam_grow_rate, c.foam_decay_rate = self.time, self.foam_grow_rate, self.foam_decay_rate


class _Descriptor:
    """stand-in for RenderingContext.Descriptor (render_context.gd:23-28): `.rid` is the device pointer"""

    def __init__(self, rid, layer_stride):
        self.rid, self.layer_stride = rid, layer_stride


class WaveGenerator:
    """assets/water/wave_generator.gd.  Typical use, as in water.gd:89-91,112-114:

        gen = WaveGenerator(); gen.map_size = 1024; gen.init_gpu(max(2, len(parameters)))
        gen.update(delta, parameters)          # once per simulation tick
        gen._process(frame_delta)              # once per rendered frame (one cascade each)
    """

    def __init__(self):
        self.map_size = 0          # :8
        self.context = None        # :9  (ow_context*)
        self.descriptors = {}      # :11
        self.pass_parameters = []  # :14
        self._lib = None
        self.depth = DEPTH
        self.debug_f32 = False
        self.kernels = None           # None = runtime picks per batch; "standard" / "layer_parallel" pin the kernel family
        self.tick_groups = True        # False: run() keeps one pair of launches per tick (OW_FLAG_NO_TICK_GROUPS)
        self.run_as_calls = False      # True: run() issues its ticks as update_all() calls, one per tick (OW_FLAG_RUN_AS_CALLS)
        self.run_as_reference = False  # True: run() issues update() + one _process() per cascade, tick by tick (OW_FLAG_RUN_AS_REFERENCE_SCHEDULE)
        self.always_regenerate_spectrum = False  # True: every dirty flag launches the spectrum kernel, as the reference does (OW_FLAG_ALWAYS_REGENERATE_SPECTRUM)
        self.lazy_scratch = False      # True: ow_create allocates one batch of scratch, the look-ahead's share on first use (OW_FLAG_LAZY_SCRATCH)
        self.single_stream = False     # True: tick-pair launches of four 1024^2 cascades stay whole, on the one stream (OW_FLAG_SINGLE_STREAM; default: two chains on two streams)
        self.group_forms = (None, None)  # tests: pin the tick groups' work-item forms -- ("lp" | "compact", "plain" | "pipe"); None = the runtime's choice
        self.device_id = -1
        self.stream = None
        self.external_maps = (None, None)  # optional caller-owned device buffers (displacement, normal)
        self.num_cascades = 0

    @property
    def pass_num_cascades_remaining(self):  # :15
        return self._lib.ow_cascades_remaining(self.context) if self.context else 0

    # ---- init_gpu (:17-54) ------------------------------------------------------------------------
    def init_gpu(self, num_cascades):
        self._lib = _lib.load()
        if self.context:
            self.free()
        cfg = ow_config(map_size=int(self.map_size), num_cascades=int(num_cascades), device_id=self.device_id,
                        depth=float(self.depth), stream=self.stream, displacement_map=self.external_maps[0],
                        normal_map=self.external_maps[1], flags=(OW_FLAG_DEBUG_F32 if self.debug_f32 else 0) | {None: 0, "lp": OW_FLAG_GROUP_P1_LP, "compact": OW_FLAG_GROUP_P1_COMPACT}[self.group_forms[0]] |
                        {None: 0, "plain": OW_FLAG_GROUP_P2_PLAIN, "pipe": OW_FLAG_GROUP_P2_PIPE}[self.group_forms[1]] | (0 if self.tick_groups else OW_FLAG_NO_TICK_GROUPS) | (OW_FLAG_RUN_AS_CALLS if self.run_as_calls else 0) | (OW_FLAG_RUN_AS_REFERENCE_SCHEDULE if self.run_as_reference else 0) |
                        (OW_FLAG_ALWAYS_REGENERATE_SPECTRUM if self.always_regenerate_spectrum else 0) | (OW_FLAG_LAZY_SCRATCH if self.lazy_scratch else 0) | (OW_FLAG_SINGLE_STREAM if self.single_stream else 0) |
                        {None: 0, "standard": OW_FLAG_KERNELS_STANDARD, "layer_parallel": OW_FLAG_KERNELS_LAYER_PARALLEL,
                               "compact": OW_FLAG_KERNELS_COMPACT,
                               "layer_parallel_compact": OW_FLAG_KERNELS_LAYER_PARALLEL | OW_FLAG_KERNELS_COMPACT}[self.kernels])
        ctx = C.c_void_p()
        _lib.check(self._lib.ow_create(C.byref(cfg), C.byref(ctx)))
        self.context = ctx
        self.num_cascades = int(num_cascades)
        d, n, stride = C.c_void_p(), C.c_void_p(), C.c_size_t()
        _lib.check(self._lib.ow_get_device_ptrs(ctx, C.byref(d), C.byref(n), C.byref(stride)))
        self.descriptors = {"displacement_map": _Descriptor(d.value, stride.value),
                            "normal_map": _Descriptor(n.value, stride.value)}

    # ---- _process (:56-63) --------------------------------------------------------------------------
    def _process(self, delta=0.0):
        if not self.context or self.pass_num_cascades_remaining == 0:
            return
        idx = self.pass_num_cascades_remaining - 1
        self._push_live(idx)                                      # parameter objects are live in the reference
        _lib.check(self._lib.ow_process(self.context))
        self.pass_parameters[idx].should_generate_spectrum = False  # :72

    def _push_live(self, idx):
        """the context holds a COPY of the armed records (a C caller's memory is borrowed during a call only); the reference
        reads the live object when it processes a cascade, so the object's current fields are pushed right before"""
        c = ow_cascade_params()
        self.pass_parameters[idx]._pack(c)
        _lib.check(self._lib.ow_set_cascade_params(self.context, idx, C.byref(c)))

    # ---- update (:90-109) ------------------------------------------------------------------------------
    def _arm(self, fn, delta, parameters, drains_all):
        assert len(parameters) != 0                               # :91
        if not self.context:
            self.init_gpu(max(2, len(parameters)))                # :92-93
        leftovers = self.pass_parameters[:self.pass_num_cascades_remaining]
        for i in range(len(leftovers)):                           # the flush (:94-98) sees the live parameter objects
            self._push_live(i)
        new_c = (ow_cascade_params * len(parameters))()
        for p, c in zip(parameters, new_c):
            p._pack(c)
            if any(p is q for q in leftovers):                    # flushed by this very call: its spectrum is regenerated there (:72)
                c.should_generate_spectrum = 0
        _lib.check(fn(self.context, float(delta), new_c, len(parameters)))
        for q in leftovers:
            q.should_generate_spectrum = False
        for p, c in zip(parameters, new_c):                       # time and the foam rates advance inside the objects (:103-106)
            p.time, p.foam_grow_rate, p.foam_decay_rate = c.time, c.foam_grow_rate, c.foam_decay_rate
            if drains_all:
                p.should_generate_spectrum = False
        self.pass_parameters = list(parameters)

    def update(self, delta, parameters):
        self._arm(self._lib.ow_update if self._lib else _lib.load().ow_update, delta, parameters, False)

    def update_all(self, delta, parameters):
        """throughput mode (not in the reference): update() + every cascade in one pair of launches"""
        self._arm(self._lib.ow_update_all if self._lib else _lib.load().ow_update_all, delta, parameters, True)

    def run(self, delta, parameters, frames):
        """throughput mode: `frames` update_all() ticks enqueued back to back by the C runtime"""
        fn = self._lib.ow_run if self._lib else _lib.load().ow_run
        self._arm(lambda ctx, d, arr, cnt: fn(ctx, d, arr, cnt, int(frames)), delta, parameters, True)

    # ---- teardown (:111-113) -------------------------------------------------------------------------------
    def free(self):
        if self.context:
            self._lib.ow_destroy(self.context)
            self.context = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    # ---- statics (:116-121) -----------------------------------------------------------------------------------
    @staticmethod
    def JONSWAP_alpha(wind_speed=20.0, fetch_length=550e3):
        return _lib.load().ow_jonswap_alpha(wind_speed, fetch_length)

    @staticmethod
    def JONSWAP_peak_angular_frequency(wind_speed=20.0, fetch_length=550e3):
        return _lib.load().ow_jonswap_peak_angular_frequency(wind_speed, fetch_length)

    # ---- host read-back helpers (RenderingDevice.texture_get_data equivalents) ------------------------------
    def sync(self):
        _lib.check(self._lib.ow_sync(self.context))

    def debug_inject_fault(self, bits):
        """test hook (ow_debug_inject_fault): fault bits for the next batch only"""
        _lib.check(self._lib.ow_debug_inject_fault(self.context, int(bits)))

    def get_maps(self, cascade):
        n = self.map_size
        disp, norm = np.empty((n, n, 4), np.float16), np.empty((n, n, 4), np.float16)
        _lib.check(self._lib.ow_get_maps(self.context, cascade, disp.ctypes.data, norm.ctypes.data))
        return disp, norm

    def set_normal_map(self, cascade, normal):
        a = np.ascontiguousarray(normal, np.float16)
        assert a.shape == (self.map_size, self.map_size, 4)
        _lib.check(self._lib.ow_set_normal_map(self.context, cascade, a.ctypes.data))

    # ---- hand-off to a host-side consumer: the bytes for RenderingDevice.texture_update (water.gd:95-100) --------
    def readback_begin(self, cascades):
        """Start the asynchronous copy of the given layers (iterable of indices) into the context's page-locked
        staging memory; returns at once, later updates overlap the PCIe transfer."""
        mask = 0
        for i in cascades:
            mask |= 1 << int(i)
        _lib.check(self._lib.ow_readback_begin(self.context, mask))

    def readback_wait(self, cascade):
        """(displacement, normal) of one layer as float16 [N][N][4] VIEWS of the staging memory: valid until the next
        readback_begin of that layer (or free())."""
        d, m = C.c_void_p(), C.c_void_p()
        _lib.check(self._lib.ow_readback_wait(self.context, cascade, C.byref(d), C.byref(m)))
        n = self.map_size
        view = lambda p: np.frombuffer((C.c_uint16 * (n * n * 4)).from_address(p.value), np.float16).reshape(n, n, 4)
        return view(d), view(m)

    # ---- consumer-side sampling on the device (water.gdshader:27-39,72-82; sea_spray_particle.gdshader:78-96) ----
    SURFACE_SAMPLE = np.dtype([("displacement", np.float32, 3), ("gradient", np.float32, 2), ("gradient_scaled", np.float32, 2),
                               ("foam", np.float32), ("normal_factor", np.float32), ("foam_factor", np.float32),
                               ("scale_factor", np.float32), ("spray_active", np.int32), ("gradient_fragment", np.float32, 2),
                               ("foam_fragment", np.float32), ("reserved", np.float32)])

    def sample_surface(self, world_xz, map_scales):
        """Evaluate the water vertex/fragment sums and the sea-spray spawn mask at world points [P][2] (x, z);
        map_scales [C][4] as built by Water.map_scales() (water.gd:105-109).  Returns a structured array."""
        xz = np.ascontiguousarray(world_xz, np.float32).reshape(-1, 2)
        sc = np.ascontiguousarray(map_scales, np.float32).reshape(-1, 4)
        out = np.zeros(len(xz), self.SURFACE_SAMPLE)
        _lib.check(self._lib.ow_sample_surface(self.context, xz.ctypes.data, len(xz), sc.ctypes.data, len(sc), out.ctypes.data))
        return out

    def get_push_constants(self, cascade):
        """(spectrum[16], modulate[8], unpack[4]) uint32 words: the reference's push-constant blocks of this cascade's most recent launch"""
        pc = _lib.ow_push_constants()
        _lib.check(self._lib.ow_get_push_constants(self.context, cascade, C.byref(pc)))
        return (np.array(pc.spectrum, np.uint32), np.array(pc.modulate, np.uint32), np.array(pc.unpack, np.uint32))

    def get_maps_f32(self, cascade):
        out = np.empty((self.map_size, self.map_size, 8), np.float32)
        _lib.check(self._lib.ow_get_maps_f32(self.context, cascade, out.ctypes.data))
        return out

    def get_spectrum(self, cascade):
        n = self.map_size
        h0, om = np.empty((n, n, 4), np.float32), np.empty((n, n), np.float32)
        _lib.check(self._lib.ow_get_spectrum(self.context, cascade, h0.ctypes.data, om.ctypes.data))
        return h0, om

    def get_intermediate(self, cascade):
        out = np.empty((4, self.map_size, self.map_size, 2), np.float32)
        _lib.check(self._lib.ow_get_intermediate(self.context, cascade, out.ctypes.data))
        return out

    KERNEL_FAMILIES = {0: None, 1: "standard", 2: "layer_parallel", 3: "compact", 4: "layer_parallel_compact", 5: "tick_groups_compact",
                       6: "tick_pairs_compact"}

    def last_kernel_family(self):
        """which kernels the most recent batch ran with: "standard", "layer_parallel", "compact" (None before the first launch)"""
        return self.KERNEL_FAMILIES[int(self._lib.ow_last_kernel_family(self.context))]

    def last_batch_cascades(self):
        """cascades in the most recent pair of launches (a tick may be split into several pairs)"""
        return int(self._lib.ow_last_batch_cascades(self.context))

    def tick_group_depth(self):
        """ticks per launch: of the most recent run() that went out in tick groups / tick pairs, else the depth planned for this
        context's small batches (0: no tick groups)"""
        return int(self._lib.ow_tick_group_depth(self.context))

    def lookahead_stats(self):
        """(hits, speculated): update_all() ticks whose pass 1 had been speculated by the previous call, and speculations launched"""
        h, sp = C.c_uint64(), C.c_uint64()
        _lib.check(self._lib.ow_lookahead_stats(self.context, C.byref(h), C.byref(sp)))
        return h.value, sp.value

    def chain_stats(self):
        """launches that went out as two chains on two streams (1024^2, four cascades a side; OW_FLAG_SINGLE_STREAM keeps them whole)"""
        n = C.c_uint64()
        _lib.check(self._lib.ow_chain_stats(self.context, C.byref(n)))
        return n.value

    def spectrum_stats(self):
        """(generated, skipped): spectrum kernels launched, and dirty flags consumed because the resident spectrum had been generated from the
        same thirteen packed constants (a whitecap / foam_amount edit: wave_cascade_parameters.gd:32-35 raise the flag, spectrum_compute never reads them)"""
        g, sk = C.c_uint64(), C.c_uint64()
        _lib.check(self._lib.ow_spectrum_stats(self.context, C.byref(g), C.byref(sk)))
        return g.value, sk.value

    def timing(self, enable):
        """False / 0: off; True / 1: per pass (run() stays on one launch per pass); 2: as launched (tick groups / pairs stay on and are
        timed per launch: timing_read_launches)"""
        _lib.check(self._lib.ow_timing_enable(self.context, 2 if enable == 2 else (1 if enable else 0)))

    def probe_kernel_times(self, reps=50):
        """(pass1_ms, pass2_ms, cascades_per_launch): each kernel alone, `reps` back-to-back launches (benchmark probe;
        the extra pass-2 launches advance the foam state)"""
        a, b, n = C.c_float(), C.c_float(), C.c_int32()
        _lib.check(self._lib.ow_probe_kernel_times(self.context, reps, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def timing_read_launches(self, reset=True):
        """(average ms, count) of the tick-group / tick-pair launches timed under timing(2)"""
        a, n = C.c_float(), C.c_int32()
        _lib.check(self._lib.ow_timing_read_launches(self.context, C.byref(a), C.byref(n), 1 if reset else 0))
        return a.value, n.value

    def timing_read(self, reset=True):
        a, b, n = C.c_float(), C.c_float(), C.c_int32()
        _lib.check(self._lib.ow_timing_read(self.context, C.byref(a), C.byref(b), C.byref(n), 1 if reset else 0))
        return a.value, b.value, n.value
