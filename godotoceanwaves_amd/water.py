"""Host-side mirror of the ocean node's UPDATE POLICY (assets/water/water.gd) -- SURVEY.md 8f row N1, the step
immediately above the hot path.  Only the scheduling is mirrored: rate limiter, parameter-array setter (dirty flags,
time offsets, seeds), the map_scales uniform, and the per-frame drive of the generator.  Rendering (mesh, materials,
global shader parameters) is out of scope.

Not reproduced: Godot's RandomNumberGenerator (PCG32; engine source is not part of the reference checkout), which
water.gd:31 uses to draw the seeds.  Seeds are explicit inputs here (default: the frozen table of presets.py).
"""
import math

from .wave_generator import WaveGenerator


class Water:
    def __init__(self, generator_factory=WaveGenerator):
        self._generator_factory = generator_factory   # tests inject a recording fake; the product uses WaveGenerator
        self.wave_generator = None
        self._parameters = []
        self._map_size = 1024                          # water.gd:38
        self._updates_per_second = 50.0                # water.gd:51
        self.time = 0.0                                # water.gd:61
        self.next_update_time = 0.0                    # water.gd:62

    # ---- parameters (water.gd:22-35) ---------------------------------------------------------------------------
    @property
    def parameters(self):
        return self._parameters

    def set_parameters(self, value, seeds=None):
        """`parameters = value` of water.gd:22-35: every cascade gets a seed and the time offset 120 + PI*i
        ("such that cascades don't interfere", :32), then the generator is rebuilt and every spectrum is dirty."""
        for i, p in enumerate(value):
            if seeds is not None:
                p.spectrum_seed = tuple(seeds[i])
            p.time = 120.0 + math.pi * i
        self._parameters = list(value)
        self._setup_wave_generator()

    # ---- map_size (water.gd:38-41) --------------------------------------------------------------------------------
    @property
    def map_size(self):
        return self._map_size

    @map_size.setter
    def map_size(self, value):
        self._map_size = int(value)
        self._setup_wave_generator()

    # ---- updates_per_second (water.gd:51-54): keeps the phase of the next update when the rate changes ------------
    @property
    def updates_per_second(self):
        return self._updates_per_second

    @updates_per_second.setter
    def updates_per_second(self, value):
        self.next_update_time = self.next_update_time - (1.0 / (self._updates_per_second + 1e-10) - 1.0 / (value + 1e-10))
        self._updates_per_second = value

    # ---- _process (water.gd:75-82) + the engine-driven child _process (wave_generator.gd:56-63) --------------------
    def _process(self, delta):
        """One rendered frame: at most one update() (rate limiter), then the generator drains ONE armed cascade.
        Returns the update delta that was issued, or None."""
        issued = None
        if self._updates_per_second == 0 or self.time >= self.next_update_time:
            target_update_delta = 1.0 / (self._updates_per_second + 1e-10)
            update_delta = delta if self._updates_per_second == 0 else target_update_delta + (self.time - self.next_update_time)
            self.next_update_time = self.time + target_update_delta
            self._update_water(update_delta)
            issued = update_delta
        self.time += delta
        if self.wave_generator is not None:            # child node: its _process runs after the parent's
            self.wave_generator._process(delta)
        return issued

    # ---- _setup_wave_generator (water.gd:84-100) ----------------------------------------------------------------------
    def _setup_wave_generator(self):
        if len(self._parameters) <= 0:
            return
        for p in self._parameters:
            p.should_generate_spectrum = True
        if self.wave_generator is not None and hasattr(self.wave_generator, "free"):
            self.wave_generator.free()                 # `wave_generator = value` queue_free()s the old node (:56-59)
        self.wave_generator = self._generator_factory()
        self.wave_generator.map_size = self._map_size
        self.wave_generator.init_gpu(max(2, len(self._parameters)))   # maxi(2, n), :91

    # ---- _update_scales_uniform (water.gd:102-110) -----------------------------------------------------------------------
    def map_scales(self):
        """the `map_scales` uniform of the water / spray materials: (1/tile.x, 1/tile.y, displacement_scale, normal_scale)"""
        return [(1.0 / p.tile_length[0], 1.0 / p.tile_length[1], p.displacement_scale, p.normal_scale) for p in self._parameters]

    # ---- _update_water (water.gd:112-114) ----------------------------------------------------------------------------------
    def _update_water(self, delta):
        if self.wave_generator is None:
            self._setup_wave_generator()
        self.wave_generator.update(delta, self._parameters)
