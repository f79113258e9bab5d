"""Parameter sets at the edges of the reference's exported ranges (wave_cascade_parameters.gd:7-35) and of the
seed / tile domain: used by the oracle-vs-reference-shaders test (CPU, bit-exact) and the HIP parity test (GPU)."""
import math


def edge_presets():
    base = dict(tile_length=(50.0, 50.0), displacement_scale=1.0, normal_scale=1.0, wind_speed=20.0, wind_direction=0.0,
                fetch_length=550.0, swell=0.8, spread=0.2, detail=1.0, whitecap=0.5, foam_amount=5.0,
                spectrum_seed=(0, 0), time=120.0)
    out = {}
    out["defaults_zero_seed"] = dict(base)                                                          # wave_cascade_parameters.gd defaults, Vector2i.ZERO seed
    out["non_square_tile"] = dict(base, tile_length=(88.0, 33.0), wind_direction=135.0)            # tile_length is a Vector2
    out["calm_min_wind_short_fetch"] = dict(base, wind_speed=0.0001, fetch_length=0.0001)          # both clamped minima (:15,:20)
    out["gale_long_fetch"] = dict(base, wind_speed=60.0, fetch_length=5000.0, tile_length=(500.0, 500.0))
    out["swell_spread_detail_extremes"] = dict(base, swell=2.0, spread=1.0, detail=0.0)            # [0,2], [0,1], [0,1] upper/lower ends
    out["no_swell_no_spread"] = dict(base, swell=0.0, spread=0.0, detail=0.5, wind_direction=-270.0)
    out["whitecap_foam_extremes"] = dict(base, whitecap=2.0, foam_amount=10.0, tile_length=(16.0, 16.0))   # everything foams, slowest decay floor
    out["no_foam"] = dict(base, whitecap=0.0, foam_amount=0.0)
    out["wrapping_seed"] = dict(base, spectrum_seed=(-10000, 10000), time=0.0)                     # randi_range bounds (water.gd:31); id + seed wraps as uvec2
    out["late_time"] = dict(base, time=120.0 + math.pi * 7 + 1000 * 0.02, tile_length=(9.0, 9.0))  # cascade 7 after the 1000-frame loop: largest phases
    return out
