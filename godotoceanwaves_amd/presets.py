"""Frozen synthetic inputs for parity tests and bench.py.

Cascades 0-2 are the demo scene's parameter sets (reference main.tscn:43-83); 3-7 are the
defaults of wave_cascade_parameters.gd:7-35 with varied tile lengths.  Seeds are explicit
(inside the reference's randi_range(-10000, 10000), water.gd:31); time offsets follow
water.gd:32 (120 + PI*i) and the update cadence water.gd:51 (50 updates/s).
"""
import math

DEPTH = 20.0            # wave_generator.gd:6
UPDATE_DELTA = 1.0 / 50.0  # water.gd:51

# tile, wind_speed, wind_direction(deg), fetch_length(km), swell, spread, detail, whitecap, foam_amount
_TABLE = [
    (88.0, 10.0, 20.0, 150.0, 0.8, 0.2, 1.0, 0.5, 8.0),
    (57.0, 5.0, 15.0, 150.0, 0.8, 0.4, 1.0, 0.5, 0.0),
    (16.0, 20.0, 20.0, 550.0, 0.8, 0.4, 1.0, 0.25, 3.0),
    (250.0, 20.0, 0.0, 550.0, 0.8, 0.2, 1.0, 0.5, 5.0),
    (33.0, 20.0, 0.0, 550.0, 0.8, 0.2, 1.0, 0.5, 5.0),
    (137.0, 20.0, 0.0, 550.0, 0.8, 0.2, 1.0, 0.5, 5.0),
    (23.0, 20.0, 0.0, 550.0, 0.8, 0.2, 1.0, 0.5, 5.0),
    (9.0, 20.0, 0.0, 550.0, 0.8, 0.2, 1.0, 0.5, 5.0),
]


def cascade_preset(i):
    """dict of WaveCascadeParameters fields for synthetic cascade i (any i >= 0; table repeats with new seeds)."""
    t = _TABLE[i % len(_TABLE)]
    return dict(tile_length=(t[0], t[0]), displacement_scale=1.0, normal_scale=1.0, wind_speed=t[1],
                wind_direction=t[2], fetch_length=t[3], swell=t[4], spread=t[5], detail=t[6], whitecap=t[7],
                foam_amount=t[8], spectrum_seed=(1000 + 17 * i, -2000 + 31 * i), time=120.0 + math.pi * i)
