"""MI355X-native drop-in for the wave-generation hot path of 2Retr0/GodotOceanWaves.

The product is godotoceanwaves_amd/libocean_waves.so (hand-written HIP for gfx950 behind the C-ABI in
include/ocean_waves.h).  This package is the thin host-side mirror of the reference's
`WaveGenerator` / `WaveCascadeParameters` interface on top of that ABI.
"""
from .wave_generator import WaveCascadeParameters, WaveGenerator  # noqa: F401
from .group import WaveGeneratorGroup  # noqa: F401
from .presets import cascade_preset, DEPTH, UPDATE_DELTA  # noqa: F401
