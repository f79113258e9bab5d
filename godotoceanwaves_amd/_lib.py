"""ctypes binding of include/ocean_waves.h (libocean_waves.so).  Fails loudly: no fallback of any kind."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OCEAN_WAVES_LIB: developer override to A/B a differently built library (scripts/build_variant.sh); same loud failure if missing
LIB_PATH = os.environ.get("OCEAN_WAVES_LIB") or os.path.join(_HERE, "libocean_waves.so")

OW_MAX_CASCADES = 8
OW_MAX_DEVICES = 8
OW_GROUP_FLAG_FORCE_PEER_PATH = 0x10000
OW_FLAG_DEBUG_F32 = 1
OW_FLAG_KERNELS_STANDARD = 2
OW_FLAG_KERNELS_LAYER_PARALLEL = 4
OW_FLAG_KERNELS_COMPACT = 8
OW_FLAG_NO_TICK_GROUPS = 16
OW_FLAG_RUN_AS_CALLS = 32
OW_FLAG_RUN_AS_REFERENCE_SCHEDULE = 64
OW_FLAG_GROUP_P1_LP, OW_FLAG_GROUP_P1_COMPACT, OW_FLAG_GROUP_P2_PLAIN, OW_FLAG_GROUP_P2_PIPE = 0x100, 0x200, 0x400, 0x800
OW_FLAG_ALWAYS_REGENERATE_SPECTRUM = 0x1000
OW_FLAG_LAZY_SCRATCH = 0x2000
OW_FLAG_SINGLE_STREAM = 0x4000
OW_OK, OW_ERR_INVALID, OW_ERR_NO_DEVICE, OW_ERR_HIP, OW_ERR_NOMEM, OW_ERR_STATE = range(6)


class OceanWavesError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"ocean_waves status {status}: {message}")
        self.status = status


class ow_cascade_params(C.Structure):
    """struct ow_cascade_params (WaveCascadeParameters, wave_cascade_parameters.gd:7-42), ABI 4: the scalar parameters are FP64 as a
    GDScript caller holds them (the library narrows where the reference does); tile_length is a Vector2 (FP32 components)"""
    _fields_ = [("tile_length", C.c_float * 2), ("displacement_scale", C.c_double), ("normal_scale", C.c_double),
                ("wind_speed", C.c_double), ("wind_direction", C.c_double), ("fetch_length", C.c_double),
                ("swell", C.c_double), ("spread", C.c_double), ("detail", C.c_double), ("whitecap", C.c_double),
                ("foam_amount", C.c_double), ("spectrum_seed", C.c_int32 * 2),
                ("should_generate_spectrum", C.c_int32), ("reserved", C.c_int32), ("time", C.c_double),
                ("foam_grow_rate", C.c_double), ("foam_decay_rate", C.c_double)]


class ow_push_constants(C.Structure):
    """the reference's three push-constant blocks of one cascade, as 32-bit words (ow_get_push_constants)"""
    _fields_ = [("spectrum", C.c_uint32 * 16), ("modulate", C.c_uint32 * 8), ("unpack", C.c_uint32 * 4)]


class ow_config(C.Structure):
    _fields_ = [("map_size", C.c_int32), ("num_cascades", C.c_int32), ("device_id", C.c_int32), ("depth", C.c_float),
                ("stream", C.c_void_p), ("displacement_map", C.c_void_p), ("normal_map", C.c_void_p),
                ("flags", C.c_uint32)]


class ow_group_config(C.Structure):
    _fields_ = [("map_size", C.c_int32), ("num_devices", C.c_int32), ("device_ids", C.c_int32 * OW_MAX_DEVICES),
                ("cascades_per_device", C.c_int32), ("root", C.c_int32), ("depth", C.c_float), ("flags", C.c_uint32),
                ("displacement_map", C.c_void_p), ("normal_map", C.c_void_p)]


class ow_group_link(C.Structure):
    """how a shard's layers reach the root device (ow_group_link_info / ow_query_link)"""
    _fields_ = [("device", C.c_int32), ("root_device", C.c_int32), ("same_device", C.c_int32), ("peer_access", C.c_int32),
                ("link_type", C.c_int32), ("hops", C.c_int32), ("staged_path", C.c_int32), ("reserved", C.c_int32)]

    LINK_TYPES = {0: "hypertransport", 1: "qpi", 2: "pcie", 3: "infiniband", 4: "xgmi", -1: "unknown"}

    def as_dict(self):
        return {"device": self.device, "root_device": self.root_device, "same_device": bool(self.same_device), "peer_access": bool(self.peer_access),
                "link": self.LINK_TYPES.get(self.link_type, str(self.link_type)), "hops": self.hops, "staged_path": bool(self.staged_path)}


# every symbol include/ocean_waves.h declares: (restype, argtypes)
_P = C.POINTER
SIGNATURES = {
    "ow_create": (C.c_int, [_P(ow_config), _P(C.c_void_p)]),
    "ow_destroy": (None, [C.c_void_p]),
    "ow_cascade_params_default": (None, [_P(ow_cascade_params)]),
    "ow_update": (C.c_int, [C.c_void_p, C.c_double, _P(ow_cascade_params), C.c_int32]),
    "ow_set_cascade_params": (C.c_int, [C.c_void_p, C.c_int32, _P(ow_cascade_params)]),
    "ow_get_cascade_params": (C.c_int, [C.c_void_p, C.c_int32, _P(ow_cascade_params)]),
    "ow_debug_inject_fault": (C.c_int, [C.c_void_p, C.c_uint32]),
    "ow_get_push_constants": (C.c_int, [C.c_void_p, C.c_int32, _P(ow_push_constants)]),
    "ow_process": (C.c_int, [C.c_void_p]),
    "ow_update_all": (C.c_int, [C.c_void_p, C.c_double, _P(ow_cascade_params), C.c_int32]),
    "ow_run": (C.c_int, [C.c_void_p, C.c_double, _P(ow_cascade_params), C.c_int32, C.c_int32]),
    "ow_lookahead_stats": (C.c_int, [C.c_void_p, _P(C.c_uint64), _P(C.c_uint64)]),
    "ow_spectrum_stats": (C.c_int, [C.c_void_p, _P(C.c_uint64), _P(C.c_uint64)]),
    "ow_chain_stats": (C.c_int, [C.c_void_p, _P(C.c_uint64)]),
    "ow_cascades_remaining": (C.c_int32, [C.c_void_p]),
    "ow_last_kernel_family": (C.c_int32, [C.c_void_p]),
    "ow_last_batch_cascades": (C.c_int32, [C.c_void_p]),
    "ow_tick_group_depth": (C.c_int32, [C.c_void_p]),
    "ow_sync": (C.c_int, [C.c_void_p]),
    "ow_get_device_ptrs": (C.c_int, [C.c_void_p, _P(C.c_void_p), _P(C.c_void_p), _P(C.c_size_t)]),
    "ow_get_maps": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "ow_set_normal_map": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "ow_readback_begin": (C.c_int, [C.c_void_p, C.c_uint32]),
    "ow_readback_wait": (C.c_int, [C.c_void_p, C.c_int32, _P(C.c_void_p), _P(C.c_void_p)]),
    "ow_sample_surface": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "ow_get_maps_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "ow_get_spectrum": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "ow_get_intermediate": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "ow_jonswap_alpha": (C.c_double, [C.c_double, C.c_double]),
    "ow_jonswap_peak_angular_frequency": (C.c_double, [C.c_double, C.c_double]),
    "ow_timing_enable": (C.c_int, [C.c_void_p, C.c_int32]),
    "ow_timing_read": (C.c_int, [C.c_void_p, _P(C.c_float), _P(C.c_float), _P(C.c_int32), C.c_int32]),
    "ow_timing_read_launches": (C.c_int, [C.c_void_p, _P(C.c_float), _P(C.c_int32), C.c_int32]),
    "ow_probe_kernel_times": (C.c_int, [C.c_void_p, C.c_int32, _P(C.c_float), _P(C.c_float), _P(C.c_int32)]),
    "ow_group_create": (C.c_int, [_P(ow_group_config), _P(C.c_void_p)]),
    "ow_group_destroy": (None, [C.c_void_p]),
    "ow_group_num_cascades": (C.c_int32, [C.c_void_p]),
    "ow_group_context": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "ow_group_update": (C.c_int, [C.c_void_p, C.c_double, _P(ow_cascade_params), C.c_int32]),
    "ow_group_process": (C.c_int, [C.c_void_p]),
    "ow_group_update_all": (C.c_int, [C.c_void_p, C.c_double, _P(ow_cascade_params), C.c_int32]),
    "ow_group_run": (C.c_int, [C.c_void_p, C.c_double, _P(ow_cascade_params), C.c_int32, C.c_int32]),
    "ow_group_cascades_remaining": (C.c_int32, [C.c_void_p]),
    "ow_group_sync": (C.c_int, [C.c_void_p]),
    "ow_group_gather_begin": (C.c_int, [C.c_void_p]),
    "ow_group_gather_wait": (C.c_int, [C.c_void_p]),
    "ow_group_gather_stats": (C.c_int, [C.c_void_p, _P(C.c_float), _P(C.c_size_t)]),
    "ow_group_link_info": (C.c_int, [C.c_void_p, C.c_int32, _P(ow_group_link)]),
    "ow_query_link": (C.c_int, [C.c_int32, C.c_int32, _P(ow_group_link)]),
    "ow_group_get_device_ptrs": (C.c_int, [C.c_void_p, _P(C.c_void_p), _P(C.c_void_p), _P(C.c_size_t)]),
    "ow_group_get_maps": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "ow_group_sample_surface": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "ow_export_maps": (C.c_int, [C.c_void_p, _P(C.c_int32), _P(C.c_int32), _P(C.c_size_t)]),
    "ow_import_buffer": (C.c_int, [C.c_int32, C.c_int32, C.c_size_t, C.c_size_t, _P(C.c_void_p), _P(C.c_void_p)]),
    "ow_release_buffer": (None, [C.c_void_p]),
    "ow_last_error": (C.c_char_p, []),
    "ow_abi_version": (C.c_int32, []),
}

_lib = None


def load():
    """Load libocean_waves.so (built in-tree by godotoceanwaves_amd.build).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OceanWavesError(-1, f"{LIB_PATH} is missing: run `python -m godotoceanwaves_amd.build` "
                                  "(hipcc, gfx950).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status != OW_OK:
        raise OceanWavesError(status, load().ow_last_error().decode("utf-8", "replace"))
