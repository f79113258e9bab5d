#!/usr/bin/env python3
"""Parity margins of the SHIPPED build against the oracle, per BASELINE configuration: for every cascade, after `frames`
ticks through ow_run (the merged launches bench.py times) on a PRODUCTION context (no OW_FLAG_DEBUG_F32),
  * the share of FP16 texel-channels of the two maps that are BIT-EQUAL to the oracle's,
  * the share within ONE FP16 ulp (of the oracle's value), and
  * the share that needs tests/helpers.fp16_close's floor (more than one ulp away, but within 1 ulp + 1e-5 x channel maximum) -- these sit at
    zero crossings, where the ulp shrinks with the value while the FP32 error of a long transform does not --, and anything beyond (must be 0);
  * and, from a debug context run side by side (bit-identical maps: tests/test_instantiations.py), the worst FP32 channel error.
    python scripts/parity_margins.py [n:c ...]  > profiles/rNN_parity_margins.txt
The same measurement is a -m gpu test with bounds on the shares: tests/test_parity_margins.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
from godotoceanwaves_amd import WaveCascadeParameters, WaveGenerator, cascade_preset, UPDATE_DELTA

CONFIGS = [(256, 4), (1024, 4), (1024, 8), (2048, 4)]   # BASELINE C2, C3 (headline), C4 (per-node total), C5
FRAMES = 4


def shares(got, ref):
    """got, ref: float16 [..., C] -> (bit-equal, <= 1 ulp, needs the floor, beyond) as fractions of all texel-channels, worst ratio of fp16_close"""
    g, r = np.asarray(got).view(np.float16), np.asarray(ref).view(np.float16)
    gb, rb = g.view(np.uint16), r.view(np.uint16)
    eq = gb == rb
    gf, rf = g.astype(np.float64), r.astype(np.float64)
    err = np.abs(gf - rf)
    ulp = np.spacing(np.abs(r)).astype(np.float64)
    chmax = np.abs(rf).reshape(-1, rf.shape[-1]).max(axis=0)
    one = err <= ulp
    floor = err <= ulp + 1e-5 * chmax
    n = eq.size
    return eq.sum() / n, (one & ~eq).sum() / n, (floor & ~one).sum() / n, (~floor).sum() / n, float((err / (ulp + 1e-5 * chmax)).max())


def measure(n, c, frames=FRAMES, verbose=False):
    """-> dict(bit_equal, one_ulp, floor, beyond: shares of the FP16 texel-channels of both maps over all cascades; worst_ratio; worst_f32; family)"""
    ids = list(range(c))
    prod = WaveGenerator(); prod.map_size = n; prod.init_gpu(max(2, c))
    dbg = WaveGenerator(); dbg.map_size = n; dbg.debug_f32 = True; dbg.init_gpu(max(2, c))
    pp = [WaveCascadeParameters(**cascade_preset(ci)) for ci in ids]
    pd = [WaveCascadeParameters(**cascade_preset(ci)) for ci in ids]
    og = H.oracle_generator(n, ids)
    try:
        prod.run(UPDATE_DELTA, pp, frames); dbg.run(UPDATE_DELTA, pd, frames)
        for _ in range(frames):
            og.update_all(UPDATE_DELTA)
        prod.sync(); dbg.sync()
        tot = np.zeros(4); worst_ratio = 0.0; worst_f32 = 0.0
        family = prod.last_kernel_family()
        for i in range(c):
            disp, norm = prod.get_maps(i)
            dd, dn = dbg.get_maps(i)
            assert np.array_equal(disp.view(np.uint16), dd.view(np.uint16)) and np.array_equal(norm.view(np.uint16), dn.view(np.uint16)), "debug and production maps differ"
            sd = shares(disp[..., :3], og.displacement(i).view(np.float16)[..., :3])
            sn = shares(norm[..., :3], og.normal(i).view(np.float16)[..., :3])
            foam_err = np.abs(norm[..., 3].astype(np.float64) - og.normal(i)[..., 3].view(np.float16).astype(np.float64)).max()
            f32, ref = dbg.get_maps_f32(i), og.f32(i)
            e32 = max(H.relmax(f32[..., ch], ref[..., ch]) for ch, name in enumerate(H.CHANNELS) if name != "foam")
            worst_f32 = max(worst_f32, e32)
            if verbose:
                print(f"{n}^2 x {c} cascade {i} ({family}): displacement {sd[0]*100:8.4f} % | {sd[1]*100:7.4f} % | {sd[2]*100:7.4f} % | {sd[3]*100:.4f} % | {sd[4]:.2f}"
                      f"    normal {sn[0]*100:8.4f} % | {sn[1]*100:7.4f} % | {sn[2]*100:7.4f} % | {sn[3]*100:.4f} % | {sn[4]:.2f}    foam max |diff| {foam_err:.2e}    FP32 worst channel {e32:.1e}", flush=True)
            tot += (np.array(sd[:4]) + np.array(sn[:4])) / 2
            worst_ratio = max(worst_ratio, sd[4], sn[4])
        tot /= c
    finally:
        prod.free(); dbg.free(); og.close()
    return {"bit_equal": float(tot[0]), "one_ulp": float(tot[1]), "floor": float(tot[2]), "beyond": float(tot[3]), "worst_ratio": worst_ratio,
            "worst_f32": worst_f32, "family": family}


if __name__ == "__main__":
    cfgs = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or CONFIGS
    print(f"# parity margins of the shipped build vs the oracle after {FRAMES} ticks through ow_run (production context; FP32 errors from a debug context side by side)")
    print("# share of FP16 texel-channels: bit-equal | within 1 ulp | needs fp16_close's 1e-5 x max floor | beyond (must be 0) | worst ratio of (1 ulp + floor)")
    for n, c in cfgs:
        m = measure(n, c, verbose=True)
        print(f"== {n}^2 x {c}: bit-equal {m['bit_equal']*100:.4f} %, within one ulp {m['one_ulp']*100:.4f} %, needs the floor {m['floor']*100:.5f} %, beyond {m['beyond']*100:.5f} %; "
              f"worst ratio {m['worst_ratio']:.2f}; worst FP32 channel error {m['worst_f32']:.1e} (tolerance 1e-4)", flush=True)
