#!/usr/bin/env python3
"""Bank-conflict model of the row transforms' LDS traffic (developer tool; nothing here runs on a GPU).

Rules (/opt/skills/guides/MI355X_MICROARCH.md, LDS): 64 banks of 4 B.  ds_read_b64: two groups of 32 lanes, bank = (a / 4) mod 64;
ds_write_b64: four groups of 16 CONTIGUOUS lanes, bank = (a / 4) mod 32.  A group needs as many passes as its busiest bank has distinct
addresses (equal addresses broadcast).  For every map size the script walks the exchange accesses of one row transform (ow_device.h
fft_stage_write / fft_stage_read; several rows per wave where a row is shorter than a wave) under three slot maps and prints the extra
passes.  Checked against the counters: map 0 gives +32 passes per 1024-point transform = the 14 % of LDS-active cycles round 2 measured as
SQ_LDS_BANK_CONFLICT; map 1 gives +64 = the doubling round 3 measured (profiles/r03_pmc_lds_counters.txt); map 2 gives 0, and the counter
reads 0.  None of it moves the kernels' time (the LDS is not their critical path) and map 2 costs registers: map 0 is what ships."""


def plan(N):
    T, S = N // 16, (2 if N <= 256 else 3)
    R = [16, 8 if N == 128 else 16, N // 256 if N > 256 else None][:S]
    s = [1, 16, 16 * R[1]][:S]
    return T, S, R, s


def region(N):
    return N + N // 16 + 4  # plan_region_cplx


def read_passes(addrs):  # addresses in 8-byte units, one per lane
    total = 0
    for h in range(2):
        banks = {}
        for a in set(addrs[32 * h:32 * h + 32]):
            for b in ((2 * a) % 64, (2 * a + 1) % 64):
                banks[b] = banks.get(b, 0) + 1
        total += max(banks.values())
    return total


def write_passes(addrs):
    total = 0
    for g in range(4):
        banks = {}
        for a in set(addrs[16 * g:16 * g + 16]):
            for b in ((2 * a) % 32, (2 * a + 1) % 32):
                banks[b] = banks.get(b, 0) + 1
        total += max(banks.values())
    return total


def transform(N, slot, lane_exchange=True, slot_b=None):
    """slot: map of the exchange after stage 0; slot_b: map of the exchange after stage 1 (default: the same)"""
    T, S, R, s = plan(N)
    waves = [0] if T <= 64 else [0, 1]
    rows = lambda w: [((w * 64 + l) // T, (w * 64 + l) % T) for l in range(64)]  # (row in block, lane of the row)
    out = {}
    for J in range(S - 1):
        if J == 1 and lane_exchange and N in (512, 1024):
            continue  # the last exchange runs on the row-swap instructions, not through LDS
        Rj, sj, Rn, sn = R[J], s[J], R[J + 1], s[J + 1]
        slot_j = slot if J == 0 or slot_b is None else slot_b
        B, Bn, mn = (N // Rj) // T, (N // Rn) // T, (N // sn) // Rn
        w = sum(write_passes([r * region(N) + slot_j(((t + T * b) % sj) + sj * (Rj * ((t + T * b) // sj) + k)) for r, t in rows(wv)])
                for wv in waves for b in range(B) for k in range(Rj))
        r_ = sum(read_passes([r * region(N) + slot_j(((t + T * b) % sn) + sn * (((t + T * b) // sn) + mn * i)) for r, t in rows(wv)])
                 for wv in waves for b in range(Bn) for i in range(Rn))
        out[f"stage {J} write"] = (w, 4 * len(waves) * B * Rj)
        out[f"stage {J + 1} read"] = (r_, 2 * len(waves) * Bn * Rn)
    return out


def swap34(e):
    return (e & ~0x18) | (((e >> 3) & 1) << 4) | (((e >> 4) & 1) << 3)


MAPS = (("0: e + (e >> 4)          [shipped]", lambda e: e + (e >> 4)),
        ("1: e + (e >> 5)          [round 3, first try]", lambda e: e + (e >> 5)),
        ("2: swap34(e) + (e >> 5)  [conflict-free, not shipped]", lambda e: swap34(e) + (e >> 5)))

if __name__ == "__main__":
    for N in (128, 256, 512, 1024, 2048):
        for name, f in MAPS:
            vals = [f(e) for e in range(N)]
            assert len(set(vals)) == N and max(vals) < region(N), (N, name)
            r = transform(N, f)
            tot, ideal = sum(v[0] for v in r.values()), sum(v[1] for v in r.values())
            print(f"N = {N:4d}  map {name:52s} +{tot - ideal:3d} passes over {ideal:3d}   " + "  ".join(f"{k} {v[0]}/{v[1]}" for k, v in r.items()))
        if N == 2048:  # two LDS exchanges per transform: each may take its own map (the region is rewritten in between)
            r = transform(N, MAPS[2][1], slot_b=MAPS[1][1])
            tot, ideal = sum(v[0] for v in r.values()), sum(v[1] for v in r.values())
            print(f"N = {N:4d}  map {'2 for the first exchange, 1 for the second':52s} +{tot - ideal:3d} passes over {ideal:3d}   " + "  ".join(f"{k} {v[0]}/{v[1]}" for k, v in r.items()))
