#!/usr/bin/env python3
"""Bank-conflict model of the row transforms' LDS traffic (developer tool; nothing here runs on a GPU).

LDS: 64 banks x 4 B.  A wave's ds_read_b64 / ds_write_b64 is served in two halves of 32 lanes, every lane taking a pair of adjacent
banks; a half needs as many passes as its busiest bank has distinct addresses (equal addresses broadcast).  For every map size the
script walks the exchange accesses of one row transform (ow_device.h fft_stage_write / fft_stage_read, several rows per wave where a
row is shorter than a wave) under the padding functions e + (e >> 4) (rounds 1-2) and e + (e >> 5), and prints passes / ideal.
Round 2's PMC counters put 14 % of the tick-pair kernel's LDS-active cycles in bank conflicts; the model gives +14.5 % for the 1024
plan under e + (e >> 4) (the stage-1 reads take two passes per half) and 0 under e + (e >> 5)."""


def plan(N):
    T, S = N // 16, (2 if N <= 256 else 3)
    R = [16, 8 if N == 128 else 16, N // 256 if N > 256 else None][:S]
    s = [1, 16, 16 * R[1]][:S]
    return T, S, R, s


def region(N):
    return N + N // 16 + 4  # plan_region_cplx


def passes(addrs):
    total = 0
    for h in range(2):
        banks = {}
        for a in set(addrs[32 * h:32 * h + 32]):
            for b in ((2 * a) % 64, (2 * a + 1) % 64):
                banks[b] = banks.get(b, 0) + 1
        total += max(banks.values())
    return total


def transform(N, slot, lane_exchange=True):
    T, S, R, s = plan(N)
    waves = [0] if T <= 64 else [0, 1]
    rows = lambda w: [((w * 64 + l) // T, (w * 64 + l) % T) for l in range(64)]  # (row in block, lane of the row)
    out = {}
    for J in range(S - 1):
        if J == 1 and lane_exchange and N in (512, 1024):
            continue  # the last exchange runs on the row-swap instructions, not through LDS
        Rj, sj, Rn, sn = R[J], s[J], R[J + 1], s[J + 1]
        B, Bn, mn = (N // Rj) // T, (N // Rn) // T, (N // sn) // Rn
        w = sum(passes([r * region(N) + slot(((t + T * b) % sj) + sj * (Rj * ((t + T * b) // sj) + k)) for r, t in rows(wv)])
                for wv in waves for b in range(B) for k in range(Rj))
        r_ = sum(passes([r * region(N) + slot(((t + T * b) % sn) + sn * (((t + T * b) // sn) + mn * i)) for r, t in rows(wv)])
                 for wv in waves for b in range(Bn) for i in range(Rn))
        out[f"stage {J} write"] = (w, 2 * len(waves) * B * Rj)
        out[f"stage {J + 1} read"] = (r_, 2 * len(waves) * Bn * Rn)
    return out


def staging_1024():
    """the other LDS accesses of the 1024^2 pass 1 (twiddle table reads, staged rows for the transposed store): all conflict-free"""
    N, T = 1024, 64
    ops = []
    for (R, s, m) in ((16, 1, 64), (16, 16, 4)):
        ops += [[(k - 1) * m + t // s for t in range(64)] for k in range(1, R)]
    ops += [[t + T * o for t in range(64)] for o in range(16)]
    ops += [[(tau % 8) * region(N) + tau // 8 + T * k for tau in range(64)] for k in range(16)]
    return sum(passes(o) for o in ops), 2 * len(ops)


if __name__ == "__main__":
    for N in (128, 256, 512, 1024, 2048):
        for name, f in (("e + (e >> 4)", lambda e: e + (e >> 4)), ("e + (e >> 5)", lambda e: e + (e >> 5))):
            r = transform(N, f)
            tot, ideal = sum(v[0] for v in r.values()), sum(v[1] for v in r.values())
            print(f"N = {N:4d}  {name}:  {tot:4d} passes / {ideal:4d} ideal   " + "  ".join(f"{k} {v[0]}/{v[1]}" for k, v in r.items()))
    st, st_ideal = staging_1024()
    for name, f in (("e + (e >> 4)", lambda e: e + (e >> 4)), ("e + (e >> 5)", lambda e: e + (e >> 5))):
        r = transform(1024, f)
        tot, ideal = sum(v[0] for v in r.values()) + st, sum(v[1] for v in r.values()) + st_ideal
        print(f"1024^2 pass 1, all LDS accesses of one row transform + staging, {name}: {tot} / {ideal} = +{100 * (tot - ideal) / ideal:.1f} %")
