#!/usr/bin/env python3
"""CPU model of fm_gemm_tn_multi's work distribution (host planning in fm_gemm_tn_multi + the segment generator at the end of
gemm_tn_multi_kernel, csrc/gemm.hip), checked exhaustively on random job lists: every k-tile of every output tile is reduced exactly
once, every workgroup's walk terminates, and the per-workgroup loads are balanced.  python tools/tn_multi_plan_check.py [n_cases]"""
import random
import sys


def plan(jobs, cus, big=True, bands=True, hybrid_on=True):
    """jobs: list of (N, K, R).  Mirrors the host code.  big: True = 256 x 256 tiles with K-step 32, False = 128 x 256 / 64,
    "ls" = the lock-step form (256 x 256 tiles, K-step 64: fm_set_gemm_tn_config(4)), "t4" = gemm_tn4.hip (256 x 384 tiles, K-step 64, the
    lock-step form's planner constants)."""
    ls = big in ("ls", "t4")
    ta, kb = (256, 64) if ls else (256, 32) if big else (128, 64)
    tb = 384 if big == "t4" else 256
    J, tiles, units = [], 0, 0
    for N, K, R in jobs:
        ntb = (K + tb - 1) // tb
        t = ((N + ta - 1) // ta) * ntb
        kt = (R + kb - 1) // kb
        J.append(dict(tiles=t, tile_start=tiles, kt=kt, q=0, lb=0))
        tiles += t
        units += t * kt
    grid = cus
    if units // 8 < grid:
        grid = (units // 8 + 7) // 8 * 8
    if grid < 8:
        grid = 8
    rem, ntail = tiles % grid, grid - tiles % grid
    tail_rr = rem >= ntail
    c = (12.0 if tail_rr else 64.0) if ls else (24.0 if tail_rr else 128.0) if big else 8.0
    cb = 16.0 if ls else 32.0
    banded = (not tail_rr) and rem > 0 and ntail > rem and bands
    hybrid = tail_rr and rem > ntail and ntail > 0 and hybrid_on
    for j in J:
        j["lb"] = 0
        if rem == 0:
            j["q"] = j["kt"]
            continue
        if banded:
            rw = rem / (ntail - rem)
            q = (rw * j["kt"] + rw * cb) / (1.0 + 2.0 * rw)
            q = max(q, 0)
            if 2 * q > j["kt"]:
                q = j["kt"] // 2          # C: integer division of kt by 2 converted to double
            j["q"] = j["lb"] = int(q)
            continue
        if hybrid:
            r = rem / ntail
            j["q"] = int((j["kt"] * r + c) / (1.0 + r))
        elif tail_rr:
            n = (rem + ntail - 1) // ntail
            left = (j["kt"] - (n - 1) * c) / (n + 1.0)
            j["q"] = j["kt"] - (int(left) if left > 0 else 0)
        else:
            r = rem / ntail
            j["q"] = int((j["kt"] * r + r * c) / (1.0 + r))
        j["q"] = min(max(j["q"], 0), j["kt"])
    return J, tiles, grid, tail_rr, banded, hybrid


def segments(J, tiles, G, tail_rr, banded, w, limit=100000, hybrid=False):
    """The device loop of workgroup w (logical index).  Yields (tile, t0, t1); raises on non-termination."""
    n_jobs = len(J)

    def job_of(tile):
        j = 0
        while j + 1 < n_jobs and tile >= J[j + 1]["tile_start"]:
            j += 1
        return j
    full = tiles // G
    T0 = full * G
    rem = tiles - T0
    ntail = G - rem
    nband = 0 if hybrid else (rem if banded else 0)
    nwalk = ntail - nband
    walk_T0 = T0 + ntail if hybrid else T0
    q2 = lambda j: J[j]["q"] + (J[j]["lb"] if (banded and not hybrid) else 0)
    u0 = u1 = 0
    if rem > 0 and w >= rem + nband and (not tail_rr or hybrid):
        Lsum = 0
        for j in range(n_jobs):
            lo, hi = max(J[j]["tile_start"], walk_T0), J[j]["tile_start"] + J[j]["tiles"]
            if hi > lo:
                Lsum += (hi - lo) * (J[j]["kt"] - q2(j))
        u0, u1 = Lsum * (w - rem - nband) // nwalk, Lsum * (w - rem - nband + 1) // nwalk
    phase, f, tj, ci, P = 0, 0, 0, -1, 0
    steps = 0
    while True:
        steps += 1
        if steps > limit:
            raise RuntimeError("segment generator does not terminate")
        tile = t0 = t1 = 0
        have = False
        if phase == 0:
            if f < full:
                tile = f * G + w; t1 = J[job_of(tile)]["kt"]; f += 1; have = True
            else:
                phase = 3 if rem == 0 else (1 if w < rem else (4 if (not tail_rr and w < rem + nband) else 2)); f = 0
        elif phase == 1:
            tile = T0 + w; t1 = J[job_of(tile)]["q"]; phase = 3; have = True
        elif phase == 4:
            tile = T0 + (w - rem); j = job_of(tile); t0 = J[j]["q"]; t1 = t0 + J[j]["lb"]; phase = 3; have = True
        elif phase == 2 and tail_rr and not (hybrid and f > 0):
            sgm = (w - rem) + f * ntail
            if sgm >= rem:
                phase = 3
            else:
                tile = T0 + sgm; j = job_of(tile); t0 = J[j]["q"]; t1 = J[j]["kt"]; f += 1; have = True
        elif phase == 2:
            if tj >= n_jobs:
                phase = 3
            else:
                lo, hi = max(J[tj]["tile_start"], walk_T0), J[tj]["tile_start"] + J[tj]["tiles"]
                q = q2(tj); left = J[tj]["kt"] - q
                advance = True
                if hi > lo and left > 0:
                    Pn = P + (hi - lo) * left
                    if u1 > P and u0 < Pn:
                        s0, s1 = max(u0, P) - P, min(u1, Pn) - P
                        if ci < 0:
                            ci = s0 // left
                        base = ci * left
                        if base < s1:
                            tile = lo + ci; t0 = q + (max(s0, base) - base); t1 = q + (min(s1, base + left) - base)
                            ci += 1; have = True; advance = False
                    if advance:
                        P = Pn
                if advance:
                    tj += 1; ci = -1
        else:
            break
        if have and t0 < t1:
            yield tile, t0, t1


def check(jobs, cus, big, bands=True, hybrid_on=True):
    J, tiles, G, rr, banded, hybrid = plan(jobs, cus, big, bands, hybrid_on)
    cover = {}
    load = []
    for w in range(G):
        tot = 0
        for tile, t0, t1 in segments(J, tiles, G, rr, banded, w, hybrid=hybrid):
            assert 0 <= tile < tiles and 0 <= t0 < t1, (tile, t0, t1)
            for k in range(t0, t1):
                key = (tile, k)
                assert key not in cover, ("k-tile reduced twice", key, jobs, cus)
                cover[key] = w
            tot += t1 - t0
        load.append(tot)
    want = sum(j["tiles"] * j["kt"] for j in J)
    assert len(cover) == want, ("k-tiles missing", want - len(cover), jobs, cus, big)
    for j in J:
        for t in range(j["tile_start"], j["tile_start"] + j["tiles"]):
            assert all((t, k) in cover for k in range(j["kt"])), (t, jobs)
    return max(load), want / G, banded, rr


def main(n):
    rng = random.Random(0)
    worst = 0.0
    shapes = [768, 2304, 2048, 1536, 64, 200, 130, 1024, 2752, 3072, 4096, 100]
    stats = dict(banded=0, rr=0, other=0)
    for it in range(n):
        nj = rng.randint(1, 16)
        R = rng.choice([60, 54, 37, 64, 1000, 4100, 32768, 16384, 8192, 300])
        jobs = [(rng.choice(shapes), rng.choice(shapes), R if rng.random() < 0.8 else rng.choice([33, 64, 500, 32768])) for _ in range(nj)]
        cus = rng.choice([256, 248, 240, 192, 64])
        big = rng.random() < 0.7
        mx, avg, banded, rr = check(jobs, cus, big)
        stats["banded" if banded else "rr" if rr else "other"] += 1
        if avg > 200:
            worst = max(worst, mx / avg)
    # the launches the engine actually makes
    enc = [(2048, 768, 32768)] * 2 + [(768, 2048, 32768), (768, 768, 32768), (2304, 768, 32768)]
    dec = enc + [(768, 768, 32768)] * 2 + [(1536, 768, 32768)]
    micro = [(96, 64, 60)] * 2 + [(64, 96, 60), (64, 64, 60), (192, 64, 60)]
    for name, jobs in (("4M-B encoder layer", enc), ("4M-B decoder layer", dec), ("micro encoder layer", micro), ("micro decoder", micro + [(64, 64, 54)] * 2 + [(128, 64, 60)])):
        for cus in (256, 240):
            mx, avg, banded, rr = check(jobs, cus, True)
            print(f"{name:22s} {cus} CUs: max load {mx} k-tiles, mean {avg:.1f}, banded={banded} rr={rr}")
    print(f"{n} random job lists: exact cover, all walks terminate; modes {stats}; worst max/mean load {worst:.2f} (lists with > 200 k-tiles per workgroup)")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 400)
