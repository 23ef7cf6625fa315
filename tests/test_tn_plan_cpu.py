"""Work distribution of fm_gemm_tn_multi (all weight-gradient GEMMs of a layer in one launch): the CPU model in
tools/tn_multi_plan_check.py restates the host planning and the device segment generator of csrc/gemm.hip line by line; here it must
cover every k-tile of every output tile exactly once and terminate, for random job lists and for the lists the engine launches."""
import importlib.util
import os
import random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("tn_multi_plan_check", os.path.join(ROOT, "tools", "tn_multi_plan_check.py"))
M = importlib.util.module_from_spec(spec)
spec.loader.exec_module(M)


def test_random_job_lists_are_covered_exactly_once():
    rng = random.Random(1)
    shapes = [768, 2304, 2048, 1536, 64, 200, 136, 1024, 2752, 3072]
    modes = set()
    for _ in range(60):
        R = rng.choice([60, 37, 64, 1000, 4100, 8192, 300])
        jobs = [(rng.choice(shapes), rng.choice(shapes), R if rng.random() < 0.8 else rng.choice([33, 64, 500])) for _ in range(rng.randint(1, 16))]
        for big in (True, False, "ls", "t4"):
            _, _, banded, rr = M.check(jobs, rng.choice([256, 240, 64]), big)
            modes.add("banded" if banded else "rr" if rr else "plain")
    assert {"banded", "rr"} <= modes


def test_engine_job_lists():
    enc = [(2048, 768, 32768)] * 2 + [(768, 2048, 32768), (768, 768, 32768), (2304, 768, 32768)]
    dec = enc + [(768, 768, 32768)] * 2 + [(1536, 768, 32768)]
    mx, avg, banded, rr = M.check(enc, 256, True)
    assert banded and mx < 1.05 * avg                       # encoder layer: bands + walkers, balanced within 5 %
    mx, avg, banded, rr = M.check(dec, 256, True)
    assert rr and mx < 1.05 * avg                           # decoder layer: one whole tail per tail workgroup + a walk over the other 32 tails
    mx0, avg0, _, _ = M.check(dec, 256, True, hybrid_on=False)
    assert mx0 > 1.15 * avg0                                # (plain round-robin: 32 tail workgroups take a second whole tail)
    for lst in (enc, dec):
        mx, avg, _, _ = M.check(lst, 256, "ls")                # the lock-step configuration
        assert mx < 1.25 * avg
        mx, avg, _, _ = M.check(lst, 256, "t4")                # 256 x 384 tiles (gemm_tn4.hip): 74 / 96 tiles, every one cut
        assert mx < 1.25 * avg
    large = [(3072, 1024, 16384), (1024, 1024, 16384), (2752, 1024, 16384), (2752, 1024, 16384), (1024, 2752, 16384)]
    M.check(large, 256, True)                               # 4M-L encoder layer: a full round + a cut
    M.check(large + [(1024, 1024, 16384)] * 2 + [(2048, 1024, 16384)], 256, True)
