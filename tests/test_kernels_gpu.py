"""Per-kernel numerics on a real MI355X: every HIP kernel against a plain fp32 PyTorch restatement of
the same op on the same (bf16-rounded) inputs.  Integer outputs must match exactly; floating point
within the tolerance written next to each check (bf16 output rounding = 2^-8 relative)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from fourm.hip import ops, _lib
    return ops, _lib


def bf(t):
    return t.to(torch.bfloat16)


def rel_err(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


def max_err(a, b):
    return float((a.float() - b.float()).abs().max())


def randn(*shape, scale=1.0, seed=None):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed if seed is not None else hash(shape) % (2 ** 31))
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


# ------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M", [1, 5, 16, 32])
@pytest.mark.parametrize("N,K", [(768, 768), (2304, 768), (30000, 768), (40, 64), (768, 2752), (1000, 4096)])
def test_gemm_nt_skinny_rows(M, N, K):
    """fm_gemm_nt on a handful of rows (a decoding step) runs the weight-streaming kernel of csrc/gemm_skinny.hip: bf16 (+ bias), fp32
    residual (+ bias) and SwiGLU epilogues against fp32 torch with gemm.hip's rounding points; rows / features outside the problem stay untouched."""
    ops, L = _ops()
    x = bf(randn(M, K, seed=11, scale=0.5))
    w, w3 = bf(randn(N, K, seed=12, scale=0.1)), bf(randn(N, K, seed=13, scale=0.1))
    bias = randn(N, seed=14, scale=0.3)
    ref = x.float() @ w.float().t()
    ldo = ops.ru(N, 8)
    for b in (None, bias):
        out = torch.full((M + 3, ldo), 7.0, device=DEV, dtype=torch.bfloat16)
        ops.gemm_nt(x, w, out, bias=b, M=M, N=N, K=K)
        want = ref + (bf(b).float() if b is not None else 0.0)
        assert rel_err(out[:M, :N], want) < 4e-3, (M, N, K, b is not None)
        assert bool((out[M:] == 7.0).all()) and bool((out[:M, N:] == 7.0).all())
        res = randn(M + 3, ldo, seed=15)
        o32 = torch.full((M + 3, ldo), 7.0, device=DEV)
        ops.gemm_nt(x, w, o32, epilogue=L.EPI_RESIDUAL, res=res, bias=b, M=M, N=N, K=K)
        assert max_err(o32[:M, :N], res[:M, :N] + bf(want).float()) < 2e-2 * float(want.abs().max()), (M, N, K)
        assert bool((o32[M:] == 7.0).all())
    if N % 8 == 0:
        Hp = ops.ru(N, 64)
        act = torch.full((M, Hp), 3.0, device=DEV, dtype=torch.bfloat16)
        gu = torch.full((M, 2 * Hp), 3.0, device=DEV, dtype=torch.bfloat16)
        ops.gemm_nt(x, w, act, epilogue=L.EPI_SWIGLU, w2=w3, out2=gu, Hp=Hp, M=M, N=N, K=K)
        g, u = bf(ref).float(), bf(x.float() @ w3.float().t()).float()
        assert rel_err(gu[:, :N], g) < 4e-3 and rel_err(gu[:, Hp:Hp + N], u) < 4e-3
        assert rel_err(act[:, :N], bf(torch.nn.functional.silu(g)).float() * u) < 8e-3
        act2 = torch.full((M, Hp), 3.0, device=DEV, dtype=torch.bfloat16)
        ops.gemm_nt(x, w, act2, epilogue=L.EPI_SWIGLU, w2=w3, Hp=Hp, M=M, N=N, K=K)          # inference: no (g | u) copy
        assert torch.equal(act2[:, :N], act[:, :N])


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12])
def test_gemm_nt_tile_configs(cfg):
    """Every tile configuration of fm_gemm_nt (fm_set_gemm_nt_config) on ragged shapes and all epilogue kinds."""
    ops, L = _ops()
    L.lib.fm_set_gemm_nt_config(cfg)
    try:
        for M, N, K in [(300, 260, 192), (1000, 768, 768), (513, 2304, 64)]:
            x = bf(randn(M, K, seed=1) + torch.arange(K, device=DEV)[None] * 0.01)
            w = bf(randn(N, K, seed=2) * 0.1 + torch.arange(N, device=DEV)[:, None] * 0.001)
            out = torch.full((M, N), 7.0, device=DEV, dtype=torch.bfloat16)
            ops.gemm_nt(x, w, out)
            assert rel_err(out, x.float() @ w.float().t()) < 4e-3, (cfg, M, N, K)
        M, K, H = 700, 128, 170
        Hp = ops.ru(H, 64)
        x, w1, w3 = bf(randn(M, K, seed=7)), bf(randn(H, K, seed=8) * 0.2), bf(randn(H, K, seed=9) * 0.2)
        gu = torch.full((M, 2 * Hp), 3.0, device=DEV, dtype=torch.bfloat16)
        act = torch.full((M, Hp), 3.0, device=DEV, dtype=torch.bfloat16)
        ops.gemm_nt(x, w1, act, epilogue=L.EPI_SWIGLU, w2=w3, out2=gu, Hp=Hp, N=H)
        g, u = bf(x.float() @ w1.float().t()).float(), bf(x.float() @ w3.float().t()).float()
        assert rel_err(gu[:, :H], g) < 4e-3 and rel_err(gu[:, Hp:Hp + H], u) < 4e-3, cfg
        assert rel_err(act[:, :H], bf(torch.nn.functional.silu(g)).float() * u) < 8e-3, cfg
        res = randn(M, 320, seed=6); buf = res.clone()
        w = bf(randn(320, K, seed=4) * 0.2)
        ops.gemm_nt(x, w, buf, epilogue=L.EPI_RESIDUAL, res=buf)
        assert rel_err(buf, res + bf(x.float() @ w.float().t()).float()) < 3e-3, cfg
    finally:
        L.lib.fm_set_gemm_nt_config(9 + 256)


NT_TILED = 9 + 256                   # automatic configuration, tile-at-a-time kernels
NT_FLAT = 9 + 256 + (1 << 29)        # + the flattened persistent kernel where it applies


@pytest.mark.parametrize("M,N,K", [(32768, 768, 768), (6000, 2304, 768), (2048, 128, 512), (4100, 768, 4096), (33000, 1536, 1024)])
def test_gemm_nt_flat_bit_identical(M, N, K):
    """The flattened persistent kernel (gemm_nt_flat.hip: one K-tile stream across all tiles of a workgroup, deferred
    stores) accumulates in the same order as the tile-at-a-time kernels: outputs must be bit-identical, run after run (a race
    in the staggered LDS ring or a miscounted vmcnt wait would show as rare wrong tiles), and right against fp32."""
    ops, L = _ops()
    x = bf(randn(M, K, seed=21) * 0.7)
    w = bf(randn(N, K, seed=22) * 0.05)
    L.lib.fm_set_gemm_nt_config(NT_TILED)
    ref = torch.full((M, N), 5.0, device=DEV, dtype=torch.bfloat16)
    ops.gemm_nt(x, w, ref)
    rows = torch.randperm(M, device=DEV)[:512]
    assert rel_err(ref[rows], x[rows].float() @ w.float().t()) < 4e-3
    L.lib.fm_set_gemm_nt_config(NT_FLAT)
    try:
        for rep in range(6):
            out = torch.full((M, N), -3.0, device=DEV, dtype=torch.bfloat16)
            ops.gemm_nt(x, w, out)
            assert torch.equal(out, ref), (rep, int((out != ref).sum()))
    finally:
        L.lib.fm_set_gemm_nt_config(NT_TILED)


@pytest.mark.parametrize("M,H,K,save", [(32768, 2048, 768, True), (5000, 448, 768, True), (4096, 2048, 768, False)])
def test_gemm_nt_flat_swiglu_bit_identical(M, H, K, save):
    ops, L = _ops()
    x, w1, w3 = bf(randn(M, K, seed=23) * 0.7), bf(randn(H, K, seed=24) * 0.05), bf(randn(H, K, seed=25) * 0.05)

    def run():
        gu = torch.full((M, 2 * H), 3.0, device=DEV, dtype=torch.bfloat16) if save else None
        act = torch.full((M, H), 3.0, device=DEV, dtype=torch.bfloat16)
        ops.gemm_nt(x, w1, act, epilogue=L.EPI_SWIGLU, w2=w3, out2=gu, Hp=H, N=H)
        return act, gu
    L.lib.fm_set_gemm_nt_config(NT_TILED)
    act0, gu0 = run()
    rows = torch.randperm(M, device=DEV)[:256]
    g, u = bf(x[rows].float() @ w1.float().t()).float(), bf(x[rows].float() @ w3.float().t()).float()
    assert rel_err(act0[rows], bf(torch.nn.functional.silu(g)).float() * u) < 8e-3
    L.lib.fm_set_gemm_nt_config(NT_FLAT)
    try:
        for rep in range(4):
            act, gu = run()
            assert torch.equal(act, act0), rep
            if save:
                assert torch.equal(gu, gu0), rep
    finally:
        L.lib.fm_set_gemm_nt_config(NT_TILED)


@pytest.mark.parametrize("M,N,K", [(32768, 768, 768), (4096, 2304, 768), (2048, 1536, 1024), (8192, 768, 4096), (2304, 384, 192), (6144, 768, 2048)])
@pytest.mark.parametrize("mode", [1, 5])
def test_gemm_nt4_bit_identical(M, N, K, mode):
    """The 4-wave 256 x 384-tile kernel (gemm_nt4.hip: 16 accumulator fragments in hand-named AGPRs, staged whole-line epilogue through the stage
    buffer a tile's last barrier freed; mode 5 = unstaged stores) accumulates in gemm_nt3's order: bit-identical outputs, run after run, and right
    against fp32.  One / several tiles per workgroup, K from 3 K-tiles up."""
    ops, L = _ops()
    x = bf(randn(M, K, seed=31) * 0.7 + torch.arange(K, device=DEV)[None] * 0.001)
    w = bf(randn(N, K, seed=32) * 0.05 + torch.arange(N, device=DEV)[:, None] * 0.0001)
    try:
        L.lib.fm_lab_set(4, 0)
        ref = torch.full((M, N), 5.0, device=DEV, dtype=torch.bfloat16)
        ops.gemm_nt(x, w, ref)
        rows = torch.randperm(M, device=DEV)[:512]
        assert rel_err(ref[rows], x[rows].float() @ w.float().t()) < 4e-3
        L.lib.fm_lab_set(4, mode)
        for rep in range(6):
            out = torch.full((M, N), -3.0, device=DEV, dtype=torch.bfloat16)
            ops.gemm_nt(x, w, out)
            assert torch.equal(out, ref), (rep, int((out != ref).sum()))
        # a column view of a wider buffer (ldo > N) and row-strided operands (ldx, ldw > K): what the engine passes for q / k / v slices
        xs = torch.zeros(M, K + 64, device=DEV, dtype=torch.bfloat16); xs[:, :K] = x
        ws = torch.zeros(N, K + 128, device=DEV, dtype=torch.bfloat16); ws[:, :K] = w
        big = torch.full((M, N + 256), 9.0, device=DEV, dtype=torch.bfloat16)
        ops.gemm_nt(xs[:, :K], ws[:, :K], big[:, 128:128 + N], K=K)
        assert torch.equal(big[:, 128:128 + N], ref) and bool((big[:, :128] == 9.0).all()) and bool((big[:, 128 + N:] == 9.0).all())
    finally:
        L.lib.fm_lab_set(4, 1)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (300, 260, 192), (1024, 768, 768), (70, 2304, 768)])
def test_gemm_nt_plain(M, N, K):
    ops, L = _ops()
    # asymmetric operands (transposed outputs would not pass)
    x = bf(randn(M, K, seed=1) + torch.arange(K, device=DEV)[None] * 0.01)
    w = bf(randn(N, K, seed=2) * 0.1 + torch.arange(N, device=DEV)[:, None] * 0.001)
    out = torch.full((M, N), 7.0, device=DEV, dtype=torch.bfloat16)
    ops.gemm_nt(x, w, out)
    ref = x.float() @ w.float().t()
    assert rel_err(out, ref) < 4e-3, rel_err(out, ref)
    assert max_err(out, bf(ref)) <= 2 ** -6 * float(ref.abs().max())


def test_gemm_nt_identity_layout():
    """A = I check with an asymmetric right operand: catches row/column swaps in the fragment maps."""
    ops, L = _ops()
    K = 128
    x = torch.eye(K, device=DEV, dtype=torch.bfloat16)                       # (M=128, K)
    w = bf(torch.arange(256 * K, device=DEV).reshape(256, K).float() % 251 - 125)  # (N=256, K), exact in bf16
    out = torch.zeros(K, 256, device=DEV, dtype=torch.bfloat16)
    ops.gemm_nt(x, w, out)
    assert torch.equal(out.float(), w.float().t())


@pytest.mark.parametrize("epi", ["bias", "gelu", "residual", "f32"])
def test_gemm_nt_epilogues(epi):
    ops, L = _ops()
    M, N, K = 200, 320, 128
    x, w = bf(randn(M, K, seed=3)), bf(randn(N, K, seed=4) * 0.2)
    bias = randn(N, seed=5)
    acc = x.float() @ w.float().t() + bf(bias).float()
    if epi == "bias":
        out = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
        ops.gemm_nt(x, w, out, bias=bias)
        assert rel_err(out, acc) < 4e-3
    elif epi == "gelu":
        out = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
        pre = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
        ops.gemm_nt(x, w, out, bias=bias, epilogue=L.EPI_GELU, out2=pre)
        assert rel_err(pre, acc) < 4e-3
        assert rel_err(out, torch.nn.functional.gelu(bf(acc).float())) < 6e-3
    elif epi == "residual":
        res = randn(M, N, seed=6)
        buf = res.clone()
        ops.gemm_nt(x, w, buf, bias=bias, epilogue=L.EPI_RESIDUAL, res=buf)     # in place
        assert rel_err(buf, res + bf(acc).float()) < 3e-3
    else:
        bias32 = bias
        out = torch.zeros(M, N, device=DEV, dtype=torch.float32)
        ops.gemm_nt(x, w, out, bias=bias32, epilogue=L.EPI_F32)
        assert rel_err(out, x.float() @ w.float().t() + bias32) < 1e-5


@pytest.mark.parametrize("H", [128, 170, 2048])
def test_gemm_nt_swiglu(H):
    ops, L = _ops()
    M, K = 192, 128
    Hp = ops.ru(H, 64)
    x = bf(randn(M, K, seed=7))
    w1, w3 = bf(randn(H, K, seed=8) * 0.2), bf(randn(H, K, seed=9) * 0.2)
    gu = torch.full((M, 2 * Hp), 3.0, device=DEV, dtype=torch.bfloat16)
    act = torch.full((M, Hp), 3.0, device=DEV, dtype=torch.bfloat16)
    ops.gemm_nt(x, w1, act, epilogue=L.EPI_SWIGLU, w2=w3, out2=gu, Hp=Hp, N=H)
    g, u = bf(x.float() @ w1.float().t()).float(), bf(x.float() @ w3.float().t()).float()
    a = bf(torch.nn.functional.silu(g)).float() * u
    assert rel_err(gu[:, :H], g) < 4e-3 and rel_err(gu[:, Hp:Hp + H], u) < 4e-3
    assert rel_err(act[:, :H], a) < 8e-3
    # columns past H inside the last written 4-group must be zero (GEMM K-padding contract)
    last = min(Hp, ops.ru(H, 4))
    assert float(act[:, H:last].float().abs().max() if last > H else 0.0) == 0.0


@pytest.mark.parametrize("tr", [0, 1])
@pytest.mark.parametrize("R,N,K", [(64, 128, 128), (256, 200, 136), (1024, 768, 384)])
def test_gemm_tn(tr, R, N, K):
    ops, L = _ops()
    a = bf(randn(R, N, seed=10) + torch.arange(N, device=DEV)[None] * 0.01)
    b = bf(randn(R, K, seed=11))
    out = torch.full((N, K), 1.0, device=DEV, dtype=torch.float32)
    ops.gemm_tn(a, b, out, force_tr=tr)
    ref = 1.0 + a.float().t() @ b.float()
    assert rel_err(out, ref) < 1e-4, (tr, rel_err(out, ref))


@pytest.mark.parametrize("R,N,K,splits", [(64, 128, 128, 0), (320, 200, 136, 1), (4096, 768, 384, 0), (32768, 304, 520, 0)])
def test_gemm_tn_pingpong_schedule(R, N, K, splits):
    """Both schedules of fm_gemm_tn: 1 (default) = K-step 64, one workgroup per CU, wave rows one barrier apart; 0 =
    K-step 32, lock-step.  Repeated to screen for races in the staggered LDS ring."""
    ops, L = _ops()
    a = bf(randn(R, N, seed=14) + torch.arange(N, device=DEV)[None] * 0.01)
    b = bf(randn(R, K, seed=15))
    ref = a.float().t() @ b.float()
    for cfg in (1, 0):
        L.lib.fm_set_gemm_tn_config(cfg)
        try:
            for _ in range(5):
                out = torch.zeros(N, K, device=DEV, dtype=torch.float32)
                ops.gemm_tn(a, b, out, splits=splits)
                assert rel_err(out, ref) < 1e-4, (cfg, rel_err(out, ref))
        finally:
            L.lib.fm_set_gemm_tn_config(TN_DEFAULT)


@pytest.mark.parametrize("cfg", [0, 1])
@pytest.mark.parametrize("R,Rbuf", [(1000, 1024), (37, 64), (4100, 4224), (300, 300)])
def test_gemm_tn_masks_rows_past_R(cfg, R, Rbuf):
    """Reduction rows >= R never reach the result, whatever the buffers hold there (a workspace reused with fewer live rows)."""
    ops, L = _ops()
    N, K = 200, 264
    a, b = bf(randn(Rbuf, N, seed=16)), bf(randn(Rbuf, K, seed=17))
    a[R:] = 1e4; b[R:] = float("nan")
    ref = a[:R].float().t() @ b[:R].float()
    L.lib.fm_set_gemm_tn_config(cfg)
    try:
        out = torch.zeros(N, K, device=DEV, dtype=torch.float32)
        ops.gemm_tn(a, b, out, R=R)
        assert rel_err(out, ref) < 1e-4, (cfg, R, rel_err(out, ref))
    finally:
        L.lib.fm_set_gemm_tn_config(TN_DEFAULT)


@pytest.mark.parametrize("M,N,K,bias", [(392, 512, 4608, True), (1568, 512, 9216, True), (392, 512, 512, False), (6272, 512, 4608, True),
                                        (25088, 256, 2304, True), (200, 132, 1024, True), (1568, 1536, 512, False), (392, 512, 4672, True)])
def test_gemm_nt_small_grids(M, N, K, bias):
    """Small-grid policy of fm_gemm_nt: a dense bf16 launch whose 256 x 256 tiling leaves more than half of the CUs idle runs on 128 x 128 tiles,
    and with a long reduction as K-slices on gridDim.y + one reduction pass (split-K, fp32 partials in the scratch the caller lends).  Against an
    fp32 matmul of the same bf16 operands; against the launch with the policy off the outputs agree to a bf16 rounding (summation order)."""
    ops, L = _ops()
    x = bf(randn(M + 3, K, seed=60))[:M]
    w = bf(randn(N, K, seed=61) * 0.05)
    b = randn(N, seed=62) if bias else None
    ldo = ops.ru(N, 64)
    ref = x.float() @ w.float().t() + (bf(b).float() if bias else 0.0)
    outs = {}
    for small in (1, 0):
        L.lib.fm_lab_set(9, small)
        try:
            for _ in range(2):
                out = torch.full((M, ldo), 7.0, device=DEV, dtype=torch.bfloat16)
                ops.gemm_nt(x, w, out, bias=b, M=M, N=N, K=K)
        finally:
            L.lib.fm_lab_set(9, 1)
        assert max_err(out[:, :N], bf(ref)) <= 2 ** -7 * float(ref.abs().max()), (small, max_err(out[:, :N], bf(ref)))
        if N < ldo:
            assert bool((out[:, ops.ru(N, 4):] == 7.0).all())          # columns past roundup4(N) untouched
        outs[small] = out[:, :N].float()
    assert rel_err(outs[1], outs[0]) < 2e-3


TN_DEFAULT = 1
TN_MULTI_LISTS = {
    # (R, N, K) per job
    "one_small": [(64, 128, 128)],
    "one_split": [(8192, 768, 768)],                                               # 18 tiles < grid: main + contiguous tails
    "ragged": [(1000, 200, 264), (37, 136, 520), (4100, 768, 136), (300, 64, 64)],  # R % 64 != 0, different R per job
    "rr_tails": [(4096, 2048, 768), (4096, 2048, 768), (4096, 768, 2048), (4096, 768, 768), (4096, 2304, 768)],   # 216 tiles: round-robin tails
    "full_round": [(2048, 2048, 768)] * 5 + [(2048, 768, 2048), (2048, 1536, 768)] + [(2048, 768, 768)] * 3,      # 330 tiles: a full round + a cut
    "t4_ragged_b": [(1024, 768, 2048), (1024, 2560, 1536), (1024, 384, 384), (1024, 128, 768)],   # 256 x 384 tiling: B tiles past K (2048 = 5.33 x 384), A tiles past N
    "t4_short": [(256, 768, 768), (512, 2304, 768)],                                # segments of one or two K-tiles
}


@pytest.mark.parametrize("tile", [256, 128, "lockstep", "t4"])
@pytest.mark.parametrize("name", sorted(TN_MULTI_LISTS))
def test_gemm_tn_multi(name, tile):
    """fm_gemm_tn_multi: every job of the list accumulates dY^T X into its own output, whatever the cut of the tile list over the
    grid (whole tiles, main + tails, round-robin or contiguous); rows past R never count; repeated to screen for races.
    "t4": the 4-wave / 512-register kernel on 256 x 384 tiles (gemm_tn4.hip, opt-in: FOURM_TN4=1) where the list fits its shape rules."""
    ops, L = _ops()
    jobs, refs = [], []
    for i, (R, N, K) in enumerate(TN_MULTI_LISTS[name]):
        Rbuf = ops.ru(R, 128)
        a = bf(randn(Rbuf, N, seed=100 + 2 * i) + torch.arange(N, device=DEV)[None] * 0.01)
        b = bf(randn(Rbuf, K, seed=101 + 2 * i))
        a[R:] = 1e4; b[R:] = float("nan")
        refs.append(1.0 + a[:R].float().t() @ b[:R].float())
        jobs.append((a, b, None, N, K, R))
    L.lib.fm_set_gemm_tn_config({256: 1, 128: 3, "lockstep": 4, "t4": 1}[tile])         # 256 x 256 / K-step 32, 128 x 256, lock-step 256 x 256 / 64
    L.lib.fm_lab_set(5, 1 if tile == "t4" else 0)
    try:
        for _ in range(3):
            outs = [torch.full((N, K), 1.0, device=DEV, dtype=torch.float32) for _, N, K in TN_MULTI_LISTS[name]]
            ops.gemm_tn_multi([(a, b, o, N, K, R) for (a, b, _, N, K, R), o in zip(jobs, outs)])
            for i, (o, ref) in enumerate(zip(outs, refs)):
                assert rel_err(o, ref) < 1e-4, (name, tile, i, rel_err(o, ref))
    finally:
        L.lib.fm_set_gemm_tn_config(TN_DEFAULT)
        L.lib.fm_lab_set(5, 0)


def test_gemm_tn_multi_column_views():
    """Jobs whose dY is a column block of a wider buffer (the SwiGLU (g | u) gradient) and whose output has a wider row stride."""
    ops, L = _ops()
    R, Hp, D = 1024, 256, 192
    dgu = bf(randn(R, 2 * Hp, seed=120))
    x = bf(randn(R, D, seed=121))
    big = torch.zeros(2 * 200, D, device=DEV, dtype=torch.float32)
    ops.gemm_tn_multi([(dgu[:, :Hp], x, big[:200], 200, D, R), (dgu[:, Hp:], x, big[200:], 200, D, R)])
    ref = torch.cat([dgu[:, :200].float().t() @ x.float(), dgu[:, Hp:Hp + 200].float().t() @ x.float()])
    assert rel_err(big, ref) < 1e-4, rel_err(big, ref)


def test_gemm_tn_identity_layout():
    ops, L = _ops()
    R = 128
    a = torch.eye(R, device=DEV, dtype=torch.bfloat16)                 # (R, N=128)
    b = bf((torch.arange(R * 256, device=DEV).reshape(R, 256) % 251 - 125).float())
    for tr in (0, 1):
        out = torch.zeros(R, 256, device=DEV, dtype=torch.float32)
        ops.gemm_tn(a, b, out, force_tr=tr, splits=1)
        assert torch.equal(out, b.float()), f"tr={tr}"


@pytest.mark.parametrize("vocabs,counts", [([512, 192, 1000, 64], [700, 0, 300, 5]), ([16384, 8192, 4096], [2000, 1500, 90])])
def test_gemm_nt_heads_dense_matches_grouped(vocabs, counts):
    """One dense gemm_nt3 launch per head with the row range read from device memory (fm_gemm_nt m_dev / row0_dev, ABI 7) against the
    grouped launch on the same segmented rows: the logits of every head's segment (pad rows included: zero), nothing outside it; an
    empty head (no rows) launches workgroups that leave at once."""
    ops, L = _ops()
    D = 192
    n_heads, R = len(vocabs), sum(counts) + 40
    head = torch.full((R,), -1, dtype=torch.int32)
    idx = torch.randperm(R, generator=torch.Generator().manual_seed(5))
    o = 0
    for h, c in enumerate(counts):
        head[idx[o:o + c]] = h
        o += c
    head = head.to(DEV)
    Rp = ops.padded_rows(R, n_heads)
    seg_start = torch.zeros(n_heads, dtype=torch.int32, device=DEV)
    seg_count = torch.zeros_like(seg_start)
    perm = torch.zeros(Rp, dtype=torch.int32, device=DEV)
    r2p = torch.zeros(R, dtype=torch.int32, device=DEV)
    tile_group = torch.zeros(Rp // ops.SEG, dtype=torch.int32, device=DEV)
    ops.segment_rows(head, n_heads, seg_start, seg_count, perm, r2p, tile_group)
    y = bf(randn(R, D, seed=112))
    yp = torch.zeros(Rp, D, device=DEV, dtype=torch.bfloat16)
    ops.gather_rows(y, perm, yp, D)
    yp[perm < 0] = 0
    ws = [bf(randn(v, D, seed=120 + i) * 0.3) for i, v in enumerate(vocabs)]
    ldl = ops.ru(max(vocabs), 64)
    want = torch.zeros(Rp, ldl, device=DEV, dtype=torch.bfloat16)
    groups = ops.make_groups([dict(W=w, N=v, K=D, ldw=D) for w, v in zip(ws, vocabs)], DEV)
    ops.gemm_nt_grouped(yp, groups, tile_group, want, max(vocabs), max_K=D)
    got = torch.full((Rp, ldl), 7.0, device=DEV, dtype=torch.bfloat16)
    assert ops.heads_dense_ok(ws, vocabs, got)
    ops.gemm_nt_heads(yp, ws, vocabs, seg_start, seg_count, got, D)
    covered = torch.zeros(Rp, ldl, dtype=torch.bool, device=DEV)
    for h, (v, c) in enumerate(zip(vocabs, counts)):
        s, n = int(seg_start[h]), ops.ru(c, ops.SEG)
        covered[s:s + n, :v] = True
        if c:
            ref = y[torch.nonzero(head == h).flatten()].float() @ ws[h].float().t()
            assert rel_err(got[s:s + c, :v], ref) < 4e-3
            assert float((got[s:s + n, :v].float() - want[s:s + n, :v].float()).abs().max()) <= 2 ** -7 * float(ref.abs().max())
            assert float(got[s + c:s + n, :v].float().abs().max() if n > c else 0.0) == 0.0      # pad rows: zero inputs
    assert bool((got[~covered] == 7.0).all())                                    # nothing written outside the heads' segments


def test_gemm_grouped():
    ops, L = _ops()
    D = 128
    vocabs = [320, 200, 64]
    counts = [130, 0, 77]
    n_heads = len(vocabs)
    R = 256
    head = torch.full((R,), -1, dtype=torch.int32)
    idx = torch.randperm(R, generator=torch.Generator().manual_seed(0))
    o = 0
    for h, c in enumerate(counts):
        head[idx[o:o + c]] = h
        o += c
    head = head.to(DEV)
    Rp = ops.padded_rows(R, n_heads)
    seg_start = torch.zeros(n_heads, dtype=torch.int32, device=DEV)
    seg_count = torch.zeros_like(seg_start)
    perm = torch.zeros(Rp, dtype=torch.int32, device=DEV)
    r2p = torch.zeros(R, dtype=torch.int32, device=DEV)
    tile_group = torch.zeros(Rp // ops.SEG, dtype=torch.int32, device=DEV)
    ops.segment_rows(head, n_heads, seg_start, seg_count, perm, r2p, tile_group)
    assert seg_count.tolist() == counts
    assert seg_start.tolist() == [0, 256, 256]
    for h in range(n_heads):
        rows = torch.nonzero(head == h).flatten()
        s = int(seg_start[h])
        assert torch.equal(perm[s:s + len(rows)].long(), rows)          # stable order
        assert torch.equal(r2p[rows].long(), torch.arange(s, s + len(rows), device=DEV))
    assert int((perm >= 0).sum()) == sum(counts)
    y = bf(randn(R, D, seed=12))
    yp = torch.full((Rp, D), 5.0, device=DEV, dtype=torch.bfloat16)
    ops.gather_rows(y, perm, yp, D)
    ws = [bf(randn(v, D, seed=20 + i) * 0.3) for i, v in enumerate(vocabs)]
    ldl = ops.ru(max(vocabs), 64)
    logits = torch.zeros(Rp, ldl, device=DEV, dtype=torch.bfloat16)
    groups = ops.make_groups([dict(W=w, N=v, K=D, ldw=D) for w, v in zip(ws, vocabs)], DEV)
    ops.gemm_nt_grouped(yp, groups, tile_group, logits, max(vocabs))
    for h in range(n_heads):
        rows = torch.nonzero(head == h).flatten()
        s = int(seg_start[h])
        if len(rows):
            ref = y[rows].float() @ ws[h].float().t()
            assert rel_err(logits[s:s + len(rows), :vocabs[h]], ref) < 4e-3
    # cross entropy on the segmented logits
    tgt = torch.stack([torch.randint(0, 64, (R,), generator=torch.Generator().manual_seed(3))]).flatten().to(DEV)
    vocab_t = torch.tensor(vocabs, dtype=torch.int32, device=DEV)
    row_loss = torch.zeros(Rp, device=DEV); head_loss = torch.zeros(n_heads, device=DEV); total = torch.zeros(1, device=DEV)
    row_lse = torch.zeros(Rp, device=DEV)
    saved = logits.clone()
    for loss_type, name in ((L.LOSS_MOD, "mod"), (L.LOSS_TOKEN, "token")):
        logits.copy_(saved)
        gs = torch.tensor([0.5], device=DEV)
        ops.cross_entropy(logits, perm, tile_group, tgt, vocab_t, seg_start, seg_count, n_heads, max(vocabs), row_loss, row_lse, head_loss,
                          total, loss_type=loss_type)
        ops.cross_entropy(logits, perm, tile_group, tgt, vocab_t, seg_start, seg_count, n_heads, max(vocabs), row_loss, row_lse, head_loss,
                          total, loss_type=loss_type, grad_scale=gs, write_grad=True)
        leafs, losses, numel = [], [], []
        for h in range(n_heads):
            rows = torch.nonzero(head == h).flatten()
            s = int(seg_start[h])
            lg = saved[s:s + len(rows), :vocabs[h]].float().clone().requires_grad_(True)
            leafs.append(lg)
            if len(rows):
                losses.append(torch.nn.functional.cross_entropy(lg, tgt[rows])); numel.append(lg.numel())
            else:
                losses.append(torch.zeros((), device=DEV)); numel.append(0)
        if name == "mod":
            tot = sum(losses) / n_heads
        else:
            tot = sum(l * n for l, n in zip(losses, numel)) / sum(numel)
        (tot * 0.5).backward()
        assert abs(float(total) - float(tot)) < 2e-5 * max(1.0, abs(float(tot)))
        for h in range(n_heads):
            assert abs(float(head_loss[h]) - float(losses[h])) < 2e-5 * max(1.0, float(losses[h]))
            rows = torch.nonzero(head == h).flatten()
            s = int(seg_start[h])
            if len(rows):
                assert rel_err(logits[s:s + len(rows), :vocabs[h]], leafs[h].grad) < 6e-3
                pad = logits[s + len(rows):s + ops.ru(len(rows), 128)]
                assert float(pad[:, :ops.ru(vocabs[h], 64)].float().abs().max()) == 0.0 if pad.numel() else True
    # grouped TN: per-head weight gradient
    dws = [torch.zeros(v, D, device=DEV) for v in vocabs]
    groups_tn = ops.make_groups([dict(out=dw, N=v) for dw, v in zip(dws, vocabs)], DEV)
    for tr in (0, 1):
        for dw in dws:
            dw.zero_()
        ops.gemm_tn_grouped(logits, yp, groups_tn, seg_start, seg_count, n_heads, max(vocabs), Rp, D, force_tr=tr)
        for h in range(n_heads):
            rows = torch.nonzero(head == h).flatten()
            s = int(seg_start[h])
            ref = logits[s:s + len(rows), :vocabs[h]].float().t() @ y[rows].float() if len(rows) else torch.zeros_like(dws[h])
            assert rel_err(dws[h], ref) < 1e-4 or float(ref.norm()) == 0.0, (tr, h)


# ------------------------------------------------------------------------------------------------
# LayerNorm
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("R,D", [(64, 128), (300, 384), (512, 768), (33, 2048)])
def test_layernorm(R, D):
    ops, L = _ops()
    x = randn(R, D, seed=30) * 2 + 0.5
    w, b = randn(D, seed=31) * 0.2 + 1, randn(D, seed=32) * 0.1
    y = torch.zeros(R, D, device=DEV, dtype=torch.bfloat16)
    mean, rstd = torch.zeros(R, device=DEV), torch.zeros(R, device=DEV)
    ops.layernorm_fwd(x, w, b, y, mean, rstd)
    ref = torch.nn.functional.layer_norm(x, (D,), w, b, eps=1e-6)
    assert max_err(y, bf(ref)) <= 2 ** -7 * float(ref.abs().max())
    y32 = torch.zeros(R, D, device=DEV)
    ops.layernorm_fwd(x, w, None, y32)
    assert max_err(y32, torch.nn.functional.layer_norm(x, (D,), w, None, eps=1e-6)) < 2e-5
    # backward
    dy = bf(randn(R, D, seed=33))
    dres = randn(R, D, seed=34)
    dx = torch.zeros(R, D, device=DEV); dxb = torch.zeros(R, D, device=DEV, dtype=torch.bfloat16)
    dw, db = torch.ones(D, device=DEV), torch.ones(D, device=DEV)
    ops.layernorm_bwd(dy, x, w, mean, rstd, dx, dres=dres, dx_bf16=dxb, dw=dw, db=db)
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (D,), wr, br, eps=1e-6).backward(dy.float())
    assert rel_err(dx, xr.grad + dres) < 1e-5
    assert rel_err(dxb, xr.grad + dres) < 4e-3
    assert rel_err(dw - 1, wr.grad) < 1e-4 and rel_err(db - 1, br.grad) < 1e-4
    # in-place accumulation into dx
    acc = dres.clone()
    ops.layernorm_bwd(dy, x, w, mean, rstd, acc, dres=acc)
    assert rel_err(acc, xr.grad + dres) < 1e-5


@pytest.mark.parametrize("R,D", [(300, 768), (129, 1024), (64, 384), (17, 2048)])
def test_layernorm_bwd_from_saved_output(R, D):
    """fm_layernorm_bwd_h: x_hat rebuilt from the saved bf16 norm output h instead of the fp32 input (bias-free norm).  Exact against
    the restatement with x_hat = float(h) / w; against the fp32 norm backward within h's bf16 rounding; weights of magnitude 0 (x_hat
    not recoverable from h) take the fp32 input for their chunk."""
    ops, L = _ops()
    x = randn(R, D, seed=35) * 2 + 0.5
    w = randn(D, seed=36) * 0.2 + 1
    w[5] = 0.0; w[D - 3] = 1e-30          # chunks 1 and D / 4 - 1 fall back to x
    h = torch.zeros(R, D, device=DEV, dtype=torch.bfloat16)
    mean, rstd = torch.zeros(R, device=DEV), torch.zeros(R, device=DEV)
    ops.layernorm_fwd(x, w, None, h, mean, rstd)
    dy, dres = bf(randn(R, D, seed=37)), randn(R, D, seed=38)
    dx = torch.zeros(R, D, device=DEV); dxb = torch.zeros(R, D, device=DEV, dtype=torch.bfloat16)
    dw = torch.ones(D, device=DEV)
    ops.layernorm_bwd(dy, x, w, mean, rstd, dx, dres=dres, dx_bf16=dxb, dw=dw, h=h)
    # restatement: x_hat from h where every weight of the 4-column chunk is usable, from x elsewhere
    xh_x = (x - mean[:, None]) * rstd[:, None]
    usable = (w.abs().view(-1, 4).min(dim=1).values >= 1e-20).repeat_interleave(4)
    xh = torch.where(usable[None, :], h.float() / torch.where(usable, w, torch.ones_like(w))[None, :], xh_x)
    g = dy.float() * w
    want = rstd[:, None] * (g - g.mean(1, keepdim=True) - xh * (g * xh).mean(1, keepdim=True)) + dres
    assert rel_err(dx, want) < 1e-5
    assert rel_err(dw - 1, (dy.float() * xh).sum(0)) < 1e-4
    assert rel_err(dxb, want) < 4e-3
    # against the fp32 norm backward: within the rounding h carries
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (D,), wr, None, eps=1e-6).backward(dy.float())
    assert rel_err(dx, xr.grad + dres) < 4e-3
    assert rel_err(dw - 1, wr.grad) < 4e-3
    # in place (dres aliases dx), no bf16 copy, no weight gradient
    acc = dres.clone()
    ops.layernorm_bwd(dy, x, w, mean, rstd, acc, dres=acc, h=h)
    assert rel_err(acc, want) < 1e-5


@pytest.mark.parametrize("R,D", [(300, 768), (129, 1024), (64, 384), (17, 2048)])
def test_layernorm_with_residual_add(R, D):
    """fm_layernorm_fwd_res: x_out = x + delta (bf16) and y = LN(x_out) in one pass - bit-identical to the residual epilogue's sum
    followed by fm_layernorm_fwd (the engine moves the add from the GEMM epilogue into the norm); row_map honoured; fm_add_bf16_f32."""
    ops, L = _ops()
    x = randn(R, D, seed=40) * 2 + 0.5
    delta = bf(randn(R, D, seed=41))
    w, b = randn(D, seed=42) * 0.2 + 1, randn(D, seed=43) * 0.1
    xo = torch.full((R, D), 7.0, device=DEV)
    y = torch.zeros(R, D, device=DEV, dtype=torch.bfloat16)
    mean, rstd = torch.zeros(R, device=DEV), torch.zeros(R, device=DEV)
    ops.layernorm_fwd(x, w, b, y, mean, rstd, delta=delta, x_out=xo)
    want = x + delta.float()
    assert torch.equal(xo, want)
    y2 = torch.zeros_like(y); m2, r2 = torch.zeros_like(mean), torch.zeros_like(rstd)
    ops.layernorm_fwd(want, w, b, y2, m2, r2)
    assert torch.equal(y, y2) and torch.equal(mean, m2) and torch.equal(rstd, r2)
    out = torch.zeros(R, D, device=DEV)
    ops.add_bf16_to_f32(x, delta, out)
    assert torch.equal(out, want)
    # scattered rows (the decoder norm writes into the head-segmented layout)
    perm = torch.randperm(R, device=DEV).int()
    perm[::7] = -1
    ys = torch.zeros(R, D, device=DEV, dtype=torch.bfloat16)
    ops.layernorm_fwd(x, w, None, ys, row_map=perm, delta=delta, x_out=xo)
    yr = torch.zeros(R, D, device=DEV, dtype=torch.bfloat16)
    ops.layernorm_fwd(want, w, None, yr, row_map=perm)
    assert torch.equal(ys, yr)


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
NEG = -torch.finfo(torch.bfloat16).max


def ref_attention(q, k, v, blocked, scale):
    """fp32 restatement with upstream's rounding points (bf16 scores, fp32 softmax, bf16 probabilities)."""
    s = bf(bf(q.float() @ k.float().transpose(-1, -2)).float() * scale).float()
    if blocked is not None:
        s = s.masked_fill(blocked, NEG)
    p = torch.softmax(s, -1)
    return bf(p).float() @ v.float(), p


def make_masks(kind, B, Nq, Nk, seed):
    g = torch.Generator().manual_seed(seed)
    out = dict(kpad=None, cs=None, modq=None, modk=None, dense=None, blocked=None)
    if kind == "keypad":
        kp = torch.rand(B, Nk, generator=g) < 0.3
        kp[0] = True                      # one fully blocked sample: uniform attention rows
        out["kpad"] = kp.to(DEV)
        out["blocked"] = kp[:, None, None, :].to(DEV)
    elif kind == "decoder":
        assert Nq == Nk
        dam = (torch.rand(B, Nq, generator=g) < 0.5).int() * torch.randint(1, 4, (B, Nq), generator=g).int()
        dam[:, 0] = 0                     # rows before the first non-zero entry are fully blocked
        cs = dam.cumsum(-1).int()
        mod = torch.randint(0, 3, (B, Nq), generator=g).short().sort(-1).values
        blk = (torch.arange(Nk)[None, None, :] >= cs[:, :, None]) | (mod[:, :, None] != mod[:, None, :])
        out.update(cs=cs.to(DEV), modq=mod.to(DEV), modk=mod.to(DEV), blocked=blk[:, None].to(DEV))
    elif kind == "dense":
        d = torch.rand(B, Nq, Nk, generator=g) < 0.4
        out.update(dense=d.to(DEV), blocked=d[:, None].to(DEV))
    return out


@pytest.mark.parametrize("tr", [0, 1])
@pytest.mark.parametrize("kind,Nq,Nk", [("none", 128, 128), ("keypad", 128, 128), ("decoder", 128, 128), ("dense", 96, 160),
                                        ("keypad", 40, 200), ("none", 196, 196), ("decoder", 256, 256),
                                        # ragged query / key counts
                                        ("dense", 200, 100), ("keypad", 70, 50), ("decoder", 96, 96), ("none", 33, 128),
                                        # beyond 256 tokens: whole sequence in LDS (<= 512 rows), then the chunked backward
                                        # (upstream trains 1024 + 1024-token configs; registers push the encoder past 256)
                                        ("keypad", 260, 260), ("decoder", 512, 512), ("dense", 500, 384), ("decoder", 1024, 1024), ("keypad", 600, 1030),
                                        ("dense", 530, 70)])
def test_attention(kind, Nq, Nk, tr):
    ops, L = _ops()
    B, H = 3, 2
    D = H * 64
    scale = 64 ** -0.5
    qkv = bf(randn(B * Nq, 3 * D, seed=40) * 1.5)
    if Nq == Nk:
        q2, k2, v2 = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    else:
        q2 = qkv[:, :D]
        kv = bf(randn(B * Nk, 2 * D, seed=41) * 1.5)
        k2, v2 = kv[:, :D], kv[:, D:]
    mk = make_masks(kind, B, Nq, Nk, seed=42)
    kinds = dict(none=L.MASK_NONE, keypad=L.MASK_KEYPAD, decoder=L.MASK_DECODER, dense=L.MASK_DENSE)
    o = torch.zeros(B * Nq, D, device=DEV, dtype=torch.bfloat16)
    sm, sl = torch.zeros(B, H, Nq, device=DEV), torch.zeros(B, H, Nq, device=DEV)
    ops.attn_fwd(q2, k2, v2, o, B, H, Nq, Nk, scale, mask_kind=kinds[kind], kpad=mk["kpad"], cs=mk["cs"], modq=mk["modq"],
                 modk=mk["modk"], dense=mk["dense"], stat_m=sm, stat_l=sl, force_tr=tr)
    qh = q2.reshape(B, Nq, H, 64).transpose(1, 2).float().requires_grad_(True)
    kh = k2.reshape(B, Nk, H, 64).transpose(1, 2).float().requires_grad_(True)
    vh = v2.reshape(B, Nk, H, 64).transpose(1, 2).float().requires_grad_(True)
    ref, p = ref_attention(qh, kh, vh, mk["blocked"], scale)
    oh = o.reshape(B, Nq, H, 64).transpose(1, 2).float()
    # (long key sequences: p = 1/Nk in fully blocked rows rounds to bf16 with a systematic 2^-9 relative error)
    assert rel_err(oh, ref) < (8e-3 if Nk <= 512 else 1.2e-2), (kind, tr, rel_err(oh, ref))
    assert max_err(oh, ref) < 0.06
    # backward (straight-through the rounding points, like autograd upstream)
    do = bf(randn(B * Nq, D, seed=43))
    s = (qh @ kh.transpose(-1, -2)) * scale
    if mk["blocked"] is not None:
        s = s.masked_fill(mk["blocked"], NEG)
    (torch.softmax(s, -1) @ vh).backward(do.reshape(B, Nq, H, 64).transpose(1, 2).float())
    dq = torch.zeros(B * Nq, D, device=DEV, dtype=torch.bfloat16)
    dk = torch.zeros(B * Nk, D, device=DEV, dtype=torch.bfloat16)
    dv = torch.zeros(B * Nk, D, device=DEV, dtype=torch.bfloat16)
    ops.attn_bwd(q2, k2, v2, o, do, dq, dk, dv, B, H, Nq, Nk, scale, sm, sl, mask_kind=kinds[kind], kpad=mk["kpad"], cs=mk["cs"],
                 modq=mk["modq"], modk=mk["modk"], dense=mk["dense"], force_tr=tr)
    for name, got, want, n in (("dq", dq, qh.grad, Nq), ("dk", dk, kh.grad, Nk), ("dv", dv, vh.grad, Nk)):
        got = got.reshape(B, n, H, 64).transpose(1, 2).float()
        assert rel_err(got, want) < 2e-2, (kind, tr, name, rel_err(got, want))


@pytest.mark.parametrize("kind", ["none", "keypad", "decoder", "causal"])
@pytest.mark.parametrize("zero_attn", [False, True])
@pytest.mark.parametrize("Nq,Nk", [(128, 128), (256, 256), (384, 256), (128, 512)])
def test_attention_fwd128_matches_the_general_kernel(kind, zero_attn, Nq, Nk):
    """The forward kernels for whole 128-token tiles - 128 x 128: all keys in LDS at once; multiples of 128 up to 512 keys: the same
    per-score arithmetic in an online softmax over 128-key tiles (mask on the raw scores, one exp2(fma) per score; force_tr=1) - against
    the general online-softmax kernel (force_tr=0 never takes those paths): the same row maxima bit for bit (fully blocked rows included:
    -finfo(bf16).max), row sums to fp32 rounding, outputs to one bf16 rounding."""
    if kind in ("decoder", "causal") and Nq != Nk:
        pytest.skip("the decoder rule is a self-attention mask")
    ops, L = _ops()
    B, H = 5, 3
    D = H * 64
    q2 = bf(randn(B * Nq, D, seed=140) * 1.5)
    kv = bf(randn(B * Nk, 2 * D, seed=142) * 1.5)
    mk = make_masks("none" if kind == "causal" else kind, B, Nq, Nk, seed=141)
    kinds = dict(none=L.MASK_NONE, keypad=L.MASK_KEYPAD, decoder=L.MASK_DECODER, causal=L.MASK_DECODER)
    res = []
    for tr in (0, 1):
        o = torch.zeros(B * Nq, D, device=DEV, dtype=torch.bfloat16)
        sm, sl = torch.zeros(B, H, Nq, device=DEV), torch.zeros(B, H, Nq, device=DEV)
        ops.attn_fwd(q2, kv[:, :D], kv[:, D:], o, B, H, Nq, Nk, 0.125, mask_kind=kinds[kind], kpad=mk["kpad"], cs=mk["cs"],
                     modq=mk["modq"], modk=mk["modk"], stat_m=sm, stat_l=sl, force_tr=tr, zero_attn=zero_attn, causal=kind == "causal")
        res.append((o.float(), sm, sl))
    (o0, m0, l0), (o1, m1, l1) = res
    assert torch.equal(m0, m1)
    if kind == "causal":                                         # (decoder_causal_mask, fm.py:466-468) against the definition itself
        qh, kh, vh = (t.reshape(B, -1, H, 64).transpose(1, 2).float() for t in (q2, kv[:, :D], kv[:, D:]))
        blocked = torch.ones(Nq, Nk, dtype=torch.bool, device=DEV).triu(1)[None, None]
        if not zero_attn:
            ref, _ = ref_attention(qh, kh, vh, blocked, 0.125)
            assert rel_err(o1.reshape(B, Nq, H, 64).transpose(1, 2), ref) < 8e-3
    if kind in ("keypad", "decoder") and not zero_attn:
        assert int((m1 < -1e38).sum()) > 0                       # the fixture holds fully blocked rows
    assert float(((l0 - l1).abs() / l0).max()) < 3e-6
    assert float((o0 - o1).abs().max()) <= 2 ** -7 * float(o0.abs().max()) and rel_err(o1, o0) < 2e-3


@pytest.mark.parametrize("R,H,bias", [(37, 2, True), (1000, 12, False)])
def test_headnorm(R, H, bias):
    """Per-head LayerNorm of q / k (qk_norm models) on a column block of a wider buffer, forward and backward."""
    ops, L = _ops()
    D = 64 * H
    buf = bf(randn(R, 3 * D, seed=90))
    x = buf[:, D:2 * D]                                   # the "k" block of a fused qkv buffer
    w, b = randn(64, seed=91) * 0.2 + 1.0, (randn(64, seed=92) * 0.1 if bias else None)
    y = torch.zeros(R, D, device=DEV, dtype=torch.bfloat16)
    st = torch.zeros(R * H, 2, device=DEV)
    ops.headnorm_fwd(x, w, b, y, st, R, H, 1e-6)
    xr = x.float().reshape(R, H, 64).requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), (b.clone().requires_grad_(True) if bias else None)
    ref = torch.nn.functional.layer_norm(xr, (64,), wr, br, 1e-6)
    assert max_err(y.float().reshape(R, H, 64), ref) < 2e-2 and rel_err(y.float().reshape(R, H, 64), ref) < 4e-3
    dy = bf(randn(R, D, seed=93))
    dx = torch.zeros(R, 3 * D, device=DEV, dtype=torch.bfloat16)
    dw, db = torch.ones(64, device=DEV), torch.ones(64, device=DEV)
    ops.headnorm_bwd(dy, x, w, st, dx[:, :D], dw, db if bias else None, R, H)
    ref.backward(dy.float().reshape(R, H, 64))
    assert rel_err(dx[:, :D].float().reshape(R, H, 64), xr.grad) < 6e-3
    assert rel_err(dw - 1, wr.grad) < 1e-4
    if bias:
        assert rel_err(db - 1, br.grad) < 1e-4
    assert float(dx[:, D:].float().abs().max()) == 0


# ------------------------------------------------------------------------------------------------
# element-wise
# ------------------------------------------------------------------------------------------------
def test_swiglu_gelu_bwd():
    ops, L = _ops()
    R, H = 100, 170
    Hp = ops.ru(H, 64)
    gu = torch.zeros(R, 2 * Hp, device=DEV, dtype=torch.bfloat16)
    g, u = bf(randn(R, H, seed=50)), bf(randn(R, H, seed=51))
    gu[:, :H], gu[:, Hp:Hp + H] = g, u
    da = torch.zeros(R, Hp, device=DEV, dtype=torch.bfloat16); da[:, :H] = bf(randn(R, H, seed=52))
    dgu = torch.full((R, 2 * Hp), 9.0, device=DEV, dtype=torch.bfloat16)
    ops.swiglu_bwd(da, gu, dgu, H, Hp)
    gr, ur = g.float().requires_grad_(True), u.float().requires_grad_(True)
    (torch.nn.functional.silu(gr) * ur).backward(da[:, :H].float())
    assert rel_err(dgu[:, :H], gr.grad) < 8e-3 and rel_err(dgu[:, Hp:Hp + H], ur.grad) < 8e-3
    assert float(dgu[:, H:Hp].float().abs().max()) == 0 and float(dgu[:, Hp + H:].float().abs().max()) == 0
    pre = torch.zeros(R, Hp, device=DEV, dtype=torch.bfloat16); pre[:, :H] = g
    dpre = torch.full((R, Hp), 9.0, device=DEV, dtype=torch.bfloat16)
    ops.gelu_bwd(da, pre, dpre, H, Hp)
    pr = g.float().requires_grad_(True)
    torch.nn.functional.gelu(pr).backward(da[:, :H].float())
    assert rel_err(dpre[:, :H], pr.grad) < 6e-3 and float(dpre[:, H:].float().abs().max()) == 0


@pytest.mark.parametrize("M,H,K", [(100, 170, 64), (300, 2730, 768)])
def test_gemm_nt_activation_backward_epilogues(M, H, K):
    """fc2 dX GEMM with the SwiGLU / GELU backward fused into its epilogue == plain GEMM followed by the
    stand-alone element-wise kernels (bit-exact: same bf16 rounding points)."""
    ops, L = _ops()
    Hp = ops.ru(H, 64)
    gy = bf(randn(M, K, scale=0.5, seed=60))
    w2t = bf(randn(H, K, scale=0.1, seed=61))            # fc2.weight transposed: (hidden, D)
    gu = torch.zeros(M, 2 * Hp, device=DEV, dtype=torch.bfloat16)
    gu[:, :H], gu[:, Hp:Hp + H] = bf(randn(M, H, seed=62)), bf(randn(M, H, seed=63))
    da = torch.zeros(M, Hp, device=DEV, dtype=torch.bfloat16)
    ops.gemm_nt(gy, w2t, da, N=H, K=K)
    want = torch.zeros(M, 2 * Hp, device=DEV, dtype=torch.bfloat16)
    ops.swiglu_bwd(da, gu, want, H, Hp)
    got = torch.zeros(M, 2 * Hp, device=DEV, dtype=torch.bfloat16)
    ops.gemm_nt(gy, w2t, got, N=H, K=K, epilogue=L.EPI_SWIGLU_BWD, res=gu, Hp=Hp)
    assert torch.equal(got, want)
    pre = gu[:, :Hp].contiguous()
    want_p = torch.zeros(M, Hp, device=DEV, dtype=torch.bfloat16)
    ops.gelu_bwd(da, pre, want_p, H, Hp)
    got_p = torch.zeros(M, Hp, device=DEV, dtype=torch.bfloat16)
    ops.gemm_nt(gy, w2t, got_p, N=H, K=K, epilogue=L.EPI_GELU_BWD, res=pre)
    assert torch.equal(got_p, want_p)


def test_weight_shadows_and_colsum():
    ops, L = _ops()
    w = randn(170, 128, seed=60)
    a = torch.full((170, 192), 4.0, device=DEV, dtype=torch.bfloat16)
    ops.cast_pad(w, a)
    assert torch.equal(a[:, :128], bf(w)) and float(a[:, 128:].float().abs().max()) == 0
    t = torch.full((128, 192), 4.0, device=DEV, dtype=torch.bfloat16)
    ops.transpose_cast_pad(w, t)
    assert torch.equal(t[:, :170], bf(w.t())) and float(t[:, 170:].float().abs().max()) == 0
    dy = bf(randn(300, 200, seed=61))
    db = torch.ones(200, device=DEV)
    ops.colsum(dy, db, 200)
    assert rel_err(db - 1, dy.float().sum(0)) < 1e-5


def test_shadow_refresh_multi_tensor():
    """One launch, many (fp32 master -> bf16 shadow) jobs: plain casts, transposes, a transposed pair written into
    column slices of one buffer, ragged shapes; padding must stay untouched."""
    ops, L = _ops()
    flat = randn(170 * 128 + 3 * 77 + 256 * 192 + 2 * 64 * 130 + 8, seed=80)
    off = 0
    def take(r, c):
        nonlocal off
        v = flat[off:off + r * c].view(r, c); off += r * c
        return v
    a, b, c_, d1, d3 = take(170, 128), take(3, 77), take(256, 192), take(64, 130), take(64, 130)
    wa = torch.zeros(170, 128, device=DEV, dtype=torch.bfloat16)
    wb = torch.full((3, 128), 7.0, device=DEV, dtype=torch.bfloat16)
    wat = torch.full((128, 192), 7.0, device=DEV, dtype=torch.bfloat16)
    wct = torch.zeros(192, 256, device=DEV, dtype=torch.bfloat16)
    pair = torch.full((130, 2 * 64), 7.0, device=DEV, dtype=torch.bfloat16)
    jobs = [(a, wa, False), (b, wb, False), (a, wat, True), (c_, wct, True), (d1, pair[:, :64], True), (d3, pair[:, 64:], True)]
    table, tiles = ops.shadow_jobs_table(jobs, DEV)
    ops.shadow_refresh(table, len(jobs), tiles)
    assert torch.equal(wa, bf(a)) and torch.equal(wb[:, :77], bf(b)) and float((wb[:, 77:].float() - 7).abs().max()) == 0
    assert torch.equal(wat[:, :170], bf(a.t())) and float((wat[:, 170:].float() - 7).abs().max()) == 0
    assert torch.equal(wct, bf(c_.t()))
    assert torch.equal(pair[:, :64], bf(d1.t())) and torch.equal(pair[:, 64:], bf(d3.t()))


def test_adamw_matches_torch():
    ops, L = _ops()
    n = 10007
    p0, g = randn(n + 1, seed=70)[:n + 1], randn(n + 1, seed=71)
    p = p0.clone(); m = torch.zeros_like(p); v = torch.zeros_like(p)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    for step in range(1, 4):
        ref.grad = g.clone() * step
        opt.step()
        ops.adamw(p, g * step, m, v, p.numel(), 1e-3, 0.9, 0.95, 1e-8, 0.05, step)
    assert max_err(p, ref.detach()) < 1e-6
    ss = torch.zeros(1, device=DEV); nrm = torch.zeros(1, device=DEV); coef = torch.zeros(1, device=DEV)
    ops.sumsq(g, ss)
    ops.clip_coef(ss, 1.0, nrm, coef)
    assert abs(float(nrm) - float(g.norm())) < 1e-3 * float(g.norm())
    assert abs(float(coef) - min(1.0, 1.0 / (float(g.norm()) + 1e-6))) < 1e-6


@pytest.mark.parametrize("kind,Nq,Nk", [("none", 128, 128), ("keypad", 100, 100), ("decoder", 64, 64), ("keypad", 40, 130), ("keypad", 128, 128),
                                        ("decoder", 128, 128)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_attention_zero_attn(kind, Nq, Nk, dtype):
    """allow_zero_attn (upstream softmax1, fm_utils.py:28-30): p = softmax(pad(scores, one zero logit))[..., :-1], forward and backward,
    on the bf16 kernels and on the fp32 verification kernels; small and large scores (the zero logit dominates / vanishes)."""
    import torch.nn.functional as F
    ops, L = _ops()
    B, H = 2, 2
    D = H * 64
    scale = 64 ** -0.5
    cast = (lambda t: bf(t)) if dtype == torch.bfloat16 else (lambda t: t.float().contiguous())
    # every key shares a component u; queries run from -30 u (all scores ~ -11: the zero logit takes the mass) to +30 u (it vanishes)
    u = torch.nn.functional.normalize(randn(1, 64, seed=59), dim=-1).repeat(1, H)
    q2 = cast(randn(B * Nq, D, seed=60) * 0.5 + torch.linspace(-30.0, 30.0, B * Nq, device=DEV)[:, None] * u)
    kv = randn(B * Nk, 2 * D, seed=61) * 1.5
    kv[:, :D] = kv[:, :D] * 0.2 + 3.0 * u
    kv = cast(kv)
    k2, v2 = kv[:, :D], kv[:, D:]
    mk = make_masks(kind, B, Nq, Nk, seed=62)
    kinds = dict(none=L.MASK_NONE, keypad=L.MASK_KEYPAD, decoder=L.MASK_DECODER)
    o = torch.zeros(B * Nq, D, device=DEV, dtype=dtype)
    sm, sl = torch.zeros(B, H, Nq, device=DEV), torch.zeros(B, H, Nq, device=DEV)
    kw = dict(mask_kind=kinds[kind], kpad=mk["kpad"], cs=mk["cs"], modq=mk["modq"], modk=mk["modk"], zero_attn=True)
    ops.attn_fwd(q2, k2, v2, o, B, H, Nq, Nk, scale, stat_m=sm, stat_l=sl, **kw)
    qh = q2.reshape(B, Nq, H, 64).transpose(1, 2).float().requires_grad_(True)
    kh = k2.reshape(B, Nk, H, 64).transpose(1, 2).float().requires_grad_(True)
    vh = v2.reshape(B, Nk, H, 64).transpose(1, 2).float().requires_grad_(True)
    s = (qh @ kh.transpose(-1, -2)) * scale
    if mk["blocked"] is not None:
        s = s.masked_fill(mk["blocked"], NEG)
    p = torch.softmax(F.pad(s, (0, 1)), -1)[..., :-1]
    ref = p @ vh
    assert float(p.sum(-1).min()) < 0.1 and float(p.sum(-1).max()) > 0.99                  # both regimes are in the data
    oh = o.reshape(B, Nq, H, 64).transpose(1, 2).float()
    tol = 8e-3 if dtype == torch.bfloat16 else 2e-5
    assert rel_err(oh, ref) < tol, rel_err(oh, ref)
    plain = torch.softmax(s, -1) @ vh
    assert rel_err(plain, ref) > 10 * tol                                                  # the flag matters here
    do = cast(randn(B * Nq, D, seed=63))
    ref.backward(do.reshape(B, Nq, H, 64).transpose(1, 2).float())
    dq, dk, dv = (torch.zeros(B * n, D, device=DEV, dtype=dtype) for n in (Nq, Nk, Nk))
    ops.attn_bwd(q2, k2, v2, o, do, dq, dk, dv, B, H, Nq, Nk, scale, sm, sl, **kw)
    for name, got, want, n in (("dq", dq, qh.grad, Nq), ("dk", dk, kh.grad, Nk), ("dv", dv, vh.grad, Nk)):
        got = got.reshape(B, n, H, 64).transpose(1, 2).float()
        assert rel_err(got, want) < (2e-2 if dtype == torch.bfloat16 else 5e-5), (name, rel_err(got, want))
