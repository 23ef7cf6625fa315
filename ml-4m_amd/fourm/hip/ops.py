"""Thin torch-tensor front end over the C ABI: pointer extraction, shape checks, stream plumbing.
torch is used here for device memory and streams only; every computation is a HIP kernel."""
import ctypes as C
from typing import Optional

import torch

from . import _lib as L

# Optional per-launch timing (bench.py): an object with .launch(name, flops, bytes) -> context manager
# that brackets the launch with events on the CURRENT stream (the one the kernels are enqueued on).
_PROFILER = None


def set_profiler(p):
    global _PROFILER
    _PROFILER = p


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NULL = _Null()


def _prof(name, flops=0.0, nbytes=0.0, tag=""):
    return _NULL if _PROFILER is None else _PROFILER.launch(name, flops, nbytes, tag)


def _timed(name):
    """Decorator: the wrapped launch shows up as family ``name`` in bench.py's per-kernel breakdown (events only while a profiler is set)."""
    def deco(fn):
        import functools

        @functools.wraps(fn)
        def wrapped(*a, **k):
            if _PROFILER is None:
                return fn(*a, **k)
            with _PROFILER.launch(name, 0.0, 0.0, ""):
                return fn(*a, **k)
        return wrapped
    return deco


BK = 64          # reduction granularity of the GEMMs (zero padded)
SEG = 256        # row-segment alignment of the grouped head GEMMs (FM_SEG_ROWS)


def ru(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libfourm_hip ops need device tensors (no CPU fallback exists)")
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ld(t: torch.Tensor) -> int:
    assert t.dim() == 2 and t.stride(1) == 1, "2-D row-major tensor expected"
    return t.stride(0)


# ---------------------------------------------------------------------------------------------
# GEMM
# ---------------------------------------------------------------------------------------------
# split-K scratch of fm_gemm_nt (fm_gemm_nt_args.splitk_ws): one fp32 buffer per device, shared by every launch (stream order)
SPLITK_MAX_OUT = 128 * 128 * 128          # outputs of up to 128 tiles of 128 x 128 can be split (the kernel decides)
_SPLITK_WS = {}


SCRATCH_TAG = None          # set while a hipGraph is being captured: scratch allocated then belongs to THAT graph (torch captures every graph on the same
                            # side stream, so the stream alone would hand one graph's scratch to the next - and two graphs may replay concurrently)


def _splitk_ws(device):
    key = (device, torch.cuda.current_stream(device).cuda_stream, SCRATCH_TAG)          # (launches on different streams must not share partial tiles)
    t = _SPLITK_WS.get(key)
    if t is None:
        t = _SPLITK_WS[key] = torch.empty(16 * SPLITK_MAX_OUT, dtype=torch.float32, device=device)      # 128 MB: 16 slices of the largest output
    return t


def gemm_nt(x, w, out, *, epilogue=L.EPI_BF16, bias=None, res=None, w2=None, bias2=None, out2=None, Hp=0, N=None, K=None, M=None, conv=None):
    """out[m][n] = sum_k x[m][k] w[n][k] (+ epilogue).  x: bf16 (M, >=K); w: bf16 (N, >=K).
    fp32 operands select the verification kernel (any strides for w).
    conv = dict(C, H, W, Ho, Wo, stride, up): x is a (B, H >> up, W >> up, C) feature map in rows and the launch is the 3 x 3 convolution
    with the (N, 9 C) weight rows w (taps outermost) as an implicit GEMM (fm_gemm_nt_args.conv_*); M = B Ho Wo and K = 9 C must be given."""
    if x.dtype == torch.float32:
        return _gemm_f32(x, w, out, epilogue=epilogue, bias=bias, res=res, w2=w2, bias2=bias2, out2=out2, Hp=Hp, N=N, K=K, M=M)
    a = L.GemmNTArgs()
    a.W, a.W2, a.X, a.out, a.out2 = _p(w), _p(w2), _p(x), _p(out), _p(out2)
    a.res, a.bias, a.bias2 = _p(res), _p(bias), _p(bias2)
    a.M = x.shape[0] if M is None else M
    a.N = w.shape[0] if N is None else N
    a.K = w.shape[1] if K is None else K
    a.ldw, a.ldx, a.ldo = _ld(w), _ld(x), _ld(out)
    a.ldo2 = _ld(out2) if out2 is not None else 0
    a.ldr = _ld(res) if res is not None else 0
    a.Hp, a.epilogue = Hp, epilogue
    if conv is not None:
        a.conv_C, a.conv_H, a.conv_W, a.conv_Ho, a.conv_Wo = conv["C"], conv["H"], conv["W"], conv["Ho"], conv["Wo"]
        a.conv_stride, a.conv_up = conv.get("stride", 1), conv.get("up", 0)
    if epilogue == L.EPI_BF16 and a.M * a.N <= SPLITK_MAX_OUT and a.K >= 512:       # (only launches that can be split need the scratch)
        ws = _splitk_ws(x.device)
        a.splitk_ws, a.splitk_ws_bytes = ws.data_ptr(), ws.numel() * 4
    two = 2 if epilogue == L.EPI_SWIGLU else 1
    nbytes = 2.0 * a.K * (a.M + two * a.N) + a.M * a.N * two * out.element_size()        # operands once + output once
    nbytes += (a.M * a.N * 4.0 if res is not None and epilogue in (L.EPI_RESIDUAL, L.EPI_F32) else 0.0) + (a.M * a.N * 4.0 if out2 is not None else 0.0)
    with _prof(f"gemm_nt/epi{epilogue}", 2.0 * a.M * a.N * a.K * two, nbytes, tag=f"M{a.M} N{a.N} K{a.K}"):
        L.check(L.gemm_nt(C.byref(a), _stream()))
    return out


def _gemm_f32(x, w, out, *, epilogue=L.EPI_BF16, bias=None, res=None, w2=None, bias2=None, out2=None, Hp=0, N=None, K=None, M=None,
              accumulate=False):
    assert w.dtype == torch.float32 and out.dtype == torch.float32 and x.stride(1) == 1 and out.stride(1) == 1
    a = L.GemmF32Args()
    a.X, a.W, a.W2, a.out, a.out2, a.res, a.bias, a.bias2 = _p(x), _p(w), _p(w2), _p(out), _p(out2), _p(res), _p(bias), _p(bias2)
    a.M = x.shape[0] if M is None else M
    a.N = w.shape[0] if N is None else N
    a.K = min(w.shape[1] if K is None else K, w.shape[1])          # padded reduction widths (Hp) stop at the weight's own width
    a.sxm, a.sxk, a.swn, a.swk = x.stride(0), 1, w.stride(0), w.stride(1)
    if w2 is not None:
        assert w2.stride() == w.stride()
    a.ldo, a.ldo2, a.ldr = out.stride(0), out2.stride(0) if out2 is not None else 0, res.stride(0) if res is not None else 0
    a.Hp, a.epilogue, a.accumulate = Hp, epilogue, 1 if accumulate else 0
    with _prof("gemm_f32", 2.0 * a.M * a.N * a.K * (2 if epilogue == L.EPI_SWIGLU else 1), 4.0 * (a.K * (a.M + a.N) + a.M * a.N), tag=f"M{a.M} N{a.N} K{a.K}"):
        L.check(L.gemm_f32(C.byref(a), _stream()))
    return out


@_timed("heads_gemm_nt (logits, dY)")
def gemm_nt_grouped(x, groups, tile_group, out, max_N, M=None, max_K=0):
    if x.dtype == torch.float32:
        a = L.GemmF32Args()
        a.X, a.out, a.groups, a.tile_group = _p(x), _p(out), _p(groups), _p(tile_group)
        a.M, a.N, a.max_N, a.seg_rows = (x.shape[0] if M is None else M), max_N, max_N, SEG
        a.sxm, a.sxk, a.ldo, a.epilogue = x.stride(0), 1, out.stride(0), L.EPI_BF16
        L.check(L.gemm_f32(C.byref(a), _stream()))
        return out
    a = L.GemmNTArgs()
    a.X, a.out = _p(x), _p(out)
    a.M = x.shape[0] if M is None else M
    a.K = max_K
    a.ldx, a.ldo = _ld(x), _ld(out)
    a.epilogue = L.EPI_BF16
    a.groups, a.tile_group, a.max_N = _p(groups), _p(tile_group), max_N
    L.check(L.gemm_nt(C.byref(a), _stream()))
    return out


def heads_dense_ok(ws, vocabs, out):
    """One dense fm_gemm_nt launch per head (row range read from device memory) can replace the grouped logits GEMM: bf16, every
    vocabulary a multiple of 8, whole-line output rows (the staged epilogue of gemm_nt3)."""
    return (out.dtype == torch.bfloat16 and all(v % 8 == 0 for v in vocabs) and all(w.dtype == torch.bfloat16 and w.shape[1] % 64 == 0 for w in ws)
            and _ld(out) % 64 == 0 and out.data_ptr() % 128 == 0)


@_timed("heads_gemm_nt (logits, dY)")
def gemm_nt_heads(x, ws, vocabs, seg_start, seg_count, out, K):
    """The logits of every modality head as ONE DENSE launch per head on the lock-step kernel (csrc/gemm_nt3.hip, DEVM): head h covers rows
    [seg_start[h], seg_start[h] + roundup(seg_count[h], SEG)) of x / out - read by the kernel from device memory, so the step stays
    capturable - against its own (V_h, K) weight.  Same outputs as gemm_nt_grouped on the rows of the segments (pad rows of x are zero)."""
    for h, (w, v) in enumerate(zip(ws, vocabs)):
        a = L.GemmNTArgs()
        a.W, a.X, a.out = _p(w), _p(x), _p(out)
        a.M, a.N, a.K = x.shape[0], v, K
        a.ldw, a.ldx, a.ldo = _ld(w), _ld(x), _ld(out)
        a.epilogue = L.EPI_BF16
        a.m_dev, a.row0_dev = seg_count.data_ptr() + 4 * h, seg_start.data_ptr() + 4 * h
        L.check(L.gemm_nt(C.byref(a), _stream()))
    return out


def gemm_tn(a_mat, b_mat, out, *, N=None, K=None, R=None, splits=0, force_tr=-1, a_cols=0, b_cols=0):
    """out[n][k] += sum_r a[r][n] b[r][k]; out fp32 (N, >=K)."""
    if a_mat.dtype == torch.float32:        # verification kernel: X := a^T, W := b^T through strides, accumulate
        g = L.GemmF32Args()
        g.X, g.W, g.out = _p(a_mat), _p(b_mat), _p(out)
        g.M = a_mat.shape[1] if N is None else N
        g.N = b_mat.shape[1] if K is None else K
        g.K = a_mat.shape[0] if R is None else R
        g.sxm, g.sxk, g.swn, g.swk = 1, a_mat.stride(0), 1, b_mat.stride(0)
        g.ldo, g.epilogue, g.accumulate = _ld(out), L.EPI_BF16, 1
        L.check(L.gemm_f32(C.byref(g), _stream()))
        return out
    a = L.GemmTNArgs()
    a.A, a.B, a.out = _p(a_mat), _p(b_mat), _p(out)
    a.R = a_mat.shape[0] if R is None else R
    a.N = a_mat.shape[1] if N is None else N
    a.K = b_mat.shape[1] if K is None else K
    a.lda, a.ldb, a.ldo = _ld(a_mat), _ld(b_mat), _ld(out)
    a.a_cols = a_cols or a_mat.shape[1]
    a.b_cols = b_cols or b_mat.shape[1]
    a.splits, a.force_tr = splits, force_tr
    with _prof("gemm_tn", 2.0 * a.R * a.N * a.K, 2.0 * a.R * (a.N + a.K) + 4.0 * a.N * a.K, tag=f"R{a.R} N{a.N} K{a.K}"):
        L.check(L.gemm_tn(C.byref(a), _stream()))
    return out


def gemm_tn_multi(jobs):
    """One launch for a list of dense weight-gradient GEMMs: jobs = [(a_mat, b_mat, out, N, K, R), ...] with gemm_tn's meaning
    per entry (all dW of a transformer layer; csrc/gemm.hip gemm_tn_multi_kernel)."""
    if not jobs:
        return
    if jobs[0][0].dtype == torch.float32:
        for a_mat, b_mat, out, N, K, R in jobs:
            gemm_tn(a_mat, b_mat, out, N=N, K=K, R=R)
        return
    for lo in range(0, len(jobs), L.TN_MAX_JOBS):
        part = jobs[lo:lo + L.TN_MAX_JOBS]
        arr = (L.GemmTNJob * len(part))()
        flops = 0.0
        for j, (a_mat, b_mat, out, N, K, R) in zip(arr, part):
            j.A, j.B, j.out = _p(a_mat), _p(b_mat), _p(out)
            j.R, j.N, j.K = R, N, K
            j.lda, j.ldb, j.ldo = _ld(a_mat), _ld(b_mat), _ld(out)
            j.a_cols, j.b_cols = a_mat.shape[1], b_mat.shape[1]
            flops += 2.0 * R * N * K
        with _prof("gemm_tn_multi", flops, 0.0, tag=f"jobs{len(part)} R{part[0][5]}"):
            L.check(L.gemm_tn_multi(arr, len(part), _stream()))


@_timed("heads_gemm_tn (dW)")
def gemm_tn_grouped(a_mat, b_mat, groups, seg_start, seg_count, n_groups, max_N, max_R, K, *, splits=0, force_tr=-1):
    if a_mat.dtype == torch.float32:
        g = L.GemmF32Args()
        g.X, g.W, g.groups, g.seg_start, g.seg_count = _p(a_mat), _p(b_mat), _p(groups), _p(seg_start), _p(seg_count)
        g.N, g.max_N, g.n_groups = K, max_N, n_groups
        g.sxm, g.sxk, g.swn, g.swk = 1, a_mat.stride(0), 1, b_mat.stride(0)
        g.ldo, g.epilogue, g.accumulate = K, L.EPI_BF16, 1
        L.check(L.gemm_f32(C.byref(g), _stream()))
        return
    a = L.GemmTNArgs()
    a.A, a.B = _p(a_mat), _p(b_mat)
    a.K = K
    a.lda, a.ldb, a.ldo = _ld(a_mat), _ld(b_mat), K
    a.a_cols, a.b_cols = a_mat.shape[1], b_mat.shape[1]
    a.splits, a.force_tr = splits, force_tr
    a.groups, a.seg_start, a.seg_count = _p(groups), _p(seg_start), _p(seg_count)
    a.n_groups, a.max_N, a.max_R = n_groups, max_N, max_R
    L.check(L.gemm_tn(C.byref(a), _stream()))


def make_groups(entries, device):
    """entries: list of dicts(W=tensor|None, out=tensor|None, N, K, ldw) -> device byte tensor of fm_gemm_group."""
    arr = (L.GemmGroup * len(entries))()
    for i, e in enumerate(entries):
        arr[i].W = e["W"].data_ptr() if e.get("W") is not None else None
        arr[i].out = e["out"].data_ptr() if e.get("out") is not None else None
        arr[i].N, arr[i].K, arr[i].ldw = e["N"], e.get("K", 0), e.get("ldw", 0)
        arr[i].pad_ = e.get("transposed", 0)         # fp32 path: W is read as W[k][n] (stride ldw along k)
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    return host.to(device)


# ---------------------------------------------------------------------------------------------
# LayerNorm
# ---------------------------------------------------------------------------------------------
def layernorm_fwd(x, w, b, y, mean=None, rstd=None, row_map=None, eps=1e-6, R=None, delta=None, x_out=None):
    """y = LN(x) - or, with ``delta`` (bf16) and ``x_out`` (f32): x_out = x + delta, y = LN(x_out) in one pass."""
    R = x.shape[0] if R is None else R
    if delta is not None:
        with _prof("layernorm_fwd", 0.0, R * w.numel() * (10 + y.element_size())):
            L.check(L.layernorm_fwd_res(_p(x), _ld(x), _p(delta), _ld(delta), _p(x_out), _ld(x_out), _p(w), _p(b), _p(y), _ld(y),
                                        1 if y.dtype == torch.float32 else 0, _p(mean), _p(rstd), _p(row_map), R, w.numel(), eps, _stream()))
        return y
    with _prof("layernorm_fwd", 0.0, R * w.numel() * (4 + y.element_size())):
        return _ln_fwd(x, w, b, y, mean, rstd, row_map, eps, R)


def _ln_fwd(x, w, b, y, mean, rstd, row_map, eps, R):
    L.check(L.layernorm_fwd(_p(x), _ld(x), _p(w), _p(b), _p(y), _ld(y), 1 if y.dtype == torch.float32 else 0,
                            _p(mean), _p(rstd), _p(row_map), R, w.numel(), eps, _stream()))
    return y


def layernorm_bwd(dy, x, w, mean, rstd, dx, *, dres=None, dx_bf16=None, dw=None, db=None, dy_row_map=None, R=None, h=None):
    """``h`` (optional): the bf16 output the forward norm wrote (bias-free norm, same row order as x) - x_hat is then rebuilt from it
    instead of from the fp32 ``x`` (fm_layernorm_bwd_h: 14 instead of 16 bytes per element)."""
    R = x.shape[0] if R is None else R
    if h is not None and dy.dtype != torch.float32 and db is None and dy_row_map is None:
        with _prof("layernorm_bwd", 0.0, R * w.numel() * (2 + 2 + 4 + 4 + (2 if dx_bf16 is not None else 0))):
            L.check(L.layernorm_bwd_h(_p(dy), _ld(dy), _p(h), _ld(h), _p(x), _ld(x), _p(w), _p(mean), _p(rstd), _p(dres), _p(dx), _ld(dx),
                                      _p(dx_bf16), _ld(dx_bf16) if dx_bf16 is not None else 0, _p(dw), R, w.numel(), _stream()))
        return dx
    with _prof("layernorm_bwd", 0.0, R * w.numel() * (2 + 4 + 4 + 4 + (2 if dx_bf16 is not None else 0))):
        return _ln_bwd(dy, x, w, mean, rstd, dx, dres, dx_bf16, dw, db, dy_row_map, R)


def _ln_bwd(dy, x, w, mean, rstd, dx, dres, dx_bf16, dw, db, dy_row_map, R):
    if dy.dtype == torch.float32:
        L.check(L.layernorm_bwd_f32(_p(dy), _ld(dy), _p(dy_row_map), _p(x), _ld(x), _p(w), _p(mean), _p(rstd), _p(dres), _p(dx), _ld(dx),
                                    _p(dx_bf16), _ld(dx_bf16) if dx_bf16 is not None else 0, _p(dw), _p(db), R, w.numel(), _stream()))
        return dx
    L.check(L.layernorm_bwd(_p(dy), _ld(dy), _p(dy_row_map), _p(x), _ld(x), _p(w), _p(mean), _p(rstd), _p(dres), _p(dx), _ld(dx),
                            _p(dx_bf16), _ld(dx_bf16) if dx_bf16 is not None else 0, _p(dw), _p(db), R, w.numel(), _stream()))
    return dx


@_timed("headnorm")
def headnorm_fwd(x, w, b, y, stats, R, H, eps):
    """Per-head (64 features) LayerNorm of the q / k column block ``x`` -> ``y`` (both bf16 2-D views)."""
    if x.dtype == torch.float32:
        return L.check(L.headnorm_f32_fwd(_p(x), _ld(x), _p(w), _p(b), _p(y), _ld(y), _p(stats), R, H, eps, _stream()))
    L.check(L.headnorm_fwd(_p(x), _ld(x), _p(w), _p(b), _p(y), _ld(y), _p(stats), R, H, eps, _stream()))


@_timed("headnorm")
def headnorm_bwd(dy, x, w, stats, dx, dw, db, R, H):
    if x.dtype == torch.float32:
        return L.check(L.headnorm_f32_bwd(_p(dy), _ld(dy), _p(x), _ld(x), _p(w), _p(stats), _p(dx), _ld(dx), _p(dw), _p(db), R, H, _stream()))
    L.check(L.headnorm_bwd(_p(dy), _ld(dy), _p(x), _ld(x), _p(w), _p(stats), _p(dx), _ld(dx), _p(dw), _p(db), R, H, _stream()))


# ---------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------
def _attn_args(q, k, v, o, B, H, Nq, Nk, scale, mask_kind, kpad, cs, modq, modk, dense, causal, stat_m, stat_l, force_tr, zero_attn=False):
    a = L.AttnArgs()
    a.Q, a.K, a.V, a.O = _p(q), _p(k), _p(v), _p(o)
    a.ldq, a.ldk, a.ldv, a.ldo = q.stride(0), k.stride(0), v.stride(0), o.stride(0)
    a.B, a.H, a.Nq, a.Nk, a.head_dim, a.mask_kind = B, H, Nq, Nk, 64, mask_kind
    a.scale, a.causal = scale, 1 if causal else 0
    a.kpad, a.cs, a.modq, a.modk, a.dense = _p(kpad), _p(cs), _p(modq), _p(modk), _p(dense)
    a.stat_m, a.stat_l = _p(stat_m), _p(stat_l)
    a.force_tr, a.zero_attn = force_tr, 1 if zero_attn else 0
    return a


def attn_fwd(q, k, v, o, B, H, Nq, Nk, scale, *, mask_kind=L.MASK_NONE, kpad=None, cs=None, modq=None, modk=None, dense=None,
             causal=False, stat_m=None, stat_l=None, force_tr=-1, kv_batch_rows=0, zero_attn=False):
    """q/k/v/o: 2-D bf16 views whose row t of sample b is row b*N + t; head h occupies columns [64h, 64h+64).
    kv_batch_rows > Nk: k / v are views of a K/V cache whose sample b starts at row b * kv_batch_rows."""
    a = _attn_args(q, k, v, o, B, H, Nq, Nk, scale, mask_kind, kpad, cs, modq, modk, dense, causal, stat_m, stat_l, force_tr, zero_attn)
    a.kv_batch_rows = kv_batch_rows
    if q.dtype == torch.float32:
        L.check(L.attn_f32_fwd(C.byref(a), _stream()))
        return o
    with _prof("attn_fwd", 4.0 * B * H * Nq * Nk * 64):
        L.check(L.attn_fwd(C.byref(a), _stream()))
    return o


def attn_bwd(q, k, v, o, do, dq, dk, dv, B, H, Nq, Nk, scale, stat_m, stat_l, *, mask_kind=L.MASK_NONE, kpad=None, cs=None,
             modq=None, modk=None, dense=None, causal=False, force_tr=-1, zero_attn=False):
    a = _attn_args(q, k, v, o, B, H, Nq, Nk, scale, mask_kind, kpad, cs, modq, modk, dense, causal, stat_m, stat_l, force_tr, zero_attn)
    a.dO, a.dQ, a.dK, a.dV = _p(do), _p(dq), _p(dk), _p(dv)
    a.lddo, a.lddq, a.lddk, a.lddv = do.stride(0), dq.stride(0), dk.stride(0), dv.stride(0)
    if q.dtype == torch.float32:
        dk.zero_(); dv.zero_()                      # the verification kernel accumulates dK / dV with atomics
        L.check(L.attn_f32_bwd(C.byref(a), _stream()))
        return
    with _prof("attn_bwd", 10.0 * B * H * Nq * Nk * 64):
        L.check(L.attn_bwd(C.byref(a), _stream()))


# ---------------------------------------------------------------------------------------------
# heads / loss
# ---------------------------------------------------------------------------------------------
def padded_rows(R: int, n_heads: int) -> int:
    return ru(R, SEG) + SEG * (n_heads - 1)


@_timed("heads_misc")
def segment_rows(head_of_row, n_heads, seg_start, seg_count, perm, row_to_padded, tile_group):
    L.check(L.segment_rows(_p(head_of_row), head_of_row.numel(), n_heads, _p(seg_start), _p(seg_count), _p(perm),
                           _p(row_to_padded), _p(tile_group), perm.numel(), _stream()))


@_timed("heads_misc")
def gather_rows(src, perm, dst, D):
    L.check(L.gather_rows(_p(src), _ld(src), _p(perm), _p(dst), _ld(dst), perm.numel(), D, _stream()))


@_timed("cross_entropy")
def cross_entropy(logits, perm, tile_group, target_ids, vocab, seg_start, seg_count, n_heads, max_vocab, row_loss, row_lse, head_loss,
                  total_loss, *, loss_type=L.LOSS_MOD, grad_scale=None, write_grad=False):
    """write_grad=False: forward (losses + row_lse); write_grad=True: in-place d(logits) from the saved row_lse."""
    fn = L.cross_entropy_f32 if logits.dtype == torch.float32 else L.cross_entropy
    L.check(fn(_p(logits), _ld(logits), _p(perm), _p(tile_group), _p(target_ids), _p(vocab), _p(seg_start),
                            _p(seg_count), _p(grad_scale), loss_type, n_heads, perm.numel(), max_vocab, _p(row_loss), _p(row_lse),
                            _p(head_loss), _p(total_loss), 1 if write_grad else 0, _stream()))


# ---------------------------------------------------------------------------------------------
# element-wise
# ---------------------------------------------------------------------------------------------
@_timed("swiglu_bwd")
def swiglu_bwd(da, gu, dgu, H, Hp, R=None):
    if da.dtype == torch.float32:
        return L.check(L.swiglu_bwd_f32(_p(da), _ld(da), _p(gu), _ld(gu), _p(dgu), _ld(dgu), da.shape[0] if R is None else R, H, Hp, _stream()))
    L.check(L.swiglu_bwd(_p(da), _ld(da), _p(gu), _ld(gu), _p(dgu), _ld(dgu), da.shape[0] if R is None else R, H, Hp, _stream()))


@_timed("gelu_bwd")
def gelu_bwd(dh, pre, dpre, H, Hp, R=None):
    if dh.dtype == torch.float32:
        return L.check(L.gelu_bwd_f32(_p(dh), _ld(dh), _p(pre), _ld(pre), _p(dpre), _ld(dpre), dh.shape[0] if R is None else R, H, _stream()))
    L.check(L.gelu_bwd(_p(dh), _ld(dh), _p(pre), _ld(pre), _p(dpre), _ld(dpre), dh.shape[0] if R is None else R, H, Hp, _stream()))


def cast_pad(src, dst):
    """dst (rows, ld>=cols) bf16 <- src (rows, cols) f32, padding zeroed."""
    s2 = src.reshape(src.shape[0], -1)
    L.check(L.cast_pad(_p(s2), s2.stride(0), _p(dst), _ld(dst), s2.shape[0], s2.shape[1], _stream()))
    return dst


def transpose_cast_pad(src, dst):
    """dst (cols, width>=rows) bf16 <- src(rows, cols)^T, columns [rows, width) zeroed (dst may be a
    column slice of a wider buffer)."""
    s2 = src.reshape(src.shape[0], -1)
    L.check(L.transpose_cast_pad(_p(s2), s2.stride(0), _p(dst), _ld(dst), dst.shape[1], s2.shape[0], s2.shape[1], _stream()))
    return dst


def shadow_jobs_table(jobs, device):
    """jobs: list of (src f32 2-D view, dst bf16 | f32 2-D view, transpose[, col_scale f32 (cols,) | None]) ->
    (device byte tensor of fm_shadow_desc, total tiles)."""
    arr = (L.ShadowDesc * len(jobs))()
    tiles = 0
    for i, job in enumerate(jobs):
        src, dst, tr = job[:3]
        scale = job[3] if len(job) > 3 else None
        rows, cols = src.shape
        need = (cols, rows) if tr else (rows, cols)
        assert src.dtype == torch.float32 and dst.dtype in (torch.bfloat16, torch.float32) and src.stride(1) == 1 and dst.stride(1) == 1
        assert dst.shape[0] >= need[0] and dst.shape[1] >= need[1], (tuple(dst.shape), need)
        d = arr[i]
        d.src, d.dst, d.ld_src, d.ld_dst = src.data_ptr(), dst.data_ptr(), src.stride(0), dst.stride(0)
        d.rows, d.cols, d.transpose, d.tile_start = rows, cols, 1 if tr else 0, tiles
        d.dst_f32 = 1 if dst.dtype == torch.float32 else 0
        if scale is not None:
            assert scale.dtype == torch.float32 and scale.numel() == cols and scale.is_contiguous()
            d.col_scale = scale.data_ptr()
        tiles += ((rows + 63) // 64) * ((cols + 63) // 64)
    raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
    return raw, tiles


@_timed("fold_colscale_grad")
def fold_colscale_grad(jobs):
    """jobs = [(dWp f32 (rows, cols) view, W f32 master, gamma f32 (cols,), gW f32 | None, ggamma f32 (cols,) | None)]:
    gW += dWp * gamma[None, :];  ggamma += sum_r dWp * W  (csrc/elementwise.hip fold_colscale_grad_kernel)."""
    for lo in range(0, len(jobs), L.FOLD_MAX_JOBS):
        part = jobs[lo:lo + L.FOLD_MAX_JOBS]
        arr = (L.FoldGradJob * len(part))()
        for j, (dwp, w, gamma, gw, gg) in zip(arr, part):
            rows, cols = w.shape[0], w[0].numel()
            assert w.is_contiguous() and (gw is None or gw.is_contiguous()) and dwp.stride(1) == 1
            j.dWp, j.W, j.gamma, j.gW, j.ggamma = _p(dwp), _p(w), _p(gamma), _p(gw), _p(gg)
            j.rows, j.cols, j.ld_dwp = rows, cols, dwp.stride(0)
        L.check(L.fold_colscale_grad(arr, len(part), _stream()))


@_timed("shadow_refresh")
def shadow_refresh(table, n_jobs, tiles):
    L.check(L.shadow_refresh(_p(table), n_jobs, tiles, _stream()))


@_timed("colsum")
def colsum(dy, db, N, R=None):
    if dy.dtype == torch.float32:
        return L.check(L.colsum_f32(_p(dy), _ld(dy), _p(db), dy.shape[0] if R is None else R, N, _stream()))
    L.check(L.colsum(_p(dy), _ld(dy), _p(db), dy.shape[0] if R is None else R, N, _stream()))


@_timed("casts")
def f32_to_bf16(src, dst):
    if dst.dtype == torch.float32:
        return dst.copy_(src)
    L.check(L.f32_to_bf16(_p(src), _p(dst), src.numel(), _stream()))
    return dst


@_timed("casts")
def bf16_to_f32_scaled(src, dst, scale=1.0):
    L.check(L.bf16_to_f32_scaled(_p(src), _p(dst), src.numel(), float(scale), _stream()))
    return dst


@_timed("drop_path")
def scale_rows_bf16(x, scale, rows_per_sample, R, N=None):
    """x[r] *= scale[r // rows_per_sample] (bf16, in place): DropPath on a branch output / on the gradient entering the branch."""
    L.check(L.scale_rows_bf16(_p(x), _ld(x), _p(scale), rows_per_sample, R, x.shape[1] if N is None else N, _stream()))
    return x


@_timed("casts")
def add_bf16_to_f32(x, delta, out, R=None):
    """out[:R] = x[:R] + delta[:R] (contiguous (rows, D) buffers of equal width)."""
    R = x.shape[0] if R is None else R
    assert x.is_contiguous() and delta.is_contiguous() and out.is_contiguous() and x.shape[1] == delta.shape[1] == out.shape[1]
    L.check(L.add_bf16_f32(_p(x), _p(delta), _p(out), R * x.shape[1], _stream()))
    return out


@_timed("adamw")
def adamw(p, g, m, v, n, lr, beta1, beta2, eps, wd, step, grad_mult=None, hyper=None, sumsq=None):
    L.check(L.adamw(_p(p), _p(g), _p(m), _p(v), n, lr, beta1, beta2, eps, wd, step, _p(grad_mult), _p(hyper), _p(sumsq), _stream()))


ADAMW_CHUNK = 8192   # FM_ADAMW_CHUNK


def adamw_jobs_table(jobs, device):
    """jobs: list of dicts(p, g, m, v: fp32 tensors of one (rows, cols) matrix; plain / t: bf16 2-D destinations or None)
    -> (device byte tensor of fm_adamw_job, total tiles)."""
    arr = (L.AdamWJob * len(jobs))()
    tiles = 0
    for i, j in enumerate(jobs):
        p = j["p"]
        rows, cols = p.shape[0], p[0].numel()
        d = arr[i]
        for k in ("p", "g", "m", "v"):
            t = j[k]
            assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == rows * cols, (k, t.dtype, tuple(t.shape))
            setattr(d, k, t.data_ptr())
        pl, tr = j.get("plain"), j.get("t")
        if pl is not None:
            assert pl.dtype == torch.bfloat16 and pl.stride(1) == 1 and pl.shape[0] >= rows and pl.shape[1] >= cols
            d.dst_plain, d.ld_plain = pl.data_ptr(), pl.stride(0)
        assert tr is None, "transposed shadows are refreshed by fm_shadow_refresh"
        d.rows, d.cols, d.tile_start = rows, cols, tiles
        tiles += (rows * cols + ADAMW_CHUNK - 1) // ADAMW_CHUNK
    raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
    return raw, tiles


@_timed("adamw")
def adamw_shadow(table, n_jobs, tiles, lr, beta1, beta2, eps, wd, step, grad_mult=None, hyper=None, sumsq=None):
    L.check(L.adamw_shadow(_p(table), n_jobs, tiles, lr, beta1, beta2, eps, wd, step, _p(grad_mult), _p(hyper), _p(sumsq), _stream()))


@_timed("grad_norm")
def sumsq(x, out):
    L.check(L.sumsq(_p(x), x.numel(), _p(out), _stream()))


@_timed("grad_norm")
def clip_coef(ss, max_norm, norm_out, coef_out):
    L.check(L.clip_coef(_p(ss), float(max_norm or 0.0), _p(norm_out), _p(coef_out), _stream()))


def dense_decoder_mask(cs, mod, B, M, causal, use_sep):
    out = torch.empty(B, M, M, dtype=torch.bool, device=(cs if cs is not None else mod).device)
    L.check(L.dense_decoder_mask(_p(cs), _p(mod), _p(out), B, M, 1 if causal else 0, 1 if cs is not None else 0,
                                 1 if use_sep else 0, _stream()))
    return out
