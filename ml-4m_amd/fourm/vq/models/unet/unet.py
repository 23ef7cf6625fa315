"""``PatchedUNetCondCat`` / ``unet_patched``: the conditional UNet behind the DiVAE detokenizers, inference on gfx950.

API of upstream ``fourm/vq/models/unet/unet.py`` (UNetModel :411-690, PatchedUNetCondCat :693-744, unet_patched :747-754): same
constructor arguments, same parameter tree (``time_embed.{0,2}``, ``input_blocks.i.j.{in_layers.0|in_layers.2|emb_layers.1|
out_layers.0|out_layers.3|skip_connection|norm|qkv|proj_out|op}``, ``middle_block.k``, ``output_blocks.i.j.(...|conv)``, ``out.{0,2}``:
upstream checkpoints load with strict=True), same ``forward(sample, timestep, encoder_hidden_states, cond_mask=None)``.

The torch modules below only HOLD the parameters; the forward runs on the HIP kernels (csrc/unet.hip + the NT GEMMs): feature maps as
(B * H * W, C) bf16 rows, every convolution a GEMM (3 x 3 through fm_unet_im2col), GroupNorm + SiLU in fp32 arithmetic, bf16 GEMM operands
with fp32 accumulation = upstream's autocast arithmetic.  Inference only (the decoder is trained upstream; no backward here).
Not implemented (rejected loudly): class conditioning, scale-shift norm, ResBlock up / down sampling, the new attention order, dropout."""
import math
from typing import Optional, Union

import torch
import torch.nn as nn

from fourm.hip import _lib as L
from fourm.hip import ops


import os

# 3 x 3 convolutions with C % 64 == 0 input channels as implicit GEMMs (no im2col round trip; bit-identical).  FOURM_UNET_IMPLICIT_CONV=0: im2col + GEMM.
IMPLICIT_CONV = os.environ.get("FOURM_UNET_IMPLICIT_CONV", "1") == "1"


# One evaluation = ~430 launches issued from Python: at batch 8 the host needs 8.3 ms to enqueue what the GPU runs in ~6.5 ms (tools/divae_host_probe.py).
# In eval mode the launch sequence of a (shape, stream) is captured into a hipGraph after two eager evaluations and replayed from then on
# (inputs copied into static buffers, the output copied out).  FOURM_UNET_GRAPH=0: always eager.
UNET_GRAPH = os.environ.get("FOURM_UNET_GRAPH", "1") == "1"


def ru(x, m):
    return (x + m - 1) // m * m


class _Res(nn.Module):
    def __init__(self, cin, cout, emb_ch):
        super().__init__()
        self.in_layers = nn.Sequential(nn.GroupNorm(32, cin), nn.SiLU(), nn.Conv2d(cin, cout, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_ch, cout))
        self.out_layers = nn.Sequential(nn.GroupNorm(32, cout), nn.SiLU(), nn.Dropout(p=0.0), nn.Conv2d(cout, cout, 3, padding=1))
        for p in self.out_layers[3].parameters():
            p.detach().zero_()                                                  # zero_module (unet.py:224-227)
        self.skip_connection = nn.Identity() if cin == cout else nn.Conv2d(cin, cout, 1)
        self.cin, self.cout = cin, cout


class _Attn(nn.Module):
    def __init__(self, ch, heads):
        super().__init__()
        self.norm = nn.GroupNorm(32, ch)
        self.qkv = nn.Conv1d(ch, 3 * ch, 1)
        self.proj_out = nn.Conv1d(ch, ch, 1)
        for p in self.proj_out.parameters():
            p.detach().zero_()
        self.ch, self.heads = ch, heads


class _Down(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.op = nn.Conv2d(ch, ch, 3, stride=2, padding=1)
        self.ch = ch


class _Up(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)
        self.ch = ch


class PatchedUNetCondCat(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, cond_channels: int, patch_size: int, image_size=224, model_channels=256,
                 num_res_blocks=3, attention_resolutions=(8, 16), dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None,
                 use_checkpoint=False, num_heads=1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 use_new_attention_order=False):
        super().__init__()
        if dims != 2 or num_classes is not None or use_scale_shift_norm or resblock_updown or use_new_attention_order or not conv_resample or dropout:
            raise NotImplementedError("PatchedUNetCondCat (HIP): only the options of unet_patched are implemented (2-D, conv resampling, no class "
                                      "conditioning / scale-shift norm / ResBlock resampling / new attention order / dropout)")
        self.P_H = self.P_W = int(patch_size)
        self.in_channels, self.out_channels, self.cond_channels = in_channels, out_channels, cond_channels
        self.image_size = self.sample_size = image_size
        self.model_channels, self.num_res_blocks, self.channel_mult = model_channels, num_res_blocks, tuple(channel_mult)
        self.attention_resolutions = tuple(attention_resolutions)
        in_p, out_p = in_channels * patch_size * patch_size + cond_channels, out_channels * patch_size * patch_size
        te = model_channels * 4
        heads = lambda c: num_heads if num_head_channels == -1 else c // num_head_channels
        self.time_embed = nn.Sequential(nn.Linear(model_channels, te), nn.SiLU(), nn.Linear(te, te))
        ch = int(channel_mult[0] * model_channels)
        self.input_blocks = nn.ModuleList([nn.Sequential(nn.Conv2d(in_p, ch, 3, padding=1))])
        chans, ds = [ch], 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [_Res(ch, int(mult * model_channels), te)]
                ch = int(mult * model_channels)
                if ds in self.attention_resolutions:
                    layers.append(_Attn(ch, heads(ch)))
                self.input_blocks.append(nn.Sequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(nn.Sequential(_Down(ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = nn.Sequential(_Res(ch, ch, te), _Attn(ch, heads(ch)), _Res(ch, ch, te))
        self.output_blocks = nn.ModuleList([])
        up_heads = (lambda c: num_heads_upsample if num_head_channels == -1 else c // num_head_channels) if num_heads_upsample != -1 else heads
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                layers = [_Res(ch + chans.pop(), int(model_channels * mult), te)]
                ch = int(model_channels * mult)
                if ds in self.attention_resolutions:
                    layers.append(_Attn(ch, up_heads(ch)))
                if level and i == num_res_blocks:
                    layers.append(_Up(ch))
                    ds //= 2
                self.output_blocks.append(nn.Sequential(*layers))
        self.out = nn.Sequential(nn.GroupNorm(32, ch), nn.SiLU(), nn.Conv2d(ch, out_p, 3, padding=1))
        for p in self.out[2].parameters():
            p.detach().zero_()
        self._engine = None

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @torch.no_grad()
    def forward(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int], encoder_hidden_states: torch.Tensor = None,
                cond_mask: Optional[torch.Tensor] = None, **kwargs) -> torch.Tensor:
        """sample (B, C, H, W); timestep: number or (B,) / (1,) tensor; encoder_hidden_states (B, D, Hc, Wc): the de-quantised latents;
        cond_mask (B, Hc, Wc) bool, True = that conditioning vector is zeroed (unet.py:727-729).  Returns f32 (B, out_channels, H, W)."""
        if not sample.is_cuda:
            raise RuntimeError("PatchedUNetCondCat runs on the HIP kernels only: move the model and its inputs to an MI355X (there is no CPU path)")
        if self._engine is None:
            self._engine = _UNetEngine(self)
        if UNET_GRAPH and not self.training and not torch.cuda.is_current_stream_capturing():
            return self._engine.forward_graphed(sample, timestep, encoder_hidden_states, cond_mask)
        return self._engine.forward(sample, timestep, encoder_hidden_states, cond_mask)


def unet_patched(**kwargs):
    return PatchedUNetCondCat(patch_size=4, model_channels=256, num_res_blocks=3, attention_resolutions=[4, 8], channel_mult=(1, 2, 2, 2), **kwargs)


class _UNetEngine:
    """Launch sequence of one UNet evaluation.  bf16 weight images are cached per parameter (refreshed when the parameter changed)."""

    def __init__(self, net: PatchedUNetCondCat):
        self.net = net
        self._w, self._buf = {}, {}
        res = [m for m in net.modules() if isinstance(m, _Res)]
        self._res_index = {id(m): i for i, m in enumerate(res)}
        self._res = res
        offs, o = [], 0
        for m in res:
            offs.append(o)
            o += ru(m.cout, 4)
        self._emb_off, self._emb_total = offs, ru(o, 64)

    # ---- cached operands -------------------------------------------------------------------------------------------------------------
    def _stamp(self, *ps):
        return tuple((p._version, p.data_ptr()) for p in ps)

    def w_conv(self, conv):
        """(Cout, k * k * Cin padded to 64) bf16: conv.weight with the taps outermost, the GEMM's W operand behind fm_unet_im2col."""
        w = conv.weight
        key = ("w", id(w))
        hit = self._w.get(key)
        if hit is None or hit[0] != self._stamp(w):
            co = w.shape[0]
            flat = (w.detach().permute(0, 2, 3, 1) if w.dim() == 4 else w.detach().permute(0, 2, 1)).reshape(co, -1).float().contiguous()
            K = flat.shape[1]
            img = torch.zeros(co, ru(K, 64), dtype=torch.bfloat16, device=w.device)
            tmp = torch.empty(co * K, dtype=torch.bfloat16, device=w.device)
            ops.f32_to_bf16(flat, tmp)
            img[:, :K] = tmp.view(co, K)
            hit = self._w[key] = (self._stamp(w), img)
        return hit[1]

    def w_emb_all(self):
        """Every ResBlock's emb_layers Linear stacked into one (sum Cout, 4 mc) operand + bias: ONE GEMM per evaluation gives all the
        per-block timestep embeddings (the input silu(emb) is the same for all of them)."""
        ps = [p for m in self._res for p in (m.emb_layers[1].weight, m.emb_layers[1].bias)]
        key = ("emb_all",)
        hit = self._w.get(key)
        if hit is None or hit[0] != self._stamp(*ps):
            te = self._res[0].emb_layers[1].weight.shape[1]
            dev = ps[0].device
            wf = torch.zeros(self._emb_total, te, dtype=torch.float32, device=dev)
            bf = torch.zeros(self._emb_total, dtype=torch.float32, device=dev)
            for m, o in zip(self._res, self._emb_off):
                wf[o:o + m.cout] = m.emb_layers[1].weight.detach()
                bf[o:o + m.cout] = m.emb_layers[1].bias.detach()
            img = torch.empty(self._emb_total, te, dtype=torch.bfloat16, device=dev)
            ops.f32_to_bf16(wf, img)
            hit = self._w[key] = (self._stamp(*ps), img, bf)
        return hit[1], hit[2]

    def buf(self, tag, rows, cols, dtype=torch.bfloat16):
        key = (tag, dtype, torch.cuda.current_stream(self.net.device).cuda_stream, ops.SCRATCH_TAG)      # one scratch set per stream / per captured graph
        b = self._buf.get(key)
        n = rows * cols
        if b is None or b.numel() < n or b.device != self.net.device:
            b = self._buf[key] = torch.empty(max(n, 1), dtype=dtype, device=self.net.device)
        return b[:n].view(rows, cols)

    # ---- building blocks -------------------------------------------------------------------------------------------------------------
    def gemm(self, x, w, bias, out, M, N, K, f32_out=False):
        ops.gemm_nt(x, w, out, epilogue=L.EPI_F32 if f32_out else L.EPI_BF16, bias=bias.detach().float().contiguous() if bias is not None else None, M=M, N=N, K=K)
        return out

    def im2col(self, tag, src1, C1, B, H, W, ksize=3, stride=1, up1=0, src2=None, C2=0, H2=0, W2=0):
        kp = ru(ksize * ksize * (C1 + C2), 64)
        pad = ksize // 2
        Ho, Wo = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
        out = self.buf(tag, B * Ho * Wo, kp)
        L.check(L.unet_im2col(ops._p(src1), src1.stride(0), C1, ops._p(src2), src2.stride(0) if src2 is not None else 0, C2, H2, W2, ops._p(out), kp, kp,
                              B, H, W, ksize, stride, up1, ops._stream()))
        return out, Ho, Wo

    def gn(self, tag, x, norm, B, HW, C, silu, add=None):
        y = self.buf(tag, B * HW, C)
        st = self.buf("gn_stats", B * norm.num_groups * ((HW + 31) // 32 + 1), 2, torch.float32)
        L.check(L.groupnorm_nhwc(ops._p(x), x.stride(0), ops._p(add), add.stride(0) if add is not None else 0, ops._p(norm.weight.detach()), ops._p(norm.bias.detach()),
                                 ops._p(y), C, ops._p(st), B, HW, C, norm.num_groups, float(norm.eps), 1 if silu else 0, ops._stream()))
        return y

    def conv3(self, tag, x, conv, B, H, W, stride=1, up1=0, out=None, f32_out=False):
        """3 x 3 convolution (padding 1) of the (B, H >> up1, W >> up1, C) rows x read on the (H, W) grid.  C % 64 == 0: implicit GEMM (the gather runs
        inside the GEMM's LDS-DMA addresses, fm_gemm_nt_args.conv_*); else fm_unet_im2col + the plain GEMM."""
        C, Co = conv.weight.shape[1], conv.weight.shape[0]
        Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        if out is None:
            out = self.buf(tag, B * Ho * Wo, Co)
        if IMPLICIT_CONV and C % 64 == 0:
            ops.gemm_nt(x, self.w_conv(conv), out, epilogue=L.EPI_F32 if f32_out else L.EPI_BF16, bias=conv.bias.detach().float().contiguous() if conv.bias is not None else None,
                        M=B * Ho * Wo, N=Co, K=9 * C, conv=dict(C=C, H=H, W=W, Ho=Ho, Wo=Wo, stride=stride, up=up1))
            return out, Ho, Wo
        col, Ho, Wo = self.im2col("col", x, C, B, H, W, 3, stride, up1)
        self.gemm(col, self.w_conv(conv), conv.bias, out, B * Ho * Wo, Co, col.shape[1], f32_out=f32_out)
        return out, Ho, Wo

    def res(self, tag, m, x, emb_all, B, H, W):
        R = B * H * W
        a = self.gn("act", x, m.in_layers[0], B, H * W, m.cin, True)
        h, _, _ = self.conv3("h1", a, m.in_layers[2], B, H, W)
        o = self._emb_off[self._res_index[id(m)]]
        a2 = self.gn("act", h, m.out_layers[0], B, H * W, m.cout, True, add=emb_all[:, o:o + m.cout])
        h2, _, _ = self.conv3("h2", a2, m.out_layers[3], B, H, W)
        if isinstance(m.skip_connection, nn.Identity):
            xs = x
        else:
            xs = self.buf("skipc", R, m.cout)
            xin = x
            if m.cin % 64:                           # the GEMM reduces over whole 64-column groups: zero-padded copy of the rows
                xin, _, _ = self.im2col("col", x, m.cin, B, H, W, ksize=1)
            self.gemm(xin, self.w_conv(m.skip_connection), m.skip_connection.bias, xs, R, m.cout, ru(m.cin, 64))
        out = torch.empty(R, m.cout, dtype=torch.bfloat16, device=x.device)
        L.check(L.add_bf16(ops._p(xs), xs.stride(0), ops._p(h2), h2.stride(0), ops._p(out), m.cout, R, m.cout, ops._stream()))
        return out

    def attn(self, m, x, B, H, W):
        T, C = H * W, m.ch
        n = self.gn("act", x, m.norm, B, T, C, False)
        qkv = self.buf("qkv", B * T, 3 * C)
        nin = n
        if C % 64:
            nin, _, _ = self.im2col("col", n, C, B, H, W, ksize=1)
        self.gemm(nin, self.w_conv(m.qkv), m.qkv.bias, qkv, B * T, 3 * C, ru(C, 64))
        a = self.buf("attn_o", B * T, ru(C, 64))
        if C % 64:
            a.zero_()
        L.check(L.unet_attention(ops._p(qkv), 3 * C, ops._p(a), a.stride(0), B, T, m.heads, C // m.heads, ops._stream()))
        pr = self.buf("h2", B * T, C)
        self.gemm(a, self.w_conv(m.proj_out), m.proj_out.bias, pr, B * T, C, ru(C, 64))
        out = torch.empty(B * T, C, dtype=torch.bfloat16, device=x.device)
        L.check(L.add_bf16(ops._p(x), x.stride(0), ops._p(pr), C, ops._p(out), C, B * T, C, ops._stream()))
        return out

    def run(self, seq, h, emb_all, B, H, W):
        for m in seq:
            if isinstance(m, _Res):
                h = self.res("r", m, h, emb_all, B, H, W)
            elif isinstance(m, _Attn):
                h = self.attn(m, h, B, H, W)
            elif isinstance(m, _Down):
                o, H, W = self.conv3("down", h, m.op, B, H, W, stride=2)
                h = o.clone()
            elif isinstance(m, _Up):
                o, H, W = self.conv3("up", h, m.conv, B, 2 * H, 2 * W, up1=1)
                h = o.clone()
            else:
                raise TypeError(type(m))
        return h, H, W

    # ---- one evaluation, replayed from a hipGraph ----------------------------------------------------------------------------------
    def _weights_stamp(self):
        return tuple((p._version, p.data_ptr()) for p in self.net.parameters())

    def forward_graphed(self, sample, timestep, cond, cond_mask):
        """forward() through a captured graph per (shapes, mask or not, stream).  The first two evaluations of a key run eagerly (they build the
        cached weight images and size every scratch buffer); the third is captured.  A parameter that changed (version / storage) drops the graphs."""
        dev = sample.device
        key = (tuple(sample.shape), tuple(cond.shape), cond_mask is not None, torch.cuda.current_stream(dev).cuda_stream)
        stamp = self._weights_stamp()
        if getattr(self, "_graph_stamp", None) != stamp:
            self._graphs, self._graph_warm, self._graph_stamp = {}, {}, stamp
        g = self._graphs.get(key)
        if g is None:
            n = self._graph_warm.get(key, 0)
            if n < 2:
                self._graph_warm[key] = n + 1
                return self.forward(sample, timestep, cond, cond_mask)
            B = sample.shape[0]
            st = dict(x=sample.detach().float().contiguous().clone(), t=torch.zeros(B, dtype=torch.float32, device=dev), c=cond.detach().float().contiguous().clone(),
                      m=cond_mask.to(dev).clone() if cond_mask is not None else None)
            torch.cuda.current_stream(dev).synchronize()
            graph = torch.cuda.CUDAGraph()
            ops.SCRATCH_TAG = ("graph", len(self._graphs), key)          # scratch allocated during the capture lives in this graph's pool and is its alone
            try:
                with torch.cuda.graph(graph):
                    st["out"] = self.forward(st["x"], st["t"], st["c"], st["m"])
            finally:
                ops.SCRATCH_TAG = None
            while len(self._graphs) >= 6:                      # (a graph keeps its scratch: bound what a stream of changing shapes can pile up)
                self._graphs.pop(next(iter(self._graphs)))
            g = self._graphs[key] = (graph, st)
        graph, st = g
        st["x"].copy_(sample.detach())
        st["c"].copy_(cond.detach())
        if st["m"] is not None:
            st["m"].copy_(cond_mask)
        if torch.is_tensor(timestep):
            st["t"].copy_(timestep.detach().reshape(-1).float().expand(st["t"].shape[0]) if timestep.numel() == 1 else timestep.detach().reshape(-1).float())
        else:
            st["t"].fill_(float(timestep))
        graph.replay()
        return st["out"].clone()

    # ---- one evaluation --------------------------------------------------------------------------------------------------------------
    def forward(self, sample, timestep, cond, cond_mask):
        net = self.net
        dev = sample.device
        B, C, H, W = sample.shape
        P = net.P_H
        if H % P or W % P:
            raise ValueError(f"Image sizes {H}x{W} must be divisible by patch sizes {P}x{P}")
        nh, nw = H // P, W // P
        x32 = sample.detach().float().contiguous()
        CP = C * P * P
        if CP % 8 or net.cond_channels % 8:
            raise NotImplementedError("patch / conditioning widths must be multiples of 8")
        rows = self.buf("patch", B * nh * nw, CP)
        if nh == nw and H == W:
            L.check(L.vq_patchify(ops._p(x32), ops._p(rows), CP, B, C, H, W, P, ops._stream()))
        else:
            raise NotImplementedError("non-square inputs")
        cnd = cond.detach().float()
        if cond_mask is not None:
            cnd = torch.where(cond_mask[:, None].to(dev), torch.zeros((), device=dev), cnd)
        D, Hc, Wc = cnd.shape[1:]
        crow32 = cnd.permute(0, 2, 3, 1).reshape(B * Hc * Wc, D).contiguous()
        crow = self.buf("cond", B * Hc * Wc, D)
        ops.f32_to_bf16(crow32, crow)
        # timestep embedding -> time_embed MLP -> silu -> every ResBlock's projection in one GEMM
        t = torch.as_tensor(timestep, device=dev).reshape(-1).float()
        if t.numel() == 1:
            t = t.expand(B)
        t = t.contiguous()
        mc, te = net.model_channels, net.model_channels * 4
        temb = self.buf("temb", B, ru(mc, 64))
        temb.zero_()
        L.check(L.timestep_embedding(ops._p(t), ops._p(temb), temb.stride(0), B, mc, 10000.0, ops._stream()))
        l0, l2 = net.time_embed[0], net.time_embed[2]
        e1 = self.buf("e1", B, te, torch.float32)
        self.gemm(temb, self.w_conv_lin(l0), l0.bias, e1, B, te, ru(mc, 64), f32_out=True)
        s1 = self.buf("s1", B, te)
        L.check(L.silu_f32_to_bf16(ops._p(e1), ops._p(s1), B * te, ops._stream()))
        e2 = self.buf("e2", B, te, torch.float32)
        self.gemm(s1, self.w_conv_lin(l2), l2.bias, e2, B, te, te, f32_out=True)
        s2 = self.buf("s2", B, te)
        L.check(L.silu_f32_to_bf16(ops._p(e2), ops._p(s2), B * te, ops._stream()))
        wall, ball = self.w_emb_all()
        emb_all = self.buf("emb_all", B, self._emb_total, torch.float32)
        ops.gemm_nt(s2, wall, emb_all, epilogue=L.EPI_F32, bias=ball, M=B, N=self._emb_total, K=te)
        # input blocks
        first = net.input_blocks[0][0]
        col, _, _ = self.im2col("col", rows, CP, B, nh, nw, 3, 1, 0, src2=crow, C2=D, H2=Hc, W2=Wc)
        ch0 = first.weight.shape[0]
        h = torch.empty(B * nh * nw, ch0, dtype=torch.bfloat16, device=dev)
        self.gemm(col, self.w_conv(first), first.bias, h, B * nh * nw, ch0, col.shape[1])
        hs, hh, ww = [(h, nh, nw)], nh, nw
        for blk in list(net.input_blocks)[1:]:
            h, hh, ww = self.run(blk, h, emb_all, B, hh, ww)
            hs.append((h, hh, ww))
        h, hh, ww = self.run(net.middle_block, h, emb_all, B, hh, ww)
        for blk in net.output_blocks:
            skip, sh, sw = hs.pop()
            assert (sh, sw) == (hh, ww)
            C1, C2 = h.shape[1], skip.shape[1]
            cat = torch.empty(B * hh * ww, C1 + C2, dtype=torch.bfloat16, device=dev)
            L.check(L.unet_im2col(ops._p(h), h.stride(0), C1, ops._p(skip), skip.stride(0), C2, hh, ww, ops._p(cat), C1 + C2, C1 + C2, B, hh, ww, 1, 1, 0, ops._stream()))
            h, hh, ww = self.run(blk, cat, emb_all, B, hh, ww)
        a = self.gn("act", h, net.out[0], B, hh * ww, h.shape[1], True)
        conv = net.out[2]
        OP = conv.weight.shape[0]
        y = self.buf("y", B * hh * ww, ru(OP, 4), torch.float32)
        self.conv3("y", a, conv, B, hh, ww, out=y, f32_out=True)
        img = torch.empty(B, net.out_channels, H, W, dtype=torch.float32, device=dev)
        L.check(L.vq_unpatchify(ops._p(y), y.stride(0), ops._p(img), B, net.out_channels, H, W, P, ops._stream()))
        return img

    def w_conv_lin(self, lin):
        w = lin.weight
        key = ("lin", id(w))
        hit = self._w.get(key)
        if hit is None or hit[0] != self._stamp(w):
            o, i = w.shape
            img = torch.zeros(o, ru(i, 64), dtype=torch.bfloat16, device=w.device)
            tmp = torch.empty(o, i, dtype=torch.bfloat16, device=w.device)
            ops.f32_to_bf16(w.detach().float().contiguous(), tmp)
            img[:, :i] = tmp
            hit = self._w[key] = (self._stamp(w), img)
        return hit[1]
