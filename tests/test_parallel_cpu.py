"""Data-parallel logic without GPUs: gloo, world_size 2.  Covers the stage-wise bucketed gradient
mean (GradReducer), the stage -> slice partition of the flat gradient store, and the synthetic batch
generator's contract."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.golden.cases import build_case
from tests.util_model import build_hip_model


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, n, stages, bucket, out_dir, min_launch_mb=0.0):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fourm.parallel import GradReducer
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red = GradReducer(flat, stages, bucket_elems=bucket, min_launch_mb=min_launch_mb)
    for window in range(2):                      # two optimizer steps reuse the reducer
        if window:
            flat.copy_(torch.arange(n, dtype=torch.float32) * (rank + 1) + window)
        red.begin()
        for s in ["b", "a"]:                     # backward order; "c" is left to finish()
            red.stage_done(s)
            if min_launch_mb == float("inf"):    # "tail" exchange: nothing leaves before finish()
                assert not red._pending
        red.finish()
        torch.save(flat.clone(), os.path.join(out_dir, f"r{rank}_w{window}.pt"))
    dist.destroy_process_group()


def test_grad_reducer_mean_gloo(tmp_path):
    n, world = 1000, 2
    stages = {"a": [(0, 100), (100, 200)], "b": [(300, 500)], "c": [(800, 150)]}    # [950, 1000) belongs to nobody
    mp.spawn(_worker, args=(world, _free_port(), n, stages, 128, str(tmp_path)), nprocs=world, join=True)
    for window in range(2):
        base = torch.arange(n, dtype=torch.float32)
        mean = (base * 1 + window + base * 2 + window) / 2 if window else base * 1.5
        for rank in range(world):
            got = torch.load(tmp_path / f"r{rank}_w{window}.pt")
            assert torch.allclose(got[:950], mean[:950]), (window, rank)
            own = base * (rank + 1) + (window if window else 0)
            assert torch.equal(got[950:], own[950:])          # untouched outside every stage


def test_grad_reducer_tail_exchange_gloo(tmp_path):
    """exchange="tail" (DataParallel) = an infinite launch threshold: every slice goes out in finish(), same means."""
    n, world = 1000, 2
    stages = {"a": [(0, 100), (100, 200)], "b": [(300, 500)], "c": [(800, 150)]}
    mp.spawn(_worker, args=(world, _free_port(), n, stages, 128, str(tmp_path), float("inf")), nprocs=world, join=True)
    base = torch.arange(n, dtype=torch.float32)
    for rank in range(world):
        assert torch.allclose(torch.load(tmp_path / f"r{rank}_w0.pt")[:950], (base * 1.5)[:950])
        assert torch.allclose(torch.load(tmp_path / f"r{rank}_w1.pt")[:950], (base * 1.5 + 1)[:950])


def _worker4(rank, world, port, n, stages, bucket, out_dir, algorithm, wire):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fourm.parallel import GradReducer
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(n, generator=g)
    torch.save(flat.clone(), os.path.join(out_dir, f"in{rank}.pt"))
    kw = {}
    if wire is not None:        # the wire-format kernels are HIP (fm_f32_to_bf16 / fm_bf16_to_f32_scaled); the bucketing logic under test is not
        kw = dict(wire_dtype=wire, pack=lambda src, dst: dst.copy_(src), unpack=lambda src, dst, scale: dst.copy_(src.float() * scale))
    red = GradReducer(flat, stages, bucket_elems=bucket, algorithm=algorithm, min_launch_mb=0.0, **kw)
    red.begin()
    for s in ["b", "a"]:
        red.stage_done(s)
    red.finish()
    torch.save(dict(flat=flat.clone(), n_coll=red.n_collectives, wire_bytes=red.bytes_on_wire), os.path.join(out_dir, f"out{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("algorithm,wire", [("reduce_scatter", None), ("reduce_scatter", torch.bfloat16), ("all_reduce", torch.bfloat16)])
def test_grad_reducer_world_size_four(tmp_path, algorithm, wire):
    """World size 4: reduce-scatter + all-gather of every bucket (heads that do not divide by 4 leave a tail to
    all_reduce) and the bf16 wire format (one rounding of each rank's slice, summed in bf16 on the wire, mean in fp32)."""
    n, world = 1003, 4
    stages = {"a": [(0, 101), (101, 200)], "b": [(301, 498)], "c": [(800, 150)]}      # odd lengths: 301 = 75 * 4 + 1, 498 = 124 * 4 + 2, 150 = 37 * 4 + 2
    mp.spawn(_worker4, args=(world, _free_port(), n, stages, 128, str(tmp_path), algorithm, wire), nprocs=world, join=True)
    ins = [torch.load(tmp_path / f"in{r}.pt") for r in range(world)]
    if wire is None:
        want = sum(ins) / world
    else:       # every rank's slice is rounded once, the ranks are summed (bf16 partial sums on the wire), the mean is taken in fp32
        want = sum(t.to(torch.bfloat16).float() for t in ins) / world
    for rank in range(world):
        out = torch.load(tmp_path / f"out{rank}.pt")
        got = out["flat"]
        tol = dict(rtol=1e-6, atol=1e-6) if wire is None else dict(rtol=2e-2, atol=2e-2)     # bf16 accumulation across 4 ranks
        for o, k in ((0, 301), (301, 498), (800, 150)):
            assert torch.allclose(got[o:o + k], want[o:o + k], **tol), (rank, o)
        assert torch.equal(got[950:], ins[rank][950:]) and torch.equal(got[799:800], ins[rank][799:800])      # outside every stage
        assert out["n_coll"] >= 3 and out["wire_bytes"] == 949 * (2 if wire is not None else 4)
    assert all(torch.equal(torch.load(tmp_path / "out0.pt")["flat"][:799], torch.load(tmp_path / f"out{r}.pt")["flat"][:799]) for r in range(1, world))


def test_reducer_rejects_overlapping_stages():
    from fourm.parallel import GradReducer
    with pytest.raises(ValueError):
        GradReducer(torch.zeros(10), {"a": [(0, 6)], "b": [(5, 5)]})


def test_grad_stage_partition_covers_every_parameter_once():
    from fourm.hip.engine import FourMEngine
    case = build_case("micro_gelu")
    model = build_hip_model(case["cfg"], case["share_embedding"], case["norm_bias"], case["learned_pos"])
    eng = FourMEngine(model)
    eng.flatten()
    stages = eng.grad_stages()
    cover = torch.zeros(eng.flat_grads.numel(), dtype=torch.int32)
    for rs in stages.values():
        for o, n in rs:
            cover[o:o + n] += 1
    assert int(cover.max()) == 1
    for n, p in model.named_parameters():
        o, k = eng._slices[id(p)]
        assert int(cover[o:o + k].min()) == 1, n
        assert p.data_ptr() == eng.flat_params.data_ptr() + 4 * o        # parameters are views of the flat store
    # block weights are exchanged when their block's backward finishes; shared / tiny tensors at the end
    # (the micro model's tables are below the 64K-element threshold, so they ride in the tail slice)
    assert {"dec0", "dec1", "enc0", "enc1", "tail"} <= set(stages)
    # flattening must not change values or the state_dict
    sd = model.state_dict()
    for k, v in case["sd"].items():
        if k in sd and not k.endswith("pos_emb"):
            assert sd[k].shape == v.shape


def test_synthetic_batch_contract():
    from fourm.data.synthetic import synthetic_batch
    case = build_case("micro_swiglu")
    model = build_hip_model(case["cfg"], learned_pos=case["learned_pos"])
    B, n_in, n_out = 5, 20, 18
    md = synthetic_batch(model, B, n_in, n_out, device="cpu", seed=0)
    assert list(md) == sorted(md)
    tot_in = sum(int((~d["input_mask"]).sum()) for m, d in md.items() if m in model.encoder_embeddings)
    tot_out = sum(int((~d["target_mask"]).sum()) for m, d in md.items() if m in model.decoder_embeddings)
    assert tot_in == B * n_in and tot_out == B * n_out
    for m, d in md.items():
        assert d["input_mask"].dtype == torch.bool and d["decoder_attention_mask"].dtype == torch.int32
        assert not bool((~d["input_mask"] & ~d["target_mask"]).any()), m          # inputs and targets are disjoint
        if m in model.decoder_embeddings:
            assert torch.equal(d["decoder_attention_mask"].sum(1), (~d["target_mask"]).sum(1).int()), m
