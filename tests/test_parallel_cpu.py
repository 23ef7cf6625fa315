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
