"""Pre-tokenised shards (upstream README_DATA.md:9-37, save_vq_tokens.py:293-305) -> raw int16 batches (host) -> device ids -> device-side
masking: the host-to-device path once the masks are produced on the GPU."""
import io
import os
import tarfile

import numpy as np
import pytest
import torch

MODS = ["tok_rgb@224", "tok_depth@224"]


def write_shards(root, n_shards=2, per_shard=5, n_crops=3, drop=()):
    rng = np.random.default_rng(0)
    truth = {}
    for m in MODS:
        os.makedirs(os.path.join(root, m), exist_ok=True)
        for sh in range(n_shards):
            with tarfile.open(os.path.join(root, m, f"shard-{sh:05d}.tar"), "w") as tar:
                for i in range(per_shard):
                    key = f"{sh:02d}{i:04d}"
                    if (m, key) in drop:
                        continue
                    arr = rng.integers(0, 16384, (n_crops, 196)).astype(np.int16)         # the tokenizer's output dtype (save_vq_tokens.py:296)
                    truth[(m, key)] = arr
                    buf = io.BytesIO(); np.save(buf, arr); data = buf.getvalue()
                    info = tarfile.TarInfo(f"{key}.npy"); info.size = len(data)
                    tar.addfile(info, io.BytesIO(data))
    return truth


def test_aligned_batches_from_shards(tmp_path):
    from fourm.data.token_shards import iter_token_batches, raw_batch_nbytes, read_token_shard, shard_path
    truth = write_shards(str(tmp_path), drop={("tok_depth@224", "000002")})
    one = read_token_shard(shard_path(str(tmp_path), MODS[0], 1))
    assert list(one) == [f"01{i:04d}" for i in range(5)] and one["010003"].dtype == np.int16
    batches = list(iter_token_batches(str(tmp_path), MODS, [0, 1], batch_size=4, crop=1))
    assert len(batches) == 2 and all(b[m].shape == (4, 196) and b[m].dtype == np.int16 for b in batches for m in MODS)
    keys = [f"00{i:04d}" for i in (0, 1, 3, 4)] + [f"01{i:04d}" for i in range(5)]              # 000002 lacks a modality: dropped for all
    for j, k in enumerate(keys[:8]):
        for m in MODS:
            assert np.array_equal(batches[j // 4][m][j % 4], truth[(m, k)][1])
    assert raw_batch_nbytes(batches[0]) == 2 * 4 * 196 * 2
    tail = list(iter_token_batches(str(tmp_path), MODS, [0, 1], batch_size=4, crop=0, drop_last=False))
    assert len(tail) == 3 and tail[2][MODS[0]].shape == (1, 196)
    # random crops: the same crop index for every modality of a sample (the augmentations are aligned)
    rnd = next(iter_token_batches(str(tmp_path), MODS, [1], batch_size=5, rng=np.random.default_rng(3)))
    for j in range(5):
        c = [i for i in range(3) if np.array_equal(rnd[MODS[0]][j], truth[(MODS[0], f"01{j:04d}")][i])]
        assert len(c) == 1 and np.array_equal(rnd[MODS[1]][j], truth[(MODS[1], f"01{j:04d}")][c[0]])
    with pytest.raises(ValueError):
        bad = tmp_path / "bad" / MODS[0]
        os.makedirs(bad)
        with tarfile.open(bad / "shard-00000.tar", "w") as tar:
            buf = io.BytesIO(); np.save(buf, np.zeros((2, 196), dtype=np.int64)); data = buf.getvalue()
            info = tarfile.TarInfo("x.npy"); info.size = len(data); tar.addfile(info, io.BytesIO(data))
        read_token_shard(str(bad / "shard-00000.tar"))


@pytest.mark.gpu
def test_shards_to_device_masking(tmp_path):
    """ids cross as int16, are widened on the device and masked there: what reaches the model has the loader contract's layout."""
    from fourm.data.masking import DeviceUnifiedMasking
    from fourm.data.token_shards import iter_token_batches, raw_batch_to_device
    truth = write_shards(str(tmp_path))
    b = next(iter_token_batches(str(tmp_path), MODS, [0, 1], batch_size=8, crop=2))
    dev = raw_batch_to_device(b)
    for m in MODS:
        assert dev[m].shape == (8, 14, 14) and dev[m].dtype == torch.int64
        assert np.array_equal(dev[m].cpu().numpy().reshape(8, 196), b[m].astype(np.int64))
    info = {m: dict(type="img", max_tokens=196, min_tokens=0, input_alphas=[1.0], target_alphas=[1.0]) for m in MODS}
    um = DeviceUnifiedMasking(info, None, input_tokens_range=64, target_tokens_range=64, max_tries=16, device="cuda",
                              sentinel_to_id={0: 4, 1: 5}, pad_id=0)
    md = um(dev, generator=torch.Generator(device="cuda").manual_seed(0))
    n_in = sum((~md[m]["input_mask"]).sum(1) for m in MODS)
    assert int(n_in.max()) <= 64 and int(n_in.min()) > 0 and md[MODS[0]]["tensor"] is dev[MODS[0]]
