#!/usr/bin/env python3
"""Headline benchmark: 4M-B (fm_base_12e_12d_swiglu_nobias, 7 modalities) masked-modeling TRAIN step on
synthetic random-token multimodal batches, per-GPU batch 256, 128 input + 128 target tokens per sample
(BASELINE.json configs[1]; configs[2] = the same per GPU on N GPUs, weak scaling).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = select+embed -> encoder -> decoder -> heads/CE -> hand-written backward -> (RCCL gradient
mean, overlapped with backward) -> gradient norm -> fused AdamW.  Inputs are resident in HBM before
the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MOD7_IN = ["caption", "det", "rgb@224", "tok_clip@224", "tok_depth@224", "tok_normal@224", "tok_rgb@224", "tok_semseg@224"]
MOD7_OUT = [m for m in MOD7_IN if m != "rgb@224"]
# BASELINE.json configs[3]: cfgs/default/4m/data/cc12m+coyo+c4/main/mix_mod21_all2allmix_rgb2all_capT5bias_C4.yaml:7-8
MOD21_IN = ("caption-t5_caption-det-metadata-rgb@224-tok_rgb@224-tok_normal@224-tok_depth@224-tok_semseg@224-tok_clip@224-human_poses-"
            "tok_dinov2@224-tok_dinov2_global-tok_imagebind@224-tok_imagebind_global-tok_sam_edge@224-tok_canny_edge@224-color_palette-"
            "sam_instance").split("-")
MOD21_OUT = [m for m in MOD21_IN if m not in ("rgb@224", "t5_caption")]
MODS = {"mod7": (MOD7_IN, MOD7_OUT), "mod21": (MOD21_IN, MOD21_OUT)}
BF16_PEAK_TFLOPS = 2500.0          # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
F32_MATRIX_PEAK_TFLOPS = 157.3     # v_mfma_f32_*_f32 (f32 in / f32 accumulate), same guide: 1/16 of the bf16 rate
HBM_PEAK_GBS = 8000.0


def build_model(name, device, mods="mod7"):
    """The way the trainer builds it (run_training_4m.py:242-253 setup_modality_info, :354-387 get_model)."""
    from fourm.data.modality_info import MODALITY_INFO
    from fourm.utils import create_model
    mods_in, mods_out = MODS[mods]
    info = {m: dict(MODALITY_INFO[m]) for m in sorted(set(mods_in) | set(mods_out))}
    geom = {m: (info[m].get("input_size", 224), info[m].get("patch_size", 16)) for m in info}
    for m in info:
        if info[m]["type"] == "img":
            info[m]["max_tokens"] = (geom[m][0] // geom[m][1]) ** 2

    def embs(names, key):
        out = {}
        for m in names:
            if info[m].get(key) is None:
                continue
            kw = dict(patch_size=geom[m][1], image_size=geom[m][0]) if info[m]["type"] == "img" else {}
            out[m] = info[m][key](**kw)
        return out
    model = create_model(name, encoder_embeddings=embs(mods_in, "encoder_embedding"), decoder_embeddings=embs(mods_out, "decoder_embedding"),
                         modality_info=info)
    return model.to(device)


def train_flops_per_sample(model, n_in, n_out, head_vocabs):
    """Algorithmic FLOPs (2*MAC) of one sample's forward, x3 for the train step."""
    D, Le, Ld = model.dim, len(model.encoder), len(model.decoder)
    Hd = model.encoder[0].mlp.hidden_features
    gated = hasattr(model.encoder[0].mlp, "fc3")
    mlp = (6 if gated else 4) * D * Hd
    N, M = n_in, n_out
    f_enc = N * (6 * D * D + 2 * D * D + mlp) + 4 * N * N * D
    f_dec = M * (6 * D * D + 2 * D * D + mlp) + 4 * M * M * D + M * 4 * D * D + N * 4 * D * D + 4 * M * N * D
    f = Le * f_enc + Ld * f_dec + 2 * N * D * D
    n_enc = len(model.encoder_embeddings)
    for e in model.encoder_embeddings.values():                   # dense input projections on the rows actually selected (uniform split)
        if hasattr(e, "proj"):
            f += 2 * (N / n_enc) * e.proj.weight.shape[1] * D
        if hasattr(e, "emb_proj"):
            f += 2 * (N / n_enc) * e.emb_proj.weight.shape[1] * D
    f += sum(2 * D * v * (M / len(head_vocabs)) for v in head_vocabs)
    return 3.0 * f


class LaunchProfiler:
    """Per-launch kernel timing with events recorded on the stream the kernels are enqueued on."""

    def __init__(self):
        self.recs = []

    class _Ctx:
        def __init__(self, prof, name, flops, nbytes, tag):
            self.p, self.name, self.flops, self.nbytes, self.tag = prof, name, flops, nbytes, tag

        def __enter__(self):
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record(torch.cuda.current_stream())
            return self

        def __exit__(self, *exc):
            self.b.record(torch.cuda.current_stream())
            self.p.recs.append((self.name, self.flops, self.nbytes, self.a, self.b, self.tag))
            return False

    def launch(self, name, flops, nbytes, tag=""):
        return self._Ctx(self, name, flops, nbytes, tag)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        self.by_shape = {}
        for name, fl, nb, a, b, tag in self.recs:
            ms = a.elapsed_time(b)
            for key, table in ((name, agg), (f"{name} {tag}".strip(), self.by_shape)):
                d = table.setdefault(key, dict(ms=0.0, flops=0.0, bytes=0.0, n=0))
                d["ms"] += ms; d["flops"] += fl; d["bytes"] += nb; d["n"] += 1
        return agg

    def shape_table(self, steps):
        rows = sorted(self.by_shape.items(), key=lambda kv: -kv[1]["ms"])
        return [f"{k:46s} {d['n'] // steps:4d} launches/step {d['ms'] / steps:8.3f} ms/step {d['ms'] / d['n'] * 1e3:8.1f} us/launch"
                + (f" {d['flops'] / d['ms'] / 1e9:7.1f} TF/s" if d["flops"] else "") for k, d in rows]


KERNEL_REGEX = {   # profiler family -> regex on the demangled kernel name (7th template argument of gemm_nt = epilogue)
    "gemm_nt": r"(gemm_nt3_kernel<\d+, {epi}, |gemm_nt_kernel<\d+, \d+, \d+, \d+, \d+, \d+, {epi}, false|gemm_nt4_kernel<\d+, \d+, \d+, \d+, (true|false)>{nt4})",
    "gemm_tn": r"gemm_tn_kernel<\w+, false", "gemm_tn_multi": r"gemm_tn_multi_kernel", "attn_fwd": r"attn_fwd\w*_kernel", "attn_bwd": r"attn_bwd\w*_kernel",
    "layernorm_fwd": r"ln_fwd_kernel", "layernorm_bwd": r"ln_bwd_kernel",
}


def pmc_traffic(kernel_regex, timeout_s=240, worker_args=()):
    """HBM-side bytes per launch of one kernel from rocprofv3 PMC counters (MI355X_MICROARCH.md, HBM section): FETCH_SIZE
    and WRITE_SIZE are collected in SEPARATE passes (TCC slot limit) over two train steps in a child process, per
    dispatch; bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 reports half of a wide coalesced read; WRITE_SIZE is
    uncalibrated).  Returns (bytes_per_launch | None, detail dict)."""
    import csv, glob, re, shutil, subprocess, tempfile
    if shutil.which("rocprofv3") is None:
        return None, {"error": "rocprofv3 not on PATH"}
    rx, kb, n_launch = re.compile(kernel_regex), {}, {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="fm_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--",
               sys.executable, os.path.abspath(__file__), "--pmc-worker", *worker_args]
        try:
            subprocess.run(cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, timeout=timeout_s, capture_output=True)
            f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
            tot, n = 0.0, 0
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    if row["Counter_Name"] == counter and rx.search(row["Kernel_Name"]):
                        tot += float(row["Counter_Value"]); n += 1
            kb[counter], n_launch[counter] = tot / max(n, 1), n
        except Exception as e:
            return None, {"error": f"{counter} pass failed: {type(e).__name__}: {e}"}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if min(n_launch.values()) == 0:
        return None, {"error": f"no dispatch matched {kernel_regex}"}
    nbytes = (2.0 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024.0
    return nbytes, {"fetch_size_kb_per_launch": kb["FETCH_SIZE"], "write_size_kb_per_launch": kb["WRITE_SIZE"],
                    "launches_counted": n_launch["FETCH_SIZE"], "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 B, separate --pmc passes"}


def parity_record():
    """What the GPU suite measured for the benched arithmetic, read from the newest committed profiles/rNN_parity.jsonl (written by
    tests/parity_log.py during `pytest -m gpu`): the bf16 product path (THE path this bench times) and the fp32 verification mode of the
    same engine, both against the unmodified reference at the benched row count (4M-B mod7, batch 256) - so that nobody reads the fp32
    mode's 1e-6 and the bf16 path's ms/step as one run."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_parity.jsonl")))
    if not files:
        return None
    src = files[-1]
    rec = {"source": os.path.relpath(src, ROOT), "benched_path": "bf16 (GEMM operands bf16, fp32 accumulate: upstream's autocast arithmetic)"}
    try:
        for line in open(src):
            d = json.loads(line)
            if d.get("test") == "model.b256_golden" and d.get("mode") in ("bf16", "fp32"):
                k = "bf16_path" if d["mode"] == "bf16" else "fp32_verification_mode"
                rec[k + "_vs_fp32_reference_batch256"] = {"logits_rel": d.get("logits_head"), "loss_rel": d.get("loss"), "grad_norm_rel_worst": d.get("grad_norm_worst")}
            if d.get("test") == "model.logits" and d.get("case") == "b_mod7":
                pm = d.get("per_modality", {})
                rec["b_mod7_logits_worst_modality"] = {k: max(v[k] for v in pm.values()) for k in ("hip_vs_fp32", "hip_vs_bf16_oracle", "bf16_oracle_vs_fp32") if pm}
        rec["note"] = ("north_star tolerance 1e-3 (logits vs reference): met by the fp32 verification mode (same engine and launch sequence on fp32 kernels); the "
                       "timed bf16 path sits at upstream autocast's own distance from fp32 (bf16_oracle_vs_fp32)")
    except Exception as e:      # noqa: BLE001
        rec["error"] = f"{type(e).__name__}: {e}"
    return rec


def _reference_tree():
    """The unmodified upstream checkout the CPU baseline times when one is reachable: FOURM_UPSTREAM (the variable the package's own
    fall-through uses, fourm/_upstream.py) if it names a tree with fourm/models/fm.py, else /root/reference (the build container)."""
    for c in (os.environ.get("FOURM_UPSTREAM"), "/root/reference"):
        if c and os.path.isfile(os.path.join(c, "fourm", "models", "fm.py")):
            return os.path.abspath(c)
    return "/root/reference"


REFERENCE_TREE = _reference_tree()


def cpu_baseline(timeout_s=300, workload="train"):
    """Run the CPU leg in a child process (bounded wall time, its own thread pool): the UNMODIFIED upstream model through
    tests/golden/ref_stubs.py when /root/reference exists (the build container), else the oracle port (the GPU box)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--workload", workload], capture_output=True, text=True,
                           timeout=timeout_s, env={**os.environ, "HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        rec = json.loads(line)
        if rec.get("kind") == "port" and workload == "train":
            # measured in the build container, where both exist (tools/cpu_port_vs_reference.py): what the port's number means in
            # units of the unmodified reference on the same cores
            try:
                conv = json.load(open(os.path.join(ROOT, "profiles", "r05_cpu_port_vs_reference.json")))
                rec["port_vs_reference"] = {"port_over_reference": conv["port_over_reference"], "measured_on": conv["cpu"], "cores": conv["cores"],
                                            "source": "profiles/r05_cpu_port_vs_reference.txt"}
            except Exception:      # noqa: BLE001
                pass
        return rec
    except Exception as e:
        return {"value": None, "unit": "images/s" if workload in ("vq", "divae") else "tokens/s", "cores": os.cpu_count(), "kind": "port",
                "reference_available": os.path.isdir(REFERENCE_TREE),
                "sample": f"CPU leg did not finish within {timeout_s}s ({type(e).__name__})"}


def extra_record(argv, timeout_s):
    """One more workload as a sub-record of the headline line: this script re-run with ``argv`` in a child process."""
    import subprocess
    t0 = time.perf_counter()
    try:
        env = dict(os.environ)
        if env.get("BENCH_SHAPE_TABLE"):      # the child writes its own per-shape table NEXT TO the headline's, not over it
            base, ext = os.path.splitext(env["BENCH_SHAPE_TABLE"])
            tag = "_".join(x.lstrip("-") for x in argv[:2])
            env["BENCH_SHAPE_TABLE"] = f"{base}_{tag}{ext}"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-extras", *argv], capture_output=True, text=True, timeout=timeout_s, env=env)
        rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        rec["wall_s"] = round(time.perf_counter() - t0, 1)
        return rec
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"[:300], "argv": argv, "wall_s": round(time.perf_counter() - t0, 1)}


def _cpu_model_name():
    try:
        return [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        return ""


def cpu_baseline_reference_worker(batch=8, steps=3):
    """The UNMODIFIED upstream FourM (imported from /root/reference through the inert stubs of tests/golden/ref_stubs.py), fp32,
    forward + backward + torch.optim.AdamW on the 4M-B mod7 shapes: SURVEY §8d's CPU baseline.  Only where the tree exists."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import ref_stubs
    ref_stubs.install()
    # this child process times UPSTREAM's package: `fourm` must resolve to the reference tree, not to ml-4m_amd/fourm
    sys.path[:] = [p for p in sys.path if os.path.abspath(p) != os.path.join(ROOT, "ml-4m_amd")]
    for name in [n for n in sys.modules if n == "fourm" or n.startswith("fourm.")]:
        del sys.modules[name]
    sys.path.insert(0, REFERENCE_TREE)
    from tests.golden.make_golden import upstream_model, clone_mod_dict
    import fourm.models.fm as ref_fm
    assert os.path.abspath(ref_fm.__file__).startswith(REFERENCE_TREE), ref_fm.__file__
    from oracle import fourm_oracle as O
    threads = min(64, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    cfg = O.named_cfg("base", O.mod7_specs())
    model = upstream_model(cfg, True, False, ())
    model.load_state_dict(O.seeded_state_dict(cfg, seed=0), strict=True)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
    md = O.synthetic_mod_dict(cfg, batch, 128, 128, seed=0)
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        loss, _ = model(clone_mod_dict(md), 128, 128)
        loss.sum().backward()
        opt.step(); opt.zero_grad()
        times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": batch * 256 / t, "unit": "tokens/s", "cores": threads, "host_cores": os.cpu_count(), "kind": "reference",
            "reference_available": True,
            "sample": f"unmodified upstream fourm.models.fm.FourM (fp32), 4M-B mod7, batch {batch}, 128+128 tokens, fwd+bwd+AdamW, median of "
                      f"{steps} steps after 1 warm-up, {t:.2f} s/step", "cpu": _cpu_model_name()}


def cpu_baseline_vq_worker(batch=8, steps=3):
    """The VQ oracle port (fp32 restatement of VQ.encode, pinned to upstream by tests/golden/make_golden_vq.py) on the host cores."""
    from oracle import vq_oracle as V
    threads = min(64, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    cfg = V.vq_cfg("vit_b_enc", image=224, patch=16, codebook=16384, post_mlp=True)
    sd = V.seeded_vq_state_dict(cfg, seed=0)
    x = V.synthetic_images(cfg, batch, seed=0)
    times = []
    with torch.no_grad():
        for it in range(steps + 1):
            t0 = time.perf_counter()
            V.vq_encode(sd, cfg, x)
            times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": batch / t, "unit": "images/s", "cores": threads, "host_cores": os.cpu_count(), "kind": "port",
            "reference_available": os.path.isdir(REFERENCE_TREE),
            "sample": f"oracle fp32 PyTorch port of VQ.encode (ViT-B/16 + 16384 x 32 cosine codebook), batch {batch} 224^2 images, median of "
                      f"{steps} passes after 1 warm-up, {t:.2f} s/batch", "cpu": _cpu_model_name()}


def cpu_baseline_divae_worker(evals=2, ddim_steps=25):
    """The DiVAE oracle port (fp32 restatement of the unet_patched decoder, pinned to upstream by tests/golden/make_golden_divae.py) on the host
    cores: UNet evaluations of ONE image; a decode is ddim_steps of them (the scheduler steps are negligible)."""
    from oracle import divae_oracle as D
    threads = min(64, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    cfg = D.unet_patched_cfg()
    P = D.seeded_unet_state_dict(cfg, seed=0)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 3, cfg.image_size, cfg.image_size, generator=g)
    cond = torch.randn(1, cfg.cond_channels, 14, 14, generator=g)
    times, t_start = [], time.perf_counter()
    with torch.no_grad():
        while len(times) < evals + 1 or (time.perf_counter() - t_start < 12.0 and len(times) < 61):      # ~12 s of CPU work
            t0 = time.perf_counter()
            D.unet_forward(P, cfg, x, 500, cond)
            times.append(time.perf_counter() - t0)
    evals = len(times) - 1
    t = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": 1.0 / (t * ddim_steps), "unit": "images/s", "cores": threads, "host_cores": os.cpu_count(), "kind": "port",
            "reference_available": os.path.isdir(REFERENCE_TREE),
            "sample": f"oracle fp32 PyTorch port of the unet_patched decoder: median of {evals} evaluations of ONE 224^2 image after 1 warm-up, "
                      f"{t:.2f} s per evaluation, x {ddim_steps} DDIM steps per decoded image", "cpu": _cpu_model_name()}


def cpu_baseline_mod21_worker(batch=1, steps=2):
    """BASELINE configs[3] on the host cores: the oracle port of 4M-L mod21 (24 + 24 blocks, 1.27 G parameters, 19 input / 17 target
    modalities, 256 + 256 tokens), forward + backward + AdamW, batch 1.  (The unmodified upstream model where its tree exists.)"""
    from oracle import fourm_oracle as O
    threads = min(64, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    cfg = O.named_cfg("large", O.mod21_specs())
    sd = O.seeded_state_dict(cfg, seed=0, learned_pos=O.MOD21_LEARNED_POS)
    kind = "port"
    md = O.synthetic_mod_dict(cfg, batch, 256, 256, seed=0)
    if os.path.isdir(REFERENCE_TREE):
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import ref_stubs
        ref_stubs.install()
        sys.path[:] = [p for p in sys.path if os.path.abspath(p) != os.path.join(ROOT, "ml-4m_amd")]
        for name in [n for n in sys.modules if n == "fourm" or n.startswith("fourm.")]:
            del sys.modules[name]
        sys.path.insert(0, REFERENCE_TREE)
        from tests.golden.make_golden import upstream_model, clone_mod_dict
        model = upstream_model(cfg, True, False, O.MOD21_LEARNED_POS)
        model.load_state_dict(sd, strict=True)
        model.train()
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
        kind = "reference"

        def one():
            loss, _ = model(clone_mod_dict(md), 256, 256)
            loss.sum().backward()
            opt.step(); opt.zero_grad()
    else:
        def is_buffer(k):
            return ("pos_emb" in k and not any(k.startswith(f"{side}_embeddings.{m}.") for side in ("encoder", "decoder") for m in O.MOD21_LEARNED_POS)) \
                or ("norm" in k and k.endswith(".bias"))
        P = {k: v.clone().requires_grad_(v.is_floating_point() and not is_buffer(k)) for k, v in sd.items()}
        for m in cfg.mods:
            if m.in_enc and m.in_dec:
                P[f"decoder_embeddings.{m.name}.mod_emb"] = P[f"encoder_embeddings.{m.name}.mod_emb"]
            if m.in_dec:
                P[f"decoder_embeddings.{m.name}.to_logits.weight"] = P[f"decoder_embeddings.{m.name}.token_emb.weight"]
        leaves = list({id(v): v for v in P.values() if v.requires_grad}.values())
        opt = torch.optim.AdamW(leaves, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
        order = [m.name for m in cfg.mods if m.in_dec]

        def one():
            loss, _ = O.fourm_forward(P, cfg, md, 256, 256, order)
            loss.sum().backward()
            opt.step(); opt.zero_grad()
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        one()
        times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": batch * 512 / t, "unit": "tokens/s", "cores": threads, "host_cores": os.cpu_count(), "kind": kind,
            "reference_available": kind == "reference",
            "sample": f"{'unmodified upstream FourM' if kind == 'reference' else 'oracle fp32 PyTorch port'}, 4M-L mod21, batch {batch}, 256+256 tokens, "
                      f"fwd+bwd+AdamW, median of {steps} steps after 1 warm-up, {t:.2f} s/step", "cpu": _cpu_model_name()}


def cpu_baseline_worker(batch=8, steps=3):
    """The oracle port (plain fp32 PyTorch restatement of the upstream model) timed on the host cores:
    forward + backward + AdamW on the same 4M-B mod7 shapes, small batch.  A reported baseline only.
    (Used where /root/reference does not exist: the GPU box.  The port is pinned to upstream by tests/golden/make_golden.py.)"""
    from oracle import fourm_oracle as O
    threads = min(64, os.cpu_count() or 1)        # more threads than this only adds contention at batch 8
    torch.set_num_threads(threads)
    cfg = O.named_cfg("base", O.mod7_specs())
    sd = O.seeded_state_dict(cfg, seed=0)
    def is_buffer(k):      # fixed sin-cos tables and the zero bias buffers of the bias-free LayerNorms
        return "pos_emb" in k or ("norm" in k and k.endswith(".bias"))
    P = {k: v.clone().requires_grad_(v.is_floating_point() and not is_buffer(k)) for k, v in sd.items()}
    for m in cfg.mods:
        if m.in_enc and m.in_dec:
            P[f"decoder_embeddings.{m.name}.mod_emb"] = P[f"encoder_embeddings.{m.name}.mod_emb"]
        if m.in_dec:
            P[f"decoder_embeddings.{m.name}.to_logits.weight"] = P[f"decoder_embeddings.{m.name}.token_emb.weight"]
    leaves = list({id(v): v for v in P.values() if v.requires_grad}.values())
    opt = torch.optim.AdamW(leaves, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
    md = O.synthetic_mod_dict(cfg, batch, 128, 128, seed=0)
    order = [m.name for m in cfg.mods if m.in_dec]
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        loss, _ = O.fourm_forward(P, cfg, md, 128, 128, order)
        loss.sum().backward()
        opt.step(); opt.zero_grad()
        times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": batch * 256 / t, "unit": "tokens/s", "cores": threads, "host_cores": os.cpu_count(), "kind": "port",
            "reference_available": False,
            "sample": f"oracle fp32 PyTorch port (the reference tree is not on this machine), 4M-B mod7, batch {batch}, 128+128 tokens, "
                      f"fwd+bwd+AdamW, median of {steps} steps after 1 warm-up, {t:.2f} s/step", "cpu": _cpu_model_name()}


def main_vq(a):
    """BASELINE.json configs[4]: images/s of VQ.encode (patchify -> ViT-B/16 -> fp32 post-MLP -> 1x1 projection -> cosine code
    search) at the reference's sub-batch of 64 (save_vq_tokens.py:387-388).  Replicas only across GPUs (no collective)."""
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        print(json.dumps({"error": "no GPU visible: the tokenizer has no CPU implementation"}))
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from fourm.hip import ops
    from fourm.vq import VQ
    torch.manual_seed(rank)
    batch = a.user_batch or 64
    model = VQ(image_size=224, enc_type="vit_b_enc", patch_size=16, post_mlp=True, codebook_size=16384, latent_dim=32, norm_codes=True,
               sync_codebook=False).to(dev).eval()
    xs = [torch.rand(batch, 3, 224, 224, device=dev) * 2 - 1 for _ in range(2)]

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    if a.pmc_worker:
        for i in range(2):
            model.tokenize(xs[i % 2])
        fence()
        return
    from fourm.vq import tokenize_sub_batches
    n_streams = a.vq_streams
    for i in range(a.warmup):
        model.tokenize(xs[i % 2])
    tokenize_sub_batches(model, [xs[i % 2] for i in range(2 * n_streams + 1)], n_streams)       # (allocates the per-stream scratch sets)
    fence()
    # one sub-batch at a time (the number of rounds 1 - 5) ...
    t0 = time.perf_counter()
    for i in range(a.steps):
        tok = model.tokenize(xs[i % 2])
    fence()
    dt1 = time.perf_counter() - t0
    # ... and the timed region: upstream's loop over sub-batches of 64 (save_vq_tokens.py:262-288) with two of them in flight on two HIP streams
    t0 = time.perf_counter()
    toks = tokenize_sub_batches(model, [xs[i % 2] for i in range(a.steps)], n_streams)
    fence()
    dt = time.perf_counter() - t0
    tok = toks[-1]
    t = torch.tensor([dt, dt1], device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, dt1 = float(t[0]), float(t[1])
    flops_img = 37.0e9                      # 36.8 GFLOP ViT + 0.21 GFLOP code search per image
    value = world * batch * a.steps / dt
    out = {"metric": "images/sec (RGB VQ tokenizer encode+quantize, whole job)", "value": value, "unit": "images/s", "n_gpus": world,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": "RGB VQ tokenizer (ViT-B/16 encoder 90.5M params, 224^2 -> 14x14 codes, 16384 x 32 cosine codebook): "
                                  "VQ.tokenize on uniform [-1,1] images", "per_gpu_batch": batch, "global_batch": batch * world,
                      "parallelism": f"replicas x{world}; {n_streams} sub-batches of {batch} in flight on {n_streams} HIP streams (fourm.vq.tokenize_sub_batches)"},
           "one_sub_batch_in_flight_images_per_s": world * batch * a.steps / dt1,
           "codes_per_sec": value * 196, "mfu": flops_img * value / world / (BF16_PEAK_TFLOPS * 1e12),
           "arithmetic": "ViT blocks bf16 operands / fp32 accumulate; post-MLP, 1x1 projection and code search fp32"}
    if not a.no_kernel_profile and rank == 0:
        prof = LaunchProfiler()
        ops.set_profiler(prof)
        for i in range(2):
            model.tokenize(xs[i % 2])
        ops.set_profiler(None)
        agg = prof.summary()
        tot_ms = sum(d["ms"] for d in agg.values()) or 1.0
        name, d = max(agg.items(), key=lambda kv: kv[1]["ms"])
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["flops"] else 0.0
        fam, _, epi = name.partition("/epi")
        traffic, detail = (None, None)
        if world == 1 and not a.no_traffic and fam in KERNEL_REGEX:
            traffic, detail = pmc_traffic(KERNEL_REGEX[fam].format(epi=epi or "0", nt4="" if (epi or "0") == "0" else "NEVER"), worker_args=["--workload", "vq", "--batch", str(batch)])
        # the fp32 tail (tanh post-MLP, upstream disables autocast there) runs on v_mfma_f32_32x32x2_f32: exact fp32 at 1/16 of the bf16
        # rate - its launches are priced against the fp32 matrix peak of MI355X_MICROARCH.md, every other kernel against the bf16 peak
        f32_kernel = name.startswith("gemm_f32")
        peak = F32_MATRIX_PEAK_TFLOPS if f32_kernel else BF16_PEAK_TFLOPS
        out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                           "kernel": f"{name} ({'csrc/fp32_verify.hip, fp32 MFMA: peak = the fp32 matrix rate' if f32_kernel else 'csrc/gemm_nt3.hip: gemm_nt3_kernel<TW, EPI_BF16, SPLIT, STG, BIAS=true> (biased Linears of the ViT blocks)' if name == 'gemm_nt/epi0' else 'csrc/gemm.hip: gemm_nt_kernel (GELU / fp32-output epilogues)' if name.startswith('gemm_nt') else 'csrc'})",
                           "traffic": traffic, "traffic_detail": detail, "launches_per_step": d["n"] // 2,
                           "avg_launch_us": 1e3 * d["ms"] / d["n"], "share_of_timed_kernels": d["ms"] / tot_ms,
                           "algorithmic_bytes_per_launch": d["bytes"] / d["n"] if d["bytes"] else None}
        out["kernel_breakdown_ms_per_step"] = {k: round(v["ms"] / 2, 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        print(json.dumps(out), file=sys.stderr)
        out["cpu_baseline"] = cpu_baseline(workload="vq")
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def main_divae(a):
    """the diffusion detokenizer DiVAE (vqvae.py:498-764) - images/s of decode_tokens (token grid -> codebook embedding ->
    25 DDIM steps of the conditional unet_patched decoder, 196 M parameters, with dynamic thresholding) at batch 8.  Replicas only."""
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        print(json.dumps({"error": "no GPU visible: the detokenizer has no CPU implementation"}))
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from fourm.hip import ops
    from fourm.vq import DiVAE
    torch.manual_seed(rank)
    batch, ddim = a.user_batch or 8, 25
    model = DiVAE(image_size=224, n_channels=3, enc_type="vit_b_enc", patch_size=16, codebook_size=16384, latent_dim=32, post_mlp=True, norm_codes=True,
                  scheduler="ddim", prediction_type="sample", beta_schedule="linear", sync_codebook=False)
    for p in model.decoder.parameters():          # upstream zero-initialises the block tails: give every GEMM real operands
        if float(p.abs().max()) == 0:
            torch.nn.init.normal_(p, std=0.02)
    model = model.to(dev).eval()
    toks = [torch.randint(0, 16384, (batch, 14, 14), device=dev) for _ in range(2)]
    gen = torch.Generator().manual_seed(rank)

    def decode(i):
        return model.decode_tokens(toks[i % 2], timesteps=ddim, generator=gen, verbose=False)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    if a.pmc_worker:          # two UNet evaluations (a decode is 25 of them)
        quant = model.tokens_to_embedding(toks[0]).float()
        x = torch.randn(batch, 3, 224, 224, device=dev)
        for _ in range(2):
            model.decoder(x, 500, quant)
        fence()
        return
    from fourm.vq import decode_token_batches
    n_streams = a.vq_streams
    steps, warmup = min(a.steps, 7), min(a.warmup, 1)
    for i in range(warmup):
        decode(i)
    decode_token_batches(model, [toks[i % 2] for i in range(n_streams + 1)], n_streams, timesteps=ddim, generator=gen, verbose=False)     # (per-stream scratch)
    fence()
    t0 = time.perf_counter()                   # one decode at a time ...
    for i in range(2):
        img = decode(i)
    fence()
    dt1 = (time.perf_counter() - t0) / 2
    t0 = time.perf_counter()                   # ... and the timed region: `steps` decodes of batch 8, n_streams of them in flight
    imgs = decode_token_batches(model, [toks[i % 2] for i in range(steps)], n_streams, timesteps=ddim, generator=gen, verbose=False)
    fence()
    dt = time.perf_counter() - t0
    img = imgs[-1]
    t = torch.tensor([dt, dt1], device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, dt1 = float(t[0]), float(t[1])
    flops_img_eval = 190.0e9               # 3x3 / 1x1 convolutions + attention of one evaluation per image
    value = world * batch * steps / dt
    out = {"metric": "images/sec (DiVAE diffusion detokenizer: decode_tokens, 25 DDIM steps, whole job)", "value": value, "unit": "images/s", "n_gpus": world,
           "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": "DiVAE detokenizer (unet_patched decoder 196 M parameters, 56 x 56 patch grid, 14 x 14 x 32 conditioning, DDIM 25 steps, "
                                  "dynamic thresholding): DiVAE.decode_tokens on random token grids", "per_gpu_batch": batch, "global_batch": batch * world,
                      "parallelism": f"replicas x{world}; {n_streams} decodes of batch {batch} in flight on {n_streams} HIP streams (fourm.vq.decode_token_batches)"},
           "ms_per_unet_evaluation_one_decode_in_flight": 1e3 * dt1 / ddim, "one_decode_in_flight_images_per_s": world * batch / dt1,
           "finite": bool(torch.isfinite(img).all()),
           "mfu": flops_img_eval * ddim * value / world / (BF16_PEAK_TFLOPS * 1e12),
           "arithmetic": "convolutions as im2col + bf16 GEMMs (fp32 accumulate), bf16 feature maps, GroupNorm / attention softmax / scheduler steps fp32"}
    if not a.no_kernel_profile and rank == 0:
        prof = LaunchProfiler()
        quant = model.tokens_to_embedding(toks[0]).float()
        x = torch.randn(batch, 3, 224, 224, device=dev)
        import fourm.vq.models.unet.unet as unet_mod
        graph_on, unet_mod.UNET_GRAPH = unet_mod.UNET_GRAPH, False          # (the launch events ride on the eager launch sequence, not on graph replays)
        ops.set_profiler(prof)
        for _ in range(2):
            model.decoder(x, 500, quant)
        ops.set_profiler(None)
        unet_mod.UNET_GRAPH = graph_on
        agg = prof.summary()
        tot_ms = sum(d["ms"] for d in agg.values()) or 1.0
        gemms = {k: v for k, v in agg.items() if v["flops"]}
        name, d = max(gemms.items(), key=lambda kv: kv[1]["ms"])
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
        fam, _, epi = name.partition("/epi")
        traffic, detail = (None, None)
        if world == 1 and not a.no_traffic and fam in KERNEL_REGEX:
            traffic, detail = pmc_traffic(KERNEL_REGEX[fam].format(epi=epi or "0", nt4="" if (epi or "0") == "0" else "NEVER"), worker_args=["--workload", "divae", "--batch", str(batch)])
        out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / BF16_PEAK_TFLOPS,
                           "kernel": f"{name} (the convolutions' GEMMs of one UNet evaluation: csrc/gemm_nt3.hip / gemm.hip on im2col rows, M = batch x 56^2 ... batch x 7^2)",
                           "traffic": traffic, "traffic_detail": detail, "launches_per_evaluation": d["n"] // 2,
                           "avg_launch_us": 1e3 * d["ms"] / d["n"], "share_of_timed_kernels": d["ms"] / tot_ms,
                           "algorithmic_bytes_per_launch": d["bytes"] / d["n"] if d["bytes"] else None}
        out["kernel_breakdown_ms_per_evaluation"] = {k: round(v["ms"] / 2, 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
        if os.environ.get("BENCH_SHAPE_TABLE"):
            with open(os.environ["BENCH_SHAPE_TABLE"], "w") as fh:
                fh.write("\n".join(prof.shape_table(2)) + "\n")
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        print(json.dumps(out), file=sys.stderr)
        out["cpu_baseline"] = cpu_baseline(workload="divae")
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="train", choices=["train", "vq", "train21", "divae"], help="train = the headline 4M train step; vq = BASELINE "
                    "configs[4]: RGB VQ tokenizer (ViT-B/16, 224^2 -> 14 x 14 codes, 16384 x 32 codebook) encode + quantize, batch 64; divae = SURVEY §8 "
                    "row f4: the diffusion detokenizer, decode_tokens with 25 DDIM steps at batch 8")
    ap.add_argument("--mods", default="mod7", choices=sorted(MODS), help="mod7 = BASELINE configs[1] (4M-B, batch 256, 128+128 tokens); "
                    "mod21 = configs[3] (4M-L, 19 / 17 modalities, batch 64, 256+256 tokens)")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: 256 for mod7, 64 for mod21)")
    ap.add_argument("--model", default=None)
    ap.add_argument("--n-in", type=int, default=None)
    ap.add_argument("--n-out", type=int, default=None)
    ap.add_argument("--masking", default="uniform", choices=["uniform", "dirichlet"], help="uniform = SURVEY §8d's synthetic budgets (the headline); "
                    "dirichlet = the batches come from the device-side masking pipeline (Dirichlet token budgets, image masks, span masking: "
                    "fourm.data.masking.DeviceUnifiedMasking), and the record carries the producer's time per batch")
    ap.add_argument("--vq-streams", type=int, default=2, help="--workload vq: sub-batches of 64 in flight (HIP streams) in the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) | gloo (test hook: several ranks on one GPU)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc passes that fill roofline.traffic")
    ap.add_argument("--no-extras", action="store_true", help="headline record only (default for non-default workloads): no extra.vq / extra.mod21 sub-records")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-kind", default="auto", choices=["auto", "port", "reference"], help=argparse.SUPPRESS)
    ap.add_argument("--pmc-worker", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    a.user_batch = a.batch
    # the sub-records of BASELINE configs[3] / [4] ride on the DEFAULT invocation only (N = 1, mod7, no size overrides)
    a.extras = not a.no_extras and a.workload == "train" and a.mods == "mod7" and a.batch is None and a.model is None and not a.pmc_worker and a.masking == "uniform" \
        and not a.cpu_baseline_worker and a.gpus == 1 and a.dist_backend == "nccl" and not a.no_cpu_baseline
    dflt = {"mod7": ("fm_base_12e_12d_swiglu_nobias", 256, 128), "mod21": ("fm_large_24e_24d_swiglu_nobias", 64, 256)}[a.mods]
    a.model = a.model or dflt[0]
    a.batch = a.batch or dflt[1]
    a.n_in, a.n_out = a.n_in or dflt[2], a.n_out or dflt[2]
    if a.cpu_baseline_worker:
        if a.workload == "vq":
            print(json.dumps(cpu_baseline_vq_worker()))
        elif a.workload == "divae":
            print(json.dumps(cpu_baseline_divae_worker()))
        elif a.workload == "train21":
            print(json.dumps(cpu_baseline_mod21_worker()))
        elif a.cpu_baseline_kind == "reference" or (a.cpu_baseline_kind == "auto" and os.path.isdir(REFERENCE_TREE)):
            print(json.dumps(cpu_baseline_reference_worker()))
        else:
            print(json.dumps(cpu_baseline_worker()))
        return
    if a.workload == "vq":
        return main_vq(a)
    if a.workload == "divae":
        return main_divae(a)

    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        print(json.dumps({"error": "no GPU visible: the 4M hot path has no CPU implementation"}))
        sys.exit(2)
    if a.dist_backend != "nccl":
        local = local % torch.cuda.device_count()          # test hook: ranks may share a device under gloo
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.dist_backend == "nccl":
            from fourm.parallel import cap_collective_channels
            cap_collective_channels()          # before the communicator exists: RCCL's workgroups fit the CUs DataParallel keeps free
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(a.dist_backend, rank=rank, world_size=world)
    if a.gpus != world:
        if rank == 0:
            print(f"[bench] --gpus {a.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    comm_check = None
    if world > 1:
        # self-check of the launch: a sum of ones over the process group = the ranks the collectives really span (a mis-launched
        # N x dp1 job would print WORLD_SIZE = N and still reduce over 1), and a sum of device ordinals shows N DISTINCT GPUs under RCCL
        ones = torch.tensor([1.0, float(local)], device=dev)
        dist.all_reduce(ones)
        comm_check = {"backend": dist.get_backend(), "rccl_ranks": int(ones[0].item()), "sum_local_ranks": int(ones[1].item()),
                      "expected_sum_local_ranks": world * (world - 1) // 2 if a.dist_backend == "nccl" else None}
        try:
            comm_check["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:      # noqa: BLE001
            comm_check["rccl_version"] = f"unavailable ({type(e).__name__})"
        assert comm_check["rccl_ranks"] == world, f"collectives span {comm_check['rccl_ranks']} ranks, WORLD_SIZE says {world}"

    from fourm.data.synthetic import synthetic_batch
    from fourm.hip import ops
    from fourm.parallel import DataParallel
    from fourm.utils.optim_factory import FusedAdamW, get_parameter_groups

    torch.manual_seed(0)
    model = build_model(a.model, dev, a.mods).train()
    dp = DataParallel(model) if world > 1 else None
    if dp is not None:
        dp.time_exchange = True              # events around GradReducer.finish(): the exchange time the backward did not hide
    fwd = dp if dp is not None else model
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        groups = get_parameter_groups(model, weight_decay=0.05, skip_list=model.no_weight_decay())
    opt = FusedAdamW(groups, lr=1e-4 * a.batch * world / 256, betas=(0.9, 0.95), eps=1e-8)
    masking_ms = None
    if a.masking == "dirichlet":      # SURVEY §8d "realistic variant": the loader's UnifiedMasking, run on the device over the whole batch
        from fourm.data.synthetic import device_masked_batch, device_masking_for
        um = device_masking_for(model, a.n_in, a.n_out, device=dev)
        gen = torch.Generator(device=dev).manual_seed(1000 * rank)
        batches = [device_masked_batch(model, um, a.batch, device=dev, generator=gen) for _ in range(2)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            device_masked_batch(model, um, a.batch, device=dev, generator=gen)
        torch.cuda.synchronize()
        masking_ms = (time.perf_counter() - t0) / 5 * 1e3
    else:
        batches = [synthetic_batch(model, a.batch, a.n_in, a.n_out, device=dev, seed=1000 * rank + i) for i in range(2)]
    import random
    random.seed(rank)

    def step(i):
        loss, mod_loss = fwd(batches[i % 2], a.n_in, a.n_out, loss_type="mod")
        loss.backward()
        norm = opt.fused_grad_norm(lazy=True)        # as the trainer without clipping (NativeScaler): the norm rides on the AdamW pass
        opt.step()
        opt.zero_grad(set_to_none=True)
        return loss, norm

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if a.pmc_worker:            # child of pmc_traffic(): two plain steps under rocprofv3 --pmc, nothing printed
        for i in range(2):
            step(i)
        fence()
        return
    for i in range(a.warmup):
        loss, norm = step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss, norm = step(i)
    fence()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t)
    last_loss = float(loss.detach())
    exch = None
    if dp is not None and dp._reducer is not None:
        ms = dp._reducer.exposed_ms()[-a.steps:]
        exch = {"exposed_ms_per_step": sum(ms) / len(ms) if ms else None, "collectives_per_step": dp._reducer.n_collectives,
                "wire_mb_per_step": dp._reducer.bytes_on_wire / 2 ** 20}

    tokens_per_step = world * a.batch * (a.n_in + a.n_out)
    value = tokens_per_step * a.steps / dt
    head_vocabs = [e.vocab_size for e in model.decoder_embeddings.values()]
    flops_step = train_flops_per_sample(model, a.n_in, a.n_out, head_vocabs) * a.batch
    n_params = sum(p.numel() for p in {id(p): p for p in model.parameters()}.values())
    family = {"fm_tiny": "4M-Ti", "fm_small": "4M-S", "fm_base": "4M-B", "fm_large": "4M-L", "fm_xlarge": "4M-XL"}.get(a.model.rsplit("_", 4)[0], a.model)
    out = {
        "metric": f"multimodal tokens/sec ({family} train step, whole job)", "value": value, "unit": "tokens/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{family} {a.mods} ({a.model}, {n_params / 1e6:.1f}M params, {len(model.encoder_embeddings)} input / "
                               f"{len(model.decoder_embeddings)} target modalities) masked-modeling train step, 224^2 token grids, "
                               f"{a.n_in} input + {a.n_out} target tokens/sample, AdamW", "model": a.model, "per_gpu_batch": a.batch,
                   "global_batch": a.batch * world, "seq_len": a.n_in + a.n_out, "parallelism": f"dp{world}"},
        "tokens_per_sec_per_gpu": value / world,
        "mfu": flops_step * a.steps / dt / (BF16_PEAK_TFLOPS * 1e12),
        "algorithmic_tflop_per_step_per_gpu": flops_step / 1e12, "final_loss": last_loss,
    }
    if dp is not None:         # how the gradients travelled (fourm.parallel.DataParallel): nothing here changes the work counted in `value`
        out["config"]["data_parallel"] = {"exchange": dp._exchange_mode, "reserved_cus": dp._reserved_cus, "min_launch_mb": dp._min_launch_mb
                                          if dp._min_launch_mb != float("inf") else None, "wire_dtype": "bf16" if dp._wire is not None else "fp32",
                                          "NCCL_MAX_NCHANNELS": os.environ.get("NCCL_MAX_NCHANNELS"), **(comm_check or {}), **(exch or {}),
                                          "note": "exposed_ms_per_step = rank 0's compute stream between entering GradReducer.finish() and holding every "
                                                  "reduced slice (collectives still running or not yet started + unpack), averaged over the timed steps"}
    if masking_ms is not None:
        out["data"] = "synthetic modalities masked on the device (Dirichlet token budgets, image masks, span masking)"
        out["masking_ms_per_batch"] = masking_ms          # the producer, outside the timed region (batches are resident when it starts)

    # ---- dominant kernel vs its roofline, measured live with events on the launch stream (rank 0) ----------
    if not a.no_kernel_profile:
        # every rank runs the two extra steps (they contain the gradient collectives); only rank 0 records
        prof = LaunchProfiler() if rank == 0 else None
        ops.set_profiler(prof)
        for i in range(2):
            step(i)
        ops.set_profiler(None)
        fence()
    if rank == 0 and not a.no_kernel_profile:
        agg = prof.summary()
        tot_ms = sum(d["ms"] for d in agg.values()) or 1.0
        symbol = {"gemm_nt": "FAMILY of template instantiations: gemm_nt4_kernel<TW=384, TX=256, 2, 2, STG=true> (csrc/gemm_nt4.hip, round 6: 4 waves x 512 registers, "
                             "256 x 384 tiles, accumulators in hand-named AGPRs) where N % 384 == 0 (N = 768, 1536, 2304), gemm_nt3_kernel<TW=192|256, EPI={epi}, SPLIT=true, "
                             "STG=true> (csrc/gemm_nt3.hip: 8 waves, lock-step 256 x 256 / 192 x 256 tiles) elsewhere; outputs staged through LDS into whole-line stores - "
                             "every dense bf16 Linear of the trunk, forward and dX",
                  "gemm_tn": "gemm_tn_kernel<true,false,128,256,2,4,64,3,PP=true>  (csrc/gemm.hip)",
                  "gemm_tn_multi": "gemm_tn_multi_kernel<MASKED=false>  (csrc/gemm.hip; all dW GEMMs of a layer per launch)",
                  "attn_fwd": "attn_fwd128_kernel<MASK> at 128 x 128 tokens, attn_fwdt_kernel<MASK> at multiples of 128 up to 512 keys, attn_fwd_kernel<true,MASK> otherwise", "attn_bwd": "attn_bwd128_kernel<MASK> (key padding) / attn_bwd128o_kernel<MASK> (decoder mask) at 128 x 128 tokens, attn_bwd_kernel<true,MASK,..> otherwise",
                  "layernorm_fwd": "ln_fwd_kernel<bf16,3>", "layernorm_bwd": "ln_bwd_kernel<3>"}
        name, d = max(agg.items(), key=lambda kv: kv[1]["ms"])
        fam, _, epi = name.partition("/epi")
        common = {"kernel": symbol.get(fam, fam).format(epi=epi or "0"), "traffic": None, "launches_per_step": d["n"] // 2,
                  "avg_launch_us": 1e3 * d["ms"] / d["n"], "share_of_timed_kernels": d["ms"] / tot_ms,
                  "algorithmic_bytes_per_launch": d["bytes"] / d["n"] if d["bytes"] else None}
        if world == 1 and not a.no_traffic and fam in KERNEL_REGEX:
            same_job = ["--mods", a.mods, "--model", a.model, "--batch", str(a.batch), "--n-in", str(a.n_in), "--n-out", str(a.n_out)]
            common["traffic"], common["traffic_detail"] = pmc_traffic(KERNEL_REGEX[fam].format(epi=epi or "0", nt4="" if (epi or "0") == "0" else "NEVER"), worker_args=same_job)
        if d["flops"] > 0:
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / BF16_PEAK_TFLOPS, **common}
        else:
            ach = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, **common}
        # the roofline entry above is a kernel FAMILY when it is gemm_nt; the largest SINGLE symbol of the step is the dW list kernel
        if "gemm_tn_multi" in agg and agg["gemm_tn_multi"]["ms"] > 0:
            t = agg["gemm_tn_multi"]
            tf = t["flops"] / (t["ms"] * 1e-3) / 1e12
            out["roofline"]["largest_single_symbol"] = {"kernel": symbol["gemm_tn_multi"], "ms_per_step": t["ms"] / 2, "launches_per_step": t["n"] // 2,
                                                        "achieved": tf, "frac": tf / BF16_PEAK_TFLOPS, "unit": "TFLOP/s"}
        out["roofline"]["peak_note"] = ("peak = the dense bf16 MFMA figure of MI355X_MICROARCH.md (2.5 PFLOP/s at 2.4 GHz); on random bf16 operands the chip sustains "
                                        "1.79-1.89 PFLOP/s at ~1.72 GHz (profiles/r03_ubench.txt): frac x 1.35 is the fraction of what a pure MFMA loop reaches")
        out["kernel_breakdown_ms_per_step"] = {k: round(v["ms"] / 2, 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
        # every launch family of the step is itemised (GEMMs, attention, norms, heads, loss, activation backward, optimizer, casts, fills);
        # the remainder is what the events do not bracket: torch's own small kernels, launch gaps, and the events' own overhead
        out["kernel_breakdown_ms_per_step"]["sum_of_families"] = round(tot_ms / 2, 3)
        out["kernel_breakdown_ms_per_step"]["step_minus_families"] = round(out["ms_per_step"] - tot_ms / 2, 3)
        out["kernel_tflops"] = {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) for k, v in agg.items() if v["flops"] > 0 and v["ms"] > 0}
        if os.environ.get("BENCH_SHAPE_TABLE"):      # per-shape table of the timed launches (tuning aid), off the JSON line
            with open(os.environ["BENCH_SHAPE_TABLE"], "w") as f:
                f.write("\n".join(prof.shape_table(2)) + "\n")
    if world > 1:
        dist.barrier()
    if rank == 0 and a.mods == "mod7":
        out["parity"] = parity_record()
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        print(json.dumps(out), file=sys.stderr)      # the GPU result is safe on stderr before the CPU leg starts
        out["cpu_baseline"] = cpu_baseline(workload="train" if a.mods == "mod7" else "train21")
    if rank == 0 and world == 1 and a.extras:
        # BASELINE configs[4] and configs[3] on the same record (child processes, bounded; the headline keys above are not touched)
        del model, opt, batches
        import gc
        gc.collect(); torch.cuda.empty_cache()
        out["extra"] = {"vq": extra_record(["--workload", "vq", "--steps", "10", "--warmup", "3"], 300),
                        "mod21": extra_record(["--mods", "mod21", "--steps", "5", "--warmup", "2"], 600),
                        "divae": extra_record(["--workload", "divae", "--steps", "7", "--warmup", "1"], 400)}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
