"""The trainer drop-in: ``ml-4m_amd/run_training_4m.py`` has upstream's command line and
YAML surface; ``fourm.utils`` carries the schedules / loggers / run-name helpers the trainer calls; everything the package does not
implement (``fourm.data`` loaders, vendored timm, ...) falls through to an upstream checkout.  Tests that need the upstream tree
(``/root/reference``: present in the build container, absent on the GPU box) run in a child interpreter with this package FIRST on
``sys.path`` and are skipped without it."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ml-4m_amd")
REF = os.environ.get("FOURM_REFERENCE_ROOT", "/root/reference")
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "fourm")), reason="no upstream checkout")
YAML_B = os.path.join(REF, "cfgs/default/4m/models/main/4m-b_mod7_500b.yaml")


def child(code: str, upstream=True, timeout=600):
    """Run ``code`` with [this package, repo root, upstream checkout] on sys.path (this package first: it shadows ``fourm``)."""
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env.pop("FOURM_UPSTREAM", None)
    if upstream:
        env["FOURM_UPSTREAM"] = REF
    pre = f"import sys; sys.path[:0] = [{PKG!r}, {ROOT!r}]\n"
    if upstream:       # inert torchvision / webdataset / ... stand-ins (install() also puts the upstream tree on sys.path: keep ours first)
        pre += f"from tests.golden import ref_stubs; ref_stubs.install(); sys.path[:] = [{PKG!r}, {ROOT!r}] + [p for p in sys.path if p not in ({PKG!r}, {ROOT!r})]\n"
    r = subprocess.run([sys.executable, "-c", pre + textwrap.dedent(code)], capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


@needs_ref
def test_upstream_get_args_runs_on_this_package_and_agrees_with_ours():
    """Upstream's own ``get_args()`` (run_training_4m.py:42-239), imported with THIS package shadowing ``fourm`` (its module-level
    ``from fourm.data import build_mixture_dataloader, ...`` and ``import fourm.utils as utils`` resolve through the fall-through),
    parses upstream's 4M-B YAML; our ``get_args`` yields the identical namespace for the same command lines."""
    out = child(f"""
        import importlib.util, json, sys
        spec = importlib.util.spec_from_file_location("upstream_trainer", {os.path.join(REF, 'run_training_4m.py')!r})
        U = importlib.util.module_from_spec(spec); spec.loader.exec_module(U)
        import run_training_4m as T
        assert T.__file__.startswith({PKG!r}), T.__file__
        import fourm; assert fourm.__file__.startswith({PKG!r})
        cases = [["-c", {YAML_B!r}],
                 ["-c", {YAML_B!r}, "--batch_size", "64", "--no_fixed_eval", "--opt_betas", "0.9", "0.99", "--clip_grad", "3.0", "--dtype", "bf16"],
                 ["--epochs", "3", "--scheduler", "inverse_sqrt-10000", "--no_auto_resume", "--eval", "--log_wandb", "--find_unused_params"],
                 ["-c", {os.path.join(REF, 'cfgs/default/4m/models/main/4m-l_mod21_500b.yaml')!r}, "--no_pin_mem", "--no_compute_grad_norm"]]
        for argv in cases:
            sys.argv = ["run_training_4m.py"] + argv
            a, b = vars(U.get_args()), vars(T.get_args(argv))
            assert a == b, sorted(set(a.items()) ^ set(b.items()), key=str)
        a = U.get_args.__globals__["utils"]
        print(json.dumps(dict(n=len(b), model=b["model"], utils_file=a.__file__)))
    """)
    info = json.loads(out.strip().splitlines()[-1])
    assert info["n"] > 70 and info["utils_file"].startswith(PKG)


@needs_ref
def test_fall_through_resolves_upstream_only_names():
    out = child(f"""
        import fourm, fourm.utils as U, fourm.data as D
        from fourm.data import build_mixture_dataloader, get_train_dataloader, get_val_dataloader, setup_sampling_mod_info
        ref = {REF!r}
        import inspect
        assert inspect.getsourcefile(build_mixture_dataloader).startswith(ref)          # upstream's function ...
        assert build_mixture_dataloader.__module__.startswith("fourm.data")             # ... living inside THIS package's namespace
        from fourm.data.masking import UnifiedMasking, image_mask_batched                 # upstream class + this package's kernel wrapper
        assert inspect.getsourcefile(UnifiedMasking).startswith(ref) and inspect.getsourcefile(inspect.unwrap(image_mask_batched)).startswith({PKG!r})
        from fourm.utils.timm.model_ema import ModelEmaV2                                 # a sub-package this package lacks
        assert inspect.getsourcefile(U.MetricLogger).startswith({PKG!r})                 # implemented here: ours wins
        assert inspect.getsourcefile(U.cosine_scheduler).startswith({PKG!r})
        assert inspect.getsourcefile(U.create_model).startswith({PKG!r})
        from fourm.utils import ModelEma                                                  # exported by upstream's fourm/utils/__init__ only
        assert inspect.getsourcefile(ModelEma).startswith(ref)
        from fourm.models.fm import FM
        assert inspect.getsourcefile(FM).startswith({PKG!r})
        print("ok")
    """)
    assert out.strip().endswith("ok")


def test_without_a_checkout_upstream_only_names_fail_loudly():
    out = child("""
        import fourm.data as D, fourm.utils as U
        from fourm.data import SyntheticLoader, synthetic_batch
        for mod, name in ((D, "build_mixture_dataloader"), (U, "ModelEma")):
            try:
                getattr(mod, name)
            except AttributeError as e:
                assert "FOURM_UPSTREAM" in str(e), e
            else:
                raise SystemExit(f"{name} resolved without an upstream checkout")
        print("ok")
    """, upstream=False)
    assert out.strip().endswith("ok")


def _load_upstream(rel, name):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@needs_ref
def test_schedules_equal_upstream():
    from fourm.utils import scheduler as S
    U = _load_upstream("fourm/utils/scheduler.py", "upstream_scheduler")
    for kw in (dict(base_value=1e-3, final_value=1e-6, epochs=5, niter_per_ep=37, warmup_epochs=1),
               dict(base_value=2e-4, final_value=0.0, epochs=3, niter_per_ep=50, warmup_epochs=-1, warmup_steps=40),
               dict(base_value=0.05, final_value=0.05, epochs=4, niter_per_ep=11)):
        np.testing.assert_allclose(S.cosine_scheduler(**kw), U.cosine_scheduler(**kw), rtol=1e-12, atol=0)
    for kw in (dict(base_value=1e-3, final_value=1e-5, epochs=6, niter_per_ep=40, warmup_epochs=1, cooldown_epochs=1, timescale=100),
               dict(base_value=1e-3, final_value=1e-3, epochs=2, niter_per_ep=40, warmup_steps=7, cooldown_steps=9),
               dict(base_value=4e-4, final_value=0.0, epochs=3, niter_per_ep=25)):
        np.testing.assert_allclose(S.inverse_sqrt_scheduler(**kw), U.inverse_sqrt_scheduler(**kw), rtol=1e-12, atol=0)
    np.testing.assert_array_equal(S.constant_scheduler(0.3, 2, 5), U.constant_scheduler(0.3, 2, 5))


@needs_ref
def test_run_name_equals_upstream():
    from argparse import Namespace
    from fourm.utils.run_name import setup_run_name
    U = _load_upstream("fourm/utils/run_name.py", "upstream_run_name")
    for kw in (dict(run_name="auto", config_path="cfgs/default/4m/models/main/4m-b_mod7_500b.yaml", wandb_run_name="auto",
                    output_dir="output/auto", s3_save_dir="s3://bucket/auto/x"),
               dict(run_name="mine", config_path="", wandb_run_name="auto", output_dir="out/auto"),
               dict(run_name="auto", config_path="/abs/cfgs/a/b/c.yaml", output_dir="")):
        a, b = Namespace(**kw), Namespace(**kw)
        setup_run_name(a); U.setup_run_name(b)
        assert vars(a) == vars(b)


def test_metric_logger_and_smoothed_value(capsys):
    import torch
    from fourm.utils import MetricLogger, SmoothedValue
    v = SmoothedValue(window_size=3)
    for x in (1.0, 2.0, 6.0, 4.0):
        v.update(x)
    assert v.count == 4 and v.total == 13.0 and v.global_avg == 3.25 and v.value == 4.0 and v.max == 6.0
    assert v.median == 4.0 and abs(v.avg - 4.0) < 1e-6 and str(v) == "4.0000 (3.2500)"
    m = MetricLogger(delimiter="  ")
    m.add_meter("lr", SmoothedValue(window_size=1, fmt="{value:.6f}"))
    seen = []
    for i in m.log_every(range(25), 10, iter_len=25, header="Epoch: [0]"):
        m.update(loss=torch.tensor(float(i)), lr=0.5, skipped=None)
        seen.append(i)
    assert seen == list(range(25)) and m.loss.count == 25 and m.meters["loss"].global_avg == 12.0 and "skipped" not in m.meters
    out = capsys.readouterr().out
    assert out.count("Epoch: [0]") == 5 and "[ 0/25]" in out and "[24/25]" in out and "lr: 0.500000" in out and "Total time" in out
    with pytest.raises(AttributeError):
        m.nope
    with pytest.raises(TypeError):
        m.update(bad="x")


def test_trainer_schedule_bookkeeping():
    """tokens -> epochs / warm-up steps and the lr / wd tables for upstream's 4M-B settings (run_training_4m.py:432-559)."""
    import run_training_4m as T
    a = T.get_args(["-c", os.path.join(ROOT, "tests/cfgs/default/4m-ti_mod7_synth.yaml"), "--epochs", "-1", "--total_tokens", "500", "--warmup_epochs", "-1",
                    "--warmup_tokens", "10", "--epoch_size", "10000000", "--batch_size", "128"])
    T.resolve_lengths(a)           # world size 1
    assert a.epochs == int(np.ceil(500e9 / (256 * 10_000_000))) == 196
    assert a.warmup_steps == int(np.ceil(10e9 / (256 * 128)))
    a.lr, a.min_lr, a.frozen_model_lr = 1e-4 * 128 / 256, 0.0, 1e-4
    steps = 10_000_000 // 128
    lr, wd = T.build_schedules(a, steps)
    assert len(lr) == len(wd) == a.epochs * steps and lr[0] == 0.0 and abs(lr[a.warmup_steps - 1] - a.lr) < 1e-12 and lr[-1] < 1e-9
    assert np.all(wd == a.weight_decay) and a.weight_decay_end == a.weight_decay
    assert a.data_config == "synthetic" and a.clip_grad == 3.0 and a.tokenizer_path.endswith(".json")
    from fourm.utils.run_name import setup_run_name
    setup_run_name(a)
    assert a.run_name == "4m-ti_mod7_synth"


def test_synthetic_data_config_variants():
    """``--data_config synthetic`` / ``synthetic:dirichlet``: our dataset type (no upstream counterpart) and its masking switch."""
    out = child(r'''
        import types
        import run_training_4m as T
        for name, want in (("synthetic", "uniform"), ("synthetic:dirichlet", "dirichlet")):
            cfg = T.load_data_config(types.SimpleNamespace(data_config=name))
            ds = cfg["train"]["datasets"]["synthetic"]
            assert ds["type"] == "synthetic" and ds["masking"] == want
            assert "rgb@224" in ds["in_domains"].split("-") and "rgb@224" not in ds["out_domains"].split("-")
        print("ok")
    ''', upstream=False)
    assert "ok" in out
