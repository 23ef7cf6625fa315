"""Per-step learning-rate / weight-decay tables of the trainer (same values as upstream ``fourm/utils/scheduler.py:22-82``; the
trainer indexes them with the global iteration, run_training_4m.py:705-711)."""
import numpy as np


def _warmup_len(warmup_epochs, warmup_steps, niter_per_ep) -> int:
    n = warmup_steps if warmup_steps > 0 else warmup_epochs * niter_per_ep
    print("Set warmup steps = %d" % n)
    return int(n)


def _ramp(lo, hi, n, enabled):
    return np.linspace(lo, hi, n) if enabled else np.zeros(0)


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0, warmup_steps=-1):
    """Linear warm-up to ``base_value``, then half a cosine period down to ``final_value`` over the remaining steps."""
    total = epochs * niter_per_ep
    w = _warmup_len(warmup_epochs, warmup_steps, niter_per_ep)
    head = _ramp(start_warmup_value, base_value, w, warmup_epochs > 0 or warmup_steps > 0)
    n = total - w
    t = np.arange(n, dtype=np.float64)
    body = final_value + 0.5 * (base_value - final_value) * (1.0 + np.cos(np.pi * t / max(n, 1)))
    out = np.concatenate((head, body))
    assert len(out) == total, (len(out), total)
    return out


def constant_scheduler(base_value, epochs, niter_per_ep):
    return np.full(epochs * niter_per_ep, base_value, dtype=np.float64)


def inverse_sqrt_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0, warmup_steps=-1,
                           cooldown_epochs=0, cooldown_steps=-1, timescale=10_000):
    """Warm-up, then base / sqrt((t + timescale) / timescale), then a linear cool-down to ``final_value``."""
    total = epochs * niter_per_ep
    w = _warmup_len(warmup_epochs, warmup_steps, niter_per_ep)
    c = int(cooldown_steps if cooldown_steps > 0 else cooldown_epochs * niter_per_ep)
    print("Set cooldown steps = %d" % c)
    head = _ramp(start_warmup_value, base_value, w, warmup_epochs > 0 or warmup_steps > 0)
    t = np.arange(total - w - c, dtype=np.float64)
    body = np.full(len(t), base_value, dtype=np.float64) if base_value == final_value else base_value / np.sqrt((t + timescale) / timescale)
    tail = _ramp(body[-1], final_value, c, cooldown_epochs > 0 or cooldown_steps > 0)
    out = np.concatenate((head, body, tail))
    assert len(out) == total, (len(out), total)
    return out


# names only upstream's same-named module defines (see fourm/_upstream.py)
from fourm import _upstream as _up
_up.merge(__name__, globals())
