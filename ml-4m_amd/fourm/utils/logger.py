"""Console / wandb bookkeeping of the trainer with upstream's interface (``fourm/utils/logger.py:34-227``): ``SmoothedValue``,
``MetricLogger`` (``update``, ``add_meter``, ``log_every``, ``synchronize_between_processes``, ``.meters[...].global_avg``) and
``WandbLogger``.  Host-side only; the one collective (count / total of every meter, once per epoch) goes through
``torch.distributed`` on whatever backend the process group uses."""
import collections
import datetime
import time

import torch
import torch.distributed as dist

from .dist import is_dist_avail_and_initialized


class SmoothedValue:
    """Last ``window_size`` values (median / avg / max / value) plus the running total over everything seen (global_avg)."""

    def __init__(self, window_size=20, fmt=None):
        self.deque = collections.deque(maxlen=window_size)
        self.total, self.count = 0.0, 0
        self.fmt = fmt or "{median:.4f} ({global_avg:.4f})"

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        """Sum count / total over the ranks (the window is left alone, as upstream)."""
        if not is_dist_avail_and_initialized():
            return
        dev = "cuda" if (torch.cuda.is_available() and dist.get_backend() == "nccl") else "cpu"
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=dev)
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.total = int(t[0].item()), float(t[1].item())

    @property
    def median(self):
        return torch.tensor(list(self.deque)).median().item()

    @property
    def avg(self):
        return torch.tensor(list(self.deque), dtype=torch.float32).mean().item()

    @property
    def global_avg(self):
        return self.total / self.count

    @property
    def max(self):
        return max(self.deque)

    @property
    def value(self):
        return self.deque[-1]

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


class MetricLogger:
    def __init__(self, delimiter="\t"):
        self.meters = collections.defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for name, v in kwargs.items():
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                v = v.item()
            if not isinstance(v, (float, int)):
                raise TypeError(f"metric {name!r}: {type(v).__name__} is not a number")
            self.meters[name].update(v)

    def __getattr__(self, attr):
        meters = self.__dict__.get("meters", {})
        if attr in meters:
            return meters[attr]
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{attr}'")

    def __str__(self):
        return self.delimiter.join(f"{name}: {meter}" for name, meter in self.meters.items())

    def synchronize_between_processes(self):
        for meter in self.meters.values():
            meter.synchronize_between_processes()

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def log_every(self, iterable, print_freq, iter_len=None, header=None):
        """Yield from ``iterable``; every ``print_freq`` items print progress, ETA, the meters, iteration / data time, peak memory."""
        n = iter_len if iter_len is not None else len(iterable)
        header = header or ""
        width = len(str(n))
        t_iter, t_data = SmoothedValue(fmt="{avg:.4f}"), SmoothedValue(fmt="{avg:.4f}")
        t0 = last = time.time()
        for i, obj in enumerate(iterable):
            t_data.update(time.time() - last)
            yield obj
            t_iter.update(time.time() - last)
            if i % print_freq == 0 or i == n - 1:
                eta = str(datetime.timedelta(seconds=int(t_iter.global_avg * (n - i)))) if n > 0 else "?"
                fields = [header, f"[{i:{width}d}/{n if n > 0 else '?'}]", f"eta: {eta}", str(self), f"time: {t_iter}", f"data: {t_data}"]
                if torch.cuda.is_available():
                    fields.append(f"max mem: {torch.cuda.max_memory_allocated() / 2 ** 20:.0f}")
                print(self.delimiter.join(fields))
            last = time.time()
        total = time.time() - t0
        per = f"{total / n:.4f}" if n > 0 else "?"
        print(f"{header} Total time: {datetime.timedelta(seconds=int(total))} ({per} s / it)")


class WandbLogger:
    """Thin wrapper over ``wandb`` (imported on construction: the package is optional)."""

    def __init__(self, args):
        import wandb
        self._wandb = wandb
        self.step = 0
        wandb.init(config=args, entity=args.wandb_entity, project=args.wandb_project, group=getattr(args, "wandb_group", None),
                   name=getattr(args, "wandb_run_name", None), tags=getattr(args, "wandb_tags", None),
                   mode=getattr(args, "wandb_mode", "online"))

    def set_step(self, step=None):
        self.step = step if step is not None else self.step + 1

    def update(self, metrics):
        row = {k: (v.item() if isinstance(v, torch.Tensor) else v) for k, v in metrics.items() if v is not None}
        try:
            self._wandb.log(row, step=self.step)
        except (self._wandb.CommError, BrokenPipeError):
            print("wandb logging failed, skipping...")

    def flush(self):
        pass

    def finish(self):
        try:
            self._wandb.finish()
        except (self._wandb.CommError, BrokenPipeError):
            print("wandb failed to finish")


# names only upstream's same-named module defines (see fourm/_upstream.py)
from fourm import _upstream as _up
_up.merge(__name__, globals())
