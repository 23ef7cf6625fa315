"""backward + (unscale) + clip / skip + optimizer step, with the call signature and state_dict of upstream
``fourm/utils/native_scaler.py:21-64``.  With ``FusedAdamW`` the gradient norm and the clipping coefficient
stay on the device (no host round trip between backward and the parameter update)."""
import torch


class NativeScalerWithGradNormCount:
    state_dict_key = "amp_scaler"

    def __init__(self, enabled=True):
        self._scaler = torch.amp.GradScaler("cuda", enabled=enabled and torch.cuda.is_available())

    def __call__(self, loss, optimizer, clip_grad=None, skip_grad=None, parameters=None, create_graph=False, update_grad=True,
                 compute_grad_norm=True):
        self._scaler.scale(loss).backward(create_graph=create_graph)
        if not update_grad:
            return None
        fused = hasattr(optimizer, "fused_grad_norm") and not self._scaler.is_enabled()
        if fused:
            norm = None
            if clip_grad is not None:
                norm = optimizer.fused_grad_norm(clip=clip_grad)        # clip applied inside the AdamW kernel
            elif skip_grad is not None:
                norm = optimizer.fused_grad_norm()
                if norm >= skip_grad:                                    # host decision, as upstream
                    return norm
            elif compute_grad_norm:
                norm = optimizer.fused_grad_norm(lazy=True)              # filled by the step below (the norm rides on the AdamW pass)
            optimizer.step()
            return norm
        return self._generic_step(optimizer, parameters, clip_grad, skip_grad, compute_grad_norm)

    def _generic_step(self, optimizer, parameters, clip_grad, skip_grad, want_norm):
        """Any other optimizer (or an enabled loss scaler): torch's own unscale / clip / step sequence."""
        sc = self._scaler
        sc.unscale_(optimizer)
        norm, do_step = None, True
        if clip_grad is not None:
            if parameters is None:
                raise ValueError("clip_grad needs the parameter list")
            norm = torch.nn.utils.clip_grad_norm_(parameters, clip_grad)
        elif skip_grad is not None:
            norm = get_grad_norm_(parameters)
            do_step = bool(norm < skip_grad)                 # an exploding step is dropped, the scaler still advances
        elif want_norm:
            norm = get_grad_norm_(parameters)
        if do_step:
            sc.step(optimizer)
        sc.update()
        return norm

    def state_dict(self):
        return self._scaler.state_dict()

    def load_state_dict(self, state_dict):
        self._scaler.load_state_dict(state_dict)


def get_grad_norm_(parameters, norm_type: float = 2.0) -> torch.Tensor:
    """Global p-norm of the gradients present on ``parameters`` (a tensor or an iterable of tensors)."""
    params = [parameters] if isinstance(parameters, torch.Tensor) else list(parameters)
    per_tensor = [p.grad.detach().norm(float(norm_type)) for p in params if p.grad is not None]
    return torch.stack(per_tensor).norm(float(norm_type)) if per_tensor else torch.tensor(0.)


# names only upstream's same-named module defines (see fourm/_upstream.py)
from fourm import _upstream as _up
_up.merge(__name__, globals())
