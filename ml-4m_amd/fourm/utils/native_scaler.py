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
                norm = optimizer.fused_grad_norm()
            optimizer.step()
            return norm
        self._scaler.unscale_(optimizer)
        if clip_grad is not None:
            assert parameters is not None
            norm = torch.nn.utils.clip_grad_norm_(parameters, clip_grad)
        elif skip_grad is not None:
            norm = get_grad_norm_(parameters)
            if norm >= skip_grad:
                self._scaler.update()
                return norm
        else:
            norm = get_grad_norm_(parameters) if compute_grad_norm else None
        self._scaler.step(optimizer)
        self._scaler.update()
        return norm

    def state_dict(self):
        return self._scaler.state_dict()

    def load_state_dict(self, state_dict):
        self._scaler.load_state_dict(state_dict)


def get_grad_norm_(parameters, norm_type: float = 2.0) -> torch.Tensor:
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    grads = [p.grad.detach() for p in parameters if p.grad is not None]
    if not grads:
        return torch.tensor(0.)
    return torch.norm(torch.stack([torch.norm(g, float(norm_type)) for g in grads]), float(norm_type))
