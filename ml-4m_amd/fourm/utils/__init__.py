"""Host-side helpers mirroring the names the upstream trainer imports from ``fourm.utils``."""
from .. import _upstream
_upstream.extend_path(__name__, __path__)          # first: sub-modules loaded below may import upstream-only siblings
from .registry import register_model, model_entrypoint, is_model, list_models, create_model
from .misc import generate_uint15_hash
from .native_scaler import NativeScalerWithGradNormCount, get_grad_norm_
from .optim_factory import create_optimizer, get_parameter_groups, FusedAdamW
from .checkpoint import save_model, auto_load_model, load_state_dict, load_safetensors
from .dist import init_distributed_mode, is_dist_avail_and_initialized, get_world_size, get_rank, is_main_process
from .scheduler import cosine_scheduler, constant_scheduler, inverse_sqrt_scheduler
from .logger import SmoothedValue, MetricLogger, WandbLogger
from .run_name import setup_run_name
from .s3_utils import setup_s3_args
from . import s3_utils

# everything else upstream's ``fourm.utils`` offers (vendored timm pieces, tokenizer helpers, generation datasets, ...) comes from an
# upstream checkout when one is configured: see fourm/_upstream.py
__getattr__ = _upstream.fallthrough(__name__, is_package=True)
_upstream.preload(__name__)
