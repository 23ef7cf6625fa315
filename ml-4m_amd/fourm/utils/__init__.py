"""Host-side helpers mirroring the names the upstream trainer imports from ``fourm.utils``."""
from .registry import register_model, model_entrypoint, is_model, list_models, create_model
from .misc import generate_uint15_hash
from .native_scaler import NativeScalerWithGradNormCount, get_grad_norm_
from .optim_factory import create_optimizer, get_parameter_groups, FusedAdamW
from .checkpoint import save_model, auto_load_model, load_state_dict, load_safetensors
from .dist import init_distributed_mode, is_dist_avail_and_initialized, get_world_size, get_rank, is_main_process
