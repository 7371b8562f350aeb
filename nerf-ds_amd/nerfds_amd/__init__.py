"""MI355X-native volume-rendering core for NeRF-DS (host-side mirror of the reference's render surface)."""
from .config import NerfModelConfig, MLPSpec, nerf_ds_config, static_config, hypernerf_config
from .params import init_params, param_count
