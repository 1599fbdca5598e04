"""raytracer_amd — MI355X-native render core for omeldar/raytracer's per-pixel ray<->scene hot path.

csrc/        HIP kernels + the C ABI (include/rtcore.h) and the C++ host mirror (include/rthost.h)
api.py       numpy-level access to both ABIs (GpuScene, HostBvh)
scenes.py    synthetic scenes for the BASELINE.json configs
dist.py      one-process-per-GPU tile sharding + RCCL framebuffer gather
"""
from .api import GpuScene, HostBvh, default_camera, make_params, tile_list, untile_host  # noqa: F401
from .scenes import SceneSpec  # noqa: F401
