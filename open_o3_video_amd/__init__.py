"""open_o3_video_amd: MI355X-native (gfx950) implementation of Open-o3-Video's generate hot path.

Layout: csrc/ (HIP kernels + C++ engine behind the C ABI of include/o3v.h), _lib.py (ctypes binding),
indexing.py / vision_process.py (host integer + policy logic), engine.py (planner), hf_api.py / vllm_api.py
(the two call surfaces of the reference), rollout.py (GSPO group rollout), dist.py (data-parallel sharding).
"""
from .config import O3VConfig  # noqa: F401

__version__ = "0.1.0"
