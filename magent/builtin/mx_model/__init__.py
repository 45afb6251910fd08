"""the reference keeps MXNet models here; this build provides the same class on PyTorch-ROCm"""
from magent_amd.builtin.torch_model import DeepQNetwork

__all__ = ["DeepQNetwork"]
