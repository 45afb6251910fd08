"""the reference keeps TensorFlow models here; this build provides the same classes on PyTorch-ROCm"""
from magent_amd.builtin.torch_model import AdvantageActorCritic, DeepQNetwork, DeepRecurrentQNetwork

__all__ = ["DeepQNetwork", "DeepRecurrentQNetwork", "AdvantageActorCritic"]
