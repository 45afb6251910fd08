"""the reference keeps TensorFlow models here; this build provides the same class on PyTorch-ROCm"""
from magent_amd.builtin.torch_model import DeepQNetwork


class DeepRecurrentQNetwork(object):
    """import-compatibility placeholder: the recurrent variant (tf_model/drqn.py) is not provided by this build"""
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("DeepRecurrentQNetwork is not provided by this build; use DeepQNetwork (--alg dqn)")


__all__ = ["DeepQNetwork", "DeepRecurrentQNetwork"]
