from .dqn import DeepQNetwork

__all__ = ["DeepQNetwork"]
