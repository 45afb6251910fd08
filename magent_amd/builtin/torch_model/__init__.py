from .a2c import AdvantageActorCritic
from .dqn import DeepQNetwork
from .drqn import DeepRecurrentQNetwork

__all__ = ["DeepQNetwork", "DeepRecurrentQNetwork", "AdvantageActorCritic"]
