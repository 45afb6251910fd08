"""magent_amd -- MI355X-native grid-world step engine behind the MAgent GridWorld interface.

Only the hot path lives here: the C-ABI library (csrc/, built into lib/libmagent.so) and the host-side mirror of
the reference's ``magent.GridWorld`` operator interface (gridworld.py).
"""
from . import gridworld
from .gridworld import EnvBatch, GridWorld, step_many

__all__ = ["gridworld", "GridWorld", "step_many", "EnvBatch"]
