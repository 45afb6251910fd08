"""rule-based actors (only the one the measurement needs: uniform random actions)"""
import numpy as np

from ..model import BaseModel


class RandomActor(BaseModel):
    def __init__(self, env, handle, *args, **kwargs):
        BaseModel.__init__(self, env, handle)
        self.env, self.handle = env, handle
        self.n_action = env.get_action_space(handle)[0]

    def infer_action(self, obs, *args, **kwargs):
        return np.random.randint(self.n_action, size=len(obs[0]), dtype=np.int32)
