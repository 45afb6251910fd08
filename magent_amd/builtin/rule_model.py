"""Rule-based actors.  Only the uniform-random one is provided (the reference's `RandomActor`,
python/magent/builtin/rule_model/random.py); its hand-written chase / flee policies sit on `temp_c_booster`, which is
outside this engine's scope.

Works on both observation forms of `GridWorld.get_observation`: numpy arrays give a numpy int32 action vector, torch
tensors (device_obs mode) give an int32 tensor on the same device, so that the actions never leave the GPU."""
import numpy as np

from ..model import BaseModel


class RandomActor(BaseModel):
    def __init__(self, env, handle, *args, seed=None, **kwargs):
        super().__init__(env, handle)
        self.env, self.handle = env, handle
        self.n_action = int(env.get_action_space(handle)[0])
        self._host_rng = np.random if seed is None else np.random.RandomState(seed)
        self._device_rng = None
        self._seed = seed

    def infer_action(self, obs, *args, **kwargs):
        view = obs[0]
        count = len(view)
        if isinstance(view, np.ndarray):
            return self._host_rng.randint(self.n_action, size=count, dtype=np.int32)
        import torch
        if self._device_rng is None and self._seed is not None:
            self._device_rng = torch.Generator(device=view.device)
            self._device_rng.manual_seed(int(self._seed))
        return torch.randint(self.n_action, (count,), dtype=torch.int32, device=view.device, generator=self._device_rng)
