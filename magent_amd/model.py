"""Model hosting (mirror of the call surface of reference python/magent/model.py).

The reference starts one sub-process per model and ships every observation batch through a named pipe
(model.py:145-155,194-196) -- gigabytes per step at a million agents.  Here the model lives in the caller's process,
next to the engine, on the same GPU; the "non-blocking" calls of the reference protocol (infer_action(block=False) /
fetch_action, sample_step(block=False) / check_done, train(block=False) / fetch_train) keep their names and order,
and simply complete eagerly."""
from . import utility


class BaseModel(object):
    def __init__(self, env, handle, *args, **kwargs):
        pass

    def infer_action(self, raw_obs, ids, *args, **kwargs):
        pass

    def train(self, sample_buffer, **kwargs):
        return 0, 0

    def save(self, *args, **kwargs):
        pass

    def load(self, *args, **kwargs):
        pass


class ProcessingModel(BaseModel):
    def __init__(self, env, handle, name, port, sample_buffer_capacity=1000, RLModel=None, **kwargs):
        BaseModel.__init__(self, env, handle)
        assert RLModel is not None
        self.model = RLModel(env=env, handle=handle, name=name, **kwargs)
        self.capacity = sample_buffer_capacity
        self.sample_buffer = utility.EpisodesBuffer(sample_buffer_capacity)
        self._last = None          # (obs, ids, actions) of the pending step
        self._train_result = None

    def infer_action(self, raw_obs, ids, policy="e_greedy", eps=0, block=True):
        acts = self.model.infer_action(raw_obs, ids, policy=policy, eps=eps)
        self._last = (raw_obs, ids, acts)
        return acts if block else None

    def fetch_action(self):
        return self._last[2]

    def sample_step(self, rewards, alives, block=True):
        obs, ids, acts = self._last
        self.sample_buffer.record_step(ids, obs, acts, rewards, alives)

    def train(self, print_every=5000, block=True):
        self._train_result = self.model.train(self.sample_buffer, print_every=print_every)
        self.sample_buffer = utility.EpisodesBuffer(self.capacity)
        if block:
            return self.fetch_train()

    def fetch_train(self):
        return self._train_result

    def save(self, save_dir, epoch, block=True):
        self.model.save(save_dir, epoch)

    def load(self, save_dir, epoch, name=None, block=True):
        self.model.load(save_dir, epoch, name)

    def check_done(self):
        pass

    def quit(self):
        pass
