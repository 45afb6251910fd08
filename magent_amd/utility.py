"""Helpers around the engine used by the training scripts (mirror of the call surface of reference
python/magent/utility.py: EpisodesBuffer, decay schedules, sample_observation, init_logger, rec_round)."""
import logging
import math

import numpy as np


class EpisodesBufferEntry(object):
    """the transitions of ONE agent during one game round"""
    __slots__ = ("views", "features", "actions", "rewards", "terminal")

    def __init__(self):
        self.views, self.features, self.actions, self.rewards, self.terminal = [], [], [], [], False

    def append(self, view, feature, action, reward, alive):
        if isinstance(view, np.ndarray):
            self.views.append(np.array(view, copy=True))
            self.features.append(np.array(feature, copy=True))
        else:                                   # torch rows of a device-resident observation: stay on the device
            self.views.append(view.clone())
            self.features.append(feature.clone())
        self.actions.append(int(action))
        self.rewards.append(reward)
        if not alive:
            self.terminal = True


class EpisodesBuffer(object):
    """Per-agent episode store, one entry per tracked agent id, at most `capacity` agents.

    Same policy as the reference (utility.py:33-77): until the buffer is full, the agents of a step are admitted in
    random order; once `capacity` agents are tracked, only those keep being recorded.  The per-step work is
    vectorised over the tracked set (the reference walks every id of the step in Python, which does not survive a
    million agents)."""

    def __init__(self, capacity):
        self.buffer = {}
        self.capacity = capacity
        self.is_full = False

    def record_step(self, ids, obs, acts, rewards, alives):
        ids = np.asarray(ids)
        views, features = obs[0], obs[1]
        n = len(ids)
        if not self.is_full:
            for i in np.random.permutation(n):
                key = int(ids[i])
                if key not in self.buffer:
                    self.buffer[key] = EpisodesBufferEntry()
                    if len(self.buffer) >= self.capacity:
                        self.is_full = True
                        break
        if n == 0 or not self.buffer:
            return
        tracked = np.fromiter(self.buffer.keys(), dtype=np.int64, count=len(self.buffer))
        rows = np.nonzero(np.isin(ids, tracked, assume_unique=False))[0]
        if not isinstance(acts, np.ndarray):    # device tensor of actions: fetch only the tracked rows
            import torch
            acts_rows = acts[torch.as_tensor(rows, device=acts.device)].cpu().numpy() if len(rows) else np.zeros(0, np.int32)
        else:
            acts_rows = acts[rows]
        for k, i in enumerate(rows):
            self.buffer[int(ids[i])].append(views[i], features[i], acts_rows[k], rewards[i], alives[i])

    def reset(self):
        self.buffer = {}
        self.is_full = False

    def episodes(self):
        return self.buffer.values()


def exponential_decay(now_step, total_step, final_value, rate):
    decay = math.exp(math.log(final_value) / total_step ** rate)
    return max(final_value, 1 * decay ** (now_step ** rate))


def linear_decay(now_step, total_step, final_value):
    return max(final_value, 1 - (1 - final_value) / total_step * now_step)


def piecewise_decay(now_step, anchor, anchor_value):
    """piecewise-linear schedule through (anchor[i], anchor_value[i]); flat after the last anchor"""
    i = 0
    while i < len(anchor) and now_step >= anchor[i]:
        i += 1
    if i == len(anchor):
        return anchor_value[-1]
    slope = (anchor_value[i] - anchor_value[i - 1]) / (anchor[i] - anchor[i - 1])
    return anchor_value[i - 1] + (now_step - anchor[i - 1]) * slope


def sample_observation(env, handles, n_obs=-1, step=-1):
    """play random actions and collect (view, feature) samples per group, e.g. as a fixed evaluation set"""
    from .builtin.rule_model import RandomActor
    actors = [RandomActor(env, h) for h in handles]
    views, feats = [[] for _ in handles], [[] for _ in handles]
    done, t = False, 0
    while not done:
        for i, h in enumerate(handles):
            v, f = env.get_observation(h)
            views[i].append(v.copy()); feats[i].append(f.copy())
            env.set_action(h, actors[i].infer_action((v, f), None))
        done = env.step()
        env.clear_dead()
        t += 1
        if step != -1 and t >= step:
            break
    out = []
    for i in range(len(handles)):
        v, f = np.concatenate(views[i]), np.concatenate(feats[i])
        if n_obs != -1 and len(v) > 0:
            pick = np.random.choice(len(v), n_obs)
            v, f = v[pick], f[pick]
        out.append((v, f))
    return out


def init_logger(filename):
    logging.basicConfig(level=logging.INFO, filename=filename + ".log")
    console = logging.StreamHandler()
    console.setLevel(logging.INFO)
    logging.getLogger("").addHandler(console)


def rec_round(x, ndigits=2):
    if isinstance(x, (list, tuple, np.ndarray)):
        return [rec_round(v, ndigits) for v in x]
    return round(float(x), ndigits)


def check_model(name):
    """the reference downloads pre-trained TensorFlow checkpoints here (utility.py:260-305); there is no network and no
    TensorFlow in this build: report what is missing instead"""
    import os
    path = os.path.join("data", name + "_model")
    if not os.path.exists(path):
        raise FileNotFoundError("pre-trained model %r is not shipped with this build (expected under %s)" % (name, path))
    return True


class FontProvider(object):
    """8x8 pixel font for the `arrange` game's goal layouts (reference python/magent/utility.py:271-305).

    The file holds one glyph per line: eight comma-separated byte literals, one byte per row, bit j = pixel j.
    `get(ch)` returns the glyph as an 8x8 list of 0/1 rows; `ch` is a character or a code point."""
    width = height = 8

    def __init__(self, filename):
        self.data = []
        with open(filename) as f:
            for line in f:
                fields = [x.strip() for x in line.split(",") if x.strip()]
                if not fields:
                    continue
                rows = [int(x, 0) for x in fields[:self.height]]
                self.data.append([[(row >> j) & 1 for j in range(self.width)] for row in rows])

    def get(self, ch):
        return self.data[ch if isinstance(ch, int) else ord(ch)]
