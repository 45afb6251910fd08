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
    random order; once `capacity` agents are tracked, only those keep being recorded.

    Storage is STEP-major: a step appends ONE gathered block per array (the tracked rows of the step's observation, actions,
    rewards, alive flags) -- one indexed copy per array, on whatever device the observation lives.  The reference walks every id
    of the step in Python and appends row by row; with device-resident observations that was two tiny device copies per tracked
    agent and step (30 ms per step at 1000 tracked agents, 20 times the engine's step).  The per-agent view the reference's
    consumers iterate (`episodes()`, `buffer`) is built on demand; `packed()` hands the same transitions to a replay memory as
    flat arrays in episode order without ever leaving the device."""

    def __init__(self, capacity):
        self.capacity = capacity
        self.is_full = False
        self._slot = {}              # agent id -> slot, in admission order
        self._ids_sorted = np.zeros(0, np.int64)
        self._slots_sorted = np.zeros(0, np.int64)
        self._steps = []             # (slots[k], views[k, ...], features[k, ...], actions[k], rewards[k], alives[k]) per recorded step
        self._entries = None
        self._sorted_n = None        # size of the group when its ids were last seen in ascending order (None: not known to be)
        self._expect_alive = 0       # tracked agents alive at the end of the last recorded step

    def record_step(self, ids, obs, acts, rewards, alives):
        ids = np.asarray(ids)
        views, features = obs[0], obs[1]
        n = len(ids)
        if not self.is_full:
            admitted = False
            for i in np.random.permutation(n):
                key = int(ids[i])
                if key not in self._slot:
                    self._slot[key] = len(self._slot)
                    admitted = True
                    if len(self._slot) >= self.capacity:
                        self.is_full = True
                        break
            if admitted:
                self._expect_alive = 1 << 62      # (new agents are tracked: this step goes through the general search)
                keys = np.fromiter(self._slot.keys(), dtype=np.int64, count=len(self._slot))
                order = np.argsort(keys, kind="stable")
                self._ids_sorted, self._slots_sorted = keys[order], np.arange(len(keys), dtype=np.int64)[order]
        if n == 0 or not self._slot:
            return
        # Which rows of this step belong to tracked agents?  The engine hands a group's ids out in ascending order (ids are given in
        # placement order and clear_dead keeps the order), so the tracked ids are LOOKED UP in them -- k log n instead of a pass over all n
        # (1.2 ms per side and step at 500k agents, all of `sample_step`'s time) -- whenever a cheap check says the ids are ascending:
        # the full check runs when the group has grown since the last step, and a look-up that misses an agent that was alive a step ago
        # sends the step through the general search.
        rows = None
        if self._sorted_n is not None and n <= self._sorted_n:
            pos = np.searchsorted(ids, self._ids_sorted)
            pos_c = np.minimum(pos, n - 1)
            hit = ids[pos_c] == self._ids_sorted
            if int(hit.sum()) >= self._expect_alive:
                rows = np.sort(pos_c[hit])
        if rows is None:
            self._sorted_n = n if n < 2 or bool(np.all(ids[1:] > ids[:-1])) else None
            rows = np.nonzero(np.isin(ids, self._ids_sorted))[0]
        else:
            self._sorted_n = n
        if len(rows) == 0:
            self._expect_alive = 0
            return
        slots = self._slots_sorted[np.searchsorted(self._ids_sorted, ids[rows])]
        self._expect_alive = int(np.asarray(alives)[rows].astype(bool).sum())      # tracked agents that will still be listed after clear_dead

        def take(a):
            if isinstance(a, np.ndarray):
                return a[rows]           # (fancy indexing copies)
            import torch
            return a[torch.as_tensor(rows, device=a.device)]
        acts_rows = take(acts) if not isinstance(acts, (list, tuple)) else np.asarray(acts)[rows]
        self._steps.append((slots, take(views), take(features), acts_rows, np.asarray(rewards)[rows].astype(np.float32),
                            np.asarray(alives)[rows].astype(bool)))
        self._entries = None

    def reset(self):
        self.__init__(self.capacity)

    # ---- the reference's per-agent view (built on demand)
    @property
    def buffer(self):
        if self._entries is None:
            entries = {key: EpisodesBufferEntry() for key in self._slot}
            by_slot = list(entries.values())
            for slots, views, features, acts, rewards, alives in self._steps:
                acts = acts if isinstance(acts, np.ndarray) else acts.cpu().numpy()
                for k, s in enumerate(slots):
                    e = by_slot[s]
                    e.views.append(views[k]); e.features.append(features[k]); e.actions.append(int(acts[k])); e.rewards.append(rewards[k])
                    if not alives[k]:
                        e.terminal = True
            self._entries = entries
        return self._entries

    def episodes(self):
        return self.buffer.values()

    def packed(self):
        """every recorded transition as flat arrays in EPISODE order -- the agents in admission order, each agent's steps in time
        order: exactly the sequence a loop over episodes() appends to a replay memory.  Returns (views, features, actions, rewards,
        terminal, mask) or None; views / features / actions are torch tensors where the observation was one, numpy arrays otherwise.
        terminal marks the last transition of an agent that died, mask == 0 the last transition of one that did not (its
        successor in the memory is not its next state: tf_model/dqn.py:249-252)."""
        if not self._steps:
            return None
        slots = np.concatenate([st[0] for st in self._steps])
        step = np.concatenate([np.full(len(st[0]), t, dtype=np.int64) for t, st in enumerate(self._steps)])
        order = np.lexsort((step, slots))
        slots_o = slots[order]
        alive = np.concatenate([st[5] for st in self._steps])
        died = np.bincount(slots, weights=~alive, minlength=len(self._slot)) > 0
        is_last = np.ones(len(order), dtype=bool)
        is_last[:-1] = slots_o[1:] != slots_o[:-1]
        terminal = is_last & died[slots_o]
        mask = np.where(is_last & ~died[slots_o], 0.0, 1.0).astype(np.float32)
        rewards = np.concatenate([st[4] for st in self._steps])[order]

        def gather(parts):
            if isinstance(parts[0], np.ndarray):
                return np.concatenate(parts)[order]
            import torch
            return torch.cat(parts)[torch.as_tensor(order, device=parts[0].device)]
        return (gather([st[1] for st in self._steps]), gather([st[2] for st in self._steps]), gather([st[3] for st in self._steps]),
                rewards, terminal, mask)


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
