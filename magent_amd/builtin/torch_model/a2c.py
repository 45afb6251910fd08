"""Advantage actor-critic on PyTorch-ROCm with the constructor / method surface of the reference's TensorFlow model
(python/magent/builtin/tf_model/a2c.py:10-14, 189-287).

Network (a2c.py:140-167): dense 256 (flattened view) || dense 256 (feature) -> dense 512 -> softmax policy head and a
scalar value head; optional CommNet block (mean of the OTHER agents' hidden units, two steps).  One update per
`train()` over every sample of the round: discounted returns bootstrapped with the value of each episode's last
observation, loss = -E[adv * log pi] + value_coef * E[(R - V)^2] + ent_coef * E[sum pi log pi], Adam."""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...model import BaseModel


class _CommStep(nn.Module):
    """h' = tanh(mean_other(h) C + h H + skip)"""
    def __init__(self, size):
        super().__init__()
        self.C = nn.Linear(size, size, bias=False)
        self.H = nn.Linear(size, size, bias=False)

    def forward(self, h, skip):
        n = h.shape[0]
        others = (h.sum(dim=0, keepdim=True) - h) / (n - 1) if n > 1 else torch.zeros_like(h)
        return torch.tanh(self.C(others) + self.H(h) + skip)


class _ActorCritic(nn.Module):
    def __init__(self, view_space, feature_space, n_action, use_comm):
        super().__init__()
        self.dense_view = nn.Linear(int(np.prod(view_space)), 256)
        self.dense_emb = nn.Linear(feature_space[0], 256)
        self.dense = nn.Linear(512, 512)
        self.comm = nn.ModuleList([_CommStep(512), _CommStep(512)]) if use_comm else None
        self.policy = nn.Linear(512, n_action)
        self.value = nn.Linear(512, 1)

    def forward(self, view, feature):
        h = torch.cat([F.relu(self.dense_view(view.reshape(view.shape[0], -1))), F.relu(self.dense_emb(feature))], dim=1)
        h = F.relu(self.dense(h))
        if self.comm is not None:
            skip = h
            for step in self.comm:
                h = step(h, skip)
        policy = F.softmax(self.policy(h), dim=1).clamp(1e-10, 1 - 1e-10)
        return policy, self.value(h).squeeze(1)


class AdvantageActorCritic(BaseModel):
    def __init__(self, env, handle, name, learning_rate=1e-3, batch_size=64, reward_decay=0.99, eval_obs=None,
                 train_freq=1, value_coef=0.1, ent_coef=0.08, use_comm=False, custom_view_space=None,
                 custom_feature_space=None, device=None):
        BaseModel.__init__(self, env, handle)
        self.env, self.handle, self.name, self.subclass_name = env, handle, name, "torcha2c"
        self.view_space = tuple(custom_view_space or env.get_view_space(handle))
        self.feature_space = tuple(custom_feature_space or env.get_feature_space(handle))
        self.num_actions = env.get_action_space(handle)[0]
        self.reward_decay, self.batch_size, self.learning_rate, self.train_freq = reward_decay, batch_size, learning_rate, train_freq
        self.value_coef, self.ent_coef, self.use_comm, self.train_ct = value_coef, ent_coef, use_comm, 0
        if device is None:
            device = torch.device("cuda", getattr(env, "device_id", 0)) if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)
        self.net = _ActorCritic(self.view_space, self.feature_space, self.num_actions, use_comm).to(self.device)
        self.optimizer = torch.optim.Adam(self.net.parameters(), lr=learning_rate)

    def _tensor(self, x, dtype=torch.float32):
        if isinstance(x, torch.Tensor):
            return x.to(self.device, dtype)
        return torch.as_tensor(np.ascontiguousarray(x)).to(self.device, dtype)

    @torch.no_grad()
    def infer_action(self, raw_obs, ids, *args, **kwargs):
        """one action per agent, sampled from the policy; raw_obs = (view [n,H,W,C], feature [n,F]), numpy or torch"""
        view, feature = raw_obs[0], raw_obs[1]
        if len(view) == 0:
            return np.empty(0, dtype=np.int32)
        policy, _ = self.net(self._tensor(view), self._tensor(feature))
        acts = torch.multinomial(policy, 1).squeeze(1).to(torch.int32)
        return acts if isinstance(view, torch.Tensor) else acts.cpu().numpy()

    def train(self, sample_buffer, print_every=1000):
        """one gradient step over all samples of the round; returns ([pg_loss, vf_loss, ent_loss], mean state value)"""
        views, features, actions, returns = [], [], [], []
        for ep in sample_buffer.episodes():
            m = len(ep.rewards)
            if m == 0:
                continue
            v = torch.stack(ep.views).to(self.device) if isinstance(ep.views[0], torch.Tensor) else self._tensor(np.stack(ep.views))
            f = torch.stack(ep.features).to(self.device) if isinstance(ep.features[0], torch.Tensor) else self._tensor(np.stack(ep.features))
            with torch.no_grad():
                keep = float(self.net(v[-1:], f[-1:])[1][0])      # bootstrap from the last observation (a2c.py:244-251)
            r = np.asarray(ep.rewards, dtype=np.float64)
            for i in reversed(range(m)):
                keep = keep * self.reward_decay + r[i]
                r[i] = keep
            views.append(v); features.append(f)
            actions.append(self._tensor(np.asarray(ep.actions), torch.int64)); returns.append(self._tensor(r))
        if not views:
            return [0.0, 0.0, 0.0], 0.0
        view, feature, action, reward = torch.cat(views), torch.cat(features), torch.cat(actions), torch.cat(returns)
        policy, value = self.net(view, feature)
        log_policy = torch.log(policy + 1e-6)
        log_prob = log_policy.gather(1, action.unsqueeze(1)).squeeze(1)
        advantage = (reward - value).detach()
        pg_loss = -(advantage * log_prob).mean()
        vf_loss = self.value_coef * ((reward - value) ** 2).mean()
        neg_entropy = self.ent_coef * (policy * log_policy).sum(dim=1).mean()
        self.optimizer.zero_grad(set_to_none=True)
        (pg_loss + vf_loss + neg_entropy).backward()
        self.optimizer.step()
        self.train_ct += 1
        losses = [float(x.detach()) for x in (pg_loss, vf_loss, neg_entropy)]
        print("sample", len(reward), *losses)
        return losses, float(value.detach().mean())

    def get_info(self):
        return "a2c train_time: %d" % self.train_ct

    def _path(self, dir_name, name, epoch):
        return os.path.join(dir_name, name, "%s_%d.pt" % (self.subclass_name, epoch))

    def save(self, dir_name, epoch):
        os.makedirs(os.path.join(dir_name, self.name), exist_ok=True)
        torch.save({"net": self.net.state_dict(), "optimizer": self.optimizer.state_dict(), "train_ct": self.train_ct},
                   self._path(dir_name, self.name, epoch))

    def load(self, dir_name, epoch=0, name=None):
        state = torch.load(self._path(dir_name, name or self.name, epoch), map_location=self.device)
        self.net.load_state_dict(state["net"])
        self.optimizer.load_state_dict(state["optimizer"])
        self.train_ct = state.get("train_ct", 0)
