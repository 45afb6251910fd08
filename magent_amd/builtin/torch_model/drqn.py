"""Deep recurrent Q network on PyTorch-ROCm with the constructor / method surface of the reference's TensorFlow model
(python/magent/builtin/tf_model/drqn.py:14-18, 205-402).

Network (drqn.py:140-187): 2 x conv3x3(32, valid, relu) -> dense 256 (view) || dense 256 (feature) -> GRU(512) ->
dueling head.  Acting keeps one hidden state per agent id (dropped when the agent disappears); training splits the
stored episodes into windows of `unroll_step` steps starting from a zero state, episodes drawn in proportion to
their length, double-DQN targets from the window shifted by one step, masked squared TD error, Adam, global-norm
clipping 10."""
import collections
import os
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...model import BaseModel


class _RecurrentQNet(nn.Module):
    STATE = 512

    def __init__(self, view_space, feature_space, n_action, use_dueling):
        super().__init__()
        h, w, c = view_space
        self.conv1, self.conv2 = nn.Conv2d(c, 32, 3), nn.Conv2d(32, 32, 3)
        self.dense_view = nn.Linear(32 * (h - 4) * (w - 4), 256)
        self.dense_emb = nn.Linear(feature_space[0], 256)
        self.rnn = nn.GRU(self.STATE, self.STATE, batch_first=True)
        self.use_dueling = use_dueling
        if use_dueling:
            self.value, self.advantage = nn.Linear(self.STATE, 1), nn.Linear(self.STATE, n_action, bias=False)
        else:
            self.value = nn.Linear(self.STATE, n_action)

    def forward(self, view, feature, batch, unroll, state=None):
        """view / feature hold batch * unroll rows, sequence-major per batch entry; returns (q [batch*unroll, A], state)"""
        x = view.permute(0, 3, 1, 2)
        x = F.relu(self.conv2(F.relu(self.conv1(x))))
        x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)
        h = torch.cat([F.relu(self.dense_view(x)), F.relu(self.dense_emb(feature))], dim=1)
        out, state = self.rnn(h.reshape(batch, unroll, self.STATE), state)
        out = out.reshape(batch * unroll, self.STATE)
        if self.use_dueling:
            adv = self.advantage(out)
            return self.value(out) + adv - adv.mean(dim=1, keepdim=True), state
        return self.value(out), state


class DeepRecurrentQNetwork(BaseModel):
    def __init__(self, env, handle, name, batch_size=32, unroll_step=8, reward_decay=0.99, learning_rate=1e-4, train_freq=1,
                 memory_size=20000, target_update=2000, eval_obs=None, use_dueling=True, use_double=True,
                 use_episode_train=False, custom_view_space=None, custom_feature_space=None, device=None):
        BaseModel.__init__(self, env, handle)
        self.env, self.handle, self.name, self.subclass_name = env, handle, name, "torchdrqn"
        self.view_space = tuple(custom_view_space or env.get_view_space(handle))
        self.feature_space = tuple(custom_feature_space or env.get_feature_space(handle))
        self.num_actions = env.get_action_space(handle)[0]
        self.batch_size, self.unroll_step, self.gamma, self.learning_rate = int(batch_size), int(unroll_step), reward_decay, learning_rate
        self.train_freq, self.target_update, self.eval_obs, self.use_double = train_freq, target_update, eval_obs, use_double
        self.train_ct, self.agent_states = 0, {}
        if use_episode_train:
            raise NotImplementedError("use_episode_train (drqn.py:404, train_keep_hidden) is not provided")
        if device is None:
            device = torch.device("cuda", getattr(env, "device_id", 0)) if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)
        self.qnet = _RecurrentQNet(self.view_space, self.feature_space, self.num_actions, use_dueling).to(self.device)
        self.target_net = _RecurrentQNet(self.view_space, self.feature_space, self.num_actions, use_dueling).to(self.device)
        self.target_net.load_state_dict(self.qnet.state_dict())
        self.optimizer = torch.optim.Adam(self.qnet.parameters(), lr=learning_rate)
        # episodes: (views, features, actions, rewards, terminals) as device tensors; the oldest fall out (drqn.py:129-131)
        self.memory_size = memory_size
        self.replay_buffer = collections.deque(maxlen=memory_size)

    def _tensor(self, x, dtype=torch.float32):
        if isinstance(x, torch.Tensor):
            return x.to(self.device, dtype)
        return torch.as_tensor(np.ascontiguousarray(x)).to(self.device, dtype)

    # ------------------------------------------------------------------ acting
    @torch.no_grad()
    def infer_action(self, raw_obs, ids, policy="e_greedy", eps=0):
        """epsilon-greedy actions; the recurrent state of every agent is carried from its previous call by id"""
        view, feature = raw_obs[0], raw_obs[1]
        n = len(ids)
        if n == 0:
            self.agent_states = {}
            return np.empty(0, dtype=np.int32)
        ids_host = ids.cpu().numpy() if isinstance(ids, torch.Tensor) else np.asarray(ids)
        zero = torch.zeros(_RecurrentQNet.STATE, device=self.device)
        states = torch.stack([self.agent_states.get(int(i), zero) for i in ids_host]).unsqueeze(0)
        q, states = self.qnet(self._tensor(view), self._tensor(feature), n, 1, states)
        self.agent_states = {int(i): states[0, k] for k, i in enumerate(ids_host)}   # agents that are gone drop out
        best = q.argmax(dim=1).to(torch.int32)
        if policy == "e_greedy":
            rnd = torch.randint(self.num_actions, best.shape, dtype=torch.int32, device=self.device)
            best = torch.where(torch.rand(best.shape, device=self.device) < eps, rnd, best)
        return best if isinstance(view, torch.Tensor) else best.cpu().numpy()

    # ------------------------------------------------------------------ learning
    def _add_to_replay_buffer(self, sample_buffer):
        n = 0
        for ep in sample_buffer.episodes():
            m = len(ep.rewards)
            if m == 0:
                continue
            terminal = np.zeros(m, dtype=bool)
            terminal[-1] = bool(ep.terminal)
            v = torch.stack(ep.views).to(self.device) if isinstance(ep.views[0], torch.Tensor) else self._tensor(np.stack(ep.views))
            f = torch.stack(ep.features).to(self.device) if isinstance(ep.features[0], torch.Tensor) else self._tensor(np.stack(ep.features))
            self.replay_buffer.append((v, f, self._tensor(np.asarray(ep.actions), torch.int64),
                                       self._tensor(np.asarray(ep.rewards, dtype=np.float32)), self._tensor(terminal, torch.bool)))
            n += m
        return n

    @torch.no_grad()
    def _calc_target(self, next_view, next_feature, rewards, terminal):
        t_q, _ = self.target_net(next_view, next_feature, self.batch_size, self.unroll_step)
        if self.use_double:
            pick = self.qnet(next_view, next_feature, self.batch_size, self.unroll_step)[0].argmax(dim=1, keepdim=True)
            nxt = t_q.gather(1, pick).squeeze(1)
        else:
            nxt = t_q.max(dim=1).values
        return torch.where(terminal, rewards, rewards + self.gamma * nxt)

    def train(self, sample_buffer, print_every=500):
        """add the round's episodes to the replay memory, then train on windows of `unroll_step` steps (zero initial state)"""
        add_num = self._add_to_replay_buffer(sample_buffer)
        B, U = self.batch_size, self.unroll_step
        n_batches = int(self.train_freq * add_num / (B * U))
        if n_batches == 0 or not self.replay_buffer:
            return 0, 0
        lens = np.array([len(item[3]) for item in self.replay_buffer], dtype=np.float64)
        weight = lens / lens.sum()
        print("batches: %d  add: %d  replay_len: %d/%d" % (n_batches, add_num, len(self.replay_buffer), self.memory_size))
        d = self.device
        view = torch.zeros((B * U + 1,) + self.view_space, device=d)
        feature = torch.zeros((B * U + 1,) + self.feature_space, device=d)
        action = torch.zeros(B * U, dtype=torch.int64, device=d)
        reward, mask = torch.zeros(B * U, device=d), torch.zeros(B * U, device=d)
        terminal = torch.zeros(B * U, dtype=torch.bool, device=d)
        start_time, total_loss, target = time.time(), 0.0, None
        for ct in range(n_batches):
            picks = np.random.choice(len(self.replay_buffer), B, p=weight)
            mask.zero_()
            for j, k in enumerate(picks):
                v, f, a, r, t = self.replay_buffer[k]
                start = np.random.randint(len(r))
                real = min(len(r) - start, U)
                beg = j * U
                view[beg:beg + real], feature[beg:beg + real] = v[start:start + real], f[start:start + real]
                action[beg:beg + real], reward[beg:beg + real] = a[start:start + real], r[start:start + real]
                terminal[beg:beg + real] = t[start:start + real]
                mask[beg:beg + real] = 1.0
                if not bool(t[start + real - 1]):
                    mask[beg + real - 1] = 0      # no successor inside the window
            target = self._calc_target(view[1:], feature[1:], reward, terminal)
            q, _ = self.qnet(view[:-1], feature[:-1], B, U)
            q_taken = q.gather(1, action.unsqueeze(1)).squeeze(1)
            loss = ((target - q_taken) ** 2 * mask).sum() / mask.sum().clamp_min(1e-12)
            self.optimizer.zero_grad(set_to_none=True)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(self.qnet.parameters(), 10.0)
            self.optimizer.step()
            total_loss += float(loss.detach())
            if ct % self.target_update == 0:
                self.target_net.load_state_dict(self.qnet.state_dict())
            if ct % print_every == 0:
                print("batch %5d, loss %.6f, qvalue %.6f" % (ct, float(loss.detach()), float(target.mean())))
            self.train_ct += 1
        total_time = time.time() - start_time
        print("batches: %d,  total time: %.2f,  1k average: %.2f" % (n_batches, total_time, total_time / max(1.0, n_batches / 1000.0)))
        return total_loss / n_batches, float(target.mean())

    def get_info(self):
        return "tfdrqn train_time: %d" % self.train_ct

    # ------------------------------------------------------------------ checkpoints
    def _path(self, dir_name, name, epoch):
        return os.path.join(dir_name, name, "%s_%d.pt" % (self.subclass_name, epoch))

    def save(self, dir_name, epoch):
        os.makedirs(os.path.join(dir_name, self.name), exist_ok=True)
        torch.save({"qnet": self.qnet.state_dict(), "target": self.target_net.state_dict(),
                    "optimizer": self.optimizer.state_dict(), "train_ct": self.train_ct}, self._path(dir_name, self.name, epoch))

    def load(self, dir_name, epoch=0, name=None):
        state = torch.load(self._path(dir_name, name or self.name, epoch), map_location=self.device)
        self.qnet.load_state_dict(state["qnet"])
        self.target_net.load_state_dict(state["target"])
        self.optimizer.load_state_dict(state["optimizer"])
        self.train_ct = state.get("train_ct", 0)
