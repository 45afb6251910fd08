"""Deep Q network on PyTorch-ROCm with the constructor / method surface of the reference's TensorFlow model
(python/magent/builtin/tf_model/dqn.py:13-18, 191-346), so that examples/train_battle.py can say
`from magent.builtin.tf_model import DeepQNetwork` and run unmodified.

Network (dqn.py:151-189): 2 x conv3x3(32, valid, relu) -> dense 256 (view) || dense 256 (feature) -> concat ->
dueling head (value + advantage without bias - mean advantage).  Double DQN targets, Adam, global-norm clipping 5,
masked squared TD error.  The replay memory lives on the model's device: with 288 GB of HBM the 2^20-entry buffer of
the reference (5 GB of views) needs no host round trip."""
import os
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...model import BaseModel


class _QNet(nn.Module):
    def __init__(self, view_space, feature_space, n_action, use_dueling, use_conv):
        super().__init__()
        h, w, c = view_space
        self.use_conv, self.use_dueling = use_conv, use_dueling
        if use_conv:
            self.conv1 = nn.Conv2d(c, 32, 3)
            self.conv2 = nn.Conv2d(32, 32, 3)
            flat = 32 * (h - 4) * (w - 4)
        else:
            flat = h * w * c
        self.dense_view = nn.Linear(flat, 256)
        self.dense_emb = nn.Linear(feature_space[0], 256)
        if use_dueling:
            self.value = nn.Linear(512, 1)
            self.advantage = nn.Linear(512, n_action, bias=False)
        else:
            self.value = nn.Linear(512, n_action)

    def forward(self, view, feature):
        if self.use_conv:
            x = view.permute(0, 3, 1, 2)                 # the engine renders NHWC
            x = F.relu(self.conv2(F.relu(self.conv1(x))))
            x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)
        else:
            x = view.reshape(view.shape[0], -1)
        h = torch.cat([F.relu(self.dense_view(x)), F.relu(self.dense_emb(feature))], dim=1)
        if self.use_dueling:
            adv = self.advantage(h)
            return self.value(h) + adv - adv.mean(dim=1, keepdim=True)
        return self.value(h)


class _Ring(object):
    """circular device buffer with batched put / gather"""
    def __init__(self, shape, dtype, device):
        self.buf = torch.empty(shape, dtype=dtype, device=device)
        self.head, self.capacity = 0, shape[0]

    def put(self, data):
        n = len(data)
        if n >= self.capacity:
            data, n = data[-self.capacity:], self.capacity
        first = min(n, self.capacity - self.head)
        self.buf[self.head:self.head + first] = data[:first]
        if n > first:
            self.buf[:n - first] = data[first:]
        self.head = (self.head + n) % self.capacity
        return n


class DeepQNetwork(BaseModel):
    def __init__(self, env, handle, name, batch_size=64, learning_rate=1e-4, reward_decay=0.99, train_freq=1,
                 target_update=2000, memory_size=2 ** 20, eval_obs=None, use_dueling=True, use_double=True, use_conv=True,
                 custom_view_space=None, custom_feature_space=None, num_gpu=1, infer_batch_size=8192, network_type=0,
                 device=None, infer_dtype=None):
        BaseModel.__init__(self, env, handle)
        self.env, self.handle, self.name, self.subclass_name = env, handle, name, "torchdqn"
        self.view_space = tuple(custom_view_space or env.get_view_space(handle))
        self.feature_space = tuple(custom_feature_space or env.get_feature_space(handle))
        self.num_actions = env.get_action_space(handle)[0]
        self.batch_size, self.learning_rate, self.gamma = int(batch_size), learning_rate, reward_decay
        self.train_freq, self.target_update, self.eval_obs = train_freq, target_update, eval_obs
        self.infer_batch_size, self.use_double, self.train_ct = infer_batch_size, use_double, 0
        if device is None:
            device = torch.device("cuda", getattr(env, "device_id", 0)) if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)
        self.qnet = _QNet(self.view_space, self.feature_space, self.num_actions, use_dueling, use_conv).to(self.device)
        self.target_net = _QNet(self.view_space, self.feature_space, self.num_actions, use_dueling, use_conv).to(self.device)
        self.target_net.load_state_dict(self.qnet.state_dict())
        self.optimizer = torch.optim.Adam(self.qnet.parameters(), lr=learning_rate)
        # The arithmetic of infer_action.  "f32" (the default) is the reference's own: its TensorFlow graph computes in float32
        # (tf_model/dqn.py:151-189), and so does the PyTorch network here.  "bf16" is an opt-in -- the constructor argument, or
        # MAGENT_POLICY_DTYPE=bf16 for scripts that are run unmodified: acting on device-resident observations then goes through the
        # hand-written MFMA kernels (bf16 inputs, weights and inter-layer activations, f32 accumulation; magent_amd/csrc/policy.hip)
        # when the network has the reference's default shape.  How far that is from the f32 network on real observations is pinned
        # in tests/test_policy.py::test_bf16_policy_against_the_f32_network (|dQ| <= 2 % of max |Q|, >= 97 % equal greedy actions).
        # Training is float32 either way.
        self.infer_dtype = (infer_dtype or os.environ.get("MAGENT_POLICY_DTYPE", "f32")).lower()
        if self.infer_dtype not in ("f32", "bf16"):
            raise ValueError("infer_dtype must be 'f32' or 'bf16', not %r" % (self.infer_dtype,))
        # "f32" on the GPU goes through hand-written kernels too since round 6: the same float32 arithmetic -- inputs, weights, activations,
        # accumulation -- on the f32 matrix instruction (magent_amd/csrc/policy_f32.hip; tests/test_policy.py pins it to the PyTorch network
        # within float32 round-off).  MAGENT_POLICY_F32=torch keeps the PyTorch / MIOpen forward pass (A/B runs, bench.py's comparison).
        self._hip = None
        if self.device.type == "cuda":
            try:
                if self.infer_dtype == "bf16":
                    from .hip_policy import HipDqnPolicy
                    self._hip = HipDqnPolicy(self.qnet, self.view_space, self.feature_space, self.num_actions, self.device)
                elif os.environ.get("MAGENT_POLICY_F32", "hip").lower() != "torch":
                    from .hip_policy import HipDqnPolicyF32
                    self._hip = HipDqnPolicyF32(self.qnet, self.view_space, self.feature_space, self.num_actions, self.device)
            except (ValueError, OSError, AttributeError):
                self._hip = None
        # replay memory; mask == 0 marks the padding transition that closes an unfinished episode (dqn.py:249-252)
        self.memory_size, self.replay_len = memory_size, 0
        d = self.device
        self.mem_view = _Ring((memory_size,) + self.view_space, torch.float32, d)
        self.mem_feature = _Ring((memory_size,) + self.feature_space, torch.float32, d)
        self.mem_action = _Ring((memory_size,), torch.int64, d)
        self.mem_reward = _Ring((memory_size,), torch.float32, d)
        self.mem_terminal = _Ring((memory_size,), torch.bool, d)
        self.mem_mask = _Ring((memory_size,), torch.float32, d)

    # ------------------------------------------------------------------ acting
    def _tensor(self, x, dtype=torch.float32):
        if isinstance(x, torch.Tensor):
            return x.to(self.device, dtype)
        return torch.as_tensor(np.ascontiguousarray(x)).to(self.device, dtype)

    @torch.no_grad()
    def infer_action(self, raw_obs, ids, policy="e_greedy", eps=0):
        """epsilon-greedy actions for a batch of agents; raw_obs = (view [n,H,W,C], feature [n,F]), numpy or torch"""
        view, feature = raw_obs[0], raw_obs[1]
        eps = 0 if policy == "greedy" else eps
        n = len(view)
        if (self._hip is not None and n > 0 and isinstance(view, torch.Tensor) and isinstance(feature, torch.Tensor) and view.is_cuda
                and (view.dtype == torch.float32 or (view.dtype == torch.bfloat16 and self.infer_dtype == "bf16")) and feature.dtype == torch.float32
                and view.is_contiguous() and feature.is_contiguous()):
            best = self._hip.infer(view, feature)
            if eps > 0:
                rnd = torch.randint(self.num_actions, best.shape, dtype=torch.int32, device=self.device)
                best = torch.where(torch.rand(best.shape, device=self.device) < eps, rnd, best)
            return best
        if isinstance(view, torch.Tensor) and view.dtype == torch.bfloat16:
            # bf16 cells [n, H, W, 8] are the MFMA kernels' operand format (channels, zeros, a constant 1); without those kernels
            # (infer_dtype "f32", an unsupported shape) the PyTorch network takes the channels back as float32
            view = view[..., :self.view_space[-1]].float()
        out = torch.empty(n, dtype=torch.int32, device=self.device)
        step = max(1, min(n, self.infer_batch_size))
        for beg in range(0, n, step):
            v, f = self._tensor(view[beg:beg + step]), self._tensor(feature[beg:beg + step])
            m = len(v)
            # group sizes change every step (deaths); pad the batch to a bucketed size so the convolution library sees
            # a handful of shapes instead of tuning a kernel for every n
            bucket = self._bucket(m)
            if bucket != m:
                v = torch.cat([v, v.new_zeros((bucket - m,) + tuple(v.shape[1:]))])
                f = torch.cat([f, f.new_zeros((bucket - m,) + tuple(f.shape[1:]))])
            best = self.qnet(v, f)[:m].argmax(dim=1).to(torch.int32)
            rnd = torch.randint(self.num_actions, best.shape, dtype=torch.int32, device=self.device)
            explore = torch.rand(best.shape, device=self.device) < eps
            out[beg:beg + step] = torch.where(explore, rnd, best)
        if isinstance(view, torch.Tensor):
            return out
        return out.cpu().numpy()

    @staticmethod
    def _bucket(m):
        if m <= 64:
            return 64
        p = 1 << (m - 1).bit_length()          # next power of two
        return p if m > (p * 3) // 4 else (p * 3) // 4 if m > p // 2 else p // 2

    # ------------------------------------------------------------------ learning
    def _add_to_replay_buffer(self, sample_buffer):
        if hasattr(sample_buffer, "packed"):      # every transition of the round in episode order, one put per array (utility.EpisodesBuffer)
            packed = sample_buffer.packed()
            if packed is None:
                return 0
            views, features, actions, rewards, terminal, mask = packed
            n = len(rewards)
            self.mem_view.put(views.to(self.device) if isinstance(views, torch.Tensor) else self._tensor(views))
            self.mem_feature.put(features.to(self.device) if isinstance(features, torch.Tensor) else self._tensor(features))
            self.mem_action.put(actions.to(self.device, torch.int64) if isinstance(actions, torch.Tensor) else self._tensor(actions, torch.int64))
            self.mem_reward.put(self._tensor(rewards))
            self.mem_terminal.put(self._tensor(terminal, torch.bool))
            self.mem_mask.put(self._tensor(mask))
            self.replay_len = min(self.memory_size, self.replay_len + n)
            return n
        n = 0
        for ep in sample_buffer.episodes():
            m = len(ep.rewards)
            if m == 0:
                continue
            mask = np.ones(m, dtype=np.float32)
            terminal = np.zeros(m, dtype=bool)
            if ep.terminal:
                terminal[-1] = True
            else:
                mask[-1] = 0
            if isinstance(ep.views[0], torch.Tensor):
                self.mem_view.put(torch.stack(ep.views).to(self.device))
                self.mem_feature.put(torch.stack(ep.features).to(self.device))
            else:
                self.mem_view.put(self._tensor(np.stack(ep.views)))
                self.mem_feature.put(self._tensor(np.stack(ep.features)))
            self.mem_action.put(self._tensor(np.asarray(ep.actions), torch.int64))
            self.mem_reward.put(self._tensor(np.asarray(ep.rewards, dtype=np.float32)))
            self.mem_terminal.put(self._tensor(terminal, torch.bool))
            self.mem_mask.put(self._tensor(mask))
            n += m
        self.replay_len = min(self.memory_size, self.replay_len + n)
        return n

    @torch.no_grad()
    def _calc_target(self, next_view, next_feature, rewards, terminal):
        t_q = self.target_net(next_view, next_feature)
        if self.use_double:
            pick = self.qnet(next_view, next_feature).argmax(dim=1, keepdim=True)
            nxt = t_q.gather(1, pick).squeeze(1)
        else:
            nxt = t_q.max(dim=1).values
        return torch.where(terminal, rewards, rewards + self.gamma * nxt)

    def train(self, sample_buffer, print_every=1000):
        """add the round's episodes to the replay memory, then run train_freq * new_samples / batch_size batches"""
        add_num = self._add_to_replay_buffer(sample_buffer)
        n_batches = int(self.train_freq * add_num / self.batch_size)
        if n_batches == 0 or self.replay_len < 2:
            return 0, 0
        print("batch number: %d  add: %d  replay_len: %d/%d" % (n_batches, add_num, self.replay_len, self.memory_size))
        start, total_loss, target = time.time(), 0.0, None
        for ct in range(n_batches):
            idx = torch.randint(self.replay_len - 1, (self.batch_size,), device=self.device)
            nxt = idx + 1
            target = self._calc_target(self.mem_view.buf[nxt], self.mem_feature.buf[nxt], self.mem_reward.buf[idx],
                                       self.mem_terminal.buf[idx])
            q = self.qnet(self.mem_view.buf[idx], self.mem_feature.buf[idx])
            q_taken = q.gather(1, self.mem_action.buf[idx].unsqueeze(1)).squeeze(1)
            mask = self.mem_mask.buf[idx]
            loss = ((target - q_taken) ** 2 * mask).sum() / mask.sum().clamp_min(1e-12)
            self.optimizer.zero_grad(set_to_none=True)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(self.qnet.parameters(), 5.0)
            self.optimizer.step()
            if self._hip is not None:
                self._hip.dirty = True
            total_loss += float(loss.detach())
            if ct % self.target_update == 0:
                self.target_net.load_state_dict(self.qnet.state_dict())
            if ct % print_every == 0:
                print("batch %5d,  loss %.6f, eval %.6f" % (ct, float(loss.detach()), self._eval(target)))
            self.train_ct += 1
        total_time = time.time() - start
        print("batches: %d,  total time: %.2f,  1k average: %.2f" % (n_batches, total_time, total_time / max(1.0, n_batches / 1000.0)))
        return total_loss / n_batches, self._eval(target)

    @torch.no_grad()
    def _eval(self, target):
        if self.eval_obs is None:
            return float(target.mean())
        return float(self.qnet(self._tensor(self.eval_obs[0]), self._tensor(self.eval_obs[1])).mean())

    def clear_buffer(self):
        self.replay_len = 0
        for ring in (self.mem_view, self.mem_feature, self.mem_action, self.mem_reward, self.mem_terminal, self.mem_mask):
            ring.head = 0

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
        if self._hip is not None:
            self._hip.dirty = True
        self.target_net.load_state_dict(state["target"])
        self.optimizer.load_state_dict(state["optimizer"])
        self.train_ct = state.get("train_ct", 0)
