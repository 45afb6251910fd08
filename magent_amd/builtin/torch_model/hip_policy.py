"""Binding of the hand-written MI355X inference kernels for the reference's deep Q network (include/magent_policy.h,
magent_amd/csrc/policy.hip) to the PyTorch model that owns the parameters (dqn.py: _QNet).

The kernels want every weight matrix in the operand order of v_mfma_f32_32x32x16_bf16 ("fragment order") and every
activation in the order a lane of the MFMA result holds its 16 outputs ("slot order"); both are plain index permutations of
the torch parameters, done here with tensor ops on the device whenever the parameters have changed."""
import ctypes

import torch

from ... import c_lib


class _Shape(ctypes.Structure):
    _fields_ = [("view_h", ctypes.c_int), ("view_w", ctypes.c_int), ("view_c", ctypes.c_int), ("feat", ctypes.c_int),
                ("n_action", ctypes.c_int)]


class _Weights(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ("conv1", "conv2", "dense_view", "dense_emb", "head", "conv2_bias",
                                               "dense_view_bias", "dense_emb_bias")] + [("value_bias", ctypes.c_float)]


def slot_channels(device):
    """channel (output) held in slot s of a 32-wide tile: (s & 3) + 8 ((s & 15) >> 2) + 4 (s >> 4)  (policy.hip: ch_of)"""
    s = torch.arange(32, device=device)
    return (s & 3) + 8 * ((s & 15) >> 2) + 4 * (s >> 4)


def fragment_order(w):
    """[N (multiple of 32)][K (multiple of 16)] -> bf16 [K / 16][N / 32][64 lanes][8]: lane l of k-step s and tile T holds
    w[32 T + (l & 31)][16 s + 8 (l >> 5) + 0..7]"""
    n, k = w.shape
    assert n % 32 == 0 and k % 16 == 0
    return w.reshape(n // 32, 32, k // 16, 2, 8).permute(2, 0, 3, 1, 4).contiguous().to(torch.bfloat16).reshape(k // 16, n // 32, 64, 8)


CONV1_TAP_ORDER = [0, 3, 1, 4, 2, 5, 6, 7, 8, 9]       # tap (ky * 3 + kx; 9 = padding) in position 2 s + g of conv1's reduction index


def _pad_k(w, k):
    return torch.cat([w, w.new_zeros(w.shape[0], k - w.shape[1])], dim=1) if w.shape[1] < k else w


class HipDqnPolicy(object):
    """greedy actions (and, for tests, the Q values) of a dueling conv _QNet, computed by k_dqn_conv + k_dqn_head"""

    def __init__(self, qnet, view_space, feature_space, n_action, device, chunk=131072):
        self._lib = c_lib.load()
        self.qnet, self.device, self.chunk = qnet, torch.device(device), int(chunk)
        h, w, c = view_space
        self.shape = _Shape(h, w, c, feature_space[0], n_action)
        if not (qnet.use_conv and qnet.use_dueling) or not self._lib.policy_dqn_supported(ctypes.byref(self.shape)):
            raise ValueError("network shape not taken by the HIP policy kernels")
        self.k_dense = (h - 4) * (w - 4) * 32
        self._packed, self._work = None, None
        self.dirty = True

    @torch.no_grad()
    def pack(self):
        q, dev = self.qnet, self.device
        ch = slot_channels(dev)
        c = self.shape.view_c
        w1 = q.conv1.weight.detach().float()                              # [32][C][3][3] -> [32][ky][kx][8] -> K = tap * 8 + channel
        w1 = torch.cat([w1, w1.new_zeros(32, 8 - c, 3, 3)], dim=1).permute(0, 2, 3, 1).contiguous()
        w1[:, 0, 0, 7] = q.conv1.bias.detach().float()       # the kernel feeds a constant 1.0 in channel 7: the MFMA adds the bias
        # the kernel pairs the taps (0|3) (1|4) (2|5) (6|7) (8|pad) into its five k-steps (policy.hip: k_dqn_conv)
        w1 = _pad_k(w1.reshape(32, 72), 80).reshape(32, 10, 8)[:, CONV1_TAP_ORDER].reshape(32, 80)
        w2 = q.conv2.weight.detach().float()[:, ch].permute(0, 2, 3, 1).reshape(32, 288)            # K = tap * 32 + slot
        wv = q.dense_view.weight.detach().float().reshape(256, -1, 32)[:, :, ch].reshape(256, self.k_dense)   # K = position * 32 + slot
        fk = (self.shape.feat + 15) // 16 * 16
        we = _pad_k(q.dense_emb.weight.detach().float(), fk)
        hidden = (torch.arange(16, device=dev)[:, None] * 32 + ch[None, :]).reshape(512)              # hidden slot -> hidden unit
        head = torch.zeros(32, 512, device=dev)
        head[:self.shape.n_action] = q.advantage.weight.detach().float()
        head[self.shape.n_action] = q.value.weight.detach().float()[0]
        t = {
            "conv1": fragment_order(w1), "conv2": fragment_order(w2), "dense_view": fragment_order(wv),
            "dense_emb": fragment_order(we), "head": fragment_order(head[:, hidden]),
            "conv2_bias": q.conv2.bias.detach().float()[ch].contiguous(),
            "dense_view_bias": q.dense_view.bias.detach().float()[hidden[:256]].contiguous(),
            "dense_emb_bias": q.dense_emb.bias.detach().float()[hidden[:256]].contiguous(),
        }
        w = _Weights()
        for k, v in t.items():
            setattr(w, k, v.data_ptr())
        w.value_bias = float(q.value.bias.detach().float().item())
        self._packed, self._w, self.dirty = t, w, False       # (the tensors stay alive as long as the pointers are in use)

    @torch.no_grad()
    def infer(self, view, feature, want_q=False):
        """view float32 [n][H][W][C] -- or bfloat16 [n][H][W][8], the engine's cells (GridWorld.get_observation_device_bf16) --,
        feature float32 [n][F]: contiguous CUDA tensors.  Returns int32 actions [n] (and Q [n][A])"""
        cells16 = view.dtype == torch.bfloat16
        assert view.is_cuda and view.is_contiguous() and feature.is_contiguous() and feature.dtype == torch.float32
        assert (cells16 and view.shape[-1] == 8) or (view.dtype == torch.float32 and view.shape[-1] == self.shape.view_c)
        call = self._lib.policy_dqn_infer_bf16 if cells16 else self._lib.policy_dqn_infer
        if self.dirty:
            self.pack()
        n = view.shape[0]
        actions = torch.empty(n, dtype=torch.int32, device=view.device)
        q = torch.empty((n, self.shape.n_action), dtype=torch.float32, device=view.device) if want_q else None
        nbytes = ctypes.c_size_t(0)
        self._lib.policy_dqn_act_bytes(ctypes.byref(self.shape), min(n, self.chunk), ctypes.byref(nbytes))
        need = nbytes.value          # (activations in the kernels' own layout + the conv kernel's dump line)
        if self._work is None or self._work.numel() < need:
            self._work = torch.empty(need, dtype=torch.uint8, device=view.device)
        stream = torch.cuda.current_stream(view.device).cuda_stream
        for beg in range(0, n, self.chunk):
            m = min(self.chunk, n - beg)
            rc = call(ctypes.byref(self.shape), ctypes.byref(self._w), view[beg:].data_ptr(), feature[beg:].data_ptr(), m,
                                            self._work.data_ptr(), actions[beg:].data_ptr(), q[beg:].data_ptr() if want_q else None, stream)
            if rc != 0:
                raise RuntimeError("policy_dqn_infer failed (%d)" % rc)
        return (actions, q) if want_q else actions


# ---------------------------------------------------------------------------------------------------- float32 (the reference's arithmetic)
def fragment_order_f32(w):
    """[N (multiple of 32)][K (multiple of 8)] -> float32 [K / 8][N / 32][64 lanes][4]: lane l of group m and tile T holds
    w[32 T + (l & 31)][8 m + 4 (l >> 5) + 0..3]  (include/magent_policy.h: "f32 fragment order")"""
    n, k = w.shape
    assert n % 32 == 0 and k % 8 == 0
    return w.reshape(n // 32, 32, k // 8, 2, 4).permute(2, 0, 3, 1, 4).contiguous().float().reshape(k // 8, n // 32, 64, 4)


class HipDqnPolicyF32(object):
    """greedy actions (and the Q values) of a dueling conv _QNet in float32 -- inputs, weights, activations, accumulation: the reference
    network's own arithmetic -- computed by k_dqn_conv_f32 + k_dqn_head_f32 on v_mfma_f32_32x32x2_f32 (magent_amd/csrc/policy_f32.hip)"""

    def __init__(self, qnet, view_space, feature_space, n_action, device, chunk=131072):
        self._lib = c_lib.load()
        self.qnet, self.device, self.chunk = qnet, torch.device(device), int(chunk)
        h, w, c = view_space
        self.shape = _Shape(h, w, c, feature_space[0], n_action)
        if not (qnet.use_conv and qnet.use_dueling) or not self._lib.policy_dqn_f32_supported(ctypes.byref(self.shape)):
            raise ValueError("network shape not taken by the HIP f32 policy kernels")
        self.k_dense = (h - 4) * (w - 4) * 32
        self._packed, self._work = None, None
        self.dirty = True

    @torch.no_grad()
    def pack(self):
        q, dev = self.qnet, self.device
        c = self.shape.view_c
        w1 = q.conv1.weight.detach().float()                              # [32][C][3][3] -> [32][ky][kx][8] -> K = tap * 8 + channel
        w1 = torch.cat([w1, w1.new_zeros(32, 8 - c, 3, 3)], dim=1).permute(0, 2, 3, 1).contiguous()
        w1[:, 0, 0, 7] = q.conv1.bias.detach().float()       # the kernel feeds a constant 1.0 in channel 7: the MFMA adds the bias
        w2 = q.conv2.weight.detach().float().permute(0, 2, 3, 1).reshape(32, 288)                      # K = tap * 32 + channel
        wv = q.dense_view.weight.detach().float()                                                      # K = position * 32 + channel (NHWC flatten)
        fk = (self.shape.feat + 7) // 8 * 8
        we = _pad_k(q.dense_emb.weight.detach().float(), fk)
        head = torch.zeros(32, 512, device=dev)
        head[:self.shape.n_action] = q.advantage.weight.detach().float()
        head[self.shape.n_action] = q.value.weight.detach().float()[0]
        t = {
            "conv1": fragment_order_f32(w1.reshape(32, 72)), "conv2": fragment_order_f32(w2), "dense_view": fragment_order_f32(wv),
            "dense_emb": fragment_order_f32(we), "head": fragment_order_f32(head),
            "conv2_bias": q.conv2.bias.detach().float().contiguous(),
            "dense_view_bias": q.dense_view.bias.detach().float().contiguous(),
            "dense_emb_bias": q.dense_emb.bias.detach().float().contiguous(),
        }
        w = _Weights()
        for k, v in t.items():
            setattr(w, k, v.data_ptr())
        w.value_bias = float(q.value.bias.detach().float().item())
        self._packed, self._w, self.dirty = t, w, False

    @torch.no_grad()
    def infer(self, view, feature, want_q=False):
        """view float32 [n][H][W][C], feature float32 [n][F]: contiguous CUDA tensors (the engine's observation tensors as they are).
        Returns int32 actions [n] (and Q [n][A])"""
        assert view.is_cuda and view.is_contiguous() and feature.is_contiguous() and view.dtype == torch.float32 and feature.dtype == torch.float32
        assert view.shape[-1] == self.shape.view_c
        if self.dirty:
            self.pack()
        n = view.shape[0]
        actions = torch.empty(n, dtype=torch.int32, device=view.device)
        q = torch.empty((n, self.shape.n_action), dtype=torch.float32, device=view.device) if want_q else None
        nbytes = ctypes.c_size_t(0)
        self._lib.policy_dqn_f32_act_bytes(ctypes.byref(self.shape), min(n, self.chunk), ctypes.byref(nbytes))
        if self._work is None or self._work.numel() < nbytes.value:
            self._work = torch.empty(nbytes.value, dtype=torch.uint8, device=view.device)
        stream = torch.cuda.current_stream(view.device).cuda_stream
        for beg in range(0, n, self.chunk):
            m = min(self.chunk, n - beg)
            rc = self._lib.policy_dqn_infer_f32(ctypes.byref(self.shape), ctypes.byref(self._w), view[beg:].data_ptr(), feature[beg:].data_ptr(), m,
                                                self._work.data_ptr(), actions[beg:].data_ptr(), q[beg:].data_ptr() if want_q else None, stream)
            if rc != 0:
                raise RuntimeError("policy_dqn_infer_f32 failed (%d)" % rc)
        return (actions, q) if want_q else actions
