"""Development helper (GPU box): time of the DQN inference kernels on n agents of the battle observation shape -- magent_amd/csrc/policy.hip
(bf16; `cells`: the engine's bf16-cell observations; `torch`: the PyTorch / MIOpen forward pass under bf16 autocast beside it) or, with `f32`,
magent_amd/csrc/policy_f32.hip (float32 on v_mfma_f32_32x32x2_f32; `torch`: PyTorch's float32 forward pass beside it).

    python tools/policy_rate.py [n] [reps] [f32] [cells] [torch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from magent_amd.builtin.torch_model.dqn import _QNet
from magent_amd.builtin.torch_model.hip_policy import HipDqnPolicy, HipDqnPolicyF32

nums = [a for a in sys.argv[1:] if a.isdigit()]
n = int(nums[0]) if len(nums) > 0 else 131072
reps = int(nums[1]) if len(nums) > 1 else 20
F32 = "f32" in sys.argv
dev = torch.device("cuda", 0)
vs, F, A = (13, 13, 7), 34, 21
qnet = _QNet(vs, (F,), A, True, True).to(dev)
view = (torch.rand((n,) + vs, device=dev) < 0.3).float()
feat = torch.rand((n, F), device=dev)
pol = (HipDqnPolicyF32 if F32 else HipDqnPolicy)(qnet, vs, (F,), A, dev, chunk=int(os.environ.get("CHUNK", min(n, 131072) if F32 else n)))
if "cells" in sys.argv:      # the engine's bf16-cell observation format (env_get_observation_device_bf16)
    cells = torch.zeros((n,) + vs[:2] + (8,), dtype=torch.bfloat16, device=dev)
    cells[..., :vs[2]] = view.to(torch.bfloat16); cells[..., 7] = 1
    view = cells
for _ in range(3):
    pol.infer(view, feat)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    pol.infer(view, feat)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
flop = n * (2 * 121 * 32 * 63 + 2 * 81 * 32 * 288 + 2 * 2592 * 256 + 2 * F * 256 + 2 * 512 * (A + 1))
print("HIP policy (%s): %.3f ms for %d agents = %.1f TFLOP/s (useful flops%s), %.2f us per 1000 agents" % (
    "float32, policy_f32.hip" if F32 else "bf16, policy.hip", dt * 1e3, n, flop / dt / 1e12, ": %.3f of the 157.3 TFLOP/s f32 matrix peak" % (flop / dt / 157.3e12) if F32 else "", dt * 1e9 / n))
if F32 and "torch" in sys.argv:
    torch.backends.cudnn.allow_tf32 = False; torch.backends.cuda.matmul.allow_tf32 = False
    with torch.no_grad():
        for _ in range(2):
            qnet(view[:65536], feat[:65536])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            qnet(view[:65536], feat[:65536])
        torch.cuda.synchronize()
        print("torch float32: %.2f us per 1000 agents" % ((time.perf_counter() - t0) / 5 * 1e9 / 65536))
if "torch" in sys.argv and "cells" not in sys.argv and not F32:
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        for _ in range(2):
            qnet(view[:65536], feat[:65536])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            qnet(view[:65536], feat[:65536])
        torch.cuda.synchronize()
        print("torch bf16 autocast: %.2f us per 1000 agents" % ((time.perf_counter() - t0) / 5 * 1e9 / 65536))
