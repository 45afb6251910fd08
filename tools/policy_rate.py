"""Development helper (GPU box): time of the DQN inference kernels (magent_amd/csrc/policy.hip) on n agents of the battle
observation shape, against the PyTorch / MIOpen forward pass under bf16 autocast."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from magent_amd.builtin.torch_model.dqn import _QNet
from magent_amd.builtin.torch_model.hip_policy import HipDqnPolicy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
vs, F, A = (13, 13, 7), 34, 21
qnet = _QNet(vs, (F,), A, True, True).to(dev)
view = (torch.rand((n,) + vs, device=dev) < 0.3).float()
feat = torch.rand((n, F), device=dev)
pol = HipDqnPolicy(qnet, vs, (F,), A, dev, chunk=int(os.environ.get("CHUNK", n)))
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
print("HIP policy: %.3f ms for %d agents = %.1f TFLOP/s (useful flops), %.2f us per 1000 agents" % (dt * 1e3, n, flop / dt / 1e12, dt * 1e9 / n))
if "torch" in sys.argv and "cells" not in sys.argv:
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        for _ in range(2):
            qnet(view[:65536], feat[:65536])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            qnet(view[:65536], feat[:65536])
        torch.cuda.synchronize()
        print("torch bf16 autocast: %.2f us per 1000 agents" % ((time.perf_counter() - t0) / 5 * 1e9 / 65536))
