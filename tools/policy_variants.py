"""Development helper (GPU box): time builds of policy.hip that were compiled with parts of a kernel switched off (elimination runs).
usage: python tools/policy_variants.py lib1.so lib2.so ...   (each built from magent_amd/csrc/policy.hip alone)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from magent_amd.builtin.torch_model.dqn import _QNet
from magent_amd.builtin.torch_model.hip_policy import HipDqnPolicy

n, reps = 131072, int(os.environ.get("REPS", "600"))
dev = torch.device("cuda", 0)
vs, F, A = (13, 13, 7), 34, 21
qnet = _QNet(vs, (F,), A, True, True).to(dev)
view = (torch.rand((n,) + vs, device=dev) < 0.3).float()
feat = torch.rand((n, F), device=dev)
pol = HipDqnPolicy(qnet, vs, (F,), A, dev, chunk=n)
cells = torch.zeros((n,) + vs[:2] + (8,), dtype=torch.bfloat16, device=dev)
cells[..., :vs[2]] = view.to(torch.bfloat16); cells[..., 7] = 1
pol.pack()
actions = torch.empty(n, dtype=torch.int32, device=dev)
work = torch.empty((n + 127) // 128 * 128 * (pol.k_dense + 64) * 2 + 2048, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream(dev).cuda_stream
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    def call():
        rc = lib.policy_dqn_infer_bf16(ctypes.byref(pol.shape), ctypes.byref(pol._w), ctypes.c_void_p(cells.data_ptr()), ctypes.c_void_p(feat.data_ptr()), n,
                                       ctypes.c_void_p(work.data_ptr()), ctypes.c_void_p(actions.data_ptr()), None, ctypes.c_void_p(stream))
        assert rc == 0
    for _ in range(reps // 2):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    print("%-28s %.3f ms" % (os.path.basename(path), (time.perf_counter() - t0) / reps * 1e3), flush=True)
