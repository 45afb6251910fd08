"""Development probe (GPU box): bench.small_world_extras (BASELINE config 2 through the call sequence and through EnvBatch, 1 / 8 / 32 worlds) under the MAGENT_TUNE of the environment."""
import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, magent_amd, bench
dev = torch.device("cuda", 0)
r = bench.small_world_extras(torch, magent_amd, dev)
print(os.environ.get("MAGENT_TUNE", "(defaults)"), {k: round(v["ms_per_cycle"], 4) for k, v in r.items()})
