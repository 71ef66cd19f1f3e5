import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import kokoro_align_amd.model as M
from kokoro_align_amd.model import load_model
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0); torch.manual_seed(0)
model = load_model(None, device=dev)
segs = []
for c in range(64):
    left = int(43000 * rng.uniform(0.6, 1.4))
    while left > 0:
        n = int(min(left, rng.integers(200, 1200))); segs.append(torch.from_numpy(rng.standard_normal((n, 40)).astype(np.float32))); left -= n
import cProfile, pstats
M.segment_logits_device(model, segs[:100])
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
out = M.segment_logits_device(model, segs); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
