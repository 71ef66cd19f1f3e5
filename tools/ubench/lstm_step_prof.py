import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from kokoro_align_amd import _lib
lib = _lib.load_library()
dev = torch.device("cuda", 0)
n, H, total = 3600, 128, 2500000
gin = torch.randn(total, 8 * H, device=dev)
out = torch.empty(total, 2 * H, device=dev)
h = torch.zeros(2, n, H, device=dev); c = torch.zeros(2, n, H, device=dev)
w = torch.randn(2, H, 4 * H, device=dev) * 0.05
rows = torch.randint(0, total, (2, 300, n), dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream(dev).cuda_stream
def loop(steps, do_bmm=True, do_k=True, rec=None):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(steps):
        if do_bmm: rec = torch.bmm(h[:, :n], w)
        if do_k: lib.ka_lstm_step_f32(gin.data_ptr(), gin.stride(0), rec.data_ptr(), rec.stride(0), c.data_ptr(), h.data_ptr(), h.stride(0), out.data_ptr(), out.stride(0), rows[:, t].data_ptr(), rows.stride(0), n, H, stream)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6
rec = torch.bmm(h, w)
print("warm", loop(50))
print("both   cpu/total us per step", loop(300))
print("bmm only", loop(300, True, False))
print("kernel only", loop(300, False, True, rec))
