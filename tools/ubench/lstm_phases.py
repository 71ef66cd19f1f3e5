"""Where the time of lstm_logits_device goes (8-hour synthetic book)."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import kokoro_align_amd.model as M
from kokoro_align_amd.model import load_model
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0); torch.manual_seed(0)
model = load_model(None, device=dev)
lens = []
for c in range(64):
    left = int(43000 * rng.uniform(0.6, 1.4))
    while left > 0:
        n = int(min(left, rng.integers(200, 1200))); lens.append(n); left -= n
data = torch.from_numpy(rng.standard_normal((sum(lens), 40)).astype(np.float32))
ends = np.cumsum(lens)
for persistent in (True, False):
    M.lstm_logits_device(model, data, ends, persistent=persistent); torch.cuda.synchronize()
    t0 = time.perf_counter(); M.lstm_logits_device(model, data, ends, persistent=persistent); torch.cuda.synchronize()
    print("persistent" if persistent else "stepwise", "total ms", (time.perf_counter() - t0) * 1e3)
# pieces
t0 = time.perf_counter(); x = data.to(dev); torch.cuda.synchronize(); print("H2D ms", (time.perf_counter() - t0) * 1e3)
w = torch.randn(1024, 40, device=dev); b = torch.randn(1024, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter(); g = torch.addmm(b, x, w.t()); torch.cuda.synchronize(); print("gin GEMM layer0 ms", (time.perf_counter() - t0) * 1e3)
x2 = torch.randn(x.shape[0], 256, device=dev); w2 = torch.randn(1024, 256, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter(); g2 = torch.addmm(b, x2, w2.t()); torch.cuda.synchronize(); print("gin GEMM layer1 ms", (time.perf_counter() - t0) * 1e3)
from kokoro_align_amd import _lib
lib = _lib.load_library()
order = np.argsort(-np.array(lens), kind="stable"); sl = np.array(lens)[order]; offs = np.concatenate([[0], np.cumsum(sl)[:-1]])
d_off = torch.from_numpy(offs.astype(np.int32)).to(dev); d_len = torch.from_numpy(sl.astype(np.int32)).to(dev)
whh = torch.randn(2, 512, 128, device=dev) * 0.05
out = torch.empty(x.shape[0], 256, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lib.ka_lstm_layer_f32(g.data_ptr(), g.stride(0), whh.data_ptr(), out.data_ptr(), out.stride(0), d_off.data_ptr(), d_len.data_ptr(), len(lens), 128, st)
    torch.cuda.synchronize(); print("lstm_layer kernel ms", (time.perf_counter() - t0) * 1e3, "steps", int(sl[0]), "tiles", (len(lens) + 31) // 32)
