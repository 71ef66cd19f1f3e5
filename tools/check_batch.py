"""Batch of cfg2 lattices in a chosen kernel form; compare a sample of them with the CPU oracle."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
import kokoro_align_amd as ka
from kokoro_align_amd.align import DeviceBatch
from oracle import oracle as O

B = int(sys.argv[1]); mode = sys.argv[2]; T = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
V, S = 64, T // 10
lib = ka.load_library()
lps = torch.empty((B, T, V), dtype=torch.float32, device="cuda")
labs = torch.empty((B, S), dtype=torch.int32, device="cuda")
assert lib.ka_hash_logprobs_batch_f32(lps.data_ptr(), B, T, V, V, T * V, 0, None) == 0
assert lib.ka_hash_labels_batch_i32(labs.data_ptr(), B, S, V, S, 0, None) == 0
torch.cuda.synchronize()
batch = DeviceBatch([lps[i] for i in range(B)], [labs[i] for i in range(B)], 1000, 4)
batch.engine.set_mode(mode)
for rep in range(3):
    batch.run()
    ends = np.array([int(p[-1]) for p in batch.path])
    nbad_end = int((ends != 2 * S).sum())
    bad = []
    wrong = np.nonzero(ends != 2 * S)[0][:4].tolist()
    sample = np.random.default_rng(rep).integers(0, B, size=10).tolist()
    for i in sorted(set([0, 1, B - 1] + wrong + sample)):
        want = O.ctc_best_path_c(O.hash_logprobs(T, V, i), O.hash_labels(S, V, i), 1000, 4)
        got = batch.path[i].cpu().numpy()
        if not np.array_equal(got, want[0]):
            j = int(np.argmax(got != want[0]))
            bad.append((i, j, int((got != want[0]).sum()), got[j:j+4].tolist(), want[0][j:j+4].tolist(), int(got[-1])))
        elif not (np.array_equal(batch.best_labels[i].cpu().numpy(), want[1])
                  and np.array_equal(batch.best_scores[i].cpu().numpy().view(np.int32), want[2].view(np.int32))):
            bad.append((i, "labels/scores differ"))
    # every lattice, reference-free: float32 chain of best_scores along the path == the forward pass's total
    off = []
    for a in range(0, B, 512):
        sc = torch.stack(batch.best_scores[a:a + 512]).cpu().numpy()
        chain = np.add.accumulate(sc, axis=1, dtype=np.float32)[:, -1]
        tot = np.asarray(batch.total, np.float32)[a:a + 512]
        off += (a + np.nonzero(chain.view(np.int32) != tot.view(np.int32))[0]).tolist()
    print(f"rep {rep} mode={mode} B={B} T={T}: ends wrong: {nbad_end} (idx {np.nonzero(ends != 2 * S)[0][:8].tolist()}); "
          f"chain != total on {len(off)} lattices {off[:8]}; sample mismatches: {bad}")
