"""Multi-GPU layout on CPU: LPT sharding is a partition; the weight broadcast works over gloo
with world_size 2 (the RCCL path uses the same code with backend 'nccl')."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kokoro_align_amd.sharding import lattice_cost, lpt_partition, shard_for_rank


def test_lpt_partition_is_a_balanced_partition():
    costs = [lattice_cost(t, s) for t, s in [(50000, 5000), (81140, 2000), (20000, 300), (160000, 22000),
                                             (30000, 4000), (30000, 4000), (1000, 10), (99999, 14000)]]
    for n in (1, 2, 3, 8):
        parts = lpt_partition(costs, n)
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(len(costs)))
        loads = [sum(costs[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(costs)
    assert lpt_partition([], 4) == [[], [], [], []]
    assert lattice_cost(100, 3, 1000) == 100 * 7


def test_shard_for_rank_covers_everything():
    shapes = [(1000 * (i + 1), 100 * (i + 1)) for i in range(13)]
    seen = sorted(i for r in range(4) for i in shard_for_rank(shapes, r, 4))
    assert seen == list(range(13))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kokoro_align_amd.sharding import broadcast_model_weights, gather_rank_stats
        model = broadcast_model_weights(torch.device("cpu"))
        n = sum(p.numel() for p in model.parameters())
        digest = float(sum(p.double().sum() for p in model.parameters()))
        shapes = [(1000 * (i + 1), 100 * (i + 1)) for i in range(9)]
        mine = shard_for_rank(shapes, rank, world)
        stats = gather_rank_stats(sum(shapes[i][0] for i in mine), 1.0 + rank, torch.device("cpu"))
        q.put((rank, n, digest, mine, stats))
    finally:
        dist.destroy_process_group()


def test_weight_broadcast_and_sharding_gloo_ws2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, n0, d0, mine0, st0), (r1, n1, d1, mine1, st1) = res
    assert n0 == n1 == 579367                     # AudioToChar parameter count
    assert d0 == d1                               # identical weights after the broadcast
    assert sorted(mine0 + mine1) == list(range(9)) and not set(mine0) & set(mine1)
    assert st0 == st1 and sum(f for f, _ in st0) == sum(1000 * (i + 1) for i in range(9))


# ------------------------------------------------------------------------------------------
# process_alignment_sharded: two ranks over gloo write the same files as one process.  The DP stage is the oracle
# here (this is the CPU suite; the product's DP needs the GPU - tests/test_pipeline_gpu.py runs the real one).
# ------------------------------------------------------------------------------------------
def _oracle_best_path_files(logits_files, voca_files, out_files, device=None, logits_on_device=None, host_softmax=False):
    import numpy as np
    from oracle import oracle as O
    from kokoro_align_amd.align import _host_log_softmax
    from kokoro_align_amd.transcript import read_transcript
    written = []
    for lf, vf, bf in zip(logits_files, voca_files, out_files):
        if os.path.exists(bf):
            continue
        with np.load(lf) as f:
            logits = f["data"]
        p, l, s = O.ctc_best_path_c(_host_log_softmax(logits), read_transcript(vf), 1000, 4)
        np.savez(bf, best_path=p, best_labels=l, best_scores=s)
        written.append(bf)
    return written


_SYLL = ["k o", "n i", "ch i", "w a", "s e", "k a", "i", "d e", "s u", "m a", "t o", "r e"]


def _make_dataset(root, n_files=5):
    """Synthetic upstream outputs of n_files recordings: voca.txt, mfcc.npz (indices only matter), split.txt,
    logits.npz + greed.txt (so that no model is needed)."""
    import numpy as np
    from kokoro_align_amd.encoder import encode_text
    rng = np.random.default_rng(11)
    audio = []
    for k in range(n_files):
        base = os.path.join(root, f"rec{k}")
        n_tok = 6 + 3 * k
        toks = [_SYLL[int(i)] for i in rng.integers(0, len(_SYLL), n_tok)]
        with open(base + ".voca.txt", "wt") as f:
            for t in toks:
                f.write(f"{t.replace(' ', '')}|{t}\n")
        labels = encode_text(" ".join(toks))
        labels = labels[labels != 0] if False else labels
        seg_lens = [int(x) for x in rng.integers(40, 90, 2 + k % 3)]
        T = sum(seg_lens)
        ends = np.cumsum(seg_lens).astype(np.int32)
        logits = rng.standard_normal((T, 39)).astype(np.float32)
        np.savez(base + ".mfcc.npz", indices=ends, data=np.zeros((T, 1), np.float32))
        np.savez(base + ".logits.npz", indices=ends, data=logits)
        with open(base + ".greed.txt", "wt") as f:
            f.write("\n".join(f"{i + 1}|x" for i in range(len(seg_lens))) + "\n")
        with open(base + ".split.txt", "wt") as f:
            f.write("\n".join(str(int(e) * 256) for e in ends) + "\n")
        audio.append(base + ".mp3")
    return audio


def _sharded_worker(rank, world, port, root, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kokoro_align_amd import pipeline
        audio = sorted(os.path.join(root, f) for f in os.listdir(root) if f.endswith(".voca.txt"))
        audio = [a[:-len(".voca.txt")] + ".mp3" for a in audio]
        out = pipeline.process_alignment_sharded("ds", audio, os.path.join(root, "out", "ds.metadata.txt"), verbose=False,
                                                 best_path_files_fn=_oracle_best_path_files)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_process_alignment_sharded_ws2_writes_what_one_process_writes(tmp_path):
    from kokoro_align_amd import pipeline
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    os.makedirs(one); os.makedirs(two)
    audio_one = _make_dataset(one)
    _make_dataset(two)
    pipeline.process_alignment("ds", audio_one, os.path.join(one, "out", "ds.metadata.txt"), verbose=False,
                               best_path_files_fn=_oracle_best_path_files)
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, two, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1] is None and res[0] == os.path.join(two, "out", "ds.metadata.txt")
    names = sorted(f for f in os.listdir(one) if f.endswith((".align.txt", ".best_path.npz")))
    assert names == sorted(f for f in os.listdir(two) if f.endswith((".align.txt", ".best_path.npz"))) and len(names) == 10
    for f in names:
        a, b = os.path.join(one, f), os.path.join(two, f)
        if f.endswith(".txt"):
            assert open(a).read() == open(b).read(), f
        else:
            import numpy as np
            with np.load(a) as x, np.load(b) as y:
                assert all(np.array_equal(x[k], y[k]) for k in ("best_path", "best_labels", "best_scores")), f
    assert open(os.path.join(one, "out", "ds.metadata.txt")).read() == open(os.path.join(two, "out", "ds.metadata.txt")).read()


# ------------------------------------------------------------------------------------------
# one rank fails: every rank raises, nobody hangs, rank 0 does not merge a partial dataset
# ------------------------------------------------------------------------------------------
def _failing_best_path_files(logits_files, voca_files, out_files, **kw):
    if any(os.path.basename(f).startswith("rec4.") for f in logits_files):
        raise OSError("rec4 cannot be aligned (injected)")
    return _oracle_best_path_files(logits_files, voca_files, out_files, **kw)


def _sharded_failing_worker(rank, world, port, root, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kokoro_align_amd import pipeline
        audio = sorted(os.path.join(root, f) for f in os.listdir(root) if f.endswith(".voca.txt"))
        audio = [a[:-len(".voca.txt")] + ".mp3" for a in audio]
        try:
            pipeline.process_alignment_sharded("ds", audio, os.path.join(root, "out", "ds.metadata.txt"), verbose=False,
                                               best_path_files_fn=_failing_best_path_files)
            q.put((rank, "returned"))
        except OSError as exc:
            q.put((rank, "own:" + str(exc)))
        except RuntimeError as exc:
            q.put((rank, "other:" + str(exc)))
    finally:
        dist.destroy_process_group()


def test_process_alignment_sharded_ws2_one_rank_fails_every_rank_raises(tmp_path):
    root = str(tmp_path / "ds")
    os.makedirs(root)
    _make_dataset(root)
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_failing_worker, args=(r, world, port, root, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    kinds = sorted(v.split(":")[0] for v in res.values())
    assert kinds == ["other", "own"], res          # the owner of rec4 re-raises its error, the other rank learns of it
    assert not os.path.exists(os.path.join(root, "out", "ds.metadata.txt"))
