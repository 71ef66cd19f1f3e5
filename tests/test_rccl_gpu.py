"""The RCCL backend, in-process: the calls the multi-GPU driver makes at start-up (bench.py: init_process_group("nccl",
device_id=...), sharding.broadcast_model_weights, barrier(device_ids=...)) and the sharded pipeline's failure flag, with a
world of ONE rank inside the pytest process - no child process, nothing exec'ed after GPU initialisation.  The gloo tests in
tests/test_sharding.py cover the logic at world size 2; this covers the backend the 8-GPU run uses
(BASELINE.json configs[3]: "sharded across 8 x MI355X, RCCL weight broadcast")."""
import datetime
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rccl_world_of_one_broadcast_barrier_and_failure_flag(tmp_path):
    import torch
    import torch.distributed as dist
    from kokoro_align_amd.sharding import broadcast_model_weights, gather_rank_stats, shard_for_rank
    assert not dist.is_initialized()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    store = dist.TCPStore("127.0.0.1", _free_port(), world_size=1, is_master=True, timeout=datetime.timedelta(seconds=60))
    dist.init_process_group("nccl", store=store, rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(minutes=5))
    try:
        assert dist.get_backend() == "nccl"
        model = broadcast_model_weights(dev)                      # one flat-buffer broadcast over RCCL
        assert sum(p.numel() for p in model.parameters()) == 579367 and next(model.parameters()).is_cuda
        # the broadcast left the weights as rank 0 initialised them (seed 0)
        from kokoro_align_amd.model import AudioToChar
        torch.manual_seed(0)
        want = AudioToChar()
        for a, b in zip(model.parameters(), want.parameters()):
            assert torch.equal(a.cpu(), b)
        dist.barrier(device_ids=[0])
        stats = gather_rank_stats(1234, 0.5, dev)
        assert stats == [(1234.0, 0.5)]
        t = torch.tensor([3], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert int(t.item()) == 3
        assert shard_for_rank([(100, 10), (50, 5), (70, 7)], 0, 1) == [0, 1, 2]
        # the sharded file driver end to end on this backend (its failure flag is an all-reduce on the device)
        from kokoro_align_amd import pipeline
        from golden_util import g4
        from oracle import oracle as O
        torch.manual_seed(5)
        voca_txt = g4()["voca_txt"]
        audio = []
        for i, seg_lens in enumerate([[150, 220], [300]]):
            base = str(tmp_path / f"ch{i}")
            audio.append(base + ".mp3")
            with pipeline.open_index_data_for_write(base + ".mfcc.npz") as w:
                for j, n in enumerate(seg_lens):
                    w.write(O.hash_logprobs(n, 40, 80 + 10 * i + j) * np.float32(0.5) + np.float32(2.0))
            with open(base + ".split.txt", "wt") as f:
                f.write("".join(f"{(j + 1) * 40000}\n" for j in range(len(seg_lens))))
            with open(base + ".voca.txt", "wt") as f:
                f.write(voca_txt)
        meta = str(tmp_path / "out" / "ds.metadata.txt")
        out = pipeline.process_alignment_sharded("ds", audio, meta, model=model.eval(), verbose=False)
        assert out == meta and os.path.exists(meta)
        assert all(os.path.exists(a[:-4] + ".align.txt") for a in audio)
    finally:
        dist.destroy_process_group()
    assert not dist.is_initialized()
