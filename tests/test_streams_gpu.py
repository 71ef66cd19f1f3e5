"""Several launches in flight on one GPU (kokoro_align_amd.streams.StreamedAligner): results must not depend on what else
runs beside a launch.  Reference: run_example.py:283-304 / :248-254 loop datasets and files one after the other; they share
nothing, so here their launches overlap on G engines / HIP streams / host threads."""
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def test_streamed_sub_batches_of_a_book_match_the_oracle():
    import torch
    from kokoro_align_amd import workloads as W
    from kokoro_align_amd.streams import StreamedAligner, split_device_batch
    _, shapes = W.kokoro_book()
    lps, labs = W.device_book(shapes)
    threads = max(1, min(16, (os.cpu_count() or 2) - 1))
    want = O.lattice_batch_c(shapes, W.V_MODEL, W.BOOK_SEED0, 1000, 4, threads=threads)
    for G, mode, bt in ((4, "auto", "auto"), (3, "wave", "serial"), (2, "tiled", "parallel")):
        sa = StreamedAligner(G, torch.cuda.current_device(), mode, bt)
        try:
            batches, parts = split_device_batch(lps, labs, 2 * G)       # two launches per stream, every one repeated
            assert sorted(i for p in parts for i in p) == list(range(len(shapes)))
            sa.bind(batches)
            status = sa.run(batches, repeat=3, stagger_s=0.002)
            for b, idx, st in zip(batches, parts, status):
                assert not (st != 0).any()
                for k, i in enumerate(idx):
                    w = want[i]
                    assert np.array_equal(b.path[k].cpu().numpy(), w[0]), (G, mode, i)
                    assert np.array_equal(b.best_labels[k].cpu().numpy(), w[1])
                    assert np.array_equal(b.best_scores[k].cpu().numpy().view(np.int32), w[2].view(np.int32))
                    assert np.float32(b.total[k]).view(np.int32) == np.float32(w[3]).view(np.int32)
        finally:
            sa.close()


def test_a_failing_launch_raises_in_the_caller_and_leaves_the_others_done():
    """an empty beam (the reference's ValueError, align.py:101) in ONE stream's batch: run() re-raises it in the calling
    thread after every worker has finished; the other streams' results are complete"""
    import torch
    from kokoro_align_amd import workloads as W
    from kokoro_align_amd.align import DeviceBatch
    from kokoro_align_amd.streams import StreamedAligner
    good = [(3000, 400), (2500, 300)]
    bad = [(20, 400)]                # 20 frames for a band that has to travel 801 positions: nothing is live at the end
    lg, bg = W.device_book(good, seed0=7)
    lb, bb = W.device_book(bad, seed0=9)
    sa = StreamedAligner(2, torch.cuda.current_device())
    try:
        batches = sa.bind([DeviceBatch(lg, bg), DeviceBatch(lb, bb)])
        with pytest.raises(ValueError, match="argmax of an empty sequence"):
            sa.run(batches)
        status = sa.run(batches, raise_on_error=False)
        assert status[0].tolist() == [0, 0] and status[1].tolist() == [-1]
        want = O.lattice_batch_c(good, W.V_MODEL, 7, threads=2)
        for k in range(2):
            assert np.array_equal(batches[0].path[k].cpu().numpy(), want[k][0])
    finally:
        sa.close()


def test_corpus_dataset_by_dataset_on_four_engines():
    """The reference's own loop (run_example.py:283-304: one dataset after the other) overlapped on four engines: several tile
    pipelines share the device and every engine reuses ONE workspace for launches of different shapes.  Round 4 regression: the
    sentinel refill behind a launch's tiles was still writing when the next, differently shaped launch laid its descriptors out
    in the same memory (a GPU memory fault; tools/stress_streams_tiled.py)."""
    import torch
    from kokoro_align_amd import workloads as W
    from kokoro_align_amd.align import DeviceBatch
    from kokoro_align_amd.streams import StreamedAligner
    per_ds = []
    for k, (_, shapes) in enumerate(W.corpus()):
        lps, labs = W.device_book(shapes, seed0=W.corpus_seed0(k))
        per_ds.append((lps, labs, shapes))
    ref = []
    for lps, labs, _ in per_ds:       # reference: one wavefront per lattice, serial backtrace, one dataset at a time
        b = DeviceBatch(lps, labs)
        b.engine.set_mode("wave"); b.engine.set_backtrace("serial")
        try:
            b.run()
        finally:
            b.engine.set_mode("auto"); b.engine.set_backtrace("auto")
        ref.append([p.clone() for p in b.path])
    sa = StreamedAligner(4)
    try:
        batches = sa.bind([DeviceBatch(lps, labs) for lps, labs, _ in per_ds])
        for rep in range(3):
            status = sa.run(batches, repeat=2)
            for k, (b, st) in enumerate(zip(batches, status)):
                assert not (st != 0).any(), (rep, k)
                assert all(torch.equal(a, c) for a, c in zip(ref[k], b.path)), (rep, k)
    finally:
        sa.close()
