"""Several launches in flight on one GPU: G engines, G HIP streams, G host threads.

Why: a launch of the checkpointed form is two kernels of different character, one after the other - the scores-only
forward pass keeps the vector ALUs ~85 % busy, the backtrace's recompute-and-walk ~75 % with the rest spent in serial
chains - and within ONE launch the second cannot start before the first has finished for every
lattice.  Independent launches have no such barrier: while one stream walks its lattices back, another one's forward
pass fills the issue slots the walk leaves empty.  Measured on MI355X with 8192 cfg2 lattices (tools/overlap_probe.py,
profiles/r03_overlap_probe.jsonl): one engine 73.7 ms per pass over all of them, two engines with 4096 each 64-66 ms,
three with 2730 each 63.7 ms, four with 2048 each 63-65 ms - as long as every stream has a hardware queue of its own (below).

This is how a caller with more than one batch of work should drive the library (the datasets of a corpus, the
sub-batches of a large one): the reference loops datasets and files one after the other (run_example.py:283-304,
:248-254); they share nothing, so their launches may overlap freely.  include/kokoro_align_amd.h: one ka_engine per
host thread / stream, distinct engines are independent.
"""
import threading
import time

from . import _lib


class StreamedAligner:
    """Runs DeviceBatch objects on ``n_streams`` engines side by side.

    Batch k is served by worker k % n_streams (its engine, its stream, its host thread); a worker runs its batches in
    the order given.  Every worker owns a device workspace sized for its largest batch, so the memory cost is
    ``n_streams`` workspaces of 1 / n_streams of the work each - the same total as one engine over everything."""

    def __init__(self, n_streams=4, device=0, mode="auto", backtrace="auto", profiling=False):
        import torch
        if n_streams < 1:
            raise ValueError("n_streams must be >= 1")
        self.device = int(device)
        self.n_streams = int(n_streams)
        self.engines = [_lib.Engine(self.device) for _ in range(self.n_streams)]
        # Worker 0 launches on the caller's current stream, the others on HIP streams of their own, created here one after
        # the other (ka_stream_create).  The HIP runtime spreads a process's streams over FOUR hardware queues per device
        # and the device's default stream holds one of them; streams that share a queue take turns.  Measured with 8192
        # cfg2 lattices (tools/overlap_probe.py): default stream + 3 own streams 63-65 ms per pass, 3 own streams 63.9,
        # 4 own streams 74.6 (two of them share a queue), 4 streams from PyTorch's pool anything between 63 and 79
        # (which pool streams share a queue depends on what was handed out before).  So: at most 4 launches in flight,
        # and none of them on a pooled stream.
        import ctypes
        lib = _lib.load_library()
        self._own = []
        with torch.cuda.device(self.device):
            self.streams = [torch.cuda.current_stream(self.device)]
            for _ in range(self.n_streams - 1):
                h = ctypes.c_void_p()
                _lib.check(lib.ka_stream_create(self.device, ctypes.byref(h)), "ka_stream_create")
                self._own.append(h)
                self.streams.append(torch.cuda.ExternalStream(h.value, device=self.device))
        for e in self.engines:
            e.set_mode(mode)
            e.set_backtrace(backtrace)
            e.set_profiling(profiling)
        self.profiling = bool(profiling)
        self.kernel_ms = []          # with profiling: (worker, batch index, {"prep", "forward", "backtrace", "gather"}) per run

    def close(self):
        for e in self.engines:
            e.close()
        self.engines = []
        lib = _lib.load_library()
        for h in getattr(self, "_own", []):
            lib.ka_stream_destroy(self.device, h)
        self._own = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bind(self, batches):
        """Hand every batch to its worker's engine and size the workspaces once (no allocation inside run())."""
        need = [0] * self.n_streams
        for k, b in enumerate(batches):
            g = k % self.n_streams
            b.engine = self.engines[g]
            need[g] = max(need[g], b.workspace_bytes() + (1 << 20))
        for g, nbytes in enumerate(need):
            if nbytes:
                self.engines[g].reserve(nbytes)
        return batches

    def run(self, batches, repeat=1, raise_on_error=True, stagger_s=0.0):
        """Every batch ``repeat`` times (a worker cycles through ITS batches, in order, ``repeat`` times).  Returns when all
        of it has finished on the device; the per-lattice status of each batch's LAST run is in ``batch.status``.
        ``stagger_s``: worker g starts g * stagger_s late (the phases of equal-sized batches then interleave from the
        first launch on instead of drifting apart by themselves)."""
        import torch
        mine = [[(k, b) for k, b in enumerate(batches) if k % self.n_streams == g] for g in range(self.n_streams)]
        errors = []
        self.kernel_ms = []
        lock = threading.Lock()
        gate = threading.Barrier(sum(1 for m in mine if m) + 1)

        def worker(g):
            try:
                torch.cuda.set_device(self.device)
                with torch.cuda.stream(self.streams[g]):
                    try:
                        gate.wait(timeout=120)
                    except threading.BrokenBarrierError:
                        return                # another worker failed before the start: its error is the one reported
                    if stagger_s:
                        time.sleep(g * stagger_s)
                    for _ in range(repeat):
                        for k, b in mine[g]:
                            b.enqueue()
                            b.finish(raise_on_error)
                            if self.profiling:
                                ms = self.engines[g].last_kernel_ms()
                                with lock:
                                    self.kernel_ms.append((g, k, ms))
            except BaseException as exc:      # re-raised in the caller's thread
                with lock:
                    errors.append(exc)
                gate.abort()                  # a worker that fails before the gate must not leave the others waiting at it

        threads = [threading.Thread(target=worker, args=(g,), name=f"ka-stream-{g}") for g in range(self.n_streams) if mine[g]]
        for t in threads:
            t.start()
        try:
            gate.wait(timeout=120)
        except threading.BrokenBarrierError:
            pass
        for t in threads:
            t.join()
        if errors:
            real = [x for x in errors if not isinstance(x, threading.BrokenBarrierError)]
            raise (real or errors)[0]
        return [b.status for b in batches]


def split_device_batch(log_probs, labels, n_parts, beam_size=1000, max_move=4):
    """One list of device-resident lattices -> ``n_parts`` DeviceBatch objects of (almost) equal size, by position:
    lattice i goes to part i * n_parts // n.  Returns (batches, index lists)."""
    from .align import DeviceBatch
    n = len(log_probs)
    n_parts = max(1, min(int(n_parts), n))
    bounds = [n * j // n_parts for j in range(n_parts + 1)]
    parts = [list(range(bounds[j], bounds[j + 1])) for j in range(n_parts)]
    batches = [DeviceBatch([log_probs[i] for i in idx], [labels[i] for i in idx], beam_size, max_move) for idx in parts]
    return batches, parts
