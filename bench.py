#!/usr/bin/env python3
"""bench.py — headline benchmark of the CTC forced-alignment hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--lattices B]

Workload (BASELINE.json configs[1]): synthetic log-probs T=50000 x V=64, S=5000 phonemes
(L=10001), beam_size=1000, max_move=4.  One "step" = one pass of the hot path (label prep +
forward DP + back-pointer recomputation and backtrace + outputs) over a batch of B independent lattices of that
shape, each with its own hash-generated inputs, already resident in HBM.  Metric = aligned
audio frames per second, whole job (all ranks).  N>1: one process per GPU (launched by
torch.distributed.run), lattices sharded across ranks, no data-path collective (weak scaling).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T, V, S, BEAM, MAX_MOVE = 50000, 64, 5000, 1000, 4
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


CK_FRAMES = 32   # kokoro-align_amd/csrc/ka_kernels.hpp kCkFrames: frames between stored score rings


def algorithmic_bytes_per_frame():
    """HBM bytes per frame that the algorithm has to move.

    Dominant kernel (forward_ck_kernel, scores only): the log-prob row read once (4V) + the score ring
    (1024 slots x 4 B) written once every CK_FRAMES frames.  Whole job: + the second kernel
    (backtrace_rc_kernel): the row once more (4V), the 128-cell window of a checkpoint and of the labels per
    chunk, and the three 4-byte outputs.
    For reference, SURVEY.md §8d's figure for the store-every-back-pointer formulation
    (KA_MODE_WAVE_EXACT): 4V + Wbar/4 forward, + 0.25 + 12 backtrace, Wbar from align.py:64-65."""
    L = 2 * S + 1
    cells = 0
    for t in range(T):
        lo = max(0, L * t // T - BEAM // 2)
        cells += min(lo + BEAM, L) - lo
    wbar = cells / T
    fwd = 4.0 * V + 4096.0 / CK_FRAMES
    job = fwd + 4.0 * V + (512.0 + 256.0) / CK_FRAMES + 12.0
    survey_fwd = 4.0 * V + wbar / 4.0
    return fwd, job, wbar, survey_fwd


def cpu_baseline(min_seconds):
    """The reference's CPU path cannot travel; time the oracle's per-frame NumPy port (same
    NumPy work per frame as kokoro_align/align.py:62-93) on a bounded sample of the same
    workload, single thread like the reference."""
    from oracle import oracle as O
    lp = O.hash_logprobs(T, V, 0)
    labels = O.hash_labels(S, V, 0)
    frames, dt = 0, 0.0
    while dt < min_seconds:              # whole 50000-frame lattices until the sample is long enough
        t0 = time.perf_counter()
        O.ctc_best_path_numpy(lp, labels, BEAM, MAX_MOVE)
        dt += time.perf_counter() - t0
        frames += T
    t1 = time.perf_counter()
    O.ctc_best_path_c(lp, labels, BEAM, MAX_MOVE)
    dt_c = time.perf_counter() - t1
    return {
        "value": frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
        "sample": f"{frames // T} x one full cfg2 lattice ({frames} frames; forward DP + backtrace + gathers), "
                  f"NumPy per-frame port of align.py:43-109, single thread like the reference, {dt:.1f} s; "
                  f"host has {os.cpu_count()} logical CPUs",
        "c_oracle_frames_per_s": T / dt_c,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--lattices", type=int, default=int(os.environ.get("KA_BENCH_LATTICES", "8192")),
                    help="lattices per GPU per step")
    ap.add_argument("--mode", default="auto", choices=["auto", "wave", "wave_exact", "workgroup"],
                    help="kernel form of the batch run (DESIGN.md section 4); auto = wave (checkpointed) at this batch size")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one rank per GPU as a CHILD process (never an exec,
        # and before anything here has touched the GPU), relay rank 0's JSON line and the exit code
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
        lines = [ln for ln in proc.stdout.decode("utf-8", "replace").splitlines() if ln.startswith("{")]
        if lines:
            print(lines[-1], flush=True)
        sys.exit(proc.returncode if proc.returncode else (0 if lines else 1))

    # stdout must carry exactly ONE JSON line: native libraries (RCCL prints a version banner) write to
    # fd 1 directly, so point fd 1 at stderr until the result is printed
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import kokoro_align_amd as ka
    from kokoro_align_amd.align import DeviceBatch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # rehearsal knobs (never set by the driver): several ranks on ONE GPU over gloo, to exercise the
    # multi-rank control flow on a single-GPU box
    rehearsal = os.environ.get("KA_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or os.environ.get("KA_FORCE_DIST") == "1":   # KA_FORCE_DIST: exercise the RCCL path with one rank
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)      # RCCL over xGMI
        # start-up collective of the pipeline: broadcast of the acoustic-model weights (2.3 MB)
        from kokoro_align_amd.sharding import broadcast_model_weights
        broadcast_model_weights(torch.device("cpu") if rehearsal else dev)

    lib = ka.load_library()
    B = args.lattices
    # ---- synthetic inputs, generated in HBM (hash generator == oracle's, so any lattice can be re-checked on CPU)
    # 8192 lattices = 8 wavefronts per SIMD = 215 GB (log-probs + back-pointers); halve on OOM.
    while True:
        try:
            lps = torch.empty((B, T, V), dtype=torch.float32, device=dev)
            labs = torch.empty((B, S), dtype=torch.int32, device=dev)
            batch = DeviceBatch([lps[i] for i in range(B)], [labs[i] for i in range(B)], BEAM, MAX_MOVE)
            batch.engine.reserve(batch.workspace_bytes() + (1 << 20))
            break
        except (RuntimeError, MemoryError) as exc:   # torch OOM or KA_ERR_NOMEM
            if B <= 64:
                raise
            print(f"[bench] {B} lattices do not fit ({type(exc).__name__}); retrying with {B // 2}", file=sys.stderr)
            lps = labs = batch = None
            torch.cuda.empty_cache()
            B //= 2
    stream = torch.cuda.current_stream().cuda_stream
    seed0 = rank * 1000003            # lattice i of this rank uses seed0 + i (rank 0, i = 0 is the golden cfg2 lattice)
    assert lib.ka_hash_logprobs_batch_f32(lps.data_ptr(), B, T, V, V, T * V, seed0, stream) == 0
    assert lib.ka_hash_labels_batch_i32(labs.data_ptr(), B, S, V, S, seed0, stream) == 0
    torch.cuda.synchronize()
    batch.engine.set_profiling(True)
    batch.engine.set_mode(args.mode)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier() if rehearsal else dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.run()
    barrier()
    fwd_ms, bt_ms, prep_ms, ga_ms = [], [], [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.run()                      # enqueue + stream sync + per-lattice status check
        k = batch.engine.last_kernel_ms()
        fwd_ms.append(k["forward"]); bt_ms.append(k["backtrace"]); prep_ms.append(k["prep"]); ga_ms.append(k["gather"])
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # parity spot check on the timed outputs (lattice 0 of rank 0 is the golden cfg2 lattice)
    ok = True
    if rank == 0:
        from tests.golden_util import g3_case
        g3 = g3_case()
        ok = bool(np.array_equal(batch.path[0].cpu().numpy(), g3["path"]))
        ends = torch.stack([p[-1] for p in batch.path[: min(B, 64)]]).cpu().numpy()
        ok = ok and bool((ends == 2 * S).all())
        # and, without any reference, on a spread of lattices: the float32 chain of the per-frame scores along the
        # returned path must be the forward pass's best cumulative score bit for bit (a path that is not THE best
        # path of its lattice misses it; this is what caught a store hazard that left every end position right)
        pick = sorted(set(np.linspace(0, B - 1, num=min(B, 256), dtype=np.int64).tolist()))
        chain = np.add.accumulate(torch.stack([batch.best_scores[i] for i in pick]).cpu().numpy(), axis=1, dtype=np.float32)[:, -1]
        ok = ok and bool(np.array_equal(chain.view(np.int32), np.asarray(batch.total, np.float32)[pick].view(np.int32)))

    # single-lattice latency (the serial T-chain; one wavefront busy on the whole chip)
    single = None
    if rank == 0:
        batch.engine.set_mode("auto")
        one = DeviceBatch([lps[0]], [labs[0]], BEAM, MAX_MOVE)   # 1 lattice, KA_MODE_AUTO
        one.engine.set_profiling(True)
        one.run()
        t1 = time.perf_counter()
        for _ in range(3):
            one.run()
        dt1 = (time.perf_counter() - t1) / 3
        k1 = one.engine.last_kernel_ms()
        single = {"frames_per_s": T / dt1, "ms": dt1 * 1e3, "forward_ms": k1["forward"], "backtrace_ms": k1["backtrace"], "gather_ms": k1["gather"]}

    if rank == 0:
        frames_per_step = B * T * world
        value = frames_per_step * args.steps / elapsed
        fwd_b, job_b, wbar, survey_fwd_b = algorithmic_bytes_per_frame()
        fwd_s = float(np.mean(fwd_ms)) * 1e-3
        checkpointed = args.mode in ("auto", "wave")   # KA_MODE_AUTO is the checkpointed form at every batch size
        if not checkpointed:     # every back-pointer stored: SURVEY.md 8d's bytes are this form's own
            fwd_b = survey_fwd_b
            job_b = survey_fwd_b + 12.25 + 4.0 * V / 2
        achieved = B * T * fwd_b / fwd_s / 1e9
        # HBM bytes of one forward launch cannot be counted from inside this process: they come from the
        # committed rocprofv3 PMC run (tools/prof.sh -> tools/summarize_profile.py) at the same batch size
        traffic = None
        tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tf):
            try:
                with open(tf) as f:
                    pt = json.load(f)
                if int(pt.get("lattices", -1)) == B and checkpointed:
                    traffic = pt.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "aligned audio-frames/sec (whole node), 50k x 5k lattice",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (hash-generated log-probs and labels, in HBM)",
            "config": {"workload": f"cfg2: T={T} x V={V} log-probs, S={S} phonemes (L={2 * S + 1}), beam_size={BEAM}, "
                                   f"max_move={MAX_MOVE}; batch of {B} independent lattices per GPU per step",
                       "lattices_per_gpu": B, "frames_per_step": frames_per_step, "parallelism": f"lattice-sharded x{world}",
                       "kernel_form": args.mode},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, same batch size)" if traffic else None,
                         "kernel": "forward_ck_kernel<4,false>" if checkpointed else
                                   ("forward_w16_kernel<4,false>" if args.mode == "wave_exact" else "forward_wg4_kernel<4,false>"),
                         "kernel_ms": fwd_s * 1e3,
                         "algorithmic_bytes_per_frame": fwd_b, "mean_band_width": wbar,
                         "survey_8d_bytes_per_frame": survey_fwd_b,
                         "achieved_with_survey_8d_bytes": B * T * survey_fwd_b / fwd_s / 1e9},
            "kernels_ms": {"prep": float(np.mean(prep_ms)), "forward": float(np.mean(fwd_ms)),
                           "backtrace": float(np.mean(bt_ms)), "gather": float(np.mean(ga_ms))},
            "job_bytes_per_frame": job_b,
            "single_lattice": single,
            "parity_spot_check": ok,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_baseline_seconds)
            out["speedup_vs_cpu_numpy_1core"] = value / out["cpu_baseline"]["value"]
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier() if rehearsal else dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
