#!/usr/bin/env python3
"""bench.py — benchmark of the CTC forced-alignment hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload cfg2|book] [--lattices B] [--mode M] [--backtrace HOW]

--workload cfg2 (default; BASELINE.json configs[1], the configuration the metric is quoted on): synthetic log-probs
    T=50000 x V=64, S=5000 phonemes (L=10001), beam_size=1000, max_move=4.  One "step" = one pass of the hot path (label
    prep + forward DP + backtrace + outputs) over a batch of B independent lattices of that shape per GPU, each with its
    own hash-generated inputs, already resident in HBM.  N>1: every rank has its own B lattices ("scaling": "weak").
--workload book (BASELINE.json configs[3]): the Meian stand-in (120 chapter lattices, 5.27 M frames, V=39), the chapters
    split over the N ranks by kokoro_align_amd.sharding.shard_for_rank; a step = the whole book once ("scaling":
    "strong").  At N=1 this is configs[2]/[3] on one GPU.
Metric = aligned audio frames per second, whole job (all ranks); no data-path collective.  N>1 without a launcher
starts one rank per GPU itself (a child process running torch.distributed.run, before anything touches the GPU).

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
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
CK_FRAMES = 32          # kokoro-align_amd/csrc/ka_kernels.hpp kCkFrames: frames between stored score rings
N_SIMD = 1024           # 256 CUs x 4


def algorithmic_bytes_per_frame():
    """HBM bytes per frame that the algorithm has to move.

    Dominant kernel (forward_ck_kernel, scores only): the log-prob row read once (4V) + the score ring (1024 slots x 4 B)
    written once every CK_FRAMES frames.  Whole job: + the second kernel (backtrace_rc_kernel): the row once more (4V),
    the 128-cell window of a checkpoint and of the labels per chunk, and the three 4-byte outputs.  For reference,
    SURVEY.md §8d's figure for the store-every-back-pointer formulation (KA_MODE_WAVE_EXACT): 4V + Wbar/4 forward,
    + 0.25 + 12 backtrace, Wbar from align.py:64-65."""
    L = 2 * S + 1
    cells = 0
    for t in range(T):
        lo = max(0, L * t // T - BEAM // 2)
        cells += min(lo + BEAM, L) - lo
    wbar = cells / T
    fwd = 4.0 * V + 4096.0 / CK_FRAMES
    job = fwd + 4.0 * V + (512.0 + 256.0) / CK_FRAMES + 12.0
    survey_fwd = 4.0 * V + wbar / 4.0
    survey_job = survey_fwd + 12.25
    return fwd, job, wbar, survey_fwd, survey_job


def valu_bound(B, fwd_ms):
    """The forward kernel is bound by vector-instruction issue, not by HBM: what the recurrence itself
    needs per frame and wavefront - 24 max (7 candidates per blank/label pair of four cells, v_max3 takes three), 8 packed
    adds of the emissions, 3 DPP moves for the cells of the lane below - at the measured issue cost of 4 cycles per
    wave-instruction with 8 wavefronts per SIMD (profiles/r01_ubench_issue_rates.txt), against the kernel's time."""
    min_instr = 24 + 8 + 3
    measured_instr = None
    f = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(f) as fh:
            measured_instr = json.load(fh).get("forward_valu_per_frame")
    except Exception:
        pass
    clock_ghz = 1.9   # under this load (GRBM_GUI_ACTIVE / kernel time, profiles/r01j_summary.json)
    waves_per_simd = max(1.0, B / N_SIMD)
    bound_ms = B * T * min_instr * 4.0 / (N_SIMD * clock_ghz * 1e9) * 1e3 if B >= N_SIMD else T * min_instr * 4.0 * waves_per_simd / (clock_ghz * 1e9) * 1e3
    return {"min_vector_instructions_per_frame": min_instr, "measured_vector_instructions_per_frame": measured_instr or 46.3,
            "cycles_per_wave_instruction": 4.0, "clock_ghz_under_load": clock_ghz, "bound_ms": bound_ms, "frac": bound_ms / fwd_ms}


def _numpy_one(_):
    from oracle import oracle as O
    lp = O.hash_logprobs(T, V, 0)
    labels = O.hash_labels(S, V, 0)
    t0 = time.perf_counter()
    O.ctc_best_path_numpy(lp, labels, BEAM, MAX_MOVE)
    return time.perf_counter() - t0


def cpu_baseline(min_seconds, whole_host=True):
    """The reference's CPU path cannot travel; time the oracle's per-frame NumPy port (same NumPy work per frame as
    kokoro_align/align.py:62-93) on a bounded sample of the same workload: one process like the reference, then
     one lattice per process on every core this job may use, and the C oracle next to both."""
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
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    out = {
        "value": frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
        "sample": f"{frames // T} x one full cfg2 lattice ({frames} frames; forward DP + backtrace + gathers), "
                  f"NumPy per-frame port of align.py:43-109, single thread like the reference, {dt:.1f} s; "
                  f"host: {cpu_model}, {os.cpu_count()} logical CPUs",
        "c_oracle_frames_per_s": T / dt_c,
    }
    if whole_host:
        import multiprocessing as mp
        from concurrent.futures import ThreadPoolExecutor
        try:
            share = len(os.sched_getaffinity(0))
        except AttributeError:
            share = os.cpu_count() or 1
        procs = max(1, min(share, 16))   # this job's share of the host (a 1-GPU box: 16 cores)
        t2 = time.perf_counter()
        with mp.get_context("spawn").Pool(procs) as pool:
            pool.map(_numpy_one, range(procs))
        dt_np = time.perf_counter() - t2
        t3 = time.perf_counter()
        with ThreadPoolExecutor(procs) as ex:      # (ctypes releases the GIL: threads are cores here)
            list(ex.map(lambda _: O.ctc_best_path_c(lp, labels, BEAM, MAX_MOVE), range(procs)))
        dt_cw = time.perf_counter() - t3
        out["whole_host"] = {
            "processes": procs, "cpu_model": cpu_model, "logical_cpus": os.cpu_count(),
            "numpy_port_frames_per_s": procs * T / dt_np, "c_oracle_frames_per_s": procs * T / dt_cw,
            "sample": f"{procs} processes x one cfg2 lattice each, wall {dt_np:.1f} s (incl. process start-up and input generation) / C oracle {dt_cw:.2f} s",
        }
    return out


def relay_to_children(args):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU as a CHILD process (never an exec, and
    before anything here has touched the GPU), relay rank 0's JSON line and the exit code."""
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


def timed_batch(batch, reps):
    batch.run()
    t0 = time.perf_counter()
    for _ in range(reps):
        batch.run()
    dt = (time.perf_counter() - t0) / reps
    k = batch.engine.last_kernel_ms()
    return dt, k


def latency_entries(lps0, labs0):
    """The few-lattice regime (rank 0, N=1): a lone cfg2 lattice and the two book stand-ins, library defaults
    (KA_MODE_AUTO: tile pipeline + chunk-parallel backtrace) next to the one-wavefront form with the serial backtrace."""
    import torch
    from kokoro_align_amd import workloads as W
    from kokoro_align_amd.align import DeviceBatch
    out = {}
    one = DeviceBatch([lps0], [labs0], BEAM, MAX_MOVE)
    one.engine.set_profiling(True)
    res = {}
    for name, mode, bt in (("auto", "auto", "auto"), ("wave_serial", "wave", "serial")):
        one.engine.set_mode(mode)
        one.engine.set_backtrace(bt)
        dt, k = timed_batch(one, 3)
        res[name] = {"ms": dt * 1e3, "frames_per_s": T / dt, "forward_ms": k["forward"], "backtrace_ms": k["backtrace"]}
    out["single_lattice"] = dict(res["auto"], one_wavefront_serial_backtrace=res["wave_serial"])
    books = {}
    for key, (name, shapes) in (("kokoro", W.kokoro_book()), ("meian", W.meian_book())):
        lps, labs = W.device_book(shapes)
        b = DeviceBatch(lps, labs)
        frames = sum(t for t, _ in shapes)
        res = {}
        for tag, mode, bt in (("auto", "auto", "auto"), ("wave_serial", "wave", "serial")):
            b.engine.set_mode(mode)
            b.engine.set_backtrace(bt)
            dt, k = timed_batch(b, 3)
            res[tag] = {"ms": dt * 1e3, "frames_per_s": frames / dt, "forward_ms": k["forward"], "backtrace_ms": k["backtrace"]}
        ends_ok = all(int(p[-1]) == 2 * s for p, (_, s) in zip(b.path, shapes))
        books[key] = dict(res["auto"], workload=name, chapters=len(shapes), frames=frames, longest_chapter=max(t for t, _ in shapes),
                          all_ends_at_trailing_blank=ends_ok, one_wavefront_serial_backtrace=res["wave_serial"])
        del lps, labs, b
        torch.cuda.empty_cache()
    one.engine.set_mode("auto")
    one.engine.set_backtrace("auto")
    out["book"] = books
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "book"])
    ap.add_argument("--lattices", type=int, default=int(os.environ.get("KA_BENCH_LATTICES", "8192")),
                    help="cfg2 workload: lattices per GPU per step")
    ap.add_argument("--mode", default="auto", choices=["auto", "wave", "wave_exact", "workgroup", "tiled"],
                    help="kernel form (DESIGN.md section 4); auto = one wavefront per lattice, checkpointed, at this batch size")
    ap.add_argument("--backtrace", default="auto", choices=["auto", "serial", "parallel"])
    ap.add_argument("--cpu-baseline-seconds", type=float, default=8.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-lattice / book entries")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relay_to_children(args)

    # The CPU baseline starts worker PROCESSES (one lattice per core): it runs first, before this process has touched the
    # GPU - a process that has initialised the GPU must not start programs on this pool.
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    cpu = None
    if world_env == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_baseline_seconds)

    # stdout must carry exactly ONE JSON line: native libraries (RCCL prints a version banner) write to
    # fd 1 directly, so point fd 1 at stderr until the result is printed
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import kokoro_align_amd as ka
    from kokoro_align_amd import workloads as W
    from kokoro_align_amd.align import DeviceBatch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # rehearsal knob (never set by the driver): several ranks on ONE GPU over gloo, to exercise the
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
    stream = torch.cuda.current_stream().cuda_stream
    if args.workload == "cfg2":
        B = args.lattices
        # synthetic inputs, generated in HBM (hash generator == oracle's, so any lattice can be re-checked on CPU)
        # 8192 lattices = 8 wavefronts per SIMD = 215 GB (log-probs + checkpoints); halve on OOM.
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
        seed0 = rank * 1000003            # lattice i of this rank uses seed0 + i (rank 0, i = 0 is the golden cfg2 lattice)
        assert lib.ka_hash_logprobs_batch_f32(lps.data_ptr(), B, T, V, V, T * V, seed0, stream) == 0
        assert lib.ka_hash_labels_batch_i32(labs.data_ptr(), B, S, V, S, seed0, stream) == 0
        frames_per_step = B * T * world
        scaling = "weak"
        workload = (f"cfg2: T={T} x V={V} log-probs, S={S} phonemes (L={2 * S + 1}), beam_size={BEAM}, max_move={MAX_MOVE}; "
                    f"batch of {B} independent lattices per GPU per step")
    else:
        from kokoro_align_amd.sharding import shard_for_rank
        name, shapes = W.meian_book()
        mine = shard_for_rank(shapes, rank, world)
        lps_all, labs_all = W.device_book([shapes[i] for i in mine], seed0=W.BOOK_SEED0)   # (seed by position in the shard: data only)
        batch = DeviceBatch(lps_all, labs_all)
        B = len(mine)
        frames_per_step = sum(t for t, _ in shapes)
        scaling = "strong"
        workload = (f"book: {name}, {len(shapes)} chapter lattices ({frames_per_step} frames, V=39, S=0.14T), split over {world} rank(s) "
                    f"by shard_for_rank; this rank: {B} chapters")
    torch.cuda.synchronize()
    batch.engine.set_profiling(True)
    batch.engine.set_mode(args.mode)
    batch.engine.set_backtrace(args.backtrace)

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

    # parity spot check on the timed outputs
    ok = True
    if rank == 0 and args.workload == "cfg2":
        from tests.golden_util import g3_case
        g3 = g3_case()       # lattice 0 of rank 0 is the golden cfg2 lattice
        ok = bool(np.array_equal(batch.path[0].cpu().numpy(), g3["path"]))
        ends = torch.stack([p[-1] for p in batch.path[: min(B, 64)]]).cpu().numpy()
        ok = ok and bool((ends == 2 * S).all())
        # and, without any reference, on a spread of lattices: the float32 chain of the per-frame scores along the
        # returned path must be the forward pass's best cumulative score bit for bit (a path that is not THE best
        # path of its lattice misses it; this is what caught a store hazard that left every end position right)
        pick = sorted(set(np.linspace(0, B - 1, num=min(B, 256), dtype=np.int64).tolist()))
        chain = np.add.accumulate(torch.stack([batch.best_scores[i] for i in pick]).cpu().numpy(), axis=1, dtype=np.float32)[:, -1]
        ok = ok and bool(np.array_equal(chain.view(np.int32), np.asarray(batch.total, np.float32)[pick].view(np.int32)))
    elif rank == 0:
        for i in range(B):
            chain = np.add.accumulate(batch.best_scores[i].cpu().numpy(), dtype=np.float32)[-1]
            ok = ok and np.float32(chain).view(np.int32) == np.float32(batch.total[i]).view(np.int32)
        ok = bool(ok)

    if rank == 0:
        value = frames_per_step * args.steps / elapsed
        out = {
            "metric": "aligned audio-frames/sec (whole node), 50k x 5k lattice" if args.workload == "cfg2" else "aligned audio-frames/sec (whole node), Meian book",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (hash-generated log-probs and labels, in HBM)",
            "config": {"workload": workload, "lattices_per_gpu": B, "frames_per_step": frames_per_step, "parallelism": f"lattice-sharded x{world}",
                       "kernel_form": args.mode, "backtrace": args.backtrace},
            "kernels_ms": {"prep": float(np.mean(prep_ms)), "forward": float(np.mean(fwd_ms)),
                           "backtrace": float(np.mean(bt_ms)), "gather": float(np.mean(ga_ms))},
            "parity_spot_check": ok,
        }
        if args.workload == "cfg2":
            fwd_b, job_b, wbar, survey_fwd_b, survey_job_b = algorithmic_bytes_per_frame()
            fwd_s = float(np.mean(fwd_ms)) * 1e-3
            checkpointed = args.mode in ("auto", "wave", "tiled")
            if not checkpointed:     # every back-pointer stored: SURVEY.md 8d's bytes are this form's own
                fwd_b, job_b = survey_fwd_b, survey_job_b + 4.0 * V / 2
            achieved = B * T * fwd_b / fwd_s / 1e9
            # HBM bytes of one forward launch cannot be counted from inside this process: they come from the
            # committed rocprofv3 PMC run (tools/prof.sh -> tools/summarize_profile.py) at the same batch size
            traffic = job_traffic = None
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    pt = json.load(f)
                if int(pt.get("lattices", -1)) == B and checkpointed:
                    traffic = pt.get("hbm_bytes_per_launch")
                    job_traffic = pt.get("job_hbm_bytes_per_step")
            except Exception:
                pass
            out["roofline"] = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, same batch size)" if traffic else None,
                "kernel": {"auto": "forward_ck_kernel<4,false>", "wave": "forward_ck_kernel<4,false>", "tiled": "forward_tp_kernel<4>",
                           "wave_exact": "forward_w16_kernel<4,false>", "workgroup": "forward_wg4_kernel<4,false>"}[args.mode],
                "kernel_ms": fwd_s * 1e3, "algorithmic_bytes_per_frame": fwd_b, "mean_band_width": wbar,
                "survey_8d_bytes_per_frame": survey_fwd_b, "achieved_with_survey_8d_bytes": B * T * survey_fwd_b / fwd_s / 1e9,
                "job_frac_with_survey_8d_bytes": frames_per_step / world * survey_job_b / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                "job_traffic_over_survey_8d_bytes": (job_traffic / (B * T * survey_job_b)) if job_traffic else None,
                "valu": valu_bound(B, fwd_s * 1e3) if checkpointed and args.mode != "tiled" else None,
            }
            out["job_bytes_per_frame"] = job_b
        if world == 1 and args.workload == "cfg2" and not args.no_latency:
            lps0, labs0 = lps[0].clone(), labs[0].clone()
            del batch
            lps = labs = None
            torch.cuda.empty_cache()
            out.update(latency_entries(lps0, labs0))
        if cpu is not None:
            out["cpu_baseline"] = cpu
            if args.workload == "cfg2":
                out["speedup_vs_cpu_numpy_1core"] = value / out["cpu_baseline"]["value"]
                out["speedup_vs_c_oracle_1core"] = value / out["cpu_baseline"]["c_oracle_frames_per_s"]
                if "single_lattice" in out:
                    out["single_lattice"]["speedup_vs_cpu_numpy_1core"] = out["single_lattice"]["frames_per_s"] / out["cpu_baseline"]["value"]
                    out["single_lattice"]["speedup_vs_c_oracle_1core"] = out["single_lattice"]["frames_per_s"] / out["cpu_baseline"]["c_oracle_frames_per_s"]
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier() if rehearsal else dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
