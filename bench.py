#!/usr/bin/env python3
"""bench.py — benchmark of the CTC forced-alignment hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload cfg2|book|corpus] [--lattices B] [--streams G] [--mode M] [--backtrace HOW]

--workload cfg2 (default; BASELINE.json configs[1], the configuration the metric is quoted on): synthetic log-probs
    T=50000 x V=64, S=5000 phonemes (L=10001), beam_size=1000, max_move=4.  One "step" = one pass of the hot path (label
    prep + forward DP + backtrace + outputs) over a batch of B independent lattices of that shape per GPU, each with its
    own hash-generated inputs, already resident in HBM.  N>1: every rank has its own B lattices ("scaling": "weak").
--workload book (BASELINE.json configs[3]): the Meian stand-in (120 chapter lattices, 5.27 M frames, V=39), the chapters
    split over the N ranks by kokoro_align_amd.sharding.shard_for_rank; a step = the whole book once ("scaling":
    "strong").  At N=1 this is configs[2]/[3] on one GPU.
--workload corpus: all 14 enabled datasets of the reference's example.json (64 h of audio, ~460 chapter lattices, 20.1 M
    frames, V=39; kokoro_align_amd.workloads.corpus), chapters split over the N ranks; a step = the whole corpus once
    ("scaling": "strong").
--streams G (cfg2: default 4; kokoro_align_amd.streams.StreamedAligner): the step's B lattices are G launches of B/G on G
    engines / HIP streams / host threads, and the K steps are issued back to back on every stream, as a caller with a queue
    of batches drives the library: the forward pass of one launch runs beside the backtrace of another.  --streams 1 is one
    engine, one launch per step, a host sync after every step (rounds 1-2).
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
CK_FRAMES = 32          # kokoro-align_amd/csrc/ka_types.hpp kCkFrames: frames between stored score rings
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
    needs per frame and wavefront - 20 max (7 candidates per blank/label pair, v_max3 takes three, and the two label cells of a
    group of four share max(l0, b0): ka_device.hpp label_pair_max; 24 up to round 4's first bench line), 8 packed
    adds of the emissions, 3 DPP moves for the cells of the lane below - at the measured issue cost of 4 cycles per
    wave-instruction with 8 wavefronts per SIMD (profiles/r01_ubench_issue_rates.txt), against the kernel's time."""
    min_instr = 20 + 8 + 3
    measured_instr = None
    clock_ghz = None
    f = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(f) as fh:
            pt = json.load(fh)
        measured_instr = pt.get("forward_valu_per_frame")
        clock_ghz = pt.get("forward_clock_ghz")      # GRBM_GUI_ACTIVE / kernel time of the same profiling run
    except Exception:
        pass
    clock_source = "profiles/pmc_traffic.json (GRBM_GUI_ACTIVE / kernel duration)" if clock_ghz else "profiles/r01j_summary.json (round 1)"
    clock_ghz = clock_ghz or 1.9
    waves_per_simd = max(1.0, B / N_SIMD)
    bound_ms = B * T * min_instr * 4.0 / (N_SIMD * clock_ghz * 1e9) * 1e3 if B >= N_SIMD else T * min_instr * 4.0 * waves_per_simd / (clock_ghz * 1e9) * 1e3
    out = {"min_vector_instructions_per_frame": min_instr, "measured_vector_instructions_per_frame": measured_instr or 46.3,
           "cycles_per_wave_instruction": 4.0, "clock_ghz_under_load": clock_ghz, "clock_source": clock_source, "bound_ms": bound_ms,
           "frac": bound_ms / fwd_ms}
    # ... and the MEASURED floor (round 4): the same kernel built without its band handling and finiteness sum (results wrong by
    # design), timed beside the real one on one box - the gap between model and kernel as a measurement
    try:
        with open(os.path.join(ROOT, "profiles", "r04_forward_floor.json")) as fh:
            fl = json.load(fh)
        if int(fl.get("lattices", -1)) == B:
            out["measured_floor"] = {"floor_build_ms": fl["forward_kernel_alone_ms"]["floor"], "real_build_ms_same_box": fl["forward_kernel_alone_ms"]["real"],
                                     "frac": fl["floor_over_real"], "source": "profiles/r04_forward_floor.json (make -C kokoro-align_amd/csrc floor; tools/ab_rc.sh real floor)"}
    except Exception:
        pass
    return out


def _numpy_one(_):
    from oracle import oracle as O
    lp = O.hash_logprobs(T, V, 0)
    labels = O.hash_labels(S, V, 0)
    t0 = time.perf_counter()
    O.ctc_best_path_numpy(lp, labels, BEAM, MAX_MOVE)
    return time.perf_counter() - t0


def cpu_baseline(min_seconds, share=True):
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
    if share:
        import multiprocessing as mp
        from concurrent.futures import ThreadPoolExecutor
        try:
            allowed = len(os.sched_getaffinity(0))
        except AttributeError:
            allowed = os.cpu_count() or 1
        # NOT the whole host: the CPU share that goes with ONE GPU of the node (a 1-GPU box of this pool may run 16
        # processes; the host has 8 GPUs for its 64 cores / 256 hardware threads, i.e. 8 cores = 32 threads per GPU)
        procs = max(1, min(allowed, 16))
        t2 = time.perf_counter()
        with mp.get_context("spawn").Pool(procs) as pool:
            pool.map(_numpy_one, range(procs))
        dt_np = time.perf_counter() - t2
        t3 = time.perf_counter()
        with ThreadPoolExecutor(procs) as ex:      # (ctypes releases the GIL: threads are cores here)
            list(ex.map(lambda _: O.ctc_best_path_c(lp, labels, BEAM, MAX_MOVE), range(procs)))
        dt_cw = time.perf_counter() - t3
        out["per_gpu_share"] = {
            "processes": procs, "cpu_model": cpu_model, "logical_cpus_of_host": os.cpu_count(), "logical_cpus_allowed": allowed,
            "what": "one cfg2 lattice per process on the CPU share of ONE GPU (16 processes: this pool's limit for a 1-GPU box), not the whole host",
            "numpy_port_frames_per_s": procs * T / dt_np, "c_oracle_frames_per_s": procs * T / dt_cw,
            "sample": f"{procs} processes x one cfg2 lattice each, wall {dt_np:.1f} s (incl. process start-up and input generation) / C oracle {dt_cw:.2f} s",
        }
    return out


def relay_to_children(args):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU as a CHILD process (never an exec, and
    before anything here has touched the GPU), relay rank 0's JSON line and the exit code.  The parent never touches the
    GPU, so it is also the one that times the CPU baseline (before the ranks start: the host cores are then idle) and
    merges it into rank 0's line."""
    import subprocess
    cpu = None
    if not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_baseline_seconds)
    # torch.distributed.run picks the rendezvous port itself (--standalone), on the loopback address
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--standalone",
           "--local-addr", "127.0.0.1", os.path.abspath(__file__)] + sys.argv[1:] + ["--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in proc.stdout.decode("utf-8", "replace").splitlines() if ln.startswith("{")]
    if lines:
        line = lines[-1]
        if cpu is not None:
            try:
                out = json.loads(line)
                attach_cpu_baseline(out, cpu, args.workload)
                line = json.dumps(out)
            except ValueError:
                pass
        print(line, flush=True)
    sys.exit(proc.returncode if proc.returncode else (0 if lines else 1))


def attach_cpu_baseline(out, cpu, workload):
    out["cpu_baseline"] = cpu
    if workload == "cfg2":
        out["speedup_vs_cpu_numpy_1core"] = out["value"] / cpu["value"]
        out["speedup_vs_c_oracle_1core"] = out["value"] / cpu["c_oracle_frames_per_s"]
        if "single_lattice" in out:
            out["single_lattice"]["speedup_vs_cpu_numpy_1core"] = out["single_lattice"]["frames_per_s"] / cpu["value"]
            out["single_lattice"]["speedup_vs_c_oracle_1core"] = out["single_lattice"]["frames_per_s"] / cpu["c_oracle_frames_per_s"]


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
    import numpy as np
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
    # BASELINE.json configs[0] (one recording of the reference's CPU case: 81 140 frames x 39 classes, S = 2000) and configs[4]
    # (the long-form stress lattice T = 500 000, S = 50 000: with the reference's band of 1000 and as the whole lattice,
    # beam_size >= 2L = the "tiled DP"), one lattice each under the library defaults.  Checked on the spot: the path ends at the
    # trailing blank, rises by at most 3 per frame, and the float32 chain of its scores equals the forward pass's total bit for bit.
    for key, cfg, V, seed, beam in (("cfg1", W.CFG1, W.V_MODEL, 77, BEAM), ("cfg5_band", W.CFG5, 64, 5, BEAM),
                                    ("cfg5_whole_lattice", W.CFG5, 64, 5, 2 * (2 * W.CFG5["S"] + 1))):
        Tk, Sk = cfg["T"], cfg["S"]
        lps, labs = W.device_book([(Tk, Sk)], V=V, seed0=seed)
        b = DeviceBatch(lps, labs, beam, MAX_MOVE)
        b.engine.set_profiling(True)
        dt, k = timed_batch(b, 2)
        path = b.path[0].cpu().numpy()
        steps = np.diff(path)
        out[key] = {"ms": dt * 1e3, "frames_per_s": Tk / dt, "forward_ms": k["forward"], "backtrace_ms": k["backtrace"],
                    "workload": f"T={Tk} x V={V}, S={Sk} (L={2 * Sk + 1}), beam_size={beam}", "cells": int(Tk) * int(min(beam, 2 * Sk + 1)),
                    "ends_at_trailing_blank": bool(int(path[-1]) == 2 * Sk), "steps_within_0_3": bool(steps.min() >= 0 and steps.max() <= 3),
                    "score_chain_equals_total": bool(np_chain(b.best_scores[0]) == np_bits(b.total[0]))}
        b.engine.set_profiling(False)
        del lps, labs, b
        torch.cuda.empty_cache()
    return out


def corpus_entry():
    """The unit of work of the reference's `run_example.py` (no --dataset): every enabled dataset of example.json, ~460
    chapter lattices of 20k..94k frames (kokoro_align_amd.workloads.corpus) - the few-hundred-lattice, mixed-length regime
    in which KA_MODE_AUTO has to pick a form.  Timed (a) as ONE launch of all chapters and (b) dataset by dataset like the
    reference's loop, the datasets' launches on 4 streams (StreamedAligner)."""
    import torch
    from kokoro_align_amd import workloads as W
    from kokoro_align_amd.align import DeviceBatch
    from kokoro_align_amd.streams import StreamedAligner
    datasets = W.corpus()
    per_ds, all_lps, all_labs, all_shapes = [], [], [], []
    for k, (name, shapes) in enumerate(datasets):
        lps, labs = W.device_book(shapes, seed0=W.corpus_seed0(k))
        per_ds.append((lps, labs))
        all_lps += lps; all_labs += labs; all_shapes += shapes
    frames = sum(t for t, _ in all_shapes)
    out = {"workload": "example.json, 14 enabled datasets", "datasets": len(datasets), "chapters": len(all_shapes), "frames": frames,
           "hours_of_audio": frames / W.FRAMES_PER_SECOND / 3600.0, "longest_chapter": max(t for t, _ in all_shapes)}
    one = DeviceBatch(all_lps, all_labs)
    one.engine.set_profiling(True)
    for tag, mode, bt in (("auto", "auto", "auto"), ("tiled", "tiled", "parallel"), ("wave_serial", "wave", "serial")):
        one.engine.set_mode(mode)
        one.engine.set_backtrace(bt)
        dt, k = timed_batch(one, 3)
        out["one_launch_" + tag] = {"ms": dt * 1e3, "frames_per_s": frames / dt, "forward_ms": k["forward"], "backtrace_ms": k["backtrace"]}
    one.engine.set_mode("auto")
    one.engine.set_backtrace("auto")
    one.engine.set_profiling(False)
    ends_ok = all(int(p[-1]) == 2 * s for p, (_, s) in zip(one.path, all_shapes))
    chain_ok = True
    for i in range(0, len(all_shapes), 7):
        chain = np_chain(one.best_scores[i])
        chain_ok = chain_ok and chain == np_bits(one.total[i])
    out["all_ends_at_trailing_blank"] = bool(ends_ok)
    out["score_chain_equals_total_on_sample"] = bool(chain_ok)
    del one
    sa = StreamedAligner(4)
    batches = sa.bind([DeviceBatch(lps, labs) for lps, labs in per_ds])
    sa.run(batches)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sa.run(batches, repeat=3)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    out["per_dataset_launches_4_streams"] = {"ms": dt * 1e3, "frames_per_s": frames / dt}
    sa.close()
    del batches, per_ds, all_lps, all_labs
    torch.cuda.empty_cache()
    return out


def pipeline_entry():
    """SURVEY.md section 8f end to end on the Kokoro stand-in (64 chapters, 2.77 M frames = 8.78 h of audio): MFCC rows on the
    host -> AudioToChar on the device (kokoro_align_amd.model.lstm_logits_device: library GEMMs + the persistent MFMA
    recurrence kernel) -> ONE log-softmax launch -> the chapters' DP as one launch.  Random-init network of the reference's
    architecture (train.py:54-65), random MFCC-shaped input cut into segments of 200..1200 frames as the reference's
    splitter produces them; stage times from HIP events, second pass (allocator and library heuristics warm)."""
    import numpy as np
    import torch
    import kokoro_align_amd as ka
    from kokoro_align_amd import workloads as W
    from kokoro_align_amd.align import DeviceBatch
    from kokoro_align_amd.model import load_model, lstm_logits_device
    dev = torch.device("cuda", torch.cuda.current_device())
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    model = load_model(None, device=dev)
    V = model.dense.out_features
    name, shapes = W.kokoro_book()
    ends, row = [], 0
    for T, _ in shapes:
        left = T
        while left > 0:
            n = int(min(left, rng.integers(200, 1200)))
            row += n
            left -= n
            ends.append(row)
    total = row
    mfcc = rng.standard_normal((total, 40), dtype=np.float32)
    labels = [torch.from_numpy(rng.integers(1, V, size=S).astype(np.int32)).to(dev) for _, S in shapes]
    out = {}
    for rep in range(2):
        tm = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        logits = lstm_logits_device(model, mfcc, np.asarray(ends), device=dev, timings=tm)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        lp = ka.log_softmax_device(logits)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        lps, k = [], 0
        for T, _ in shapes:
            lps.append(lp[k:k + T])
            k += T
        b = DeviceBatch(lps, labels)
        b.engine.set_profiling(True)
        b.run()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        km = b.engine.last_kernel_ms()
        b.engine.set_profiling(False)
        ok = all(int(p[-1]) == 2 * S for p, (_, S) in zip(b.path, shapes))
        net = sum(tm.values())
        out = {"workload": name + ", 64 chapters as %d MFCC segments" % len(ends), "frames": total,
               "network_ms": (t1 - t0) * 1e3, "network_stages_ms": {k_: round(v, 3) for k_, v in tm.items()}, "network_stages_sum_ms": net,
               "log_softmax_ms": (t2 - t1) * 1e3, "ctc_best_path_ms": (t3 - t2) * 1e3,
               "ctc_kernels_ms": {"forward": km["forward"], "backtrace": km["backtrace"]},
               "end_to_end_ms": (t3 - t0) * 1e3, "frames_per_s_end_to_end": total / (t3 - t0),
               "all_paths_end_at_the_trailing_blank": bool(ok)}
        del b, lps, lp, logits
    # the recurrence kernel against the f32 MFMA peak (MI355X_MICROARCH.md: 157.3 TFLOP/s); layer 0 carries its K = 40 input
    # projection inside the step (ka_lstm_layer0_f32)
    rec = sum(v for k_, v in out["network_stages_ms"].items() if "recurrence" in k_)
    fused = any(k_.startswith("projection+recurrence") for k_ in out["network_stages_ms"])
    flops = 2.0 * total * 2 * 2 * 128 * 512 + (2.0 * total * 40 * 1024 if fused else 0.0)   # 2 flop per MAC x frames x directions x layers x (128 x 512)
    out["recurrence"] = {"ms_both_layers": rec, "tflops": flops / (rec * 1e-3) / 1e12, "f32_mfma_peak_tflops": 157.3,
                         "frac_of_peak": flops / (rec * 1e-3) / 1e12 / 157.3, "layer0_projection_inside": fused,
                         "note": "chain-bound: the longest segment (1199 steps) x 5.4 us per step (6.6 with the layer-0 projection); 16 sequences per workgroup"}
    torch.cuda.empty_cache()
    return out


def np_bits(x):
    import numpy as np
    return int(np.float32(x).view(np.int32))


def np_chain(scores):
    """float32 running sum of the per-frame scores along a path (device tensor) -> bits of the last partial sum"""
    import numpy as np
    return int(np.add.accumulate(scores.cpu().numpy(), dtype=np.float32)[-1].view(np.int32))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "book", "corpus"])
    ap.add_argument("--lattices", type=int, default=int(os.environ.get("KA_BENCH_LATTICES", "8192")),
                    help="cfg2 workload: lattices per GPU per step")
    ap.add_argument("--streams", type=int, default=None,
                    help="launches in flight per GPU (cfg2: default 4 sub-batches on 4 engines / streams; book, corpus: 1)")
    ap.add_argument("--mode", default="auto", choices=["auto", "wave", "wave_exact", "tiled"],
                    help="kernel form (DESIGN.md section 4); auto = one wavefront per lattice, checkpointed, at this batch size")
    ap.add_argument("--backtrace", default="auto", choices=["auto", "serial", "parallel"])
    ap.add_argument("--cpu-baseline-seconds", type=float, default=8.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-lattice / book / corpus entries")
    args = ap.parse_args()

    world_env = int(os.environ.get("WORLD_SIZE", "0"))
    if world_env:
        args.gpus = world_env            # under a launcher the launcher decides
    elif args.gpus > 1:
        relay_to_children(args)

    # The CPU baseline starts worker PROCESSES (one lattice per core): it runs first, before this process has touched the
    # GPU - a process that has initialised the GPU must not start programs on this pool.  Under a launcher rank 0 does it
    # (the other ranks wait for it in the process group's first barrier); started by relay_to_children the parent has.
    rank_env = int(os.environ.get("RANK", "0"))
    cpu = None
    if rank_env == 0 and not args.no_cpu_baseline:
        # (under a launcher with several ranks only the one-core figures: the others are waiting for rank 0 in the process
        #  group's rendezvous, and the 16-process pool belongs to the parent of relay_to_children, which has no ranks waiting)
        cpu = cpu_baseline(args.cpu_baseline_seconds, share=world_env <= 1)

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
    from kokoro_align_amd.streams import StreamedAligner, split_device_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # rehearsal knob (never set by the driver): several ranks on ONE GPU over gloo, to exercise the
    # multi-rank control flow on a single-GPU box
    rehearsal = os.environ.get("KA_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or os.environ.get("KA_FORCE_DIST") == "1":   # KA_FORCE_DIST: exercise the RCCL path with one rank
        import datetime
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=20))
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=20))      # RCCL over xGMI
        # start-up collective of the pipeline: broadcast of the acoustic-model weights (2.3 MB)
        from kokoro_align_amd.sharding import broadcast_model_weights
        broadcast_model_weights(torch.device("cpu") if rehearsal else dev)

    lib = ka.load_library()
    stream = torch.cuda.current_stream().cuda_stream
    G = args.streams if args.streams else (4 if args.workload == "cfg2" else 1)
    lps = labs = None
    if args.workload == "cfg2":
        B = args.lattices
        if rehearsal:
            B = min(B, 1024 // world)
        # synthetic inputs, generated in HBM (hash generator == oracle's, so any lattice can be re-checked on CPU)
        # 8192 lattices = 8 wavefronts per SIMD = 215 GB (log-probs + checkpoints); halve on OOM.
        while True:
            try:
                lps = torch.empty((B, T, V), dtype=torch.float32, device=dev)
                labs = torch.empty((B, S), dtype=torch.int32, device=dev)
                G = max(1, min(G, B))
                batches, _ = split_device_batch([lps[i] for i in range(B)], [labs[i] for i in range(B)], G, BEAM, MAX_MOVE)
                aligner = StreamedAligner(G, dev_index, args.mode, args.backtrace, profiling=True)
                aligner.bind(batches)
                break
            except (RuntimeError, MemoryError) as exc:   # torch OOM or KA_ERR_NOMEM
                if B <= 64:
                    raise
                print(f"[bench] {B} lattices do not fit ({type(exc).__name__}); retrying with {B // 2}", file=sys.stderr)
                lps = labs = batches = None
                torch.cuda.empty_cache()
                B //= 2
        seed0 = rank * 1000003            # lattice i of this rank uses seed0 + i (rank 0, i = 0 is the golden cfg2 lattice)
        assert lib.ka_hash_logprobs_batch_f32(lps.data_ptr(), B, T, V, V, T * V, seed0, stream) == 0
        assert lib.ka_hash_labels_batch_i32(labs.data_ptr(), B, S, V, S, seed0, stream) == 0
        frames_per_step = B * T * world
        scaling = "weak"
        workload = (f"cfg2: T={T} x V={V} log-probs, S={S} phonemes (L={2 * S + 1}), beam_size={BEAM}, max_move={MAX_MOVE}; "
                    f"batch of {B} independent lattices per GPU per step" + (f", as {G} launches of {B // G} on {G} streams" if G > 1 else ""))
    else:
        from kokoro_align_amd.sharding import shard_for_rank
        if args.workload == "book":
            name, shapes = W.meian_book()
            seeds = [W.BOOK_SEED0 + i for i in range(len(shapes))]
            what = f"book: {name}"
        else:
            shapes, seeds = [], []
            for k, (_, sh) in enumerate(W.corpus()):
                shapes += sh
                seeds += [W.corpus_seed0(k) + i for i in range(len(sh))]
            what = "corpus: the 14 enabled datasets of example.json"
        mine = shard_for_rank(shapes, rank, world)
        lps_all, labs_all = [], []
        for i in mine:                       # (same inputs whatever the sharding: lattice i is seeded by its index)
            lp1, lab1 = W.device_book([shapes[i]], seed0=seeds[i])
            lps_all += lp1; labs_all += lab1
        B = len(mine)
        G = max(1, min(G, B))
        batches, _ = split_device_batch(lps_all, labs_all, G)
        aligner = StreamedAligner(G, dev_index, args.mode, args.backtrace, profiling=True)
        aligner.bind(batches)
        frames_per_step = sum(t for t, _ in shapes)
        scaling = "strong"
        workload = (f"{what}, {len(shapes)} chapter lattices ({frames_per_step} frames, V=39, S=0.14T), split over {world} rank(s) "
                    f"by shard_for_rank; this rank: {B} chapters" + (f" as {G} launches on {G} streams" if G > 1 else ""))
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier() if rehearsal else dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    # warm-up: W untimed steps; with several streams worker g starts g/G of a step late, so that from the first timed
    # launch on one stream's forward pass runs beside another's backtrace (they also drift apart by themselves)
    if args.warmup > 0:
        aligner.run(batches, repeat=args.warmup, stagger_s=(0.07 / G if G > 1 and args.workload == "cfg2" else 0.0))
    barrier()
    t0 = time.perf_counter()
    aligner.run(batches, repeat=args.steps)      # per launch: enqueue + stream sync + per-lattice status check
    barrier()
    elapsed = time.perf_counter() - t0
    launch_ms = {k: float(np.mean([ms[k] for _, _, ms in aligner.kernel_ms])) for k in ("prep", "forward", "backtrace", "gather")}
    n_launches = len(aligner.kernel_ms)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # parity spot check on the timed outputs
    path = [p for b in batches for p in b.path]
    best_scores = [x for b in batches for x in b.best_scores]
    total = np.concatenate([np.asarray(b.total, np.float32) for b in batches])
    status_ok = all(not (b.status != 0).any() for b in batches)
    ok = status_ok
    if rank == 0 and args.workload == "cfg2":
        from tests.golden_util import g3_case
        g3 = g3_case()       # lattice 0 of rank 0 is the golden cfg2 lattice
        ok = ok and bool(np.array_equal(path[0].cpu().numpy(), g3["path"]))
        ends = torch.stack([p[-1] for p in path[: min(B, 64)]]).cpu().numpy()
        ok = ok and bool((ends == 2 * S).all())
        # and, without any reference, on a spread of lattices: the float32 chain of the per-frame scores along the
        # returned path must be the forward pass's best cumulative score bit for bit (a path that is not THE best
        # path of its lattice misses it; this is what caught a store hazard that left every end position right)
        pick = sorted(set(np.linspace(0, B - 1, num=min(B, 256), dtype=np.int64).tolist()))
        chain = np.add.accumulate(torch.stack([best_scores[i] for i in pick]).cpu().numpy(), axis=1, dtype=np.float32)[:, -1]
        ok = ok and bool(np.array_equal(chain.view(np.int32), total[pick].view(np.int32)))
    elif rank == 0:
        for i in range(0, B, max(1, B // 128)):
            ok = ok and np_chain(best_scores[i]) == np_bits(total[i])
        ok = bool(ok)

    if rank == 0:
        value = frames_per_step * args.steps / elapsed
        out = {
            "metric": {"cfg2": "aligned audio-frames/sec (whole node), 50k x 5k lattice", "book": "aligned audio-frames/sec (whole node), Meian book",
                       "corpus": "aligned audio-frames/sec (whole node), example.json corpus"}[args.workload],
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (hash-generated log-probs and labels, in HBM)",
            "config": {"workload": workload, "lattices_per_gpu": B, "frames_per_step": frames_per_step, "parallelism": f"lattice-sharded x{world}",
                       "kernel_form": args.mode, "backtrace": args.backtrace, "streams_per_gpu": G,
                       "step": (f"{G} launches of {B // G} lattices on {G} engines / HIP streams, steps issued back to back per stream (no host barrier "
                                "between steps): a launch's forward pass overlaps another launch's backtrace") if G > 1 else
                               "one launch per step, host sync after every step"},
            "kernels_ms": dict(launch_ms, per="launch, HIP events on the launch's own stream inside the timed region"
                                              + (f" ({G} launches in flight: a launch shares the GPU)" if G > 1 else ""), launches=n_launches),
            "parity_spot_check": ok,
        }
        if args.workload == "cfg2":
            fwd_b, job_b, wbar, survey_fwd_b, survey_job_b = algorithmic_bytes_per_frame()
            checkpointed = args.mode in ("auto", "wave", "tiled")
            if not checkpointed:     # every back-pointer stored: SURVEY.md 8d's bytes are this form's own
                fwd_b, job_b = survey_fwd_b, survey_job_b + 4.0 * V / 2
            # The dominant kernel on its own: with G > 1 launches in flight a launch's duration is not the kernel's own
            # speed (it shares the SIMDs with G-1 others), so its roofline is measured live in a SERIAL pass right after
            # the timed region - ONE launch over all B lattices, HIP events on its stream, 3 steps - the same grid size
            # as the rocprofv3 summary under profiles/.  The timed region's own per-launch numbers are kept beside it.
            timed_fwd_s = launch_ms["forward"] * 1e-3
            serial = None
            if G > 1:
                aligner.close()
                torch.cuda.empty_cache()
                whole = DeviceBatch([lps[i] for i in range(B)], [labs[i] for i in range(B)], BEAM, MAX_MOVE)
                whole.engine.set_mode(args.mode)
                whole.engine.set_backtrace(args.backtrace)
                whole.engine.set_profiling(True)
                whole.run()
                ts, ks = time.perf_counter(), []
                for _ in range(3):
                    whole.run()
                    ks.append(whole.engine.last_kernel_ms())
                serial = {"ms_per_step": (time.perf_counter() - ts) / 3 * 1e3, "steps": 3,
                          "kernels_ms": {k: float(np.mean([x[k] for x in ks])) for k in ("prep", "forward", "backtrace", "gather")}}
                serial["frames_per_s"] = B * T / (serial["ms_per_step"] * 1e-3)
                whole.engine.set_profiling(False)
                whole.engine.set_mode("auto")
                whole.engine.set_backtrace("auto")
                del whole
                fwd_s = serial["kernels_ms"]["forward"] * 1e-3
            else:
                fwd_s = timed_fwd_s
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
            step_s = elapsed / args.steps
            out["roofline"] = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, same batch size)" if traffic else None,
                "kernel": {"auto": "forward_ck_kernel<4,false>", "wave": "forward_ck_kernel<4,false>", "tiled": "forward_ts_kernel<4> (128-position tiles) or forward_tp2_kernel<4> (256) by the number of tiles alive",
                           "wave_exact": "forward_w16_kernel<4,false>"}[args.mode],
                "kernel_ms": fwd_s * 1e3, "launch_lattices": B,
                "measured": ("serial pass after the timed region: one launch over all lattices, alone on the GPU (HIP events on its stream)" if G > 1
                             else "HIP events on the launch stream inside the timed region"),
                "algorithmic_bytes_per_frame": fwd_b, "mean_band_width": wbar,
                "survey_8d_bytes_per_frame": survey_fwd_b, "achieved_with_survey_8d_bytes": B * T * survey_fwd_b / fwd_s / 1e9,
                "timed_region": {"launch_lattices": B // G, "launches_in_flight": G, "kernel_ms": timed_fwd_s * 1e3,
                                 "achieved_per_launch": (B // G) * T * fwd_b / timed_fwd_s / 1e9,
                                 "job_algorithmic_GBps": B * T * job_b / step_s / 1e9, "job_frac": B * T * job_b / step_s / 1e9 / HBM_PEAK_GBS},
                "job_frac_with_survey_8d_bytes": frames_per_step / world * survey_job_b / step_s / 1e9 / HBM_PEAK_GBS,
                "job_traffic_over_survey_8d_bytes": (job_traffic / (B * T * survey_job_b)) if job_traffic else None,
                "valu": valu_bound(B, fwd_s * 1e3) if checkpointed and args.mode != "tiled" else None,
            }
            out["job_bytes_per_frame"] = job_b
            if serial is not None:
                out["one_launch_per_step"] = serial
        if world == 1 and args.workload == "cfg2" and not args.no_latency:
            lps0, labs0 = lps[0].clone(), labs[0].clone()
            aligner.close()
            del batches, path, best_scores
            lps = labs = None
            torch.cuda.empty_cache()
            out.update(latency_entries(lps0, labs0))
            out["corpus"] = corpus_entry()
            out["pipeline"] = pipeline_entry()
        if cpu is not None:
            attach_cpu_baseline(out, cpu, args.workload)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier() if rehearsal else dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
