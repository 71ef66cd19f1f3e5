#!/usr/bin/env python3
"""Latency regime: a lone cfg2 lattice, the two book stand-ins and the cfg5 stress lattice in the one-wavefront
checkpointed form and in the tiled form, every result checked against the first form's.

    python tools/bench_tiled.py [--cfg5-full T S]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import kokoro_align_amd as ka
from kokoro_align_amd import workloads as W
from kokoro_align_amd.align import DeviceBatch


def timed(lps, labs, mode, beam=1000, reps=3):
    b = DeviceBatch(lps, labs, beam)
    b.engine.set_mode(mode)
    b.engine.set_tile_width(int(os.environ.get("KA_TILE_WIDTH", "0")))
    b.engine.set_profiling(True)
    b.run()
    t0 = time.perf_counter()
    for _ in range(reps):
        b.run()
    dt = (time.perf_counter() - t0) / reps
    k = b.engine.last_kernel_ms()
    b.engine.set_mode("auto")
    b.engine.set_tile_width(0)
    return b, {"ms": dt * 1e3, "forward_ms": k["forward"], "backtrace_ms": k["backtrace"], "gather_ms": k["gather"]}


def compare(name, shapes, V, seed0, modes, beam=1000):
    lps, labs = W.device_book(shapes, V=V, seed0=seed0)
    frames = sum(t for t, _ in shapes)
    out = {"workload": name, "lattices": len(shapes), "frames": frames, "beam_size": beam}
    ref = None
    for mode in modes:
        b, r = timed(lps, labs, mode, beam)
        r["frames_per_s"] = frames / (r["ms"] * 1e-3)
        if ref is None:
            ref = [p.clone() for p in b.path]
            tot = b.total.copy()
        else:
            r["identical_paths"] = all(torch.equal(a, c) for a, c in zip(ref, b.path))
            r["identical_totals"] = bool((tot.view("int32") == b.total.view("int32")).all())
        out[mode] = r
    del lps, labs
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-cfg5", action="store_true")
    ap.add_argument("--full", type=int, nargs=2, default=None, metavar=("T", "S"), help="full-lattice (beam >= 2L) case of this size, tiled form only")
    args = ap.parse_args()
    res = []
    res.append(compare("cfg2 single lattice", [(W.CFG2["T"], W.CFG2["S"])], 64, 0, ["wave", "tiled"]))
    print(json.dumps(res[-1]), flush=True)
    res.append(compare("cfg1 gongitsune", [(W.CFG1["T"], W.CFG1["S"])], 39, 77, ["wave", "tiled"]))
    print(json.dumps(res[-1]), flush=True)
    name, shapes = W.kokoro_book()
    res.append(compare(name, shapes, 39, W.BOOK_SEED0, ["wave", "tiled"]))
    print(json.dumps(res[-1]), flush=True)
    name, shapes = W.meian_book()
    res.append(compare(name, shapes, 39, W.BOOK_SEED0, ["wave", "tiled"]))
    print(json.dumps(res[-1]), flush=True)
    if not args.skip_cfg5:
        res.append(compare("cfg5 long form, band 1000", [(W.CFG5["T"], W.CFG5["S"])], 64, 5, ["wave", "tiled"]))
        print(json.dumps(res[-1]), flush=True)
    if args.full:
        T, S = args.full
        res.append(compare(f"full lattice T={T} S={S} (beam >= 2L)", [(T, S)], 64, 5, ["tiled"], beam=2 * (2 * S + 1)))
        print(json.dumps(res[-1]), flush=True)
