#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE config 2, on N MI355X of one node.

  metric    directed edge-message updates / s = 2*E*batch*iters / t  
  workload  HQC-128 graph (N=17669, W=50, R=4000, H=[Hin|I], E=204000), batch 4096
            codewords per GPU, 50 fixed iterations, fp32 min-sum, alpha=1
  step      one decode_batch of the resident batch (inputs already in HBM, device I/O)
  N > 1     trials are independent: each rank decodes its own 4096 trials (seed =
            base + global trial index), no data-path collective; one RCCL all_gather
            of the success flags after the timed region  -> "scaling": "weak"

Also reported on the same JSON line:
  roofline      dominant kernel (check-node update), algorithmic bytes per launch (8 B per
                edge per codeword of the cache-resident tile group it sweeps) / HIP-event
                launch duration vs the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle's f32 restatement (oracle/, a "port": the reference's own
                decoder binaries cannot run here) on a bounded sample, host cores stated
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (hqc set, first-row key, method, max_iter)
    "hqc128_minsum": ("hqc128", "N17669_W50_s0", "min_sum", 50),
    "hqc192_minsum": ("hqc192", "N35851_W50_s0", "min_sum", 50),
    "hqc256_tanh": ("hqc256", "N57637_W50_s0", "product_sum", 50),
    "hqc128_tanh": ("hqc128", "N17669_W50_s0", "product_sum", 50),
    # BASELINE config 5: Monte-Carlo sweep, trials generated/decoded/compared on the device,
    # product_sum, early exit, max_iter 100 (hqc.py:696); --trials = whole-job trial count
    "hqc128_mc": ("hqc128", "N17669_W50_s0", "product_sum", 100),
    # BASELINE config 4: q-ary (Q = 3) min-sum, the reference's DecoderN450R150V3C7B1 on its doctest
    # H (150 x 450, decode.py:192-209), 5 iterations, --batch codewords (1024) per call, host pmf
    # arrays in and symbols out as the PyO3 class takes them.  ALU-bound: no HBM roofline claim.
    "qary_config4": (None, None, "qary_min_sum", 5),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--workload", default="hqc128_minsum", choices=sorted(WORKLOADS))
    ap.add_argument("--eps", type=float, default=0.05)
    ap.add_argument("--tile-group", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--trials", type=int, default=65536, help="hqc128_mc: total trials over all ranks")
    ap.add_argument("--mc-batch", type=int, default=131072, help="hqc128_mc: trials per device call (the stragglers "
                    "of one call share its compact passes, so larger is better: 356k / 365k / 371k / 367k trials/s at "
                    "32768 / 131072 / 262144 / 524288, profiles/r02/mc_batch_sweep.log)")
    ap.add_argument("--pmc", choices=["live", "file", "off"], default="live", help="roofline.traffic: 'live' = two short "
                    "rocprofv3 --pmc child runs (FETCH_SIZE, WRITE_SIZE) of this very workload before the timed run "
                    "(N=1 only; falls back to 'file'), 'file' = the committed profiles/*_pmc_traffic_*.json, 'off' = null")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # the short run the PMC passes profile
    ap.add_argument("--no-hbm-streaming", action="store_true", help="skip the extra pass with tile groups far beyond "
                    "the Infinity Cache (roofline.hbm_streaming_GBps)")
    ap.add_argument("--rendezvous-only", action="store_true", help="launch, rendezvous, one all_gather of the rank "
                    "ids, print {rccl_ranks}; no decode (checks the N>1 launch path; works over gloo without a GPU)")
    ap.add_argument("--parity-rows", type=int, default=64, help="codewords of the timed output checked against the "
                    "CPU oracle after the timed region (0 = skip); a mismatch makes the run exit non-zero")
    args = ap.parse_args()
    self_launch(args)
    # PMC passes first: children of a process that has not touched the GPU yet
    live_traffic = None
    if args.gpus == 1 and args.pmc == "live" and not args.pmc_child and WORKLOADS[args.workload][0] and args.workload != "hqc128_mc":
        live_traffic = pmc_live(args.workload)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.rendezvous_only:
        return rendezvous_only(args, torch, dist, rank, world, local)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    # Rehearsal knobs (one-GPU box only): several ranks on one device over gloo.  The real
    # multi-GPU run uses one device per rank and backend nccl (= RCCL).
    if os.environ.get("BENCH_FORCE_DEVICE") is not None:
        local = int(os.environ["BENCH_FORCE_DEVICE"])
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)

    S = importlib.import_module("sca-ldpc_amd")
    bp = importlib.import_module("sca-ldpc_amd.bp")
    lib = importlib.import_module("sca-ldpc_amd._lib")
    trials = importlib.import_module("sca-ldpc_amd.trials")
    lib.check(lib.load().scaldpc_set_device(local))

    hqc, key, method, iters = WORKLOADS[args.workload]
    if args.workload == "qary_config4":
        return qary_config4(args, S, rank, world, dist, backend, local, iters)
    rows = json.load(open(os.path.join(ROOT, "tests", "golden", "hqc_first_rows.json")))
    H, Hin, _ = S.codes.hqc_bench_graph(hqc, rows[key])
    N, omega = S.codes.HQC_PARAMS[hqc]
    R, E, n = Hin.m, H.nnz, H.n
    batch = args.batch
    probs = trials.hqc_priors(N, R, omega, args.eps)
    if args.workload != "hqc128_mc":
        msg, ys = trials.hqc_trials(Hin, omega, args.eps, batch, base_seed=2, first_index=rank * batch)

    if args.workload == "hqc128_mc":
        return mc_sweep(args, S, bp, lib, trials, H, N, omega, R, E, probs, iters, method, rank, world, local, dist, backend)
    if args.pmc_child:  # what the PMC passes profile: one cache-resident group's worth of launches, nothing else
        iters = 6
    dec = bp.bp_decoder(H, max_iter=iters, bp_method=method, channel_probs=probs)
    if args.tile_group:
        dec.set_tile_group(args.tile_group)
    dev = torch.device("cuda", local)
    d_in = torch.from_numpy(msg).to(dev)
    d_out = torch.empty((batch, n), dtype=torch.uint8, device=dev)
    d_conv = torch.empty(batch, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        dec.decode_batch_device(d_in.data_ptr(), lib.IN_RECEIVED, batch, d_out.data_ptr(), early_exit=False,
                                stream=stream, d_out_conv=d_conv.data_ptr())

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.pmc_child:
        step()
        fence()
        kt = dec.time_kernels(1, stream=stream)
        print(json.dumps({"pmc_child": True, "codewords_per_launch": kt["codewords"], "lanes": kt["lanes"]}), flush=True)
        dec.close()
        return
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # the same kernels streaming from HBM: one tile group = the whole batch (far beyond the 256 MiB
    # Infinity Cache), same launches otherwise -- the other regime next to the cache-resident one
    hbm_stream = None
    if not args.no_hbm_streaming and rank == 0 and world == 1:
        T = (batch + 63) // 64
        dec.set_tile_group(T)
        step()
        fence()
        t1 = time.perf_counter()
        for _ in range(2):
            step()
        fence()
        ht = (time.perf_counter() - t1) / 2
        hbm_stream = {"GBps": 16.0 * E * iters * batch / ht / 1e9, "ms_per_step": ht * 1e3,
                      "group_MB": 4.0 * E * 64 * T / 1e6}
        dec.set_tile_group(args.tile_group)
        step()  # leave the default schedule's state behind for the kernel timing below
        fence()

    # per-kernel launch durations, HIP events on the launch stream
    kt = dec.time_kernels(50, stream=stream)
    ms_check = kt["ms_check"] / max(1, kt["launches_check"])
    ms_var_pass = kt["ms_var"] / max(1, kt["launches_var"])
    swept = kt["codewords"]  # codewords per check launch (tile padded)
    lanes = kt["lanes"]  # 2: timed in the decode's own two-stream launch pattern (one event per launch)
    iso = None
    if lanes == 2:  # the same kernels alone on the chip, one series after the other over the whole tile group
        dec.configure(split=1)
        iso = dec.time_kernels(50, stream=stream)
        dec.configure(split=2)

    # measured device copy ceiling on this very GPU (SURVEY.md 8d asks for it next to the
    # datasheet peak): 1 GiB float copy, read + write bytes / time
    src = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    for _ in range(2):
        dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    copy_gbs = 2.0 * src.numel() * 4 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst

    # success statistics + the one end-of-run collective
    shard = importlib.import_module("sca-ldpc_amd.shard")
    out_host = d_out.cpu().numpy()
    ok = trials.success(out_host, ys, N).astype(np.uint8)
    ok_all = shard.gather_results(ok, batch * world, rank, world, device=dev if backend == "nccl" else None)  # RCCL all_gather
    ranks_seen = shard.gather_results(np.array([rank], dtype=np.int32), world, rank, world, device=dev if backend == "nccl" else None)
    succ = float(ok_all.mean())
    conv = float(d_conv.float().mean().item())

    rc = 0
    if rank == 0:
        total_cw = batch * world * args.steps
        updates = 2.0 * E * iters * total_cw
        value = updates / dt
        check_gbs = 8.0 * E * swept / (ms_check * 1e-3) / 1e9  # 4 B read + 4 B written per edge per codeword
        var_gbs = 8.0 * E * kt["codewords_var"] / (ms_var_pass * 1e-3) / 1e9
        cname = "k_check_minsum_x" if method == "min_sum" else "k_check_tanh"
        out = {
            "metric": "edge_message_updates_per_s",
            "value": value,
            "unit": "directed edge-message updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{hqc} H=[Hin|I] N={N} W=50 R={R} E={E}, batch {batch}/GPU, {iters} fixed iters, {method}",
                "batch_per_gpu": batch,
                "iters": iters,
                "eps": args.eps,
                "tile_group": args.tile_group,
            },
            "rccl_ranks": int(len(set(ranks_seen.tolist()))),  # ranks the end-of-run all_gather saw
            "codewords_per_s": total_cw / dt,
            "whole_job_algorithmic_GBps": 16.0 * E * iters * total_cw / dt / 1e9,
            "decode_success_rate": succ,
            "converged_rate": conv,
            "kernel_ms": {"check_per_launch": ms_check, "var_pass": ms_var_pass},
            "hbm_copy_ceiling_GBps": copy_gbs,  # measured: 1 GiB device copy, read+write bytes/s
        }
        # The kernel with the larger share of GPU time is the dominant one (k_var on every workload so
        # far).  Under the two-lane schedule no kernel runs alone: each stream alternates check and
        # variable launches over its half of the tile group, one kernel out of phase with the other
        # stream, so `achieved` is the CHIP's algorithmic rate over a (check + variable) launch pair:
        # lanes x (bytes of a check launch + bytes of a variable launch) / (t_check + t_var)
        # (DESIGN.md section 5).  `per_launch` = each kernel's in-situ launch duration (what rocprofv3
        # reports for it), `dominant` = the larger one, `isolated` = the same kernels alone on the chip.
        # The messages of a tile group are served by the 256 MiB Infinity Cache by design (the group is
        # sized for it), which is why `achieved` can exceed what HBM streaming sustains
        # (`hbm_copy_ceiling_GBps`, `hbm_streaming_GBps`); `peak` stays the HBM datasheet figure the
        # contract names.
        dom_is_var = ms_var_pass >= ms_check
        dname = "k_var" if dom_is_var else cname
        achieved = lanes * (8.0 * E * swept + 8.0 * E * kt["codewords_var"]) / ((ms_check + ms_var_pass) * 1e-3) / 1e9
        traffic, traffic_src = None, None
        if live_traffic and live_traffic.get("codewords_per_launch") == swept:
            tk = live_traffic["kernels"].get("k_var" if dom_is_var else ("k_check_minsum" if method == "min_sum" else "k_check_tanh"))
            if tk:
                traffic, traffic_src = tk["traffic_bytes"], "live rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run"
        if traffic is None and args.pmc != "off":
            traffic = pmc_traffic(args.workload, batch, swept, "k_var" if dom_is_var else ("k_check_minsum" if method == "min_sum" else "k_check_tanh"))
            traffic_src = "profiles/ (committed PMC passes of this geometry)" if traffic is not None else None
        out["roofline"] = {
            "bound": "infinity-cache",  # what serves the bytes; the contract's class for this path is "hbm" (no MFMA)
            "bound_class": "hbm",
            "kernel": dname,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": 8.0 * E * (kt["codewords_var"] if dom_is_var else swept),
            "lanes": lanes,
            "dominant": {"name": dname, "us": (ms_var_pass if dom_is_var else ms_check) * 1e3,
                         "share_of_pair_time": max(ms_var_pass, ms_check) / (ms_check + ms_var_pass),
                         "algorithmic_GBps": var_gbs if dom_is_var else check_gbs,
                         # alone on the chip (one lane: its own rate; two lanes: filled in from `isolated` below)
                         "frac": (var_gbs if dom_is_var else check_gbs) / HBM_PEAK_GBS if lanes == 1 else None},
            "per_launch": {
                cname: {"codewords": swept, "us": ms_check * 1e3, "algorithmic_GBps": check_gbs},
                "k_var": {"codewords": kt["codewords_var"], "us": ms_var_pass * 1e3, "algorithmic_GBps": var_gbs},
            },
            "hbm_streaming_GBps": hbm_stream["GBps"] if hbm_stream else None,
            "hbm_streaming": hbm_stream,
        }
        if live_traffic:
            out["roofline"]["traffic_all_kernels"] = live_traffic["kernels"]
        if iso:
            ic = iso["ms_check"] / max(1, iso["launches_check"])
            iv = iso["ms_var"] / max(1, iso["launches_var"])
            ig = {cname: 8.0 * E * iso["codewords"] / (ic * 1e-3) / 1e9, "k_var": 8.0 * E * iso["codewords_var"] / (iv * 1e-3) / 1e9}
            out["roofline"]["isolated"] = {
                cname: {"codewords": iso["codewords"], "us": ic * 1e3, "algorithmic_GBps": ig[cname]},
                "k_var": {"codewords": iso["codewords_var"], "us": iv * 1e3, "algorithmic_GBps": ig["k_var"]},
            }
            out["roofline"]["dominant"]["frac"] = ig[dname] / HBM_PEAK_GBS  # the dominant kernel alone on the chip
        if args.parity_rows > 0:
            out.update(parity_check(H, probs, msg, out_host, iters, method, min(args.parity_rows, batch)))
            rc = 0 if out["parity_ok"] else 3
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(H, probs, msg, iters, method, E, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    dec.close()
    if world > 1:
        dist.destroy_process_group()
    if rc:
        raise SystemExit(rc)


def parity_check(H, probs, msg, out_host, iters, method, rows):
    """After the timed region: the first `rows` codewords of the timed output against the CPU oracle
    (f32, same operation order; test infrastructure, never part of what is timed).  min-sum: every bit
    equal.  tanh rule: bits equal wherever the oracle's posterior is outside the fp32 tolerance of
    tests/helpers.compare (|L| > 2e-4 + 2e-4 |L|)."""
    from oracle import pyoracle

    om = {"min_sum": "min_sum", "product_sum": "tanh_complement"}[method]
    threads = max(1, min(os.cpu_count() or 1, pyoracle.max_threads(), rows))
    ref = pyoracle.bp_decode_batch(H, probs, msg[:rows], 1, iters, om, dtype="f32", threads=threads, early_exit=False)
    diff = out_host[:rows] != ref["bits"]
    if method != "min_sum":
        with np.errstate(invalid="ignore"):
            diff &= np.abs(ref["llr"]) > 2e-4
    return {"parity_checked": int(rows), "parity_mismatched_bits": int(diff.sum()), "parity_ok": bool(not diff.any()),
            "parity_against": f"oracle f32 {om}, {iters} fixed iterations, first {rows} codewords of the timed output"}


def pmc_live(workload):
    """roofline.traffic measured in THIS run: two rocprofv3 passes (FETCH_SIZE and WRITE_SIZE do not fit
    one) of `bench.py --pmc-child` -- the same workload, one 256-codeword slice (the launch geometry of
    the timed run: a cache-resident tile group per lane), 6 iterations -- started as child processes
    BEFORE this process touches the GPU.  Units and gfx950 corrections as MI355X_MICROARCH.md's HBM
    section prescribes (KiB; FETCH_SIZE x2: 128-byte read requests are tallied at 64 B; WRITE_SIZE x1;
    re-verified on profiles/microbench/rmw_stream).  These are L2 <-> fabric bytes: Infinity-Cache hits
    are counted, so they bound HBM bytes from above.  Returns None if anything goes wrong."""
    import csv
    import re
    import shutil
    import tempfile
    from collections import defaultdict

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe) or "rocprof" in os.environ.get("LD_PRELOAD", "") or os.environ.get("ROCPROFILER_REGISTER_FORCE_LOAD"):
        return None  # no profiler, or this process is itself being profiled
    env = dict(os.environ, TMPDIR="/tmp")
    per = {}
    geom = None
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        for counter, scale in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", os.path.join(td, counter), "-o", "p",
                   "--", sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", workload, "--batch", "256",
                   "--no-cpu-baseline", "--pmc", "off", "--parity-rows", "0"]
            try:
                r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=240)  # (a pass takes 10-20 s)
            except Exception:
                return None
            if r.returncode != 0:
                return None
            for ln in r.stdout.splitlines():
                if ln.startswith("{") and "pmc_child" in ln:
                    geom = json.loads(ln)
            path = None
            for root, _, files in os.walk(os.path.join(td, counter)):
                for f in files:
                    if f.endswith("counter_collection.csv"):
                        path = os.path.join(root, f)
            if not path:
                return None
            tot, cnt = defaultdict(float), defaultdict(int)
            with open(path) as fh:
                for row in csv.DictReader(fh):
                    if row["Counter_Name"] != counter:
                        continue
                    name = re.sub(r"^void ", "", row["Kernel_Name"].replace("(anonymous namespace)::", "")).split("(")[0]
                    if not name.startswith(("k_var", "k_check")) or name.endswith("true>"):
                        continue  # (FIRST = true: the first iteration reads the priors, not the messages)
                    base = name.split("<")[0]
                    key = (base[:-2] if base.endswith("_x") else base, row["Grid_Size"])
                    tot[key] += float(row["Counter_Value"]) * scale
                    cnt[key] += 1
            best = {}
            for (name, grid), n in cnt.items():  # the most common grid = the schedule's own launches
                if name not in best or n > best[name][1]:
                    best[name] = (tot[(name, grid)] / n, n)
            for name, (b, n) in best.items():
                per.setdefault(name, {})[counter] = b
                per[name]["dispatches"] = n
    if not geom or not per:
        return None
    kernels = {k: {"fetch_bytes": v.get("FETCH_SIZE"), "write_bytes": v.get("WRITE_SIZE"),
                   "traffic_bytes": (v.get("FETCH_SIZE") or 0.0) + (v.get("WRITE_SIZE") or 0.0), "dispatches": v["dispatches"]}
               for k, v in per.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v}
    return {"codewords_per_launch": geom["codewords_per_launch"], "kernels": kernels}


def self_launch(args):
    """`python bench.py --gpus N` from a plain shell (N > 1, no RANK in the environment): start the
    N ranks as a CHILD `python -m torch.distributed.run ... bench.py <same arguments>` -- one process
    per GPU over RCCL, rendezvous on 127.0.0.1 -- forward its output and exit with its return code.
    This runs before torch is imported or any HIP call is made: a process that has touched the GPU
    is never replaced or re-executed.  (The reference's only parallelism is a pool of independent
    decode calls, simulate/decode.py:247-262, and a shell loop over independent runs,
    run-parallel-hqc-simulation.sh:10-43: nothing to coordinate but the launch.)"""
    if args.gpus <= 1 or "RANK" in os.environ:
        return
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def rendezvous_only(args, torch, dist, rank, world, local):
    """The N > 1 launch path without the decode: process group up, one all_gather of the rank ids (the
    same collective the end-of-run gather uses), rank 0 prints what it saw."""
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dev = None
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("BENCH_FORCE_DEVICE", local)))
        dev = torch.device("cuda", torch.cuda.current_device())
    if world > 1:
        dist.init_process_group(backend, rank=rank, world_size=world)
    shard = importlib.import_module("sca-ldpc_amd.shard")
    seen = shard.gather_results(np.array([rank], dtype=np.int32), world, rank, world, device=dev)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"rendezvous_only": True, "n_gpus": world, "backend": backend, "rccl_ranks": int(len(set(seen.tolist())))}),
              flush=True)


def qary_config4(args, S, rank, world, dist, backend, local, iters):
    """BASELINE config 4 through the simulate_rs-shaped class: `--steps` calls of min_sum_batch
    on `--batch` (default here: 1024) codewords per rank; value = symbol-edge message updates / s
    (2 * E * batch * iterations per call), plus ms per call."""
    import torch

    qary = importlib.import_module("sca-ldpc_amd.qary")
    gens = json.load(open(os.path.join(ROOT, "tests", "golden", "generators.json")))
    g = S.TannerGraph.from_coo(gens["regular_identity_300_150_3_6_s1"])
    batch = 1024 if args.batch == 4096 else args.batch
    rng = np.random.RandomState(7 + rank)
    p = 1 / 3
    good, bad = np.array([p, 1.75 * p, 0.25 * p]), np.array([p, 0.25 * p, 1.75 * p])  # decode.py:232-237
    mask = rng.rand(batch, g.n) < 0.005
    pmf = np.where(mask[:, :, None], bad, good).astype(np.float32)
    dec = qary.decoder_class("DecoderN450R150V3C7B1")(g.to_dense(np.int8), iters)
    for _ in range(max(1, args.warmup)):
        out = dec.min_sum_batch(pmf)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = dec.min_sum_batch(pmf)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=torch.device("cuda", local) if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        print(json.dumps({
            "metric": "edge_message_updates_per_s", "value": 2.0 * g.nnz * batch * iters * args.steps * world / dt,
            "unit": "directed symbol-edge message updates/s", "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup),
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"q-ary min-sum DecoderN450R150V3C7B1 (150x450, E={g.nnz}, Q=3), {iters} iterations, "
                                   f"batch {batch}/GPU, host pmf in / symbols out (PCIe and the host-side probability->LLR "
                                   f"conversion included)"},
            "codewords_per_s": batch * args.steps * world / dt, "all_zero_rate": float((out == 0).all(axis=1).mean()),
        }), flush=True)
    if world > 1:
        dist.destroy_process_group()


def mc_sweep(args, S, bp, lib, trials, H, N, omega, R, E, probs, iters, method, rank, world, local, dist, backend):
    """BASELINE config 5: `--trials` synthetic hqc.decode() trials sharded over the ranks by
    GLOBAL trial index (results independent of the GPU count), generated, decoded (early
    exit, max_iter 100) and compared on the device in sub-batches of `--batch`; one gather
    of the per-trial success flags at the end.  A "step" is one sub-batch."""
    import torch

    shard = importlib.import_module("sca-ldpc_amd.shard")
    dec = bp.bp_decoder(H, max_iter=iters, bp_method=method, channel_probs=probs)
    a, b = shard.trial_range(args.trials, rank, world)
    for _ in range(args.warmup):  # untimed: workspace allocation, code-object load
        dec.mc_hqc_run(min(args.mc_batch, max(1, b - a)), omega, args.eps, seed=1, first_trial=0, early_exit=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    succ, its = [], []
    for s0 in range(a, b, args.mc_batch):
        r = dec.mc_hqc_run(min(args.mc_batch, b - s0), omega, args.eps, seed=2, first_trial=s0, early_exit=True)
        succ.append(r["success"])
        its.append(r["iters"])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    succ = np.concatenate(succ) if succ else np.zeros(0, np.uint8)
    its = np.concatenate(its) if its else np.zeros(0, np.int32)
    dev = torch.device("cuda", local)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    gdev = dev if backend == "nccl" else None
    all_succ = shard.gather_results(succ, args.trials, rank, world, device=gdev)
    all_its = shard.gather_results(its, args.trials, rank, world, device=gdev)
    if rank == 0:
        updates = 2.0 * E * float(all_its.astype(np.int64).sum())
        print(json.dumps({
            "metric": "edge_message_updates_per_s", "value": updates / dt, "unit": "directed edge-message updates/s",
            "n_gpus": world, "steps": int(np.ceil((b - a) / args.mc_batch)), "warmup": args.warmup,
            "ms_per_step": dt / max(1, np.ceil((b - a) / args.mc_batch)) * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"hqc128 Monte-Carlo sweep, {args.trials} trials, eps={args.eps}, {method}, early exit, "
                                   f"max_iter {iters}, sub-batches of {args.mc_batch}, device-side trial generation"},
            "trials_per_s": args.trials / dt, "decode_success_rate": float(all_succ.mean()),
            "mean_iterations": float(all_its.mean()), "wall_s": dt,
            "success_checksum": int(np.flatnonzero(all_succ == 0)[:1000].sum()),
        }), flush=True)
    dec.close()
    if world > 1:
        dist.destroy_process_group()


def pmc_traffic(workload, batch, swept, kernel):
    """Bytes per launch of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE x2,
    WRITE_SIZE x1 -- the gfx950 corrections of MI355X_MICROARCH.md, re-verified on
    profiles/microbench), recorded under profiles/ for exactly this workload geometry.
    Counters cannot be read from inside the process, so this is the committed measurement,
    or None when the run's geometry differs from the profiled one."""
    for path in (os.path.join(ROOT, "profiles", "r02", f"pmc_traffic_{workload}.json"),
                 os.path.join(ROOT, "profiles", f"r01_pmc_traffic_{workload}.json")):
        try:
            d = json.load(open(path))
            if d["batch"] == batch and d["tile_group_codewords"] == swept:
                for name, v in d["kernels"].items():  # name prefix; the steady-state instantiation (FIRST = false)
                    if name.startswith(kernel) and not name.endswith("true>"):
                        return v["traffic_bytes"]
        except Exception:
            pass
    return None


def cpu_baseline(H, probs, msg, iters, method, E, budget_s):
    """The CPU oracle (f32 restatement, oracle/) on a bounded sample of the same workload,
    all host threads over the batch.  A reported baseline -- NOT the reference binary
    (ldpc==0.1.3 / simulate_rs cannot run here) and not the optimisation target."""
    from oracle import pyoracle

    om = {"min_sum": "min_sum", "product_sum": "tanh_complement"}[method]
    threads = max(1, min(os.cpu_count() or 1, pyoracle.max_threads()))
    t0 = time.perf_counter()
    pyoracle.bp_decode_batch(H, probs, msg[:1], 1, iters, om, dtype="f32", threads=1, early_exit=False)
    one = time.perf_counter() - t0
    sample = int(max(threads, min(msg.shape[0], budget_s / max(one, 1e-6) * threads)))
    sample = max(threads, (sample // threads) * threads)
    sample = min(sample, msg.shape[0])
    t0 = time.perf_counter()
    pyoracle.bp_decode_batch(H, probs, msg[:sample], 1, iters, om, dtype="f32", threads=threads, early_exit=False)
    dt = time.perf_counter() - t0
    # the arithmetic the reference's package actually runs -- float64 probability-ratio product-sum, one
    # codeword at a time on one core (oracle method 0) -- on two codewords of the same batch
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = time.perf_counter()
        pyoracle.bp_decode_batch(H, probs, msg[:2], 1, iters, "product_sum", dtype="f64", threads=1, early_exit=False)
        one64 = (time.perf_counter() - t1) / 2
    return {
        "reference_form_single_thread": {
            "value": 2.0 * E * iters / one64, "unit": "directed edge-message updates/s", "cores": 1, "codewords_per_s": 1.0 / one64,
            "what": f"float64 ratio-domain product-sum (the arithmetic of ldpc 0.1.3's bp_decoder, restated), one codeword at a "
                    f"time on one core, {iters} fixed iterations, 2 codewords of the same batch"},
        "value": 2.0 * E * iters * sample / dt,
        "unit": "directed edge-message updates/s",
        "cores": threads,
        "kind": "port",
        "sample": f"first {sample} codewords of the same batch, {iters} fixed iterations, f32 {om}, "
        f"{threads} OpenMP threads over codewords ({dt:.1f} s); restated CPU path, not the reference binary",
        "codewords_per_s": sample / dt,
        "single_thread_value": 2.0 * E * iters / one,  # one codeword on one core, same code
    }


if __name__ == "__main__":
    main()
