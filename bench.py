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
  roofline      `achieved` / `frac` = bytes REALLY MOVED (rocprofv3 FETCH_SIZE + WRITE_SIZE, taken live by two short child
                runs) by a (check + variable) launch pair, all stream lanes, / the pair's HIP-event time, against the
                8 TB/s HBM peak; `kernel` = the one with the larger share of the pair's time; `hbm_frac` = the same in
                the HBM-streaming regime (one tile group = the whole batch); `cache_ceiling_GBps` = the measured ceiling
                of the cache-resident regime (in-place stream of the tile group's size, scaldpc_measure_rmw_stream);
                `algorithmic_*` = SURVEY 8d's figure (16 B per edge-iteration) over the same time -- for min-sum in its
                RECORD form (k_check_minsum_rec / k_var_rec) about 1.6x what is moved, hence not a
                fraction of anything; `updates_elided_frac` = the share of `value`'s updates served without computing a
                message (iteration 1's table, degree-1 columns).  No field labelled `frac` exceeds 1 (self_check).
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
    # the decoder "used in the paper" for Kyber (kyber.py:381-382): DecoderN1280R512SW6 (lib.rs:66-75), B = 2, BSUM = 12,
    # H = make_qary_qc_parity_check_matrix(256, 6, 3, RandomState(0), 2) (512 x 1280), --batch codewords per call
    "kyber_sw6": (None, None, "qary_special", 5),
    # the only perf workloads the reference itself defines: criterion "small decoder" / "medium decoder"
    # (simulate_rs/benches/decoder.rs:38-62, 66-92): Q = 15, 10 iterations, ONE min_sum call on a point-mass channel
    # output with one bad symbol; value = calls / s at batch 1 (latency)
    "criterion_small": (None, None, "qary_min_sum", 10),
    "criterion_medium": (None, None, "qary_min_sum", 10),
}
QARY_WORKLOADS = ("qary_config4", "kyber_sw6", "criterion_small", "criterion_medium")
VALU_PEAK_OPS = 78.6e12  # MI355X_MICROARCH.md: 256 CUs x 128 fp32 lanes x 2.4 GHz, one (non-FMA) VALU op per lane per cycle


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
    ap.add_argument("--pmc-save", default=None, help="write the live PMC bytes per kernel to this file (profiles/r04/"
                    "pmc_traffic_<workload>.json is what a later run without counters falls back to)")
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
        if live_traffic and args.pmc_save:
            with open(args.pmc_save, "w") as fh:
                json.dump(live_traffic, fh, indent=1)

    args.sq_live = None
    if (args.gpus == 1 and args.pmc == "live" and not args.pmc_child and args.workload in ("qary_config4", "kyber_sw6")):
        qb = {"qary_config4": 1024, "kyber_sw6": 256}[args.workload] if args.batch == 4096 else args.batch
        args.sq_live = sq_live(args.workload, qb)
        if args.sq_live and args.pmc_save:
            with open(args.pmc_save, "w") as fh:
                json.dump(args.sq_live, fh, indent=1)

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
    torch.cuda.set_device(local)
    # gloo first (barriers, fallback), then RCCL tried for the end-of-run gathers: a transport failure must not cost the
    # run its line (sca-ldpc_amd/shard.py: Collectives)
    shard = importlib.import_module("sca-ldpc_amd.shard")
    coll = shard.Collectives(rank, world, device=torch.device("cuda", local), want=os.environ.get("BENCH_BACKEND", "nccl"))
    if rank == 0 and coll.note:
        print(f"bench.py: {coll.note}", file=sys.stderr)

    S = importlib.import_module("sca-ldpc_amd")
    bp = importlib.import_module("sca-ldpc_amd.bp")
    lib = importlib.import_module("sca-ldpc_amd._lib")
    trials = importlib.import_module("sca-ldpc_amd.trials")
    lib.check(lib.load().scaldpc_set_device(local))

    hqc, key, method, iters = WORKLOADS[args.workload]
    if args.workload in QARY_WORKLOADS:
        return finish(coll, qary_bench(args, S, rank, world, coll, local, iters))
    rows = json.load(open(os.path.join(ROOT, "tests", "golden", "hqc_first_rows.json")))
    H, Hin, _ = S.codes.hqc_bench_graph(hqc, rows[key])
    N, omega = S.codes.HQC_PARAMS[hqc]
    R, E, n = Hin.m, H.nnz, H.n
    batch = args.batch
    probs = trials.hqc_priors(N, R, omega, args.eps)
    if args.workload != "hqc128_mc":
        msg, ys = trials.hqc_trials(Hin, omega, args.eps, batch, base_seed=2, first_index=rank * batch)

    if args.workload == "hqc128_mc":
        return finish(coll, mc_sweep(args, S, bp, lib, trials, H, N, omega, R, E, probs, iters, method, rank, world, local, coll))
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
        coll.barrier()
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
    per_rank_ms = None
    if world > 1:  # every rank's own time travels to rank 0 (a straggler must not hide behind the MAX)
        per_rank_ms = coll.gather_scalars(dt / args.steps * 1e3)
        dt = max(per_rank_ms) * args.steps / 1e3

    # per-kernel launch durations, HIP events on the launch stream -- taken right behind the timed region, on the very
    # state it left (before the HBM-streaming pass re-sizes the handle's arrays), and held to the driver-timed step: the
    # pairs of a step cannot take longer than the step.  A measurement that says otherwise (seen once: a variable pass at
    # 58 us beside a 61.6 ms step that only fits 42 us) is repeated, up to three times; the attempt with the shortest pair
    # is the one reported, `pair.attempts_us` keeps them all.
    def pair_ms(k):
        return k["ms_check"] / max(1, k["launches_check"]) + k["ms_var"] / max(1, k["launches_var"])

    kt, attempts = None, []
    for _ in range(3):
        k = dec.time_kernels(50, stream=stream)
        attempts.append(pair_ms(k) * 1e3)
        if kt is None or pair_ms(k) < pair_ms(kt):
            kt = k
        groups = -(-batch // max(1, k["codewords"] * k["lanes"]))
        if pair_ms(kt) * groups * (iters - 1) <= 1.02 * dt / args.steps * 1e3:
            break
    iso = None
    if kt["lanes"] == 2:  # the same kernels alone on the chip, one series after the other over the whole tile group
        dec.configure(split=1)
        iso = dec.time_kernels(50, stream=stream)
        dec.configure(split=2)

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

    ms_check = kt["ms_check"] / max(1, kt["launches_check"])
    ms_var_pass = kt["ms_var"] / max(1, kt["launches_var"])
    swept = kt["codewords"]  # codewords per check launch (tile padded)
    lanes = kt["lanes"]  # 2: timed in the decode's own two-stream launch pattern (one event per launch)

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
    out_host = d_out.cpu().numpy()
    ok = trials.success(out_host, ys, N).astype(np.uint8)
    ok_all = coll.gather(ok, batch * world)  # RCCL all_gather (gloo if RCCL did not come up)
    ranks_seen = coll.gather(np.array([rank], dtype=np.int32), world)
    succ = float(ok_all.mean())
    conv = float(d_conv.float().mean().item())

    rc = 0
    if rank == 0:
        total_cw = batch * world * args.steps
        updates = 2.0 * E * iters * total_cw
        value = updates / dt
        check_gbs = 8.0 * E * swept / (ms_check * 1e-3) / 1e9  # 4 B read + 4 B written per edge per codeword
        var_gbs = 8.0 * E * kt["codewords_var"] / (ms_var_pass * 1e-3) / 1e9
        rec = bool(kt.get("record_form"))  # min-sum in its record form (knob minsum_rec): other kernels, same algorithmic bytes
        kt_first_fused = os.environ.get("SCALDPC_FIRST_FUSED", "1") != "0"  # (HQC graphs: every row and column is register-resident)
        cname = ("k_check_minsum_rec" if rec else "k_check_minsum_x") if method == "min_sum" else "k_check_tanh"
        vname = "k_var_rec" if rec else "k_var"
        cpmc = ("k_check_minsum_rec" if rec else "k_check_minsum") if method == "min_sum" else "k_check_tanh"
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
            "collective_backend": coll.backend,  # "nccl" = RCCL; "gloo" = RCCL did not come up (collective_note says why); "none" = one rank
            "collective_note": coll.note or None,
            "codewords_per_s": total_cw / dt,
            "whole_job_algorithmic_GBps": 16.0 * E * iters * total_cw / dt / 1e9,
            "decode_success_rate": succ,
            "converged_rate": conv,
            "kernel_ms": {"check_per_launch": ms_check, "var_pass": ms_var_pass},
            "hbm_copy_ceiling_GBps": copy_gbs,  # measured: 1 GiB device copy, read+write bytes/s
        }
        if per_rank_ms:
            out["per_rank_ms_per_step"] = per_rank_ms
            out["rank_ms_min_max"] = [min(per_rank_ms), max(per_rank_ms)]
        # ---- roofline: what the hardware did --------------------------------------------------------------------------
        # Under the two-lane schedule no kernel runs alone: each stream alternates check and variable launches over its
        # half of the tile group, one kernel out of phase with the other stream.  The unit of account is therefore the
        # (check + variable) launch PAIR of one lane; `lanes` of them run side by side.
        #   achieved / frac   bytes REALLY MOVED over a pair (rocprofv3 FETCH_SIZE + WRITE_SIZE of both kernels, gfx950
        #                     corrections applied) x lanes / the pair's HIP-event time, against the 8 TB/s HBM datasheet
        #                     peak.  These are L2 <-> fabric bytes; a cache-resident tile group is served by the 256 MiB
        #                     Infinity Cache, which is why the figure can sit above what HBM streaming sustains
        #                     (hbm_copy_ceiling_GBps) -- `cache_ceiling_GBps` is the measured ceiling of that regime.
        #   algorithmic_*     SURVEY 8(d)'s figure (two fp32 message arrays: 16 B per edge, codeword and iteration) over
        #                     the same time.  Not a hardware rate: the record form of min-sum moves about half of it.
        dom_is_var = ms_var_pass >= ms_check
        dname = vname if dom_is_var else cname
        pair_s = (ms_check + ms_var_pass) * 1e-3
        algorithmic = lanes * (8.0 * E * swept + 8.0 * E * kt["codewords_var"]) / pair_s / 1e9
        tr = live_traffic if (live_traffic and live_traffic.get("codewords_per_launch") == swept) else None
        if tr is None and args.pmc != "off":
            tr = pmc_traffic(args.workload, swept)
        tc = tr["kernels"].get(cpmc) if tr else None
        tv = tr["kernels"].get(vname) if tr else None
        n1 = int((H.col_degrees() == 1).sum())  # edges into columns of degree 1 (the identity block of an HQC graph)
        if tc and tv:
            bytes_check, bytes_var, bytes_src = tc["traffic_bytes"], tv["traffic_bytes"], tr["source"]
        else:  # no counters: the design's byte model, per launch
            if rec:
                bytes_check = swept * (4.0 * E + 8.0 * R + E / 4.0)
                bytes_var = kt["codewords_var"] * (4.0 * (E - (n1 if kt.get("var_slim") else 0)) + E / 4.0 + 8.0 * R)
            else:
                bytes_check, bytes_var = 8.0 * E * swept, 8.0 * E * kt["codewords_var"]
            bytes_src = "byte MODEL of DESIGN.md section 4 (no PMC counters available in this run, none committed for this geometry)"
        moved = lanes * (bytes_check + bytes_var) / pair_s / 1e9
        # iteration 1 and the elided updates, stated: with first_fused the first check pass is a table (no message is
        # computed per codeword), and in the record form's passes without output the columns of degree <= 1 are not rewritten
        elided = (E if kt_first_fused else 0) + (n1 * max(0, iters - 2) if (rec and kt.get("var_slim")) else 0)
        out["roofline"] = {
            "bound": "hbm",
            "served_by": "infinity-cache (a tile group's message array is sized to stay resident: 4 tiles = 209 MB on this graph); "
                         "the bytes counted are L2 <-> fabric bytes, which HBM would carry if the cache did not",
            "kernel": dname,
            "achieved": moved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": moved / HBM_PEAK_GBS,
            "achieved_is": "bytes really moved by a (check + variable) launch pair (FETCH_SIZE + WRITE_SIZE of both kernels) x lanes "
                           "/ the pair's HIP-event time",
            "traffic": bytes_var if dom_is_var else bytes_check,
            "traffic_source": bytes_src,
            "pair": {"lanes": lanes, "us": pair_s * 1e6, "bytes_per_pair_and_lane": bytes_check + bytes_var,
                     "bytes": {cname: bytes_check, vname: bytes_var}, "attempts_us": attempts,
                     # the step's own pairs: (ms_per_step - everything else) / pairs per step cannot be shorter than this
                     "consistent_with_step": bool(pair_s * 1e3 * (-(-batch // max(1, swept * lanes))) * (iters - 1) <= 1.02 * dt / args.steps * 1e3)},
            "algorithmic_GBps": algorithmic,
            "algorithmic_frac": algorithmic / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": 8.0 * E * (kt["codewords_var"] if dom_is_var else swept),
            "scheme": "algorithmic_* = SURVEY 8(d): 8 B per directed edge-message update (one fp32 read + one written), 16 E per "
                      "codeword and iteration, over the same pair time.  " + (
                          "Min-sum runs in its RECORD form: the check pass reads 4E and writes two magnitudes per row + two lane masks "
                          "per edge (8m + E/4), the variable pass writes 4E and reads E/4 of masks + the row records: about 8.7E per "
                          "codeword and iteration through the fabric instead of 16E, so the algorithmic figure exceeds the hardware's "
                          "rate and is NOT a roofline fraction" if rec else
                          "Message form: both passes read and write every message, moved bytes = algorithmic bytes + index / prior / "
                          "plane traffic"),
            "lanes": lanes,
            "timed_variable_pass": {"writes_output": bool(kt.get("var_writes_out")), "without_degree_le1_columns": bool(kt.get("var_slim"))},
            "dominant": {"name": dname, "us": (ms_var_pass if dom_is_var else ms_check) * 1e3,
                         "share_of_pair_time": max(ms_var_pass, ms_check) / (ms_check + ms_var_pass)},
            "per_launch": {
                cname: {"codewords": swept, "us": ms_check * 1e3, "moved_bytes": bytes_check, "algorithmic_GBps": check_gbs},
                vname: {"codewords": kt["codewords_var"], "us": ms_var_pass * 1e3, "moved_bytes": bytes_var, "algorithmic_GBps": var_gbs},
            },
            "updates_elided_frac": elided / (2.0 * E * iters),
            "updates_elided_what": "`value` counts 2 E updates per codeword and iteration; of those, iteration 1's check-to-variable "
                                   "messages come from a per-edge table (first_fused) and, in fixed-iteration record-form runs, the "
                                   "messages out of columns of degree <= 1 (always the prior) are written once instead of every pass: "
                                   "identical results, fewer message updates actually performed",
        }
        # whole step in moved bytes: (iters - 1) pairs per codeword group + the first variable pass, over ms_per_step
        per_cw_pair = (bytes_check / max(1, swept) + bytes_var / max(1, kt["codewords_var"]))
        first_cw = None
        if tr and tr["kernels"].get("k_var_first"):
            first_cw = tr["kernels"]["k_var_first"]["traffic_bytes"] / max(1, swept)
        step_bytes = batch * ((iters - 1) * per_cw_pair + (first_cw if first_cw is not None else per_cw_pair))
        out["roofline"]["whole_step"] = {"moved_GBps": step_bytes / (dt / args.steps) / 1e9,
                                         "frac": step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                                         "what": "PMC bytes per codeword of every launch of a step (first variable pass + iters - 1 pairs) "
                                                 "x batch / ms_per_step: the driver-timed figure, launch gaps and I/O kernels included"}
        if hbm_stream:
            # the same kernels with ONE tile group = the whole batch, far beyond the cache: bytes per codeword-iteration
            # from the counters above x what the pass ran / its time.  This is an HBM rate and must not exceed the copy ceiling.
            sb = batch * ((iters - 1) * per_cw_pair + (first_cw if first_cw is not None else per_cw_pair))
            hbm_stream["moved_GBps"] = sb / (hbm_stream["ms_per_step"] * 1e-3) / 1e9
            hbm_stream["algorithmic_GBps"] = hbm_stream.pop("GBps")
            hbm_stream["what"] = ("one tile group = the whole batch (%.0f MB of messages): every pass streams from HBM; moved_GBps = PMC "
                                  "bytes per codeword-iteration (cache-resident geometry) x codeword-iterations / time" % hbm_stream["group_MB"])
            out["roofline"]["hbm_streaming"] = hbm_stream
            out["roofline"]["hbm_frac"] = hbm_stream["moved_GBps"] / HBM_PEAK_GBS
        # measured ceilings of the two regimes, this GPU, this run: the in-place read-all / write-all stream an in-place
        # BP pass is made of (a wave reads 51 consecutive 256-B rows and writes them back), at the size of a cache-resident
        # tile group and far beyond the cache
        try:
            group_bytes = 4.0 * E * swept * lanes
            out["roofline"]["cache_ceiling_GBps"] = lib.measure_rmw_stream(int(group_bytes), 51, 50)
            out["roofline"]["cache_ceiling_what"] = ("in-place read-all / write-all stream (51 rows of 256 B per wave, the row in registers: every "
                                                     "load issued before the first use, as the row kernels do) over %.0f MB = the tile group's "
                                                     "message array: the Infinity-Cache regime; any read : write mix of this shape measures the "
                                                     "same total (profiles/r04/stream_modes.log)" % (group_bytes / 1e6))
            out["roofline"]["hbm_rmw_ceiling_GBps"] = lib.measure_rmw_stream(int(16 * group_bytes), 51, 6)
            out["roofline"]["frac_of_cache_ceiling"] = moved / out["roofline"]["cache_ceiling_GBps"]
        except Exception as ex:  # (a measurement aid must not cost the run its line)
            out["roofline"]["cache_ceiling_GBps"] = None
            out["roofline"]["cache_ceiling_error"] = str(ex)
        if tr:
            out["roofline"]["traffic_all_kernels"] = tr["kernels"]
        if iso:
            ic = iso["ms_check"] / max(1, iso["launches_check"])
            iv = iso["ms_var"] / max(1, iso["launches_var"])
            scale_c, scale_v = iso["codewords"] / max(1, swept), iso["codewords_var"] / max(1, kt["codewords_var"])
            out["roofline"]["isolated"] = {
                cname: {"codewords": iso["codewords"], "us": ic * 1e3, "moved_GBps": bytes_check * scale_c / (ic * 1e-3) / 1e9,
                        "algorithmic_GBps": 8.0 * E * iso["codewords"] / (ic * 1e-3) / 1e9},
                vname: {"codewords": iso["codewords_var"], "us": iv * 1e3, "moved_GBps": bytes_var * scale_v / (iv * 1e-3) / 1e9,
                        "algorithmic_GBps": 8.0 * E * iso["codewords_var"] / (iv * 1e-3) / 1e9},
                "what": "the same kernels alone on the chip, one series after the other over the whole tile group",
            }
            out["roofline"]["dominant"]["frac_alone"] = out["roofline"]["isolated"][dname]["moved_GBps"] / HBM_PEAK_GBS
        # self-check: nothing labelled a fraction of the HBM peak may exceed 1, and the HBM-streaming rate in moved bytes
        # cannot beat this GPU's own copy ceiling by more than measurement noise
        sc = {"frac_le_1": out["roofline"]["frac"] <= 1.0,
              "whole_step_frac_le_1": out["roofline"]["whole_step"]["frac"] <= 1.0}
        if hbm_stream:
            sc["hbm_streaming_le_copy_ceiling"] = hbm_stream["moved_GBps"] <= 1.08 * copy_gbs
        if out["roofline"].get("cache_ceiling_GBps"):  # the pair cannot beat a plain stream over the same cache-resident bytes
            sc["pair_le_cache_ceiling"] = moved <= 1.08 * out["roofline"]["cache_ceiling_GBps"]  # (the tanh pair reads 7.2 against 6.97: same noise band as above)
        out["roofline"]["self_check"] = sc
        if not all(sc.values()):
            print(f"bench.py: roofline self-check failed: {sc}", file=sys.stderr)
        if args.parity_rows > 0:
            out.update(parity_check(H, probs, msg, out_host, iters, method, min(args.parity_rows, batch), swept, lanes))
            rc = 0 if out["parity_ok"] else 3
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(H, probs, msg, iters, method, E, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    dec.close()
    return finish(coll, rc)


def finish(coll, rc):
    """Tear the process groups down and leave with `rc`.  A run that had to abandon an RCCL collective in flight (probe
    timeout) leaves through os._exit: its teardown may never return, and its line is already printed."""
    sys.stdout.flush()
    sys.stderr.flush()
    if coll.abandoned_collective:
        os._exit(rc or 0)
    coll.close()
    if rc:
        raise SystemExit(rc)


def parity_check(H, probs, msg, out_host, iters, method, rows, swept=128, lanes=2):
    """After the timed region: `rows` codewords of the timed output against the CPU oracle (f32, same operation order;
    test infrastructure, never part of what is timed), spread over the FIRST, a MIDDLE and the LAST tile group of the
    step and over every stream lane of each (a group's tiles are dealt to the lanes in order: `swept` codewords per
    lane).  min-sum: every bit equal.  tanh rule: bits equal wherever the oracle's posterior is outside the fp32
    tolerance of tests/helpers.compare (|L| > 2e-4 + 2e-4 |L|)."""
    from oracle import pyoracle

    om = {"min_sum": "min_sum", "product_sum": "tanh_complement"}[method]
    batch = msg.shape[0]
    group = max(64, swept * lanes)
    ngroups = max(1, -(-batch // group))
    starts = []
    for g in sorted({0, ngroups // 2, ngroups - 1}):
        for ln in range(lanes):
            starts.append(min(batch - 1, g * group + ln * swept))
    per = max(1, rows // len(starts))
    idx = np.unique(np.concatenate([np.arange(st, min(batch, st + per)) for st in starts]))
    threads = max(1, min(os.cpu_count() or 1, pyoracle.max_threads(), len(idx)))
    ref = pyoracle.bp_decode_batch(H, probs, msg[idx], 1, iters, om, dtype="f32", threads=threads, early_exit=False)
    diff = out_host[idx] != ref["bits"]
    if method != "min_sum":
        with np.errstate(invalid="ignore"):
            diff &= np.abs(ref["llr"]) > 2e-4 + 2e-4 * np.abs(ref["llr"])
    return {"parity_checked": int(len(idx)), "parity_mismatched_bits": int(diff.sum()), "parity_ok": bool(not diff.any()),
            "parity_against": f"oracle f32 {om}, {iters} fixed iterations, {per} codewords from each stream lane of the first, a "
                              f"middle and the last tile group of the timed output (codewords {', '.join(str(x) for x in starts)} ...)"}


def rocprof_pmc_passes(passes, child_args, prefixes, skip=None, timeout=300):
    """rocprofv3 counter passes of `bench.py --pmc-child <child_args>` -- short runs of the very workload, started as
    child processes BEFORE this process touches the GPU (the profiler's preloaded library initialises the GPU in front
    of the program it is given, so the program itself comes right after `--`: python3 bench.py, no wrapper).
    `passes` = [(counter names of one pass, {counter: scale})]; counters that do not fit one pass go in separate ones
    (FETCH_SIZE and WRITE_SIZE: MI355X_MICROARCH.md, PMC slots), never together with a trace domain other than
    --kernel-trace.  Returns ({kernel base name: {counter: average per dispatch, "dispatches": n}}, the child's
    geometry line) or (None, None) if anything goes wrong.  Per kernel the most common grid is taken: the schedule's
    own steady-state launches."""
    import csv
    import re
    import shutil
    import tempfile
    from collections import defaultdict

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe) or "rocprof" in os.environ.get("LD_PRELOAD", "") or os.environ.get("ROCPROFILER_REGISTER_FORCE_LOAD"):
        return None, None  # no profiler, or this process is itself being profiled
    env = dict(os.environ, TMPDIR="/tmp")
    per, geom = {}, None
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        for idx, (counters, scales) in enumerate(passes):
            out_dir = os.path.join(td, f"pass{idx}")
            cmd = [exe, "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", out_dir, "-o", "p",
                   "--", sys.executable, os.path.abspath(__file__), "--pmc-child", *child_args,
                   "--no-cpu-baseline", "--pmc", "off", "--parity-rows", "0"]
            try:
                r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=timeout)  # (a pass takes 10-20 s)
            except Exception:
                return None, None
            if r.returncode != 0:
                return None, None
            for ln in r.stdout.splitlines():
                if ln.startswith("{") and "pmc_child" in ln:
                    geom = json.loads(ln)
            path = None
            for root, _, files in os.walk(out_dir):
                for f in files:
                    if f.endswith("counter_collection.csv"):
                        path = os.path.join(root, f)
            if not path:
                return None, None
            tot, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
            with open(path) as fh:
                for row in csv.DictReader(fh):
                    c = row["Counter_Name"]
                    if c not in counters:
                        continue
                    name = re.sub(r"^void ", "", row["Kernel_Name"].replace("(anonymous namespace)::", "")).split("(")[0]
                    if not name.startswith(prefixes) or (skip and re.search(skip, name)):
                        continue
                    base = name.split("<")[0]
                    key = (base[:-2] if base.endswith("_x") else base, row["Grid_Size"])
                    tot[key][c] += float(row["Counter_Value"]) * scales.get(c, 1.0)
                    cnt[key][c] += 1
            best = {}
            for (name, grid), c in cnt.items():
                n = max(c.values())
                if name not in best or n > best[name][1]:
                    best[name] = (grid, n)
            for name, (grid, n) in best.items():
                d = per.setdefault(name, {})
                for c in tot[(name, grid)]:
                    d[c] = tot[(name, grid)][c] / cnt[(name, grid)][c]
                d["dispatches"] = n
    if not geom or not per:
        return None, None
    return per, geom


def pmc_live(workload, batch=256, extra=()):
    """roofline.traffic measured in THIS run: two rocprofv3 passes (FETCH_SIZE and WRITE_SIZE do not fit one) of
    `bench.py --pmc-child` -- the same workload, one 256-codeword slice (the launch geometry of the timed run: a
    cache-resident tile group per lane), 6 iterations.  Units and gfx950 corrections as MI355X_MICROARCH.md's HBM
    section prescribes (KiB; FETCH_SIZE x2: 128-byte read requests are tallied at 64 B; WRITE_SIZE x1; re-verified on
    profiles/microbench/rmw_stream).  These are L2 <-> fabric bytes: Infinity-Cache hits are counted, so they bound HBM
    bytes from above.  Returns None if anything goes wrong."""
    per, geom = rocprof_pmc_passes([(("FETCH_SIZE",), {"FETCH_SIZE": 2048.0}), (("WRITE_SIZE",), {"WRITE_SIZE": 1024.0})],
                                   ["--workload", workload, "--batch", str(batch), *extra], ("k_var", "k_check"),
                                   skip=r"^k_check[^<]*<[^,>]+, true, false")  # (FIRST = true: iteration 1 reads the priors, not the messages)
    if not per:
        return None
    kernels = {k: {"fetch_bytes": v.get("FETCH_SIZE"), "write_bytes": v.get("WRITE_SIZE"),
                   "traffic_bytes": (v.get("FETCH_SIZE") or 0.0) + (v.get("WRITE_SIZE") or 0.0), "dispatches": v["dispatches"]}
               for k, v in per.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v}
    return {"workload": workload, "codewords_per_launch": geom["codewords_per_launch"], "kernels": kernels,
            "source": "live rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run"}


def sq_live(workload, batch):
    """q-ary workloads: ONE rocprofv3 pass of SQ counters over a short child run -- what the check kernel really
    ISSUED (SQ_INSTS_VALU: wave-level VALU instructions; SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES / SQ_WAVE_CYCLES in
    quad-cycles), beside the reference-operation count the roofline line is defined on."""
    per, geom = rocprof_pmc_passes([(("SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES",
                                      "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"), {})],
                                   ["--workload", workload, "--batch", str(batch)], ("k_q_",))
    if not per:
        return None
    return {"kernels": per, "batch": geom.get("batch"), "source": "live rocprofv3 --pmc SQ_* pass of this run"}


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
    # preflight: are there N GPUs?  (not for the gloo rehearsals, which put several ranks on one device or none)
    if (os.environ.get("BENCH_BACKEND", "nccl") == "nccl" and os.environ.get("BENCH_FORCE_DEVICE") is None
            and not (args.rendezvous_only and os.environ.get("SCALDPC_FORCE_NCCL_FAILURE") == "1")):  # (the CPU rehearsal of the fallback)
        have, how = visible_gpu_count()
        if have is not None and have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s) ({how}); nothing launched", file=sys.stderr)
            raise SystemExit(2)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    # dmabuf IPC: RCCL across processes needs it on this pool's driver.  Only a default: a value the environment already
    # carries is left alone, and the value in effect is printed with the reason if RCCL does not come up (collective_note)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def visible_gpu_count():
    """GPUs a child process would see, WITHOUT initialising HIP here (the parent of the ranks must never touch the
    GPU).  KFD topology: one node per agent, GPUs are the nodes with simd_count > 0; a *_VISIBLE_DEVICES list narrows
    it.  Falls back to a 3-line child that asks the library (scaldpc_device_count); (None, reason) if neither works."""
    import glob

    n, how = None, None
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if nodes:
        n = 0
        for f in nodes:
            try:
                for ln in open(f):
                    if ln.startswith("simd_count") and int(ln.split()[1]) > 0:
                        n += 1
            except OSError:
                pass
        how = "KFD topology"
    if not n:
        code = ("import ctypes,sys;l=ctypes.CDLL(sys.argv[1]);c=ctypes.c_int(0);"
                "rc=l.scaldpc_device_count(ctypes.byref(c));print(c.value if rc==0 else -1)")
        try:
            so = os.path.join(ROOT, "sca-ldpc_amd", "libscaldpc.so")
            r = subprocess.run([sys.executable, "-c", code, so], capture_output=True, text=True, timeout=120)
            n, how = int(r.stdout.strip().splitlines()[-1]), "scaldpc_device_count in a child process"
            if n < 0:
                return None, "device count unavailable"
        except Exception:
            return None, "device count unavailable"
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
            how += f", {var}={v}"
    return n, how


def rendezvous_only(args, torch, dist, rank, world, local):
    """The N > 1 launch path without the decode: process groups up (gloo, then RCCL tried: shard.Collectives), one
    all_gather of the rank ids (the same collective the end-of-run gather uses), rank 0 prints what it saw and over
    which backend."""
    want = os.environ.get("BENCH_BACKEND", "nccl")
    dev = None
    if want == "nccl" and torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("BENCH_FORCE_DEVICE", local)))
        dev = torch.device("cuda", torch.cuda.current_device())
    shard = importlib.import_module("sca-ldpc_amd.shard")
    coll = shard.Collectives(rank, world, device=dev, want=want)
    seen = coll.gather(np.array([rank], dtype=np.int32), world)
    coll.barrier()
    if rank == 0:
        print(json.dumps({"rendezvous_only": True, "n_gpus": world, "backend": coll.backend, "collective_note": coll.note or None,
                          "rccl_ranks": int(len(set(seen.tolist())))}), flush=True)
    return finish(coll, 0)


def qary_case(workload, S, batch, rank):
    """(class name, H int8, inputs tuple, oracle callable, description) of a q-ary workload."""
    from oracle import pyoracle  # (only the cpu_baseline / parity legs call it)

    gens = json.load(open(os.path.join(ROOT, "tests", "golden", "generators.json")))
    rng = np.random.RandomState(7 + rank)
    if workload == "qary_config4":
        g = S.TannerGraph.from_coo(gens["regular_identity_300_150_3_6_s1"])
        p = 1 / 3
        good, bad = np.array([p, 1.75 * p, 0.25 * p]), np.array([p, 0.25 * p, 1.75 * p])  # decode.py:232-237
        mask = rng.rand(batch, g.n) < 0.005
        pmf = np.where(mask[:, :, None], bad, good).astype(np.float32)
        return ("DecoderN450R150V3C7B1", g, (pmf,), lambda x, it, th: pyoracle.qary_min_sum_batch(g, 3, x[0], it, threads=th),
                f"q-ary min-sum DecoderN450R150V3C7B1 (150x450, E={g.nnz}, Q=3), good/bad pmfs of decode.py:232-237 at error rate 0.005")
    if workload == "kyber_sw6":
        g = S.TannerGraph.from_coo(gens["qary_qc_256_6_3_s0_cb2"])
        pb = rng.dirichlet(np.ones(5), size=(batch, 768)).astype(np.float32)
        ps = rng.dirichlet(np.ones(25), size=(batch, 512)).astype(np.float32)
        return ("DecoderN1280R512SW6", g, (pb, ps), lambda x, it, th: pyoracle.qary_special_batch(g, 2, 12, x[0], x[1], it, threads=th),
                f"DecoderSpecial N1280R512SW6 (512x1280, E={g.nnz}, B=2, BSUM=12: 5^6 assignments per check), Dirichlet pmfs")
    # criterion cases: zero message, zero noise, one bad symbol (benches/decoder.rs:48-56, 80-88)
    if workload == "criterion_small":
        Hd = np.array([[1, 1, 1, 1, 0, 0], [0, 0, 1, 1, 0, 1], [1, 0, 0, 1, 1, 0]], dtype=np.int8)
        g, name = S.TannerGraph.from_dense(Hd), "DecoderN6R3V3C4B7"
    else:
        g = S.TannerGraph.from_coo(json.load(open(os.path.join(ROOT, "tests", "golden", "parity_check_150_450.json"))))
        name = "DecoderN450R150V3C7B7"
    ch = np.zeros((batch, g.n, 15), dtype=np.float32)
    ch[:, :, 7] = 1.0
    ch[:, 1, 7], ch[:, 1, 14] = 0.1, 0.9
    return (name, g, (ch,), lambda x, it, th: pyoracle.qary_min_sum_batch(g, 15, x[0], it, threads=th),
            f"criterion '{workload.split('_')[1]} decoder' (simulate_rs/benches/decoder.rs): {name}, Q=15, point-mass channel output with one bad symbol")


def qary_reference_ops(g, cls, special):
    """VALU work of one iteration of one codeword AS THE REFERENCE DOES IT: every check enumerates its assignments
    (decoder.rs:585-631: Q^(k-1) for a check of degree k when every message entry is finite, the last value being
    minus the sum of the others; decoder_special.rs:531-554: (2B+1)^(k-1), unfiltered) and does, per assignment, k adds
    for the sum, k subtractions `sum - alpha_j` and k minima: 3k operations."""
    deg = np.diff(g.row_ptr).astype(np.int64)
    q = cls.Q
    return float(((float(q) ** (deg - 1)) * 3 * deg).sum()), float((float(q) ** (deg - 1)).sum())


def qary_bench(args, S, rank, world, coll, local, iters):
    """BASELINE config 4 and its q-ary siblings through the simulate_rs-shaped classes: `--steps` calls of
    min_sum_batch on `--batch` codewords per rank, channel outputs resident in HBM when the timed region starts and
    symbols left there (the probability -> LLR conversion runs on the device, k_q_into_llr, inside the call); the
    same call with host arrays in and out -- what the PyO3 class takes, PCIe included -- is timed beside it
    (`host_buffers`) and must give the same symbols.
    value = directed symbol-edge message updates / s (2 * E * batch * iterations per call).
    roofline: ALU-bound, no HBM claim: reference-ops / s of the check kernel from its HIP-event
    time against the fp32 VALU peak.  cpu_baseline: the C oracle (decoder.rs restated) on the same inputs."""
    import torch

    qary = importlib.import_module("sca-ldpc_amd.qary")
    special = args.workload == "kyber_sw6"
    default_batch = {"qary_config4": 1024, "kyber_sw6": 256}.get(args.workload, 1)
    batch = default_batch if args.batch == 4096 else args.batch
    name, g, inputs, oracle_call, what = qary_case(args.workload, S, batch, rank)
    cls = qary.decoder_class(name)
    dec = cls(g.to_dense(np.int8), iters)
    dev = torch.device("cuda", local)
    d_in = [torch.from_numpy(x).to(dev) for x in inputs]  # resident in HBM before the timed region (the contract's `value`)
    d_out = torch.empty((batch, g.n), dtype=torch.int8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        dec.min_sum_batch_device(*[t.data_ptr() for t in d_in], batch, d_out.data_ptr(), stream=stream)

    if args.pmc_child:  # what the SQ counter pass profiles: two calls, nothing else
        with np.errstate(divide="ignore"):
            step()
            step()
        torch.cuda.synchronize()
        print(json.dumps({"pmc_child": True, "batch": batch}), flush=True)
        dec.close()
        return
    with np.errstate(divide="ignore"):
        for _ in range(max(1, args.warmup)):
            step()
        torch.cuda.synchronize()
        coll.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        coll.barrier()
        dt = time.perf_counter() - t0
        out = d_out.cpu().numpy()
        # the same call as the PyO3 class takes it: host arrays in, symbols out (PCIe both ways) -- reported, never `value`
        for _ in range(2):
            out_host = dec.min_sum_batch(*inputs)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            out_host = dec.min_sum_batch(*inputs)
        dt_host = (time.perf_counter() - t1) / args.steps
        assert np.array_equal(out_host, out), "host-buffer and device-buffer calls disagree"
        # kernel times: one more call with the launches bracketed by HIP events on the handle's stream
        dec.configure(timing=1)
        step()
        kt = dec.last_timing()
        dec.configure(timing=0)
    per_rank_ms = None
    if world > 1:
        per_rank_ms = coll.gather_scalars(dt / args.steps * 1e3)
        dt = max(per_rank_ms) * args.steps / 1e3
    rc = 0
    if rank == 0:
        ops, assignments = qary_reference_ops(g, cls, special)
        ms_check = kt["ms_check"] / kt["iterations"]
        line = {
            "metric": "edge_message_updates_per_s", "value": 2.0 * g.nnz * batch * iters * args.steps * world / dt,
            "unit": "directed symbol-edge message updates/s", "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup),
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{what}, {iters} iterations, batch {batch}/GPU, pmf resident in HBM / symbols left in HBM "
                                   f"(probability->LLR conversion on the device, inside the timed call)", "batch_per_gpu": batch, "iters": iters},
            "codewords_per_s": batch * args.steps * world / dt, "calls_per_s": args.steps * world / dt,
            "collective_backend": coll.backend, "collective_note": coll.note or None,
            "host_buffers": {"ms_per_step": dt_host * 1e3, "value": 2.0 * g.nnz * batch * iters / dt_host,
                             "what": "the same call with host arrays in and out, as the PyO3 class takes them (PCIe both ways included)"},
            "all_zero_rate": float((out == 0).all(axis=1).mean()),
            "kernel_ms": {"check_per_launch": ms_check, "var_per_launch": kt["ms_var"] / kt["iterations"],
                          "iteration_loop": kt["ms_loop"]},
            "roofline": {
                "bound": "valu", "kernel": kt["check_kernel"], "unit": "Tops/s",
                "achieved": ops * batch / (ms_check * 1e-3) / 1e12, "peak": VALU_PEAK_OPS / 1e12,
                "frac": ops * batch / (ms_check * 1e-3) / VALU_PEAK_OPS, "traffic": None,
                "frac_is": "REFERENCE operations (a derived count, below) / time / peak; what the kernel really issued: `executed`",
                "reference_ops_per_launch": ops * batch, "assignments_per_launch": assignments * batch,
                "what": "reference-ops = 3k per enumerated assignment of a degree-k check (k adds, k subtractions, k minima: "
                        "decoder.rs:600-627), Q^(k-1) assignments per check; achieved = that count / the check kernel's "
                        "HIP-event launch time; peak = fp32 VALU lane-ops/s (no FMA: add / sub / min only).  A kernel "
                        "that prunes or shares work between assignments can exceed what it 'should' -- the count is the "
                        "reference's, not the kernel's",
                "share_of_loop_time": kt["ms_check"] / max(kt["ms_loop"], 1e-9),
            },
        }
        sq = getattr(args, "sq_live", None)
        if not sq and args.pmc != "off":  # instruction counts do not vary from run to run: the committed pass of this batch serves
            try:
                sq = json.load(open(os.path.join(ROOT, "profiles", "r04", f"sq_counters_{args.workload}_b{batch}.json")))
                sq["source"] = f"profiles/r04/sq_counters_{args.workload}_b{batch}.json (committed rocprofv3 --pmc SQ_* pass of this workload and batch)"
            except Exception:
                sq = None
        ck = (kt["check_kernel"] or "").split("<")[0]
        if sq and sq.get("batch") == batch and ck in sq["kernels"]:
            c = sq["kernels"][ck]
            wc = c.get("SQ_WAVE_CYCLES") or 0.0
            line["roofline"]["executed"] = {
                "valu_wave_instructions_per_launch": c.get("SQ_INSTS_VALU"),
                "executed_valu_frac": (c.get("SQ_INSTS_VALU") or 0.0) * 64.0 / (ms_check * 1e-3) / VALU_PEAK_OPS,
                "instructions_per_reference_op": (c.get("SQ_INSTS_VALU") or 0.0) * 64.0 / max(ops * batch, 1.0),
                "wave_cycle_shares": {"issuing_valu": (c.get("SQ_ACTIVE_INST_VALU") or 0.0) / wc, "parked_on_waitcnt": (c.get("SQ_WAIT_ANY") or 0.0) / wc,
                                      "issue_stalled": (c.get("SQ_WAIT_INST_ANY") or 0.0) / wc} if wc else None,
                "waves_per_launch": c.get("SQ_WAVES"), "dispatches_profiled": c.get("dispatches"), "source": sq["source"],
                "what": "what the check kernel ISSUED: SQ_INSTS_VALU (wave-level VALU instructions, all 64 lanes counted whether "
                        "active or not) x 64 / the un-profiled HIP-event launch time / the fp32 VALU lane-op peak",
            }
            # the roofline fraction is the MEASURED one; the reference-operation figure (which credits the kernel with work it
            # no longer does -- shared prefix sums, one subtraction per output instead of one per candidate -- and exceeds 1)
            # moves aside
            ex = line["roofline"]["executed"]
            line["roofline"].update({
                "reference_op": {"Tops": line["roofline"]["achieved"], "frac_of_peak": line["roofline"]["frac"],
                                 "what": line["roofline"].pop("what"), "not_a_hardware_fraction": True},
                "achieved": ex["valu_wave_instructions_per_launch"] * 64.0 / (ms_check * 1e-3) / 1e12,
                "frac": ex["executed_valu_frac"],
                "frac_is": "executed VALU lane-operations of the check kernel (SQ_INSTS_VALU x 64) / its HIP-event launch time / the fp32 "
                           "VALU peak; the reference's operation count over the same time is under `reference_op`",
            })
        else:  # no counters at all: say what the figure is, and never call a derived count above 1 a fraction
            line["roofline"]["reference_op"] = {"Tops": line["roofline"]["achieved"], "frac_of_peak": line["roofline"]["frac"],
                                                "not_a_hardware_fraction": True}
            if line["roofline"]["frac"] > 1.0:
                line["roofline"]["frac"] = None
        if args.workload.startswith("criterion"):
            # point-mass channel outputs: almost every message entry is +inf and the reference enumerates finite
            # supports only (decoder.rs:281-401) -- a handful of assignments per check, nothing like Q^(k-1).  The call
            # is LAUNCH-bound: 2 * iterations + 4 launches and two copies around microseconds of arithmetic.
            launches = 2 * kt["iterations"] + 3  # (status fill, conversion, the loop, unpack)
            line["roofline"] = {
                "bound": "launch", "kernel": kt["check_kernel"], "unit": "launches/s", "achieved": launches * args.steps * world / dt,
                "peak": None, "frac": None, "traffic": None, "launches_per_call": launches,
                "device_loop_ms": kt["ms_loop"], "call_ms": dt / args.steps * 1e3,
                "what": "batch-1 latency of the reference's criterion case; no VALU / HBM fraction is claimed: the work is a few "
                        "finite-support assignments per check, the time is launch and copy latency (compare "
                        "cpu_baseline.single_thread_ms_per_call: one host core runs the same call in 0.04 ms (6x3) / 0.37 ms (150x450))",
            }
        if per_rank_ms:
            line["per_rank_ms_per_step"] = per_rank_ms
        if args.parity_rows > 0:
            rows = min(args.parity_rows, batch)
            with np.errstate(divide="ignore"):
                ref = oracle_call(tuple(x[:rows] for x in inputs), iters, max(1, min(os.cpu_count() or 1, rows)))
            bad = int((ref != out[:rows]).sum())
            line.update({"parity_checked": rows, "parity_mismatched_symbols": bad, "parity_ok": bad == 0,
                         "parity_against": "C oracle (decoder.rs / decoder_special.rs restated, f32, same operation order), first "
                                           f"{rows} codewords of the timed output"})
            rc = 0 if bad == 0 else 3
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = qary_cpu_baseline(oracle_call, inputs, iters, g, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    dec.close()
    return rc


def qary_cpu_baseline(oracle_call, inputs, iters, g, budget_s):
    """The C oracle (oracle/qary_oracle.c: decoder.rs / decoder_special.rs restated in f32, the reference's operation
    order) on a bounded sample of the same inputs: one thread, then all host threads over codewords."""
    from oracle import pyoracle

    threads = max(1, min(os.cpu_count() or 1, pyoracle.max_threads()))
    n = inputs[0].shape[0]
    with np.errstate(divide="ignore"):
        t0 = time.perf_counter()
        oracle_call(tuple(x[:1] for x in inputs), iters, 1)
        one = time.perf_counter() - t0
        reps = 1
        if one < 0.05:  # (criterion cases: microseconds per call -- repeat for a stable figure)
            reps = int(min(2000, max(3, 0.5 / max(one, 1e-6))))
            t0 = time.perf_counter()
            for _ in range(reps):
                oracle_call(tuple(x[:1] for x in inputs), iters, 1)
            one = (time.perf_counter() - t0) / reps
        sample = int(min(n, max(1, budget_s / max(one, 1e-6) * threads)))
        tile = tuple(np.concatenate([x] * (-(-sample // n)), axis=0)[:sample] for x in inputs) if sample > n else tuple(x[:sample] for x in inputs)
        t0 = time.perf_counter()
        oracle_call(tile, iters, min(threads, sample))
        dt = time.perf_counter() - t0
    return {"value": 2.0 * g.nnz * iters * sample / dt, "unit": "directed symbol-edge message updates/s", "cores": min(threads, sample),
            "kind": "port", "codewords_per_s": sample / dt,
            "sample": f"{sample} codewords of the same inputs, {iters} iterations, {min(threads, sample)} OpenMP threads over codewords "
                      f"({dt:.2f} s); restated CPU path (C), not the reference's Rust binary (no rustc here)",
            "single_thread_value": 2.0 * g.nnz * iters / one, "single_thread_ms_per_call": one * 1e3}


def mc_sweep(args, S, bp, lib, trials, H, N, omega, R, E, probs, iters, method, rank, world, local, coll):
    """BASELINE config 5: `--trials` synthetic hqc.decode() trials sharded over the ranks by
    GLOBAL trial index (results independent of the GPU count), generated, decoded (early
    exit, max_iter 100) and compared on the device in sub-batches of `--mc-batch`; one gather
    of the per-trial success flags at the end.  A "step" is one sub-batch.
    roofline: the sweep's own algorithmic byte rate (16 E bytes per codeword-iteration actually run) and, from the
    same HIP-event machinery as the default workload, the in-situ launch times of the check / variable kernels.
    cpu_baseline: the CPU oracle with early exit on the first trials of the same sweep (same seeds, same inputs)."""
    import torch

    shard = importlib.import_module("sca-ldpc_amd.shard")
    dec = bp.bp_decoder(H, max_iter=iters, bp_method=method, channel_probs=probs)
    a, b = shard.trial_range(args.trials, rank, world)
    for _ in range(args.warmup):  # untimed: workspace allocation, code-object load
        dec.mc_hqc_run(min(args.mc_batch, max(1, b - a)), omega, args.eps, seed=1, first_trial=0, early_exit=True)
    torch.cuda.synchronize()
    coll.barrier()
    t0 = time.perf_counter()
    succ, its = [], []
    for s0 in range(a, b, args.mc_batch):
        r = dec.mc_hqc_run(min(args.mc_batch, b - s0), omega, args.eps, seed=2, first_trial=s0, early_exit=True)
        succ.append(r["success"])
        its.append(r["iters"])
    torch.cuda.synchronize()
    coll.barrier()
    dt = time.perf_counter() - t0
    succ = np.concatenate(succ) if succ else np.zeros(0, np.uint8)
    its = np.concatenate(its) if its else np.zeros(0, np.int32)
    dev = torch.device("cuda", local)
    per_rank_s = None
    if world > 1:
        per_rank_s = coll.gather_scalars(dt)
        dt = max(per_rank_s)
    all_succ = coll.gather(succ, args.trials)
    all_its = coll.gather(its, args.trials)
    rc = 0
    if rank == 0:
        total_iters = float(all_its.astype(np.int64).sum())
        updates = 2.0 * E * total_iters
        nsteps = int(np.ceil((b - a) / args.mc_batch))
        line = {
            "metric": "edge_message_updates_per_s", "value": updates / dt, "unit": "directed edge-message updates/s",
            "n_gpus": world, "steps": nsteps, "warmup": args.warmup,
            "ms_per_step": dt / max(1, nsteps) * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"hqc128 Monte-Carlo sweep, {args.trials} trials, eps={args.eps}, {method}, early exit, "
                                   f"max_iter {iters}, sub-batches of {args.mc_batch}, device-side trial generation"},
            "trials_per_s": args.trials / dt, "decode_success_rate": float(all_succ.mean()),
            "mean_iterations": float(all_its.mean()), "wall_s": dt,
            "success_checksum": shard.success_checksum(all_succ),
            "collective_backend": coll.backend, "collective_note": coll.note or None,
        }
        if per_rank_s:
            line["per_rank_wall_s"] = per_rank_s
            line["rank_wall_s_min_max"] = [min(per_rank_s), max(per_rank_s)]
        # kernel launch times in the sweep's own schedule: one short fixed-iteration decode of 4096 of the sweep's
        # trials leaves a full tile group's message state behind, then the HIP-event series of time_kernels
        r4 = dec.mc_hqc_run(4096, omega, args.eps, seed=2, first_trial=0, early_exit=True, want_inputs=True)
        d_in = torch.from_numpy(r4["msg"]).to(dev)
        d_out = torch.empty((4096, H.n), dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        # (early exit, as the sweep: time_kernels then launches the variable pass in the form the sweep's passes have -- with the
        # decision output every early-exit pass writes; compaction off so that the full tile group's state stays)
        dec.configure(compact_after=0)
        dec.decode_batch_device(d_in.data_ptr(), lib.IN_RECEIVED, 4096, d_out.data_ptr(), max_iter=4, early_exit=True, stream=stream)
        dec.configure(compact_after=-1)
        torch.cuda.synchronize()
        kt = dec.time_kernels(50, stream=stream)
        ms_check = kt["ms_check"] / max(1, kt["launches_check"])
        ms_var = kt["ms_var"] / max(1, kt["launches_var"])
        lanes = kt["lanes"]
        pair = lanes * (8.0 * E * kt["codewords"] + 8.0 * E * kt["codewords_var"]) / ((ms_check + ms_var) * 1e-3) / 1e9
        whole = 16.0 * E * total_iters / dt / 1e9
        # bytes really moved per algorithmic byte by this rule's launch pair: the committed PMC passes of the same kernels on the
        # same graph and launch geometry (the sweep itself cannot be profiled per launch: its schedule is data-dependent).
        # Message form: every message is read and written by both passes, so moved >= algorithmic (x1.02-1.05: records, priors, planes)
        tr = pmc_traffic("hqc128_tanh", kt["codewords"]) if args.pmc != "off" else None
        ratio, ratio_src = 1.0, "none (no committed PMC passes of this geometry): algorithmic bytes, a lower bound for the message form"
        if tr and tr["kernels"].get("k_check_tanh") and tr["kernels"].get("k_var"):
            ratio = (tr["kernels"]["k_check_tanh"]["traffic_bytes"] + tr["kernels"]["k_var"]["traffic_bytes"]) / (16.0 * E * kt["codewords"])
            ratio_src = tr["source"]
        line["roofline"] = {
            "bound": "hbm", "served_by": "infinity-cache (cache-resident tile groups)", "kernel": "k_var" if ms_var >= ms_check else "k_check_tanh",
            "achieved": whole * ratio, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": whole * ratio / HBM_PEAK_GBS, "traffic": None,
            "achieved_is": "the sweep's whole-job byte rate: 16 E algorithmic bytes per codeword-iteration actually run (early exit: "
                           "mean_iterations per trial) x the moved / algorithmic ratio of this rule's launch pair (PMC) / wall time, trial "
                           "generation, convergence tests, compaction and the success compare included",
            "moved_per_algorithmic_byte": ratio, "moved_per_algorithmic_byte_source": ratio_src,
            "algorithmic_GBps": whole, "algorithmic_frac": whole / HBM_PEAK_GBS,
            "steady_state_pair": {"lanes": lanes, "algorithmic_GBps": pair, "moved_GBps": pair * ratio, "frac": pair * ratio / HBM_PEAK_GBS,
                                  "per_launch_us": {"check": ms_check * 1e3, "var": ms_var * 1e3},
                                  "timed_variable_pass_writes_output": bool(kt.get("var_writes_out")),
                                  "what": "the check + variable launch pair of a full tile group in the decode's own two-lane "
                                          "schedule (HIP events, scaldpc_bp_time_kernels), as in the default workload"},
            "sweep_efficiency": whole / pair,  # how much of the steady-state kernel rate the whole sweep keeps
        }
        del d_in, d_out
        if args.parity_rows > 0 or (world == 1 and not args.no_cpu_baseline):
            line.update(mc_cpu_leg(H, probs, r4, E, iters, args, world))
            rc = 0 if line.get("parity_ok", True) else 3
        print(json.dumps(line), flush=True)
    dec.close()
    return rc


def mc_cpu_leg(H, probs, r4, E, iters, args, world):
    """After the timed region: the CPU oracle (f32 tanh rule, EARLY EXIT -- the reference's decode loop) on the first
    trials of the sweep, inputs taken from the device's own trial generator.  Serves as the parity check (iteration
    counts and success-relevant decisions of the same trials) and as cpu_baseline (bounded sample, all host threads)."""
    from oracle import pyoracle

    out = {}
    threads = max(1, min(os.cpu_count() or 1, pyoracle.max_threads()))
    msg = r4["msg"]
    t0 = time.perf_counter()
    one = pyoracle.bp_decode_batch(H, probs, msg[:threads], 1, iters, "tanh_complement", dtype="f32", threads=threads, early_exit=True)
    t_first = time.perf_counter() - t0
    sample = int(min(msg.shape[0], max(threads, args.cpu_seconds / max(t_first, 1e-6) * threads)))
    t0 = time.perf_counter()
    ref = pyoracle.bp_decode_batch(H, probs, msg[:sample], 1, iters, "tanh_complement", dtype="f32", threads=threads, early_exit=True)
    dt = time.perf_counter() - t0
    if args.parity_rows > 0:
        same = int((ref["iters"] == r4["iters"][:sample]).sum())
        out.update({"parity_checked": sample, "parity_same_iteration_count": same,
                    "parity_ok": bool(same >= sample - max(1, sample // 500)),  # a tie of `L <= 0 -> 1` may move one count by one
                    "parity_against": f"oracle f32 tanh_complement with early exit, iteration counts of the first {sample} trials of the sweep"})
    if world == 1 and not args.no_cpu_baseline:
        it_sum = float(ref["iters"].astype(np.int64).sum())
        out["cpu_baseline"] = {
            "value": 2.0 * E * it_sum / dt, "unit": "directed edge-message updates/s", "cores": threads, "kind": "port",
            "trials_per_s": sample / dt, "mean_iterations": it_sum / sample,
            "sample": f"first {sample} trials of the same sweep (inputs from the device's trial generator), f32 tanh rule, early "
                      f"exit, max_iter {iters}, {threads} OpenMP threads over trials ({dt:.1f} s); restated CPU path, not the reference binary",
        }
    return out


def pmc_traffic(workload, swept):
    """Fallback for roofline.traffic when this run cannot take counters itself (the process is being profiled, N > 1, no
    rocprofv3): the per-kernel bytes of an earlier live run of exactly this launch geometry, committed under
    profiles/ by `bench.py --pmc-save` (same format as pmc_live's result).  None when the geometry differs."""
    for rnd in ("r04",):
        path = os.path.join(ROOT, "profiles", rnd, f"pmc_traffic_{workload}.json")
        try:
            d = json.load(open(path))
            if d["codewords_per_launch"] == swept and d.get("kernels"):
                d["source"] = f"profiles/{rnd}/pmc_traffic_{workload}.json (committed rocprofv3 --pmc passes of this launch geometry)"
                return d
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
