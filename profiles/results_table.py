#!/usr/bin/env python3
"""The results table of DESIGN.md section 5 from the bench lines committed under profiles/<round>/bench_*.json.
usage: python3 profiles/results_table.py r04 > table.md"""
import glob
import json
import os
import sys


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
    here = os.path.dirname(os.path.abspath(__file__))
    R = {}
    for f in sorted(glob.glob(os.path.join(here, rnd, "bench_*.json"))):
        name = os.path.basename(f)[6:-5]
        if "gpus2" in name:
            continue
        R[name] = json.loads(open(f).read().strip().splitlines()[-1])

    def hq(n):
        d = R[n]
        r = d["roofline"]
        return (f"| `{n}` | **{d['value']:.3g} updates/s, {d['ms_per_step']:.1f} ms/step** ({d['codewords_per_s'] / 1e3:.1f} k codewords/s) | "
                f"`{r['kernel']}`; pair {r['pair']['us']:.1f} µs × {r['lanes']} lane{'s' if r['lanes'] > 1 else ''}, "
                f"{r['pair']['bytes_per_pair_and_lane'] / 1e6:.0f} MB moved per pair and lane: **{r['achieved'] / 1e3:.2f} TB/s = {r['frac']:.3f}** "
                f"(whole step {r['whole_step']['frac']:.3f}); algorithmic {r['algorithmic_frac']:.2f}; HBM-streaming regime {r['hbm_frac']:.2f}; "
                f"in-place stream of the group's size {r['cache_ceiling_GBps'] / 1e3:.2f} TB/s | "
                f"{d['cpu_baseline']['value']:.2g} updates/s on {d['cpu_baseline']['cores']} threads (f32 port); "
                f"{d['cpu_baseline']['reference_form_single_thread']['value']:.2g} on one core in the float64 ratio form |")

    rows = [hq(n) for n in ("hqc128_minsum", "hqc192_minsum", "hqc128_tanh", "hqc256_tanh") if n in R]
    if "hqc128_mc" in R:
        d = R["hqc128_mc"]
        r = d["roofline"]
        rows.append(f"| `hqc128_mc` (config 5) | {d['value']:.3g} updates/s, **{d['trials_per_s'] / 1e3:.0f} k trials/s**, 1 M trials in {d['wall_s']:.2f} s, "
                    f"mean {d['mean_iterations']:.2f} iterations | whole sweep **{r['frac']:.2f}** moved ({r.get('algorithmic_frac', r['frac']):.2f} algorithmic; message "
                    f"form: ×{r.get('moved_per_algorithmic_byte', 1.0):.3f}); steady-state pair {r['steady_state_pair']['frac']:.2f}; `sweep_efficiency` {r['sweep_efficiency']:.2f} | "
                    f"oracle with early exit on the sweep's first {d['parity_checked']} trials: {d['cpu_baseline']['value']:.2g} updates/s, "
                    f"{d['cpu_baseline']['trials_per_s']:.0f} trials/s on {d['cpu_baseline']['cores']} threads; iteration counts equal on "
                    f"{d['parity_same_iteration_count']} of {d['parity_checked']} |")

    def qx(d):
        r = d["roofline"]
        ex = r.get("executed", {})
        ro = r.get("reference_op", {})
        return r.get("frac"), ro.get("frac_of_peak"), ex

    if "qary_config4" in R:
        d = R["qary_config4"]
        fr, ro, ex = qx(d)
        rows.append(f"| `qary_config4` (config 4) | {d['value']:.3g} symbol-edge updates/s, **{d['ms_per_step']:.2f} ms** per 1024-codeword call "
                    f"({d['codewords_per_s'] / 1e6:.2f} M codewords/s; host arrays in / out {d['host_buffers']['ms_per_step']:.2f} ms) | `bound: valu`, "
                    f"`{d['roofline']['kernel']}` {d['kernel_ms']['check_per_launch'] * 1e3:.1f} µs per launch: **executed {fr:.2f}** of the VALU peak "
                    f"({ex['instructions_per_reference_op']:.2f} lane-instructions per reference operation: the min-plus recursion does not enumerate; the reference's own "
                    f"operation count over the same time would read {ro:.2f}) | C port of `decoder.rs`: {d['cpu_baseline']['value']:.2g} updates/s on "
                    f"{d['cpu_baseline']['cores']} threads, {d['cpu_baseline']['single_thread_ms_per_call']:.1f} ms per call on one core |")
    ks = [R.get(f"kyber_sw6_b{b}") for b in ("256", "64", "1")]
    if all(ks):
        fr = [qx(k)[0] for k in ks]
        ro = [qx(k)[1] for k in ks]
        rows.append(f"| `kyber_sw6` batch 256 / 64 / 1 | **{ks[0]['ms_per_step']:.2f} / {ks[1]['ms_per_step']:.2f} / {ks[2]['ms_per_step']:.2f} ms** per call | "
                    f"{' / '.join('`' + k['roofline']['kernel'] + '`' for k in ks)} **executed {fr[0]:.2f} / {fr[1]:.2f} / {fr[2]:.2f}** of the VALU peak "
                    f"(the reference's enumeration would be {ro[0]:.2f} / {ro[1]:.2f} / {ro[2]:.2f} of it in the same time: the min-plus recursion does not enumerate) | {ks[0]['cpu_baseline']['value']:.2g} updates/s on {ks[0]['cpu_baseline']['cores']} threads; one core "
                    f"{ks[2]['cpu_baseline']['single_thread_ms_per_call']:.0f} ms per codeword |")
    c = [R.get("criterion_small"), R.get("criterion_medium")]
    if all(c):
        rows.append(f"| `criterion_small` / `criterion_medium` | **{c[0]['ms_per_step']:.2f} / {c[1]['ms_per_step']:.2f} ms** per `min_sum` call at batch 1 | "
                    f"`bound: launch` ({c[0]['roofline']['launches_per_call']} launches per call); no VALU / HBM fraction claimed | one host core: "
                    f"{c[0]['cpu_baseline']['single_thread_ms_per_call']:.3f} / {c[1]['cpu_baseline']['single_thread_ms_per_call']:.2f} ms per call |")
    print(f"**Results, one MI355X, round {rnd[1:].lstrip('0')}** (`python bench.py --workload …`; `profiles/{rnd}/bench_*.json`; every line `parity_ok`, "
          "`self_check` all true):\n")
    print("| workload | `value` / time | `roofline` | `cpu_baseline` (GPU box host, \"port\") |\n|---|---|---|---|")
    print("\n".join(rows))


if __name__ == "__main__":
    main()
