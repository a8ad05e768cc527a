"""Regenerate the "## r05" section of profiles/README.md from the data files of profiles/r05/ (bench lines, rocprofv3
kernel statistics, PMC traffic, rank-share tables), so that every number in the tables can be traced to a committed file:
    python tools/profiles_readme_r05.py        (run from the repository root after tools/profile_round.sh r05)"""
import csv
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
R = ROOT / "profiles" / "r05"
MPX = 4.194304  # 2048^2 in units of 1e6 pixels


def stats(label):
    with open(R / f"{label}_n1_kernel_stats.csv") as fh:
        return {r["Name"].replace("(anonymous namespace)::", ""): (float(r["AverageNs"]) / 1e3, int(r["Calls"])) for r in csv.DictReader(fh)}


def find(d, key, which=0):
    for k, v in d.items():
        if key in k:
            return v[which]
    return float("nan")


def pmc(config, key):
    """(MB read, MB written) per launch: reads = 2 x FETCH_SIZE (gfx950 correction), writes = WRITE_SIZE"""
    fetch = write = float("nan")
    with open(R / "pmc_hbm_traffic.csv") as fh:
        for r in csv.DictReader(fh):
            if r["config"] == config and key in r["kernel"]:
                if r["counter"] == "FETCH_SIZE":
                    fetch = float(r["avg_per_launch_KB"])
                else:
                    write = float(r["avg_per_launch_KB"])
    return 2 * fetch * 1024 / 1e6, write * 1024 / 1e6


def line(name):
    return json.loads((R / f"{name}.json").read_text().strip().splitlines()[-1])


def shard(name):
    """{N: (max us, min us, max / min)} of a tools/shard_table.py output"""
    out = {}
    for raw in (R / name).read_text().splitlines():
        if raw.startswith("{"):
            for n, w in json.loads(raw)["worlds"].items():
                out[int(n)] = (1e3 * w["max_ms"], 1e3 * w["min_ms"], w["max_over_min"])
    return out


def main():
    c3, c4, c6, f3 = stats("c3"), stats("c4"), stats("c6"), stats("c3fft")
    b1, b2, b3, b3d, b4, b5, b6, be = (line(n) for n in ("c1_n1_bench", "c2_n1_bench", "c3_n1_bench", "c3_n1_bench_driver_flags", "c4_n1_bench",
                                                         "c5_n1_bench", "c6_n1_bench", "e0102_bench"))
    br = line("c3_rank2_of_8_bench")
    shared = (f"; side run `shared_psf` (one PSF per dataset for both components: evaluated as their sum) {b5['shared_psf']['value']:.0f} it/s"
              if "shared_psf" in b5 else "")
    fw = find(c3, "walk_mixed_kernel")
    ad17, ad33 = find(c3, "walk_kernel<17, 4, 3, false, false, 6"), find(c3, "walk_kernel<33, 2, 2, false, false, 2")
    sc, ex, ga, stg = find(c3, "gmm_screen_kernel<2, false, true, false, false>"), find(c3, "gmm_exact_kernel"), find(c3, "gmm_gather_tile"), find(c3, "gmm_stage")
    sca, cnt, bs, best, dn = (find(c3, k) for k in ("bucket_scatter", "bucket_count", "bucket_binscan", "gmm_best", "gmm_fwd_kernel"))
    f4 = find(c4, "walk_kernel<17, 4, 2, true")
    fw_mb = (16 * 8 + 4) * MPX
    rf, rw = pmc("c3", "walk_mixed_kernel")
    roof, r6 = b3["roofline"], b6["roofline_c6"]
    n6 = max(find(c6, "gmm_screen_kernel", 1), 1)  # steps in the profiled c6 run
    c6rows = sorted(((v[0] * v[1] / n6, k, v[0], v[1] / n6) for k, v in c6.items() if v[1] >= n6 and "elementwise_kernel" not in k), reverse=True)[:14]
    c6table = "\n".join(f"| `{k[:96]}` | {per:.1f} | {avg:.1f} | {tot / 1e3:.2f} |" for tot, k, avg, per in c6rows)
    c6l = "\n".join(
        f"| {name} | {v['what']} | {v['bytes_per_observation'] / 1e6:.0f} MB | {1e3 * v['ms_per_observation']:.1f} us | {v['achieved'] / 1e3:.2f} TB/s = **{v['frac']:.2f}** |"
        for name, v in r6["launches"].items())
    pm6 = {k: pmc("c6", k) for k in ("fftn_rows_fwd_kernel<8", "fftn_cols_kernel<128", "fftn_rows_pooled_kernel<2, 8", "fftn_rows_inv_kernel<true, 8", "shift_bwd4_kernel")}
    pm6t = "; ".join(f"`{k.split('<')[0]}` {a:.0f} + {w:.0f} MB" for k, (a, w) in pm6.items())
    s3c, s3r, s5c, s5r = shard("shard_c3_cost.txt"), shard("shard_c3_rr.txt"), shard("shard_c5_cost.txt"), shard("shard_c5_rr.txt")
    row = lambda t, n: f"{t[n][0]:.0f} / {t[n][1]:.0f} us ({t[n][2]:.2f})" if n in t else ""  # noqa: E731
    steps3 = max(find(f3, "gmm_screen_kernel", 1), 1)
    per_obs = lambda key: sum(v[0] * v[1] for k, v in f3.items() if key in k) / (steps3 * 8)  # noqa: E731
    rows, cols, mid, inv = per_obs("fftn_rows_fwd"), per_obs("fftn_cols") / 2, per_obs("fftn_rows_poisson"), per_obs("fftn_rows_inv")
    small = "\n".join("    " + raw for raw in (R / "small_fits.txt").read_text().splitlines() if raw.startswith("flux grid"))
    text = f"""## r05 (round 5)

Kernel statistics, PMC traffic and SQ counters: `tools/profile_round.sh r05` in one `gpurun` call (by-value epochs on one
stream under the tracer: a kernel's own duration).  Bench lines and kernel statistics: `tools/gpu/r5_final2.sh` (third session: `tools/gpu/r5_profile_lite.sh`); rank-share tables, small fits, SQ counters, RCCL probe: `tools/gpu/r5_g20.sh` (second session; those kernels are unchanged)
on the last build of the round (board clock {round(b3['clock_mhz'])} MHz by `jd_clock_probe`).  This section is generated from the
files by `tools/profiles_readme_r05.py`.

| file | what |
|---|---|
| `r05/c3_n1_bench.json` | `python bench.py`: **{b3['value']:.0f} it/s, {b3['ms_per_step']:.4f} ms/step** (regions {b3['ms_per_step_min']:.4f}-{b3['ms_per_step_max']:.4f}); policy: {b3['graph_policy']}; host enqueue {b3['host_enqueue_ms_per_step']:.3f} ms/step; `graph_replay` (every epoch replayed) {b3['graph_replay']['value']:.0f} it/s at {b3['graph_replay']['host_enqueue_ms_per_step']:.3f} ms of host time per step; `fft_psf` {b3.get('fft_psf', {}).get('value', float('nan')):.0f} it/s; `odd_size_fft` (2047 x 2050 image through the FFT path) native {b3['odd_size_fft']['native']['value']:.0f} against rocFFT {b3['odd_size_fft']['rocfft']['value']:.0f} it/s; `dense_fp32_gmm` {b3['dense_fp32_gmm']['value']:.0f} it/s; CPU oracle {b3['cpu_baseline']['value']:.2f} it/s on {b3['cpu_baseline']['cores']} cores |
| `r05/c3_n1_bench_driver_flags.json` | the same box, `--steps 20 --warmup 5` (the driver's flags): {b3d['value']:.0f} it/s, {b3d['ms_per_step']:.4f} ms/step |
| `r05/c1_n1_bench.json` | `--config c1` (BASELINE configs[0]: 128^2, one observation, uniform prior, the reference's sequential loop): **{b1['value']:.0f} epochs/s** ({1e3 * b1['ms_per_step']:.1f} us per epoch; {b1['graph_policy']}); CPU oracle, the full workload: {b1['cpu_baseline']['value']:.0f} epochs/s |
| `r05/c2_n1_bench.json`, `c4_n1_bench.json`, `c5_n1_bench.json` | c2 {b2['value']:.0f} it/s ({b2['ms_per_step']:.3f} ms; {b2['graph_policy'].split(';')[-1].strip(' )')}), c4 {b4['value']:.0f} it/s ({b4['ms_per_step']:.3f} ms), c5 {b5['value']:.0f} it/s ({b5['ms_per_step']:.3f} ms; {b5['graph_policy'].split(';')[-1].strip(' )')}){shared} |
| `r05/c6_n1_bench.json` | `--config c6` (2048^2 counts grid, up-sampling x2, 8 calibrated observations, general 65x65 PSFs, K = 128): **{b6['value']:.0f} it/s, {b6['ms_per_step']:.2f} ms/step** (round 4: 187); `roofline_c6`: {r6['frac']:.2f} of 8 TB/s over the six launches of an observation (table below) |
| `r05/e0102_bench.json` | `--config e0102`: the reference's only published runtime ("about 30 min on an M1 cpu"), 24 observations, 128x128 PSFs, x2 up-sampling, calibrations, 250 sequential epochs on an ASSUMED 256^2 counts grid, through `MAPDeconvolver.run()`: **{be['value']:.2f} s** ({be['ms_per_step']:.3f} ms per optimizer step; {be['graph_policy'].split(';')[0]}) |
| `r05/c3_rank2_of_8_bench.json` | `bench.py --shard-of 8 --rank 2`: the share of rank 2 of an 8-rank job, no transport: {1e3 * br['ms_per_step']:.0f} us per step |
| `r05/shard_c3_cost.txt`, `shard_c3_rr.txt`, `shard_c5_cost.txt`, `shard_c5_rr.txt` | `tools/shard_table.py`: every rank's share of an N-rank joint step, one rank after the other in one process (table below); `*_before_band_fix.txt`: the same before the band kernels lost their scratch copy |
| `r05/small_fits.txt` | `tools/gpu/small_fits.py`: 8 calibrated, up-sampled observations at 512^2 / 1024^2 / 2048^2 flux pixels, by value (rounds 1-4 loop) / planned / replayed / default policy (below); `small_fits_overlap.txt`, `small_fits_one_stream.txt`, `small_fits_fetch_kernel.txt`, `small_fits_copy_upload.txt`: earlier builds of the round |
| `r05/*_n1_kernel_stats.csv` | rocprofv3 summaries of c3, c4, c5, c6 and of c3 through the FFT path (`c3fft`) |
| `r05/pmc_hbm_traffic.csv` (+ `.commit`) | FETCH_SIZE / WRITE_SIZE per kernel, separate passes; rows `c3`, `c4`, `c6`, `c3fft`; HBM reads = 2 x FETCH_SIZE (`MI355X_MICROARCH.md`) |
| `r05/sq_counters.txt` | SQ counters of the c3 step, per launch |
| `r05/rccl_probe.json` | `tools/rccl_probe.py`: a one-rank RCCL group running the calls of a sharded step: host time of the three `torch.distributed` calls, on-stream floor of the two collectives |
| `r05/conv_method_crossover.txt` | MFMA Toeplitz against native FFT convolution by PSF size (the rule of the method "auto") |
| `r05/ab_*.txt` | A/B runs of the round: `ab_fft1_*` (packed complex arithmetic, 576-thread rows, linear LDS indices), `ab_fft3_*` / `ab_fft4_*` (batching, columns per block, pooled prefetch), `ab_c6_transposed_shift_rows_outside.txt`, `ab_prior_overlap.txt` (the prior's first phase beside the likelihood), `ab_c3_tile_heights_with_overlap.txt`; third session: `ab_prior_schedule.txt` (side-stream priority, the likelihood fenced behind the screen launch), `ab_shared_operator.txt` (flux components on one PSF evaluated as their sum), `ab_gmm_backend_loads.txt` and `ab_fft_global_loads.txt` (batched loads, global-memory accessors) |

c3, per kernel (`r05/c3_n1_kernel_stats.csv`, AverageNs; algorithmic bytes / flop of DESIGN.md section 3):

| kernel | rocprofv3 avg | achieved | PMC traffic per launch |
|---|---|---|---|
| `gmm_screen_kernel<2, false, true, false, false>` | {sc:.1f} us | 0.2054 TFLOP fp16 / {sc:.1f} us = {0.2054 / sc * 1e6:.0f} TFLOP/s = {0.2054 / sc * 1e6 / 2516.6 * 100:.1f} % of 2516.6 at the nominal 2400 MHz; `bench.py` live: {roof['frac']:.3f}, and **{roof.get('frac_at_in_kernel_clock') or float('nan'):.2f} at the {roof.get('in_kernel_clock_mhz') or float('nan'):.0f} MHz its own blocks measure** (`s_memtime` around {roof.get('in_kernel_clock_blocks_sampled')} blocks) | {pmc('c3', 'gmm_screen_kernel<2, false, true, false, false>')[0]:.0f} MB read + {pmc('c3', 'gmm_screen_kernel<2, false, true, false, false>')[1]:.0f} MB written |
| `walk_mixed_kernel<4, 2>` (8 forward models + Poisson passes) | {fw:.1f} us | {fw_mb:.1f} MB / {fw:.1f} us = {fw_mb / fw:.2f} TB/s = **{fw_mb / fw / 8 * 100:.1f} %** of 8 TB/s | {rf:.0f} MB read + {rw:.0f} MB written = {(rf + rw) / fw_mb:.2f} x algorithmic |
| `walk_kernel<17, 4, 3, false, false, 6, 8>` (adjoints of the 6 17-tap observations) | {ad17:.1f} us | {(8 * 6 + 8) * MPX:.0f} MB -> {(8 * 6 + 8) * MPX / ad17:.2f} TB/s = {(8 * 6 + 8) * MPX / ad17 / 8 * 100:.1f} % | {pmc('c3', 'walk_kernel<17, 4, 3')[0]:.0f} + {pmc('c3', 'walk_kernel<17, 4, 3')[1]:.0f} MB |
| `walk_kernel<33, 2, 2, false, false, 2, 8>` (adjoints of the 2 33-tap observations) | {ad33:.1f} us | {(8 * 2 + 8) * MPX:.0f} MB -> {(8 * 2 + 8) * MPX / ad33:.2f} TB/s = {(8 * 2 + 8) * MPX / ad33 / 8 * 100:.1f} % | {pmc('c3', 'walk_kernel<33, 2, 2, false, false, 2')[0]:.0f} + {pmc('c3', 'walk_kernel<33, 2, 2, false, false, 2')[1]:.0f} MB |
| `gmm_exact_kernel<true>` | {ex:.1f} us | | {pmc('c3', 'gmm_exact')[0]:.0f} + {pmc('c3', 'gmm_exact')[1]:.0f} MB |
| `gmm_gather_tile_kernel` (+ optimizer step) | {ga:.1f} us | | {pmc('c3', 'gmm_gather')[0]:.0f} + {pmc('c3', 'gmm_gather')[1]:.0f} MB |
| `gmm_stage_kernel` {stg:.1f}, record sort {sca:.1f} + {cnt:.1f} + {bs:.1f}, `gmm_best_kernel` {best:.1f}, gated dense kernel {dn:.1f} | {stg + sca + cnt + bs + best + dn:.1f} us | | |

The step is {1e3 * b3['ms_per_step']:.0f} us against {fw + ad17 + ad33 + sc + ex + ga + stg + sca + cnt + bs + best + dn:.0f} us of kernels: the prior's first phase (stage .. arg-max, {sc + ex + stg + sca + cnt + bs + best + dn:.0f} us) runs on
a second stream beside the likelihood's three launches ({fw + ad17 + ad33:.0f} us) -- `ab_prior_overlap.txt`.  These kernels are the round-4
kernels; the third session of round 5 batched the loads of the GMM back end (record sort, arg-max pass, gather: `ab_gmm_backend_loads.txt`).
c4 (`r05/c4_n1_kernel_stats.csv`): forward + Poisson {f4:.1f} us (335 MB -> {335 / f4:.2f} TB/s = {335 / f4 / 8 * 100:.1f} %; PMC {sum(pmc('c4', 'walk_kernel<17, 4, 2, true')):.0f} MB).
c3 through the native FFT convolution (`r05/c3fft_n1_kernel_stats.csv`, per observation): rows {rows:.1f} us, columns {cols:.1f} us, rows^-1 + Poisson + rows
of g {mid:.1f} us, rows^-1 + adjoint epilogue {inv:.1f} us; {rows + 2 * cols + mid + inv:.0f} us per observation (round 4: 76).

c6, the six launches of an observation (`bench.py` `roofline_c6`: library timers around the launches over all 8 observations,
divided by 8; padded FFT grid {r6['padded_fft_grid'][0]} row pairs x {r6['padded_fft_grid'][1]} columns):

| timer | what | algorithmic bytes | time | of 8 TB/s |
|---|---|---|---|---|
{c6l}
| all six | | {r6['bytes_per_observation'] / 1e6:.0f} MB | {1e3 * r6['ms_per_observation']:.0f} us | **{r6['frac']:.2f}** (round 4: 0.31) |

PMC traffic per launch over the 8 observations (reads + writes): {pm6t}.

c6 per kernel (`r05/c6_n1_kernel_stats.csv`; per step = 8 calibrated observations at 4096^2 flux pixels + the prior):

| kernel | launches per step | avg us | ms per step |
|---|---|---|---|
{c6table}

Rank shares of a joint step (`shard_*.txt`; max / min over the ranks, in brackets max / min):

| config | N | cost-aware placement | round-robin |
|---|---|---|---|
| c3 | 2 | {row(s3c, 2)} | (the same table) |
| c3 | 4 | {row(s3c, 4)} | {row(s3r, 4)} |
| c3 | 8 | {row(s3c, 8)} | {row(s3r, 8)} |
| c5 | 4 | {row(s5c, 4)} | |
| c5 | 8 | {row(s5c, 8)} | {row(s5r, 8)} |

Small calibrated fits (`small_fits.txt`; wall time per joint step over 200 steps / host time per step into an empty queue):

{small}

"""
    readme = ROOT / "profiles" / "README.md"
    s = readme.read_text()
    i1 = s.index("## r04 (round 4)")
    if "## r05 (round 5)" in s:
        s = s[: s.index("## r05 (round 5)")] + text + s[i1:]
    else:
        s = s[:i1] + text + s[i1:]
    readme.write_text(s)
    print(text)


if __name__ == "__main__":
    main()
