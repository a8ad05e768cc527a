"""Regenerate the "## r04" section of profiles/README.md from the data files of profiles/r04/ (bench lines, rocprofv3
kernel statistics, PMC traffic, SQ counters), so that every number in the tables can be traced to a committed file:
    python tools/profiles_readme_r04.py        (run from the repository root after tools/profile_round.sh r04)"""
import csv
import json
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
R = ROOT / "profiles" / "r04"
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


def sq(kernel):
    """{counter: value per launch} of one kernel from sq_counters.txt (tools/pmc_summary.py)"""
    out, on = {}, False
    for raw in (R / "sq_counters.txt").read_text().splitlines():
        if not raw.startswith("   "):
            on = kernel in raw
        elif on:
            name, value = raw.split()[:2]
            out[name] = float(value)
    return out


def main():
    c3, c4, c5, f3, c6 = stats("c3"), stats("c4"), stats("c5"), stats("c3fft"), stats("c6")
    b3, b3d, b2, b4, b5, b6 = (line(n) for n in ("c3_n1_bench", "c3_n1_bench_driver_flags", "c2_n1_bench", "c4_n1_bench", "c5_n1_bench", "c6_n1_bench"))
    fw = find(c3, "walk_mixed_kernel")
    ad17, ad33 = find(c3, "walk_kernel<17, 4, 3, false, false, 6"), find(c3, "walk_kernel<33, 2, 2, false, false, 2")
    sc, ex, ga, stg = find(c3, "gmm_screen_kernel"), find(c3, "gmm_exact_kernel"), find(c3, "gmm_gather_tile"), find(c3, "gmm_stage")
    sca, cnt, bs = find(c3, "bucket_scatter"), find(c3, "bucket_count"), find(c3, "bucket_binscan")
    be, dn = find(c3, "gmm_best"), find(c3, "gmm_fwd_kernel")
    f4, a4 = find(c4, "walk_kernel<17, 4, 2, true"), find(c4, "walk_kernel<17, 4, 2, false")
    m5 = find(c5, "walk_multi_kernel")
    # per OBSERVATION (the batched joint step of the FFT path covers the 8 observations of a step in one launch per phase)
    steps3 = max(find(f3, "gmm_screen_kernel", 1), 1)
    per_obs = lambda key: sum(v[0] * v[1] for k, v in f3.items() if key in k) / (steps3 * 8)  # noqa: E731
    rows, cols2, mid, inv = per_obs("fftn_rows_fwd"), per_obs("fftn_cols"), per_obs("fftn_rows_poisson"), per_obs("fftn_rows_inv")
    cols = cols2 / 2
    launches3 = sum(v[1] for k, v in f3.items() if "fftn_" in k and "spectrum" not in k) / steps3
    fw_mb = (16 * 8 + 4) * MPX
    rf, rw = pmc("c3", "walk_mixed_kernel")
    s = sq("gmm_screen_kernel")
    busy = s.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan")) / 4 / max(s.get("SQ_WAVE_CYCLES", float("nan")), 1)
    valu = s.get("SQ_ACTIVE_INST_VALU", float("nan")) / max(s.get("SQ_WAVE_CYCLES", float("nan")), 1)
    wait = s.get("SQ_WAIT_INST_ANY", float("nan")) / max(s.get("SQ_WAVE_CYCLES", float("nan")), 1)
    c6k = b6["kernel_ms_per_step"]
    waitany = s.get("SQ_WAIT_ANY", float("nan")) / max(s.get("SQ_WAVE_CYCLES", float("nan")), 1)
    coexec = s.get("SQ_VALU_MFMA_COEXEC_CYCLES", float("nan")) / max(s.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan")), 1)
    per_wc = s.get("SQ_INSTS_VALU", float("nan")) / (2040 * 128)
    n6 = max(c6.get(next(k for k in c6 if 'gmm_screen' in k))[1], 1)  # steps in the profiled c6 run
    c6rows = sorted(((v[0] * v[1] / n6, k, v[0], v[1] / n6) for k, v in c6.items() if v[1] >= n6 and 'elementwise_kernel' not in k), reverse=True)[:14]
    c6table = '\n'.join(f'| `{k[:90]}` | {per:.1f} | {avg:.1f} | {tot / 1e3:.2f} |' for tot, k, avg, per in c6rows)
    fft = b3.get("fft_psf", {})
    text = f"""## r04 (round 4)

Made by `tools/profile_round.sh r04` in ONE `gpurun` call on the last build of the round (board clock {round(b3['clock_mhz'])} MHz by
`jd_clock_probe`); this section is generated from the files by `tools/profiles_readme_r04.py`.  The c3 workload is now
SURVEY section 8(d)'s: 6 x 17x17 + 2 x 33x33 PSFs (`config.psf_shapes` in the bench line).

| file | what |
|---|---|
| `r04/c3_n1_bench.json` | `python bench.py` (200 steps x 9 regions after 20 warm-up + >= 0.3 s settle): **{b3['value']:.0f} it/s, {b3['ms_per_step']:.4f} ms/step** (regions {b3['ms_per_step_min']:.4f}-{b3['ms_per_step_max']:.4f}), host enqueue {b3['host_enqueue_ms_per_step']:.3f} ms/step; `general_psf` {b3['general_psf']['value']:.0f} (17x17 MFMA Toeplitz, 33x33 native FFT; everything MFMA Toeplitz, `direct_psf`: {b3.get('direct_psf', {}).get('value', float('nan')):.0f}), `fft_psf` {fft.get('value', float('nan')):.0f} it/s ({fft.get('ms_per_step', float('nan')):.3f} ms; round 3: 375 through rocFFT; start of round 4: 690), `dense_fp32_gmm` {b3['dense_fp32_gmm']['value']:.0f} it/s, `sequential_mode` {b3['sequential_mode']['epochs_per_s']:.0f} epochs/s, `c6_chandra_like` {b3['c6_chandra_like']['value']:.0f} it/s; CPU oracle {b3['cpu_baseline']['value']:.2f} it/s on {b3['cpu_baseline']['cores']} cores |
| `r04/c3_n1_bench_driver_flags.json` | the same box, `--steps 20 --warmup 5` (the driver's flags): {b3d['value']:.0f} it/s, {b3d['ms_per_step']:.4f} ms/step |
| `r04/c3_n1_kernel_stats.csv` | `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-general-psf`: the per-kernel table below |
| `r04/c2_n1_bench.json`, `c4_n1_bench.json`, `c5_n1_bench.json` | `--config c2|c4|c5 --steps 100 --warmup 10`: c2 {b2['value']:.0f} it/s ({b2['ms_per_step']:.3f} ms), c4 {b4['value']:.0f} it/s ({b4['ms_per_step']:.3f} ms), c5 {b5['value']:.0f} it/s ({b5['ms_per_step']:.3f} ms; 16 observations: 6 x 17x17 + 10 x 33x33 "extended" PSFs) |
| `r04/c6_n1_bench.json` | `--config c6`: 2048^2 counts grid, up-sampling x2, 8 calibrated observations, general 65x65 PSFs (the reference's Chandra example at the benchmark's size): **{b6['value']:.0f} it/s, {b6['ms_per_step']:.2f} ms/step** (first measurement of the round: 119 it/s, 8.41 ms), host enqueue {b6['host_enqueue_ms_per_step']:.2f} ms/step; per step: native FFT rows (+ shift) {c6k.get('fft_r2c', 0):.2f} + columns {c6k.get('cmul', 0):.2f} + rows^-1 {c6k.get('fft_c2r', 0):.2f} ms, rows^-1 x 2 + pool + Poisson + rows of g {c6k.get('poisson_fused', 0):.2f}, prior {c6k.get('gmm_fwd', 0):.2f} + gather {c6k.get('gmm_gather', 0):.2f} ms (the transposed-shift kernel carries no timer) |
| `r04/c6_n1_kernel_stats.csv` | rocprofv3 summary of `--config c6`: the table at the end of this section |
| `r04/c4_n1_kernel_stats.csv`, `r04/c5_n1_kernel_stats.csv`, `r04/c3fft_n1_kernel_stats.csv` | rocprofv3 summaries of c4, c5 and of c3 through the FFT path (`JOLIDECO_CONV_METHOD=fft`: the native FFT convolution) |
| `r04/pmc_hbm_traffic.csv` (+ `.commit`) | FETCH_SIZE / WRITE_SIZE per kernel, separate passes; rows `c3`, `c4`, `c3fft`; HBM reads = 2 x FETCH_SIZE (`MI355X_MICROARCH.md`) |
| `r04/sq_counters.txt` | SQ counters of the c3 step (five passes of four counters), per launch: the screen kernel's matrix / vector / wait shares below |
| `r04/conv_method_crossover.txt` | `tools/conv_bench.py`: MFMA Toeplitz against native FFT convolution at 1024^2 / 2048^2 / 4096^2, 17-33 taps (the rule of the method "auto") |
| `r04/ab_*.txt` | the A/B runs of the round (one process or one call each): packed row pass, mixed-launch balance, 33-tap adjoint tiling, native FFT column kernel with parts switched off; second session: `ab_fft8_*` (FFT kernels of the round's start against the last build: c3 through the FFT path and c6), `ab_fft9_*` (columns per block), `ab_ilv_*` (dataset order of the walk forward launch), `ab_scr_*` (screen kernel builds), `ab_fb_*` / `ab_fb4096_*` (batched against per-dataset FFT joint step), `ab_cb1024_*` (the same for calibrated + up-sampled datasets, 512^2-2048^2 flux pixels), `ab_blk_*` (blocked layout of the spectrum arrays), `ab_raw_*` (exact stage on staged fp32 patches), `ab_c4w_*` (tile shapes of the forward launch at 4096^2) |

Fractions of the roofs, recomputable from `r04/c3_n1_kernel_stats.csv` (AverageNs) and the algorithmic bytes of DESIGN.md section 3:

| kernel | rocprofv3 avg | achieved | PMC traffic per launch |
|---|---|---|---|
| `gmm_screen_kernel<2, false, true, false>` | {sc:.1f} us | 0.2054 TFLOP fp16 / {sc:.1f} us = {0.2054 / sc * 1e6:.0f} TFLOP/s = **{0.2054 / sc * 1e6 / 2516.6 * 100:.1f} %** of 2516.6; SQ: matrix pipe busy {busy * 100:.0f} % of the wave cycles, vector instructions active {valu * 100:.0f} %, parked on s_waitcnt {waitany * 100:.0f} %, waiting on an instruction's operands {wait * 100:.0f} %; {per_wc:.0f} vector + matrix instructions per wave and component; vector and matrix pipes busy together in {coexec * 100:.0f} % of the matrix pipe's busy cycles | {pmc('c3', 'gmm_screen')[0]:.0f} MB read + {pmc('c3', 'gmm_screen')[1]:.0f} MB written |
| `walk_mixed_kernel<4, 2>` (8 forward models + Poisson passes: 6 in the 17-tap frame at 4 columns per lane, 2 in the 33-tap frame at 2) | **{fw:.1f} us** | {fw_mb:.1f} MB (flux counted once) / {fw:.1f} us = {fw_mb / fw:.2f} TB/s = **{fw_mb / fw / 8 * 100:.1f} %** of 8 TB/s | {rf:.0f} MB read + {rw:.0f} MB written = {(rf + rw) / fw_mb:.2f} x algorithmic |
| `walk_kernel<17, 4, 3, false, false, 6, 8>` (adjoints of the 6 17-tap observations) | {ad17:.1f} us | {(8 * 6 + 8) * MPX:.0f} MB / {ad17:.1f} us = {(8 * 6 + 8) * MPX / ad17:.2f} TB/s = {(8 * 6 + 8) * MPX / ad17 / 8 * 100:.1f} % | {pmc('c3', 'walk_kernel<17, 4, 3')[0]:.0f} MB read + {pmc('c3', 'walk_kernel<17, 4, 3')[1]:.0f} MB written |
| `walk_kernel<33, 2, 2, false, false, 2, 8>` (adjoints of the 2 33-tap observations, accumulated) | {ad33:.1f} us | {(8 * 2 + 8) * MPX:.0f} MB / {ad33:.1f} us = {(8 * 2 + 8) * MPX / ad33:.2f} TB/s = {(8 * 2 + 8) * MPX / ad33 / 8 * 100:.1f} % (vector-instruction bound: 36 + 32 warm-up rows per tile) | {pmc('c3', 'walk_kernel<33, 2, 2, false, false, 2')[0]:.0f} MB read + {pmc('c3', 'walk_kernel<33, 2, 2, false, false, 2')[1]:.0f} MB written |
| `gmm_exact_kernel<true>` | {ex:.1f} us | | {pmc('c3', 'gmm_exact')[0]:.0f} MB read + {pmc('c3', 'gmm_exact')[1]:.0f} MB written |
| `gmm_gather_tile_kernel` (+ optimizer step) | {ga:.1f} us | {sum(pmc('c3', 'gmm_gather')):.0f} MB / {ga:.1f} us = {sum(pmc('c3', 'gmm_gather')) / ga:.1f} TB/s of measured traffic | {pmc('c3', 'gmm_gather')[0]:.0f} MB read + {pmc('c3', 'gmm_gather')[1]:.0f} MB written |
| `gmm_stage_kernel` | {stg:.1f} us | | |
| record sort: scatter {sca:.1f}, count {cnt:.1f}, binscan {bs:.1f} | {sca + cnt + bs:.1f} us | | |
| `gmm_best_kernel` {be:.1f}, gated dense kernel {dn:.1f} | {be + dn:.1f} us | | |

c3 through the native FFT convolution (`r04/c3fft_n1_kernel_stats.csv`; the batched joint step: {launches3:.0f} launches per step, each over
the 8 observations; per OBSERVATION): rows {rows:.1f} us (52 MB -> {52.4 / rows:.2f} TB/s), columns {cols:.1f} us (61 MB -> {61.3 / cols:.2f} TB/s; PMC per
launch of 8: {sum(pmc('c3fft', 'fftn_cols')):.0f} MB), rows^-1 + Poisson + rows of g {mid:.1f} us, rows^-1 + adjoint epilogue {inv:.1f} us
(55 MB -> {54.8 / inv:.2f} TB/s); {rows + 2 * cols + mid + inv:.0f} us per observation (start of the round: 132).
c4 (`r04/c4_n1_kernel_stats.csv`): screen {find(c4, 'gmm_screen'):.0f} us, exact {find(c4, 'gmm_exact'):.0f}, gather + optimizer step {find(c4, 'gmm_gather'):.0f}, forward + Poisson
{f4:.1f} us (335 MB -> {335 / f4:.2f} TB/s = {335 / f4 / 8 * 100:.1f} %; PMC {sum(pmc('c4', 'walk_kernel<17, 4, 2, true')):.0f} MB = {sum(pmc('c4', 'walk_kernel<17, 4, 2, true')) / 335.5:.2f} x), adjoint {a4:.1f} us (268 MB -> {268 / a4:.1f} TB/s = {268 / a4 / 8 * 100:.1f} %).
c5 (`r04/c5_n1_kernel_stats.csv`): `walk_multi_kernel<2, 2, 33>` {m5:.0f} us (16 x 2 forward models + 16 Poisson passes, the "extended" component of
10 observations in the 33-tap frame: 32 B per (pixel, dataset) -> {32 * 16 * MPX / m5:.2f} TB/s), five adjoint launches
({', '.join(f'{v[0]:.0f} us x {v[1] // max(c5[next(iter(c5))][1] // c5[next(iter(c5))][1], 1)}' for k, v in c5.items() if 'walk_kernel<' in k and 'false, false' in k)} calls in the profiled run), prior as in c3.

c6 (`r04/c6_n1_kernel_stats.csv`; per step = 8 calibrated observations at 4096^2 flux pixels + the prior):

| kernel | launches per step | avg us | ms per step |
|---|---|---|---|
{c6table}

"""
    text = re.sub(r" x (\d+) calls", r" x \1 calls", text)
    readme = ROOT / "profiles" / "README.md"
    s = readme.read_text()
    if "## r04 (round 4)" in s:
        i0, i1 = s.index("## r04 (round 4)"), s.index("## r03 (round 3)")
        s = s[:i0] + text + s[i1:]
    else:
        i1 = s.index("## r03 (round 3)")
        s = s[:i1] + text + s[i1:]
    readme.write_text(s)
    print(text)


if __name__ == "__main__":
    main()
