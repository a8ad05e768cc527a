"""Regenerate the "## r03" section of profiles/README.md from the data files of profiles/r03/ (bench lines, rocprofv3
kernel statistics, PMC traffic), so that every number in the tables can be traced to a committed file:
    python tools/profiles_readme.py            (run from the repository root after tools/profile_round.sh r03)"""
import csv
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
R = ROOT / "profiles" / "r03"


def stats(cfg):
    with open(R / f"{cfg}_n1_kernel_stats.csv") as fh:
        return {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(fh)}


def find(d, key):
    for k, v in d.items():
        if key in k:
            return v
    return float("nan")


def pmc(cfg, key):
    """(MB read, MB written) per launch: reads = 2 x FETCH_SIZE (gfx950 correction), writes = WRITE_SIZE"""
    fetch = write = float("nan")
    with open(R / "pmc_hbm_traffic.csv") as fh:
        for r in csv.DictReader(fh):
            if r["config"] == cfg and key in r["kernel"]:
                if r["counter"] == "FETCH_SIZE":
                    fetch = float(r["avg_per_launch_KB"])
                else:
                    write = float(r["avg_per_launch_KB"])
    return 2 * fetch * 1024 / 1e6, write * 1024 / 1e6


def line(name):
    return json.loads((R / f"{name}.json").read_text().strip().splitlines()[-1])


def side(d, key, field="value"):
    v = d.get(key)
    return round(v[field]) if isinstance(v, dict) and v.get(field) is not None else None


def main():
    c3, c4, c5 = stats("c3"), stats("c4"), stats("c5")
    b3, b3d, b2, b4, b5 = (line(n) for n in ("c3_n1_bench", "c3_n1_bench_driver_flags", "c2_n1_bench", "c4_n1_bench", "c5_n1_bench"))
    fw, ad = find(c3, "walk_kernel<4, 2, true, true, 0, 8>"), find(c3, "walk_kernel<4, 3, false, false, 6, 8>")
    sc, ex, ga, stg = find(c3, "gmm_screen_kernel"), find(c3, "gmm_exact_kernel"), find(c3, "gmm_gather_tile"), find(c3, "gmm_stage")
    sca, cnt, bs = find(c3, "bucket_scatter"), find(c3, "bucket_count"), find(c3, "bucket_binscan")
    be, dn = find(c3, "gmm_best"), find(c3, "gmm_fwd_kernel")
    f4, a4 = find(c4, "walk_kernel<4, 2, true"), find(c4, "walk_kernel<4, 2, false")
    m5, a5 = find(c5, "walk_multi_kernel"), find(c5, "walk_kernel<2, 2, false, false, 6, 16>")
    mpx = 4.194304  # 2048^2 in units of 1e6 pixels
    text = f"""## r03 (round 3)

Made by `tools/profile_round.sh r03` in ONE `gpurun` call on the last build of the round (board clock {round(b3['clock_mhz'])} MHz by
`jd_clock_probe`); this section is generated from the files by `tools/profiles_readme.py`.

| file | what |
|---|---|
| `r03/c3_n1_bench.json` | `python bench.py` (200 steps × 9 regions after 20 warm-up + ≥ 0.3 s settle): **{b3['value']:.0f} it/s, {b3['ms_per_step']:.4f} ms/step** (regions {b3['ms_per_step_min']:.4f}–{b3['ms_per_step_max']:.4f}); forward + Poisson launch {b3['roofline_poisson']['avg_launch_ms'] * 1e3:.1f} µs by event pairs = {b3['roofline_poisson']['frac'] * 100:.1f} %, batched adjoint {b3['roofline_conv']['avg_launch_ms'] * 1e3:.1f} µs = {b3['roofline_conv']['frac'] * 100:.1f} %; `general_psf` {side(b3, 'general_psf')}, `fft_psf` {side(b3, 'fft_psf')}, `dense_fp32_gmm` {side(b3, 'dense_fp32_gmm')} it/s, `sequential_mode` {side(b3, 'sequential_mode', 'epochs_per_s')} epochs/s; CPU oracle {b3['cpu_baseline']['value']:.2f} it/s on 16 cores |
| `r03/c3_n1_bench_driver_flags.json` | the same box, `--steps 20 --warmup 5` (the driver's flags): {b3d['value']:.0f} it/s, {b3d['ms_per_step']:.4f} ms/step — {abs(1 - b3d['value'] / b3['value']) * 100:.1f} % from the default run |
| `r03/c3_n1_kernel_stats.csv` | `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-general-psf`: the per-kernel table below |
| `r03/c2_n1_bench.json`, `c4_n1_bench.json`, `c5_n1_bench.json` | `--config c2|c4|c5 --steps 100 --warmup 10`: c2 {b2['value']:.0f} it/s ({b2['ms_per_step']:.3f} ms), c4 {b4['value']:.0f} it/s ({b4['ms_per_step']:.3f} ms), c5 {b5['value']:.0f} it/s ({b5['ms_per_step']:.3f} ms) |
| `r03/c4_n1_kernel_stats.csv`, `r03/c5_n1_kernel_stats.csv` | the rocprofv3 summaries of the c4 (4096², 1 observation) and c5 (2048², 16 observations, 2 components) runs |
| `r03/pmc_hbm_traffic.csv` (+ `.commit`) | FETCH_SIZE / WRITE_SIZE per kernel, separate passes; rows `c3h`, `c4h`; HBM reads = 2 × FETCH_SIZE (`MI355X_MICROARCH.md`) |
| `r03/exact_stamps.txt` | s_memtime phase stamps of `gmm_exact_kernel` (diagnostic build), four variants |
| `r03/lse_screen.txt` | `marginalize=True`: value + gradient per call through the logsumexp screen and through the dense one-pass kernel, two mixtures × three images at 2048² (0.48–1.97 ms against 2.87 ms) and 4096² (1.65 against 11.4 ms) |

Fractions of the roofs, recomputable from `r03/c3_n1_kernel_stats.csv` (AverageNs) and the algorithmic bytes of DESIGN.md §3:

| kernel | rocprofv3 avg | achieved | PMC traffic per launch |
|---|---|---|---|
| `gmm_screen_kernel<2, false, true, false>` | {sc:.1f} µs | 0.2054 TFLOP fp16 / {sc:.1f} µs = {0.2054 / sc * 1e6:.0f} TFLOP/s = **{0.2054 / sc * 1e6 / 2516.6 * 100:.1f} %** of 2516.6 | {pmc('c3h', 'gmm_screen')[0]:.0f} MB read + {pmc('c3h', 'gmm_screen')[1]:.0f} MB written |
| `walk_kernel<4, 2, true, true, 0, 8>` (8 forward models + Poisson passes) | **{fw:.1f} µs** | 671 MB / {fw:.1f} µs = {671 / fw:.2f} TB/s = **{671 / fw / 8 * 100:.1f} %** of 8 TB/s (111 µs = 76 % in a loop of likelihood steps: DESIGN.md §7d) | {pmc('c3h', 'walk_kernel<4, 2, true')[0]:.0f} MB read + {pmc('c3h', 'walk_kernel<4, 2, true')[1]:.0f} MB written = {sum(pmc('c3h', 'walk_kernel<4, 2, true')) / 671:.2f} × algorithmic (7 of the 8 flux reads are cache hits) |
| `walk_kernel<4, 3, false, false, 6, 8>` (8 adjoints, one launch, one wave per dataset) | **{ad:.1f} µs** | 302 MB / {ad:.1f} µs = {302 / ad:.2f} TB/s = **{302 / ad / 8 * 100:.1f} %** | {pmc('c3h', 'walk_kernel<4, 3')[0]:.0f} MB read + {pmc('c3h', 'walk_kernel<4, 3')[1]:.0f} MB written = {sum(pmc('c3h', 'walk_kernel<4, 3')) / 302:.1f} × algorithmic (halo rows of g and exposure) |
| `gmm_exact_kernel<true>` | {ex:.1f} µs | fabric traffic, not flop: ≈ 6.4 TB/s of cache-line fetches (DESIGN.md §7d) | {pmc('c3h', 'gmm_exact')[0]:.0f} MB read + {pmc('c3h', 'gmm_exact')[1]:.0f} MB written |
| `gmm_gather_tile_kernel` (+ optimizer step) | {ga:.1f} µs | {sum(pmc('c3h', 'gmm_gather')):.0f} MB / {ga:.1f} µs = {sum(pmc('c3h', 'gmm_gather')) / ga:.1f} TB/s of measured traffic | {pmc('c3h', 'gmm_gather')[0]:.0f} MB read + {pmc('c3h', 'gmm_gather')[1]:.0f} MB written |
| `gmm_stage_kernel` | {stg:.1f} µs | | {pmc('c3h', 'gmm_stage')[0]:.0f} MB read + {pmc('c3h', 'gmm_stage')[1]:.0f} MB written |
| record sort: scatter {sca:.1f}, count {cnt:.1f}, binscan {bs:.1f} | {sca + cnt + bs:.1f} µs | | |
| `gmm_best_kernel` {be:.1f}, gated dense kernel {dn:.1f} | {be + dn:.1f} µs | a dependent launch costs ≥ 4.5 µs | |

c4 (`r03/c4_n1_kernel_stats.csv`): screen {find(c4, 'gmm_screen'):.0f} µs, exact {find(c4, 'gmm_exact'):.0f}, gather + optimizer step {find(c4, 'gmm_gather'):.0f}, forward + Poisson
{f4:.1f} µs (335 MB → {335 / f4:.2f} TB/s = {335 / f4 / 8 * 100:.1f} %; 69 µs = 61 % in a loop of likelihood steps: DESIGN.md §7d "cold and warm"),
adjoint {a4:.1f} µs (268 MB → {268 / a4:.1f} TB/s = {268 / a4 / 8 * 100:.1f} %).
c5 (`r03/c5_n1_kernel_stats.csv`): `walk_multi_kernel<4, 2>` {m5:.0f} µs (16 × 2 forward models + 16 Poisson passes: 32 B per
(pixel, dataset) → {32 * 16 * mpx / m5:.2f} TB/s = {32 * 16 * mpx / m5 / 8 * 100:.0f} %), `walk_kernel<2, 2, false, false, 6, 16>` {a5:.0f} µs (the 32 adjoints of both
components in one launch: 2 × 136 B per pixel → {2 * 136 * mpx / a5:.2f} TB/s = {2 * 136 * mpx / a5 / 8 * 100:.0f} %), prior as in c3.

"""
    readme = ROOT / "profiles" / "README.md"
    s = readme.read_text()
    i0, i1 = s.index("## r03 (round 3)"), s.index("## r02 (round 2)")
    readme.write_text(s[:i0] + text + s[i1:])
    print(text)


if __name__ == "__main__":
    main()
