"""The hand-written FFT convolution (csrc/fftnative.hip): `convolve_fft_torch` (jolideco/utils/torch.py:347-370) as three
launches -- rows (two image rows per complex transform: upper half real, lower half imaginary), columns (FFT, kernel
spectrum, inverse FFT in one kernel), rows^-1 + epilogue -- on complex FFTs of length 2^a * {1, 3, 9} held in LDS.
Against float64 and against the rocFFT path it replaces on those sizes."""
import numpy as np
import pytest
import torch

from conftest import rel_linf

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [((64, 128), (5, 7)), ((200, 328), (17, 17)), ((130, 516), (33, 17)), ((256, 256), (65, 65)), ((96, 132), (32, 33)),
         ((512, 1024), (129, 129)), ((72, 2048), (9, 9)), ((1024, 64), (3, 41)),
         ((2600, 64), (9, 9)), ((4096, 128), (17, 17)),  # (column lengths 2048 and 2304 -- two waves per column)
         # column schedules with the odd radix in the fused middle pass, generic kernel: 288 = 8 * 4 * 9, 144 = 16 * 9, 72 = 8 * 9;
         # compile-time schedules 1024 = 16 * 8 * 8 and 1152 = 16 * 8 * 9 on a narrow image
         ((520, 64), (9, 9)), ((264, 64), (9, 5)), ((120, 68), (9, 9)), ((1800, 64), (9, 9)), ((2048, 64), (17, 9)),
         # rows of 4096 pixels -> row transforms of length 4608 = 8 * 8 * 8 * 9, the compile-time schedule bench.py's c6 and a
         # 4096^2 FFT fit time (round-4 verdict): on a short image with a wide general PSF, and at full size (rows 4608,
         # columns 2304 = 16 * 16 * 9 on two waves)
         ((256, 4096), (65, 65)), ((4096, 4096), (17, 17)), ((4096, 4096), (130, 130)),
         # any image size (round 5): an odd number of rows (the lower half of the row pairs is one row short), widths that
         # are not multiples of 4 (rows read and written at 4-byte alignment, the last piece element by element)
         ((65, 131), (5, 7)), ((201, 330), (17, 17)), ((127, 513), (33, 17)), ((64, 67), (9, 9)), ((2047, 2050), (33, 33))]


def _psf(kshape, seed):
    rs = np.random.RandomState(seed)
    psf = rs.uniform(0.0, 1.0, size=kshape) ** 3  # a general (full-rank) kernel
    return (psf / psf.sum()).astype(np.float32)


@pytest.mark.parametrize("shape,kshape", CASES, ids=[f"{s[0]}x{s[1]}_k{k[0]}x{k[1]}" for s, k in CASES])
def test_native_fft_convolution_and_adjoint_match_float64_and_rocfft(jd_option, shape, kshape):
    from scipy.signal import fftconvolve

    from jolideco_amd.ops import ConvPlan

    H, W = shape
    rs = np.random.RandomState(H + W)
    image = rs.gamma(2.0, size=shape).astype(np.float32)
    image[rs.randint(0, H, 20), rs.randint(0, W, 20)] += 500.0  # point sources: a large dynamic range
    scale = rs.uniform(0.5, 1.5, size=shape).astype(np.float32)
    psf = _psf(kshape, 7)
    oy, ox = (kshape[0] - 1) // 2, (kshape[1] - 1) // 2
    full = fftconvolve(image.astype(np.float64) * scale, psf.astype(np.float64), mode="full")
    ref = full[oy:oy + H, ox:ox + W]
    full_adj = fftconvolve(image.astype(np.float64), psf[::-1, ::-1].astype(np.float64), mode="full")
    ay, ax = kshape[0] - 1 - oy, kshape[1] - 1 - ox
    ref_adj = full_adj[ay:ay + H, ax:ax + W] * scale
    out = {}
    for native in (1, 0):
        jd_option("JD_FFT_NATIVE", native)
        plan = ConvPlan(H, W, kshape[0], kshape[1], DEV, method="fft")
        assert plan.native_fft == bool(native)
        khat = plan.psf_spectrum(torch.from_numpy(psf).to(DEV))
        x, s = torch.from_numpy(image).to(DEV), torch.from_numpy(scale).to(DEV)
        conv = plan.conv_same(x, s, khat)
        base = torch.full(shape, 0.5, device=DEV)
        adj = plan.conv_same_adjoint(x, s, khat, grad_image=base.clone(), accumulate=True)
        adj0 = plan.conv_same_adjoint(x, s, khat)
        torch.cuda.synchronize()
        out[native] = (conv.cpu().numpy(), adj.cpu().numpy() - 0.5, adj0.cpu().numpy())
        plan.close()
    for name, got, want in (("conv", out[1][0], ref), ("adjoint (accumulated)", out[1][1], ref_adj), ("adjoint", out[1][2], ref_adj)):
        assert rel_linf(got, want) < 2e-6, name
    assert rel_linf(out[1][0], out[0][0]) < 3e-6 and rel_linf(out[1][2], out[0][2]) < 3e-6
    # the two are transposes of each other: <conv(x), y> = <x, adjoint(y)>
    # (checked against float64 above on both sides; here directly, with the scale image on the input side of both)
    y = rs.uniform(size=shape).astype(np.float32)
    jd_option("JD_FFT_NATIVE", 1)
    plan = ConvPlan(H, W, kshape[0], kshape[1], DEV, method="fft")
    assert plan.native_fft
    khat = plan.psf_spectrum(torch.from_numpy(psf).to(DEV))
    adj_y = plan.conv_same_adjoint(torch.from_numpy(y).to(DEV), torch.from_numpy(scale).to(DEV), khat).cpu().numpy()
    plan.close()
    lhs = float((out[1][0].astype(np.float64) * y).sum())
    rhs = float((image.astype(np.float64) * adj_y).sum())
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs)


@pytest.mark.parametrize("shape", [(96, 132), (97, 131)], ids=["96x132", "97x131-odd-rows-ragged-width"])
def test_native_fft_fit_matches_the_oracle(monkeypatch, shape):
    """A joint fit through the native FFT path (general PSFs of two sizes, conv_method = fft) against the oracle."""
    from jolideco_amd import MAPDeconvolver, SpatialFluxComponent, UniformPrior
    from jolideco_amd.data import synthetic_observations
    from oracle import cpu_ref

    monkeypatch.setenv("JOLIDECO_CONV_METHOD", "fft")
    datasets, _, flux_init = synthetic_observations(shape=shape, n_obs=8, seed=2)
    comp = SpatialFluxComponent.from_numpy(flux=flux_init, prior=UniformPrior())
    deco = MAPDeconvolver(n_epochs=5, display_progress=False, device=DEV, fit_mode="joint")
    session = deco.session(datasets, components=comp)
    plans = {m.plan for models in session.total_loss.poisson_loss.npred_models_all for m in models.values()}
    # (17x17 and 33x33 PSFs: embedded to one 33x33 plan -- an FFT convolution costs the same for every PSF size, and one
    # plan is what the batched joint step of the FFT path needs; the oracle convolves every PSF at its own size)
    assert all(p.native_fft for p in plans) and len(plans) == 1 and session.batch_joint
    res = deco.run(datasets, components=comp)
    final, _ = cpu_ref.map_fit_joint(datasets, {"flux": flux_init}, {"flux": cpu_ref.UniformPriorRef()}, n_epochs=5)
    assert rel_linf(res.flux_total, final["flux"]) < 1e-5
