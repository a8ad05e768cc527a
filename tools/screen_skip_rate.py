"""How often could the screen kernel skip half of its work?  CPU estimate (numpy / torch-CPU on the host-side GMM
constants of jolideco_amd, tuning only): the
last 32 whitened coordinates of a patch (y_j, j >= 32: 4 of the 6 MFMA blocks, half of the squares) already give the
upper bound c_k - q_2 / 2 on its log-likelihood; if that is below the running lower bound L for all 64 patches of a
tile pair, the first coordinate block (2 MFMAs + 16 squares per tile + the bound arithmetic) is not needed.
Round 1, 256^2 crop of the bench scene, K = 128, components in popularity order, rigorous fp16 error bound:
skip rate 41 % on the initial (gamma noise) flux, 54 % on the smooth truth image; with the FINAL L 51 % / 70 %;
the FIRST 32 coordinates reject almost nothing (3-5 % of the groups).  See DESIGN.md section 8."""
import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from jolideco_amd.data import synthetic_observations, synthetic_gmm
from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta
torch.set_num_threads(8)
shape=(256,256)
datasets, truth, flux_init = synthetic_observations(shape=shape, n_obs=1, seed=0)
means, covs, weights = synthetic_gmm(128, 64, seed=0)
gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))
for label, img in (("init(gamma30)", flux_init.astype(np.float32)), ("truth", truth.astype(np.float32))):
    x = torch.from_numpy(img)[None,None]
    patches = torch.nn.functional.unfold(x, kernel_size=8, stride=4)[0].T  # (Np, 64)
    patches = patches - patches.mean(dim=1, keepdim=True)
    P = torch.from_numpy(gmm.precisions_cholesky_numpy.astype(np.float32))  # (K, 64, 64)
    mP = torch.from_numpy(gmm.means_precisions_cholesky_numpy.astype(np.float32))  # (K,64)
    w = torch.from_numpy(gmm.pixel_weights_numpy.astype(np.float32))
    Np = patches.shape[0]
    y = torch.einsum('ni,kij->nkj', patches, P) - mP[None]
    wv = w.reshape(1,1,64)
    ysq = (y*y) * wv
    q = ysq.sum(-1)               # (Np,K)
    q1 = ysq[..., :32].sum(-1)
    const = torch.from_numpy((gmm.log_det_cholesky_numpy + gmm.log_weights_numpy).astype(np.float32))[None]  # up to a common constant
    l = const - 0.5*q
    ub1 = const - 0.5*q1
    L = l.max(dim=1, keepdim=True).values
    rej = ub1 < L - 1e-3*np.abs(L)     # per (patch,k) rejection by the half product (optimistic L = final)
    print(label, "Np", Np, "per-pair reject rate", float(rej.float().mean()))
    for group in (32, 64):
        ng = Np // group
        r = rej[:ng*group].reshape(ng, group, -1).all(dim=1)
        print("   group", group, "all-reject rate", float(r.float().mean()))
    # also the other half (last 32 outputs)
    q2 = ysq[..., 32:].sum(-1); ub2 = const - 0.5*q2; rej2 = ub2 < L - 1e-3*np.abs(L)
    print("   last-half per-pair reject", float(rej2.float().mean()), "group64", float(rej2[: (Np//64)*64].reshape(Np//64,64,-1).all(dim=1).float().mean()))

print("---- running-L estimate (popularity order, 64-patch groups)")
for label, img in (("init(gamma30)", flux_init.astype(np.float32)), ("truth", truth.astype(np.float32))):
    x = torch.from_numpy(img)[None,None]
    patches = torch.nn.functional.unfold(x, kernel_size=8, stride=4)[0].T
    patches = patches - patches.mean(dim=1, keepdim=True)
    y = torch.einsum('ni,kij->nkj', patches, P) - mP[None]
    ysq = (y*y) * wv
    q = ysq.sum(-1); q2 = ysq[..., 32:].sum(-1)
    l = (const - 0.5*q).numpy(); ub2 = (const - 0.5*q2).numpy()
    xn = patches.norm(dim=1).numpy()
    efro = 1e-3 * torch.linalg.norm(P.reshape(P.shape[0], -1), dim=1).numpy()
    e = xn[:, None] * efro[None, :]
    Bnd = np.sqrt(q.numpy()) * e + 0.5 * e * e
    ub2r = (const.numpy() - 0.5 * np.maximum(np.sqrt(q2.numpy()) - e, 0.0) ** 2)   # rigorous partial upper bound
    win = l.argmax(1)
    order = np.argsort(-np.bincount(win, minlength=l.shape[1]), kind="stable")
    Np = l.shape[0]; ng = Np // 64
    skipped = 0
    for gi in range(ng):
        sl = slice(gi*64, gi*64+64)
        L = np.full(64, -np.inf)
        for k in order:
            if np.all(ub2r[sl, k] < L):
                skipped += 1
                continue
            L = np.maximum(L, l[sl, k] - Bnd[sl, k])
    print(label, "skip rate of (64-patch group, component) pairs:", skipped / (ng * l.shape[1]))
