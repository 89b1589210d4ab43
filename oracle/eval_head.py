"""TEST INFRASTRUCTURE -- a calibrated `fc` head for the evaluator's native classifiers (`tools/fooling_parity.py`,
`tests/test_gpu_size_parity.py`).  Never imported by the product package.

Why.  The reference scores adversarial clips with Kinetics-400 classifiers on a clip list that was SELECTED so that every model
classifies every clean clip correctly (`/root/reference/utils.py:29`, `reference.py:108-129`: fooling rate = 100 - top-1 against
`gt_label`).  No checkpoint exists offline; a seeded random `fc` knows no labels, so top-1 against `gt_label` is chance level for any
set of clips and "the two sets score alike" says nothing.  What CAN be built offline is the list's defining
property: a head under which every CLEAN clip of the list is classified as its `gt_label` with a stated margin.  The backbone stays the
seeded network; only `fc` is fitted, on the pooled pre-`fc` features of the clean clips (`NativeClassifier.pooled_features`).

How sensitive such a head is to the attack is a choice, and it is made here in the open.  The clips are white noise and the backbone is
random, so the pooled features of the 400 clips are 400 nearly orthogonal directions in a 2048-/2304-dimensional space, and ANY head
that uses all of them (ridge least squares at every lambda from 1e-6 to 10, nearest centroid = lambda -> infinity) keeps every attacked
clip on its own side: measured fooling rate 0 % (I3D-NL 4.5 % at lambda = 10) although the attack moves the features by 0.64 (I3D-NL) /
0.30 (SlowFast) of the spread between clips -- an evaluator as blind as the seeded one.  A head's sensitivity is set by its RANK: in r
whitened dimensions 400 points sit (1/400)^(1/r) of their spread apart, and the same feature shift crosses a boundary for a mid-range
share of the clips when r is small.  So the head is

    logits = W_r Q_r^T  S^-1 V^T (f - mean)  + b          V S: PCA of the clean features (399 directions), S^-1: whitening,
                                                          Q_r: the first r columns of a seeded random rotation (no direction preferred)

with (W_r, b) trained on the 400 clean clips by a multiclass hinge loss until EVERY clip holds its label with margin >= 1 over every
other class (least-squares start, full-batch Adam, float64; converges in 10^2..10^3 steps because 400 points in general position in
r >= 8 dimensions are all vertices of their hull).  `rank` is the one knob.  The rule for it, stated on the HIP set alone (nothing about
the ORACLE's set enters): of RANK_SWEEP, the LARGEST rank -- the least sensitive head, which also amplifies the fp32-level noise between
two correct runs of the attack least -- under which the attack still fools at least 10 % of the list.  On round 6's data (400 clips):
I3D-NL 32 (11.75 %; 48 gives 3.5 %), SlowFast 8 (11.75 %; 12 gives 0.25 %) = DEFAULT_RANK; `rank_by_rule` re-derives it from a run's
own features and `tools/fooling_parity.py` records whether it agrees.  (The first n = 400 run of the round used rank 16 for the I3D-NL --
the probe's mid-range pick -- and is kept in the sweep: fooling rates 38.25 / 42.00 %, 31 + 46 discordant clips, McNemar p = 0.11.)
"""
import numpy as np

DEFAULT_RANK = {"i3d_resnet50": 32, "slowfast_resnet50": 8}
RANK_SWEEP = (8, 12, 16, 24, 32, 48)
MIN_FOOLING_FOR_RANK = 10.0
MARGIN = 1.0


def fit_head(clean_feats, labels, rank, num_classes=400, seed=0, max_iters=6000):
    """clean_feats (n, C) float, labels (n,) -> (fc_weight (num_classes, C) float32, fc_bias (num_classes,) float32, info)."""
    import torch
    F = np.asarray(clean_feats, np.float64)
    lab = np.asarray(labels, np.int64)
    n = F.shape[0]
    mu = F.mean(0)
    X = F - mu
    _, S, Vt = np.linalg.svd(X, full_matrices=False)
    d = min(n - 1, int((S > S[0] * 1e-10).sum()))
    Pw = Vt[:d] / S[:d, None] * np.sqrt(n)                        # whitening of the clean span: (d, C)
    Q, _ = np.linalg.qr(np.random.default_rng(seed).standard_normal((d, d)))
    P = Q[:, :rank].T @ Pw                                        # (rank, C)
    Z = torch.tensor(X @ P.T)
    y = torch.tensor(lab)
    Y = torch.zeros(n, num_classes, dtype=torch.float64)
    Y[torch.arange(n), y] = 1
    W = torch.linalg.lstsq(Z, Y).solution.T.clone().requires_grad_(True)
    b = torch.zeros(num_classes, dtype=torch.float64, requires_grad=True)
    opt = torch.optim.Adam([W, b], lr=0.01)
    it, worst = 0, float("inf")
    for it in range(max_iters):
        lg = Z @ W.T + b
        viol = lg - lg[torch.arange(n), y][:, None] + MARGIN
        viol[torch.arange(n), y] = 0
        v = torch.clamp(viol, min=0)
        worst = float(v.max().detach())
        if worst == 0:
            break
        loss = v.max(1).values.sum() + 1e-4 * (W ** 2).sum()
        opt.zero_grad()
        loss.backward()
        opt.step()
    Wf = W.detach().numpy() @ P                                   # (num_classes, C)
    bf = b.detach().numpy() - Wf @ mu
    Wf32, bf32 = Wf.astype(np.float32), bf.astype(np.float32)
    lg = F.astype(np.float32).astype(np.float64) @ Wf32.astype(np.float64).T + bf32.astype(np.float64)
    top2 = np.sort(lg, 1)[:, -2:]
    info = {"rank": int(rank), "iterations": int(it), "hinge_violation_left": worst, "clean_top1": float((lg.argmax(1) == lab).mean() * 100),
            "min_clean_margin": float((top2[:, 1] - top2[:, 0])[lg.argmax(1) == lab].min()), "n": int(n), "feature_dim": int(F.shape[1]),
            "centered_feature_norm_mean": float(np.linalg.norm(X, axis=1).mean())}
    return Wf32, bf32, info


def mcnemar_exact(b, c):
    """Two-sided exact McNemar test on the discordant pairs (b: only set A right, c: only set B right): the probability, under
    exchangeable sets, of a split at least this uneven."""
    from math import comb
    nd = b + c
    if nd == 0:
        return 1.0
    k = min(b, c)
    p = sum(comb(nd, i) for i in range(k + 1)) / 2.0 ** nd * 2.0
    return min(1.0, p)


def own_margin(logits, labels):
    """Own-label logit minus the best other per row: > 0 <=> classified as its label."""
    lg = np.asarray(logits, np.float64)
    idx = np.arange(len(labels))
    own = lg[idx, labels]
    other = lg.copy()
    other[idx, labels] = -np.inf
    return own - other.max(1)


def rank_sweep(clean_feats, labels, sets, ranks=RANK_SWEEP, num_classes=400):
    """The head at every rank of the sweep, applied to pooled features of each set in `sets` ({name: (n, C)}; rows = the first n rows of
    the list): fooling rate per set and, for every pair of sets, the discordant split and the exact McNemar p.  float64 on the host from
    the features the device pooled (the device's `fc` accumulates in double as well)."""
    lab = np.asarray(labels, np.int64)
    out = []
    for r in ranks:
        W, b, info = fit_head(clean_feats, lab, r, num_classes=num_classes)
        fooled = {}
        for name, F in sets.items():
            n = len(F)
            fooled[name] = own_margin(np.asarray(F, np.float64) @ W.astype(np.float64).T + b.astype(np.float64), lab[:n]) <= 0
        row = {"rank": int(r), "clean_top1": info["clean_top1"], "fooling_rate": {k: round(100.0 * float(v.mean()), 4) for k, v in fooled.items()}, "pairs": {}}
        names = list(fooled)
        for i in range(len(names)):
            for j in range(i + 1, len(names)):
                a, c = fooled[names[i]], fooled[names[j]]
                n = min(len(a), len(c))
                oa, oc = int((a[:n] & ~c[:n]).sum()), int((c[:n] & ~a[:n]).sum())
                row["pairs"][f"{names[i]}|{names[j]}"] = {"n": n, "only_first_fooled": oa, "only_second_fooled": oc, "mcnemar_exact_p": round(mcnemar_exact(oa, oc), 4)}
        out.append(row)
    return out


def rank_by_rule(sweep, on="hip"):
    """The largest rank of a `rank_sweep` whose fooling rate on set `on` is at least MIN_FOOLING_FOR_RANK (None if there is none)."""
    ok = [row["rank"] for row in sweep if row["fooling_rate"].get(on, 0.0) >= MIN_FOOLING_FOR_RANK]
    return max(ok) if ok else None
