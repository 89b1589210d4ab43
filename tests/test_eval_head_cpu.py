"""CPU: the calibrated evaluator head (`oracle/eval_head.py`, test infrastructure of the fooling-rate parity measurement): on features
shaped like the native classifiers' (n near-orthogonal directions around a large common mean) the fitted `fc` classifies every
training clip as its label with the stated margin AFTER the cast to float32, is deterministic, and its rank sets how far a feature
has to move before the label changes; the exact McNemar test behaves."""
import numpy as np

from oracle import eval_head


def _features(n=60, C=256, seed=0):
    rng = np.random.default_rng(seed)
    return (1000.0 + rng.standard_normal((1, C)) * 50 + rng.standard_normal((n, C)) * 5).astype(np.float32)


def test_every_clean_clip_holds_its_label_with_margin_and_rank_sets_sensitivity():
    n, classes = 60, 80
    F = _features(n)
    labels = np.random.default_rng(1).permutation(classes)[:n]
    fooled = {}
    for rank in (6, 12, 40):
        W, b, info = eval_head.fit_head(F, labels, rank, num_classes=classes)
        assert W.dtype == np.float32 and W.shape == (classes, F.shape[1]) and b.shape == (classes,)
        lg = F.astype(np.float64) @ W.astype(np.float64).T + b
        assert (lg.argmax(1) == labels).all() and info["clean_top1"] == 100.0 and info["min_clean_margin"] >= 0.99, info
        assert np.linalg.matrix_rank(W.astype(np.float64), tol=1e-6 * np.abs(W).max()) <= rank
        shift = np.random.default_rng(2).standard_normal(F.shape) * 2.5            # half the spread between "clips"
        fooled[rank] = float(((F + shift).astype(np.float64) @ W.astype(np.float64).T + b).argmax(1).__ne__(labels).mean())
        W2, b2, _ = eval_head.fit_head(F, labels, rank, num_classes=classes)
        assert np.array_equal(W, W2) and np.array_equal(b, b2)
    assert fooled[6] > fooled[40], fooled                # fewer dimensions: closer boundaries
    assert fooled[40] <= 0.05, fooled


def test_mcnemar_exact():
    assert eval_head.mcnemar_exact(0, 0) == 1.0
    assert abs(eval_head.mcnemar_exact(3, 5) - 0.7265625) < 1e-12
    assert eval_head.mcnemar_exact(0, 10) < 0.01 and eval_head.mcnemar_exact(10, 0) == eval_head.mcnemar_exact(0, 10)
    assert eval_head.mcnemar_exact(25, 25) == 1.0
