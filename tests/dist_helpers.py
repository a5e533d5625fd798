"""Two-sample chi-square comparison of categorical outcomes (the device's draws against the reference's law)."""
import numpy as np
from scipy import stats


def chi2_two_sample(a, b, n_bins=None):
    """a, b: integer outcome codes of two samples.  -> (statistic, dof, p) of the homogeneity test on the bins that
    occur in either sample."""
    a, b = np.asarray(a).ravel(), np.asarray(b).ravel()
    n_bins = int(max(a.max(), b.max())) + 1 if n_bins is None else n_bins
    ca = np.bincount(a, minlength=n_bins).astype(np.float64)
    cb = np.bincount(b, minlength=n_bins).astype(np.float64)
    keep = (ca + cb) > 0
    ca, cb = ca[keep], cb[keep]
    na, nb = ca.sum(), cb.sum()
    # general form for unequal sample sizes
    stat = float((((ca * np.sqrt(nb / na) - cb * np.sqrt(na / nb)) ** 2) / (ca + cb)).sum())
    dof = int(keep.sum()) - 1
    return stat, dof, float(stats.chi2.sf(stat, dof))


def assert_same_law(a, b, what, p_min=1e-4, n_bins=None):
    stat, dof, p = chi2_two_sample(a, b, n_bins)
    assert p > p_min, "%s: chi2 %.1f on %d dof, p = %.2e -- the two samples do not follow one law" % (what, stat, dof, p)
    return p


def assert_different_law(a, b, what, p_max=1e-9, n_bins=None):
    stat, dof, p = chi2_two_sample(a, b, n_bins)
    assert p < p_max, "%s: chi2 %.1f on %d dof, p = %.2e -- the test has no power here" % (what, stat, dof, p)
