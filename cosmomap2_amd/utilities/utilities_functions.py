"""
Synthetic-problem generators and small helpers with the reference's names
(utilities/utilities_functions.py of the reference; generators :99-122, :148-212).

The generators draw from the same global ``numpy.random`` / ``random`` streams in the
same order as the reference, so seeding both modules reproduces the reference's
inputs exactly (checked against tests/golden/reference_golden.npz).
"""
import random as rd
import warnings

import numpy as np

__all__ = ["is_sorted", "bash_colors", "filter_warnings", "profile_run", "output_profile",
           "rescalepixels", "angles_gen", "pairs_gen", "checking_output", "noise_val",
           "subscan_resize", "system_setup"]


def is_sorted(seq):
    """True when ``seq`` is non-decreasing."""
    a = np.asarray(seq)
    return bool(np.all(a[:-1] <= a[1:]))


class bash_colors(object):
    """ANSI colour wrappers used by the reference's progress messages (:26-53)."""
    HEADER, OKBLUE, OKGREEN = '\033[95m', '\033[94m', '\033[92m'
    WARNING, FAIL, ENDC, BOLD, UNDERLINE = '\033[93m', '\033[91m', '\033[0m', '\033[1m', '\033[4m'

    def _wrap(self, code, s):
        return code + str(s) + self.ENDC

    def header(self, s): return self._wrap(self.HEADER, s)
    def blue(self, s): return self._wrap(self.OKBLUE, s)
    def green(self, s): return self._wrap(self.OKGREEN, s)
    def warning(self, s): return self._wrap(self.WARNING, s)
    def fail(self, s): return self._wrap(self.FAIL, s)
    def bold(self, s): return self._wrap(self.BOLD, s)
    def underline(self, s): return self._wrap(self.UNDERLINE, s)


def filter_warnings(wfilter):
    """'ignore' or 'always' (:55-62)."""
    warnings.simplefilter(wfilter)


def profile_run():
    """A fresh cProfile.Profile (:65-71)."""
    import cProfile
    return cProfile.Profile()


def output_profile(pr):
    """Print the cumulative-time table of a profile (:73-89)."""
    import io
    import pstats
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats()
    print(s.getvalue())


def rescalepixels(pixs):
    minpix, maxpix = min(pixs), max(pixs)
    return minpix, pixs - minpix, maxpix


def angles_gen(theta0, n, sample_freq=200., whwp_freq=2.5):
    """HWP angle ramp theta0 + 2 pi f_hwp / f_samp * i  (:99-107)."""
    return theta0 + 2 * np.pi * whwp_freq / sample_freq * np.arange(n)


def pairs_gen(nrows, ncols):
    """Uniform random pixel per sample (:111-122)."""
    if ncols < 3:
        raise RuntimeError("Not enough pixels!\n Please set Npix >=3, you have set Npix=%d" % ncols)
    return np.random.randint(0, high=ncols, size=nrows)


def checking_output(info):
    """Solver exit code -> True, or RuntimeError (:125-140)."""
    if info == 0:
        return True
    if info < 0:
        raise RuntimeError("illegal input or breakdown during the execution")
    raise RuntimeError("convergence not achieved after %d iterations" % info)


def noise_val(nb, bandwidth=1):
    """nb random bands of ``bandwidth`` entries and their first entries (:148-177)."""
    t = [np.random.random(size=bandwidth) for _ in range(nb)]
    return t, [b[0] for b in t]


def subscan_resize(data, subscan):
    """Keep only the sub-scan intervals of a TOD-sized array (:179-188)."""
    return np.concatenate([data[s:s + n] for n, s in zip(subscan[0], subscan[1])])


def system_setup(nt, npix, nb):
    """d, pairs, phi, t, diag of a random test problem (:190-212)."""
    d = np.random.random(nt)
    pairs = pairs_gen(nt, npix)
    phi = angles_gen(rd.uniform(0, np.pi), nt)
    t, diag = noise_val(nb, 2)
    return d, pairs, phi, t, diag
