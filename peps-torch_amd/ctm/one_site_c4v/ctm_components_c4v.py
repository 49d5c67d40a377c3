"""Enlarged corner of the one-site C4v network (reference ctm/one_site_c4v/ctm_components_c4v.py:9-130)."""
from backend import get_engine


def c2x2_sl(a, C, T, verbosity=0):
    return get_engine().c2x2_c4v(a, C, T, open_=False)


c2x2_dl = c2x2_sl
