"""Read the tables written by driver.writeTab back in (interface of the reference's
MCMC/submmSED/readtab.py:11-23; its plotting helpers are not rebuilt)."""
import numpy as np


def readtab(f, verbose=True):
    """-> structured array, one field per column (header names are separated by more than one
    space; numpy turns their blanks into underscores: "ML param 0" -> ML_param_0; the Name
    column, not a number, reads as nan), and prints mean +- std of every column."""
    with open(f) as fp:
        hdr = fp.readline()
        names = [s.strip() for s in hdr.split("  ") if s.strip()]
        ret = np.genfromtxt(fp, names=names)
    if verbose:
        for col in ret.dtype.fields.keys():
            print("%s = %f +- %f" % (col, ret[col].mean(), ret[col].std()))
    return ret
