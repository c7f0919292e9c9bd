"""Factory and list reader of the reference's MCMC/ClData/ClData.py:7-38, plus readbins
(readbins.py:1-25)."""
from . import ClData_CosmoMC, ClData_bcp


def ClData(file, **kw):
    """the appropriate data class for the file name: *.newdat -> ClData_bcp, else the CosmoMC
    .dataset format.  ('MAP'/'WMAP' selected the WMAP likelihood wrapper, which is not rebuilt.)"""
    if file in ("MAP", "WMAP"):
        raise NotImplementedError("the WMAP likelihood wrapper (ClData_WMAP.py) is out of scope")
    if file.split(".")[-1] == "newdat":
        return ClData_bcp.ClData_bcp(file, **kw)
    return ClData_CosmoMC.ClData_CosmoMC(file, **kw)


def getClData(listfile, verbose=True, no_pol=False, **kw):
    """one data set per non-blank, non-# line of listfile"""
    out = []
    with open(listfile) as f:
        for name in f:
            name = name.strip()
            if len(name) and name[0] != "#":
                if verbose:
                    print("Getting %s" % name)
                dset = ClData(name, **kw)
                if no_pol:
                    dset.has_pol_really = dset.has_pol
                    dset.has_pol = False                   # explicitly ignore polarization
                if verbose:
                    print(" got %s" % dset.name)
                out.append(dset)
    return out


def readbins(filename):
    """bins in C. Contaldi's format, lines `lmin lmax type` with type 1, 2, 3 = TT, TE, EE ->
    [[(TTstart, TTend), ...], [(TE...)], [(EE...)]]"""
    binlist = [[], [], []]
    with open(filename) as f:
        for line in f:
            if not line.strip():
                continue
            lo, hi, kind = [int(i) for i in line.split()]
            binlist[kind - 1].append((lo, hi))
    return binlist
