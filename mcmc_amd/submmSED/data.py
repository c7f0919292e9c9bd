"""SED photometry containers and catalogue readers with the interface of the reference's
MCMC/submmSED/data.py.  Host-side I/O only: every reader returns a list of submmData, and
`catalogue_arrays` packs such a list (ragged band counts) into the padded [S, nb_max] arrays that
mi355_sed_load takes - the hot path starts there.

Numerics: text fields are parsed straight to float64.  The reference's fixed-width MRR reader goes
through float32 fields (`'f'` formats, data.py:331-337), so its outputs agree with these to
float32 rounding only (tests/test_catalogue_readers.py states the tolerance); the other readers
agree bit for bit."""
import csv
import fnmatch
import os
import re

import numpy as np

from ..GaussianData import GaussianData

speed_of_light = 299792.458   # micron GHz      data.py:28
verbose = False


class submmData(GaussianData):
    """SED measurements: rest-frame frequencies freq[], fluxes and errors.  data.py:30-45"""

    def __init__(self, freq, flux, sigma, name, z, nu_obs=None, z_alt=None, name_alt=None):
        GaussianData.__init__(self, np.asarray(flux), np.asarray(sigma))
        self.freq = np.asarray(freq)
        self.name = str(name)
        self.z = np.asarray(z)
        self.freq_obs = np.asarray(nu_obs)
        self.z_alt = z_alt
        self.name_alt = name_alt


def catalogue_arrays(datasets, pad_nu=1000.0):
    """list of submmData (ragged) -> nb[S] int32, nu[S,nb_max], d[S,nb_max], sig[S,nb_max] for
    mi355_sed_load (include/mi355mcmc.h).  Padding bands carry d=0, sig=1 and are never read
    (the kernels loop to nb[s])."""
    nb = np.array([np.size(d.d) for d in datasets], dtype=np.int32)
    nbm = max(int(nb.max()), 1) if len(nb) else 1
    S = len(datasets)
    nu = np.full((S, nbm), float(pad_nu))
    dd = np.zeros((S, nbm))
    sg = np.ones((S, nbm))
    for s, d in enumerate(datasets):
        n = nb[s]
        nu[s, :n] = np.ravel(d.freq)
        dd[s, :n] = np.ravel(d.d)
        sg[s, :n] = np.broadcast_to(np.ravel(d.sig), (n,))
    return nb, nu, dd, sg


def _index_is_true(idx):
    """`if l25:` of the reference (data.py:235, :250) on the result of np.where(...)[0]: an index
    array is true only when it holds exactly one *non-zero* index.  So a 25 um point in the first
    column is NOT turned into an upper limit - reproduced, the chains depend on it.  (NumPy >= 2.2
    raises on the empty array where older NumPy said False; more than one match always raised.)"""
    if len(idx) == 0:
        return False
    if len(idx) > 1:
        raise ValueError("The truth value of an array with more than one element is ambiguous "
                         "(two 25 micron points; data.py:235)")
    return bool(idx[0])


def _upper_limit_25(lam, flux, sig):
    l25 = np.where(np.round(lam) == 25)[0]
    if _index_is_true(l25):
        sig[l25] = flux[l25]
        flux[l25] = 0
    return flux, sig


def read_wavelength_table(filename, z, name, UL25=True):
    """A (wavelength[um], flux[Jy], error[Jy]) text table such as the reference's
    txt/Arp220.txt -> submmData, following the Arp220 branch of readfluxes_DLC_2014
    (data.py:243-256): nu_obs = c/lambda, the 25 um point becomes an upper limit
    (sig = flux, flux = 0) when UL25, nu_rest = nu_obs (1+z)."""
    lam, flux, sig = np.loadtxt(filename, unpack=True, usecols=(0, 1, 2))
    lam, flux, sig = (np.ascontiguousarray(a, dtype=np.float64) for a in (lam, flux, sig))
    nu_obs = speed_of_light / lam
    if UL25:
        flux, sig = _upper_limit_25(lam, flux, sig)
    return submmData(nu_obs * (1 + z), flux, sig, name, z, nu_obs=nu_obs)


def readfluxes_Arp220(filename, UL25=True):
    return [read_wavelength_table(filename, 0.018, "Arp220", UL25=UL25)]


# ---------------------------------------------------------------------------------------------
# DLC tables.  data.py:75-138
# ---------------------------------------------------------------------------------------------
def readfluxes_DLC(filename, format=0, delnu=None):
    """format 0: [name f1 e1 f2 e2 f3 e3 f4 e4 z] at 60, 100, 450, 850 um, two header lines
    format 1: [z f1 e1 f2 e2 f3 e3] at 250, 350, 500 um, one header line (the reference's branch
              never initialises its list, data.py:102-113; same columns, made to work)
    format 2: [name f1..f10 e1..e10] at z=0, the 25 um point always an upper limit
    format 3: Peel pixel table (readfluxes_peel)"""
    if format == 3:
        return readfluxes_peel(filename, delnu=delnu)
    data = []
    if format == 0:
        nu_obs = speed_of_light / np.asarray([60.0, 100, 450, 850])
        for row in np.atleast_2d(np.loadtxt(filename, skiprows=2)):
            z = row[9]
            data.append(submmData(nu_obs * (1.0 + z), row[1:9:2], row[2:9:2], str(int(row[0])), z,
                                  nu_obs=nu_obs))
    elif format == 1:
        nu_obs = speed_of_light / np.asarray([250., 350., 500.])
        for i, row in enumerate(np.atleast_2d(np.loadtxt(filename, skiprows=1))):
            z = row[0]
            data.append(submmData(nu_obs * (1.0 + z), row[1::2], row[2::2], str(i) + "_" + str(z), z,
                                  nu_obs=nu_obs))
    elif format == 2:
        nu_obs = speed_of_light / np.asarray([25.0, 60, 100, 250, 350, 350, 500, 550, 850, 1400])
        nband = len(nu_obs)
        for row in np.atleast_2d(np.loadtxt(filename, skiprows=1)):
            flux = row[1:nband + 1].copy()
            sig = row[nband + 1:].copy()
            sig[0] = flux[0]                              # detection -> 1-sigma upper limit
            flux[0] = 0.0
            data.append(submmData(nu_obs, flux, sig, str(int(row[0])), 0, nu_obs=nu_obs))
    else:
        raise ValueError("readfluxes_DLC: unknown format %r" % (format,))
    return data


_NAME_PAT = r"Name|Source"                               # data.py:170-173
_ERR_PAT = r"^E\d+_|^E\d+$"
_FLUX_PAT = r"^F\d+_|^S\_\d+$|^F\d+$"
_Z_PAT = r"Redshift|Redfshift|z|Z"


def _header_wavelength(h):
    return float(re.search(r"\d+(?:\.\d+)?", h).group(0))


def _one_column(headers, pat, what):
    cols = [i for i, h in enumerate(headers) if re.match(pat, h)]
    if len(cols) != 1:
        raise AssertionError("expected exactly one %s column, header has %d" % (what, len(cols)))
    return cols[0]


def toFloat(s):                                                      # data.py:140-141
    return float(s) if s != '' else np.nan


def readfluxes_DLC_2014(filename="./dat/herus_phot.csv", UL25=True, getArp220=True, del157=True,
                        is_csv=None, arp220_file=None):
    """HERUS-style photometry table, data.py:143-260: `name [obsID] z f1 e1 f2 e2 ...` with header
    names F<lam>[_Jy] / S_<lam> for fluxes and E<lam>[_Jy] for errors, CSV or whitespace-separated
    with a leading '#'.  Cells without a flux are skipped, entries without a positive error are
    dropped (an empty error cell reads as NaN and NaN > 0 is false), the 157 um point is deleted
    when a 160 um point exists (del157), a source called "Arp220" is renamed "Arp220-short".
    UL25: see _index_is_true.  getArp220 appends the 27-band Arp220 table (z = 0.018) from
    `arp220_file` (the reference ships it as package data dat/Arp220.txt)."""
    data = []
    with open(filename) as f:
        first = f.readline()
        if is_csv is None:
            is_csv = "," in first
        f.seek(0)
        if is_csv:
            rd = csv.reader(f)
            headers = next(rd)
            rows = [r for r in rd]
        else:
            headers = f.readline().split()
            assert headers[0][0] == "#"
            headers[0] = headers[0][1:]
            rows = [ln.split() for ln in f if ln.strip()]
    name_col = _one_column(headers, _NAME_PAT, "name")
    z_col = _one_column(headers, _Z_PAT, "redshift")
    fcols = [i for i, h in enumerate(headers) if re.match(_FLUX_PAT, h)]
    ecols = [i for i, h in enumerate(headers) if re.match(_ERR_PAT, h)]
    lam_f = np.array([_header_wavelength(headers[i]) for i in fcols])
    lam_e = np.array([_header_wavelength(headers[i]) for i in ecols])
    assert lam_f.shape == lam_e.shape and np.all(lam_f == lam_e)   # flux / error columns pair up
    ncol = len(headers)
    for row in rows:
        row = list(row) + [""] * (ncol - len(row))
        if all(c == "" for c in row):
            continue
        name = row[name_col]
        if name == "Arp220":
            name = "Arp220-short"
        z = float(row[z_col])
        ent = [(float(row[i]), float(row[j]) if row[j] != "" else np.nan, lam)
               for i, j, lam in zip(fcols, ecols, lam_f) if row[i] != ""]
        ent = np.array([e for e in ent if e[1] > 0], dtype=np.float64).reshape(-1, 3)
        if del157:
            l157 = np.where(np.round(ent[:, 2]) == 157.0)[0]
            l160 = np.where(np.round(ent[:, 2]) == 160.0)[0]
            if len(l157) and len(l160):
                if verbose:
                    print(name, "deleting 157 micron")
                ent = np.delete(ent, l157, axis=0)
        flux, sig, lam = ent[:, 0].copy(), ent[:, 1].copy(), ent[:, 2].copy()
        nu_obs = speed_of_light / lam
        if UL25:
            flux, sig = _upper_limit_25(lam, flux, sig)
        data.append(submmData(nu_obs * (1 + z), flux, sig, name, z, nu_obs=nu_obs))
    if getArp220:
        if arp220_file is None:
            arp220_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dat", "Arp220.txt")
        if not os.path.exists(arp220_file):
            raise FileNotFoundError("getArp220=True needs the Arp220 wavelength table: pass "
                                    "arp220_file= (not found: %s)" % arp220_file)
        data.append(read_wavelength_table(arp220_file, 0.018, "Arp220", UL25=UL25))
    return data


# ---------------------------------------------------------------------------------------------
# MRR fixed-width IIFSCz x ERCSC catalogue.  data.py:264-450
# ---------------------------------------------------------------------------------------------
def _mrr_layout(old):
    """(name, width, kind) of every field of FORMAT 135 (data.py:282-294); kind 'a' text, 'i'
    integer, 'f' real.  The new files carry `next` between glat and dist."""
    lay = [("nameIRAS", 13, "a"), ("ra", 11, "f"), ("dec", 11, "f"), ("posFlag", 4, "i")]
    lay += [(n, 10, "f") for n in ("s12", "s25", "s60", "s100")]
    lay += [("nq1", 3, "i"), ("nq2", 3, "i"), ("nq3", 3, "i"), ("nq4", 6, "i")]
    lay += [("am%d" % k, 8, "f") for k in range(1, 9)] + [("em%d" % k, 8, "f") for k in range(1, 9)]
    lay += [("photFlag", 6, "i"), ("FINT", 14, "f"), ("EFINT", 11, "f"), ("zspec", 11, "f"),
            ("zspecFlag", 4, "i")]
    lay += [(n, 11, "f") for n in ("zneur", "zneurerr", "ztem", "z")]
    lay += [("j2", 3, "i"), ("av1", 6, "f"), ("err1", 11, "f"), ("zneurFlag", 4, "i"),
            ("amb2", 8, "f"), ("alb", 10, "f")]
    lay += [(n, 8, "f") for n in ("alp1", "alp2", "alp3", "alp4")]
    lay += [("alcirr", 8, "f"), ("alsb", 8, "f"), ("ala220", 8, "f"), ("alagn", 11, "f")]
    lay += [("alir", 8, "f"), ("nirtem", 7, "i"), ("errir3", 11, "f")]
    lay += [(n, 7, "f") for n in ("als12", "als25", "als60", "als90", "als100", "als110", "als140",
                                  "als160", "als250", "als350", "als500", "als850", "als1250",
                                  "als1380")]
    lay += [("nirflag", 7, "i"), ("ra1", 10, "f"), ("dec1", 10, "f")]
    lay += [(n, 11, "f") for n in ("s857", "e857", "s217", "e217", "s353", "e353", "s545", "e545")]
    lay += [("glon", 10, "f"), ("glat", 10, "f")]
    if not old:
        lay += [("next", 3, "i")]
    lay += [("dist", 15, "f"), ("nedtp", 7, "a"), ("sdsstp", 7, "a"), ("nameNED", 24, "a"),
            ("nameSDSS", 23, "a"), ("name2MASS", 22, "a")]
    return lay


def _mrr_field(text, kind):
    t = text.strip()
    if kind == "a":
        return text
    if kind == "i":
        try:
            return int(t)
        except ValueError:
            return -1                                     # genfromtxt's missing integer
    try:
        return float(t)
    except ValueError:
        return np.nan


def readfluxes_MRR(filename, IRAS_ignore=None, Planck_ignore=None, DLC_ul=False, old_MRR=False,
                   next0=True, colorx=(1.02, 1.1, 1.13, 1.0)):
    """One submmData per line of an MRR catalogue file.  Bands: the Planck points
    857/545/353/217 GHz (mJy -> Jy, divided by the colour corrections `colorx`) minus
    Planck_ignore, then the IRAS points 12/25/60/100 um minus IRAS_ignore, with errors from the
    IRAS quality flags: nq=1 upper limit (sig = flux, flux = 0), nq=2 sig = flux/2, nq=3 or 5
    sig = flux/10; non-positive IRAS fluxes are skipped; DLC_ul forces the 25 um point to an
    upper limit.  next0 keeps only objects with next == 0 (new-format files)."""
    want = ("nameIRAS", "s12", "s25", "s60", "s100", "nq1", "nq2", "nq3", "nq4", "z",
            "s857", "e857", "s217", "e217", "s353", "e353", "s545", "e545", "next")
    lay = _mrr_layout(old_MRR)
    if next0 and old_MRR:
        raise ValueError("no field named next: old-format MRR files have no `next` column, "
                         "pass next0=False")
    spans, pos = {}, 0
    for n, w, k in lay:
        if n in want:
            spans[n] = (pos, pos + w, k)
        pos += w
    IRAS_ignore = set() if IRAS_ignore is None else set(IRAS_ignore)
    Planck_ignore = set() if Planck_ignore is None else set(Planck_ignore)
    lambda_IRAS = (12.0, 25.0, 60.0, 100.0)
    planck = [(nu, (1.0 if colorx is None else colorx[i]))
              for i, nu in enumerate((857., 545., 353., 217.)) if i not in Planck_ignore]
    data = []
    with open(filename) as f:
        for line in f:
            if not line.strip():
                continue
            line = line.rstrip("\n")
            obj = {n: _mrr_field(line[a:b], k) for n, (a, b, k) in spans.items()}
            if next0 and obj["next"] != 0:
                if verbose:
                    print("Skipping object %s" % obj["nameIRAS"])
                continue
            z = obj["z"]
            name = obj["nameIRAS"].strip()
            nu_obs = [nu for nu, _ in planck]
            flux = [1e-3 * obj["s%d" % nu] / cx for nu, cx in planck]
            sig = [1e-3 * obj["e%d" % nu] / cx for nu, cx in planck]
            sg = None
            for i, lam in enumerate(lambda_IRAS):
                if i in IRAS_ignore:
                    continue
                nq = obj["nq%d" % (i + 1)]
                flx = obj["s%d" % int(lam)]
                if DLC_ul and int(lam) == 25:
                    nq = 1
                if flx <= 0:
                    continue
                if nq == 1:
                    sg, flx = flx, 0.0
                elif nq == 2:
                    sg = 0.5 * flx
                elif nq in (3, 5):
                    sg = 0.1 * flx
                elif sg is None:
                    # the reference only prints a warning and reuses the previous band's error
                    # (data.py:436-441); with no previous band it has nothing to reuse
                    raise ValueError("IRAS quality flag nq=%d at %s and no earlier band" % (nq, name))
                nu_obs.append(speed_of_light / lam)
                flux.append(flx)
                sig.append(sg)
            nu_obs = np.asarray(nu_obs)
            data.append(submmData((1 + z) * nu_obs, flux, sig, name, z, nu_obs=nu_obs))
    return data


# ---------------------------------------------------------------------------------------------
# ERCSC TopCat ASCII table.  data.py:453-540
# ---------------------------------------------------------------------------------------------
def readfluxes_ERCSC_TopCat(filename, upperlim=2.0, delete_upperlim=False):
    """Rows of a TopCat '|'-separated ASCII table below its '+---' ruled header: Planck fluxes
    at 217/353/545/857 GHz (mJy -> Jy) with errors, then IRAS 25/60/100 um with fractional errors
    1.0/0.1/0.1 (12 um ignored); rest frame from the spectroscopic z (column 15), the other z kept
    as z_alt.  upperlim: detections below upperlim*sigma become upper limits (flux = 0; sig =
    2*flux for positive, 2*sig for negative fluxes) and the name gets a 'U'.  delete_upperlim
    drops them instead (name + 'D'); the reference's version of that branch shrinks the shared
    nu_obs for all later objects (data.py:521) - here it is per object."""
    err_IRAS = np.asarray([1.0, 0.1, 0.1])
    nu_all = np.concatenate((np.asarray([217., 353., 545., 857.]),
                             speed_of_light / np.asarray([25.0, 60.0, 100.0])))
    planck_idx, iras_idx = (0, 2, 4, 6), (12, 13, 14)
    rows = []
    with open(filename) as f:
        for line in f:
            if line.startswith("+-"):
                break
        else:
            raise ValueError("%s: no '+-' ruled header found" % filename)
        f.readline()                                      # column names
        f.readline()                                      # ruler
        for line in f:
            if line.startswith("+-") or not line.strip():
                continue
            rows.append([c.strip() for c in line.split("|")[1:20]])
    data = []
    for obj in rows:
        name, name_alt = obj[8], obj[17]
        zspec, z = float(obj[15]), float(obj[16])
        flux = np.asarray([1e-3 * float(obj[i]) for i in planck_idx] + [float(obj[i]) for i in iras_idx])
        sig = np.asarray([1e-3 * float(obj[i + 1]) for i in planck_idx] +
                         [ef * float(obj[i]) for i, ef in zip(iras_idx, err_IRAS)])
        nu_obs = nu_all.copy()
        with np.errstate(all="ignore"):
            if delete_upperlim:
                good = np.logical_and(flux > 0, flux / sig > upperlim)
                flux, sig, nu_obs = flux[good], sig[good], nu_obs[good]
                if not np.all(good):
                    name += "D"
            elif upperlim:
                lt0 = flux < 0
                gt0 = np.logical_and(flux > 0, flux / sig < upperlim)
                idx = np.logical_or(lt0, gt0)
                if np.any(idx):
                    f_gt0 = flux[gt0]
                    sig[lt0] = 2 * sig[lt0]
                    flux[idx] = 0
                    sig[gt0] = 2 * f_gt0
                    name += "U"
        data.append(submmData(nu_obs * (1. + zspec), flux, sig, name, zspec, nu_obs=nu_obs, z_alt=z,
                              name_alt=name_alt))
    return data


# ---------------------------------------------------------------------------------------------
# Peel pixel table, M31 integrated SED, Mortier directory.  data.py:542-607
# ---------------------------------------------------------------------------------------------
def readfluxes_peel(filename, delnu=None):
    """`# i j 217 Err 353 Err ... 25000 Err`: one SED per pixel at z=0, 12 bands given in GHz;
    delnu = frequencies to drop (located with searchsorted, as the reference)."""
    nu_obs = np.array([217, 353, 545, 857, 1763, 1870, 3000, 4280, 5000, 12000, 12490, 25000])
    didx = None
    if delnu is not None:
        didx = np.searchsorted(nu_obs, delnu)
        nu_obs = np.delete(nu_obs, didx)
    data = []
    for i, row in enumerate(np.atleast_2d(np.loadtxt(filename))):
        flux, sig = row[2::2], row[3::2]
        if didx is not None:
            flux, sig = np.delete(flux, didx), np.delete(sig, didx)
        name = "_".join(str(int(c)) for c in (i, row[0], row[1]))
        data.append(submmData(nu_obs * 1.0, flux, sig, name, 0.0, nu_obs=nu_obs))
    return data


def readfluxes_M31(filename="M31/M31Flux-v2.dat", delnu=None):
    """single SED: columns frequency [GHz], flux density [Jy], error [Jy]"""
    nu, flux, err = np.loadtxt(filename, unpack=True)
    if delnu is not None:
        didx = np.searchsorted(nu, delnu)
        nu, flux, err = np.delete(nu, didx), np.delete(flux, didx), np.delete(err, didx)
    return [submmData(nu, flux, err, "M31", 0.0, nu_obs=nu)]


def readfluxes_mortier(dirname, delnu=None):
    """A directory of `..._<name>_fittype0.txt` files with columns nu_emit [Hz], flux [Jy],
    error [Jy] under one header line -> GHz and mJy.  Files are taken in sorted order (the
    reference takes os.listdir order)."""
    data = []
    for fn in sorted(fnmatch.filter(os.listdir(dirname), "*.txt")):
        name = fn.split("_")[3]
        nu, flux, err = np.loadtxt(os.path.join(dirname, fn), skiprows=1, unpack=True, ndmin=2)
        data.append(submmData(nu / 1e9, flux * 1e3, err * 1e3, name, 0.0))
    return data
