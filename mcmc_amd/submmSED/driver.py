"""Catalogue driver: the counterpart of the reference's MCMC/submmSED/driver.py (main / many /
postprocess / writeTab, driver.py:76-534) for a device that runs every object of a catalogue at
once.

Where the reference loops `for i in idata:` over objects and runs one Python chain each, `main`
here loads all selected objects into one Engine, runs `nchains` chains per object through the
same stages (stage k+1 restarts every chain from its own stage-k moments with
sigma = 2.4/sqrt(P) * stdev, MCMC.py:343-413), and returns one record per object in the layout
of the reference's `ret_i`:

    (ana, (maxlnLike, maxLikeParams, meanlnProb), name, z, [(d, sig), ...], MLmodel.flux(nu1, nu2),
     ML_chi2)          with ana = (means, stdevs, normalised covariance)   [driver.py:237-244]

so `postprocess` / `writeTab` produce the same tables.  Plotting, the getdist hooks and the
pickle checkpoints are not rebuilt (results are kept as .npz); statistics are pooled over the
chains of an object: mean and stdev from the exact on-device moments of the last stage,
covariance and maximum-likelihood sample from the thinned trace of that stage."""
import collections.abc

import numpy as np

from . import data, likelihood, model
from ..engine import Engine

delnu = 1763                                               # driver.py:66
nMC = (50000, 500000, 500000)                              # driver.py:69

_RUN_IDX = [[0, 2, 3, 5], [0, 1, 2], [0, 2], [0, 1, 2, 3, 4, 5], [0, 1, 2, 3]]   # driver.py:364
_RUN_NT = [2, 1, 1, 2, 1]                                  # temperature components per run


def read_catalogue(filename, filetype="MRR", DLC_ul=False, next0=True, format=0):
    """the reader dispatch of main(), driver.py:94-114"""
    ft = filetype.upper()
    if ft == "DLC":
        return data.readfluxes_DLC(filename, format=format, delnu=delnu)
    if ft == "MRR":
        if DLC_ul == 1:                                    # no 12 um, no 217 GHz, 25 um upper limit
            return data.readfluxes_MRR(filename, IRAS_ignore=[0], Planck_ignore=[3], DLC_ul=True,
                                       next0=next0)
        if DLC_ul == 2:                                    # ... and no 353 GHz
            return data.readfluxes_MRR(filename, IRAS_ignore=[0], Planck_ignore=[2, 3], DLC_ul=True,
                                       next0=next0)
        return data.readfluxes_MRR(filename)
    if ft == "PEEL":
        return data.readfluxes_peel(filename, delnu=delnu)
    if ft == "MORTIER":
        return data.readfluxes_mortier(filename)
    if ft == "M31":
        return data.readfluxes_M31(filename)
    if ft == "DLC_2014":
        return data.readfluxes_DLC_2014(filename, del157=True)
    return data.readfluxes_ERCSC_TopCat(filename)


def select_model(onecomponent=True, getNorm=True, opticallyThick=False, linear=False):
    """model and likelihood classes as main() pairs them, driver.py:141-159"""
    if opticallyThick:
        mod = model.submmModel1_opticallythick if linear else model.submmModel1_opticallythick_logA
        return mod, likelihood.SEDLikelihood_normalized
    if getNorm:
        if onecomponent:
            mod = model.submmModel1_normalized if linear else model.submmModel1_normalized_logA
        else:
            mod = model.submmModel2_normalized if linear else model.submmModel2_normalized_logA
        return mod, likelihood.SEDLikelihood_normalized
    if onecomponent:
        return model.submmModel1, likelihood.SEDLikelihood1
    return model.submmModel2, likelihood.SEDLikelihood2


def _pooled_stats(mom, samples, lnPr, params, burn_rows):
    """per object: pooled mean / stdev over chains (exact, from the device moments), normalised
    covariance (chain_analyze's definition, MCMC.py:445-477, on the pooled thinned trace) and the
    maximum-lnPr sample of the trace after `burn_rows` rows"""
    mean_c, var_c = mom["mean"], mom["var"]                # [S, C, P]
    S, C, P = mean_c.shape
    mean = mean_c.mean(axis=1)
    var = var_c.mean(axis=1) + mean_c.var(axis=1)          # law of total variance, ddof 0
    std = np.sqrt(var)
    tr = samples[burn_rows:]                               # [n, S, C, P]
    n = tr.shape[0]
    flat = np.transpose(tr, (1, 0, 2, 3)).reshape(S, n * C, P)
    dev = flat - flat.mean(axis=1, keepdims=True)
    cov = np.einsum("sni,snj->sij", dev, dev) / (n * C)
    sd = np.sqrt(np.einsum("sii->si", cov))
    ncov = np.zeros((S, P, P))
    with np.errstate(all="ignore"):
        for i in params:
            for j in params:
                ncov[:, i, j] = cov[:, i, j] / (sd[:, i] * sd[:, j])
    lp = np.transpose(lnPr[burn_rows:], (1, 0, 2)).reshape(S, n * C)
    best = np.argmax(lp, axis=1)
    return mean, std, ncov, lp[np.arange(S), best], flat[np.arange(S), best]


def fit_catalogue(datasets, mod, like_cls=None, start=None, sigmas=None, nMC=nMC, nchains=64,
                  fac=None, random=True, randomrestart=False, seed=0x5EDC0FFEE, max_rows=128,
                  burn=0.2, device=0, return_engine=False, batch_sources=8192):
    """All objects of `datasets` x `nchains` chains through the stages nMC on one GPU; returns the
    per-object records described in the module docstring plus, as a second value, a dict of
    per-object diagnostics (Rhat [S,P], ESS [S,P], acceptance).  Catalogues larger than
    `batch_sources` objects are fitted in consecutive batches: the thinned trace the statistics are
    taken from comes back to the host (max_rows x nchains x P doubles per object)."""
    if len(datasets) > batch_sources and not return_engine:
        records, parts = [], []
        for lo in range(0, len(datasets), batch_sources):
            r, dg = fit_catalogue(datasets[lo:lo + batch_sources], mod, like_cls, start, sigmas, nMC,
                                  nchains, fac, random, randomrestart, seed + lo, max_rows, burn, device,
                                  False, batch_sources)
            records += r
            parts.append(dg)
        diag = {k: np.concatenate([p[k] for p in parts]) for k in ("rhat", "ess", "accept_rate")}
        diag["thin"] = parts[0]["thin"]
        diag["resampled"] = [x for p in parts for x in p["resampled"]]
        return records, diag
    if like_cls is None:
        like_cls = likelihood.SEDLikelihood_normalized
    start = np.asarray(mod.startfrom(random=random) if start is None else start, np.float64)
    sigmas = np.asarray(start / 4.0 if sigmas is None else sigmas, np.float64)   # driver.py:165-168
    P = mod.nparam
    eng = Engine(device=device, seed=seed)
    nb, nu, d, sg = data.catalogue_arrays(datasets)
    eng.sed_load(mod.model_id, nb, nu, d, sg, consts=model.device_constants())
    eng.mh_init(nchains, start, sigmas, param_blocks=np.asarray(mod.paramBlocks, np.int32),
                randomize_start=True)                      # doBlock=True, driver.py:173-175
    params = np.where(sigmas > 0)[0]
    resampled = []

    def stuck_chains():
        """chains whose stage statistics are unusable: max(stdev/mean) < 1e-9 over the sampled
        parameters, the test of sampler() (MCMC.py:549-560)"""
        m = eng.mh_moments()
        with np.errstate(all="ignore"):
            ratio = np.sqrt(m["var"][..., params]) / m["mean"][..., params]
        return int(np.sum(np.nanmax(ratio, axis=-1) < 1.e-9))

    for n in nMC[:-1]:
        # A stage is repeated (appended to) while it produced no usable statistics, at most ten
        # times in all (nresample_max, MCMC.py:491).  One launch serves every chain, so all
        # chains of the catalogue get the extra steps, not only the stuck ones.
        eng.mh_run(int(n))
        while len(resampled) < 10 and stuck_chains() > 0:
            resampled.append(int(n))
            eng.mh_run(int(n))
        eng.mh_adapt(fac=fac, random_restart=randomrestart)
    n_last = int(nMC[-1])
    thin = max(1, -(-n_last // max_rows))                  # <= max_rows rows (mi355_mh_ess takes 128)
    eng.mh_run(n_last, thin=thin)
    mom = eng.mh_moments()
    samples, lnPr, _ = eng.mh_trace()
    burn_rows = int(burn * samples.shape[0])               # histgrid(burn=0.2), driver.py:182-184
    mean, std, ncov, maxlnL, mlpar = _pooled_stats(mom, samples, lnPr, params, burn_rows)
    # package() as MCMC.mean()/stdev() apply it (MCMC.py:297,311)
    _, mean_pk = eng.sed_prior_package(mean)
    _, std_pk = eng.sed_prior_package(std)
    S = len(datasets)
    lnl_mean = eng.sed_lnlike(mean_pk.reshape(S, 1, P))[:, 0]
    out = eng.sed_lnlike(mlpar.reshape(S, 1, P), derived=like_cls.nDerived > 0)
    lnl_ml = (out[0] if like_cls.nDerived > 0 else out)[:, 0]
    nu2, nu1 = 299792. / 8.0, 299792. / 1000.0             # rest-frame 8-1000 um, driver.py:83-84
    marginalised = like_cls is not likelihood.SEDLikelihood_normalized
    records = []
    for s, dat in enumerate(datasets):
        ana = (mean_pk[s], std_pk[s], ncov[s])
        name = dat.name + "(z=%5.3f)" % (dat.z)
        rec = (ana, (float(maxlnL[s]), mlpar[s], float(lnl_mean[s])), name, dat.z,
               list(zip(dat.d, np.broadcast_to(dat.sig, np.shape(dat.d)))))
        mlmod = mod(*mlpar[s])
        rec += (mlmod.flux(nu1, nu2) if hasattr(mlmod, "flux") else np.zeros(0),)
        # Likelihood.chi2 (Likelihood.py:107-123); -2 lnL for the explicit-amplitude likelihoods
        rec += (float(-2.0 * lnl_ml[s]) if not marginalised else float("nan"),)
        records.append(rec)
    diag = dict(rhat=eng.rhat(), ess=eng.mh_ess(), accept_rate=mom["naccept"] / float(eng.nsamples + 1),
                thin=thin, resampled=resampled)
    if return_engine:
        return records, diag, eng
    eng.close()
    return records, diag


def main(filename, i=None, rotateParams=False, onecomponent=True, getNorm=True, start=None,
         sigmas=None, nMC=nMC, opticallyThick=False, random=True, randomrestart=False,
         DLC_ul=False, next0=True, format=0, filetype="MRR", linear=False, nchains=64,
         seed=0x5EDC0FFEE, check=None, **ignored):
    """driver.py:76-269 without the plots: read `filename`, select the objects `i` (None = all, an
    index or a list), fit them all at once.  Returns the list of records (a single record when one
    object was asked for, as the reference).  check = file name for an .npz checkpoint."""
    if rotateParams:
        raise NotImplementedError("rotateParams needs the step-wise engine: use mcmc_amd.MCMC.sampler "
                                  "per object")
    alldata = read_catalogue(filename, filetype, DLC_ul=DLC_ul, next0=next0, format=format)
    if i is None:
        idata = list(range(len(alldata)))
    elif not isinstance(i, collections.abc.Sequence):
        idata = [i]
    else:
        idata = [k for k in i if k < len(alldata)]         # "no more objects", driver.py:133-135
    mod, like_cls = select_model(onecomponent, getNorm, opticallyThick, linear)
    records, diag = fit_catalogue([alldata[k] for k in idata], mod, like_cls, start=start,
                                  sigmas=sigmas, nMC=nMC, nchains=nchains, random=random,
                                  randomrestart=randomrestart, seed=seed)
    if check is not None:
        save_records(check, records)
    return records[0] if len(records) == 1 else records


def many(which=tuple(range(5)), idata=None, nMC=nMC, fil=None, cdir="./", next0=True, **keywords):
    """the five production fits of driver.py:292-342 (proposal widths 2, 4, 4, 10; beta fixed by
    sigma = 0 in the "b2" runs)"""
    sA, sB, sT, snu = 2, 4, 4, 10
    runs = {
        0: dict(start=(1, 2., 10, 0.1, 2., 20), sigmas=(sA, 0, sT) * 2, onecomponent=False),
        1: dict(start=(1, 2., 10), sigmas=(sA, sB, sT), onecomponent=True),
        2: dict(start=(1, 2., 10), sigmas=(sA, 0, sT), onecomponent=True),
        3: dict(start=(1, 2., 10, 0.1, 2., 20), sigmas=(sA, sB, sT) * 2, onecomponent=False),
        4: dict(start=(1, 2., 10, 100.0), sigmas=(sA, sB, sT, snu), onecomponent=False,
                opticallyThick=True),
    }
    ret = [[] for _ in range(5)]
    for k in range(5):
        if k in which:
            kw = dict(runs[k])
            kw.update(keywords)
            r = main(fil, getNorm=True, i=idata, nMC=nMC, check=cdir + "check%d.npz" % k,
                     next0=next0, **kw)
            ret[k] = r if isinstance(r, list) else [r]
    return tuple(ret)


# ---------------------------------------------------------------------------------------------
# results tables.  driver.py:345-534
# ---------------------------------------------------------------------------------------------
def save_records(fname, records):
    """checkpoint in .npz form (the reference pickles the list, driver.py:260-262)"""
    ndat = max(len(r[4]) for r in records)
    dat = np.full((len(records), ndat, 2), np.nan)
    for k, r in enumerate(records):
        dat[k, :len(r[4])] = np.asarray(r[4]).reshape(-1, 2)
    np.savez(fname, mean=np.array([r[0][0] for r in records]), sig=np.array([r[0][1] for r in records]),
             covar=np.array([r[0][2] for r in records]), ML=np.array([r[1][0] for r in records]),
             MLpar=np.array([r[1][1] for r in records]), MeanL=np.array([r[1][2] for r in records]),
             name=np.array([r[2] for r in records]), z=np.array([float(r[3]) for r in records]),
             dat=dat, ndat=np.array([len(r[4]) for r in records]),
             flux=np.array([np.asarray(r[5], float) for r in records]),
             ML_chi2=np.array([r[6] for r in records]))


def load_records(fname):
    with np.load(fname) as f:
        out = []
        for k in range(len(f["name"])):
            n = int(f["ndat"][k])
            out.append(((f["mean"][k], f["sig"][k], f["covar"][k]),
                        (float(f["ML"][k]), f["MLpar"][k], float(f["MeanL"][k])), str(f["name"][k]),
                        float(f["z"][k]), [tuple(x) for x in f["dat"][k, :n]], f["flux"][k],
                        float(f["ML_chi2"][k])))
        return out


def results_table(records, irun, nodat=False):
    """The structured array postprocess() builds for run `irun` (0..4) from a list of records:
    the run's free parameters (driver.py:364), the Laplace "evidence" columns exactly as the
    reference computes them (ML + det(covar)/2 + npar/2 ln 2 pi, driver.py:436-441), data and
    greybody fluxes."""
    ix = _RUN_IDX[irun]
    npar = len(ix)
    ndat = 0 if (nodat or not records) else len(records[0][4])
    if ndat and any(len(r[4]) != ndat for r in records):
        raise ValueError("objects have different numbers of data points: pass nodat=True "
                         "(driver.py:481)")
    dt = np.dtype([("name", "U21"), ("mean", float, (npar,)), ("sig", float, (npar,)),
                   ("covar", float, (npar, npar)), ("ML", float), ("ev", float),
                   ("MLpar", float, (npar,)), ("MeanL", float), ("evMean", float),
                   ("dlnLike", float), ("z", float), ("dat", float, (ndat, 2)),
                   ("flux", float, (_RUN_NT[irun],)), ("ML_chi2", float)])
    tab = np.zeros(len(records), dtype=dt)
    for k, obj in enumerate(records):
        covar = np.array(obj[0][2])[:, ix][ix]
        ML, meanL = obj[1][0], obj[1][2]
        hdet, norm = 0.5 * np.linalg.det(covar), npar * 0.5 * np.log(2 * np.pi)   # summed left to right
        tab[k]["name"] = obj[2]
        tab[k]["mean"] = np.array(obj[0][0])[ix]
        tab[k]["sig"] = np.array(obj[0][1])[ix]
        tab[k]["covar"] = covar
        tab[k]["ML"] = ML
        tab[k]["ev"] = ML + hdet + norm
        tab[k]["MLpar"] = np.asarray(obj[1][1])[ix]
        tab[k]["MeanL"] = meanL
        tab[k]["evMean"] = meanL + hdet + norm
        tab[k]["dlnLike"] = ML - meanL
        tab[k]["z"] = obj[3]
        if ndat > 0:
            tab[k]["dat"] = np.array(obj[4])
        tab[k]["flux"] = np.array(obj[5])
        tab[k]["ML_chi2"] = obj[6]
    return tab


def postprocess(runs, nodat=False):
    """runs = the tuple many() returns, or a list of checkpoint file names per run (None to skip);
    -> list of results tables, driver.py:345-460"""
    out = []
    for irun, r in enumerate(runs):
        if isinstance(r, str):
            r = load_records(r)
        out.append(results_table(r, irun, nodat=nodat) if r is not None and len(r) else [])
    return out


def writeTab(ret, fname, names=None, nodat=False, plain=False):
    """One text table per results table, columns and "%21s" cells as driver.py:478-534: Name, z,
    ML / mean / sigma of every parameter, dlnLike, evidence1, evidence2, [flux k, sigflux k ...],
    greybody flux k ..., ML_chi2.

    By default the file is byte for byte what the reference writes under Python 3
    (tests/golden/writetab_g11.json): its table is one bytes array (names 'S21' stacked with the
    numbers), so every cell prints as b'...'; numbers are NumPy's shortest round-trip text cut to
    32 characters; and with nodat=True only the HEADER drops the flux / sigflux names - the rows keep
    whatever the table's 'dat' field holds (nothing for tables made with nodat=True).  plain=True
    writes the same layout with bare cells, which is what readtab() parses."""
    anames = ret["name"] if names is None else np.asarray(names)
    nn = ret.shape[0]
    npar = ret["MLpar"].shape[1]
    ndat = 0 if nodat else ret["dat"].shape[1]
    nt = ret["flux"].shape[1]
    nums = np.hstack([ret["z"].reshape(nn, 1), ret["MLpar"], ret["mean"], ret["sig"],
                      ret["dlnLike"].reshape(nn, 1), ret["ev"].reshape(nn, 1), ret["evMean"].reshape(nn, 1),
                      ret["dat"].reshape(nn, -1), ret["flux"].reshape(nn, -1), ret["ML_chi2"].reshape(nn, 1)])
    hdr = ["Name", "z"]
    hdr += ["ML param %d" % k for k in range(npar)]
    hdr += ["Mean param %d" % k for k in range(npar)]
    hdr += ["sigma param %d" % k for k in range(npar)]
    hdr += ["dlnLike", "evidence1", "evidence2"]
    for k in range(ndat):
        hdr += ["flux %d" % k, "sigflux %d" % k]
    hdr += ["greybody flux %d" % k for k in range(nt)]
    hdr.append("ML_chi2")
    cells = np.hstack([np.asarray(anames).astype("S21").astype("S32").reshape(nn, 1), nums.astype("S32")])
    with open(fname, "w") as f:
        f.write(("%21s " * len(hdr)) % tuple(hdr) + "\n")
        for row in cells:
            f.write(" ".join("%21s" % (c.decode() if plain else c) for c in row) + "\n")


def writeTabAll(ret123, fbase, ext=".npy", nodat=False, plain=True):
    """driver.py:463-475: one file fbase<i><ext> per non-empty run.  Tables with bare cells by default,
    which is what readtab() reads back (plain=False: the reference's Python-3 text with b'...' cells,
    see writeTab).  With nodat=True the tables must have been made with nodat=True as well, or the
    header would have fewer columns than the rows (writeTab keeps that quirk of the reference for its
    byte-parity mode; here it is refused)."""
    for k, r in enumerate(ret123):
        if len(r) > 0:
            if nodat and r["dat"].size:
                raise ValueError("nodat=True on a table that holds its data columns: build it with "
                                 "postprocess(..., nodat=True)")
            writeTab(r, fbase + str(k) + ext, nodat=nodat, plain=plain)
    return ret123
