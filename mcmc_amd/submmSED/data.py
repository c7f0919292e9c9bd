"""SED photometry containers with the interface of the reference's MCMC/submmSED/data.py.
Host-side I/O only (the hot path starts at mi355_sed_load)."""
import numpy as np

from ..GaussianData import GaussianData

speed_of_light = 299792.458   # micron GHz      data.py:28


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


def read_wavelength_table(filename, z, name, UL25=True):
    """A (wavelength[um], flux[Jy], error[Jy]) text table such as the reference's
    txt/Arp220.txt -> submmData, following the Arp220 branch of readfluxes_DLC_2014
    (data.py:243-256): nu_obs = c/lambda, the 25 um point becomes an upper limit
    (sig = flux, flux = 0) when UL25, nu_rest = nu_obs (1+z)."""
    lam, flux, sig = np.loadtxt(filename, unpack=True, usecols=(0, 1, 2))
    lam, flux, sig = (np.ascontiguousarray(a, dtype=np.float64) for a in (lam, flux, sig))
    nu_obs = speed_of_light / lam
    if UL25:
        l25 = np.where(np.round(lam) == 25)[0]
        if len(l25):
            sig[l25] = flux[l25]
            flux[l25] = 0
    return submmData(nu_obs * (1 + z), flux, sig, name, z, nu_obs=nu_obs)


def readfluxes_Arp220(filename, UL25=True):
    return [read_wavelength_table(filename, 0.018, "Arp220", UL25=UL25)]
