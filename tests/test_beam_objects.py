"""Object-level API of mcmc_amd/BeamFit against the reference's BeamModel.py and the plane
helpers of OffsetNormalizedBeamModel.py
(tests/golden/beamfuncs_g10.json, made by `oracle/make_golden.py beamfuncs` from the reference):
host-side scalars and packaging on the CPU (no library call), beam values and the driver
(ProcessBeamData.setup_sampler / sample1beam) on the GPU."""
import json
import os
import types

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def g10():
    with open(os.path.join(HERE, "golden", "beamfuncs_g10.json")) as f:
        return json.load(f)


def test_shape_scalars_packaging_and_plane_helpers(g10):
    """The packed inverse covariance against the reference's setParameters_MajMinAng + set_Cinv (the
    rebuilt class derives it directly from the rotation, so to rounding, not bit for bit);
    package / unpackage and the plane helpers bit for bit."""
    from mcmc_amd.BeamFit import BeamModel as BM
    from mcmc_amd.BeamFit import OffsetNormalizedBeamModel as ONB
    x, y = np.array(g10["x"]), np.array(g10["y"])
    for c in g10["cases"]:
        m = BM.GaussianBeamModel2D(tuple(c["center"]), tuple(c["sigmas"]), c["angle"])
        assert m.angle == c["angle_mod"]
        np.testing.assert_allclose(m.Cinv, c["Cinv"], rtol=1e-13, atol=1e-16)
        flat = BM.GaussianBeamModel2D.unpackage((c["center"], c["sigmas"], c["angle"]))
        np.testing.assert_array_equal(flat, c["flat5"])
        pk = BM.GaussianBeamModel2D.package(flat)
        assert [list(pk[0]), list(pk[1]), float(pk[2])] == c["pack5"]
        np.testing.assert_array_equal(ONB.planexy(0.93, 1.1, x, y), c["planexy"])
        np.testing.assert_array_equal(ONB.plane(0.93, 1.1, types.SimpleNamespace(x=x, y=y)), c["plane"])
    assert ONB.planexy(1, 0.3, 1.0, 2.0) == g10["planexy_mu1"] == 0


def test_range_setters():
    from mcmc_amd.BeamFit import BeamModel as BM

    class G(BM.GaussianBeamModel2D):
        pass
    G.setxyRange([-2.0, 6.0], [1.0, 3.0])
    assert (G.centerMin, G.centerMax) == ((-2.0, 1.0), (6.0, 3.0))
    G.setxyRange(types.SimpleNamespace(x=[-2.0, 6.0], y=[1.0, 3.0]), scale=0.5)
    assert (G.centerMin, G.centerMax) == ((0.0, 1.5), (4.0, 2.5))
    G.setsigMax([0.0, 5.0], [0.0, 9.0])
    assert G.sigMax == 9.0 and BM.GaussianBeamModel2D.sigMax is None


@pytest.mark.gpu
def test_beam_values_on_the_device(g10):
    """gauss2d / at / atxy through mi355_beam_gauss2d, to 1e-13 (exp of the reference's own
    exponent)."""
    from mcmc_amd.BeamFit import BeamModel as BM
    x, y = np.array(g10["x"]), np.array(g10["y"])
    np.testing.assert_allclose(BM.gauss2d(x, y, 0.5, -0.25, 0.04, -0.01, 0.09), g10["gauss2d"], rtol=1e-13)
    assert BM.gauss2d(0.5, -0.25, 0.5, -0.25, 0.04, -0.01, 0.09) == 1.0            # scalar in, scalar out
    data = types.SimpleNamespace(x=x, y=y)
    for c in g10["cases"]:
        m = BM.GaussianBeamModel2D(tuple(c["center"]), tuple(c["sigmas"]), c["angle"])
        np.testing.assert_allclose(m.atxy(x, y), c["atxy"], rtol=1e-12)
        np.testing.assert_allclose(m(data), c["atxy"], rtol=1e-12)


@pytest.mark.gpu
def test_prior_of_the_plain_class():
    from mcmc_amd.BeamFit import BeamModel as BM
    G = BM.GaussianBeamModel2D
    saved = (G.centerMin, G.centerMax, G.sigMin, G.sigMax)
    try:
        G.setxyRange([-20.0, 20.0])
        G.sigMin, G.sigMax = 3, 8
        assert G.prior((0.0, 1.0), (4.0, 5.0), 0.3) == 1
        assert G.prior((0.0, 21.0), (4.0, 5.0), 0.3) == 0           # centre outside the box
        assert G.prior((0.0, 1.0), (2.0, 5.0), 0.3) == 0            # below sigMin
        assert G.prior((0.0, 1.0), (4.0, 9.0), 0.3) == 0            # above sigMax
    finally:
        G.centerMin, G.centerMax, G.sigMin, G.sigMax = saved


@pytest.mark.gpu
def test_setup_sampler_matches_the_reference():
    """G12 (tests/golden/setup_sampler_g12.json): the reference's own ProcessBeamData.setup_sampler
    (:124-216; imported by oracle/make_golden.py with its Python-2 sibling imports aliased) on
    synthetic maps - class-level prior state, start point, proposal widths, and the likelihood at
    the start point - against the rebuilt driver over the device's BeamData.stats."""
    from oracle import refbeam as RB
    from mcmc_amd.BeamFit import OffsetNormalizedBeamModel as ONBM
    from mcmc_amd.BeamFit import ProcessBeamData as PBD
    from mcmc_amd.BeamFit.BeamData import BeamData
    with open(os.path.join(HERE, "golden", "setup_sampler_g12.json")) as f:
        g12 = json.load(f)
    mod = ONBM.OffsetNormalizedBeamModel
    saved = (mod.centerMin, mod.centerMax, mod.sigMin, mod.sigMax, mod.paramBlocks, mod.nBlock)
    try:
        for c in g12["cases"]:
            x, y, d, noise, truth, prior = RB.synth_beam_maps(1, n=c["n"], seed=c["seed"])
            assert float(np.abs(d[0]).sum()) == c["input_abs_sum"] and noise == c["noise"]
            data = BeamData(x, y, d[0], noise)
            like, prop_sigmas, start = PBD.setup_sampler([data], data, useNormalizedBeam=True,
                                                         rangeScale=c["rangeScale"],
                                                         sigminmax=tuple(c["sigminmax"]), verbose=False)
            assert [float(v) for v in mod.centerMin] == c["centerMin"]
            assert [float(v) for v in mod.centerMax] == c["centerMax"]
            assert (float(mod.sigMin), float(mod.sigMax)) == (c["sigMin"], c["sigMax"])
            assert list(mod.paramBlocks) == c["paramBlocks"] and mod.nBlock == c["nBlock"]
            # the moments behind the start point are summed on the device: to rounding
            np.testing.assert_allclose(np.asarray(mod.unpackage(start), float), c["start"], rtol=1e-12, atol=1e-13)
            np.testing.assert_allclose(np.asarray(mod.unpackage(prop_sigmas), float), c["prop_sigmas"], rtol=1e-12)
            assert mod.prior(*start) == c["prior_at_start"]
            lp = like.lnLike(start)
            assert abs(lp - c["lnlike_at_start"]) <= 1e-11 * abs(c["lnlike_at_start"])
    finally:
        (mod.centerMin, mod.centerMax, mod.sigMin, mod.sigMax, mod.paramBlocks, mod.nBlock) = saved


@pytest.mark.gpu
def test_setup_sampler_and_sample1beam():
    """The beam driver end to end: setup_sampler (pinned by G12 above; here its pieces against the
    oracle's BeamData.stats), then sample1beam recovers the map's parameters."""
    from oracle import refbeam as RB
    from mcmc_amd.BeamFit import OffsetNormalizedBeamModel as ONBM
    from mcmc_amd.BeamFit import ProcessBeamData as PBD
    from mcmc_amd.BeamFit.BeamData import BeamData
    mod = ONBM.OffsetNormalizedBeamModel
    saved = (mod.centerMin, mod.centerMax, mod.sigMin, mod.sigMax, mod.paramBlocks, mod.nBlock)
    try:
        x, y, d, noise, truth, prior = RB.synth_beam_maps(1, n=64, seed=8)
        data = BeamData(x, y, d[0], noise)
        like, prop_sigmas, start = PBD.setup_sampler([data], data, useNormalizedBeam=True, verbose=False)
        st = RB.beam_stats(RB.BeamMap(x, y, d[0], noise), sigcut=0.0)
        sig0 = np.maximum(np.minimum([np.sqrt(st[2]), np.sqrt(st[3])], 8), 3)
        np.testing.assert_allclose(start[0], st[:2], rtol=1e-10)
        np.testing.assert_allclose(start[1], sig0, rtol=1e-10)
        assert start[2] == 0 and start[4] == (0, 1, 0)
        assert start[3] == (d[0].max() - d[0].min()) / 2
        ctr = d[0][(abs(x - start[0][0]) < sig0[0]) & (abs(y - start[0][1]) < sig0[1])]
        assert prop_sigmas[3] == 2 * (ctr.max() - ctr.min()) and prop_sigmas[4] == (0.3, 0, 0)
        delx = x.max() - x.min()
        np.testing.assert_allclose(prop_sigmas[0], (delx / 10, delx / 10))
        assert (mod.sigMin, mod.sigMax, mod.nBlock) == (3, 8, 7)
        assert mod.prior(*start) == 1
        stages, ana = PBD.sample1beam(like, prop_sigmas, start, nMC=(1500, 1500), doBlock=True,
                                      seed=11, verbose=False)
        last = stages[-1]
        lp = like.lnLike(mod.package(last.samples[-1]))
        assert abs(last.lnPr[-1] - lp) <= 1e-10 * abs(lp)
        mean = np.asarray(mod.unpackage(mod.package(ana[0])))
        t = truth[0]
        assert abs(mean[0] - t[0]) < 0.2 and abs(mean[1] - t[1]) < 0.2     # centre
        assert abs(mean[5] - t[5]) < 0.5                                   # amplitude (truth 10)
        with pytest.raises(NotImplementedError):
            PBD.setup_sampler([data], data, useNormalizedBeam=False)
    finally:
        (mod.centerMin, mod.centerMax, mod.sigMin, mod.sigMax, mod.paramBlocks, mod.nBlock) = saved
