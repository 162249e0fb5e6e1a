"""Input-table reader and per-source set-up against the shipped tables."""
import numpy as np

from radex_emcee_amd import data_io, workloads


def test_flux_dat_sources():
    d = data_io.read_data()
    assert len(d) == 16 and list(d)[0] == "G09v1.97" and list(d)[-1] == "G15v2.779"
    z, lw, Jup, flux, eflux = data_io.get_source("G09v1.97", d)
    assert z == 3.6345 and lw == 348.3 and list(Jup) == [3, 4, 5, 6, 7]
    assert np.allclose(flux, [5.699, 7.8, 9.734, 9.979, 7.962]) and np.allclose(eflux, [2.248, 1.5, 1.188, 1.672, 0.915])
    z, lw, Jup, flux, eflux = data_io.get_source("SDP81", d)
    assert list(Jup) == [1, 3, 5, 8, 10]
    z, lw, Jup, flux, eflux = data_io.get_source("NAv1.195", d)
    assert list(Jup) == [5] and flux[0] == 9.89
    assert max(len(data_io.get_source(s, d)[2]) for s in d) == 7


def test_flux_for2p_has_dust_temperature():
    d = data_io.read_data(data_io.FLUX_2COMP)
    assert len(d) == 15 and "NAv1.195" not in d            # commented out in the 2-component table
    z, T_d, lw, Jup, flux, eflux = data_io.get_source("SDP81", d)
    assert (z, T_d, lw) == (3.0413, 34.0, 559.5) and list(Jup) == [1, 3, 5, 8, 10]


def test_source_setup():
    tbg, b = data_io.source_setup(3.6345)
    assert abs(tbg - 2.7315 * 4.6345) < 1e-12 and b.shape == (4, 2)
    assert abs(b[3, 0] - (-9.17931735162758 - 4)) < 1e-12 and b[1, 0] == np.log10(tbg)
    tbg, b = data_io.source_setup(2.5, ncomp=2)
    assert b.shape == (8, 2) and np.array_equal(b[:4], b[4:])
    assert np.array_equal(b, workloads.bounds_2comp(2.5))
