"""CPU tests of the result layout / summary statistics."""
import numpy as np

from radex_emcee_amd import fit


def test_result_tuple_layout_and_pickle_roundtrip(tmp_path):
    chain = np.zeros((5, 8, 4)); lnp = np.zeros((5, 8))
    Jup, f, e = np.array([1, 3]), np.array([1.0, 2.0]), np.array([0.1, 0.2])
    b = np.zeros((4, 2))
    t = fit.result_tuple("SDP81", 3.04, b, Jup, f, e, np.ones(4), None, np.ones(4), np.ones(4), chain, lnp)
    assert len(t) == 8 and t[0] == "SDP81" and t[3][0] is Jup and t[7][0].shape == (5, 8, 4)
    t2 = fit.result_tuple("SDP81", 3.04, np.zeros((8, 2)), Jup, f, e, np.ones(8), None, np.ones(8), np.ones(8),
                          chain, lnp, T_d=34.0)
    assert len(t2) == 9 and t2[3] == 34.0 and t2[4][1] is f       # T_d inserted after bounds
    p = tmp_path / "x.pickle"
    fit.save_result(p, t)
    back = fit.load_result(p)
    assert back[0] == "SDP81" and np.array_equal(back[7][1], lnp)


def test_summary_percentiles():
    rng = np.random.RandomState(0)
    flat = rng.randn(20000, 4) * np.array([0.5, 0.1, 0.3, 1.0]) + np.array([3.0, 2.0, 17.5, -9.0])
    s = fit.summarize(flat)[0]
    assert abs(s["n_H2"][0] - 3.0) < 0.02 and abs(s["n_H2"][1] - 0.5) < 0.03 and abs(s["n_H2"][2] - 0.5) < 0.03
    assert abs(s["P"][0] - 5.0) < 0.03                            # log P = log n + log T
    s2 = fit.summarize(np.hstack([flat, flat + 1.0]), ncomp=2)
    assert abs(s2[1]["T_kin"][0] - 3.0) < 0.02
