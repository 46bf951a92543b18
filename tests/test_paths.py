"""CPU: the §8(f) adapter against goldens captured FROM THE REFERENCE (tests/golden/G4.json,
written by oracle/make_golden_paths.py importing GPmap.py on a synthetic CSV): CSV parse
rules, the signed travel-sum validity quirk, and the oracle's restatement of
calc_distance / calc_mean_traj."""
import json
import os

import numpy as np
import pytest

from gaussianprocesspathmodelling_amd import paths as gpaths
from oracle import gp_oracle


@pytest.fixture(scope="module")
def g4(golden_dir):
    return json.load(open(os.path.join(golden_dir, "G4.json")))


def test_read_csv_keeps_what_the_reference_keeps(g4):
    t = gpaths.read_csv(g4["csv"], numoftrajstoread=10)
    assert t.keys() == g4["kept_ids"]                   # INWARD and SHORT are dropped, order kept
    for k in g4["kept_ids"]:
        assert np.array_equal(t.pathdict[k].timestamp, g4["paths"][k]["t"])
        assert np.array_equal(t.pathdict[k].xs, g4["paths"][k]["x"])
        assert np.array_equal(t.pathdict[k].ys, g4["paths"][k]["y"])
        assert t.pathdict[k].xs.dtype == np.float64      # parsed as int, stored as float
    assert len(gpaths.read_csv(g4["csv"])) == 12          # 0 = read everything
    assert len(gpaths.read_csv(g4["csv"], 3)) == 3


def test_read_csv_edge_cases(tmp_path):
    assert len(gpaths.read_csv("")) == 0
    with pytest.raises(ValueError):
        gpaths.read_csv("###,,,\n")
    # last block without a closing ### row is dropped, like the reference
    rows = "\n".join(f"{i},0,{1000 * i},{1000 * i}" for i in range(33))
    assert len(gpaths.read_csv("h,A,x,y\n" + rows + "\n")) == 0
    assert len(gpaths.read_csv("h,A,x,y\n" + rows + "\n###,,,\n")) == 1
    f = tmp_path / "t.csv"
    f.write_text("h,A,x,y\n" + rows + "\n###,,,\n")
    assert gpaths.read_csv(str(f)).keys() == ["A"]


def test_validity_is_the_signed_sum_quirk(g4):
    inward = gpaths.Trajectory()
    for i in range(33):
        inward.add_point(float(i), 30000 - 500 * i, 30000 - 500 * i)
    t = gpaths.read_csv(g4["csv"], 10)
    first = t.pathdict[g4["kept_ids"][0]]
    assert gpaths.check_if_valid_trajectory(inward, 1000) is g4["valid"]["inward_1000"] is False
    assert gpaths.check_if_valid_trajectory(first, 1000) is g4["valid"]["first_kept_1000"]
    assert gpaths.check_if_valid_trajectory(first) is g4["valid"]["first_kept_default"]
    # closed form == the reference's double loop
    xs, ys = np.abs(first.xs), np.abs(first.ys)
    loop = sum((xs[j] - xs[i]) + (ys[j] - ys[i]) for i in range(33) for j in range(i + 1, 33))
    assert abs(gpaths.travel_score(first) - loop) <= 1e-9 * abs(loop)
    assert abs(gp_oracle.travel_score(first.xs, first.ys) - loop) <= 1e-9 * abs(loop)


def test_oracle_distance_and_mean_match_reference_values(g4):
    P = g4["paths"]
    for a, b, ref in g4["distance_pairs"]:
        got = gp_oracle.path_distance(P[a]["x"], P[a]["y"], P[b]["x"], P[b]["y"])
        assert abs(got - ref) <= 1e-12 * ref
    arr = np.array([[P[k]["t"], P[k]["x"], P[k]["y"]] for k in g4["mean_of"]]).transpose(0, 2, 1)
    m = gp_oracle.mean_path(arr)
    for c, name in enumerate("txy"):
        assert np.allclose(m[:, c], g4["mean_traj"][name], rtol=1e-14, atol=0)
    assert np.allclose(gpaths.mean_path(arr), m, rtol=1e-15)
    with pytest.raises(ZeroDivisionError):
        gpaths.mean_path(np.zeros((0, 33, 3)))


def test_to_gp_inputs(g4):
    t = gpaths.read_csv(g4["csv"], 10)
    X, Y, (lo, span) = gpaths.to_gp_inputs(t, inputs=("t",), targets=("x", "y"))
    assert X.shape == (330, 1) and Y.shape == (330, 2)
    assert X.min() == 0.0 and X.max() == 1.0
    X3, Y0, _ = gpaths.to_gp_inputs(t, keys=t.keys()[:2], inputs=("t", "x", "y"), targets=(), normalise=False)
    assert X3.shape == (66, 3) and Y0.shape == (66, 0)
    assert np.array_equal(X3[:33, 1], t.pathdict[t.keys()[0]].xs)


def test_fit_path_models_argument_checks(g4):
    """Host-side checks of the per-cluster model front end run before any GPU call."""
    t = gpaths.read_csv(g4["csv"])
    with pytest.raises(KeyError):
        gpaths.fit_path_models(t, {0: ["no-such-path"]})
    with pytest.raises(ValueError):
        gpaths.fit_path_models(t, {0: t.keys()[:2]}, devices=[])
    with pytest.raises(ValueError):
        gpaths.fit_path_models(t, {0: t.keys()[:2]}, devices=0)
    assert gpaths.fit_path_models(t, {0: [], 1: []}) == {}
