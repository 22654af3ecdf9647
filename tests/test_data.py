"""``Dataset`` (host pandas code): the reference's tests/test_data.py restated against this
package, with the same mocks, plus value checks the reference does not make."""
from unittest.mock import patch

import numpy as np
import pandas as pd
import pytest

from pybmc_amd import Dataset


@pytest.fixture
def sample():
    df = pd.DataFrame({"x": [1, 2, 3, 4], "y": [1, 2, 3, 4], "target": [10, 20, 30, 40],
                       "modelA": [9, 19, 29, 39], "modelB": [11, 21, 31, 41]})
    ds = Dataset(data_source="fake_path.h5")
    ds.data = {"target": df}
    ds.domain_keys = ["x", "y"]
    return ds, df


@patch("pybmc_amd.data.os.path.exists", return_value=True)
@patch("pybmc_amd.data.pd.read_csv")
def test_load_data_csv(mock_read_csv, mock_exists):  # reference tests/test_data.py:27-53
    mock_read_csv.return_value = pd.DataFrame(
        {"x": [1, 1, 2], "y": [1, 1, 2], "target": [10, 11, 20],
         "model": ["modelA", "modelB", "modelB"]})
    res = Dataset("fake_path.csv").load_data(["modelA", "modelB"], keys=["target"],
                                             domain_keys=["x", "y"], model_column="model")
    assert list(res["target"].columns) == ["x", "y", "modelA", "modelB"]
    assert res["target"].values.tolist() == [[1, 1, 10, 11]]      # inner join on the domain


@patch("pybmc_amd.data.os.path.exists", return_value=True)
@patch("pybmc_amd.data.pd.read_hdf")
def test_load_data_h5(mock_read_hdf, mock_exists):  # reference tests/test_data.py:55-76
    mock_read_hdf.side_effect = lambda file, key: pd.DataFrame(
        {"x": [1, 2], "y": [1, 2], "target": [10, 20]})
    ds = Dataset("fake_path.h5")
    res = ds.load_data(["modelA", "modelB"], keys=["target"], domain_keys=["x", "y"])
    assert isinstance(res["target"], pd.DataFrame)
    assert all(c in res["target"].columns for c in ["x", "y", "modelA", "modelB"])
    assert ds.data is res and ds.domain_keys == ["x", "y"]


@patch("pybmc_amd.data.os.path.exists", return_value=True)
@patch("pybmc_amd.data.pd.read_hdf")
def test_load_data_skips_models_without_the_property(mock_read_hdf, mock_exists, capsys):
    frames = {"a": pd.DataFrame({"x": [1], "y": [1], "BE": [5.0]}),
              "b": pd.DataFrame({"x": [1], "y": [1]})}
    mock_read_hdf.side_effect = lambda file, key: frames[key]
    res = Dataset("f.h5").load_data(["a", "b"], keys=["BE", "Rad"], domain_keys=["x", "y"])
    out = capsys.readouterr().out
    assert "[Skipped] Model 'b' missing columns ['BE'] for property 'BE'." in out
    assert "[Warning] No models with property 'Rad'" in out
    assert list(res["BE"].columns) == ["x", "y", "a"] and res["Rad"].empty


def test_load_data_errors():  # reference tests/test_data.py:78-92 and data.py:57-62
    with pytest.raises(ValueError, match="Data source must be specified"):
        Dataset().load_data(["m"], keys=["t"], domain_keys=["x"])
    with pytest.raises(FileNotFoundError):
        Dataset("/nonexistent/file.h5").load_data(["m"], keys=["t"], domain_keys=["x"])
    with patch("pybmc_amd.data.os.path.exists", return_value=True):
        with pytest.raises(ValueError, match="specify which properties"):
            Dataset("f.h5").load_data(["m"], domain_keys=["x"])
        with pytest.raises(ValueError, match="Unsupported file format"):
            Dataset("f.txt").load_data(["m"], keys=["t"], domain_keys=["x", "y"])
        with patch("pybmc_amd.data.pd.read_csv", return_value=pd.DataFrame({"x": [1], "y": [1]})):
            with pytest.raises(ValueError, match="Expected column 'model' not found in CSV"):
                Dataset("f.csv").load_data(["m"], keys=["t"], domain_keys=["x", "y"])


def test_split_data_random(sample):  # reference tests/test_data.py:94-105
    ds, df = sample
    tr, va, te = ds.split_data({"target": df}, "target", "random", train_size=0.6, val_size=0.2,
                               test_size=0.2)
    assert len(tr) + len(va) + len(te) == len(df)
    assert sorted(pd.concat([tr, va, te])["x"]) == [1, 2, 3, 4]
    again = ds.split_data({"target": df}, "target", "random", train_size=0.6, val_size=0.2,
                          test_size=0.2)
    assert tr.equals(again[0])                                    # random_state=1: repeatable
    with pytest.raises(ValueError):
        ds.split_data({"target": df}, "target", "random", train_size=0.6, val_size=0.2)
    with pytest.raises(ValueError):
        ds.split_data({"target": df}, "target", "random", train_size=0.6, val_size=0.3, test_size=0.3)
    with pytest.raises(ValueError):
        ds.split_data({"target": df}, "nope", "random")
    with pytest.raises(ValueError):
        ds.split_data({"target": df}, "target", "other")
    with pytest.raises(TypeError):
        ds.split_data({"target": [1, 2]}, "target", "random")


def test_split_data_inside_to_outside(sample):  # reference tests/test_data.py:107-127
    ds, df = sample
    coords = df[["x", "y"]].copy()
    tr, va, te = ds.split_data({"target": coords}, "target", "inside_to_outside",
                               stable_points=[(1, 1)], distance1=0.1, distance2=100)
    assert len(tr) + len(va) + len(te) == 4
    assert tr.values.tolist() == [[1, 1]] and len(va) == 3 and len(te) == 0


def test_separate_points_matches_the_double_loop():  # reference tests/test_data.py:153-171
    ds = Dataset()
    tr, va, te = ds.separate_points_distance_allSets([(1, 1), (2, 2)], [(1.1, 1.1), (3, 3)], 0.2, 1.5)
    assert (tr, va, te) == ([0], [1], [])
    rng = np.random.default_rng(0)
    a, b = rng.uniform(0, 10, (200, 2)), rng.uniform(0, 10, (7, 2))
    got = ds.separate_points_distance_allSets([tuple(p) for p in a], [tuple(p) for p in b], 1.0, 2.5)
    want = ([], [], [])
    for i, p in enumerate(a):                                      # the reference's rule
        d = min(np.linalg.norm(p - q) for q in b)
        want[0 if d <= 1.0 else 1 if d <= 2.5 else 2].append(i)
    assert got == want


def test_get_subset_and_view_data(sample):  # reference tests/test_data.py:129-151
    ds, df = sample
    r = ds.get_subset("target", filters={"x": lambda x: x > 2}, models_to_include=["modelA", "modelB"])
    assert list(r.columns) == ["modelA", "modelB"] and r["modelA"].tolist() == [29, 39]
    assert ds.get_subset("target", filters={"x": (2, 3)})["x"].tolist() == [2, 3]
    assert ds.get_subset("target", filters={"x": [1, 4]})["x"].tolist() == [1, 4]
    assert ds.get_subset("target", filters={"x": 2})["x"].tolist() == [2]
    assert ds.get_subset("target", filters={"multi": lambda row: row["x"] + row["y"] > 6})["x"].tolist() == [4]
    with pytest.raises(ValueError):
        ds.get_subset("nope")
    v = ds.view_data()
    assert v["available_properties"] == ["target"]
    assert v["available_models"] == ["modelA", "modelB", "target"]
    assert ds.view_data(property_name="target") is df
    assert ds.view_data("target", "modelA").tolist() == [9, 19, 29, 39]
    assert list(ds.view_data(model_name="modelB")["target"].columns) == ["x", "y", "modelB"]
    assert ds.view_data(model_name="zzz")["target"] == "[Model 'zzz' not available]"
    with pytest.raises(KeyError):
        ds.view_data("nope")
    with pytest.raises(KeyError):
        ds.view_data("target", "zzz")
    with pytest.raises(RuntimeError):
        Dataset().view_data()
