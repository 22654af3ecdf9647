"""f4 parity: ``pybmc_amd.data.Dataset`` against outputs of the reference's ``Dataset``
(pybmc/data.py:30-129 load/align, :131-192 view, :194-245 distance split, :247-330 splits with
random_state = 1, :332-374 subset filters) on the committed stand-in CSV.  The fixture
``tests/golden/dataset_standin.npz`` was written by ``tests/golden/make_golden.py dataset``
importing the unmodified reference; everything here must match it exactly (frames value for
value, index for index, column for column)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from pybmc_amd import Dataset

CSV = os.path.join(GOLDEN, "dataset_standin.csv")
MODELS = ["truth", "FRDM", "HFB24", "UNEDF1", "SKM"]
STABLE = [(26, 24), (30, 28), (34, 30)]


@pytest.fixture(scope="module")
def loaded():
    ds = Dataset(CSV)
    return ds, ds.load_data(MODELS, keys=["BE", "Rad"], domain_keys=["N", "Z"]), \
        load_golden("dataset_standin")


def same_frame(df, g, prefix):
    assert list(df.columns) == list(g[prefix + "_columns"])
    assert np.array_equal(df.index.to_numpy(), g[prefix + "_index"])
    assert np.array_equal(df.to_numpy(float), g[prefix + "_values"])


def test_load_and_align(loaded):
    ds, data, g = loaded
    assert list(data) == ["BE", "Rad"]
    for prop in data:
        same_frame(data[prop], g, f"load_{prop}")
    assert len(data["BE"]) < 144          # the inner join really dropped nuclei


def test_view(loaded):
    ds, data, g = loaded
    v = ds.view_data()
    assert v["available_properties"] == list(g["view_properties"])
    assert v["available_models"] == list(g["view_models"])
    assert np.array_equal(ds.view_data(model_name="FRDM")["BE"].to_numpy(float), g["view_model_BE"])
    assert np.array_equal(ds.view_data("Rad", "SKM").to_numpy(float), g["view_series"])


def test_random_split_uses_the_reference_seed(loaded):
    ds, data, g = loaded
    tr, va, te = ds.split_data(data, "BE", splitting_algorithm="random",
                               train_size=0.6, val_size=0.2, test_size=0.2)
    assert np.array_equal(tr.index.to_numpy(), g["random_train"])
    assert np.array_equal(va.index.to_numpy(), g["random_val"])
    assert np.array_equal(te.index.to_numpy(), g["random_test"])
    assert np.array_equal(tr.to_numpy(float), g["random_train_values"])


def test_distance_split(loaded):
    ds, data, g = loaded
    dom = {"dom": data["BE"][["N", "Z"]]}
    tr, va, te = ds.split_data(dom, "dom", splitting_algorithm="inside_to_outside",
                               stable_points=STABLE, distance1=2.0, distance2=4.5)
    assert np.array_equal(tr.index.to_numpy(), g["dist_train"])
    assert np.array_equal(va.index.to_numpy(), g["dist_val"])
    assert np.array_equal(te.index.to_numpy(), g["dist_test"])
    pts = [tuple(r) for r in data["BE"][["N", "Z"]].to_numpy()[:60]]
    a, b, c = ds.separate_points_distance_allSets(pts, STABLE, 1.5, 3.0)
    assert a == list(g["sep_a"]) and b == list(g["sep_b"]) and c == list(g["sep_c"])


@pytest.mark.parametrize("name,kw", [
    ("range", dict(filters={"Z": (22, 27)})),
    ("list", dict(filters={"N": [24, 25, 30, 41]})),
    ("value", dict(filters={"Z": 25})),
    ("callable", dict(filters={"N": lambda s: s % 2 == 0})),
    ("multi", dict(filters={"multi": lambda r: r["N"] - r["Z"] >= 6, "Z": (21, 30)})),
    ("models", dict(filters={"Z": (20, 24)}, models_to_include=["FRDM", "SKM", "nope"])),
])
def test_subsets(loaded, name, kw):
    ds, data, g = loaded
    same_frame(ds.get_subset("BE", **kw), g, f"subset_{name}")
