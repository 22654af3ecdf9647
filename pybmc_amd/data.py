"""``Dataset``: host-side loading, alignment and splitting of model-prediction tables
(reference pybmc/data.py).  Pure pandas bookkeeping that runs once
before the sampler; nothing here touches the GPU.  Same methods, arguments, return values,
prints and exception types as the reference."""
from __future__ import annotations

import os

import numpy as np
import pandas as pd


def apply_domain_filters(df, filters, between_inclusive=True):
    """Row filters shared by ``Dataset.get_subset`` (reference data.py:353-365) and
    ``BayesianModelCombination.evaluate`` (reference bmc.py:352-364): ``{"multi": row
    predicate}``, ``{column: predicate on the column}``, ``(lo, hi)`` inclusive range, list of
    allowed values, or a single value."""
    for column, cond in (filters or {}).items():
        if column == "multi" and callable(cond):
            df = df[df.apply(cond, axis=1)]
        elif callable(cond):
            df = df[cond(df[column])]
        elif isinstance(cond, tuple) and len(cond) == 2:
            df = df[(df[column] >= cond[0]) & (df[column] <= cond[1])]
        elif isinstance(cond, list):
            df = df[df[column].isin(cond)]
        else:
            df = df[df[column] == cond]
    return df


class Dataset:
    """Loads per-model prediction tables (HDF5: one key per model; CSV: one ``model`` column),
    aligns them on the domain columns, and splits / filters the result
    (reference data.py:7-28)."""

    def __init__(self, data_source=None):
        self.data_source = data_source
        self.data = {}
        self.domain_keys = ["X1", "X2"]

    # ------------------------------------------------------------------ loading
    def _model_frames(self, models, prop, domain_keys, model_column):
        """(frames with columns domain_keys + [model], skipped models) for one property."""
        src = self.data_source
        frames, skipped = [], []
        if src.endswith(".h5"):
            tables = ((m, pd.read_hdf(src, key=m)) for m in models)
            kind = "property"
        elif src.endswith(".csv"):
            table = pd.read_csv(src)

            def per_model():
                for m in models:
                    if model_column not in table.columns:
                        raise ValueError(f"Expected column '{model_column}' not found in CSV.")
                    yield m, table[table[model_column] == m]
            tables = per_model()
            kind = "key"
        else:
            raise ValueError("Unsupported file format. Only .h5 and .csv are supported.")
        for model, df in tables:
            missing = [c for c in domain_keys + [prop] if c not in df.columns]
            if missing:
                print(f"[Skipped] Model '{model}' missing columns {missing} for {kind} '{prop}'.")
                skipped.append(model)
                continue
            frames.append(df[domain_keys + [prop]].rename(columns={prop: model}))
        return frames, skipped

    def load_data(self, models, keys=None, domain_keys=None, model_column="model"):
        """One inner-joined DataFrame per property in ``keys`` with the domain columns and one
        column per model (reference data.py:30-129).  ``ValueError`` without a data source or
        ``keys``; ``FileNotFoundError`` for a missing file."""
        self.domain_keys = domain_keys
        if self.data_source is None:
            raise ValueError("Data source must be specified.")
        if not os.path.exists(self.data_source):
            raise FileNotFoundError(f"Data source '{self.data_source}' not found.")
        if keys is None:
            raise ValueError("You must specify which properties to extract via 'keys'.")
        result = {}
        for prop in keys:
            frames, skipped = self._model_frames(models, prop, domain_keys, model_column)
            if not frames:
                print(f"[Warning] No models with property '{prop}'. "
                      "Resulting DataFrame will be empty.")
                result[prop] = pd.DataFrame(
                    columns=domain_keys + [m for m in models if m not in skipped])
                continue
            merged = frames[0]
            for other in frames[1:]:
                merged = pd.merge(merged, other, on=domain_keys, how="inner")
            result[prop] = merged
            self.data = result
        return result

    # ------------------------------------------------------------------- viewing
    def view_data(self, property_name=None, model_name=None):
        """Overview dict, per-model dict, property frame or one model's column
        (reference data.py:131-192)."""
        if not self.data:
            raise RuntimeError("No data loaded. Run `load_data(...)` first.")
        if property_name is None and model_name is None:
            models = sorted({c for df in self.data.values() for c in df.columns
                             if c not in self.domain_keys})
            return {"available_properties": list(self.data.keys()), "available_models": models}
        if property_name is None:
            return {prop: (df[self.domain_keys + [model_name]] if model_name in df.columns
                           else f"[Model '{model_name}' not available]")
                    for prop, df in self.data.items()}
        if property_name not in self.data:
            raise KeyError(f"Property '{property_name}' not found.")
        df = self.data[property_name]
        if model_name is None:
            return df
        if model_name not in df.columns:
            raise KeyError(f"Model '{model_name}' not found in property '{property_name}'.")
        return df[model_name]

    # ----------------------------------------------------------------- splitting
    def separate_points_distance_allSets(self, list1, list2, distance1, distance2):
        """Indices of ``list1`` within ``distance1`` of any reference point, within
        ``distance2`` only, and beyond (reference data.py:194-245).  All pairwise distances in
        one broadcast instead of the reference's Python double loop; same Euclidean rule."""
        a = np.asarray(list1, dtype=float).reshape(len(list1), -1)
        b = np.asarray(list2, dtype=float).reshape(len(list2), -1)
        near1 = np.zeros(len(a), dtype=bool)
        near2 = np.zeros(len(a), dtype=bool)
        if len(b):
            step = max(1, 4_000_000 // max(len(b), 1))
            for s in range(0, len(a), step):
                d = np.linalg.norm(a[s:s + step, None, :] - b[None, :, :], axis=2)
                near1[s:s + step] = (d <= distance1).any(axis=1)
                near2[s:s + step] = (d <= distance2).any(axis=1)
        idx = np.arange(len(a))
        return (idx[near1].tolist(), idx[~near1 & near2].tolist(), idx[~near1 & ~near2].tolist())

    def split_data(self, data_dict, property_name, splitting_algorithm="random", **kwargs):
        """(train, validation, test) frames (reference data.py:247-330): ``"random"`` with
        ``train_size/val_size/test_size`` (sklearn, ``random_state=1``) or
        ``"inside_to_outside"`` with ``stable_points/distance1/distance2``."""
        if property_name not in data_dict:
            raise ValueError(
                f"Property '{property_name}' not found in the provided data dictionary.")
        data = data_dict[property_name]
        if not isinstance(data, pd.DataFrame):
            raise TypeError("Data for the specified property must be a pandas DataFrame.")
        table = data.reset_index(drop=True)
        if splitting_algorithm == "random":
            need = ["train_size", "val_size", "test_size"]
            if any(k not in kwargs for k in need):
                raise ValueError(f"Missing required kwargs for 'random': {need}")
            tr, va, te = (kwargs[k] for k in need)
            if not np.isclose(tr + va + te, 1.0):
                raise ValueError("train_size + val_size + test_size must equal 1.0")
            from sklearn.model_selection import train_test_split
            train_idx, rest = train_test_split(table.index, train_size=tr, random_state=1)
            val_idx, test_idx = train_test_split(rest, test_size=1 - va / (va + te), random_state=1)
        elif splitting_algorithm == "inside_to_outside":
            need = ["stable_points", "distance1", "distance2"]
            if any(k not in kwargs for k in need):
                raise ValueError(f"Missing required kwargs for 'inside_to_outside': {need}")
            points = list(table.itertuples(index=False, name=None))
            train_idx, val_idx, test_idx = self.separate_points_distance_allSets(
                points, kwargs["stable_points"], kwargs["distance1"], kwargs["distance2"])
        else:
            raise ValueError("splitting_algorithm must be either 'random' or 'inside_to_outside'")
        return table.iloc[train_idx], table.iloc[val_idx], table.iloc[test_idx]

    # ----------------------------------------------------------------- filtering
    def get_subset(self, property_name, filters=None, models_to_include=None):
        """Filtered copy of one property's frame, optionally restricted to some models plus
        the N/Z domain columns (reference data.py:332-374)."""
        if property_name not in self.data:
            raise ValueError(f"Property '{property_name}' not found in dataset.")
        df = apply_domain_filters(self.data[property_name].copy(), filters)
        if models_to_include is not None:
            keep = [c for c in ["N", "Z"] if c in df.columns]
            keep += [m for m in models_to_include if m in df.columns]
            df = df[keep]
        return df
