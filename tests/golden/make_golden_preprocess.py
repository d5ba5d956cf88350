"""Golden vectors for the pose-table preprocessing, produced by running the REFERENCE's own
scale_table / _pp_* functions (deepof/utils.py:2342-3027) in place.  Build container only:
``python tests/golden/make_golden_preprocess.py`` -> tests/golden/preprocess.npz (inputs + expected outputs)."""
import json
import os
import sys
from itertools import combinations

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_import import load_reference_preprocessing  # noqa: E402

U = load_reference_preprocessing()


class Tables(dict):
    """Just enough of deepof.data.TableDict for the _pp_* functions: a dict of DataFrames + two attributes."""

    def __init__(self, d=None, typ="merged", table_path=""):
        super().__init__(d or {})
        self._type, self._table_path = typ, table_path


def labels(bodyparts, n_angles):
    cols = []
    for bp in bodyparts:
        cols += [(bp, "x"), (bp, "y")]
    cols += list(bodyparts)
    cols += [tuple(p) for p in combinations(bodyparts, 2)]
    cols += [(bodyparts[i], bodyparts[i + 1], bodyparts[i + 2]) for i in range(n_angles)]
    return cols


def synth_table(rng, n_frames, cols, bodyparts, size):
    """Smooth random walk per body part, speeds, true pairwise distances, angles; then defects."""
    pos = {bp: np.cumsum(rng.standard_normal((n_frames, 2)) * 2.0, axis=0) + rng.uniform(-40, 40, 2) * size for bp in bodyparts}
    tab = np.zeros((n_frames, len(cols)))
    for j, c in enumerate(cols):
        if isinstance(c, tuple) and len(c) == 2 and c[1] in ("x", "y"):
            tab[:, j] = pos[c[0]][:, 0 if c[1] == "x" else 1]
        elif isinstance(c, str):
            tab[:, j] = np.r_[0.0, np.hypot(*np.diff(pos[c], axis=0).T)]
        elif len(c) == 2:
            tab[:, j] = np.hypot(*(pos[c[0]] - pos[c[1]]).T)
        else:
            tab[:, j] = rng.uniform(0, np.pi, n_frames)
    return tab


def damage(rng, tab, cols):
    n, c = tab.shape
    tab[:3, 1] = np.nan                       # leading gap
    tab[-4:, 2] = np.nan                      # trailing gap
    tab[n // 3: n // 3 + 7, 0] = np.nan       # interior gaps
    for _ in range(12):
        r, j, g = rng.integers(1, n - 6), rng.integers(0, c), rng.integers(1, 5)
        tab[r:r + g, j] = np.nan
    for _ in range(5):                        # outliers -> clipped to NaN after standardisation
        r, j = rng.integers(0, n), rng.integers(0, c)
        tab[r, j] = tab[r, j] * 60.0 + 500.0
    d = [j for j, cc in enumerate(cols) if isinstance(cc, tuple) and len(cc) == 2 and cc[1] not in ("x", "y")]
    tab[n // 2, d[0]] = -3.0                  # negative distance -> clamped before log1p
    return tab


def run_reference(tables, cols, animal_ids, kw, pretrained=None):
    index = pd.Index(cols, tupleize_cols=False)
    td = Tables({k: pd.DataFrame(v.copy(), columns=index) for k, v in tables.items()})
    keys = sorted(td.keys())
    bins = {k: np.arange(len(td[k])) for k in keys}
    modes = dict(dist_standardize=kw["dist"], speed_standardize=kw["speed"], coord_standardize=kw["coord"],
                 log_distances=kw["log"])
    valid, samples, _ = U._pp_pass1_collect_samples(td, keys_list=keys, animal_ids=animal_ids, bin_info=bins,
                                                     samples_max=kw["samples_max"], scale="standard", pretrained_scaler=pretrained,
                                                     filter_low_variance=False, quality_to_load=None, **modes)
    gs = U._pp_fit_global_scaler(scale="standard", pretrained_scaler=pretrained, samples=samples, **modes)
    out = U._pp_pass2_scale_and_save(td, coordinates=None, valid_keys=valid, bin_info=bins, animal_ids=animal_ids,
                                     scale="standard", global_scaler=gs, filter_low_variance=False,
                                     interpolate_normalized=kw["clip"], file_name="pp", save_as_paths=False,
                                     quality_to_load=None, **modes)
    return {k: out[k].to_numpy(float) for k in valid}, gs


def main():
    store = {}
    cases = []
    rng = np.random.default_rng(7)
    pair_bps = ["B_Nose", "B_Center", "B_Tail_base", "B_Left_ear", "W_Nose", "W_Center", "W_Tail_base", "W_Right_ear"]
    single_bps = ["Nose", "Left_ear", "Right_ear", "Center", "Tail_base", "Tail_tip"]
    data = {}
    for tag, bps, aids, n_ang, lens in [("pair", pair_bps, ["B", "W"], 2, (60, 45, 80, 20)),
                                        ("single", single_bps, [""], 0, (50, 70))]:
        cols = labels(bps, n_ang)
        tabs = {}
        for v, n in enumerate(lens):
            t = damage(rng, synth_table(rng, n, cols, bps, 1.0 + 0.5 * v), cols)
            tabs[f"vid{v}"] = t
        if tag == "pair":
            tabs["vid1"][:, 5] = np.nan            # a column that is all-NaN in one video
            tabs["vid3"][:] = np.nan               # a table that is skipped altogether
        data[tag] = (cols, aids, tabs)
        store[f"{tag}::columns"] = np.array(json.dumps([list(c) if isinstance(c, tuple) else c for c in cols]))
        store[f"{tag}::animal_ids"] = np.array(json.dumps(aids))
        for k, t in tabs.items():
            store[f"{tag}::raw::{k}"] = t
    grid = [("pair", "gw", dict(dist="groupwise", speed="groupwise", coord="groupwise", log=True, samples_max=227272, clip=10)),
            ("pair", "pc", dict(dist="per_column", speed="per_column", coord="per_column", log=True, samples_max=227272, clip=10)),
            ("pair", "mixed", dict(dist="groupwise", speed=None, coord="per_column", log=False, samples_max=227272, clip=4)),
            ("pair", "sub", dict(dist="per_column", speed="groupwise", coord="groupwise", log=True, samples_max=30, clip=10)),
            ("pair", "noclip", dict(dist=None, speed="per_column", coord=None, log=True, samples_max=227272, clip=0)),
            ("single", "gw", dict(dist="groupwise", speed="groupwise", coord="groupwise", log=True, samples_max=227272, clip=10)),
            ("single", "pc", dict(dist="per_column", speed="per_column", coord="per_column", log=True, samples_max=40, clip=10))]
    for tag, name, kw in grid:
        cols, aids, tabs = data[tag]
        out, gs = run_reference(tabs, cols, aids, kw)
        case = f"{tag}::{name}"
        cases.append(dict(case=case, data=tag, **kw))
        for k, t in out.items():
            store[f"{case}::out::{k}"] = t
        for part in ("speed", "dist", "dist_inner", "dist_intra", "coord"):
            if gs is not None and gs.get(part) is not None:
                store[f"{case}::scaler::{part}::mean"] = np.atleast_1d(gs[part].mean_)
                store[f"{case}::scaler::{part}::scale"] = np.atleast_1d(gs[part].scale_)
        if name == "gw" and tag == "pair":       # the fitted scalers re-applied to other videos (pretrained_scaler path)
            other = {k: damage(rng, synth_table(rng, 40 + 9 * i, cols, pair_bps, 0.8 + i), cols) for i, k in enumerate(["new0", "new1"])}
            out2, _ = run_reference(other, cols, aids, kw, pretrained=gs)
            for k, t in other.items():
                store[f"pair::pre::raw::{k}"] = t
            for k, t in out2.items():
                store[f"pair::pre::out::{k}"] = t
    # scale_table on its own (per-video statistics only), incl. size factors via standardize=False
    cols, aids, tabs = data["pair"]
    df = pd.DataFrame(tabs["vid0"].copy(), columns=pd.Index(cols, tupleize_cols=False))
    for nm, kw in [("size_only", dict(standardize=False)), ("geom", dict(inter_scale="geom", standardize=False)),
                   ("full_pc", dict()), ("full_gw", dict(dist_standardize="groupwise", speed_standardize="groupwise",
                                                         coord_standardize="groupwise")), ("infer_ids", dict(animal_ids=None))]:
        kw = dict(kw)
        aid = kw.pop("animal_ids", aids)
        store[f"scale_table::{nm}"] = U.scale_table(df, scale="standard", animal_ids=aid, **kw).to_numpy(float)
    store["cases"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(HERE, "preprocess.npz"), **store)
    print("preprocess.npz", os.path.getsize(os.path.join(HERE, "preprocess.npz")), "bytes,", len(cases), "cases")


if __name__ == "__main__":
    main()
