"""Golden vectors for the remaining rows of the pose-table preprocessing: ``scale="minmax"``
(_pp_make_scaler, deepof/utils.py:2570) and ``filter_low_variance`` (_pp_filter_low_variance, utils.py:2604), produced by
running the REFERENCE's own scale_table / _pp_* functions in place.  Build container only:
``python tests/golden/make_golden_preprocess_r03.py`` -> tests/golden/preprocess_r03.npz (inputs + expected outputs)."""
import json
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_preprocess as G  # noqa: E402  (the synthetic tables and the TableDict stand-in of the round-1 fixture)

U = G.U


def run_reference(tables, cols, animal_ids, kw, scale, pretrained=None, filter_low_variance=False):
    index = pd.Index(cols, tupleize_cols=False)
    td = G.Tables({k: pd.DataFrame(v.copy(), columns=index) for k, v in tables.items()})
    keys = sorted(td.keys())
    bins = {k: np.arange(len(td[k])) for k in keys}
    modes = dict(dist_standardize=kw["dist"], speed_standardize=kw["speed"], coord_standardize=kw["coord"],
                 log_distances=kw["log"])
    valid, samples, _ = U._pp_pass1_collect_samples(td, keys_list=keys, animal_ids=animal_ids, bin_info=bins,
                                                     samples_max=kw["samples_max"], scale=scale, pretrained_scaler=pretrained,
                                                     filter_low_variance=filter_low_variance, quality_to_load=None, **modes)
    gs = U._pp_fit_global_scaler(scale=scale, pretrained_scaler=pretrained, samples=samples, **modes)
    out = U._pp_pass2_scale_and_save(td, coordinates=None, valid_keys=valid, bin_info=bins, animal_ids=animal_ids,
                                     scale=scale, global_scaler=gs, filter_low_variance=filter_low_variance,
                                     interpolate_normalized=kw["clip"], file_name="pp", save_as_paths=False,
                                     quality_to_load=None, **modes)
    return {k: out[k] for k in valid}, gs


def store_scaler(store, case, gs, scale):
    for part in ("speed", "dist", "dist_inner", "dist_intra", "coord"):
        if gs is not None and gs.get(part) is not None:
            sc = gs[part]
            if scale == "minmax":
                store[f"{case}::scaler::{part}::data_min"] = np.atleast_1d(sc.data_min_)
                store[f"{case}::scaler::{part}::data_range"] = np.atleast_1d(sc.data_range_)
            elif scale == "robust":
                store[f"{case}::scaler::{part}::center"] = np.atleast_1d(sc.center_)
                store[f"{case}::scaler::{part}::scale"] = np.atleast_1d(sc.scale_)
            else:
                store[f"{case}::scaler::{part}::mean"] = np.atleast_1d(sc.mean_)
                store[f"{case}::scaler::{part}::scale"] = np.atleast_1d(sc.scale_)


def main():
    store, cases = {}, []
    rng = np.random.default_rng(11)
    pair_bps = ["B_Nose", "B_Center", "B_Tail_base", "B_Left_ear", "W_Nose", "W_Center", "W_Tail_base", "W_Right_ear"]
    single_bps = ["Nose", "Left_ear", "Right_ear", "Center", "Tail_base", "Tail_tip"]
    data = {}
    for tag, bps, aids, n_ang, lens in [("pair", pair_bps, ["B", "W"], 2, (60, 45, 80)), ("single", single_bps, [""], 0, (50, 70))]:
        cols = G.labels(bps, n_ang)
        tabs = {f"vid{v}": G.damage(rng, G.synth_table(rng, n, cols, bps, 1.0 + 0.5 * v), cols) for v, n in enumerate(lens)}
        if tag == "pair":
            tabs["vid1"][:, 5] = np.nan            # a column that is all-NaN in one video
            j = cols.index(("B_Center", "B_Left_ear"))
            tabs["vid2"][:, j] = 7.25              # a column that is constant in one video (range 0 -> divisor 1)
        data[tag] = (cols, aids, tabs)
    # the filter case: 2 angle columns (uniform in [0, pi]: variance 0.8) and one near-constant distance column fall below
    # the threshold in EVERY video; everything else stays (checked below)
    cols = G.labels(pair_bps, 2)
    ftabs = {}
    for v, n in enumerate((70, 55, 90)):
        t = G.damage(rng, G.synth_table(rng, n, cols, pair_bps, 1.0 + 0.4 * v), cols)
        t[:, cols.index(("B_Nose", "B_Tail_base"))] = 30.0 + 0.1 * rng.standard_normal(n)
        sp = [cols.index(bp) for bp in pair_bps]
        t[:, sp] *= 3.0                            # speeds comfortably above the threshold
        ftabs[f"vid{v}"] = t
    data["filt"] = (cols, ["B", "W"], ftabs)
    # ragged: one more distance column and one coordinate fall below the threshold in ONE video only (groupwise sections
    # pool whatever each video kept; the dropped columns come back as zeros in that video)
    rtabs = {k: t.copy() for k, t in ftabs.items()}
    rtabs["vid1"][:, cols.index(("W_Nose", "W_Center"))] = 12.0 + 0.05 * rng.standard_normal(len(rtabs["vid1"]))
    rtabs["vid2"][:, cols.index(("B_Left_ear", "y"))] = -3.0 + 0.2 * rng.standard_normal(len(rtabs["vid2"]))
    data["ragged"] = (cols, ["B", "W"], rtabs)
    for tag, (cols, aids, tabs) in data.items():
        store[f"{tag}::columns"] = np.array(json.dumps([list(c) if isinstance(c, tuple) else c for c in cols]))
        store[f"{tag}::animal_ids"] = np.array(json.dumps(aids))
        for k, t in tabs.items():
            store[f"{tag}::raw::{k}"] = t
    full = 227272
    grid = [("pair", "mm_gw", "minmax", False, dict(dist="groupwise", speed="groupwise", coord="groupwise", log=True, samples_max=full, clip=10)),
            ("pair", "mm_pc", "minmax", False, dict(dist="per_column", speed="per_column", coord="per_column", log=True, samples_max=full, clip=10)),
            ("pair", "mm_mixed", "minmax", False, dict(dist="groupwise", speed=None, coord="per_column", log=False, samples_max=25, clip=0.9)),
            ("single", "mm_sub", "minmax", False, dict(dist="per_column", speed="groupwise", coord="groupwise", log=True, samples_max=30, clip=10)),
            ("pair", "rb_gw", "robust", False, dict(dist="groupwise", speed="groupwise", coord="groupwise", log=True, samples_max=full, clip=10)),
            ("pair", "rb_pc", "robust", False, dict(dist="per_column", speed="per_column", coord="per_column", log=True, samples_max=35, clip=10)),
            ("single", "rb_mixed", "robust", False, dict(dist="per_column", speed="groupwise", coord=None, log=False, samples_max=full, clip=10)),
            ("filt", "rb_filter", "robust", 1.2, dict(dist="groupwise", speed="per_column", coord="groupwise", log=True, samples_max=45, clip=10)),
            ("filt", "std_filter", "standard", 1.2, dict(dist="groupwise", speed="groupwise", coord="groupwise", log=True, samples_max=full, clip=10)),
            ("filt", "mm_filter", "minmax", 1.2, dict(dist="per_column", speed="per_column", coord="per_column", log=True, samples_max=40, clip=10)),
            ("ragged", "std_ragged", "standard", 1.2, dict(dist="groupwise", speed="groupwise", coord="groupwise", log=True, samples_max=50, clip=10))]
    first_mm = None
    for tag, name, scale, flt, kw in grid:
        cols, aids, tabs = data[tag]
        out, gs = run_reference(tabs, cols, aids, kw, scale, filter_low_variance=flt)
        case = f"{tag}::{name}"
        for k, df in out.items():
            assert [tuple(c) if isinstance(c, tuple) else c for c in df.columns] == cols
            store[f"{case}::out::{k}"] = df.to_numpy(float)
        dropped = {}
        if flt:   # what the filter removed per video (pass 2's view: the angles are set aside first)
            index = pd.Index(cols, tupleize_cols=False)
            for k, t in tabs.items():
                df = pd.DataFrame(t.copy(), columns=index)
                df = df.drop(columns=U.infer_column_types(df)["angles"])
                left = set(U._pp_filter_low_variance(df, flt).columns)
                dropped[k] = [list(c) if isinstance(c, tuple) else c for c in df.columns if c not in left]
            print(case, "dropped", dropped)
        cases.append(dict(case=case, data=tag, scale=scale, filter=flt, dropped=dropped, **kw))
        store_scaler(store, case, gs, scale)
        if first_mm is None and scale == "minmax":
            first_mm = (kw, gs)
    # the fitted MinMaxScalers re-applied to other videos (pretrained_scaler path)
    cols, aids, _ = data["pair"]
    kw, gs = first_mm
    other = {k: G.damage(rng, G.synth_table(rng, 40 + 9 * i, cols, pair_bps, 0.8 + i), cols) for i, k in enumerate(["new0", "new1"])}
    out2, _ = run_reference(other, cols, aids, kw, "minmax", pretrained=gs)
    for k, t in other.items():
        store[f"pair::pre::raw::{k}"] = t
    for k, df in out2.items():
        store[f"pair::pre::out::{k}"] = df.to_numpy(float)
    # scale_table on its own with the MinMaxScaler (per-video statistics only)
    df = pd.DataFrame(data["pair"][2]["vid0"].copy(), columns=pd.Index(cols, tupleize_cols=False))
    for nm, kw2 in [("mm_pc", dict()), ("mm_gw", dict(dist_standardize="groupwise", speed_standardize="groupwise",
                                                      coord_standardize="groupwise"))]:
        store[f"scale_table::{nm}"] = U.scale_table(df, scale="minmax", animal_ids=aids, **kw2).to_numpy(float)
    store["cases"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(HERE, "preprocess_r03.npz"), **store)
    print("preprocess_r03.npz", os.path.getsize(os.path.join(HERE, "preprocess_r03.npz")), "bytes,", len(cases), "cases")


if __name__ == "__main__":
    main()
