"""Host logic of tools_parity_distribution.py (no GPU): which utterances it samples and how it summarises."""
import numpy as np

import tools_parity_distribution as tpd


def test_sample_covers_all_shards_and_keeps_389():
    for per_shard in (9, 16):
        ids = tpd.pick_ids(per_shard)
        assert ids == sorted(set(ids)) and 389 in ids and len(ids) >= 8 * per_shard
        assert min(ids) >= 0 and max(ids) < 1024
        per = np.bincount(np.array(ids) // 128, minlength=8)
        assert per.min() >= per_shard                     # every shard of the 8-GPU split is sampled


def test_summary_counts_and_ratios():
    rows = []
    for i, (e32, e) in enumerate([(4e-5, 2e-5), (1e-4, 2e-4), (2e-3, 5e-4), (3e-5, 1.5e-3)]):
        rows.append({"id": i, "peak": 1.0, "f32": {"err": e32, "at": 0},
                     "gpu_pin27": {"err": e, "at": 0, "p999": e / 2, "sc_delta": 1e-8 * (i + 1)}})
    s = tpd.summarise(rows, ["gpu_pin27"])
    assert s["n"] == 4 and s["numpy_f32"]["over_tol"] == 1 and s["numpy_f32"]["worst"] == 2e-3
    g = s["gpu_pin27"]
    assert g["over_tol"] == 1 and g["over_tol_ids"] == [3] and g["worst_id"] == 3 and g["worst"] == 1.5e-3
    ratios = np.array([0.5, 2.0, 0.25, 50.0])
    assert abs(g["ratio_to_numpy_f32"]["geomean"] - float(np.exp(np.mean(np.log(ratios))))) < 1e-12
    assert g["ratio_to_numpy_f32"]["max"] == 50.0 and g["worst_sc_delta"] == 4e-8
