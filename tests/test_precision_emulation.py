"""CPU: the rounding-point model of the 16-bit configurations (tests/precision/emulate.py) against the reference's golden outputs.
It pins the ABSOLUTE bounds of tests/precision/bounds.py from a computation that is independent of the device kernels: the same
numbers are then asserted on the GPU (tests/test_gpu_net.py, tests/test_gpu_f32x3.py)."""
import os
import sys

import numpy as np
import pytest
import torch

import cases
import golden_util as gu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "precision"))
import emulate          # noqa: E402
from bounds import BOUNDS   # noqa: E402


@pytest.fixture(scope="module")
def emulated():
    from object_keypoints_amd import synth
    out = {}
    for name, case in cases.NET_CASES.items():
        _, emu = emulate.build(case["heatmaps_out"], case["weight_seed"])
        x = torch.from_numpy(synth.frames(1, seed=case["frame_seed"], start=case["frame_index"]))
        g = gu.golden_net(name)
        runs = {"f32": emu.forward(x, emulate.Policy(None)), "f16": emu.forward(x, emulate.Policy(torch.float16)),
                "bf16": emu.forward(x, emulate.Policy(torch.bfloat16)), "f32x3": emu.forward(x, emulate.Policy(None, x3=True))}
        out[name] = {tag: {k: np.abs(t.numpy().astype(np.float64) - g[k]) for k, t in zip(("heat", "depth", "centers"), r)} for tag, r in runs.items()}
    return out


@pytest.mark.parametrize("name", sorted(cases.NET_CASES))
def test_the_emulator_without_rounding_is_the_reference(emulated, name):
    e = emulated[name]["f32"]
    assert e["heat"].max() <= 1e-5 and e["depth"].max() <= 5e-5 and e["centers"].max() <= 5e-5


@pytest.mark.parametrize("name", sorted(cases.NET_CASES))
@pytest.mark.parametrize("tag", ["f16", "bf16"])
def test_16bit_rounding_model_sits_inside_the_absolute_bounds(emulated, name, tag):
    e, b = emulated[name][tag], BOUNDS[tag]
    for key in ("heat", "depth", "centers"):
        assert e[key].max() <= 0.75 * b[key + "_max"], (key, e[key].max())          # the bounds are ~2x the model
        assert e[key].mean() <= 0.75 * b[key + "_mean"], (key, e[key].mean())
        assert e[key].max() >= 0.2 * b[key + "_max"]                                  # ... and not vacuous


@pytest.mark.parametrize("name", sorted(cases.NET_CASES))
def test_bf16_rounds_eight_times_coarser_than_fp16_and_neither_meets_the_fp32_bar(emulated, name):
    f16, bf16 = emulated[name]["f16"]["heat"], emulated[name]["bf16"]["heat"]
    assert 5.0 <= bf16.mean() / f16.mean() <= 11.0            # 2^3 = three mantissa bits
    assert f16.max() > BOUNDS["f32"]["heat_max"]              # 16-bit storage misses the 1e-3 heat bar on the maximum ...
    assert f16.mean() < BOUNDS["f32"]["heat_max"]             # ... though not on average


@pytest.mark.parametrize("name", sorted(cases.NET_CASES))
def test_three_term_split_is_fp32_grade(emulated, name):
    """x*w = x_hi*w_hi + x_lo*w_hi + x_hi*w_lo on fp16 halves (the f32x3 configuration): three orders of magnitude inside the bar."""
    e = emulated[name]["f32x3"]
    assert e["heat"].max() <= 2e-5 and e["depth"].max() <= 1e-4 and e["centers"].max() <= 1e-4


def test_attribution_table_says_no_layer_subset_reaches_the_bar():
    """tests/golden/precision_attribution.json (tests/precision/attribute.py): the fp16 error variance is spread over all rounding
    points; the squared contributions add up to the all-fp16 error, so the table can be used to price mixed-precision choices."""
    import json
    with open(os.path.join(gu.GOLDEN, "precision_attribution.json")) as f:
        t = json.load(f)
    assert len(t["points"]) == 276
    assert max(p["heat_share"] for p in t["points"]) <= 0.05
    assert abs(t["sum_of_parts_rms"]["heat"] / t["all_rounded"]["heat"]["rms"] - 1.0) <= 0.1      # independence: parts add in quadrature
    # to bring the maximum from 1.8e-3 under 1e-3 with a 2x margin the variance has to drop 12x: that takes the 63 largest of the
    # 276 points - every tensor and every weight set of the trunk (all of the network's large convolutions) and most of both
    # hourglasses in more than 16 bits, i.e. the whole MFMA budget again
    top = sorted((p["heat_share"] for p in t["points"]), reverse=True)
    assert sum(top[:40]) < 1.0 - 1.0 / 12.0 < sum(top[:80])
    big = [p for p in t["points"] if p["heat_share"] >= top[62]]
    trunk = [p for p in big if not p["name"].startswith("backbone.hgs.") or p["name"].count(".") <= 3]
    assert len(trunk) >= 25


def test_plan_tool_selects_the_shipped_mixed_plan_for_the_test_network():
    """tests/precision/plan.py prices every float32mix plan on given weights with the rounding-point model and names the fastest one
    inside a heat-error bound: for the synthetic network that is the plan object_keypoints_amd ships as default (fp16 hourglass levels
    n <= 2, single-term residual branches); one more fp16 level is predicted - and measured - at ~9e-4."""
    import plan
    from object_keypoints_amd import ops, synth
    _, emu = emulate.build(3, 0)
    x = torch.from_numpy(synth.frames(2, seed=1))
    rows, best = plan.choose(emu, x, bound=5.5e-4)
    assert best is not None and (best["fp16_levels"], best["branch_single_term"]) == (ops.MIX_FP16_LEVELS, ops.MIX_BRANCH_SINGLE) == (2, True)
    by = {(r["fp16_levels"], r["branch_single_term"]): r for r in rows}
    assert by[(3, True)]["heat_max"] > 7e-4 and by[(4, True)]["heat_max"] > 1e-3          # where the margin goes
    assert by[(2, False)]["heat_max"] < by[(2, True)]["heat_max"] < 5.5e-4


def test_float32mix_on_weight_families_it_was_not_derived_on():
    """tests/precision/families.py -> tests/golden/precision_families.json: the shipped float32mix plan priced with the rounding-point
    model on five weight families.  What the table establishes, asserted here (and one family recomputed, so the file cannot go stale):
      * on the family the plan was derived on (BatchNorm gains of 0.3 closing every branch; two seeds, and with logits twice as large)
        it stays inside the 1e-3 heat bar, worst 8.1e-4: the bound for this family is 9e-4 (tests/precision/bounds.py);
      * on families whose branches enter the stream with gain 1 (torch's default BatchNorm initialisation, or the synthetic gamma range
        without the 0.3, running statistics calibrated so that activations stay O(10)) it MISSES the bar by 3-3.6x: the single-term
        fp16 branches are only cheap where a small BatchNorm gain attenuates them.  float32x3 holds everywhere (<= 1e-5).
    Hence load_keypoint_net(..., compute_dtype="float32mix", audit_frames=...) - an on-device audit with a fall-back to float32x3
    (tests/test_gpu_f32x3.py::test_audit_frames_falls_back_to_float32x3_on_unit_gain_weights)."""
    import json
    import families
    with open(os.path.join(gu.GOLDEN, "precision_families.json")) as f:
        table = json.load(f)["families"]
    assert set(table) == set(families.FAMILIES)
    for name in ("derived-on", "other-seed", "head-gain-x2"):
        r = table[name]
        assert r["float32mix"]["heat_max"] <= BOUNDS["f32mix"]["heat_max_any_frame"] and r["float32mix"]["meets_heat_1e-3"], name
        assert r["float16"]["heat_max"] > 1e-3
    for name in ("torch-default", "branch-gain-1"):
        r = table[name]
        assert 2e-3 <= r["float32mix"]["heat_max"] <= 6e-3 and not r["float32mix"]["meets_heat_1e-3"], name
        assert r["activation_abs_max"] < 100.0                       # a sane network: nothing near the fp16 range limit
    for name, r in table.items():
        assert r["float32x3"]["heat_max"] <= 2e-5 and r["float32x3"]["finite"] and r["float32mix"]["finite"], name
    again = families.price("torch-default", n_frames=table["torch-default"]["frames"])
    assert abs(again["float32mix"]["heat_max"] / table["torch-default"]["float32mix"]["heat_max"] - 1.0) <= 0.02
