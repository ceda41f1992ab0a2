"""TEST INFRASTRUCTURE (CPU): where does the 16-bit error of the heat / depth maps come from?

For every rounding point of the HIP network (tests/precision/emulate.py: 137 stored tensors, 139 folded weight sets) run
the network in fp32 with ONLY that point rounded to the 16-bit type and record the error of the deployed outputs against
the all-fp32 run.  Rounding errors at different points are independent to first order, so the squared contributions add
up to the all-16-bit error (the script checks that: "sum of parts" vs "all rounded").  Output: a JSON table
(tests/golden/precision_attribution.json) sorted by contribution, and the grouped shares DESIGN.md quotes.

    python tests/precision/attribute.py [--dtype f16|bf16] [--frames 2] [--out tests/golden/precision_attribution.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import emulate                                           # noqa: E402
from object_keypoints_amd import synth                   # noqa: E402


def group_of(kind, name):
    if name == "frames" or name.startswith("backbone.pre."):
        return "pre (stem + two residuals)"
    if name.startswith("backbone.hgs."):
        i = name.split(".")[2]
        if name.endswith(".up2") or name.count(".") == 2 or (kind == "act" and name.split(".")[-1].startswith("low2") and not name.split(".")[-1][-1].isdigit()):
            return f"hourglass {i}: transposed convolutions / merges"
        return f"hourglass {i}: fire modules"
    if name.startswith("backbone.cnvs."):
        return "cnvs (3x3 after each hourglass)"
    if name.startswith("backbone.inters") or name.startswith("backbone.cnvs_") or name.startswith("backbone.merge"):
        return "inter-stack merge + residual"
    return "heads"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden", "precision_attribution.json"))
    a = ap.parse_args()
    dt = {"f16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
    net, emu = emulate.build(3, 0)
    x = torch.from_numpy(synth.frames(a.frames, seed=1, start=0))
    ref = emu.forward(x, emulate.Policy(None))
    probe = emulate.Policy(dt)
    full = emu.forward(x, probe)

    def err(out):
        return {n: (float((o - r).abs().max()), float(((o - r).double() ** 2).mean())) for n, o, r in zip(("heat", "depth", "centers"), out, ref)}

    e_full = err(full)
    rows = []
    t0 = time.time()
    points = [("act", n) for n in probe.seen_acts] + [("weight", n) for n in probe.seen_weights]
    for i, (kind, name) in enumerate(points):
        p = emulate.Policy(None, acts={name: dt} if kind == "act" else None, weights={name: dt} if kind == "weight" else None)
        e = err(emu.forward(x, p))
        rows.append({"kind": kind, "name": name, "group": group_of(kind, name),
                     "heat_max": e["heat"][0], "heat_ms": e["heat"][1], "depth_max": e["depth"][0], "depth_ms": e["depth"][1]})
        if i % 25 == 0:
            print(f"{i}/{len(points)} {time.time() - t0:.0f}s", flush=True)
    tot_h = sum(r["heat_ms"] for r in rows)
    tot_d = sum(r["depth_ms"] for r in rows)
    for r in rows:
        r["heat_share"] = r["heat_ms"] / tot_h
        r["depth_share"] = r["depth_ms"] / tot_d
    rows.sort(key=lambda r: -r["heat_ms"])
    groups = {}
    for r in rows:
        g = groups.setdefault((r["group"], r["kind"]), {"heat_share": 0.0, "depth_share": 0.0, "points": 0})
        g["heat_share"] += r["heat_share"]
        g["depth_share"] += r["depth_share"]
        g["points"] += 1
    out = {"dtype": a.dtype, "frames": a.frames,
           "all_rounded": {k: {"max": v[0], "rms": v[1] ** 0.5} for k, v in e_full.items()},
           "sum_of_parts_rms": {"heat": tot_h ** 0.5, "depth": tot_d ** 0.5},
           "groups": [{"group": k[0], "kind": k[1], **v} for k, v in sorted(groups.items(), key=lambda kv: -kv[1]["heat_share"])],
           "points": rows}
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["all_rounded"]), json.dumps(out["sum_of_parts_rms"]))
    for g in out["groups"]:
        print(f"{g['heat_share']:.3f} {g['depth_share']:.3f} {g['points']:3d} {g['kind']:6s} {g['group']}")
    print("top 25 points (heat share, depth share):")
    for r in rows[:25]:
        print(f"{r['heat_share']:.3f} {r['depth_share']:.3f} {r['kind']:6s} {r['name']}")


if __name__ == "__main__":
    main()
