"""Helpers shared by the parity tests: load a golden fixture (tests/golden/*.npz, produced by
oracle/make_golden.py from the imported reference classes) and rebuild its inputs."""
import ast
import os

import numpy as np
import torch

from i2v_amd import graphs, weights

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    fx = {k: z[k] for k in z.files}
    for k in ("steps", "b", "f", "hw", "clip_seed", "wseed"):
        if k in fx:
            fx[k] = int(fx[k])
    if "lr" in fx:
        fx["lr"] = float(fx["lr"])
    if "depth" in fx:
        fx["depth"] = ast.literal_eval(str(fx["depth"]))
        fx["kw"] = ast.literal_eval(str(fx["kw"]))
        fx["models"] = [str(m) for m in fx["models"]]
        fx["kind"], fx["prec"] = str(fx["kind"]), str(fx["prec"])
    return fx


def videos_of(fx, dtype=torch.float32):
    u8 = torch.from_numpy(fx["clip_u8"])
    mean = torch.tensor(MEAN, dtype=dtype).view(1, 3, 1, 1, 1)
    std = torch.tensor(STD, dtype=dtype).view(1, 3, 1, 1, 1)
    return (u8.to(dtype) / 255 - mean) / std


def hook_lists(fx):
    """[(graph, state_dict, [hook tensor ids])] per model, in the reference's hook order."""
    out = []
    for m in fx["models"]:
        g = graphs.build_tiny(m, (fx["hw"], fx["hw"]))
        sd = weights.synthetic_state_dict(g, fx["wseed"])
        d = fx["depth"][m] if isinstance(fx["depth"], dict) else fx["depth"]
        ds = d if isinstance(d, list) else [d]
        out.append((g, sd, [g.hooks[k] for k in ds]))
    return out
