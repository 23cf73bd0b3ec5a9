"""Helpers shared by the tests: load golden cases, build oracle objects."""
import os

import numpy as np
import torch

import weightgen
from oracle import endosurf_oracle as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["init_deform", "trained_deform", "trained_nodeform"]
RENDER_CFG = dict(net_chunk=80000, anneal_end=50000, n_samples=32, n_importance=32, important_begin_iter=0,
                  up_sample_steps=4, perturb=True)


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return {k: z[k] for k in z.files}


def oracle_for(case, dtype=torch.float32, requires_grad=False):
    seed = int(case["meta/seed"]); mode = str(case["meta/mode"]); use_deform = bool(case["meta/use_deform"])
    state = weightgen.make_state(seed, mode, use_deform)
    params = {k: torch.tensor(v, dtype=dtype, requires_grad=requires_grad) for k, v in state.items()}
    net = O.OracleNet(params, use_deform)
    return O.OracleRenderer(net, RENDER_CFG), params


def T(a, dtype=torch.float32):
    return torch.tensor(a, dtype=dtype)
