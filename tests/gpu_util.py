"""Shared helpers for the GPU tests: build an endosurf_amd renderer carrying a golden case's weights."""
import torch
import yaml

import weightgen
from oracle_util import RENDER_CFG

NET_CFG = yaml.safe_load("""
bound: 1.0
use_deform: True
deform_network:
  enc_pos_cfg: {enc_type: frequency, input_dim: 3, multires: 6}
  enc_time_cfg: {enc_type: frequency, input_dim: 1, multires: 6}
  n_layers: 9
  hidden_dim: 256
  skips: [4]
  out_dim: 3
sdf_network:
  enc_pos_cfg: {enc_type: frequency, input_dim: 3, multires: 6}
  n_layers: 9
  hidden_dim: 256
  skips: [4]
  out_dim: 257
  geometric_init: True
  geometric_init_bias: 0.8
color_network:
  enc_pos_cfg: {enc_type: frequency, input_dim: 3, multires: 10}
  enc_dir_cfg: {enc_type: frequency, input_dim: 3, multires: 4}
  n_layers: 9
  hidden_dim: 256
  skips: [4]
  feat_dim: 256
  out_dim: 3
deviation_network: {init_val: 0.3}
""")


def net_cfg(use_deform=True):
    import copy
    c = copy.deepcopy(NET_CFG)
    c["use_deform"] = use_deform
    return c


def state_to_ckpt(state, use_deform):
    nets = ["sdf_network", "color_network", "deviation_network"] + (["deform_network"] if use_deform else [])
    return {net: {k[len(net) + 1:]: torch.tensor(v) for k, v in state.items() if k.startswith(net + ".")} for net in nets}


def renderer_for(seed, mode, use_deform, render_cfg=None):
    from endosurf_amd import EndoSurfRenderer
    r = EndoSurfRenderer(dict(render_cfg or RENDER_CFG), net_cfg(use_deform), device="cuda")
    r.load_checkpoint(state_to_ckpt(weightgen.make_state(seed, mode, use_deform), use_deform))
    return r


def renderer_for_case(c):
    return renderer_for(int(c["meta/seed"]), str(c["meta/mode"]), bool(c["meta/use_deform"]))
