"""Attribute-style config for the hot path (the reference uses EasyDict over a JSON file,
utils/config.py:50-66; only the keys the encode/decode path reads are checked here)."""
from __future__ import annotations

import json


class Config(dict):
    """dict with attribute access (what the reference's EasyDict provides for this path)."""
    __getattr__ = dict.__getitem__

    def __setattr__(self, k, v):
        self[k] = v


# configs/llicti_A.json:13-32 -- the one combination the released model exercises
CONFIG_A = {
    "ycocg": True, "clrchs": 3, "clr_joint_mode": 2, "clrjnt0seqmd": False, "mwsa_joint": False,
    "chs": [88, 1, 1, 1, 1], "conv_layers": 3, "combine_layers1toL": False,
    "Evens": [4, 4, 4, 4, 4], "Odds": [3, 3, 3, 3, 3], "dwtlevels": [0, 1, 2, 3, 4],
    "useprevlevNN": [False, True, True, True, True], "wtr_type": "lazydwt", "net_type": "regular",
    "lif_prec_bits": 8, "ent_mdl_num": 4, "activfun": "ReLU", "subtract_mean": False,
    "distribution": "normal", "num_mixtures": 5,
}
_CHECKED = ["ycocg", "clrchs", "clr_joint_mode", "mwsa_joint", "conv_layers", "combine_layers1toL", "Evens", "Odds",
            "dwtlevels", "useprevlevNN", "lif_prec_bits", "ent_mdl_num", "activfun", "subtract_mean", "distribution",
            "num_mixtures"]


def default_config(**over):
    c = Config(CONFIG_A)
    c.update({"agent": "LLICTIAgent", "mode": "eval_model", "cuda": True, "gpu_device": 0, "seed": 1337,
              "exp_name": "llicti_amd", "test_data": "synthetic:64x64x2"})
    c.update(over)
    return c


def load_json(path):
    with open(path) as f:
        return Config(json.load(f))


def check_supported(config):
    """The HIP kernels implement exactly the released configuration (config A).  Anything else the
    reference's constructor would accept is rejected loudly instead of being silently mis-coded."""
    for k in _CHECKED:
        have = config[k] if k in config else None
        if have != CONFIG_A[k]:
            raise NotImplementedError(f"config.{k}={have!r}: the MI355X hot path implements configs/llicti_A.json "
                                      f"only ({k}={CONFIG_A[k]!r})")
    if int(config["chs"][0]) != 88:
        raise NotImplementedError("config.chs[0] must be 88")
