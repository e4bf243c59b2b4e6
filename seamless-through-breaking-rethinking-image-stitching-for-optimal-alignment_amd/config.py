"""Config plugin surface of the stitching path (reference: out.py:15-54).

``--inf_cfg NAME`` loads ``inf_configs/NAME.py`` which must export ``get_infernce_config()`` and
``get_tps_pipline_config(cfg)`` (spelling as in out.py:45-46); the model hyper-parameters come from
``configs/<model_config_name>.py:config_dict``.  ``CfgNode`` is a minimal attribute dict with the
yacs behaviours the path relies on (attribute access, ``hasattr`` gating, ``merge_from_other_cfg``)."""
from __future__ import annotations

import importlib


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def merge_from_other_cfg(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge_from_other_cfg(v)
            else:
                self[k] = v
        return self

    def clone(self):
        return CfgNode(self)


def load_model_config(name="last_config"):
    """configs/<name>.py:config_dict -> CfgNode (out.py:32)."""
    return CfgNode(importlib.import_module(f"configs.{name}").config_dict)


def load_inference_config(inf_cfg, model_config_name="last_config"):
    """The merge of out.py:43-54: model dict + inference overlay; returns (cfg, tps_pipeline_cfg)."""
    plug = importlib.import_module(f"inf_configs.{inf_cfg}")
    for fn in ("get_infernce_config", "get_tps_pipline_config"):
        if not hasattr(plug, fn):
            raise AttributeError(f"inf_configs/{inf_cfg}.py must export {fn}()")
    cfg = load_model_config(model_config_name)
    cfg.merge_from_other_cfg(CfgNode(plug.get_infernce_config()))
    return cfg, CfgNode(plug.get_tps_pipline_config(cfg))
