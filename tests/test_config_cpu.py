"""Typed-JSON optimizer config loader (format of the reference's config/optim_config, util/config.py:4-43)."""
import json

from vlsfr_amd.main import OPTIM_CONFIG, load_config


def test_load_config_reference_format(tmp_path):
    raw = {"scheduler": ["str", "multistep"], "epochs": ["int", "1"], "warmup": ["int", "0"], "patience": ["int", "4"],
           "milestones": ["int", [8, 14, 17]], "gammas": ["float", [0.1, 0.1, 0.1]], "LR_min": ["float", "0.00001"],
           "optim": ["str", "SGD"], "LR": ["float", "0.1"], "decay": ["float", "0.0001"], "momentum": ["float", "0.9"],
           "nesterov": ["bool", "1"]}
    p = tmp_path / "optim_config"
    p.write_text(json.dumps(raw))
    assert load_config(str(p)) == OPTIM_CONFIG
