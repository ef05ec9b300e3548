#!/usr/bin/env python
"""Entry point: ``python train.py --config configs/<file>.yml [--set key=value ...]``.

Same command line as the reference (``train.py:5-13``) plus ``--set`` overrides.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from d4pg_pytorch_amd.config import read_config          # noqa: E402
from d4pg_pytorch_amd.engine import load_engine          # noqa: E402

parser = argparse.ArgumentParser(description="Run training")
parser.add_argument("--config", type=str, help="Path to the config file.")
parser.add_argument("--set", nargs="*", default=[], metavar="KEY=VALUE",
                    help="override config entries, e.g. --set device=cpu num_steps_train=1000")
parser.add_argument("--check", action="store_true",
                    help="race detection: validate the sequence numbers of the transition rings (check_rings)")

if __name__ == "__main__":
    args = vars(parser.parse_args())
    if not args["config"]:
        parser.error("--config is required")
    config = read_config(args["config"], overrides=args["set"] + (["check_rings=true"] if args["check"] else []))
    engine = load_engine(config)
    exp_dir = engine.train()
    print(f"results in {exp_dir}")
