"""Configuration surface of the projection path: the reference's YAML keys and CLI flag names.

Counterpart of /root/reference/utils/config.py:36-94 (YAML ``<cfg>.yml`` merged over ``default.yml``
of the same directory, or ``<dir>/cfg.yml``) and of the reconstruction flags of whitebox.py:358-395 /
blackbox.py:714-762: ``--rec_iters``, ``--rec_rr``, ``--rec_lr``, ``--batch_size``, ``--override``,
``--same_init``.  Plain dict + argparse instead of tf.app.flags.
"""
from __future__ import annotations

import argparse
import os
import re
from typing import Dict, Optional

import yaml

DEFAULTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfgs", "projection_defaults.yml")

REC_KEYS = ("REC_ITERS", "REC_LR", "REC_RR", "BATCH_SIZE", "LATENT_DIM", "NET_DIM", "USE_BN", "IMAGE_DIM")


def builtin_cfg(name: str) -> str:
    """Spec of a shipped config, ``<defaults file>#<dataset>``: mnist | fmnist (f-mnist) | celeba."""
    name = {"fmnist": "f-mnist"}.get(name.lower(), name.lower())
    return DEFAULTS + "#" + name


def _from_defaults(path: str, dataset: str) -> Dict:
    with open(path, "r") as f:
        d = yaml.safe_load(f)
    if dataset not in d["datasets"]:
        raise RuntimeError("[!] No built-in configuration for {}.".format(dataset))
    c, e = d["common"], d["datasets"][dataset]
    return {"BATCH_SIZE": c["batch_size"], "TEST_BATCH_SIZE": c["test_batch_size"], "USE_BN": c["use_bn"],
            "LATENT_DIM": c["latent_dim"], "NET_DIM": e.get("net_dim", c["net_dim"]), "NUM_GPUS": c["num_gpus"],
            "DATASET_NAME": dataset, "ARCH_TYPE": e["arch"], "IMAGE_DIM": list(e["image_dim"]),
            "REC_ITERS": e["steps"], "REC_LR": float(e["lr"]), "REC_RR": e["restarts"]}


def load_config(cfg_path: str) -> Dict:
    """A built-in spec (``builtin_cfg``), a reference-style flat ``<cfg>.yml`` (merged over ``default.yml`` of the same
    directory, utils/config.py:50-71) or a directory holding ``cfg.yml``."""
    if "#" in cfg_path and os.path.exists(cfg_path.split("#", 1)[0]):
        path, dataset = cfg_path.split("#", 1)
        cfg = _from_defaults(path, dataset)
        cfg["cfg_path"] = cfg_path
        return cfg
    if not os.path.exists(cfg_path):
        raise RuntimeError("[!] Configuration path {} does not exist.".format(cfg_path))
    if os.path.isdir(cfg_path):
        cfg_path = os.path.join(cfg_path, "cfg.yml")
        with open(cfg_path, "r") as f:
            cfg = yaml.safe_load(f) or {}
    else:
        with open(cfg_path, "r") as f:
            loaded = yaml.safe_load(f) or {}
        cfg = {}
        default = os.path.join(os.path.dirname(cfg_path), "default.yml")
        if os.path.exists(default):
            with open(default, "r") as f:
                cfg = yaml.safe_load(f) or {}
        cfg.update(loaded)
    cfg["cfg_path"] = cfg_path
    return cfg


def add_rec_flags(parser: argparse.ArgumentParser) -> argparse.ArgumentParser:
    parser.add_argument("--cfg", type=str, default=builtin_cfg("mnist"), help="config .yml or directory")
    parser.add_argument("--rec_iters", type=int, default=None, help="L: GD steps per restart (REC_ITERS)")
    parser.add_argument("--rec_rr", type=int, default=None, help="R: random restarts (REC_RR)")
    parser.add_argument("--rec_lr", type=float, default=None, help="GD learning rate (REC_LR)")
    parser.add_argument("--batch_size", type=int, default=None, help="images per reconstruct() call")
    parser.add_argument("--override", action="store_true",
                        help="let --rec_rr/--rec_lr/--rec_iters win over a --rec_path (whitebox.py:261-264)")
    parser.add_argument("--same_init", action="store_true",
                        help="same z0 for every batch, drawn with sigma=1 (whitebox.py:181-183)")
    parser.add_argument("--rec_path", type=str, default=None,
                        help="recs_rr{R}_lr{lr}_iters{L} directory name (gan.py:467-478)")
    return parser


_REC_PATH_RE = re.compile(r"recs_rr(.*)_lr(.*)_iters(.*)")


def resolve_rec_params(cfg: Dict, args: Optional[argparse.Namespace] = None) -> Dict:
    """Resolution order of whitebox.py:246-264 / blackbox.py:640-658: config, then the values parsed
    from ``rec_path``, then -- only with ``--override`` -- the command-line values."""
    out = {"rec_rr": int(cfg.get("REC_RR", 10)), "rec_lr": float(cfg.get("REC_LR", 10.0)),
           "rec_iters": int(cfg.get("REC_ITERS", 200)), "batch_size": int(cfg.get("BATCH_SIZE", 50))}
    if args is None:
        return out
    cli = {k: getattr(args, k, None) for k in ("rec_rr", "rec_lr", "rec_iters")}
    rec_path = getattr(args, "rec_path", None)
    if rec_path:
        m = _REC_PATH_RE.findall(rec_path)
        if m:
            rr, lr, it = m[0]
            out.update(rec_rr=int(rr), rec_lr=float(lr), rec_iters=int(re.match(r"\d+", it).group(0)))
        if getattr(args, "override", False):
            out.update({k: v for k, v in cli.items() if v is not None})
    else:
        out.update({k: v for k, v in cli.items() if v is not None})
    if getattr(args, "batch_size", None):
        out["batch_size"] = int(args.batch_size)
    return out


def rec_dir_name(rec_rr: int, rec_lr: float, rec_iters: int) -> str:
    """Directory name the reference encodes the hyper-parameters in (gan.py:467-478)."""
    return "recs_rr{:d}_lr{:.5f}_iters{:d}".format(rec_rr, rec_lr, rec_iters)
