"""The handful of helpers of the reference's utils/common.py + utils/logger.py that the piecewise-ICP entry needs:
load_yaml (utils/common.py:20-39), dir_exist (:13-17), access_device (:97-99), setup_seed (:124-131), get_logger
(utils/logger.py:27-51, plain `logging` instead of coloredlogs) and an attribute dict in place of easydict."""
import logging
import os
import os.path as osp
import random

import numpy as np
import yaml


class AttrDict(dict):
    """Minimal easydict.EasyDict: nested dicts become attribute-accessible."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def dir_exist(path, sub_folders=None):
    os.makedirs(path, exist_ok=True)
    for sub in sub_folders or []:
        os.makedirs(osp.join(path, sub), exist_ok=True)


def load_yaml(path, keep_sub_directory=False):
    """keep_sub_directory=False merges the sections into one flat dict (utils/common.py:31-39)."""
    with open(path, 'r') as f:
        cfg = yaml.safe_load(f)
    if keep_sub_directory:
        return cfg
    flat = dict()
    for _, value in cfg.items():
        if value:
            flat.update(value)
    return flat


def access_device():
    """The reference falls back to the CPU (utils/common.py:97-99); this engine has no CPU path."""
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("fusion4landslide_amd needs an AMD GPU")
    return torch.device("cuda:0")


def setup_seed(seed=0):
    import torch
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def get_logger(log_file=None, name="fusion4landslide_amd"):
    logger = logging.getLogger(name)
    logger.setLevel(logging.INFO)
    if not logger.handlers:
        fmt = logging.Formatter('[%(asctime)s] [%(levelname)s] %(message)s')
        h = logging.StreamHandler()
        h.setFormatter(fmt)
        logger.addHandler(h)
        if log_file:
            fh = logging.FileHandler(log_file)
            fh.setFormatter(fmt)
            logger.addHandler(fh)
    return logger
