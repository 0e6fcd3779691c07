"""Helpers the example scripts import from inferix.core.utils (`set_random_seed` :17-28, `divide`)."""
import random

import numpy as np
import torch


def set_random_seed(seed):
    """Seed python, numpy and torch (host + every visible GPU); returns the seed."""
    if seed is None:
        raise AssertionError("Please provide a seed in config.json")
    for seeder in (random.seed, np.random.seed, torch.manual_seed, torch.cuda.manual_seed_all):
        seeder(seed)
    return seed


def divide(numerator, denominator):
    q, r = divmod(numerator, denominator)
    assert r == 0, f"{numerator} is not divisible by {denominator}"
    return q
