"""Schedules and small helpers; same names and meaning as the reference's utils/functional.py."""
import math

import numpy as np


def fold(f, l, a):
    """utils/functional.py:1-2."""
    for x in l:
        a = f(a, x)
    return a


def parameters_allocation_check(module):
    """utils/functional.py:13-15 — all parameters on the GPU, or none. (The reference folds tensors with `and`,
    which no longer evaluates under torch >= 0.4; the intent is kept.)"""
    flags = [p.is_cuda for p in module.parameters()]
    return all(flags) or not any(flags)


def kld_coef(i, extended=False, py2_floor=True):
    """KLD annealing weight, utils/functional.py:28-33.

    The reference is Python 2, where `(i - 3500)/1000` with an int step is FLOOR division (a staircase).
    `py2_floor=True` (default) reproduces the reference as it actually ran; `py2_floor=False` evaluates the same
    source line with Python-3 true division. Float steps always use true division (as in Python 2)."""
    floor = py2_floor and isinstance(i, (int, np.integer))
    if not extended:
        x = (i - 3500) // 1000 if floor else (i - 3500) / 1000
    else:
        x = (i - 50000) // 15000 if floor else (i - 50000) / 15000
    return (math.tanh(x) + 1) / 2


def temp_coef(i):
    """Softmax temperature, utils/functional.py:35-42: max(exp(-2e-5 * i), 0.1)."""
    initial_temp = 1.0
    min_temp = 0.1
    ANNEAL_RATE = 0.00002
    return float(np.maximum(initial_temp * np.exp(-ANNEAL_RATE * i), min_temp))
