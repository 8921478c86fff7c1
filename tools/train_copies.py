"""Which Python lines issue the copy-like torch ops of one training step (a TorchDispatchMode that records the calling frame).
python tools/train_copies.py"""
import os
import sys
import traceback
from collections import Counter

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--mode", "train", "--image-size", "384", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
import bench  # noqa: E402

CNT = Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("copy", "clone", "cat", "fill", "zero", "index_put", "contiguous", "_to_copy")):
            fr = [f for f in traceback.extract_stack(limit=14) if ROOT in f.filename and "train_copies" not in f.filename]
            where = f"{os.path.relpath(fr[-1].filename, ROOT)}:{fr[-1].lineno}" if fr else "?"
            CNT[(name, where)] += 1
        return func(*args, **(kwargs or {}))


_orig = bench.train_mode


def wrapped(args, m2, m1, dev, dt):
    with Spy():
        out = _orig(args, m2, m1, dev, dt)
    print("ops over 3 steps (1 warm-up, 1 timed, 1 with timed legs):")
    for (name, where), c in CNT.most_common(50):
        print(f"{c:6d}  {name:28s} {where}")
    return out


bench.train_mode = wrapped
bench.main()
