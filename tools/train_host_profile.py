"""Host-side cost of one stage-II training step: cProfile over `bench.py --mode train`'s step (python tools/train_host_profile.py).
The step is launch-bound from the host (legs_host_enqueue_ms ~ legs_ms): this lists where the Python time goes."""
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--mode", "train", "--image-size", "384", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"]
import bench  # noqa: E402

prof = cProfile.Profile()
_orig = bench.train_mode


def wrapped(args, m2, m1, dev, dt):
    import torch
    # warm everything once outside the profile, then profile a second call's timed loop
    prof.enable()
    try:
        return _orig(args, m2, m1, dev, dt)
    finally:
        torch.cuda.synchronize()
        prof.disable()


bench.train_mode = wrapped
bench.main()
s = io.StringIO()
pstats.Stats(prof, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
