"""TEST INFRASTRUCTURE — NOT A MEASUREMENT.  Runs bench.py's rank logic (self-spawn through torch.distributed.run, process group, per-rank shards, barrier +
max-over-ranks timing, one JSON line from rank 0) on CPU tensors, `gloo`, the host emulator build of the kernel sources and a two-block toy stack:
``python tests/bench_emu_harness.py --gpus 2 --steps 2 ...`` (tests/test_bench_spawn.py).  bench.py itself imports nothing from tests/."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402


def _setup():
    from deformablelka_amd import _lib
    from tests import emu
    _lib._set_backend_for_tests(emu.load())
    torch.set_num_threads(1)


HARNESS = {"device": torch.device("cpu"), "process_group": "gloo", "stages": ((32, (2, 2, 3), 1), (64, (2, 2, 2), 1)), "setup": _setup,
           "data": "synthetic (EMULATOR TEST RUN on CPU: exercises the rank logic only, NOT a measurement)"}

if __name__ == "__main__":
    bench.main(HARNESS)
