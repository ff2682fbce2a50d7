import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "circuitgen"), os.path.join(ROOT, "tools")):   # circuitgen: the stand-in circuit builder; tools: export_circuits (the exporter front end: tests may run it, the product may not)
    if p not in sys.path:
        sys.path.insert(0, p)


# PyTorch must be imported before the oracle's first OpenMP region: torch brings its own OpenMP runtime, and when it is loaded AFTER
# libgomp has already spun up the oracle's thread pool, every later oracle call on a many-core box runs ~7x slower (measured on the
# 256-thread GPU host: 23 s -> 160 s for the GPU suite).  The product library imports torch anyway (it shares torch's HIP runtime).
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is part of the image
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
