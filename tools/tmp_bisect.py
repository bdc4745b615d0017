import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest
import edge_yolo_amd  # noqa
from edge_yolo_amd import _lib as L
for name in sys.argv[1:]:
    k, v = name.split("=")
    L.check(L.lib().ey_tune_set(k.encode(), int(v)), k)
sys.exit(pytest.main(["-q", "-x", "tests/test_gpu_dist.py", "-k", "behind_the_bench_pipeline"]))
