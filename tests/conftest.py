import os
import sys

# The golden vectors were produced with OMP_NUM_THREADS=8 (the reference's float32 column statistics are
# summed per thread, math_ops.cpp:255-300, so their last bit depends on the thread count).  Pin it before
# any OpenMP runtime starts.
os.environ.setdefault("OMP_NUM_THREADS", "8")
# Every categorical step of the test suite replays the reference's string-keyed candidate container beside the engine's
# hash-replay of it and throws when the two iteration orders differ (engine_step.hip, device_categorical_candidates).
os.environ.setdefault("GBRL_HIP_CAT_CHECK", "1")
# Every root level that takes its class counts from the radix selection's ranks (k_hist_build without the count atomic) also accumulates
# them and compares entry by entry; a difference raises (engine_step.hip, grow_tree, root_mode).
os.environ.setdefault("GBRL_HIP_ROOT_COUNTS", "2")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
