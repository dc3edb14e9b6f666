import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Order of the suite: parity evidence first, multi-process rehearsals last, so that `-x` on a fragile rehearsal can
# never hide a kernel / step parity result.  Files not listed keep their alphabetical place between the two groups.
_FIRST = ["test_kernels_gpu.py", "test_step_gpu.py", "test_configs_gpu.py", "test_properties_gpu.py",
          "test_end_to_end_gpu.py", "test_data_feed.py"]
_LAST = ["test_bench_entry.py", "test_dist_gloo.py", "test_dist_gpu.py"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.basename(str(item.fspath))
        if name in _FIRST:
            return _FIRST.index(name)
        if name in _LAST:
            return 1000 + _LAST.index(name)
        return 500

    items.sort(key=rank)  # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def release_captured_programs(request):
    """Engines, their plans and the closures of a plan reference each other: without a collection the captured graphs of
    every earlier test (each with the runtime-internal streams of its forked branches) stay alive until the cyclic
    collector happens to run.  Dozens of live multi-stream graph executables made later launches crash inside the
    runtime on some boxes (null stream, tools/debug/seg_hunt.sh); a training process holds a handful."""
    yield
    if request.node.get_closest_marker("gpu") is None or os.environ.get("MMVAE_TEST_NO_GC", "0") != "0":
        return
    import gc

    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        gc.collect()
