import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "lab: exercises superseded / rejected kernels that only the lab build of the library carries "
                                       "(F5_LAB=1 bash f5_tts_mlx_amd/csrc/build.sh); skipped on the product library")


def _lab_library() -> bool:
    try:
        from f5_tts_mlx_amd import engine as E
        return bool(E.load_library().f5_lab_build())
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    import torch
    if not _lab_library():
        skip_lab = pytest.mark.skip(reason="product library loaded: the experiment kernels exist in the lab build only (F5_LAB=1)")
        for item in items:
            if "lab" in item.keywords:
                item.add_marker(skip_lab)
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
