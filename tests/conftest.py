import os
import sys

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.abspath(ROOT))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs /root/reference (authoring container only)")


def pytest_collection_modifyitems(config, items):
    import torch
    has_gpu = torch.cuda.is_available()
    skip_gpu = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords and not has_gpu:
            item.add_marker(skip_gpu)


def pytest_terminal_summary(terminalreporter):
    """what the parity helpers observed (tests/scenes.py: PARITY_LOG), whatever the capture mode: per frame the largest pixel
    error against the oracle and the number of pixels that needed the threshold-adjacent exception"""
    try:
        import scenes
    except Exception:
        return
    if scenes.PARITY_LOG:
        n_exc = sum(int(line.rsplit("= ", 1)[1]) for line in scenes.PARITY_LOG)
        terminalreporter.write_sep("-", f"image parity: {len(scenes.PARITY_LOG)} frames against the oracle, "
                                        f"{n_exc} threshold-adjacent exception pixels in total")
        for line in scenes.PARITY_LOG[:200]:
            terminalreporter.write_line("[parity] " + line)
