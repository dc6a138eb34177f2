"""The two-line replacement modules of integration/ import under both of the reference's import styles
(src/train_lightgcn.py:2-3: script directory on the path; src/inference_lightgcn.py:2-3: repository root on the path;
torchserve/lightgcn_handler.py:5: the handler's own directory) and hand out the drop-in class."""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INTEGRATION = os.path.join(ROOT, "integration")


def _fresh_import(name, path_entry):
    for mod in [m for m in sys.modules if m == name or m.startswith(name + ".") or m in ("lightgcn", "src", "src.lightgcn")]:
        del sys.modules[mod]
    sys.path.insert(0, path_entry)
    try:
        return importlib.import_module(name)
    finally:
        sys.path.remove(path_entry)


@pytest.mark.parametrize("name,entry", [("lightgcn", os.path.join(INTEGRATION, "src")),          # train_lightgcn.py:3
                                        ("src.lightgcn", INTEGRATION),                            # inference_lightgcn.py:2
                                        ("lightgcn", os.path.join(INTEGRATION, "torchserve"))])    # lightgcn_handler.py:5
def test_replacement_module_imports_like_the_reference_file(name, entry):
    import gnn_ecommerce_amd as lg
    mod = _fresh_import(name, entry)
    try:
        assert os.path.dirname(os.path.abspath(mod.__file__)).startswith(INTEGRATION)
        assert mod.LightGCN is lg.LightGCN and mod.BPRLoss is lg.BPRLoss
        model = mod.LightGCN(12, 8, 2)                      # the reference's constructor (src/lightgcn.py:58-65)
        assert sorted(model.state_dict()) == ["alpha", "embedding.weight"]
    finally:
        for m in [m for m in sys.modules if m in ("lightgcn", "src", "src.lightgcn")]:
            del sys.modules[m]
