"""Import shim: the package directory is named `edge-yolo_amd` (not an identifier); this module loads it under
the importable name `edge_yolo_amd`, sub-packages included (`import edge_yolo_amd.nn.tasks` works)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "edge-yolo_amd")
_spec = importlib.util.spec_from_file_location("edge_yolo_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["edge_yolo_amd"] = _mod
_spec.loader.exec_module(_mod)
