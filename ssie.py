"""Loader for the hyphen-named product package (registers it as module `ssie_amd`)."""
import importlib.util
import os
import sys

PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                       "self-supervised-image-enhancement-network-training-with-low-light-images-only_amd")


def load():
    if "ssie_amd" in sys.modules:
        return sys.modules["ssie_amd"]
    spec = importlib.util.spec_from_file_location(
        "ssie_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ssie_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
