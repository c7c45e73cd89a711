"""Import alias: the package directory is named `mdn-vae_amd`, which is not a Python
identifier, so `import mdnvae_amd` loads that directory as the package `mdnvae_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mdn-vae_amd")
_spec = importlib.util.spec_from_file_location("mdnvae_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mdnvae_amd"] = _mod
_spec.loader.exec_module(_mod)
