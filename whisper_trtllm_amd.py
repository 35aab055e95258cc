"""Import alias: `import whisper_trtllm_amd` loads the package that lives in `whisper-trtllm_amd/`.

The package directory keeps the repository's name (a hyphen is not a legal Python identifier),
so this one-file loader registers it under an importable name.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "whisper-trtllm_amd")
_spec = importlib.util.spec_from_file_location(
    "whisper_trtllm_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["whisper_trtllm_amd"] = _mod
_spec.loader.exec_module(_mod)
