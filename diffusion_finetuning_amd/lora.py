"""`lora_diffusion.lora` API surface on the MI355X-native hot path.

Every public name of the reference's lora_diffusion/lora.py resolves here.  The implementation is split by
concern: core.py (module, finders, injection, merge, monkeypatch family — the part that touches the HIP path),
formats.py (.pt / safetensors I/O) and pipeline.py (diffusers-pipeline conveniences).
"""
from .core import *  # noqa: F401,F403
from .core import _find_children, _find_modules, _find_modules_old, _find_modules_v2, _matches, _wrap_linear  # noqa: F401
from .formats import *  # noqa: F401,F403
from .formats import _text_lora_path, _ti_lora_path  # noqa: F401
from .pipeline import *  # noqa: F401,F403
