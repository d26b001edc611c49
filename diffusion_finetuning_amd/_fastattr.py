"""Sub-module parameters of a wrapped layer without `nn.Module.__getattr__`.

`layer.lora_down.weight` is two passes through `nn.Module.__getattr__` (a Python function that probes `_parameters`,
`_buffers`, `_modules` in turn): ≈0.75 µs per hop, and the unchanged-trainer route made ≈20 such hops per wrapped-layer call and
≈10 per group member per validity check — 2–3 ms of a 39 ms step (profiles/r06_dropin_sections_before.log).  The registries
nn.Module keeps are plain dicts: reading them directly is the same lookup `__getattr__` ends in, an order of magnitude cheaper,
and sees the same object when a Parameter or a sub-module has been re-assigned.  Anything unusual (a parametrized weight, a
tensor set as a plain attribute) falls back to the attribute walk.
"""


def factor_weights(layer):
    """`(layer.lora_down.weight, layer.lora_up.weight)` of a LoraInjectedLinear (lora_diffusion/lora.py:43-44)."""
    try:
        mods = layer._modules
        return mods["lora_down"]._parameters["weight"], mods["lora_up"]._parameters["weight"]
    except (KeyError, AttributeError):
        return layer.lora_down.weight, layer.lora_up.weight


def linear_params(lin):
    """`(lin.weight, lin.bias)` of an nn.Linear."""
    try:
        params = lin._parameters
        return params["weight"], params["bias"]
    except (KeyError, AttributeError):
        return lin.weight, lin.bias


def frozen_linear(layer):
    """`(layer.linear.weight, layer.linear.bias)` — the frozen nn.Linear a LoraInjectedLinear wraps (lora.py:42)."""
    try:
        params = layer._modules["linear"]._parameters
        return params["weight"], params["bias"]
    except (KeyError, AttributeError):
        lin = layer.linear
        return lin.weight, lin.bias
