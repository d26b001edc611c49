"""Build-owned caller of the hot path (NOT part of the product package): an SD-shaped UNet whose LoRA targets
enumerate exactly like diffusers' UNet2DConditionModel does for the reference.  Used by bench.py, the parity tests,
__graft_entry__.smoke() and oracle/make_golden.py, because diffusers is not available offline."""
