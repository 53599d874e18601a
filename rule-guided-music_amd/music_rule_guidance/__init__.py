"""Drop-in `music_rule_guidance` package: the FUNC_DICT / LOSS_DICT plugin surface (rule_maps.py) with the
built-in rules running as HIP kernels.  User-added Python rules keep working: they receive the decoded
roll as a torch tensor exactly like in the reference."""
