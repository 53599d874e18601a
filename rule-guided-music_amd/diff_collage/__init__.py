"""Drop-in `diff_collage` package (the part scripts/sample_rule.py uses): CondIndSimple, CondIndCircle,
split_wimg, avg_merge_wimg, SimpleWork.  The legacy workers of the reference (CondInd*SR, AvgLong, w_loss,
generic_sampler's Heun sampler) are never called by the sampling CLI and are out of scope."""
from .generic_sampler import SimpleWork
from .w_img import split_wimg, avg_merge_wimg
from .condind_long import CondIndSimple
from .condind_circle import CondIndCircle

__all__ = ["SimpleWork", "split_wimg", "avg_merge_wimg", "CondIndSimple", "CondIndCircle"]
