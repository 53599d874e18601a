"""CondIndCircle: the circular variant (reference diff_collage/condind_circle.py:7-84) -- the first `overlap`
columns are appended at the end before splitting, and the wrapped seam of the merged eps is averaged."""
from .condind_long import CondIndSimple


class CondIndCircle(CondIndSimple):
    circle = True

    def _final_width(self, w, n):
        return w * n - self.overlap_size * n

    def circle_split(self, in_x):
        from .w_img import split_windows
        return split_windows(in_x, self.num_img, self.overlap_size)

    def circle_merge(self, xs, overlap_size=None):
        from .w_img import merge_windows
        ov = self.overlap_size if overlap_size is None else overlap_size
        return merge_windows(xs, None, ov, self.num_img, circle=True, is_avg=True)
