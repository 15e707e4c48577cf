"""Mirror of vgtk/vgtk/pc, index-based operators only (vgtk/vgtk/pc/sample.py:46-77) + load_ply."""
from .sample import group_nd, ball_query_index, furthest_sample_index, furthest_sample  # noqa: F401
from .io import load_ply  # noqa: F401
