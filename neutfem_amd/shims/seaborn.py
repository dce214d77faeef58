"""Minimal stand-in so the reference benchmark drivers (tests/iaea2d/iaea2d.py:19 `import seaborn as sns`)
import in this image, where seaborn is not installed.  Only reached through plotting flags."""


def heatmap(*args, **kwargs):
    raise RuntimeError("seaborn is not installed in this image; plotting is disabled")
