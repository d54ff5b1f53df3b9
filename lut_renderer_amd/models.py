"""Import-compatible name for the reference's `lut_renderer.models` (models.py:11-143)."""
from .params import ProcessingParams, Task, TaskStatus  # noqa: F401
