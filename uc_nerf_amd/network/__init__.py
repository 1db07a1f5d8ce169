"""Mirror of the reference's `network` package for the render hot path (renderer.py, models.py)."""
