"""Mirror of the reference's `utils` package for the render hot path (utils.py, run_nerf_helpers.py)."""
