"""Mirror of the reference's `data.ray_utils` (the dataset classes are out of scope)."""
