"""Mirror of the reference's `models` package (common blocks, Detect/Model, loader)."""
