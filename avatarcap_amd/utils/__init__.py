"""Host-side mirror of the reference's `utils/` functions that sit on the hot path."""
