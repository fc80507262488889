"""Host-side mirror of the reference's `network/` package for the hot path."""
