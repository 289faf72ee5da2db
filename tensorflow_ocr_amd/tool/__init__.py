"""Host-side mirrors of the reference's `tool/` modules on the hot path."""
