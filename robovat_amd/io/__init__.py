"""Episode generation loop (host side)."""
