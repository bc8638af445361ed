"""oracle."""
