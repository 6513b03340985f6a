"""Test-only empty stand-in: the reference imports tensorflow_probability but never uses it."""
