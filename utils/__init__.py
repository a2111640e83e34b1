"""Drop-in for the reference package `utils` (compute functions of utils/tools.py only)."""
