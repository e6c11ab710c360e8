"""Measurement / diagnosis scripts (none of them is imported by the product)."""
