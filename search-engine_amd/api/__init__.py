"""Serving-side entry points (mirror of the reference's app/api package for the search path)."""
