"""Batch-job entry points (mirror of the reference's app/batch package for the bm25 job)."""
