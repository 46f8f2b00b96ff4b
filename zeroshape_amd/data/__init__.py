"""Datasets with the reference's sample-dict keys (data/synthetic.py:126-176)."""
